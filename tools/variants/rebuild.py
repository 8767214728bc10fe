#!/usr/bin/env python3
"""Rebuild the full source of a measured-and-dropped kernel variant from its diff:  python tools/variants/rebuild.py <name> [...]
Each tools/variants/<name>.diff names the file and the commit it was cut from (`# base:`); the variant is that blob with the
diff applied, written to tools/variants/_build/<name>.hip (untracked).  Build and link as tools/variants/README.md says."""
import os, re, subprocess, sys
here = os.path.dirname(os.path.abspath(__file__))
root = os.path.dirname(os.path.dirname(here))
os.makedirs(os.path.join(here, "_build"), exist_ok=True)
for name in sys.argv[1:] or sorted(f[:-5] for f in os.listdir(here) if f.endswith(".diff")):
    d = os.path.join(here, name + ".diff")
    m = re.search(r"^# base:\s+(\S+) at commit ([0-9a-f]+)", open(d).read(), re.M)
    out = os.path.join(here, "_build", name + ".hip")
    with open(out, "wb") as f:
        f.write(subprocess.check_output(["git", "-C", root, "show", f"{m.group(2)}:{m.group(1)}"]))
    subprocess.check_call(["patch", "-s", out, d])
    print(f"{name}: {out}  (base {m.group(1)} @ {m.group(2)[:12]})")
