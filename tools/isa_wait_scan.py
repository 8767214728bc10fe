#!/usr/bin/env python3
"""Which kernels wait for every global load right where they issue it?

    python tools/isa_wait_scan.py [source.hip ...]        (default: every source of posetraj_amd/csrc)

Compiles each source to gfx950 assembly (hipcc -S, device only; no GPU needed) and counts, per kernel, the global / buffer loads
and the `s_waitcnt vmcnt(0)` that follow a load within three instructions.  A kernel whose loads are nearly all followed by a
full wait has no load in flight under its arithmetic: branches around the loads (alignment tests, scalar tails) or a store that
may alias the next load are the usual cause - `pt_gemm_f16`'s loader and `+=` epilogue stood at 176 of 208 and lost 29 % to it
(profiles/r04/gemm_loader_ab.txt).  A hint, not a verdict: a wait directly behind the last load of a batch is normal."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "posetraj_amd", "csrc")


def scan(src: str):
    with tempfile.TemporaryDirectory() as tmp:
        out = os.path.join(tmp, "k.s")
        r = subprocess.run([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-S",
                            "--cuda-device-only", src, "-o", out], stderr=subprocess.PIPE, text=True)
        if r.returncode != 0:
            raise RuntimeError(r.stderr[-2000:])
        lines = open(out).read().splitlines()
    stats, name, idx, last = {}, None, 0, -99
    for ln in lines:
        m = re.match(r"^(_Z\w+):", ln)
        if m:
            name, idx, last = m.group(1), 0, -99
            stats[name] = [0, 0]
            continue
        t = ln.strip()
        if name is None or not t or t[0] in ".;":
            continue
        idx += 1
        if t.startswith(("global_load", "buffer_load", "flat_load")):
            stats[name][0] += 1
            last = idx
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and idx - last <= 3:
            stats[name][1] += 1
        elif t.startswith("s_endpgm"):
            name = None
    return stats


def short(mangled: str) -> str:
    """`gemm_kernel<4, 4, 1>` from the mangled name (binutils' c++filt here does not know _Float16)."""
    m = re.match(r"_ZN12_GLOBAL__N_1(\d+)", mangled) or re.match(r"_Z(\d+)", mangled)
    if not m:
        return mangled
    n, at = int(m.group(1)), m.end()
    name, rest = mangled[at:at + n], mangled[at + n:]
    if rest.startswith("I"):
        args = re.findall(r"L[ib](\d+)E", rest.split("EE")[0] + "E")
        if args:
            name += "<" + ", ".join(args) + ">"
    return name


def main():
    srcs = sys.argv[1:] or sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".hip"))
    print(f"{'source':18s} {'loads':>6s} {'waited':>7s}  kernel")
    for src in srcs:
        st = scan(src)
        for k, (loads, waited) in sorted(st.items(), key=lambda kv: -kv[1][1]):
            if loads >= 3 and 2 * waited >= loads:
                print(f"{os.path.basename(src):18s} {loads:6d} {waited:7d}  {short(k)}")


if __name__ == "__main__":
    main()
