#!/usr/bin/env python3
"""Samples the GPU's socket power and shader clock (hwmon / pp_dpm_sclk via sysfs, amd-smi as a fallback) while a command
runs:   python tools/power_trace.py -- python bench.py --no-cpu-baseline --no-profile --steps 1 --warmup 1
Prints min / median / max of both over the samples taken while the command's GPU phase ran (power above 40 % of max seen)."""
import glob, subprocess, sys, time, statistics


def rd(p):
    try:
        return float(open(p).read().strip())
    except (OSError, ValueError):
        return None


HW = sorted(glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*"))


def sample():
    out = []
    for h in HW:
        pw = rd(h + "/power1_average")
        if pw is None:
            pw = rd(h + "/power1_input")
        out.append((pw / 1e6 if pw is not None else None, (rd(h + "/freq1_input") or 0) / 1e6, (rd(h + "/power1_cap") or 0) / 1e6))
    return out


cmd = sys.argv[sys.argv.index("--") + 1:]
proc = subprocess.Popen(cmd)
rows = []
while proc.poll() is None:
    rows.append(sample())
    time.sleep(0.05)
if not rows or not HW:
    print("power_trace: no hwmon found")
    sys.exit(proc.returncode)
# the device the command used = the card whose power moved most (the box shows every GPU of the node)
rng = []
for i in range(len(HW)):
    pw = [r[i][0] for r in rows if r[i][0] is not None]
    rng.append((max(pw) - min(pw)) if pw else -1)
i = rng.index(max(rng))
pw = [r[i][0] for r in rows if r[i][0] is not None]
hi = [r[i] for r in rows if r[i][0] is not None and r[i][0] > min(pw) + 0.5 * (max(pw) - min(pw))]
p = [r[0] for r in hi]; f = [r[1] for r in hi]
print(f"power_trace: {len(rows)} samples x {len(HW)} cards; busiest card {HW[i].split('/')[4]} (power range {rng[i]:.0f} W), cap {rows[0][i][2]:.0f} W, "
      f"{len(hi)} samples under load")
print(f"  socket power under load: min {min(p):.0f}  median {statistics.median(p):.0f}  max {max(p):.0f} W   (idle {min(pw):.0f} W)")
print(f"  shader clock under load (freq1_input): min {min(f):.0f}  median {statistics.median(f):.0f}  max {max(f):.0f} MHz")
sys.exit(proc.returncode)
