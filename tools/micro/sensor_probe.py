#!/usr/bin/env python3
"""What the box's power sensors and the in-kernel clock probe report over time under a steady GEMM load (diagnostic for
tools/energy_table.py): lists the card's hwmon files, then 12 s of back-to-back pt_igemm_f16 launches (16128 x 1280 x 11520)
with the power file(s) sampled every 20 ms and the clock probe every 1 ms; prints 0.5-s averages."""
import glob, os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
from posetraj_amd import ops
from posetraj_amd.packing import pack_linear
import energy_table as ET
dev = torch.device("cuda:0")
pr = torch.cuda.get_device_properties(0)
bdf = f"{pr.pci_domain_id:04x}:{pr.pci_bus_id:02x}:{pr.pci_device_id:02x}."
mine = [h for h in glob.glob("/sys/class/drm/card*/device/hwmon/hwmon*") if bdf in os.path.realpath(h.split("/hwmon/")[0])]
print("hwmon dirs of this card:", mine)
files = []
for h in mine:
    for f in sorted(os.listdir(h)):
        p = os.path.join(h, f)
        if os.path.isfile(p) and (f.startswith("power") or f.startswith("energy") or f.startswith("freq")):
            try:
                print(f"  {f} = {open(p).read().strip()}")
                if f.startswith("power1_") and f.split("_")[1] in ("input", "average"):
                    files.append(p)
            except OSError as e:
                print(f"  {f}: {e}")
dv = os.path.realpath(mine[0].split("/hwmon/")[0]) if mine else None
if dv:
    for f in ("pp_dpm_sclk", "gpu_busy_percent"):
        try:
            print(f"  {f}: " + open(os.path.join(dv, f)).read().strip().replace("\n", " | "))
        except OSError as e:
            print(f"  {f}: {e}")
g = torch.Generator().manual_seed(0)
M, N, K = 16128, 1280, 11520
x = (torch.randn(M, K, generator=g)).half().to(dev)
pw = pack_linear((torch.randn(N, K, generator=g) * K ** -0.5).half(), None, dev)
out = torch.empty(M, N, dtype=torch.float16, device=dev)
rows, stop = [], threading.Event()


def sampler():
    while not stop.is_set():
        rows.append((time.perf_counter(), [float(open(f).read()) / 1e6 for f in files]))
        stop.wait(0.02)


th = threading.Thread(target=sampler, daemon=True); th.start()
time.sleep(1.0)
clock = ET.Clock(13.0); clock.start()
t0 = time.perf_counter()
n = 0
while time.perf_counter() - t0 < 12.0:
    for _ in range(50):
        ops.igemm(x, pw, out=out)
    n += 50
    torch.cuda.current_stream().synchronize()                  # (a device-wide synchronize would wait for the probe kernel)
t1 = time.perf_counter()
print(f"{n} launches in {t1 - t0:.2f} s: {2.0 * M * N * K * n / (t1 - t0) / 1e12:.0f} TFLOP/s")
time.sleep(2.0)
stop.set()
clock.stop[0] = 1; clock.stream.synchronize()
v = clock.buf.cpu().view(-1, 2); v = v[v[:, 1] > 0]
ct, rt = v[:, 0].double(), v[:, 1].double()
print("files:", [os.path.basename(f) for f in files])
print("   t(s)   power(W) per file          in-kernel GHz (0.5 s mean)")
for k in range(-2, 28):
    a, b = t0 + 0.5 * k, t0 + 0.5 * (k + 1)
    sel = [r[1] for r in rows if a <= r[0] < b]
    if not sel:
        continue
    pw_ = [sum(s[i] for s in sel) / len(sel) for i in range(len(files))]
    # clock samples: realtime ticks are 100 MHz; align the probe's first sample with its launch (~t0)
    ra, rb = rt[0] + (0.5 * k) * 1e8, rt[0] + (0.5 * (k + 1)) * 1e8
    m = (rt >= ra) & (rt < rb)
    ghz = float((ct[m][-1] - ct[m][0]) / (rt[m][-1] - rt[m][0]) * 0.1) if int(m.sum()) > 2 else float("nan")
    print(f"{0.5 * k:7.1f}   " + "  ".join(f"{p:7.0f}" for p in pw_) + f"      {ghz:.3f}")
