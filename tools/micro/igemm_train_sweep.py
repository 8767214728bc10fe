"""How much of the training step's pt_igemm_f16 time is the tile choice?  Every shape of one step (profiles/r05/
train_step_igemm_shapes_r05v.txt: M, N, K, kernel, count) is timed alone under the automatic choice and under each forced tile
configuration (0: 256 x 256, 1: 128 x 320, 2: 128 x 128, 3: 256 x 320 (+ split-K), 4: 128 x 160); printed: sum over the step of
count x time for the automatic choice and for the per-shape best.
    python tools/micro/igemm_train_sweep.py [table]"""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from posetraj_amd import hip, ops
from posetraj_amd.packing import pack_conv2d, pack_linear
dev = torch.device("cuda:0")
L = hip.lib()
g = torch.Generator().manual_seed(0)
table = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles/r05/train_step_igemm_shapes_r05v.txt")
rows = []
for ln in open(table):
    f = ln.split()
    if len(f) == 13 and "x" in f[3] and f[0].isdigit():
        M, N, K = int(f[0]), int(f[1]), int(f[2]); kh, kw = (int(v) for v in f[3].split("x"))
        st, up, act, n = int(f[4]), int(f[5]), int(f[7]), int(f[9])
        if st == 1 and up == 0 and M * N >= 4096:
            rows.append((M, N, K, kh, kw, act, n, float(f[10])))
tot_auto = tot_best = tot_tab = 0.0
geoms = {40320: (14, 40, 72), 10080: (14, 20, 36), 2520: (14, 10, 18), 630: (14, 5, 9), 2880: (1, 40, 72), 720: (1, 20, 36), 180: (1, 10, 18), 45: (1, 5, 9), 2580480: (14, 320, 576),
         645120: (14, 160, 288), 161280: (14, 80, 144)}
for M, N, K, kh, kw, act, n, tab_ms in rows:
    C = K // (kh * kw)
    if kh * kw > 1 and M not in geoms:
        continue
    Nw = N * 2 if act == 1 else N
    if kh * kw == 1:
        pw = pack_linear(torch.randn(N, C, generator=g) * C ** -0.5, torch.zeros(N), dev, geglu=(act == 1))
        x = torch.randn(M, C, device=dev).half(); geom = None
    else:
        if (kh, kw) == (3, 3):
            pw = pack_conv2d(torch.randn(N, C, 3, 3, generator=g) * (9 * C) ** -0.5, torch.zeros(N), dev)
        else:
            continue                                         # the 3 x 1 temporal convolutions keep the automatic choice here
        nimg, h, w = geoms[M]
        x = torch.randn(nimg, h, w, C, device=dev).half(); geom = (nimg, h, w)
    res = {}
    for cfg in (-1, 0, 1, 2, 3, 4):
        L.pt_igemm_force_config(cfg)
        try:
            for _ in range(2): ops.igemm(x, pw, geom=geom)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(10): ops.igemm(x, pw, geom=geom)
            e1.record(); torch.cuda.synchronize()
            res[cfg] = e0.elapsed_time(e1) / 10 * 1e3
        except RuntimeError:
            pass
    L.pt_igemm_force_config(-1)
    best = min(res, key=res.get)
    tot_auto += n * res[-1]; tot_best += n * res[best]; tot_tab += tab_ms * 1e3
    flag = "" if res[best] > 0.93 * res[-1] else f"   <- cfg {best} {100 * (1 - res[best] / res[-1]):.0f} % faster, {n * (res[-1] - res[best]):.0f} us per step"
    print(f"{M:8d} {N:6d} {K:6d} {kh}x{kw} a{act} n={n:3d}  auto {res[-1]:7.1f} us  " + "  ".join(f"{c}:{t:7.1f}" for c, t in res.items() if c >= 0) + flag)
print(f"# per step: in the step's own table {tot_tab / 1e3:.2f} ms; alone, automatic choice {tot_auto / 1e3:.2f} ms; per-shape best {tot_best / 1e3:.2f} ms")
