#!/usr/bin/env python3
"""Joins the FETCH_SIZE / WRITE_SIZE passes of tools/traffic_run.py with the recorded launch shapes.
    python3 tools/traffic_summary.py <fetch dir> <write dir> <out prefix>
Writes <out prefix>.json (per-launch averages, read by bench.py for roofline.traffic) and prints a per-shape table.
FETCH_SIZE / WRITE_SIZE are in KiB; FETCH_SIZE is doubled (gfx950 tallies 128-B requests at 64 B: MI355X_MICROARCH.md, HBM)."""
import collections, json, os, sys
fd, wd, out = sys.argv[1:4]
F = json.load(open(f"{fd}/igemm_dispatches.json")); W = json.load(open(f"{wd}/igemm_dispatches.json"))
meta = json.load(open(f"{fd}/shapes.json")); S = meta["shapes"]
wmeta = json.load(open(f"{wd}/shapes.json"))
# the digest is stamped when the counters are collected (tools/traffic_run.py); both passes must come from the same build
digest = meta.get("csrc_sha256")
if digest is None or digest != wmeta.get("csrc_sha256"):
    raise SystemExit("FETCH and WRITE passes carry no / different csrc_sha256: re-collect both with the current tools/traffic_run.py")
n = len(S); F = F[-n:]; W = W[-n:]
agg = collections.OrderedDict()
tot = dict(fetch=0.0, write=0.0, alg_rd=0.0, alg_wr=0.0, ns=0.0)
for s, f, w in zip(S, F, W):
    M, N, K, KH, KW, st, up, C1, act, epi = s
    Cin = K // (KH * KW); nout = N // 2 if act == 1 else N
    in_rows = M // 4 if up else M * st * st
    alg_rd = in_rows * Cin * 2 + N * K * 2 + (M * nout * 2 if epi & 1 else 0) + (M * nout * 2 if epi & 4 else 0) + (M * nout * 2 if epi & 8 else 0)
    if act == 3:                                             # fused feed-forward (N = C, K = inner): x [M, C] + W1 [2 inner, C] + W2 [C, inner] + side inputs
        alg_rd = M * N * 2 + 2 * K * N * 2 + N * K * 2 + (M * N * 2 if epi & 1 else 0) + (M * N * 2 if epi & 4 else 0)
        if epi & 32:                                         # pre=: x is the attention output; + the projection's residual and weights
            alg_rd += M * N * 2 + N * N * 2
    if act == 4:                                             # LayerNorm + linear in one launch: x [M, K] + W [N, K] + gamma, beta
        alg_rd = M * K * 2 + N * K * 2 + 4 * K
    alg_wr = M * nout * 2 * (2 if epi & 16 else 1)           # wide-stream outputs are fp16 pairs
    e = agg.setdefault(tuple(s) + (f[0],), [0, 0.0, 0.0, 0.0, 0.0, 0.0])
    e[0] += 1; e[1] += 2 * f[2] * 1024; e[2] += w[2] * 1024; e[3] += f[3]; e[4] += alg_rd; e[5] += alg_wr
    tot["fetch"] += 2 * f[2] * 1024; tot["write"] += w[2] * 1024; tot["alg_rd"] += alg_rd; tot["alg_wr"] += alg_wr; tot["ns"] += f[3]
summary = {"workload": meta["workload"], "launches": n, "csrc_sha256": digest,                   # the build these counters were read on (stamped at collection)
           "hbm_bytes_per_launch": (tot["fetch"] + tot["write"]) / n,
           "fetch_bytes_per_launch": tot["fetch"] / n, "write_bytes_per_launch": tot["write"] / n,
           "algorithmic_bytes_per_launch": (tot["alg_rd"] + tot["alg_wr"]) / n,
           "note": "one denoise iteration; rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; FETCH_SIZE x2 (gfx950); "
                   "Infinity-Cache hits are counted as fetches, so this is an upper bound of true HBM traffic"}
json.dump(summary, open(out + ".json", "w"), indent=1)
print(f"# total fetch {tot['fetch'] / 1e9:.1f} GB (algorithmic reads {tot['alg_rd'] / 1e9:.1f}), write {tot['write'] / 1e9:.1f} GB "
      f"(algorithmic {tot['alg_wr'] / 1e9:.1f}); per launch {summary['hbm_bytes_per_launch'] / 1e6:.0f} MB vs algorithmic "
      f"{summary['algorithmic_bytes_per_launch'] / 1e6:.0f} MB; igemm time under the profiler {tot['ns'] / 1e6:.1f} ms")
print(f"{'M':>7} {'N':>6} {'K':>6} k s u {'C1':>5} a e {'cfg':16s} {'n':>3} {'ms':>8} {'fetchGB':>8} {'algRdGB':>8} {'ratio':>6} {'wrGB':>7} {'algWr':>7} {'TB/s':>5}")
for k, e in sorted(agg.items(), key=lambda kv: -kv[1][3])[:60]:
    M, N, K, KH, KW, st, up, C1, act, epi, cfg = k
    print(f"{M:7d} {N:6d} {K:6d} {KH}x{KW} {st} {up} {C1:5d} {act} {epi} {cfg:16s} {e[0]:3d} {e[3] / 1e6:8.2f} {e[1] / 1e9:8.2f} "
          f"{e[4] / 1e9:8.2f} {e[1] / e[4]:6.2f} {e[2] / 1e9:7.2f} {e[5] / 1e9:7.2f} {(e[1] + e[2]) / e[3] / 1e3:5.2f}")
