#!/usr/bin/env python3
"""Matrix-pipe utilisation of the SHIPPED kernels from one real denoise iteration.

    rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d out -o p \
        -- python3 tools/traffic_run.py out/shapes.json [L|M]
    python3 tools/mfma_busy.py out [> profiles/rNN/mfma_busy_<workload>_<tag>.txt]

tools/traffic_run.py records the shape of every igemm-family launch of the iteration in launch order; the last len(shapes)
igemm-family dispatches of the process are that iteration's, in the same order (the join tools/traffic_extract.py makes).
MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128): GRBM_GUI_ACTIVE is the sum over the 8 XCDs and a chip has
1 024 SIMDs (tools/pmc_summary.py, MI355X_MICROARCH.md, PMC units).  Two tables: every kernel symbol of the iteration with its
share of the device time, and the igemm family per shape beside the TFLOP/s the same dispatches ran at under the profiler."""
import collections, csv, glob, json, re, sys

d = sys.argv[1]
meta = json.load(open(f"{d}/shapes.json"))
S = meta["shapes"]
disp = collections.OrderedDict()                 # dispatch id -> [kernel, ns, {counter: value}]
paths = glob.glob(f"{d}/**/*counter_collection.csv", recursive=True)
assert paths, f"no counter_collection.csv under {d}"
for path in paths:
    for r in csv.DictReader(open(path)):
        e = disp.setdefault(int(r["Dispatch_Id"]), [r["Kernel_Name"], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]), {}])
        e[2][r["Counter_Name"]] = e[2].get(r["Counter_Name"], 0.0) + float(r["Counter_Value"])
rows = [disp[k] for k in sorted(disp)]


def family(n):
    m = re.search(r"igemm_kernel<[^>]*Cfg<([^>]*)>, (true|false)", n)
    if m:
        return "igemm<" + m.group(1).replace(" ", "") + ">"
    for k in ("igemm8_kernel", "igemm10_kernel", "ffn320_kernel", "lnlin320_kernel"):
        if k in n:
            return k
    return None


ig = [r for r in rows if family(r[0])]
n = len(S)
assert len(ig) >= n, (len(ig), n)
first = len(rows) - 1
cnt = 0
for i in range(len(rows) - 1, -1, -1):           # the iteration = everything from its first igemm-family dispatch on
    if family(rows[i][0]):
        cnt += 1
        if cnt == n:
            first = i
            break
it = rows[first:]
busy = lambda c: 100.0 * c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) / (c["GRBM_GUI_ACTIVE"] * 128) if c.get("GRBM_GUI_ACTIVE") else float("nan")


def short(name):
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)[:78]


print(f"# matrix-pipe utilisation, workload {meta['workload']}, one denoise iteration under rocprofv3 --pmc (build {meta['csrc_sha256'][:12]}); "
      f"MFMA busy = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE x 128)")
agg = collections.OrderedDict()
tot_ns = sum(r[1] for r in it)
for name, ns, c in it:
    e = agg.setdefault(short(name), [0, 0, collections.Counter()])
    e[0] += 1; e[1] += ns; e[2].update(c)
print(f"{'kernel':78s} {'n':>4} {'ms':>8} {'%time':>6} {'MFMA busy %':>11} {'GHz':>5}")
for k, (cnt_, ns, c) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
    if ns < 0.002 * tot_ns:
        continue
    ghz = c["GRBM_GUI_ACTIVE"] / 8 / ns if ns else 0.0
    print(f"{k:78s} {cnt_:4d} {ns / 1e6:8.3f} {100 * ns / tot_ns:6.2f} {busy(c):11.1f} {ghz:5.2f}")
allc = collections.Counter()
for _, _, c in it:
    allc.update(c)
print(f"{'WHOLE ITERATION (sum of kernel times, eager, under the profiler)':78s} {len(it):4d} {tot_ns / 1e6:8.3f} {100.0:6.2f} {busy(allc):11.1f}")

print(f"\n# igemm family per shape (columns as tools/shape_report.py; a = 3: fused GEGLU feed-forward, N = C, K = inner)")
print(f"{'M':>7} {'N':>6} {'K':>6} k s u {'C1':>5} a {'e':>2} {'kernel':22s} {'n':>3} {'ms':>8} {'TFLOP/s':>8} {'MFMA busy %':>11}")
per = collections.OrderedDict()
fl_tot = 0.0
for s, (name, ns, c) in zip(S, ig[-n:]):
    M, N, K, KH, KW, st, up, C1, act, epi = s
    fl = 2.0 * M * N * K * (3 if act == 3 else 1)                                   # fused feed-forward: [M,C]x[C,2I] + [M,I]x[I,C] = 3 x 2MCI
    if act == 3 and epi & 32:
        fl += 2.0 * M * N * N                                                        # + the attention out-projection of its prologue
    e = per.setdefault(tuple(s) + (family(name),), [0, 0, collections.Counter(), 0.0])
    e[0] += 1; e[1] += ns; e[2].update(c); e[3] += fl
    fl_tot += fl
for k, (cnt_, ns, c, fl) in sorted(per.items(), key=lambda kv: -kv[1][1])[:48]:
    M, N, K, KH, KW, st, up, C1, act, epi, fam = k
    print(f"{M:7d} {N:6d} {K:6d} {KH}x{KW} {st} {up} {C1:5d} {act} {epi:2d} {fam:22s} {cnt_:3d} {ns / 1e6:8.3f} {fl / ns / 1e3:8.1f} {busy(c):11.1f}")
igc = collections.Counter(); ig_ns = 0
for name, ns, c in ig[-n:]:
    igc.update(c); ig_ns += ns
print(f"# igemm family: {n} launches, {ig_ns / 1e6:.2f} ms, {fl_tot / ig_ns / 1e3:.1f} TFLOP/s, MFMA busy {busy(igc):.1f} %  "
      f"(a 16x16x32 f16 MFMA holds the pipe 8 cycles: 2 500 TFLOP/s at 100 % and 2.4 GHz)")
