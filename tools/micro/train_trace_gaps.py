"""Device idle time inside one training step, from a rocprofv3 kernel trace:

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 tools/train_step_bench.py --json --steps 4 --warmup 2
    python tools/micro/train_trace_gaps.py gpurun_out/trace

One step = from the end of one pt_adamw_f32 launch to the end of the next.  Prints the union of the kernels' busy intervals
(all streams), the idle remainder split over ten equal windows of the step, and the sum of kernel durations (> union when
streams overlap)."""
import csv, glob, os, sys

root = sys.argv[1]
files = glob.glob(os.path.join(root, "**", "*kernel_trace.csv"), recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
ends = [e for s, e, n in rows if "adamw" in n]
print(f"{len(rows)} kernel records, {len(ends)} optimizer launches")
for a, b in list(zip(ends[:-1], ends[1:]))[-3:]:
    ks = [(s, e, n) for s, e, n in rows if s >= a and e <= b]
    span = (b - a) / 1e6
    busy, cur_s, cur_e = 0, None, None
    W = 10
    idle_w = [0.0] * W
    def add_idle(x0, x1):
        for w in range(W):
            lo, hi = a + (b - a) * w / W, a + (b - a) * (w + 1) / W
            ov = min(x1, hi) - max(x0, lo)
            if ov > 0:
                idle_w[w] += ov / 1e6
    last = a
    gaps = 0
    for s, e, n in ks:
        if s > last:
            add_idle(last, s)
            gaps += (s - last) > 5000
        last = max(last, e)
    if b > last:
        add_idle(last, b)
    total = sum((e - s) for s, e, n in ks) / 1e6
    idle = sum(idle_w)
    print(f"step {span:7.2f} ms: {len(ks)} kernels, sum of durations {total:7.2f} ms, device busy (union) {span - idle:7.2f} ms, idle {idle:6.2f} ms in {gaps} gaps > 5 us")
    print("   idle ms per tenth of the step: " + " ".join(f"{v:5.2f}" for v in idle_w))
    conc = 0.0
    ev = sorted([(s, 1) for s, e, n in ks] + [(e, -1) for s, e, n in ks])
    depth, prev, hist = 0, a, {}
    for t, d in ev:
        hist[depth] = hist.get(depth, 0) + (t - prev)
        depth += d; prev = t
    print("   time with k kernels in flight: " + " ".join(f"{k}:{v / 1e6:.1f}" for k, v in sorted(hist.items())))

# the last complete step by kernel name
a, b = ends[-2], ends[-1]
agg = {}
for s_, e_, n in rows:
    if s_ >= a and e_ <= b:
        key = n.replace("(anonymous namespace)::", "").replace("_GLOBAL__N_1", "")[:110]
        v = agg.setdefault(key, [0, 0])
        v[0] += 1; v[1] += e_ - s_
print("   last step by kernel (launches, ms):")
for k, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:int(sys.argv[2]) if len(sys.argv) > 2 else 40]:
    print(f"   {c:5d} {t / 1e6:8.2f}  {k}")
