set -u
O=$(pwd)/gpurun_out/r06c; mkdir -p $O
R=$(pwd)
python tools/igemm_cfg_sweep.py > $O/igemm_cfg_sweep_L.txt 2>&1
SWEEP_WORKLOAD=M python tools/igemm_cfg_sweep.py > $O/igemm_cfg_sweep_M.txt 2>&1
python tools/energy_table.py --only "conv (3x3 / 3x1) L2" --no-probe > $O/energy_convL2_noprobe.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_convL2 -o p -- python3 $R/tools/energy_table.py --only "conv (3x3 / 3x1) L2" --seconds 1.0 > $O/prof_convL2.log 2>&1
cd $R
python - <<'PY' > gpurun_out/r06c/convL2_trace_summary.txt 2>&1
import csv, glob, collections
f = glob.glob("gpurun_out/r06c/prof_convL2/**/*kernel_trace.csv", recursive=True)[0]
rows = [r for r in csv.DictReader(open(f))]
ig = [r for r in rows if "igemm" in r["Kernel_Name"]]
tail = ig[-24 * 20:]                                   # the last 20 passes of the family replay
d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in tail]
gaps = [int(b["Start_Timestamp"]) - int(a["End_Timestamp"]) for a, b in zip(tail[:-1], tail[1:])]
span = int(tail[-1]["End_Timestamp"]) - int(tail[0]["Start_Timestamp"])
print(f"last {len(tail)} igemm dispatches of the replay: kernel time {sum(d) / 1e6:.2f} ms, span {span / 1e6:.2f} ms, "
      f"per pass of 24: kernels {sum(d) / 20 / 1e6:.3f} ms, span {span / 20 / 1e6:.3f} ms; gaps: mean {sum(gaps) / len(gaps) / 1e3:.1f} us, max {max(gaps) / 1e3:.1f} us")
per = collections.OrderedDict()
for r, dd in zip(tail, d):
    k = (r["Kernel_Name"][:60], r["Grid_Size"] if "Grid_Size" in r else "")
    e = per.setdefault(k, [0, 0]); e[0] += 1; e[1] += dd
for k, (n, t) in per.items():
    print(f"  {k[0]:60s} grid {k[1]:>8} n {n:4d} avg {t / n / 1e3:8.1f} us")
PY
find $O -name '*kernel_trace.csv' -delete; find $O -name '*agent_info.csv' -delete; find $O -name '*.db' -delete
ls -la $O
