#!/bin/bash
# Round-6 measurement run on one MI355X box (ONE gpurun call, so that every number of a tag comes from the same device):
#   SECTIONS="base cam M busy L" bash tools/r06_measure.sh <tag>       writes everything under gpurun_out/<tag>/
# base  bench line at L (driver's command, CPU baseline included unless NOCPU=1)
# cam   bench line of the camera branch (BASELINE configs[4] on one GPU)
# M     BASELINE configs[1] (14 x 320 x 576): bench line, 2 clips per GPU, per-shape table, rocprofv3 kernel stats, PMC traffic, energy table
# busy  MFMA-busy PMC pass over one real iteration at L and at M (tools/mfma_busy.py)
# L     per-shape table, rocprofv3 kernel stats and PMC traffic at L
set -u
TAG=${1:-r06}
SECTIONS=${SECTIONS:-"base cam M busy L"}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
has() { [[ " $SECTIONS " == *" $1 "* ]]; }
NOCPU=${NOCPU:-0}; cpuflag=""; [ "$NOCPU" = "1" ] && cpuflag="--no-cpu-baseline"

traffic() {   # $1 = workload
  for c in FETCH_SIZE WRITE_SIZE; do
    d=$OUT/pmc_${1}_$c; mkdir -p $d
    (cd /tmp && rocprofv3 --kernel-trace --pmc $c --output-format csv -d $d -o p -- python3 $REPO/tools/traffic_run.py $d/shapes.json $1 > $d.log 2>&1)
    f=$(find $d -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && [ "$f" != "$d/p_counter_collection.csv" ] && cp $f $d/p_counter_collection.csv
    python3 $REPO/tools/traffic_extract.py $d >> $d.log 2>&1
  done
  python3 tools/traffic_summary.py $OUT/pmc_${1}_FETCH_SIZE $OUT/pmc_${1}_WRITE_SIZE $OUT/summary_${1}_$TAG > $OUT/per_shape_traffic_${1}_$TAG.txt 2>&1
  for c in FETCH_SIZE WRITE_SIZE; do cp $OUT/pmc_${1}_$c/igemm_dispatches.json $OUT/dispatches_${c}_${1}_$TAG.json; done
  cp $OUT/pmc_${1}_FETCH_SIZE/shapes.json $OUT/shapes_${1}_$TAG.json
}
stats() {     # $1 = workload
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_$1 -o p -- python3 $REPO/bench.py --workload $1 --infer-steps 2 --steps 1 --warmup 1 --no-graph --no-profile --no-cpu-baseline --no-decode > $OUT/prof_$1.log 2>&1)
  f=$(find $OUT/prof_$1 -name '*kernel_stats.csv' | head -1); [ -n "$f" ] && cp $f $OUT/rocprofv3_kernel_stats_${1}_2iters_$TAG.csv
}
busy() {      # $1 = workload
  d=$OUT/pmc_busy_$1; mkdir -p $d
  (cd /tmp && rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $d -o p -- python3 $REPO/tools/traffic_run.py $d/shapes.json $1 > $d.log 2>&1)
  python3 tools/mfma_busy.py $d > $OUT/mfma_busy_${1}_$TAG.txt 2>> $d.log
}

# PMC traffic first: bench.py replays the summary of THIS build (profiles/rNN/traffic/summary_<workload>_*.json, sha256 of the kernel sources
# inside) as roofline.traffic, so the summaries are put where it looks - on the GPU box, for the bench lines of this very call; the same
# files are committed from gpurun_out/<tag>/ afterwards
if has L; then traffic L; mkdir -p profiles/r06/traffic; cp $OUT/summary_L_$TAG.json profiles/r06/traffic/; fi
if has M; then traffic M; mkdir -p profiles/r06/traffic; cp $OUT/summary_M_$TAG.json profiles/r06/traffic/; fi
if has base; then python bench.py $cpuflag > $OUT/bench_L_default.json 2> $OUT/bench_L_default.err; fi
if has cam; then python bench.py --camera --no-cpu-baseline > $OUT/bench_L_camera.json 2> $OUT/bench_L_camera.err; fi
if has legs; then
  python bench.py --end-to-end --clips-per-gpu 2 --no-cpu-baseline > $OUT/bench_L_legs.json 2> $OUT/bench_L_legs.err
  python bench.py --train-step --no-cpu-baseline --no-decode --no-profile > $OUT/bench_L_train_step.json 2> $OUT/bench_L_train_step.err
fi
if has M; then
  python bench.py --workload M --no-cpu-baseline > $OUT/bench_M.json 2> $OUT/bench_M.err
  python bench.py --workload M --no-cpu-baseline --clips-per-gpu 2 --no-decode > $OUT/bench_M_2clips.json 2> $OUT/bench_M_2clips.err
  python tools/shape_report.py --workload M > $OUT/igemm_shapes_M_$TAG.txt 2>&1
  stats M
  python tools/energy_table.py --workload M > $OUT/energy_table_M_$TAG.txt 2>&1
fi
if has busy; then busy L; busy M; fi
if has L; then
  python tools/shape_report.py --workload L > $OUT/igemm_shapes_L_$TAG.txt 2>&1
  stats L
fi
if has energyL; then python tools/energy_table.py --workload L > $OUT/energy_table_L_$TAG.txt 2>&1; fi
# keep the small files only
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
find $OUT -name '*.db' -delete
ls -la $OUT
