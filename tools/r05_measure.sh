#!/bin/bash
# Round-5 measurement run on one MI355X box (one gpurun call, so that every number comes from the same device):
#   bash tools/r05_measure.sh <tag>        writes everything under gpurun_out/<tag>/
# 1 bench line (default run + decode leg + end-to-end leg)  2 rocprofv3 --kernel-trace --stats of 2 eager loop iterations
# 3 PMC traffic (FETCH_SIZE / WRITE_SIZE, separate passes) + PMC of the feed-forward kernels   4 per-shape igemm table   5 rocprofv3 stats of the VAE decode
set -u
TAG=${1:-r05}
REPO=$(pwd)
OUT=$REPO/gpurun_out/$TAG
mkdir -p $OUT
export TMPDIR=/tmp
if [ "${ONLY_PMC:-0}" = "0" ]; then
python bench.py > $OUT/bench_L_default.json 2> $OUT/bench_L_default.err                      # exactly the driver's command (CPU baseline: one real 72 x 128 iteration)
python bench.py --end-to-end --clips-per-gpu 2 --no-cpu-baseline > $OUT/bench_L_legs.json 2> $OUT/bench_L_legs.err
python bench.py --train-step --no-cpu-baseline --no-decode --no-profile > $OUT/bench_L_train_step.json 2> $OUT/bench_L_train_step.err   # (no CPU leg beside the training child: its 15 host threads doubled the step's enqueue time)
python bench.py --workload M --no-cpu-baseline > $OUT/bench_M.json 2> $OUT/bench_M.err
fi
cd /tmp
[ "${ONLY_PMC:-0}" = "0" ] && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_loop -o p -- python3 $REPO/bench.py --infer-steps 2 --steps 1 --warmup 1 --no-graph --no-profile --no-cpu-baseline --no-decode > $OUT/prof_loop.log 2>&1
[ "${ONLY_PMC:-0}" = "0" ] && rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_vae -o p -- python3 $REPO/tools/vae_bench.py --workload L --reps 1 > $OUT/prof_vae.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  mkdir -p $OUT/pmc_$c
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $OUT/pmc_$c -o p -- python3 $REPO/tools/traffic_run.py $OUT/pmc_$c/shapes.json > $OUT/pmc_$c.log 2>&1
  f=$(find $OUT/pmc_$c -name '*counter_collection.csv' | head -1)
  [ -n "$f" ] && [ "$f" != "$OUT/pmc_$c/p_counter_collection.csv" ] && cp $f $OUT/pmc_$c/p_counter_collection.csv
  python3 $REPO/tools/traffic_extract.py $OUT/pmc_$c >> $OUT/pmc_$c.log 2>&1
done
# PMC of the level-0 feed-forward in both forms (MFMA busy, VALU, waits, LDS bank conflicts; counters in passes of their own)
mkdir -p $OUT/pmc_ffn
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ffn -o p -- python3 $REPO/tools/ffn_one.py > $OUT/pmc_ffn.log 2>&1
mkdir -p $OUT/pmc_ffn2
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_INSTS_VALU SQ_INSTS_MFMA GRBM_GUI_ACTIVE --output-format csv -d $OUT/pmc_ffn2 -o p -- python3 $REPO/tools/ffn_one.py > $OUT/pmc_ffn2.log 2>&1
cd $REPO
python3 tools/pmc_summary.py $OUT/pmc_ffn > $OUT/pmc_ffn_L0.txt 2>&1
python3 tools/pmc_summary.py $OUT/pmc_ffn2 >> $OUT/pmc_ffn_L0.txt 2>&1
python3 tools/traffic_summary.py $OUT/pmc_FETCH_SIZE $OUT/pmc_WRITE_SIZE $OUT/summary_L_$TAG > $OUT/per_shape_L_$TAG.txt 2>&1
[ "${ONLY_PMC:-0}" = "0" ] && python tools/shape_report.py --workload L > $OUT/igemm_shapes_L_$TAG.txt 2>&1
# keep the small files only
find $OUT -name '*counter_collection.csv' -delete
find $OUT -name '*kernel_trace.csv' -delete
find $OUT -name '*agent_info.csv' -delete
ls -la $OUT
