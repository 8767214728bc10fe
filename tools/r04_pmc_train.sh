# PMC readings of the training step's kernels (one gpurun call): MFMA busy / VALU / wait cycles per kernel family, then HBM-side bytes.
#   bash tools/r04_pmc_train.sh      -> gpurun_out/pmc_train/{pmc_cycles.txt, pmc_fetch.txt, pmc_write.txt}
mkdir -p gpurun_out/pmc_train
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_train/cycles -o p -- python3 $GRAFT_REPO_ROOT/tools/train_step_bench.py --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_train/cycles.log 2>&1
for c in FETCH_SIZE WRITE_SIZE; do
  rocprofv3 --kernel-trace --pmc $c --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/pmc_train/$c -o p -- python3 $GRAFT_REPO_ROOT/tools/train_step_bench.py --steps 1 --warmup 1 > $GRAFT_REPO_ROOT/gpurun_out/pmc_train/$c.log 2>&1
done
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/pmc_train/cycles > gpurun_out/pmc_train/pmc_cycles.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_train/FETCH_SIZE > gpurun_out/pmc_train/pmc_fetch.txt 2>&1
python3 tools/pmc_summary.py gpurun_out/pmc_train/WRITE_SIZE > gpurun_out/pmc_train/pmc_write.txt 2>&1
find gpurun_out/pmc_train -name '*.csv' -delete
