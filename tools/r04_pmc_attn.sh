mkdir -p gpurun_out/r04b
python tools/ab_bench.py --rounds 2 base= split=BENCH_FLAGS=--split-cfg > gpurun_out/r04b/clip_ab_split_cfg_box2.txt 2>&1
python tools/attn_general_one.py > gpurun_out/r04b/attn_general_timing.txt 2>&1
python tools/attn_general_one.py 2 257 16 80 >> gpurun_out/r04b/attn_general_timing.txt 2>&1
python tools/attn_general_one.py 28 576 10 128 >> gpurun_out/r04b/attn_general_timing.txt 2>&1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU GRBM_GUI_ACTIVE --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/r04b/pmc_attn -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_general_one.py > $GRAFT_REPO_ROOT/gpurun_out/r04b/pmc_attn.log 2>&1
cd $GRAFT_REPO_ROOT
python3 tools/pmc_summary.py gpurun_out/r04b/pmc_attn > gpurun_out/r04b/pmc_attn_general_512.txt 2>&1
find gpurun_out/r04b -name '*.csv' -delete
