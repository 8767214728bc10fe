#!/bin/bash
# PMC passes over the level-0 launch of the spatial attention kernel (PT_LIB selects an experimental build):
#   bash tools/attn_pmc.sh <outdir-under-gpurun_out> [lib]
out=$GRAFT_REPO_ROOT/gpurun_out/$1; mkdir -p $out
[ -n "$2" ] && export PT_LIB=$GRAFT_REPO_ROOT/$2
export ATTN_PRE=1
cd /tmp; export TMPDIR=/tmp
rocprofv3 --kernel-trace --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p1 -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > $out/p1.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_LDS_ADDR_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_LDS SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE --output-format csv -d $out/p2 -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > $out/p2.log 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_MFMA SQ_INSTS_SALU SQ_INSTS_VMEM SQ_WAIT_ANY SQ_INST_CYCLES_SALU GRBM_GUI_ACTIVE --output-format csv -d $out/p3 -o p -- python3 $GRAFT_REPO_ROOT/tools/attn_one.py > $out/p3.log 2>&1
cd $GRAFT_REPO_ROOT
for p in p1 p2 p3; do python3 tools/pmc_summary.py $out/$p 2>&1 | grep -A12 attn_spatial; done > $out/summary.txt
find $out -name '*.csv' -delete
cat $out/summary.txt
