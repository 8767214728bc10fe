set -u
O=gpurun_out/r06h; mkdir -p $O
for d in 0 1 2 3 4 8; do echo "## PT_LNLIN_DBG=$d" >> $O/lnlin_ablations.txt; PT_LNLIN_DBG=$d python tools/lnlin_bench.py --rows 258048 2>&1 | grep "ln_linear\|igemm alone" >> $O/lnlin_ablations.txt; done
cat $O/lnlin_ablations.txt
