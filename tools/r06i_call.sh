set -u
O=gpurun_out/r06i; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ln_linear" > $O/tests_lnlin.txt 2>&1
python tools/lnlin_bench.py > $O/lnlin_bench_alone.txt 2>&1
for d in 1 8; do echo "## PT_LNLIN_DBG=$d" >> $O/lnlin_ablations.txt; PT_LNLIN_DBG=$d python tools/lnlin_bench.py --rows 258048 2>&1 | grep "ln_linear" >> $O/lnlin_ablations.txt; done
python tools/ab_bench.py --rounds 3 two=PT_FUSED_LNLIN=0 one=PT_FUSED_LNLIN=1 > $O/clip_ab_lnlin_L.txt 2>&1
for f in $O/*.txt; do echo "== $f"; tail -n 7 $f; done
