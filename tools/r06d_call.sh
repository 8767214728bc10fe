set -u
O=gpurun_out/r06d; mkdir -p $O
python -m pytest tests/test_kernels_gpu.py -m gpu -x -q -k "ln_linear or splitk or test_linear" > $O/tests_lnlin.txt 2>&1
python tools/lnlin_bench.py > $O/lnlin_bench_alone.txt 2>&1
python tools/ab_bench.py --rounds 3 two=PT_FUSED_LNLIN=0 one=PT_FUSED_LNLIN=1 > $O/clip_ab_lnlin_L.txt 2>&1
python tools/ab_bench.py --rounds 3 --workload M two=PT_FUSED_LNLIN=0 one=PT_FUSED_LNLIN=1 > $O/clip_ab_lnlin_M.txt 2>&1
python -m pytest tests/test_parity_ladder_gpu.py -m gpu -x -q -k "level0 or network_ladder or one_loop" > $O/tests_ladder.txt 2>&1
tail -3 $O/*.txt
