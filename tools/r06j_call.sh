set -u
O=gpurun_out/r06j; mkdir -p $O
python -m pytest tests -m gpu -x -q > $O/gpu_tests.txt 2>&1
tail -5 $O/gpu_tests.txt
