python bench.py --train-step --end-to-end > gpurun_out/bench_all.json 2> gpurun_out/bench_all.err; echo "bench rc $?"; tail -c 1500 gpurun_out/bench_all.json
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/gpu_tests.txt; cat gpurun_out/gpu_tests.txt
