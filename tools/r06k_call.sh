set -u
O=gpurun_out/r06k; mkdir -p $O
python -m pytest tests/test_backward_gpu.py -m gpu -x -q -k "hipgraph or training_loop or full_width_training" -s > $O/tests_graph.txt 2>&1
tail -30 $O/tests_graph.txt
python tools/train_step_bench.py --json --steps 7 --warmup 3 > $O/train_step_eager.json 2> $O/train_step_eager.err
python tools/train_step_bench.py --json --steps 7 --warmup 3 --graph > $O/train_step_graph.json 2> $O/train_step_graph.err
cat $O/train_step_eager.json $O/train_step_graph.json | grep ms_per_step
