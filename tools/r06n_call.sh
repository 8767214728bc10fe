set -u
O=gpurun_out/r06n; mkdir -p $O
python -m pytest tests/test_backward_gpu.py tests/test_training_objective_gpu.py -m gpu -x -q > $O/tests_backward.txt 2>&1
tail -4 $O/tests_backward.txt
for v in "eager_default:" "graph_default:--graph"; do
  n=${v%%:*}; f=${v#*:}
  python tools/train_step_bench.py --json --steps 9 --warmup 3 $f 2> $O/$n.err | grep ms_per_step > $O/$n.json
  echo "$n $(python -c "import json;d=json.load(open('$O/$n.json'));print(d['ms_per_step'], d['host_enqueue_ms'], d['ms_per_step_all'])")"
done
