set -u
O=gpurun_out/r06l; mkdir -p $O
python -m pytest tests/test_backward_gpu.py -m gpu -x -q -k "hipgraph" -s > $O/tests_graph.txt 2>&1
tail -12 $O/tests_graph.txt
for v in "eager_default:" "eager_no_enc:--no-encoder-stream" "graph_default:--graph" "graph_no_wgrad:--graph --no-wgrad-stream" "graph_one_stream:--graph --no-wgrad-stream --no-encoder-stream --no-spatial-stream" "eager_one_stream:--no-wgrad-stream --no-encoder-stream --no-spatial-stream"; do
  n=${v%%:*}; f=${v#*:}
  python tools/train_step_bench.py --json --steps 7 --warmup 3 $f 2> $O/$n.err | grep ms_per_step > $O/$n.json
  echo "$n $(python -c "import json;d=json.load(open('$O/$n.json'));print(d['ms_per_step'], d['host_enqueue_ms'], d['ms_per_step_all'])")"
done
