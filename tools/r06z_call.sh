set -u
SECTIONS="base cam legs M busy L energyL" bash tools/r06_measure.sh r06zz > gpurun_out/r06zz_measure.log 2>&1
python -m pytest tests -m gpu -x -q > gpurun_out/r06zz/gpu_tests.txt 2>&1
python -c "import __graft_entry__ as g; g.smoke()" > gpurun_out/r06zz/smoke.txt 2>&1
tail -3 gpurun_out/r06zz/gpu_tests.txt gpurun_out/r06zz/smoke.txt
