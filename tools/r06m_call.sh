python -m pytest tests/test_backward_gpu.py -m gpu -x -q -k "hipgraph" -s > gpurun_out/r06m_tests.txt 2>&1
grep -v amdgpu gpurun_out/r06m_tests.txt | grep "Error\|assert\|step\|passed\|failed" | tail -20
