#!/bin/bash
# A/B of experimental builds of the spatial attention kernel (tools/variants/_build/libpt_attn_*.so) against the in-tree library:
# correctness against an fp32 softmax first, then three interleaved timing rounds of the level-0 launch, separate processes.
#   bash tools/attn_ab.sh v4 v6 > gpurun_out/attn_ab.txt
for v in "$@"; do echo "== $v"; PT_LIB=tools/variants/_build/libpt_attn_$v.so ATTN_CHECK=1 ATTN_PRE= python3 tools/attn_one.py 1 64 5 2>&1 | grep -v amdgpu.ids; done
echo "== in-tree"; ATTN_CHECK=1 python3 tools/attn_one.py 1 64 5 2>&1 | grep -v amdgpu.ids
for r in 1 2 3; do
  ATTN_PRE=1 python3 tools/attn_one.py 2>&1 | grep attn_spatial
  for v in "$@"; do PT_LIB=tools/variants/_build/libpt_attn_$v.so ATTN_PRE=1 python3 tools/attn_one.py 2>&1 | grep attn_spatial; done
done
