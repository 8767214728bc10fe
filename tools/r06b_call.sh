set -u
O=gpurun_out/r06b; mkdir -p $O
python tools/igemm_cfg_sweep.py > $O/igemm_cfg_sweep_L.txt 2>&1
SWEEP_WORKLOAD=M python tools/igemm_cfg_sweep.py > $O/igemm_cfg_sweep_M.txt 2>&1
python tools/energy_table.py --only L2 --repeat-each 1 > $O/energy_L2_repeat1.txt 2>&1
python tools/energy_table.py --only L2 --repeat-each 4 > $O/energy_L2_repeat4.txt 2>&1
python tools/energy_table.py --only L3 --repeat-each 4 > $O/energy_L3_repeat4.txt 2>&1
python tools/ab_bench.py --rounds 3 base= chunk128=PT_FF_CHUNK_MB=128 > $O/clip_ab_ff_chunk_L1.txt 2>&1
ls -la $O
