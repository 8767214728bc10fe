#!/usr/bin/env python3
"""A/B of library tuning switches on ONE box (devices differ by up to ~10 % in wall time, so numbers from different gpurun
calls do not compare): runs bench.py once per variant per round, interleaved, and prints the clip time of each.

    python tools/ab_bench.py [--rounds 2] [--workload L] NAME=ENV1=V1,ENV2=V2 ...
e.g. python tools/ab_bench.py base= nostagger=PT_IGEMM_STAGGER=0 attn4=PT_ATTN_8WAVE=0
"""
import json
import os
import subprocess
import sys

root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
args = sys.argv[1:]
rounds, workload = 2, "L"
while args and args[0].startswith("--"):
    if args[0] == "--rounds":
        rounds = int(args[1])
    elif args[0] == "--workload":
        workload = args[1]
    args = args[2:]
variants = []
for a in args:
    name, _, envs = a.partition("=")
    env = dict(kv.split("=", 1) for kv in envs.split(",") if kv)          # BENCH_FLAGS=--no-overlap passes bench.py flags
    variants.append((name, env))
res = {n: [] for n, _ in variants}
for r in range(rounds):
    for name, env in variants:
        p = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--steps", "2", "--warmup", "1", "--no-cpu-baseline",
                            "--no-profile", "--no-decode", "--workload", workload] + env.get("BENCH_FLAGS", "").split(),
                           env=dict(os.environ, **{k: v for k, v in env.items() if k != "BENCH_FLAGS"}), capture_output=True, text=True)
        line = [ln for ln in p.stdout.splitlines() if ln.startswith("{")]
        ms = json.loads(line[-1])["ms_per_step"] if line else float("nan")
        res[name].append(ms)
        print(f"round {r} {name:14s} {ms:9.1f} ms/clip   {env}", flush=True)
base = min(res[variants[0][0]])
for name, _ in variants:
    v = res[name]
    print(f"{name:14s} min {min(v):9.1f}  mean {sum(v) / len(v):9.1f}  vs {variants[0][0]} {min(v) / base:6.3f}")
