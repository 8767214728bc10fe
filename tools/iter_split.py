#!/usr/bin/env python3
"""Per-iteration kernel split from a rocprofv3 --kernel-trace --stats CSV of `bench.py --infer-steps 2 --steps 1 --warmup 1 --no-graph ...`
(4 eager loop iterations: warm-up clip + timed clip, 2 iterations each):   python tools/iter_split.py <kernel_stats.csv> [iterations=4]"""
import csv, re, sys
it = int(sys.argv[2]) if len(sys.argv) > 2 else 4
groups = [("igemm10 (256x320)", r"igemm10_kernel"), ("igemm8 (256x256)", r"igemm8_kernel"), ("other igemm tiles", r"igemm_kernel<"),
          ("fused feed-forward", r"ffn320_kernel"), ("LayerNorm + QKV", r"lnlin320_kernel"), ("spatial attention", r"attn_spatial_kernel"),
          ("temporal attention", r"attn_temporal_kernel"), ("GroupNorm apply", r"gn_apply_kernel"), ("GroupNorm partial sums", r"gn_partial_kernel"),
          ("LayerNorm", r"layernorm"), ("split-K reduce", r"splitk_reduce")]
tot = {g: 0.0 for g, _ in groups}
rest = 0.0
for r in csv.DictReader(open(sys.argv[1])):
    ns = float(r["TotalDurationNs"])
    for g, pat in groups:
        if re.search(pat, r["Name"]):
            tot[g] += ns
            break
    else:
        if not re.search(r"at::native|rocclr|distribution_elementwise", r["Name"]):      # set-up (random init, packing copies) is not the loop
            rest += ns
s = sum(tot.values()) + rest
print(f"per loop iteration ({it} iterations in the trace), ms: " + ", ".join(f"{g} {v / it / 1e6:.1f}" for g, v in tot.items() if v) +
      f", other loop kernels {rest / it / 1e6:.1f}; sum {s / it / 1e6:.1f}")
