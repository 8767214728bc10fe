#!/usr/bin/env python3
"""VERDICT r03 #4b: GroupNorm partial sums in the producing epilogue - measured as a LOWER BOUND on one shape.

    python tools/variants/gn_epilogue_ab.py build      (here: patches a scratch copy of csrc/igemm.hip, builds tools/variants/_build/libposetraj_hip_gn.so)
    python tools/variants/gn_epilogue_ab.py run        (on the MI355X: base library vs variant, interleaved, one process each)

The variant adds to the row-wise tail of the wide / one-side-input variant (the `conv2 + residual` epilogue that produces the
tensor the next GroupNorm reads) what per-channel statistics need at the least: for each finished row pass, 8 sums + 8 sums of
squares per lane accumulated into a per-workgroup [320][2] fp32 table in LDS (ds_add_f32; the kernel's idle landing rows), and
the table written out per tile.  Missing from the real thing (so the real cost is higher): the per-group fold of 10-channel
groups, the second reduction over the M tiles of a sample, the apply pass's switch to those partials.  What it would replace:
the statistics pass over the [258048, 320] tensor (pt_groupnorm_stats: ~33-36 us, profiles/r03)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tools", "variants", "_build")
LIB = os.path.join(OUT, "libposetraj_hip_gn.so")


def build():
    sys.path.insert(0, ROOT)
    from posetraj_amd import hip
    hip.build()
    os.makedirs(OUT, exist_ok=True)
    src = open(os.path.join(hip.CSRC, "igemm.hip")).read()
    anchor = "        f16x8 o;\n#pragma unroll\n        for (int j = 0; j < 8; ++j) o[j] = (f16)v[j];\n        if constexpr (WIDE) {"
    assert src.count(anchor) == 1
    # form 1 (first measurement, profiles/r04/gn_epilogue_ab.txt: 590 -> 1226 us): 16 ds_add_f32 per lane and row pass straight into
    # the table - the 16 lanes that share a column (one row each) hit the same address and serialise.
    # form 2 (below): the 16 rows are first folded in registers by four cross-lane steps (lane ^ 4, 8, 16, 32: the 4 lanes of a row
    # stay apart), then ONE lane per column adds into the table.
    patch = '''        if constexpr (WIDE && NS == 1) {                     // EXPERIMENT: per-channel sum / sum of squares of the finished values
            if (kp.dbg & 8) {
                float s8[8], q8[8];
#pragma unroll
                for (int j = 0; j < 8; ++j) { s8[j] = v[j]; q8[j] = v[j] * v[j]; }
#pragma unroll
                for (int off = 4; off < 64; off <<= 1) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) { s8[j] += __shfl_xor(s8[j], off); q8[j] += __shfl_xor(q8[j], off); }
                }
                if (lane < 4) {
                    __attribute__((address_space(3))) float* S = (__attribute__((address_space(3))) float*)(smem + CF::SMEM);
                    const int cb = (wcol0 % 320) + c8;
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        __hip_atomic_fetch_add(S + 2 * (cb + j), s8[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                        __hip_atomic_fetch_add(S + 2 * (cb + j) + 1, q8[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                    }
                }
            }
        }
'''
    src = src.replace(anchor, patch + anchor)
    # write the table out at the end of the tail (one 8-byte store per channel and tile), so that nothing above is dead code
    tail_end = "    if (PREFETCH && fastpath) {\n#pragma unroll\n        for (int g = 0; g < NPASS; ++g) load_side(0, g);\n    }"
    assert src.count(tail_end) == 1
    src = src.replace('#include "pt_common.h"', '#include "%s/pt_common.h"' % hip.CSRC, 1)
    flush = '''
    struct GnFlush {                                          // EXPERIMENT: runs when the tail returns
        const KParams& kp; char* smem; int lane, wave, smem_bytes;
        __device__ ~GnFlush() {
            if ((kp.dbg & 8) && kp.p.splitk_ws) {
                __builtin_amdgcn_s_waitcnt(0xC07F);
                const float* S = (const float*)(smem + smem_bytes);
                const int t = wave * 64 + lane;
                if (t < 320) ((f32x2*)kp.p.splitk_ws)[(size_t)blockIdx.x * 320 + t] = (f32x2){S[2 * t], S[2 * t + 1]};
            }
        }
    } gn_flush{kp, smem, lane, wave, CF::SMEM};
'''
    src = src.replace(tail_end, flush + tail_end)
    tmp = os.path.join(OUT, "igemm_gn.hip")
    open(tmp, "w").write(src)
    obj = os.path.join(OUT, "igemm_gn.o")
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    t0 = time.time()
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I", hip.CSRC, "-c", tmp, "-o", obj])
    others = [os.path.join(hip.CSRC, "_obj", s + ".o") for s in hip.SOURCES if s != "igemm.hip"]
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-fPIC", "-shared", "-o", LIB, obj] + others)
    print(f"built {LIB} in {time.time() - t0:.0f} s")


CHILD = r'''
import os, sys, torch
sys.path.insert(0, %r)
from posetraj_amd import hip, ops
from posetraj_amd.packing import pack_conv2d
dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
N, H, W, C = 28, 72, 128, 320
x = torch.randn(N, H, W, C, generator=g).half().to(dev)
w = (torch.randn(C, C, 3, 3, generator=g) * (9 * C) ** -0.5)
pw = pack_conv2d(w, torch.zeros(C), dev)
res = torch.randn(N * H * W, C, generator=g).half().to(dev)
hip.check(hip.lib().pt_igemm_force_config(3))
for _ in range(3):
    y = ops.igemm(x, pw, geom=(N, H, W), res=res, wide=True)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record(); y = ops.igemm(x, pw, geom=(N, H, W), res=res, wide=True); b.record(); torch.cuda.synchronize()
    ts.append(a.elapsed_time(b) * 1e3)
ts.sort()
print("%%s  conv3x3 258048x320x2880 + res, wide: median %%.1f us  min %%.1f us" %% (os.environ.get("TAG"), ts[len(ts) // 2], ts[0]))
'''


def run():
    for rnd in range(3):
        for tag, env in (("base   ", {}), ("variant", {"PT_LIB": LIB, "PT_IGEMM_DBG": "8"})):
            e = dict(os.environ, TAG=f"round {rnd} {tag}", **env)
            subprocess.run([sys.executable, "-c", CHILD % ROOT], env=e, check=True)


if __name__ == "__main__":
    {"build": build, "run": run}[sys.argv[1]]()
