"""Time of the reference's ControlNet training step (scripts/train_svd_traj_VIPSeg_14.py:1264-1425; start_ft.sh: 14 frames of
320 x 576, batch 1, fp16 mixed precision) on one MI355X: full-size U-Net (frozen) + ControlNet (681 M trainable parameters),
random init, synthetic latents / trajectory maps.  Prints ms per step (forward + backward + AdamW), the executed matrix flops
of the three families (pt_igemm_f16: forward and data gradients; pt_gemm_f16: weight gradients and attention backward) and
the memory high-water mark.

    python tools/train_step_bench.py [--steps 5] [--height 320 --width 576] [--frames 14] [--tiny] [--gemm-table] [--igemm-table]
    python tools/train_step_bench.py --json --steps 7 --warmup 3      # what bench.py --train-step runs as a child process:
        wall-clock median without hipEvent brackets, the host's enqueue time, full garbage collections inside the timed steps
"""
import argparse
import ctypes as C
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=320)
    ap.add_argument("--width", type=int, default=576)
    ap.add_argument("--frames", type=int, default=14)
    ap.add_argument("--tiny", action="store_true", help="the test-sized networks (plumbing check)")
    ap.add_argument("--gemm-table", action="store_true", help="per-shape table of the pt_gemm_f16 launches of one step")
    ap.add_argument("--igemm-table", action="store_true", help="per-shape table of the pt_igemm_f16 launches of one step (forward + data gradients)")
    ap.add_argument("--no-spatial", action="store_true", help="skip the single-frame spatial loss pass")
    ap.add_argument("--graph", action="store_true", help="ControlNetTrainer(use_graph=True): forward + backward replayed as a hipGraph (one per spatial "
                                                       "frame index, all captured before the timed steps)")
    ap.add_argument("--no-wgrad-stream", action="store_true"); ap.add_argument("--no-encoder-stream", action="store_true"); ap.add_argument("--no-spatial-stream", action="store_true"); ap.add_argument("--no-pack-stream", action="store_true")
    ap.add_argument("--json", action="store_true", help="bench.py's train_step leg: time the steps without hipEvent brackets (median), count the matrix "
                                                      "flops in one extra bracketed step, print ONE JSON object")
    a = ap.parse_args()
    from posetraj_amd import ControlNetSDVModel, UNetSpatioTemporalConditionControlNetModel, hip
    from posetraj_amd.training import ControlNetTrainer
    dev = torch.device("cuda:0")
    cfg = dict(num_attention_heads=(5, 10, 20, 20), num_frames=a.frames)
    ce = (16, 32, 96, 256)
    if a.tiny:
        cfg = dict(block_out_channels=(64, 64, 128, 128), num_attention_heads=(1, 1, 2, 2), cross_attention_dim=16, addition_time_embed_dim=8,
                   projection_class_embeddings_input_dim=24, layers_per_block=1, num_frames=a.frames)
        ce = (8, 8, 16, 32)
    t0 = time.time()
    unet = UNetSpatioTemporalConditionControlNetModel(**cfg).init_random_(seed=1, device=dev, keep_source=True)
    cn = ControlNetSDVModel.from_unet(unet, conditioning_embedding_out_channels=ce)
    sd = cn.state_dict()
    g = torch.Generator().manual_seed(3)
    for k in sd:                                              # a ControlNet some way into training: the zero-convs have moved
        if k.startswith(("controlnet_down_blocks", "controlnet_mid_block", "controlnet_cond_embedding.conv_out")):
            sd[k] = (torch.randn(sd[k].shape, generator=g) * 0.02).half()
    ccfg = dict(cn.config)
    del cn
    tr = ControlNetTrainer(ccfg, sd, unet, learning_rate=1e-5, conditioning_dropout_prob=0.1, freeze_gc=True, use_graph=a.graph,
                           wgrad_stream=not a.no_wgrad_stream, encoder_stream=not a.no_encoder_stream, spatial_stream=not a.no_spatial_stream, pack_stream=not a.no_pack_stream)
    print(f"set-up {time.time() - t0:.1f} s; {tr.params.numel / 1e6:.1f} M trainable parameters (fp32 master + gradient + 2 Adam moments)")
    h, w = a.height // 8, a.width // 8
    D = unet.config.cross_attention_dim
    lat = torch.randn(1, a.frames, 4, h, w, generator=g) * 0.18215 * 5
    emb = torch.randn(1, 1, D, generator=g)
    traj = torch.rand(1, a.frames, 3, a.height, a.width, generator=g) * 2 - 1
    mv = torch.tensor([127.0])
    gen = torch.Generator().manual_seed(5)
    L = hip.lib()
    for _ in range(a.warmup):
        out = tr.step(lat, emb, mv, traj, generator=gen, use_spatial=not a.no_spatial)
    torch.cuda.synchronize()
    n_graphs = 0
    if a.graph:
        tg = time.time()
        n_graphs = tr.warm_graphs(lat, emb, mv, traj, generator=gen, use_spatial=not a.no_spatial)
        print(f"captured {n_graphs} step graphs in {time.time() - tg:.1f} s", file=sys.stderr if a.json else sys.stdout)
    if a.json:
        import gc
        import json
        tt, host, gc_log, gc_t0, replayed = [], [], [], [0.0], []

        def on_gc(phase, info):                                  # host-side hiccups: which collections ran inside the timed steps
            if phase == "start":
                gc_t0[0] = time.perf_counter()
            else:
                gc_log.append((info["generation"], round(1000 * (time.perf_counter() - gc_t0[0]), 1), info["collected"]))
        gc.callbacks.append(on_gc)
        torch.cuda.reset_peak_memory_stats()
        for _ in range(a.steps):
            ts = time.perf_counter()
            out = tr.step(lat, emb, mv, traj, generator=gen, use_spatial=not a.no_spatial)
            torch.cuda.synchronize()
            tt.append(time.perf_counter() - ts)
            host.append(out["host_enqueue_ms"])
            replayed.append(bool(out.get("graph_replay")))
        gc.callbacks.remove(on_gc)
        peak = torch.cuda.max_memory_allocated() / 2 ** 30
        L.pt_prof_enable(1)
        tr.step(lat, emb, mv, traj, generator=gen, use_spatial=not a.no_spatial)
        torch.cuda.synchronize()
        flops = 0.0
        for f in (0, 1, 2):
            n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
            L.pt_prof_collect(f, C.byref(n), C.byref(ms), C.byref(fl))
            flops += fl.value
        L.pt_prof_enable(0)
        med = sorted(tt)[len(tt) // 2]
        print(json.dumps({"ms_per_step": round(1000 * med, 1), "clips_per_s": round(1.0 / med, 2), "ms_per_step_all": [round(1000 * v, 1) for v in tt],
                          "host_enqueue_ms": round(sorted(host)[len(host) // 2], 1),
                          "matrix_TFLOP_per_step": round(flops / 1e12, 2), "peak_device_GiB": round(peak, 1),
                          "trainable_params_M": round(tr.params.numel / 1e6, 1), "loss_finite": bool(out["loss"] == out["loss"]),
                          "optimizer_stepped": bool(out["stepped"]), "hipgraph": bool(a.graph), "step_graphs": n_graphs,
                          "graph_replays_in_timed_steps": int(sum(replayed)),
                          "host_gc_in_timed_steps": [g for g in gc_log if g[1] >= 1.0], "streams": 1 + int(tr.wgrad_stream) + int(tr.spatial_stream and not a.no_spatial) + int(tr.encoder_stream)}))
        return
    L.pt_prof_enable(1)
    torch.cuda.reset_peak_memory_stats()
    t1 = time.time()
    for _ in range(a.steps):
        out = tr.step(lat, emb, mv, traj, generator=gen, use_spatial=not a.no_spatial)
    torch.cuda.synchronize()
    dt = (time.time() - t1) / a.steps
    fam = {}
    for f, name in ((0, "pt_igemm_f16"), (1, "pt_attn_spatial_f16"), (2, "pt_gemm_f16")):
        n, ms, fl = C.c_int64(), C.c_double(), C.c_double()
        L.pt_prof_collect(f, C.byref(n), C.byref(ms), C.byref(fl))
        fam[name] = (n.value / a.steps, ms.value / a.steps, fl.value / a.steps)
    L.pt_prof_enable(0)
    if a.gemm_table:
        from posetraj_amd import autodiff as AD
        AD.GEMM_LOG = []
        L.pt_prof_enable(1)
        tr.step(lat, emb, mv, traj, generator=gen, use_spatial=not a.no_spatial)
        torch.cuda.synchronize()
        cap = len(AD.GEMM_LOG) + 16
        ms, fl = (C.c_double * cap)(), (C.c_double * cap)()
        n = L.pt_prof_collect_list(2, ms, fl, cap)
        L.pt_prof_enable(0)
        assert n == len(AD.GEMM_LOG), (n, len(AD.GEMM_LOG))
        agg = {}
        for i, key in enumerate(AD.GEMM_LOG):
            e = agg.setdefault(key, [0, 0.0, 0.0])
            e[0] += 1; e[1] += ms[i]; e[2] += fl[i]
        print("  pt_gemm_f16 shapes of one step: M x N x K, batch, A/B form (T: unit stride across k), out_mode, splits, conv gather")
        for key, (cnt, t, f) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
            M_, N_, K_, nb_, fa, fb, om, sp, ga = key
            print(f"    {M_:5d} x {N_:5d} x {K_:6d}  b{nb_:6d} {fa}{fb} mode {om} splits {sp:3d} {'gather' if ga else '      '}  {cnt:3d} x  {t / cnt * 1e3:8.1f} us  {f / max(t, 1e-9) / 1e9:7.1f} TFLOP/s  total {t:6.2f} ms")
        AD.GEMM_LOG = None
    if a.igemm_table:
        import collections
        from posetraj_amd import ops
        ops.Profiler.collect_list("igemm")                       # drop what earlier legs of this run left in the family
        ops.Profiler.shapes = []
        with ops.Profiler():
            tr.step(lat, emb, mv, traj, generator=gen, use_spatial=not a.no_spatial)
            torch.cuda.synchronize()
        ms, fl = ops.Profiler.collect_list("igemm")
        shapes, ops.Profiler.shapes = ops.Profiler.shapes, None
        assert len(ms) == len(shapes), (len(ms), len(shapes))
        agg = collections.OrderedDict()
        for sh, m, f in zip(shapes, ms, fl):
            e = agg.setdefault(sh, [0, 0.0, 0.0]); e[0] += 1; e[1] += m; e[2] += f
        tot = sum(ms)
        print(f"  pt_igemm_f16 shapes of one step: {len(ms)} launches, {tot:.2f} ms, {sum(fl) / tot / 1e9:.1f} TFLOP/s")
        print(f"  {'M':>8} {'N':>6} {'K':>6} k s u {'C1':>5} a e {'n':>4} {'ms':>9} {'%':>6} {'TFLOP/s':>8}")
        for sh, (n, m, f) in list(sorted(agg.items(), key=lambda kv: -kv[1][1]))[:45]:
            M_, N_, K_, kh, kw, st, up, c1, act, epi = sh
            print(f"  {M_:8d} {N_:6d} {K_:6d} {kh}x{kw} {st} {up} {c1:5d} {act} {epi} {n:4d} {m:9.3f} {100 * m / tot:6.2f} {f / m / 1e9:8.1f}")
    print(f"{a.frames} x {a.height} x {a.width}, batch 1: {dt * 1e3:.1f} ms per training step (wall, incl. host; prof events on); "
          f"loss {out['loss']:.4f}, grad norm {out.get('grad_norm', float('nan')):.3e}, stepped {out['stepped']}, loss scale {tr.loss_scale:g}")
    tot_fl = 0.0
    for name, (n, ms, fl) in fam.items():
        tot_fl += fl
        print(f"  {name:22s} {n:8.0f} launches  {ms:9.2f} ms  {fl / 1e12:8.2f} TFLOP  {fl / max(ms, 1e-9) / 1e9:8.1f} TFLOP/s")
    print(f"  matrix flops per step {tot_fl / 1e12:.1f} TFLOP -> {tot_fl / dt / 1e12:.0f} TFLOP/s over the whole step; "
          f"peak device memory {torch.cuda.max_memory_allocated() / 2 ** 30:.1f} GiB")


if __name__ == "__main__":
    main()
