#!/usr/bin/env python3
"""Where the joules of one denoise iteration go (VERDICT r04 #1a): every kernel family of the loop replayed ALONE for >= --seconds on
the clip's own launches (shapes, weights and the activations of a real iteration, which are kept alive), with the socket power
(hwmon) and the time per pass:  J per iteration = mean W x ms per pass; the in-kernel shader clock (tools/micro/clock_probe.hip:
s_memtime / s_memrealtime sampled by one resident wave, guide 'DVFS give-back' item 6) comes from a SECOND, shorter replay, because the
resident wave itself slows one-round launches (see replay()).

How: one eager iteration runs through a recording proxy of the C-ABI library (every pt_* call with a copy of its argument struct);
a family's calls are then re-issued in their original order, pass after pass.  Launches that accumulate in place (the zero-convs'
res_post epilogues) are left out of the family replays (they would drift) and kept in the whole-iteration replay, where their
target is rewritten first.  The same harness times an A/B of two kernels for one family (tools/variants/).

    python tools/energy_table.py [--workload L] [--seconds 2.5] > profiles/r05/energy_table_L.txt"""
import argparse, ctypes as C, json, os, statistics, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
from posetraj_amd import (ControlNetSDVModel, EulerDiscreteScheduler, StableVideoDiffusionPipelineControlNet,
                          SVD_SCHEDULER_CONFIG, UNetSpatioTemporalConditionControlNetModel, hip, ops)


class Recorder:
    """Proxy of the loaded library: forwards every call and keeps (name, args) with by-reference structs copied."""

    def __init__(self, real):
        self.real, self.calls, self.on = real, [], False

    def __getattr__(self, name):
        fn = getattr(self.real, name)
        if not name.startswith("pt_") or name in ("pt_last_error", "pt_igemm_splitk_ws_bytes", "pt_abi_version", "pt_set_zero_page"):
            return fn

        def call(*args):
            if self.on:
                kept = []
                for a in args:
                    obj = getattr(a, "_obj", None)              # C.byref(struct)
                    if obj is not None:
                        cp = type(obj).from_buffer_copy(obj)
                        kept.append((C.byref(cp), cp))
                    else:
                        kept.append((a, None))
                self.calls.append((name, kept))
            return fn(*args)
        return call


def classify(name, kept):
    """-> (family label, flops, in_place)"""
    if name == "pt_igemm_f16":
        p = kept[0][1]
        fl = 2.0 * p.M * p.N * p.K
        lvl = {258048: "L0", 64512: "L1", 16128: "L2", 4032: "L3"}.get(p.M, f"M{p.M}")
        inplace = bool(p.res_post) or (p.res and p.res == p.out)
        if p.act == 1:
            return f"igemm GEGLU {lvl}", fl, inplace
        if p.KH * p.KW > 1:
            return f"igemm conv (3x3 / 3x1) {lvl}", fl, inplace
        if p.K >= 4 * p.N:
            return f"igemm FF out-projection {lvl}", fl, inplace
        return f"igemm linear K={p.K} {lvl}", fl, inplace
    if name == "pt_ffn_geglu_f16":
        p = kept[0][1]
        lvl = {258048: "L0", 64512: "L1", 16128: "L2", 4032: "L3"}.get(p.M, f"M{p.M}")
        return f"fused GEGLU feed-forward {lvl}", 2.0 * p.M * (2.0 * p.inner) * p.C + 2.0 * p.M * p.C * p.inner, False
    if name == "pt_attn_spatial_f16":
        return "spatial attention", None, False
    if "groupnorm" in name:
        return "GroupNorm (stats + apply)", 0.0, False
    if "layernorm" in name:
        return "LayerNorm", 0.0, False
    if name == "pt_attn_temporal_f16":
        return "temporal attention", 0.0, False
    return "other (element-wise, embeddings)", 0.0, True


class Clock:
    def __init__(self, seconds, enabled=True):
        so = os.path.join(ROOT, "tools", "micro", "libclock_probe.so")
        self.lib = C.CDLL(so) if enabled and os.path.exists(so) else None
        if self.lib is None:
            return
        self.lib.clock_probe_launch.argtypes = [C.c_void_p, C.c_int, C.c_uint64, C.c_void_p, C.c_void_p]
        self.n = int(seconds * 1000 * 3) + 2000
        self.buf = torch.zeros(2 * self.n, dtype=torch.int64, device="cuda")
        self.stop = torch.zeros(1, dtype=torch.int32).pin_memory()
        self.stream = torch.cuda.Stream()

    def start(self):
        if self.lib is None:
            return
        self.stop[0] = 0
        self.buf.zero_()
        torch.cuda.synchronize()                                # (the probe is not running yet)
        rc = self.lib.clock_probe_launch(self.buf.data_ptr(), self.n, 100000, self.stop.data_ptr(), self.stream.cuda_stream)   # 1 ms
        assert rc == 0, rc

    def finish(self, skip_s=0.6, until_s=float("inf")):
        """-> (median, min, max) GHz over the 1-ms samples taken between skip_s and until_s after the probe's start."""
        if self.lib is None:
            return None
        self.stop[0] = 1
        self.stream.synchronize()
        v = self.buf.cpu().view(-1, 2)
        v = v[v[:, 1] > 0]
        if len(v) < 10:
            return None
        ct, rt = v[:, 0].double(), v[:, 1].double()
        keep = ((rt - rt[0]) >= skip_s * 1e8) & ((rt - rt[0]) <= until_s * 1e8)
        if int(keep.sum()) < 5:
            return None
        dct, drt = ct[keep][1:] - ct[keep][:-1], rt[keep][1:] - rt[keep][:-1]
        ghz = (dct / drt * 0.1)
        return float(ghz.median()), float(ghz.min()), float(ghz.max())


class Power(bench.PowerSampler):
    def _run(self):
        while not self._stop.is_set():
            self.rows.append((time.perf_counter(), [self._read(f) for f in self.files]))
            self._stop.wait(0.02)

    def mean_between(self, t_from, t_to=float("inf")):
        self._stop.set()
        v = [r[1][0] for r in self.rows if t_from <= r[0] <= t_to and r[1] and r[1][0] is not None]
        return (sum(v) / len(v), len(v)) if v else (None, 0)

    def mean_after(self, t_from):
        return self.mean_between(t_from)


def replay(calls, seconds, clock, label, repeat_each=1):
    """Two replays: the first WITHOUT the clock probe gives ms, W and J; a second, shorter one with the probe resident gives the GHz
    column only.  Round 6 found the probe to be anything but free for launches of ONE round: its single wave keeps one CU from taking a
    256-thread-x-256-VGPR workgroup, the XCD that CU belongs to then serves its 32 workgroups of a 252-tile launch in TWO rounds, and
    every level-2 / level-3 family read 35-40 % low and 300-400 W under the cap in round 5's tables (conv L2: 806 TFLOP/s at 990 W with
    the probe, 1 337 at 1 397 W without; profiles/r06/energy_table_probe_artifact.txt)."""
    r = _replay(calls, seconds, Clock(0, False), label, repeat_each)
    if r is not None and clock.lib is not None:
        c = _replay(calls, min(seconds, 1.2), clock, label, repeat_each)
        r["clock"], r["ms_with_probe"] = c["clock"], c["ms"]
    return r


def _replay(calls, seconds, clock, label, repeat_each=1):
    if not calls:
        return None
    fns = [(getattr(hip._lib.real, n), [k[0] for k in kept]) for n, kept in calls for _ in range(repeat_each)]
    for fn, args in fns:                                        # one warm pass
        fn(*args)
    torch.cuda.synchronize()
    clock.start()
    pw = Power(0).start()
    cur = torch.cuda.current_stream()                           # NEVER a device-wide synchronize from here on: it would wait for the probe
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    passes, pending = 0, []
    ev0.record()
    while time.perf_counter() - t0 < seconds:
        for fn, args in fns:
            fn(*args)
        passes += 1
        e = torch.cuda.Event(); e.record(); pending.append(e)
        if len(pending) > 2:
            pending.pop(0).synchronize()
    ev1.record()
    ev1.synchronize()
    t1 = time.perf_counter()
    ms = ev0.elapsed_time(ev1) / passes
    skip = min(0.6, seconds / 3)
    w, nw = pw.mean_between(t0 + skip, t1)
    ck = clock.finish(skip, t1 - t0)
    return dict(label=label, launches=len(calls), ms=ms / repeat_each, W=w, samples=nw, clock=ck, passes=passes)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--workload", default="L"); ap.add_argument("--seconds", type=float, default=2.5)
    ap.add_argument("--json", default=None)
    ap.add_argument("--only", default=None, help="substring: replay only the families whose label contains it (and skip the whole iteration)")
    ap.add_argument("--repeat-each", type=int, default=1,
                    help="issue every launch of a family this many times in a row: all but the first find their operands where the "
                         "previous launch left them (L2 / Infinity Cache), as a launch inside the clip finds its producer's output")
    ap.add_argument("--no-probe", action="store_true", help="no resident clock-probe wave beside the replays (GHz column: n/a)")
    a = ap.parse_args()
    H, W = bench.WORKLOADS[a.workload]
    dev = torch.device("cuda:0")
    unet = UNetSpatioTemporalConditionControlNetModel(**bench.SVD).init_random_(seed=100, device=dev)
    cn = ControlNetSDVModel(**bench.SVD).init_random_(seed=200, device=dev)
    sched = EulerDiscreteScheduler(**SVD_SCHEDULER_CONFIG)
    pipe = StableVideoDiffusionPipelineControlNet(unet=unet, controlnet=cn, scheduler=sched)
    sched.set_timesteps(25)
    clip = bench.synth_clip(H, W, 14, 1024, 1234, dev, sched.init_noise_sigma)
    pipe.denoise(*clip, num_inference_steps=2)                  # warm-up: packs, condition-encoder cache, allocator
    torch.cuda.synchronize()
    hip.lib()
    rec = Recorder(hip._lib)
    hip._lib = rec
    keep = []
    orig_ptr = torch.Tensor.data_ptr

    def keeping_ptr(self):
        keep.append(self)
        return orig_ptr(self)
    torch.Tensor.data_ptr = keeping_ptr
    # the SECOND iteration of a 2-step schedule is recorded (the first one's launches are dropped by the callback): the loop's
    # launches are the same in every iteration, only the data differ
    rec.on = True

    def cb(pipe_, i, t, kw):
        if i == 0:
            rec.calls.clear()
            keep.clear()
        return {}
    pipe.denoise(*clip, num_inference_steps=2, callback_on_step_end=cb)
    rec.on = False
    torch.Tensor.data_ptr = orig_ptr
    torch.cuda.synchronize()
    calls = list(rec.calls)
    fam = {}
    for name, kept in calls:
        label, fl, inplace = classify(name, kept)
        fam.setdefault(label, []).append((name, kept, fl, inplace))
    print(f"# energy per kernel family, workload {a.workload} (14 x {H} x {W}), one loop iteration = {len(calls)} launches recorded; "
          f"{torch.cuda.get_device_name(0)}; kept alive {sum(t.numel() * t.element_size() for t in {id(t): t for t in keep}.values()) / 2**30:.1f} GiB")
    clock = Clock(a.seconds, not a.no_probe)
    # idle
    time.sleep(0.5)
    pw = Power(0).start(); t0 = time.perf_counter(); time.sleep(1.0); idle, _ = pw.mean_after(t0)
    print(f"# idle socket power {idle:.0f} W" if idle else "# no power sensor")
    rows = []
    whole = None
    if not a.only:
        whole = replay([(n, k) for n, k in calls], a.seconds, clock, "WHOLE ITERATION (eager replay, one stream)")
        rows.append(whole)
    if a.repeat_each > 1:
        print(f"# every launch issued {a.repeat_each} x in a row; ms/iter, J/iter are per ONE pass over the family's launches")
    for label in sorted(fam, key=lambda l: -sum(1 for _ in fam[l])):
        if a.only and a.only not in label:
            continue
        sel = [(n, k) for n, k, fl, ip in fam[label] if not ip or label.startswith("other")]
        if label.startswith("other"):
            continue
        r = replay(sel, a.seconds, clock, label, a.repeat_each)
        if r:
            r["flops"] = sum(fl or 0.0 for n, k, fl, ip in fam[label] if not ip)
            r["skipped_inplace"] = sum(1 for n, k, fl, ip in fam[label] if ip)
            rows.append(r)
    att = [r for r in rows if r["label"] == "spatial attention"]
    if att:                                                     # 4 N heads S^2 64 per launch: N = 28 frames, S and heads per level
        fl = 0.0
        for n, kept in [(n, k) for n, k in calls if n == "pt_attn_spatial_f16"]:
            Nimg, S, heads, hd = (kept[i][0] for i in (6, 7, 8, 9))    # (qkv, ld, koff, voff, out, ldo, Nimg, S, heads, head_dim, ...)
            fl += 4.0 * Nimg * heads * S * S * hd
        att[0]["flops"] = fl
    hdr = f"{'family':46s} {'launch':>6s} {'ms/iter':>8s} {'TFLOP/s':>8s} {'W':>6s} {'GHz med (min-max)':>20s} {'J/iter':>7s} {'pJ/flop':>8s}"
    print(hdr)
    tot_ms = tot_j = 0.0
    for r in rows:
        j = (r["W"] or 0.0) * r["ms"] * 1e-3
        fl = r.get("flops") or 0.0
        ck = r["clock"]
        slow = f"   [with the probe resident: {r['ms_with_probe']:.3f} ms]" if r.get("ms_with_probe", 0) > 1.1 * r["ms"] else ""
        print(f"{r['label']:46s} {r['launches']:6d} {r['ms']:8.3f} {(fl / r['ms'] / 1e9 if fl else 0):8.1f} {(r['W'] or 0):6.0f} "
              f"{(f'{ck[0]:.2f} ({ck[1]:.2f}-{ck[2]:.2f})' if ck else 'n/a'):>20s} {j:7.2f} {(j / fl * 1e12 if fl else 0):8.3f}{slow}"
              + (f"   [{r['skipped_inplace']} in-place launches left out]" if r.get("skipped_inplace") else ""))
        if r is not whole:
            tot_ms += r["ms"]; tot_j += j
    print(f"{'sum of the families alone':46s} {'':6s} {tot_ms:8.3f} {'':8s} {'':6s} {'':20s} {tot_j:7.2f}")
    if a.json:
        with open(a.json, "w") as f:
            json.dump(dict(workload=a.workload, idle_W=idle, rows=rows, csrc_sha256=hip.source_digest()), f, indent=1, default=str)


if __name__ == "__main__":
    main()
