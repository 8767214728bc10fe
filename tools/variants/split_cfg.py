"""Measured and dropped (round 4, VERDICT r03 #4a; moved out of the product in round 5): the two CFG halves of a loop iteration as
two INDEPENDENT network evaluations on two HIP streams.  +4.8 % clip time on two boxes (profiles/r04/clip_ab_split_cfg.txt,
_box2.txt): half-batch GEMMs read every weight panel twice and levels 2-3 fill half the chip per launch.

    from tools.variants.split_cfg import networks_split
    pipe.denoise(..., _networks=networks_split)          # bit-compatible with the full-batch loop up to fp32 summation order

It rides on the product's one-CFG-half forward (``half=`` of ``_encode`` / ``_features``: Q3's batch-interleaved temporal context as
an index into a two-row table), which tests/test_model_gpu.py keeps covered through this function."""
import torch


def networks_split(self, sample, t, emb_, cond_, cam_, added_time_ids, controlnet_cond_scale):
    """EXPERIMENT (VERDICT r03 #4a; results in DESIGN section 8): the two CFG halves are independent evaluations (GroupNorm
    is per sample, attention per frame / position; Q3's interleave is an index into a two-row table, handled by
    ``half=``), so each half runs its ControlNet + U-Net on its own HIP stream, the second one starting with the
    U-Net encoder while the first starts with the ControlNet.  Same kernels on half the rows; bit-identical results."""
    dev = sample.device
    F = sample.shape[1]
    main = torch.cuda.current_stream(dev)
    if self._side_stream is None or self._side_stream.device != dev:
        self._side_stream = torch.cuda.Stream(device=dev)
    side = self._side_stream
    Bh = sample.shape[0] // 2
    pred = torch.empty((2 * Bh, F, sample.shape[3], sample.shape[4], 4), dtype=torch.float32, device=dev)
    cam2 = cam_ if self.controlnet.config.camera else None
    if cond_ is not None:
        self.controlnet._cond_embedding(cond_, cam2)                       # once per clip, for the whole batch, before the fork
    side.wait_stream(main)

    def one(hh, unet_first):
        sl = slice(hh * Bh, (hh + 1) * Bh)
        if unet_first:
            enc = self.unet._encode(sample[sl], t, emb_, added_time_ids[sl], half=hh)
            taps, xm = self.controlnet._features(sample[sl], t, emb_, added_time_ids[sl], cond_, cam2, half=hh)
        else:
            taps, xm = self.controlnet._features(sample[sl], t, emb_, added_time_ids[sl], cond_, cam2, half=hh)
            enc = self.unet._encode(sample[sl], t, emb_, added_time_ids[sl], half=hh)
        self.controlnet._accumulate_into(taps, xm, controlnet_cond_scale, enc["skips"],
                                         self.unet._multiplicity(enc, len(taps)), enc["x"])
        self.unet._decode(enc, None, None, return_dict=False, residuals_added=True, out_f32=True,
                          out=pred[sl].view(-1, 4))
    one(0, False)
    with torch.cuda.stream(side):
        one(1, True)
    main.wait_stream(side)
    pred.record_stream(main)
    return pred

