"""The forward half of the ControlNet training step on the MI355X against the reference run (tests/golden/train.npz):
network input, time ids, dropout, both losses.  The step with its backward: tests/test_backward_gpu.py."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def rel(a, b):
    a, b = torch.as_tensor(a).double().cpu(), torch.as_tensor(b).double().cpu()
    return float((a - b).norm() / b.norm().clamp_min(1e-30))


@pytest.fixture(scope="module")
def dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


# ------------------------------------------------------------------------------------------------- training objective (forward + loss)
@pytest.mark.parametrize("case", ["b1", "b1_nodrop", "b1_dropped"])
def test_training_step_forward_and_loss_against_the_reference_run(dev, golden, case):
    """posetraj_amd.training.controlnet_training_loss vs what the reference's own training-step statements produced
    (tests/golden/train.npz; scripts/train_svd_traj_VIPSeg_14.py:1275-1407): the network input built by pt_edm_train_input, the
    training-order added_time_ids, dropout, ControlNet + U-Net forward (incl. the one-frame "spatial" pass with per-frame
    residual slices), both losses from pt_edm_loss.  Forward and loss on the inference kernels (the step with its backward: test_backward_gpu.py)."""
    import contextlib, io
    from oracle import init as OI, nets as ON
    from posetraj_amd import ControlNetSDVModel, UNetSpatioTemporalConditionControlNetModel, training as T
    from tests.golden.make_golden import TRAIN_CE, TRAIN_CFG
    g = golden("train")
    k = case + "_"
    with contextlib.redirect_stdout(io.StringIO()):
        cn_o = OI.seeded_init_(ON.ControlNetSDVModel(**TRAIN_CFG, conditioning_embedding_out_channels=TRAIN_CE), seed=81).eval()
        un_o = OI.seeded_init_(ON.UNetSpatioTemporalConditionControlNetModel(**TRAIN_CFG), seed=82).eval()
    cn = ControlNetSDVModel(**TRAIN_CFG, conditioning_embedding_out_channels=TRAIN_CE).load_state_dict(cn_o.state_dict(), dev)
    un = UNetSpatioTemporalConditionControlNetModel(**TRAIN_CFG).load_state_dict(un_o.state_dict(), dev)
    t = lambda n: torch.from_numpy(g[k + n])
    drop = float(g[k + "drop"])
    r = T.controlnet_training_loss(cn, un, t("latents"), t("emb"), torch.tensor([127.0]), t("traj"), scaling_factor=0.18215,
                                   conditioning_dropout_prob=None if drop < 0 else drop, noise=t("noise"), sigmas=t("sigmas"),
                                   random_p=t("random_p"), ran_idx=int(g[k + "ran_idx"]))
    want_inp = t("inp_noisy_latents")
    assert float((r["inp_noisy_latents"].float().cpu() - want_inp).abs().max()) <= 6e-4 * float(want_inp.abs().max())   # one fp16 rounding
    assert np.array_equal(r["timesteps"].numpy(), g[k + "timesteps"])
    assert np.array_equal(r["added_time_ids"].cpu().numpy(), g[k + "added_time_ids"])
    assert np.array_equal(r["encoder_hidden_states"].cpu().numpy(), g[k + "ehs"])
    rp = rel(r["model_pred"], g[k + "model_pred"])
    rl, rs = abs(r["loss"] / float(g[k + "loss"]) - 1), abs(r["loss_spatial"] / float(g[k + "loss_spatial"]) - 1)
    print(f"training step {case}: model_pred rel-L2 {rp:.2e}; loss {r['loss']:.6f} vs {float(g[k + 'loss']):.6f} ({rl:.1e}), spatial ({rs:.1e})")
    assert rp < 2e-3 and rl < 5e-4 and rs < 5e-4            # measured: model_pred 0.7 - 1.3e-3, losses 2 - 7e-5


def test_training_loss_samples_its_own_draws(dev):
    from posetraj_amd import training as T
    s = T.rand_cosine_interpolated([64], generator=torch.Generator().manual_seed(1))
    assert tuple(s.shape) == (64,) and float(s.min()) > 1.9e-3 and float(s.max()) < 701 and bool((s[1:] < s[:-1]).all())   # stratified: monotone
    ids = T.train_add_time_ids(6, torch.tensor([127.0, 10.0]), 0.02, torch.float32, 2)
    assert ids.tolist() == [[6.0, 0.019999999552965164, 127.0], [6.0, 0.019999999552965164, 10.0]]
