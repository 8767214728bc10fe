"""The trajectory-map rasteriser on the MI355X: pt_rasterize_tracks through posetraj_amd.trajectory against
oracle/raster.py on the tracks of the reference-run fixture (tests/golden/tracks.npz), pixel for pixel."""
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


# ------------------------------------------------------------------------------------------------- trajectory rasteriser
@pytest.mark.parametrize("name", ["a", "b", "d"])
@pytest.mark.parametrize("mode", ["inference", "dataset"])
def test_trajectory_rasteriser_against_the_restated_primitives(dev, golden, name, mode):
    """pt_rasterize_tracks vs oracle/raster.py (numpy, integer arithmetic) on the tracks of the reference-run fixture: every
    pixel of every map identical - line / disc coverage, overwrite order, per-map vs per-track channel flip, black last map."""
    from oracle import raster as R
    from posetraj_amd import trajectory as T
    g = golden("tracks")
    keys = [str(k) for k in g[f"{name}_keys"]]
    tracks = {k: g[f"{name}_tracks"][i].tolist() for i, k in enumerate(keys)}
    size, osz = [int(v) for v in g[f"{name}_size"]], tuple(int(v) for v in g[f"{name}_original_size"])
    nf = 14 if mode == "inference" else 6
    want = R.trajectory_maps(tracks, size, osz, num_frames=nf, mode=mode, start=0 if mode == "inference" else 2)
    got = T.trajectory_maps(tracks, size, osz, num_frames=nf, mode=mode, start=0 if mode == "inference" else 2, device=dev,
                            dtype=torch.float32).cpu().numpy()
    assert got.shape == want.shape == (nf, 3, size[0], size[1])
    assert np.array_equal(got, want)
    assert float(got[-1].max()) == -1.0 and float(got[0].max()) == 1.0
    h16 = T.trajectory_maps(tracks, size, osz, num_frames=nf, mode=mode, start=0 if mode == "inference" else 2, device=dev)
    assert h16.dtype == torch.float16 and np.array_equal(h16.float().cpu().numpy(), want)


def test_trajectory_rasteriser_edge_cases(dev):
    from posetraj_amd import trajectory as T
    empty = T.trajectory_maps({}, [32, 48], (64, 96, 3), num_frames=14, device=dev)
    assert tuple(empty.shape) == (14, 3, 32, 48) and float(empty.max()) == -1.0                    # no tracks: all maps black
    still = {"0": [[10, 10]] * 14}                                                                  # a point that never moves
    m = T.trajectory_maps(still, [32, 48], (32, 48, 3), num_frames=14, device=dev, dtype=torch.float32).cpu()
    assert int((m[0, 1] > 0).sum()) == 29 and float(m[0, 0].max()) == -1.0                         # only the green disc
    with pytest.raises(ValueError, match="need 14 points"):
        T.trajectory_maps({"0": [[1, 1]] * 5}, [32, 48], (32, 48, 3), num_frames=14, device=dev)
