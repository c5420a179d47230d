"""GPU: b3gs_knn_mean_dist2 (the reference's simple_knn distCUDA2) against an exact k-d tree (scipy) and the
create_from_pcd initialisation built on it."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _exact(pts):
    from scipy.spatial import cKDTree
    d, _ = cKDTree(pts.astype(np.float64)).query(pts.astype(np.float64), k=4)
    return (d[:, 1:] ** 2).mean(axis=1)


@pytest.mark.parametrize("P,kind", [(5000, "uniform"), (1025, "clustered"), (100000, "uniform"), (4, "uniform"), (2049, "plane")])
def test_matches_exact_knn(P, kind):
    from binocular3dgs_amd.init_points import knn_mean_dist2
    rng = np.random.default_rng(P)
    pts = rng.uniform(-3, 5, size=(P, 3)).astype(np.float32)
    if kind == "clustered":
        pts = (pts * 0.01 + rng.integers(0, 4, size=(P, 1)) * 7.0).astype(np.float32)
    if kind == "plane":
        pts[:, 2] = 1.5
    got = knn_mean_dist2(torch.from_numpy(pts).cuda()).cpu().numpy()
    np.testing.assert_allclose(got, _exact(pts), rtol=2e-5, atol=1e-9)


def test_duplicates_and_tiny_sets():
    from binocular3dgs_amd.init_points import knn_mean_dist2
    pts = torch.tensor([[0.0, 0, 0], [0, 0, 0], [1, 0, 0], [0, 2, 0], [0, 0, 3]]).cuda()
    got = knn_mean_dist2(pts).cpu().numpy()
    np.testing.assert_allclose(got[:2], [(0 + 1 + 4) / 3.0] * 2, rtol=1e-6)      # the twin is excluded by index only
    three = knn_mean_dist2(pts[:3]).cpu().numpy()
    assert np.all(three > 1e30)                                                  # fewer than 3 neighbours: FLT_MAX terms, as upstream


def test_create_from_points_follows_create_from_pcd():
    from binocular3dgs_amd.init_points import SH_C0, create_from_points
    rng = np.random.default_rng(0)
    pts = rng.normal(size=(3000, 3)).astype(np.float32)
    rgb = rng.uniform(size=(3000, 3)).astype(np.float32)
    m = create_from_points(pts, rgb, sh_degree=3)
    assert m._features_dc.shape == (3000, 1, 3) and m._features_rest.shape == (3000, 15, 3) and m.active_sh_degree == 0
    np.testing.assert_allclose(m._features_dc.detach().cpu().numpy()[:, 0, :], (rgb - 0.5) / SH_C0, rtol=1e-6, atol=1e-7)
    assert float(m._features_rest.abs().max()) == 0
    ref = np.log(np.sqrt(np.maximum(_exact(pts), 1e-7)))
    np.testing.assert_allclose(m._scaling.detach().cpu().numpy(), np.repeat(ref[:, None], 3, 1), rtol=1e-4, atol=1e-5)
    np.testing.assert_allclose(torch.sigmoid(m._opacity).detach().cpu().numpy(), 0.1, rtol=1e-5)
    assert torch.equal(m._rotation.detach().cpu(), torch.tensor([[1.0, 0, 0, 0]]).repeat(3000, 1))
