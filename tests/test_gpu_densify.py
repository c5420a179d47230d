"""GPU: densify / clone / split / prune + Adam-state surgery (b3gs_densify_*) against the golden
fixture produced by the reference's own densify_and_prune (tests/golden/make_golden_densify.py), for both
optimiser flavours, and the model / step object staying usable afterwards."""
import os

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")
# fixture names (reference param-group order) -> model attributes / ABI order
REF = {"_xyz": "xyz", "_features_dc": "f_dc", "_features_rest": "f_rest", "_scaling": "scaling", "_rotation": "rotation",
       "_opacity": "opacity"}
ATTRS = ("_xyz", "_features_dc", "_features_rest", "_scaling", "_rotation", "_opacity")


def _load(case):
    from binocular3dgs_amd.gaussian_model import GaussianModel
    g = np.load(os.path.join(GOLD, "densify.npz"))
    t = lambda k: torch.from_numpy(g[k]).cuda()  # noqa: E731
    m = GaussianModel.from_tensors(t(f"{case}_in_xyz"), t(f"{case}_in_f_dc"), t(f"{case}_in_f_rest"), t(f"{case}_in_scaling"),
                                   t(f"{case}_in_rotation"), t(f"{case}_in_opacity"), sh_degree=1, device="cuda")
    m.init_densification_stats()
    m.xyz_gradient_accum, m.denom = t(f"{case}_accum"), t(f"{case}_denom")
    m.max_radii2D = t(f"{case}_max_radii2D")
    P = m.get_xyz.shape[0]
    # the reference selected rows in index order; its noise rows are (k, j-th selected): rebuild [2,P,3]
    thr, min_op, extent, pd, size = [float(x) for x in g[f"{case}_scalars"]]
    grads = np.nan_to_num(g[f"{case}_accum"][:, 0] / g[f"{case}_denom"][:, 0], nan=0.0)
    smax = np.exp(g[f"{case}_in_scaling"]).max(1)
    sel = np.nonzero((grads >= thr) & (smax > pd * extent))[0]
    nz = g[f"{case}_noise"]
    assert nz.shape[0] == 2 * len(sel)
    noise = np.zeros((2, P, 3), np.float32)
    noise[0, sel], noise[1, sel] = nz[:len(sel)], nz[len(sel):]
    return g, m, torch.from_numpy(noise).cuda(), (thr, min_op, extent, None if size < 0 else size, pd)


@pytest.mark.parametrize("case", ["a", "b"])
@pytest.mark.parametrize("opt_kind", ["fused", "torch"])
def test_matches_the_reference_densify_and_prune(case, opt_kind):
    from binocular3dgs_amd.densify import densify_and_prune
    from binocular3dgs_amd.step import FusedAdam
    g, m, noise, (thr, min_op, extent, size, pd) = _load(case)
    lrs = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]
    if opt_kind == "fused":
        opt = FusedAdam(m.parameters(), lrs, eps=1e-15)
        off = 0
        for a, p in zip(ATTRS, m.parameters()):
            opt.exp_avg[off:off + p.numel()] = torch.from_numpy(g[f"{case}_in_{REF[a]}_exp_avg"]).cuda().reshape(-1)
            opt.exp_avg_sq[off:off + p.numel()] = torch.from_numpy(g[f"{case}_in_{REF[a]}_exp_avg_sq"]).cuda().reshape(-1)
            off += p.numel()
    else:
        opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(m.parameters(), lrs)], eps=1e-15)
        for a, p in zip(ATTRS, m.parameters()):
            opt.state[p] = {"step": torch.tensor(2.0), "exp_avg": torch.from_numpy(g[f"{case}_in_{REF[a]}_exp_avg"]).cuda(),
                            "exp_avg_sq": torch.from_numpy(g[f"{case}_in_{REF[a]}_exp_avg_sq"]).cuda()}
    newP = densify_and_prune(m, opt, thr, min_op, extent, size, percent_dense=pd, noise=noise)
    assert newP == g[f"{case}_out_xyz"].shape[0] and newP != g[f"{case}_in_xyz"].shape[0]
    off = 0
    for a in ATTRS:
        got = getattr(m, a).detach().cpu().numpy()
        ref = g[f"{case}_out_{REF[a]}"]
        assert got.shape == ref.shape, a
        if a in ("_xyz", "_scaling"):            # children: R(q)(s*noise)+xyz and log(s/1.6) -- float math
            np.testing.assert_allclose(got, ref, rtol=2e-5, atol=2e-6, err_msg=a)
        else:                                     # pure copies
            np.testing.assert_array_equal(got, ref, err_msg=a)
        n = got.size
        if opt_kind == "fused":
            gm, gv = opt.exp_avg[off:off + n].cpu().numpy().reshape(ref.shape), opt.exp_avg_sq[off:off + n].cpu().numpy().reshape(ref.shape)
        else:
            st = opt.state[getattr(m, a)]
            gm, gv = st["exp_avg"].cpu().numpy(), st["exp_avg_sq"].cpu().numpy()
            assert float(st["step"]) == 2.0
        np.testing.assert_array_equal(gm, g[f"{case}_out_{REF[a]}_exp_avg"], err_msg=a + " exp_avg")
        np.testing.assert_array_equal(gv, g[f"{case}_out_{REF[a]}_exp_avg_sq"], err_msg=a + " exp_avg_sq")
        off += n
    stats = g[f"{case}_out_stats"]
    assert m.xyz_gradient_accum.shape == (newP, 1) and float(m.xyz_gradient_accum.abs().max()) == 0 == float(stats[0].max())
    assert float(m.denom.abs().max()) == 0 and float(m.max_radii2D.abs().max()) == 0 == float(stats[2].max())


def test_opacity_decay_matches_the_reference():
    from binocular3dgs_amd.step import FusedAdam
    g = np.load(os.path.join(GOLD, "densify.npz"))
    o = torch.nn.Parameter(torch.from_numpy(g["decay_in"]).cuda())
    o.grad = torch.zeros_like(o)
    opt = FusedAdam([o], [0.05], eps=1e-15, opacity_decay=0.995, opacity_index=0)
    opt.step()
    torch.cuda.synchronize()
    np.testing.assert_allclose(o.detach().cpu().numpy(), g["decay_out"], rtol=2e-5, atol=2e-6)


def test_step_object_survives_densification():
    """ViewShardedStep.densify_and_prune: statistics gathered by the fused backward drive a densification, then
    slab / slots are re-created for the new P and training continues."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam, ViewShardedStep
    W, H, P = 160, 120, 6000
    model = synth.synth_model(P, seed=2, device="cuda", width=W, height=H)
    model.init_densification_stats()
    pairs = synth.synth_view_set(W, H, device="cuda")
    bg = torch.zeros(3, device="cuda")
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=1, device="cuda")
    opt = FusedAdam(model.parameters(), [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05], eps=1e-15)
    fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
    st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr)
    fn = lambda i, pkg, spkg: [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]  # noqa: E731
    for _ in range(3):
        st.step(pair_grad_fn=fn)
    assert float(model.denom.max()) == 9.0          # 3 primary views x 3 steps
    thr = float((model.xyz_gradient_accum / model.denom.clamp(min=1)).median())
    newP = st.densify_and_prune(thr, 0.005, 5.0, generator=torch.Generator(device="cuda").manual_seed(0))
    assert newP != P and model.get_xyz.shape[0] == newP and fr.P == newP
    assert opt.exp_avg.numel() == sum(p.numel() for p in model.parameters())
    for _ in range(2):
        st.step(pair_grad_fn=fn)
    torch.cuda.synchronize()
    assert all(torch.isfinite(p).all() for p in model.parameters())
    assert float(model.denom.max()) == 6.0 and not fr.overflowed()
