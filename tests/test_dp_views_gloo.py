"""CPU, gloo, world 2 / 4 / 8: VIEW-granular sharding of one iteration's views (step.ViewShardedStep.from_global),
split (input, shifted) pairs exchanging the shifted image and its gradient point to point, reduce-scatter ->
sharded Adam -> all-gather (step.ShardedAdam), densification statistics.  The oracle stands in for the GPU
rasterizer and a torch expression for the one-launch Adam kernel (tests only: the product has no CPU path); what is
tested is the host logic.  N ranks must equal ONE process that renders all views: `denom` bit for bit, parameters
to 1e-6."""
import math
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LRS = [1.6e-3, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]
STEPS = 2


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def torch_adam_impl(segs, step_count, betas, eps, decay, opacity_seg, decay_first):
    """torch statement of csrc/optim.hip::adam_kernel (torch.optim.Adam arithmetic + optional opacity decay)."""
    t = int(step_count.item()) + 1
    b1, b2 = betas
    bc1, bc2s = 1.0 - b1 ** t, math.sqrt(1.0 - b2 ** t)
    with torch.no_grad():
        for k, (p, g, m, v, lr) in enumerate(segs):
            m.mul_(b1).add_(g, alpha=1.0 - b1)
            v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
            delta = (lr / bc1) * (m / (v.sqrt() / bc2s + eps))
            dec = decay > 0 and k == opacity_seg
            if dec and decay_first:
                o = torch.sigmoid(p) * decay
                p.copy_(torch.log(o / (1 - o)) - delta)
            else:
                p.sub_(delta)
                if dec:
                    o = torch.sigmoid(p) * decay
                    p.copy_(torch.log(o / (1 - o)))
        step_count += 1


def _build(views=6, P=260, W=48, H=32):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import binocular3dgs_amd.render as R
    from binocular3dgs_amd import synth
    from cpu_render import OracleRasterizer
    R.GaussianRasterizer = OracleRasterizer      # test-only: oracle instead of the HIP rasterizer
    model = synth.synth_model(P, seed=4, device="cpu", width=W, height=H)
    with torch.no_grad():
        model._scaling += 1.0                      # bigger splats for the tiny image
    model.init_densification_stats()
    if views == 8:      # config 5: eight input views without partners
        pairs = [(c, None, 0.0) for c in synth.synth_cameras(W, H, yaws=synth.YAWS_8)]
    else:
        pairs = synth.synth_view_set(W, H)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(10 + i)) for i in range(len(pairs))]
    return model, pairs, gts, R.render


def _loss_fn(gts):
    from binocular3dgs_amd.loss import binocular_loss

    def fn(i, cam, pkg, spkg, t):
        total, _ = binocular_loss(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[i],
                                  shifted_image=None if spkg is None else spkg["render"], focal_x=cam.get_focal()[0],
                                  trans_dist=t, bg_mask=torch.ones(1, *gts[i].shape[1:]) * 0.1)
        return total
    return fn


def _worker(rank, world, port, out_dir, views):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(1)
    os.environ["OMP_NUM_THREADS"] = "1"
    from binocular3dgs_amd.step import ShardedAdam, ViewShardedStep
    model, pairs, gts, render = _build(views)
    opt = ShardedAdam(model.parameters(), LRS, eps=1e-15, adam_impl=torch_adam_impl)
    st = ViewShardedStep.from_global(model, pairs, torch.zeros(3), optimizer=opt, render_fn=render)
    assert st.slab.flat.numel() == opt.padded_numel and opt.padded_numel % world == 0
    for _ in range(STEPS):
        st.step(loss_fn=_loss_fn(gts))
    st.sync_densify_stats()
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"),
             params=torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy(),
             denom=model.denom.numpy(), accum=model.xyz_gradient_accum.numpy(), radii=model.max_radii2D.numpy(),
             nviews=np.array(len(st.views)), split=np.array(sum(v.peer is not None for v in st.views)))
    dist.destroy_process_group()


def _single_process(views):
    from binocular3dgs_amd.step import ViewShardedStep
    model, pairs, gts, render = _build(views)
    opt = torch.optim.Adam([{"params": [p], "lr": lr} for p, lr in zip(model.parameters(), LRS)], eps=1e-15)
    st = ViewShardedStep(model, pairs, torch.zeros(3), optimizer=opt, render_fn=render)
    for _ in range(STEPS):
        st.step(loss_fn=_loss_fn(gts))
    return (torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy(), model.denom.numpy(),
            model.xyz_gradient_accum.numpy(), model.max_radii2D.numpy())


def test_assign_views_blocks():
    from binocular3dgs_amd.step import assign_views
    six = [True, True, True]
    assert assign_views(six, 1) == [[(0, 0), (0, 1), (1, 0), (1, 1), (2, 0), (2, 1)]]
    assert assign_views(six, 2) == [[(0, 0), (0, 1), (1, 0)], [(1, 1), (2, 0), (2, 1)]]
    assert assign_views(six, 3) == [[(0, 0), (0, 1)], [(1, 0), (1, 1)], [(2, 0), (2, 1)]]            # no pair split
    assert assign_views(six, 4) == [[(0, 0), (0, 1)], [(1, 0), (1, 1)], [(2, 0)], [(2, 1)]]          # one pair split
    assert [len(b) for b in assign_views(six, 8)] == [1, 1, 1, 1, 1, 1, 0, 0]
    assert assign_views([False] * 8, 8) == [[(i, 0)] for i in range(8)]
    for world in (1, 2, 3, 4, 5, 6, 7, 8):
        flat = [v for b in assign_views(six, world) for v in b]
        assert flat == assign_views(six, 1)[0]
        sizes = [len(b) for b in assign_views(six, world)]
        assert max(sizes) - min(sizes) <= 1 and sizes == sorted(sizes, reverse=True)


def test_sharded_adam_segments_cover_the_flat_buffer_once():
    from binocular3dgs_amd.step import ShardedAdam
    params = [torch.nn.Parameter(torch.randn(37, 3)), torch.nn.Parameter(torch.randn(37, 1, 3)),
              torch.nn.Parameter(torch.randn(37, 0, 3)), torch.nn.Parameter(torch.randn(37, 4))]
    before = [p.detach().clone() for p in params]
    opt = ShardedAdam(params, [1e-2] * 4, adam_impl=torch_adam_impl)
    assert opt.world == 1 and opt.padded_numel % 64 == 0 and opt.padded_numel >= opt.numel == 37 * 10
    for p, b in zip(params, before):
        assert torch.equal(p.detach(), b) and (p.numel() == 0 or p.data_ptr() >= opt.pflat.data_ptr())
    seen = sum(e - s for _, s, e in opt.my_segments())
    assert seen == opt.numel
    # one step equals torch.optim.Adam on ordinary tensors
    from binocular3dgs_amd.step import FlatGradSlab
    slab = FlatGradSlab(params, opt.padded_numel)
    ref = [torch.nn.Parameter(b.clone()) for b in before]
    ropt = torch.optim.Adam(ref, lr=1e-2, eps=1e-15)
    for _ in range(3):
        for p, r in zip(params, ref):
            if p.numel():
                g = torch.randn_like(p)
                p.grad.copy_(g)
                r.grad = g.clone()
        opt.step(slab)
        ropt.step()
    for p, r in zip(params, ref):
        if p.numel():
            np.testing.assert_allclose(p.detach().numpy(), r.detach().numpy(), rtol=2e-6, atol=1e-7)


@pytest.mark.timeout(900)
@pytest.mark.parametrize("world,views", [(2, 6), (4, 6), (8, 6), (8, 8)])
def test_view_granular_ranks_equal_one_process(tmp_path, world, views):
    mp.spawn(_worker, args=(world, _free_port(), str(tmp_path), views), nprocs=world, join=True)
    rs = [np.load(tmp_path / f"rank{r}.npz") for r in range(world)]
    assert sum(int(r["nviews"]) for r in rs) == views
    expect_split = {(2, 6): 2, (4, 6): 2, (8, 6): 6, (8, 8): 0}[(world, views)]     # views whose partner is remote
    assert sum(int(r["split"]) for r in rs) == expect_split
    for r in rs[1:]:
        np.testing.assert_array_equal(r["params"], rs[0]["params"])                  # replicas stay identical
        np.testing.assert_array_equal(r["denom"], rs[0]["denom"])
    ref_params, ref_denom, ref_accum, ref_radii = _single_process(views)
    np.testing.assert_array_equal(rs[0]["denom"], ref_denom)                         # bit for bit
    np.testing.assert_array_equal(rs[0]["radii"], ref_radii)
    np.testing.assert_allclose(rs[0]["accum"], ref_accum, rtol=1e-5, atol=1e-9)
    rel = np.linalg.norm(rs[0]["params"] - ref_params) / np.linalg.norm(ref_params)
    assert rel < 1e-6, rel
    assert np.abs(rs[0]["params"] - ref_params).max() < 1e-5
