"""CPU, world_size 2 over gloo: the view-sharded step (binocular3dgs_amd/step.py).  The oracle
stands in for the GPU rasterizer (tests/cpu_render.py); what is tested is the host logic: pair
sharding, in-place accumulation into the flat gradient slab, ONE all-reduce, identical replicas."""
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


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _build(P=300, W=48, H=32):
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import binocular3dgs_amd.render as R
    from binocular3dgs_amd import synth
    from cpu_render import OracleRasterizer
    R.GaussianRasterizer = OracleRasterizer      # test-only: oracle instead of the HIP rasterizer
    model = synth.synth_model(P, seed=4, device="cpu", width=W, height=H)
    with torch.no_grad():
        model._scaling += 1.0                      # bigger splats for the tiny image
    pairs = synth.synth_view_set(W, H)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(10 + i)) for i in range(len(pairs))]
    return model, pairs, gts, R.render


def _loss_fn(gts):
    from binocular3dgs_amd.loss import binocular_loss

    def fn(i, cam, pkg, spkg, t):
        total, _ = binocular_loss(pkg["render"], pkg["rendered_depth"], pkg["rendered_alpha"], gts[i],
                                  shifted_image=spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t,
                                  bg_mask=torch.ones(1, *gts[i].shape[1:]) * 0.1)
        return total
    return fn


def _worker(rank, world, port, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.set_num_threads(2)
    os.environ["OMP_NUM_THREADS"] = "2"
    from binocular3dgs_amd.step import ViewShardedStep, shard_pairs
    model, pairs, gts, render = _build()
    mine = shard_pairs(len(pairs), rank, world)
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-15)
    st = ViewShardedStep(model, [pairs[i] for i in mine], torch.zeros(3), optimizer=opt, render_fn=render)
    st.step(loss_fn=lambda i, cam, pkg, spkg, t: _loss_fn([gts[j] for j in mine])(i, cam, pkg, spkg, t))
    # densification statistics: every rank contributes its own pairs, then one sync
    model.init_densification_stats()
    model.denom += float(len(mine))
    model.xyz_gradient_accum += float(rank + 1)
    model.max_radii2D += float(10 * (rank + 1))
    st.sync_densify_stats()
    assert float(model.denom[0]) == 3.0 and float(model.xyz_gradient_accum[0]) == 3.0
    assert float(model.max_radii2D[0]) == 20.0
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), grad=st.slab.flat.numpy(),
             params=torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy(), mine=np.array(mine))
    dist.destroy_process_group()


def test_shard_pairs_partition():
    from binocular3dgs_amd.step import shard_pairs
    for world in (1, 2, 3, 4, 8):
        seen = sorted(i for r in range(world) for i in shard_pairs(6, r, world))
        assert seen == list(range(6))
    assert shard_pairs(3, 0, 2) == [0, 2] and shard_pairs(3, 1, 2) == [1]


def test_flat_slab_accumulates_in_place():
    from binocular3dgs_amd.step import FlatGradSlab
    a = torch.nn.Parameter(torch.ones(5, 3))
    b = torch.nn.Parameter(torch.ones(4))
    slab = FlatGradSlab([a, b])
    ptr = slab.flat.data_ptr()
    for _ in range(3):
        ((a * 2).sum() + (b * 3).sum()).backward()
    assert a.grad.data_ptr() == ptr and slab.flat.data_ptr() == ptr
    np.testing.assert_array_equal(slab.flat.numpy(), np.concatenate([np.full(15, 6.0), np.full(4, 9.0)]).astype(np.float32))
    slab.zero()
    assert float(slab.flat.abs().sum()) == 0 and a.grad.data_ptr() == ptr


@pytest.mark.timeout(600)
def test_two_ranks_equal_one_rank(tmp_path):
    world, port = 2, _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert sorted(list(r0["mine"]) + list(r1["mine"])) == [0, 1, 2]
    # replicas hold identical summed gradients and identical parameters after the step
    np.testing.assert_array_equal(r0["grad"], r1["grad"])
    np.testing.assert_array_equal(r0["params"], r1["params"])
    # and they equal a single process that renders all three pairs
    from binocular3dgs_amd.step import ViewShardedStep
    model, pairs, gts, render = _build()
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, eps=1e-15)
    st = ViewShardedStep(model, pairs, torch.zeros(3), optimizer=opt, render_fn=render)
    st.step(loss_fn=_loss_fn(gts))
    ref = st.slab.flat.numpy()
    assert np.abs(ref).max() > 0
    np.testing.assert_allclose(r0["grad"], ref, rtol=1e-5, atol=1e-9)
    np.testing.assert_allclose(r0["params"], torch.cat([p.detach().reshape(-1) for p in model.parameters()]).numpy(),
                               rtol=1e-6, atol=1e-7)


def test_rank_without_pairs_contributes_zero_gradients():
    """More ranks than (input, shifted) pairs: the idle rank still owns a zeroed slab for the all-reduce."""
    from binocular3dgs_amd.step import ViewShardedStep, shard_pairs
    model, pairs, gts, render = _build()
    assert shard_pairs(3, 3, 4) == []
    st = ViewShardedStep(model, [], torch.zeros(3), render_fn=render)
    st.slab.flat.fill_(7.0)
    assert st.compute_grads(loss_fn=lambda *a: None) == 0
    assert float(st.slab.flat.abs().max()) == 0.0
