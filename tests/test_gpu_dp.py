"""GPU, world_size 2 on ONE MI355X (two processes sharing the device, gloo carrying the device tensors: RCCL needs
one GPU per rank, which a 1-GPU box cannot give): the real HIP kernels under the view-sharded step -- pair
sharding, the flat gradient slab written by the fused backward, ONE all-reduce, replicated one-launch Adam, and the
pipelined range-major tail -- against a single process that renders all pairs."""
import os
import socket
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LRS = [1.6e-4, 2.5e-3, 1.25e-4, 5e-3, 1e-3, 0.05]


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _run(pairs_idx, steps, pipeline_ranges=0, dist_on=False):
    sys.path.insert(0, ROOT)
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam, ViewShardedStep
    W, H, P = 160, 120, 9000
    dev = "cuda"
    model = synth.synth_model(P, seed=13, device=dev, width=W, height=H)
    model.init_densification_stats()
    all_pairs = synth.synth_view_set(W, H, device=dev)
    pairs = [all_pairs[i] for i in pairs_idx]
    bg = torch.zeros(3, device=dev)
    grads = {i: synth.synth_pixel_grads(W, H, seed=20 + i, device=dev) for i in range(len(all_pairs))}
    opt = FusedAdam(model.parameters(), LRS, eps=1e-15)
    fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
    st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr, pipeline_ranges=pipeline_ranges)

    def fn(k, pkg, spkg):
        gc, gd, ga = grads[pairs_idx[k]]
        return [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]
    for _ in range(steps):
        st.step(pair_grad_fn=fn)
    if dist_on:
        st.sync_densify_stats()
    torch.cuda.synchronize()
    return (torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy(), model.denom.cpu().numpy())


def _worker(rank, world, port, out_dir, pipeline_ranges):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from binocular3dgs_amd.step import shard_pairs
    mine = shard_pairs(3, rank, world)
    params, denom = _run(mine, steps=2, pipeline_ranges=pipeline_ranges, dist_on=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=params, denom=denom, mine=np.array(mine))
    dist.destroy_process_group()


@pytest.mark.parametrize("pipeline_ranges", [0, 3])
def test_two_ranks_on_one_gpu_equal_one_process_with_all_pairs(tmp_path, pipeline_ranges):
    import torch.multiprocessing as mp
    ref_params, ref_denom = _run([0, 1, 2], steps=2)
    mp.spawn(_worker, args=(2, _free_port(), str(tmp_path), pipeline_ranges), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert sorted(list(r0["mine"]) + list(r1["mine"])) == [0, 1, 2]
    np.testing.assert_array_equal(r0["params"], r1["params"])            # replicas stay identical
    rel = np.linalg.norm(r0["params"] - ref_params) / np.linalg.norm(ref_params)
    assert rel < 1e-6, rel                                               # sum over ranks == sum over pairs (fp32 order)
    np.testing.assert_array_equal(r0["denom"], ref_denom)
    np.testing.assert_array_equal(r1["denom"], ref_denom)


def _run_views(world_rank, steps, sharded):
    """3 pairs through ViewShardedStep.from_global (view-granular blocks), fused rasterizer, the fused binocular loss
    (so that a split pair really exchanges the shifted image and its gradient), ShardedAdam or FusedAdam."""
    sys.path.insert(0, ROOT)
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.fused_loss import binocular_loss_fused_batch
    from binocular3dgs_amd.step import FusedAdam, ShardedAdam, ViewShardedStep
    W, H, P = 160, 120, 9000
    dev = "cuda"
    rank, world = world_rank
    model = synth.synth_model(P, seed=13, device=dev, width=W, height=H)
    with torch.no_grad():
        model._scaling += 0.5
    model.init_densification_stats()
    pairs = synth.synth_view_set(W, H, device=dev)
    bg = torch.zeros(3, device=dev)
    gts = [torch.rand(3, H, W, generator=torch.Generator().manual_seed(40 + i)).to(dev) for i in range(3)]
    opt = (ShardedAdam if sharded else FusedAdam)(model.parameters(), LRS, eps=1e-15)
    from binocular3dgs_amd.step import assign_views
    nloc = len(assign_views([True] * 3, world)[rank])
    fr = FusedRasterizer(model, W, H, num_slots=max(nloc, 1), want_means2D=False)
    st = ViewShardedStep.from_global(model, pairs, bg, rank=rank, world=world, optimizer=opt, fused=fr)

    def batch_loss(items):
        return binocular_loss_fused_batch(
            [dict(image=pkg["render"], depth=pkg["rendered_depth"], alpha=pkg["rendered_alpha"], gt_image=gts[i],
                  shifted_image=None if spkg is None else spkg["render"], focal_x=cam.get_focal()[0], trans_dist=t)
             for i, cam, pkg, spkg, t in items], unit_grad=True)
    for _ in range(steps):
        st.step(batch_loss_fn=batch_loss)
    st.sync_densify_stats()
    torch.cuda.synchronize()
    return (torch.cat([p.detach().reshape(-1) for p in model.parameters()]).cpu().numpy(), model.denom.cpu().numpy(),
            sum(v.peer is not None for v in st.views))


def _worker_views(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    params, denom, split = _run_views((rank, world), steps=2, sharded=True)
    np.savez(os.path.join(out_dir, f"rank{rank}.npz"), params=params, denom=denom, split=np.array(split))
    dist.destroy_process_group()


def test_view_granular_split_pair_and_sharded_adam_on_one_gpu(tmp_path):
    """Two ranks, 3+3 views: pair 1 straddles the ranks (its shifted image and the gradient of it travel point to
    point), reduce-scatter -> Adam on half of the flat parameter buffer -> all-gather.  Equals one process."""
    import torch.multiprocessing as mp
    ref_params, ref_denom, _ = _run_views((0, 1), steps=2, sharded=False)
    mp.spawn(_worker_views, args=(2, _free_port(), str(tmp_path)), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "rank0.npz"), np.load(tmp_path / "rank1.npz")
    assert int(r0["split"]) == 1 and int(r1["split"]) == 1
    np.testing.assert_array_equal(r0["params"], r1["params"])
    rel = np.linalg.norm(r0["params"] - ref_params) / np.linalg.norm(ref_params)
    assert rel < 1e-6, rel
    np.testing.assert_array_equal(r0["denom"], ref_denom)


def _worker_overflow(rank, world, port, out_dir, pipeline_ranges):
    """Rank 1 renders from truncated tile lists (its slots claim a tiny capacity); rank 0 does not overflow."""
    import torch.distributed as dist
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, ROOT)
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import FusedRasterizer
    from binocular3dgs_amd.step import FusedAdam, ViewShardedStep, shard_pairs
    W, H, P = 160, 120, 9000
    dev = "cuda"
    model = synth.synth_model(P, seed=13, device=dev, width=W, height=H)
    model.init_densification_stats()
    all_pairs = synth.synth_view_set(W, H, device=dev)
    mine = shard_pairs(3, rank, world)
    pairs = [all_pairs[i] for i in mine]
    bg = torch.zeros(3, device=dev)
    grads = {i: synth.synth_pixel_grads(W, H, seed=20 + i, device=dev) for i in range(3)}
    opt = FusedAdam(model.parameters(), LRS, eps=1e-15)
    fr = FusedRasterizer(model, W, H, num_slots=2 * len(pairs), want_means2D=False)
    st = ViewShardedStep(model, pairs, bg, optimizer=opt, fused=fr, pipeline_ranges=pipeline_ranges, overflow_check_every=0)
    before = torch.cat([p.detach().reshape(-1) for p in model.parameters()]).clone()
    if rank == 1:
        for s in fr.slots:
            s.capacity = 500           # (the buffers are larger: the kernels clamp to what they are told)

    def fn(k, pkg, spkg):
        gc, gd, ga = grads[mine[k]]
        return [(pkg["render"], gc), (pkg["rendered_depth"], gd), (pkg["rendered_alpha"], ga), (spkg["render"], gc)]
    calls = []
    real_all_reduce = dist.all_reduce

    def counting_all_reduce(t, *a, **k):
        calls.append(int(t.numel()))
        return real_all_reduce(t, *a, **k)
    dist.all_reduce = counting_all_reduce
    try:
        for _ in range(2):
            st.step(pair_grad_fn=fn)
    finally:
        dist.all_reduce = real_all_reduce
    torch.cuda.synchronize()
    after = torch.cat([p.detach().reshape(-1) for p in model.parameters()])
    np.savez(os.path.join(out_dir, f"ovf{rank}.npz"), unchanged=bool(torch.equal(before, after)),
             denom=float(model.denom.sum()), accum=float(model.xyz_gradient_accum.sum()), flag=int(fr.overflow_flag.item()),
             step=int(opt.step_count.item()), collectives=np.array(calls))
    dist.destroy_process_group()


@pytest.mark.parametrize("pipeline_ranges", [0, 3])
def test_an_overflow_on_one_rank_drops_the_step_and_its_statistics_on_every_rank(tmp_path, pipeline_ranges):
    """the overflow word is agreed (max over the ranks) BEFORE the chain-rule pass that folds the densification
    statistics in -- a rank that did not overflow itself must not count a step the others drop.  Two ranks on cuda:0 over
    gloo; rank 1 overflows: parameters, step counter and statistics stay untouched on BOTH."""
    import torch.multiprocessing as mp
    mp.spawn(_worker_overflow, args=(2, _free_port(), str(tmp_path), pipeline_ranges), nprocs=2, join=True)
    r0, r1 = np.load(tmp_path / "ovf0.npz"), np.load(tmp_path / "ovf1.npz")
    for r in (r0, r1):
        assert bool(r["unchanged"]) and int(r["step"]) == 0 and int(r["flag"]) != 0
        assert float(r["denom"]) == 0.0 and float(r["accum"]) == 0.0
        if pipeline_ranges:
            # round 5: ONE collective chain per step -- the overflow word rides in the pad of the first gradient range (no
            # 4-byte all-reduce in front of the chain rule); every collective of the two steps is a gradient range
            assert len(r["collectives"]) == 2 * pipeline_ranges and int(r["collectives"].min()) > 1000, r["collectives"]
        else:
            assert sorted(set(r["collectives"].tolist()))[0] == 1          # (the un-pipelined tail keeps the 4-byte agreement)
