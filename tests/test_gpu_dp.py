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
