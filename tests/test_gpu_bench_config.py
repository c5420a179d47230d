"""The EXACT configuration bench.py times, checked first hand: `bench.Job` at the headline workload
(1M Gaussians @ 800x600, 3 input + 3 binocular-shifted views) -- FusedRasterizer(want_means2D=False, seg1_fraction="auto")
-> two binning rounds with a settled open-tile prediction, sparse gradient rows, ShardedAdam, the whole iteration captured
as ONE HIP graph.

  1. one GRAPH REPLAY and one EAGER step from the same snapshot leave the same model: step counter, `denom`,
     `max_radii2D` bit-equal (no atomics involved); parameters, Adam moments and `xyz_gradient_accum` to 1e-5 (the fp32
     atomics of the blend backward reorder between two runs);
  2. the gradients that eager step produced (sparse rows: rows whose bit is clear count as zero) against the oracle:
     tile_ref forward + backward of the six views on the same pixel gradients, summed and chained through the fp64
     activations -- the bar of tests/test_gpu_fullsize_oracle.py (2e-4 relative L2), statistics included;
  3. the replayed state is what the timed region walks: a second replay continues from it (step counter 2).
"""
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def _job(P, W, H):
    import bench
    argv, sys.argv = sys.argv, ["bench.py", "--gaussians", str(P), "--width", str(W), "--height", str(H)]
    try:
        args = bench.parse()
    finally:
        sys.argv = argv
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(dev)
    return bench.Job(args, dev, 0, 1, False, P, W, H, args.fov, 6, "weak", path="fused", graph=True, loss="synthetic")


def _state(job):
    m, o = job.model, job.opt
    return dict(params=[p.detach().clone() for p in m.parameters()], exp_avg=o.exp_avg.clone(), exp_avg_sq=o.exp_avg_sq.clone(),
                step=int(o.step_count.item()), denom=m.denom.clone(), accum=m.xyz_gradient_accum.clone(),
                max_radii=m.max_radii2D.clone())


def _rel(a, b):
    a, b = a.double().flatten(), b.double().flatten()
    return float((a - b).norm() / b.norm().clamp(min=1e-30))


@pytest.mark.parametrize("P,W,H", [pytest.param(1_000_000, 800, 600, id="headline_1M_800x600")])
def test_graph_replay_equals_eager_step_and_oracle(P, W, H):
    import fullsize
    job = _job(P, W, H)
    job.prepare(3)                                     # warm-up, snapshot, graph capture, restore
    assert job.use_graph and job.run is not job.eager_step, "bench.py must be replaying a HIP graph here"
    fr, st = job.fused, job.stepper
    assert 0.0 < fr.seg1_fraction <= 0.125 and not fr._want_m2d and st.sparse_grad_rows and fr.schedule == "batched"
    step0 = int(job.opt.step_count.item())

    job.restore()
    job.run()                                          # ONE graph replay
    torch.cuda.synchronize()
    a = _state(job)
    n2_replay = [int(s.img[:64].view(torch.int32)[2]) for s in fr.slots]
    job.run()                                          # ... and the next one continues from it
    torch.cuda.synchronize()
    assert int(job.opt.step_count.item()) == step0 + 2

    job.restore()
    job.eager_step()                                   # ONE eager step from the same snapshot
    torch.cuda.synchronize()
    b = _state(job)
    assert a["step"] == b["step"] == step0 + 1
    assert torch.equal(a["denom"], b["denom"]) and torch.equal(a["max_radii"], b["max_radii"])
    assert _rel(a["accum"], b["accum"]) <= 1e-5
    for x, y in zip(a["params"], b["params"]):
        if y.numel():
            assert _rel(x, y) <= 1e-5
    assert _rel(a["exp_avg"], b["exp_avg"]) <= 1e-5 and _rel(a["exp_avg_sq"], b["exp_avg_sq"]) <= 1e-5
    assert not fr.check_overflow()
    # settled prediction: the replayed second binning round has nothing to repair (a tile at the margin may still flip
    # once in a while: the repair kernel then takes its slow path inside the graph -- same results)
    assert sum(n != 0 for n in n2_replay) <= 1, n2_replay

    # ---- the eager gradients of this very path against the oracle --------------------------------------------------
    job.restore()
    job.model.init_densification_stats()
    st.compute_grads(**job.step_kw)
    torch.cuda.synchronize()
    mask = st._row_mask
    assert mask is not None, "the benchmarked step stores sparse gradient rows"
    bits = ((mask.view(-1, 1) >> torch.arange(64, device=mask.device, dtype=torch.int64)) & 1).reshape(-1)[:P].bool()
    assert 0.05 < float(bits.float().mean()) < 0.6
    vlist = []
    for v in st.views:
        gp = job.pix[v.key] if v.role == 0 else (job.pix2[v.key], None, None)
        vlist.append((v.cam, v.role == 0, gp))
    raw, st_norm, st_cnt, st_rad = fullsize.oracle_raw_grads(job.model, vlist, job.bg, W, H)
    for n in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity"):
        g = getattr(job.model, "_" + n).grad
        g = torch.where(bits.view((-1,) + (1,) * (g.dim() - 1)), g, torch.zeros_like(g))   # stale rows count as zero
        e = fullsize.rel_l2(g.cpu().numpy(), raw[n])
        assert e <= 2e-4, f"{n}: rel L2 {e:.3e}"
        # ... and a row the bitmap calls untouched has no gradient in the oracle either (up to its fp64 noise floor)
        # (a Gaussian whose only contribution sits on a 1/255-rule flip may differ: one minimal contribution)
        un = np.abs(raw[n][~bits.cpu().numpy()]).max() if (~bits).any() else 0.0
        assert un <= 1e-4 * max(np.abs(raw[n]).max(), 1e-30), (n, un, np.abs(raw[n]).max())
    assert fullsize.rel_l2(job.model.xyz_gradient_accum.cpu().numpy().ravel(), st_norm) <= 2e-4
    # the integer state of the exact timed configuration is EXACT, as in tests/test_gpu_fullsize_oracle.py (since round 5 the
    # in-kernel activations are torch-ROCm's bits: no radius / visibility flips are tolerated any more)
    d_mis = int((job.model.denom.cpu().numpy().ravel() != st_cnt).sum())
    r_mis = int((job.model.max_radii2D.cpu().numpy().ravel() != st_rad).sum())
    assert d_mis == 0 and r_mis == 0, (d_mis, r_mis)
