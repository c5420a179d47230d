"""GPU (row g1 of the scope table; SURVEY 8d last row / north_star last sentence): the REFERENCE SCHEDULE -- one input
view (+ its randomly shifted binocular partner) per iteration, exponential xyz learning rate, SH-degree ramp, opacity
decay before the optimiser step, densify_and_prune at the densification interval with shared split noise, Adam --
run for a few hundred iterations on a synthetic ground-truth scene three times (tests/ref_schedule.py):

    A  CPU, oracle-backed rasterizer + torch densification      (the comparison target)
    B  MI355X, drop-in render() -> _C.rasterize_gaussians, HIP densification, torch.optim.Adam: the reference's schedule
    C  MI355X, the build's own step: FusedRasterizer (raw parameters, batched pair), fused loss block, one-launch Adam
       with the reference's decay order, HIP densification

Two tests: LOCKSTEP (A and B take every iteration from identical state: identical Gaussian counts after each
densification, parameters and PSNR equal to rounding) and FREE RUN (A, B, C and a perturbed twin of A that measures the
schedule's own sensitivity; see the docstrings).  LLFF fern does not exist in this environment; the target is the build's
own oracle-backed run, as SURVEY 8d allows."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

ITERS = 500
KW = dict(densify_grad_threshold=0.001, densification_interval=50)   # threshold scaled to the 160x120 scene: P settles near 12k


def _train_fused(scene, iterations, **kw):
    import ref_schedule as rs
    return rs.train_fused(scene, iterations=iterations, **kw)


def _flat(tr):
    return torch.cat([g["params"][0].detach().reshape(-1).cpu() for g in tr.opt.param_groups])


def test_reference_schedule_lockstep_hip_vs_oracle_backed_cpu():
    """Every iteration of the schedule, from IDENTICAL state: the HIP trainer leads, the oracle-backed CPU trainer is
    handed its state (parameters, Adam moments and step, densification statistics, SH degree) before each iteration, both
    take the step.  Iterations 1..260 cover the plain phase, the decay start (60), the binocular start and the SH ramp
    (100, 200) and four densifications (100, 150, 200, 250).  Asserted per iteration: same loss (1e-5 relative), the
    Gaussian count after a densification IDENTICAL, updated parameters within 1e-3 relative L2 (Adam with eps = 1e-15
    turns a gradient element that is zero up to rounding into a +-lr step: a handful of elements differ by 2 lr), and
    every 20 iterations PSNR within 0.01 dB."""
    import ref_schedule as rs
    torch.set_num_threads(8)
    scene = rs.make_scene()
    n = 260
    hip, cpu = rs.Trainer(scene, "cuda", iterations=ITERS, **KW), rs.Trainer(scene, "cpu", iterations=ITERS, **KW)
    worst_rel, worst_psnr, densified = 0.0, 0.0, []
    for it in range(1, n + 1):
        cpu.set_state(hip.get_state())
        lh, lc = hip.step(it), cpu.step(it)
        assert abs(lh - lc) <= 1e-5 * abs(lc) + 1e-7, (it, lh, lc)
        assert (hip.last_newP is None) == (cpu.last_newP is None)
        if hip.last_newP is not None:
            assert int(hip.last_newP) == int(cpu.last_newP), (it, hip.last_newP, cpu.last_newP)
            densified.append((it, int(hip.last_newP)))
        a, b = _flat(hip), _flat(cpu)
        assert a.shape == b.shape
        rel = float((a - b).norm() / b.norm())
        worst_rel = max(worst_rel, rel)
        assert rel <= 1e-3, (it, rel)
        if it % 20 == 0:
            d = abs(hip.mean_psnr() - cpu.mean_psnr())
            worst_psnr = max(worst_psnr, d)
            assert d < 0.01, (it, d)
    assert [i for i, _ in densified] == [100, 150, 200, 250] and densified[-1][1] > densified[0][1]
    print(f"lockstep: worst rel-L2 {worst_rel:.2e}, worst |dPSNR| {worst_psnr:.2e} dB, P after densifications {densified}")


def test_reference_schedule_lockstep_swapped_imports_vs_oracle_backed_cpu():
    """What a user who only swaps imports runs: the reference loop's call sequence -- golden
    G11, driven by binocular3dgs_amd/schedule.py -- over the build's drop-ins: the loss functions, inverse_warp_images, the
    GaussianModel methods, gaussians.optimizer, scene.getShiftedCamera each one HIP launch behind the reference's signature
    (tests/ref_schedule.py::SwappedTrainer) -- LEADS; the oracle-backed CPU trainer (the same schedule with PyTorch ops,
    torch.optim.Adam and the torch densification) is handed its full state before every iteration and both step.  210 iterations: decay start (60), binocular start and SH ramp (100, 200), three densifications.  Same bars as
    the other lockstep runs: loss 1e-5 relative, IDENTICAL Gaussian counts, updated parameters 1e-3 relative L2, PSNR 0.01 dB."""
    import ref_schedule as rs
    torch.set_num_threads(8)
    scene = rs.make_scene()
    n = 210
    hip, cpu = rs.SwappedTrainer(scene, iterations=ITERS, **KW), rs.Trainer(scene, "cpu", iterations=ITERS, **KW)
    from binocular3dgs_amd.optim import Adam
    assert isinstance(hip.opt, Adam) and isinstance(hip.opt, torch.optim.Adam)
    worst_rel, worst_psnr, densified = 0.0, 0.0, []
    for it in range(1, n + 1):
        cpu.set_state(hip.get_state())
        lh, lc = hip.step(it), cpu.step(it)
        assert abs(lh - lc) <= 1e-5 * abs(lc) + 1e-7, (it, lh, lc)
        assert (hip.last_newP is None) == (cpu.last_newP is None)
        if hip.last_newP is not None:
            assert int(hip.last_newP) == int(cpu.last_newP), (it, hip.last_newP, cpu.last_newP)
            densified.append((it, int(hip.last_newP)))
        a, b = _flat(hip), _flat(cpu)
        assert a.shape == b.shape
        rel = float((a - b).norm() / b.norm())
        worst_rel = max(worst_rel, rel)
        assert rel <= 1e-3, (it, rel)
        if it % 30 == 0:
            d = abs(hip.mean_psnr() - cpu.mean_psnr())
            worst_psnr = max(worst_psnr, d)
            assert d < 0.01, (it, d)
    assert [i for i, _ in densified] == [100, 150, 200] and densified[-1][1] > densified[0][1]
    print(f"lockstep swapped imports: worst rel-L2 {worst_rel:.2e}, worst |dPSNR| {worst_psnr:.2e} dB, P after densifications {densified}")


@pytest.mark.parametrize("seg1,n", [pytest.param("auto", 160, id="rule"), pytest.param(0.125, 260, id="two_rounds_forced")])
def test_reference_schedule_lockstep_fused_step_vs_oracle_backed_cpu(seg1, n):
    """The strict statement for the build's OWN step (row g1): FusedRasterizer pair batch + fused loss
    block + one-launch Adam with the reference's decay order + HIP densification LEADS, the oracle-backed CPU trainer --
    the reference loop statement by statement -- is handed its full state before every iteration and both step.  Same
    bars as the drop-in lockstep: loss 1e-5 relative, identical Gaussian count after every densification, updated
    parameters 1e-3 relative L2, PSNR 0.01 dB.  `two_rounds_forced`: two binning rounds with the open-tile prediction
    under a camera that changes every iteration, slots that are re-created at every densification and a prediction that is
    thrown away every 7th iteration -- the second round's persistent repair kernel takes its slow path again and again; `rule`: what FusedRasterizer picks by itself at
    this size (one round)."""
    import ref_schedule as rs
    torch.set_num_threads(8)
    # (segment 1 is a whole number of 4096-Gaussian tiles of the depth order: the forced variant starts from 6000 Gaussians,
    # segment 1 = the nearest two thirds)
    scene = rs.make_scene() if seg1 == "auto" else rs.make_scene(P_gt=12000)
    hip = rs.FusedTrainer(scene, iterations=ITERS, seg1_fraction=seg1, **KW)
    cpu = rs.Trainer(scene, "cpu", iterations=ITERS, **KW)
    worst_rel, worst_psnr, densified, repaired = 0.0, 0.0, [], 0
    # loss bar: 1e-5 relative.  The forced variant blends three times as many Gaussians per pixel: ONE pixel of the 19200 on
    # a 1/255-rule flip moves the L1 term by ~1e-5 relative (seen: 1.2e-5 at iteration 34), so its bar is 3e-5.
    loss_tol = 1e-5 if seg1 == "auto" else 3e-5
    for it in range(1, n + 1):
        cpu.set_state(hip.get_state())
        if seg1 != "auto" and it % 7 == 0:
            # forget the open-tile prediction (this scene's tiles stay open and are soon all predicted): the forward finds
            # them unterminated after segment 1 and the persistent repair kernel takes its slow path
            for sl in hip.fused.slots:
                sl.img.zero_()
        lh, lc = hip.step(it), cpu.step(it)
        assert abs(lh - lc) <= loss_tol * abs(lc) + 1e-7, (it, lh, lc)
        assert (hip.last_newP is None) == (cpu.last_newP is None)
        if hip.last_newP is not None:
            assert int(hip.last_newP) == int(cpu.last_newP), (it, hip.last_newP, cpu.last_newP)
            densified.append((it, int(hip.last_newP)))
        repaired += sum(int(s.img[:64].view(torch.int32)[2]) > 0 for s in hip.fused.slots)
        a, b = hip.flat_params(), _flat(cpu)
        assert a.shape == b.shape
        rel = float((a - b).norm() / b.norm())
        worst_rel = max(worst_rel, rel)
        assert rel <= 1e-3, (it, rel)
        if it % 20 == 0:
            d = abs(hip.mean_psnr() - cpu.mean_psnr())
            worst_psnr = max(worst_psnr, d)
            assert d < 0.01, (it, d)
    want = [i for i in (100, 150, 200, 250) if i <= n]
    assert [i for i, _ in densified] == want
    if seg1 != "auto":
        assert hip.fused.seg1_fraction == seg1 and repaired > 0, "the second binning round never had work: nothing was tested"
    print(f"lockstep fused ({seg1}): worst rel-L2 {worst_rel:.2e}, worst |dPSNR| {worst_psnr:.2e} dB, P after "
          f"densifications {densified}, forwards with a repaired view {repaired}")


def test_reference_schedule_free_run_psnr():
    """Free runs of the whole schedule.  The schedule is chaotic (Adam with eps = 1e-15, densification decisions on
    thresholded statistics, split noise addressed by index): A' = the SAME CPU implementation started from initial
    positions scaled by (1 + 1e-7) drifts away from A by tenths of a dB, so 0.1 dB is resolvable only up to that
    spread; the HIP runs (fp32 atomics: summation order varies) scatter by the same amount from run to run.  Asserted:
    before the first densification is amplified (iteration 100) all runs agree to 0.01 dB and produce the same Gaussian
    count; later the runs must stay inside the chaotic band -- |PSNR_hip - PSNR_A| < max(1 dB, 0.1 dB + 3 x the largest
    |A - A'| seen so far) -- reach the same quality regime (every run gains > 8 dB) and keep the Gaussian counts within
    5 %.  The measured deviations are printed.  The strict statement is the lockstep test above."""
    import ref_schedule as rs
    torch.set_num_threads(8)
    scene = rs.make_scene()
    a = rs.train(scene, "cpu", iterations=ITERS, **KW)
    twin = dict(scene, init=dict(scene["init"], xyz=scene["init"]["xyz"] * (1.0 + 1e-7)))
    a2 = rs.train(twin, "cpu", iterations=ITERS, **KW)
    b = rs.train(scene, "cuda", iterations=ITERS, **KW)
    c = _train_fused(scene, ITERS, **KW)
    msg = f"A(cpu oracle)={a}  A'(cpu oracle, perturbed)={a2}  B(hip drop-in)={b}  C(hip fused)={c}"
    print(msg)
    assert a["psnr"][-1][1] > a["psnr"][0][1] + 8.0, msg                       # the schedule really trains
    assert len(a["P"]) >= 8 and a["P"][-1][1] > 4 * scene["init"]["xyz"].shape[0], msg
    for other in (b, c):
        assert other["psnr"][-1][1] > other["psnr"][0][1] + 8.0, msg
        spread = 0.0
        for k, ((ia, pa), (io, po)) in enumerate(zip(a["psnr"], other["psnr"])):
            spread = max(spread, abs(pa - a2["psnr"][k][1]))
            assert ia == io and abs(pa - po) < (0.01 if k == 0 else max(1.0, 0.1 + 3.0 * spread)), (ia, pa, po, spread, msg)
        assert [i for i, _ in other["P"]] == [i for i, _ in a["P"]], msg
        assert other["P"][0][1] == a["P"][0][1], msg
        for (_, na), (_, no) in zip(a["P"], other["P"]):
            assert abs(na - no) <= 0.05 * na, msg
