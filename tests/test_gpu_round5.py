"""GPU: small round-5 items (ADVICE r4 lows)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


def test_shifted_cameras_far_ahead_of_a_busy_stream_keep_their_own_matrices():
    """Camera.shifted() assembles its matrices in a row of a 64-row pinned ring and uploads asynchronously.  Behind a busy
    queue the host can run hundreds of cameras ahead of the copies: a row is only rewritten once its copy has run (an event
    per row), so every camera keeps ITS matrices."""
    from binocular3dgs_amd import synth
    cam = synth.synth_cameras(160, 120, yaws=(4.0,), device="cuda")[0]
    a = torch.rand(4096, 4096, device="cuda")
    for _ in range(20):                                    # ~tens of milliseconds of queued work
        a = a @ a
        a = a / a.abs().max()
    shifts = [0.001 * (k + 1) for k in range(300)]
    cams = [cam.shifted(t) for t in shifts]                # 300 uploads through 64 rows while the queue is busy
    torch.cuda.synchronize()
    base = cam.world_view_transform.cpu().numpy()
    for t, c in zip(shifts, cams):
        w = c.world_view_transform.cpu().numpy()
        want = base.copy()
        want[3, 0] = want[3, 0] - np.float32(t)
        assert np.array_equal(w, want), t


def test_groups_schedule_never_exceeds_the_batch_limit():
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.fused import MAX_BATCH, FusedRasterizer
    W, H = 96, 64
    model = synth.synth_model(3000, seed=1, device="cuda", width=W, height=H)
    cams = synth.synth_cameras(W, H, yaws=tuple(float(k) for k in range(-9, 9)), device="cuda")     # 18 views, two groups
    assert len(cams) > 2 * MAX_BATCH
    fr = FusedRasterizer(model, W, H, num_slots=len(cams), schedule="groups")
    bg = torch.zeros(3, device="cuda")
    with torch.no_grad():
        out = fr.render_batch([(c, k) for k, c in enumerate(cams)], bg)
        fr2 = FusedRasterizer(model, W, H, num_slots=len(cams))
        ref = fr2.render_batch([(c, k) for k, c in enumerate(cams)], bg)
    for a, b in zip(out, ref):
        assert torch.equal(a["render"], b["render"])


def test_in_kernel_activations_are_torch_bits():
    """the raw-parameter path evaluates exp / normalize / sigmoid inside its kernels; the reference feeds
    its rasterizer torch.exp(_scaling), F.normalize(_rotation), torch.sigmoid(_opacity) (scene/gaussian_model.py:95-115).
    Integer radii can only be exact when the two are the same BITS: the hook returns what the kernels compute."""
    import ctypes as C
    from binocular3dgs_amd import _lib
    g = torch.Generator().manual_seed(77)
    P = 1_000_003
    scaling = (-3.9 + 0.9 * torch.randn(P, 3, generator=g)).cuda()
    rotation = torch.randn(P, 4, generator=g).cuda()
    rotation[:5] = torch.tensor([[0.0, 0, 0, 0], [1e-20, 0, 0, 0], [1, 0, 0, 0], [3e18, 3e18, 3e18, 3e18], [-0.0, 2, 0, 0]]).cuda()
    opacity = (3.0 * torch.randn(P, 1, generator=g)).cuda()
    rp = _lib.B3gsRawParams()
    rp.scaling, rp.rotation, rp.opacity = scaling.data_ptr(), rotation.data_ptr(), opacity.data_ptr()
    s, q, o = torch.empty_like(scaling), torch.empty_like(rotation), torch.empty(P, device="cuda")
    rc = _lib.lib().b3gs_debug_activations(P, C.byref(rp), s.data_ptr(), q.data_ptr(), o.data_ptr(),
                                           torch.cuda.current_stream().cuda_stream)
    _lib.check(rc, "b3gs_debug_activations")
    want_s, want_q, want_o = torch.exp(scaling), torch.nn.functional.normalize(rotation), torch.sigmoid(opacity).reshape(-1)

    def bits(t):
        return t.view(torch.int32)
    for name, got, want in (("exp", s, want_s), ("normalize", q, want_q), ("sigmoid", o, want_o)):
        ok = (bits(got) == bits(want)) | (torch.isnan(got) & torch.isnan(want))
        assert bool(ok.all()), (name, int((~ok).sum()), got[~ok][:4], want[~ok][:4])
