"""TEST INFRASTRUCTURE: HIP-vs-oracle metrics at BASELINE.json's full sizes (shared by tests/test_gpu_fullsize_oracle.py
and tools/fullsize_report.py).  Two legs, both against oracle/tile_ref.c on the same seeded synth scene:

  dropin_metrics   the reference-shaped surface (_C.rasterize_gaussians[_backward], reference binning rule):
                   integer state bit-exact, images, eight gradient tensors
  fused_metrics    the BENCHMARKED path (FusedRasterizer: raw parameters, in-kernel activations, batched launches,
                   shared depth sort, tight binning, multi-view chain rule, densification statistics): images per
                   view, parameter gradients summed over the views, statistics, and the structure of the tight tile
                   lists (order-preserving subsequence of the oracle's lists; every dropped entry has
                   alpha < 1/255 at every pixel of its tile)

The oracle gets torch-activated inputs (sigmoid / exp / normalize / cat on the CPU, float32): the same bits the
drop-in path receives; the fused path evaluates the activations in-kernel (1-ulp differences).
"""
import math

import numpy as np
import torch

from helpers import rel_l2

GRAD_KEYS = ("dL_dmeans2D", "dL_dcolors", "dL_dopacity", "dL_dmeans3D", "dL_dcov3D", "dL_dsh", "dL_dscales",
             "dL_drotations")


def view_set(W, H, fov, views):
    """views == 6: BASELINE.md section 3 (3 input + 3 binocular-shifted);  8: config 5 (YAWS_8 input views, the first
    four with a binocular partner are not needed: 8 input views, no partners)."""
    from binocular3dgs_amd import synth
    if views == 8:
        return [(c, None, 0.0) for c in synth.synth_cameras(W, H, fovx_deg=fov, yaws=synth.YAWS_8, device="cuda")]
    return synth.synth_view_set(W, H, fovx_deg=fov, device="cuda")


def activated(model):
    """CPU float32 activated inputs (what render() hands the rasterizer)."""
    with torch.no_grad():
        return dict(means3D=model.get_xyz.detach().cpu(), opacities=model.get_opacity.detach().cpu(),
                    scales=model.get_scaling.detach().cpu(), rotations=model.get_rotation.detach().cpu(),
                    shs=model.get_features.detach().cpu())


def oracle_kw(act, cam, bg, W, H, sh_degree):
    return dict(means3D=act["means3D"].numpy(), opacities=act["opacities"].numpy(), scales=act["scales"].numpy(),
                rotations=act["rotations"].numpy(), shs=act["shs"].numpy(),
                viewmatrix=cam.world_view_transform.cpu().numpy(), projmatrix=cam.full_proj_transform.cpu().numpy(),
                campos=cam.camera_center.cpu().numpy(), bg=bg.cpu().numpy(), W=W, H=H,
                tanfovx=math.tan(cam.FoVx / 2), tanfovy=math.tan(cam.FoVy / 2), sh_degree=sh_degree)


def image_err(got, ref):
    """(max, fraction above the 2e-5 band) of |got - ref| / (1 + |ref|)"""
    err = np.abs(got - ref) / (1 + np.abs(ref))
    return float(err.max()), float((err > 2e-5).mean())


def dropin_metrics(P, W, H, fov=60.0, seed=0, yaw=3.0):
    from oracle import tile_ref
    from binocular3dgs_amd import _C, synth
    from binocular3dgs_amd.debug import state_views
    model = synth.synth_model(P, seed=seed, device="cpu", width=W, height=H, fovx_deg=fov, requires_grad=False)
    cam = synth.synth_cameras(W, H, fovx_deg=fov, yaws=(yaw,), device="cpu")[0]
    bg = torch.tensor([0.1, 0.2, 0.3])
    act = activated(model)
    st = tile_ref.forward(**oracle_kw(act, cam, bg, W, H, 1))
    g = {k: v.cuda() for k, v in act.items()}
    e = torch.empty(0, device="cuda")
    vm, pm, cp, bgd = (cam.world_view_transform.cuda(), cam.full_proj_transform.cuda(), cam.camera_center.cuda(),
                       bg.cuda())
    tfx, tfy = math.tan(cam.FoVx / 2), math.tan(cam.FoVy / 2)
    n, color, depth, alpha, radii, geom, binning, img = _C.rasterize_gaussians(
        bgd, g["means3D"], e, g["opacities"], g["scales"], g["rotations"], 1.0, e, vm, pm, tfx, tfy, H, W, g["shs"], 1,
        cp, False, False)
    v = state_views(P, W, H, n, geom, binning, img)
    m = dict(P=P, W=W, H=H, N=int(st.N), V=int((st.radii > 0).sum()))
    m["n_equal"] = int(n) == int(st.N)
    m["radii_equal"] = bool(np.array_equal(radii.cpu().numpy(), st.radii))
    m["tiles_touched_equal"] = bool(np.array_equal(v["tiles_touched"].cpu().numpy().astype(np.uint32), st.tiles_touched))
    m["point_list_equal"] = bool(np.array_equal(v["point_list"].cpu().numpy().astype(np.uint32), st.point_list))
    m["tile_ids_equal"] = bool(np.array_equal(v["tile_ids"].cpu().numpy().astype(np.uint64), st.keys >> np.uint64(32)))
    m["ranges_equal"] = bool(np.array_equal(v["ranges"].cpu().numpy().astype(np.uint32), st.ranges))
    vis = st.radii > 0
    rec = v["records"].cpu().numpy()
    m["records_equal"] = bool(np.array_equal(rec[vis, 0:2], st.means2D[vis]) and
                              np.array_equal(rec[vis][:, [2, 3, 4, 5]], st.conic_opacity[vis]) and
                              np.array_equal(rec[vis][:, [6, 7, 8]], st.rgb[vis]) and
                              np.array_equal(rec[vis, 9], st.depths[vis]))
    for name, got, ref in (("color", color, st.color), ("depth", depth, st.depth), ("alpha", alpha, st.alpha)):
        m[name + "_max"], m[name + "_frac"] = image_err(got.cpu().numpy(), ref)
    m["n_contrib_frac"] = float((v["n_contrib"].cpu().numpy().astype(np.uint32) != st.n_contrib).mean())
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=seed)
    ref = tile_ref.backward(st, gc.numpy(), gd.numpy(), ga.numpy())
    res = _C.rasterize_gaussians_backward(bgd, g["means3D"], radii, e, g["scales"], g["rotations"], 1.0, e, vm, pm, tfx,
                                          tfy, gc.cuda(), gd.cuda(), ga.cuda(), g["shs"], 1, cp, geom, n, binning, img,
                                          alpha, False)
    for k, t in zip(GRAD_KEYS, res):
        m[k] = rel_l2(t.cpu().numpy(), ref[k])
    return m


def _tight_list_metrics(P, W, H, st, fv, radii_hip):
    """Structure of the fused path's tile lists of one view against the oracle's full lists (device tensors, int64).
    A tile's list is segment 1 followed by segment 2 (two-round binning).  Checked: every entry is in the oracle's list
    of the same tile, in the oracle's order (segment 2 after segment 1), and every oracle entry that is MISSING either
    cannot reach alpha = 1/255 at any pixel of the tile (tight binning) or sits at or behind the position where the
    oracle's own forward stopped reading the tile (largest n_contrib of the tile: two-round binning never emits it)."""
    dev = "cuda"
    agree = torch.from_numpy(st.radii).to(dev) == radii_hip          # Gaussians whose integer radius agrees
    o_pl = torch.from_numpy(st.point_list.astype(np.int64)).to(dev)
    o_tile = torch.from_numpy((st.keys >> np.uint64(32)).astype(np.int64)).to(dev)
    o_start = torch.from_numpy(st.ranges[:, 0].astype(np.int64)).to(dev)
    o_q = torch.arange(o_pl.numel(), device=dev) - o_start[o_tile]   # position inside the oracle's tile list
    n1, n2 = int(fv["counts"][0]), int(fv["counts"][2]) if "point_list2" in fv else 0
    segs = [(fv["point_list"][:n1].to(torch.int64), fv["tile_ids"][:n1].to(torch.int64))]
    if n2:   # segment 2 sits behind segment 1 in the same arrays
        segs.append((fv["point_list2"][n1:n1 + n2].to(torch.int64), fv["tile_ids2"][n1:n1 + n2].to(torch.int64)))
    o_keep = agree[o_pl]
    o_pl, o_tile, o_q = o_pl[o_keep], o_tile[o_keep], o_q[o_keep]
    o_key = o_tile * P + o_pl                                          # (tile, Gaussian) is unique inside a list
    so, perm = torch.sort(o_key)
    out = dict(radius_flips=int((~agree).sum()), N_seg1=n1, N_seg2=n2, N_tight=n1 + n2, N_oracle=int(st.N), subset=True,
               order_preserved=True)
    present = torch.zeros(o_key.numel(), dtype=torch.bool, device=dev)
    tiles = st.ranges.shape[0]
    last1 = torch.full((tiles,), -1, dtype=torch.int64, device=dev)
    for k, (t_pl, t_tile) in enumerate(segs):
        keep = agree[t_pl]
        t_pl, t_tile = t_pl[keep], t_tile[keep]
        t_key = t_tile * P + t_pl
        idx = torch.searchsorted(so, t_key).clamp(max=so.numel() - 1)
        member = so[idx] == t_key
        pos = perm[idx]
        out["subset"] = out["subset"] and bool(member.all())
        out["order_preserved"] = out["order_preserved"] and bool((pos[1:] > pos[:-1]).all())
        present[pos[member]] = True
        if k == 0:
            last1.scatter_reduce_(0, t_tile, o_q[pos], reduce="amax")
        else:   # every segment-2 entry of a tile lies behind every segment-1 entry of that tile
            out["order_preserved"] = out["order_preserved"] and bool((o_q[pos] > last1[t_tile]).all())
    d_pl, d_tile, d_q = o_pl[~present], o_tile[~present], o_q[~present]
    out["dropped"] = int(d_pl.numel())
    # where the oracle's forward stopped reading each tile
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nc = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
    nc[:H, :W] = torch.from_numpy(st.n_contrib.astype(np.int64)).to(dev)
    tile_stop = nc.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(1).values
    behind = d_q >= tile_stop[d_tile]
    out["dropped_unreached"] = int(behind.sum())
    d_pl, d_tile = d_pl[~behind], d_tile[~behind]
    # largest alpha any REACHABLE dropped entry attains at any pixel of its tile, from the oracle's own record
    m2d = torch.from_numpy(st.means2D).to(dev).double()
    co = torch.from_numpy(st.conic_opacity).to(dev).double()
    ox = torch.arange(16, device=dev, dtype=torch.float64)
    worst, n_bad = 0.0, 0
    for c0 in range(0, d_pl.numel(), 400_000):
        g_, t_ = d_pl[c0:c0 + 400_000], d_tile[c0:c0 + 400_000]
        px = ((t_ % gx) * 16).double()[:, None] + ox[None, :]          # [n,16] pixel x of the tile's columns
        py = ((t_ // gx) * 16).double()[:, None] + ox[None, :]
        inx, iny = px < W, py < H
        dx = (m2d[g_, 0][:, None] - px)[:, None, :]                    # [n,1,16]
        dy = (m2d[g_, 1][:, None] - py)[:, :, None]                    # [n,16,1]
        c = co[g_]
        power = -0.5 * (c[:, 0, None, None] * dx * dx + c[:, 2, None, None] * dy * dy) - c[:, 1, None, None] * dx * dy
        a = c[:, 3, None, None] * torch.exp(power.clamp(max=0.0))
        a = torch.where((power > 0) | ~(iny[:, :, None] & inx[:, None, :]), torch.zeros_like(a), a)
        amax = a.reshape(a.shape[0], -1).max(1).values if a.numel() else a.reshape(0)
        n_bad += int((amax * 255.0 >= 1.0).sum())
        # entries that could contribute but were dropped: only legitimate next to a pixel whose termination the two
        # implementations decide differently (T within an ulp of 1e-4); counted, and excluded from the alpha bound
        ok = amax * 255.0 < 1.0
        worst = max(worst, float(amax[ok].max()) if ok.any() else 0.0)
    out["dropped_max_alpha_x255"] = worst * 255.0
    out["dropped_reachable_contributors"] = n_bad
    return out


def oracle_raw_grads(model, vlist, bg, W, H, per_view_cb=None):
    """Oracle reference for one iteration of `vlist` = [(camera, is_primary, (dL_dcolor, dL_ddepth | None, dL_dalpha | None))]:
    tile_ref forward + backward per view, gradients summed over the views in fp64 and chained through the fp64 activations
    (sigmoid / exp / normalize / cat) to the RAW parameters; plus the densification statistics of the primary views.
    per_view_cb(k, oracle_state, oracle_grads) is called for every view.  Returns (raw_grads {name: ndarray},
    st_norm, st_cnt, st_rad)."""
    from oracle import tile_ref
    P = model.get_xyz.shape[0]
    act = activated(model)
    acc = {k: np.zeros(tuple(act[n].shape), np.float64) for k, n in
           (("dL_dmeans3D", "means3D"), ("dL_dopacity", "opacities"), ("dL_dscales", "scales"),
            ("dL_drotations", "rotations"), ("dL_dsh", "shs"))}
    st_norm, st_cnt, st_rad = np.zeros(P, np.float64), np.zeros(P, np.float64), np.zeros(P, np.float64)
    for k, (cam, is_primary, (a, b, c)) in enumerate(vlist):
        st = tile_ref.forward(**oracle_kw(act, cam, bg, W, H, 1))
        ref = tile_ref.backward(st, a.cpu().numpy(), None if b is None else b.cpu().numpy(),
                                None if c is None else c.cpu().numpy())
        for kk in acc:
            acc[kk] += ref[kk].astype(np.float64).reshape(acc[kk].shape)
        if is_primary:
            vis = st.radii > 0
            st_norm[vis] += np.linalg.norm(ref["dL_dmeans2D"][vis, :2].astype(np.float64), axis=1)
            st_cnt[vis] += 1
            st_rad[vis] = np.maximum(st_rad[vis], st.radii[vis])
        if per_view_cb is not None:
            per_view_cb(k, st, ref)
    raw = {n: getattr(model, "_" + n).detach().cpu().double().requires_grad_(True)
           for n in ("xyz", "features_dc", "features_rest", "scaling", "rotation", "opacity")}
    acts = [raw["xyz"], torch.sigmoid(raw["opacity"]), torch.exp(raw["scaling"]),
            torch.nn.functional.normalize(raw["rotation"]), torch.cat((raw["features_dc"], raw["features_rest"]), 1)]
    torch.autograd.backward(acts, [torch.from_numpy(acc[k]) for k in
                                   ("dL_dmeans3D", "dL_dopacity", "dL_dscales", "dL_drotations", "dL_dsh")])
    return {n: raw[n].grad.numpy() for n in raw}, st_norm, st_cnt, st_rad


def fused_metrics(P, W, H, fov=60.0, seed=0, views=6, check_lists=True, seg1_fraction=0.125, want_means2D=True,
                  reference_binning=False):
    """All views of one iteration through FusedRasterizer.render_batch (the bench.py path) against the oracle.
    seg1_fraction: 0.125 = two-round binning forced on, first forward (no open-tile prediction yet: the second round
    repairs); "auto" = what bench.py builds (the deterministic rule of FusedRasterizer.fit_capacity, prediction settled by
    its forward); 0.0 = one round.  want_means2D=False is bench.py's setting (no per-view screen-space gradient tensors:
    the densification statistics consume them inside the chain-rule pass).  reference_binning=True (with one round): the
    fused path bins by the reference's rectangle rule (B3gsForwardView::reference_binning) -- its lists are then compared with
    the oracle's BIT FOR BIT ("lists_exact") instead of as subsequences."""
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.debug import state_views
    from binocular3dgs_amd.fused import FusedRasterizer
    dev = "cuda"
    model = synth.synth_model(P, seed=seed, device=dev, width=W, height=H, fovx_deg=fov)
    model.init_densification_stats()
    pairs = view_set(W, H, fov, views)
    bg = torch.zeros(3, device=dev)
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=seed, device=dev)
    gc2 = synth.synth_pixel_grads(W, H, seed=100 + seed, device=dev)[0]
    vlist, slot = [], 0
    for cam, scam, _t in pairs:
        vlist.append((cam, slot, True, (gc, gd, ga)))
        slot += 1
        if scam is not None:
            vlist.append((scam, slot, False, (gc2, None, None)))
            slot += 1
    fr = FusedRasterizer(model, W, H, num_slots=len(vlist), want_means2D=want_means2D, seg1_fraction=seg1_fraction,
                         reference_binning=reference_binning)
    if seg1_fraction == "auto":
        fr.fit_capacity([(c, s) for c, s, _, _ in vlist], bg)
    else:
        with torch.no_grad():                              # size the persistent binning buffers for this scene
            fr.seg1_fraction, keep = 0.0, fr.seg1_fraction
            fr.render_batch([(c, s, False) for c, s, _, _ in vlist], bg)
            while fr.overflowed():
                fr.grow()
                fr.render_batch([(c, s, False) for c, s, _, _ in vlist], bg)
            fr.seg1_fraction = keep
            fr.high_water.zero_()
            fr.overflow_flag.zero_()
    for p in model.parameters():
        p.grad = torch.zeros_like(p)
    outs = fr.render_batch([(c, s, st_) for c, s, st_, _ in vlist], bg)
    o_t, g_t = [], []
    for o, (_, _, _, (a, b, c)) in zip(outs, vlist):
        o_t.append(o["render"]); g_t.append(a)
        if b is not None:
            o_t += [o["rendered_depth"], o["rendered_alpha"]]
            g_t += [b, c]
    torch.autograd.backward(o_t, g_t)
    torch.cuda.synchronize()
    # every scratch row the blend backward flushed into was read and reset by the chain rule's scan, which looks only at the
    # rows of the Gaussians the blend forward marked (GeomView::staged): nothing may be left behind
    for _c, s_, _p, _g in vlist:
        assert float(fr.slots[s_].scratch.abs().max()) == 0.0, "a scratch row outside the marked set received a gradient"
    nr = fr.num_rendered()
    assert max(nr) <= fr.capacity and int(fr.overflow_flag.item()) == 0, "binning capacity overflow"
    m = dict(P=P, W=W, H=H, views=len(vlist), N_binned=nr, seg1_fraction=fr.seg1_fraction)
    per_view = [None] * len(vlist)

    def per_view_cb(k, st, ref):
        _, s, _, _ = vlist[k]
        o = outs[k]
        pv = dict(N_oracle=int(st.N))
        radii = o["radii"]
        pv["radius_flips"] = int((radii.cpu().numpy() != st.radii).sum())
        for name, key, refimg in (("color", "render", st.color), ("depth", "rendered_depth", st.depth),
                                  ("alpha", "rendered_alpha", st.alpha)):
            pv[name + "_max"], pv[name + "_frac"] = image_err(o[key].detach().cpu().numpy(), refimg)
        sl = fr.slots[s]
        # the slot's binning buffer is carved for `capacity` instances (that fixes where the tile-id array of the
        # two-word layout starts); the first N entries are the lists
        fv = state_views(P, W, H, sl.capacity, sl.geom, sl.binning, sl.img)
        pv["N_seg2"] = int(fv["counts"][2])
        if check_lists and reference_binning:
            n1 = int(fv["counts"][0])
            pv["lists_exact"] = dict(
                n_equal=n1 == int(st.N) and int(fv["counts"][2] if fr.seg1_fraction > 0 else 0) == 0,
                point_list_equal=bool(np.array_equal(fv["point_list"][:n1].cpu().numpy().astype(np.uint32), st.point_list)),
                tile_ids_equal=bool(np.array_equal(fv["tile_ids"][:n1].cpu().numpy().astype(np.uint64), st.keys >> np.uint64(32))),
                ranges_equal=bool(np.array_equal(fv["ranges"].cpu().numpy().astype(np.uint32), st.ranges)),
                tiles_touched_equal=bool(np.array_equal(fv["tiles_touched"].cpu().numpy().astype(np.uint32), st.tiles_touched)))
        elif check_lists and (k < 2 or k == len(vlist) - 1):
            pv["lists"] = _tight_list_metrics(P, W, H, st, fv, radii)
        if sl.means2D_grad is not None:
            pv["dL_dmeans2D"] = rel_l2(sl.means2D_grad.cpu().numpy(), ref["dL_dmeans2D"])
        per_view[k] = pv

    raw, st_norm, st_cnt, st_rad = oracle_raw_grads(model, [(c, prim, g) for c, _, prim, g in vlist], bg, W, H, per_view_cb)
    m["per_view"] = per_view
    for n in raw:
        got = getattr(model, "_" + n).grad
        if got.numel():
            m["grad_" + n] = rel_l2(got.cpu().numpy(), raw[n])
    m["stat_accum"] = rel_l2(model.xyz_gradient_accum.cpu().numpy().ravel(), st_norm)
    m["stat_denom_mismatch"] = int((model.denom.cpu().numpy().ravel() != st_cnt).sum())
    m["stat_max_radii_mismatch"] = int((model.max_radii2D.cpu().numpy().ravel() != st_rad).sum())
    return m


def render_node_metrics(P, W, H, fov=60.0, seed=0, views=6):
    """The zero-change surface at full size against the oracle: render() per view exactly as an unchanged train.py calls it
    (the raw-parameter node: in-kernel activations, tight binning, the forward of a pair as ONE two-view launch with one
    depth sort, every node of the backward launched as one batch that accumulates into .grad), all views of one iteration,
    one backward.  Per view: integer radii, visibility, images; over the iteration: parameter gradients (oracle gradients
    chained through fp64 activations), every input view's `viewspace_points.grad`."""
    import binocular3dgs_amd.rasterizer as R
    from binocular3dgs_amd import synth
    from binocular3dgs_amd.render import PipelineParams, render
    dev = "cuda"
    model = synth.synth_model(P, seed=seed, device=dev, width=W, height=H, fovx_deg=fov)
    pairs = view_set(W, H, fov, views)
    bg = torch.zeros(3, device=dev)
    gc, gd, ga = synth.synth_pixel_grads(W, H, seed=seed, device=dev)
    gc2 = synth.synth_pixel_grads(W, H, seed=100 + seed, device=dev)[0]
    pipe = PipelineParams()
    with torch.no_grad():                                  # (the first render of a shape is the exact, synchronous one)
        render(pairs[0][0], model, pipe, bg)
    R._flush_pending()
    old_idle, R._LAZY_WHEN_IDLE = R._LAZY_WHEN_IDLE, True  # the two-view launch, whatever the queue holds
    s0 = dict(R._stats)
    try:
        for p in model.parameters():
            p.grad = None
        vlist, pkgs, o_t, g_t = [], [], [], []
        for cam, scam, _t in pairs:
            a = render(cam, model, pipe, bg)
            pkgs.append(a)
            vlist.append((cam, True, (gc, gd, ga)))
            o_t += [a["render"], a["rendered_depth"], a["rendered_alpha"]]
            g_t += [gc, gd, ga]
            if scam is not None:
                b = render(scam, model, pipe, bg)
                pkgs.append(b)
                vlist.append((scam, False, (gc2, None, None)))
                o_t.append(b["render"])
                g_t.append(gc2)
        torch.autograd.backward(o_t, g_t)
        torch.cuda.synchronize()
    finally:
        R._LAZY_WHEN_IDLE = old_idle
    m = dict(P=P, W=W, H=H, views=len(vlist), stats={k: R._stats[k] - s0[k] for k in R._stats})
    per_view = [None] * len(vlist)

    def cb(k, st, ref):
        o = pkgs[k]
        pv = dict(radius_flips=int((o["radii"].cpu().numpy() != st.radii).sum()),
                  visibility_flips=int((o["visibility_filter"].cpu().numpy() != (st.radii > 0)).sum()))
        for name, key, refimg in (("color", "render", st.color), ("depth", "rendered_depth", st.depth),
                                  ("alpha", "rendered_alpha", st.alpha)):
            pv[name + "_max"], pv[name + "_frac"] = image_err(o[key].detach().cpu().numpy(), refimg)
        g = o["viewspace_points"].grad
        pv["dL_dmeans2D"] = None if g is None else rel_l2(g.cpu().numpy(), ref["dL_dmeans2D"])
        per_view[k] = pv

    raw, _, _, _ = oracle_raw_grads(model, vlist, bg, W, H, cb)
    m["per_view"] = per_view
    for n in raw:
        got = getattr(model, "_" + n).grad
        if got.numel():
            m["grad_" + n] = rel_l2(got.cpu().numpy(), raw[n])
    return m
