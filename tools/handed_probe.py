"""usage (GPU box): python tools/handed_probe.py [P W H] -- at the headline's steady state (two binning rounds, settled prediction):
per view, how many Gaussians are visible, how many have at least one instance HANDED to the blend (segment 1 + flagged lists),
how many distinct Gaussians the walked list prefixes hold -- the sets a chain-rule scan would have to look at if something
other than the scratch rows themselves told it where gradients can have arrived."""
import sys
import os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
from binocular3dgs_amd import synth
from binocular3dgs_amd.debug import state_views
from binocular3dgs_amd.fused import FusedRasterizer
from fullsize import view_set

P, W, H = (int(a) for a in sys.argv[1:4]) if len(sys.argv) >= 4 else (1_000_000, 800, 600)
dev = "cuda"
model = synth.synth_model(P, seed=0, device=dev, width=W, height=H, fovx_deg=60.0)
model.init_densification_stats()
pairs = view_set(W, H, 60.0, 6)
bg = torch.zeros(3, device=dev)
vlist, slot = [], 0
for cam, scam, _t in pairs:
    vlist.append((cam, slot)); slot += 1
    if scam is not None:
        vlist.append((scam, slot)); slot += 1
fr = FusedRasterizer(model, W, H, num_slots=len(vlist), want_means2D=False, seg1_fraction="auto")
fr.fit_capacity(vlist, bg)
with torch.no_grad():
    for _ in range(4):                       # settle the open-tile prediction
        outs = fr.render_batch([(c, s, False) for c, s in vlist], bg)
torch.cuda.synchronize()
print("seg1_fraction", fr.seg1_fraction, "num_rendered", fr.num_rendered())
for k, (cam, s) in enumerate(vlist):
    sl = fr.slots[s]
    fv = state_views(P, W, H, sl.capacity, sl.geom, sl.binning, sl.img)
    n1, n2 = int(fv["counts"][0]), int(fv["counts"][2])
    vis = int((outs[k]["radii"] > 0).sum())
    pl = fv["point_list"][:n1 + n2].to(torch.int64)
    handed = torch.zeros(P, dtype=torch.bool, device=dev)
    handed[pl] = True
    # walked prefixes: positions < max n_contrib of the tile
    ranges = fv["ranges"].to(torch.int64)
    ranges2 = fv["ranges2"].to(torch.int64)
    gx, gy = (W + 15) // 16, (H + 15) // 16
    nc = fv["n_contrib"].to(torch.int64)
    pad = torch.zeros(gy * 16, gx * 16, dtype=torch.int64, device=dev)
    pad[:H, :W] = nc
    work = pad.view(gy, 16, gx, 16).permute(0, 2, 1, 3).reshape(gy * gx, 256).max(dim=1).values
    len1 = ranges[:, 1] - ranges[:, 0]
    walked = torch.zeros(P, dtype=torch.bool, device=dev)
    pos = torch.arange(n1 + n2, device=dev)
    # segment 1 entries with local position < work
    tile_of = torch.repeat_interleave(torch.arange(gy * gx, device=dev), len1.clamp(min=0))
    if tile_of.numel() == n1:
        local = pos[:n1] - ranges[tile_of, 0]
        walked[pl[:n1][local < work[tile_of]]] = True
    print(f"view {k}: visible {vis}  N1 {n1} N2 {n2}  handed Gaussians {int(handed.sum())} ({handed.sum().item() / max(vis, 1):.3f} of visible)"
          f"  in walked segment-1 prefixes {int(walked.sum())} ({walked.sum().item() / max(vis, 1):.3f})")
