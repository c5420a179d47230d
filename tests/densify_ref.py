"""Test infrastructure: plain-PyTorch (CPU) restatement of the NET EFFECT of the reference's densify_and_prune
(scene/gaussian_model.py:258-407) in the formulation the HIP kernels use -- per-Gaussian decisions, output order
kept originals | clones | children k=0 | children k=1, zero Adam state for new rows.  Pinned against the
reference's own output by tests/test_golden.py (G8); never imported by the product."""
import torch


def build_rotation(q):
    q = q / q.norm(dim=1, keepdim=True)
    r, x, y, z = q[:, 0], q[:, 1], q[:, 2], q[:, 3]
    return torch.stack([1 - 2 * (y * y + z * z), 2 * (x * y - r * z), 2 * (x * z + r * y),
                        2 * (x * y + r * z), 1 - 2 * (x * x + z * z), 2 * (y * z - r * x),
                        2 * (x * z - r * y), 2 * (y * z + r * x), 1 - 2 * (x * x + y * y)], dim=1).reshape(-1, 3, 3)


def densify_and_prune(params, exp_avg, exp_avg_sq, accum, denom, thr, min_opacity, extent, max_screen_size,
                      percent_dense, noise):
    """params / exp_avg / exp_avg_sq: dict name -> tensor for xyz, f_dc, f_rest, scaling, rotation, opacity;
    noise [2,P,3] addressed by original index.  Returns (new_params, new_exp_avg, new_exp_avg_sq)."""
    g = accum.reshape(-1) / denom.reshape(-1)
    g[g.isnan()] = 0.0
    s = torch.exp(params["scaling"])
    smax = s.max(dim=1).values
    clone = (g.abs() >= thr) & (smax <= percent_dense * extent)
    split = (g >= thr) & (smax > percent_dense * extent)
    op = torch.sigmoid(params["opacity"]).reshape(-1)
    world = bool(max_screen_size)

    def drop(sm):
        d = op < min_opacity
        return d | (sm > 0.1 * extent) if world else d
    child_scaling = torch.log(s / 1.6)
    keep = ~split & ~drop(smax)
    kclone = clone & ~drop(smax)
    kchild = split & ~drop(torch.exp(child_scaling).max(dim=1).values)
    R = build_rotation(params["rotation"])
    out_p, out_m, out_v = {}, {}, {}
    for n, p in params.items():
        rows = [p[keep], p[kclone]]
        for k in range(2):
            c = p[kchild].clone()
            if n == "xyz":
                c = torch.bmm(R[kchild], (s[kchild] * noise[k][kchild]).unsqueeze(-1)).squeeze(-1) + p[kchild]
            elif n == "scaling":
                c = child_scaling[kchild]
            rows.append(c)
        out_p[n] = torch.cat(rows, dim=0)
        new = out_p[n].shape[0] - int(keep.sum())
        z = torch.zeros((new,) + tuple(p.shape[1:]))
        out_m[n] = torch.cat([exp_avg[n][keep], z], dim=0)
        out_v[n] = torch.cat([exp_avg_sq[n][keep], z], dim=0)
    return out_p, out_m, out_v
