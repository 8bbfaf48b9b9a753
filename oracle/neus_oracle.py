"""ORACLE — test infrastructure, not product code.

A CPU restatement (PyTorch, fp32 or fp64) of the reference's NeuS volume-rendering hot path
(BoifZ/VDN-NeRF `dpt_models/{embedder,fields,renderer}.py`). Only `tests/`,
`__graft_entry__.smoke()` and `bench.py`'s `cpu_baseline` leg may import this module, and only
as the checker / the timed CPU baseline. The product path (`vdn-nerf_amd/`) never imports it.

Parity pin: the reference has no tests or golden vectors of its own, so this
restatement is pinned against outputs of the reference itself, imported in the build container by
`tests/golden/make_golden.py`; the resulting vectors live in `tests/golden/*.npz` and
`tests/test_oracle_golden.py` checks the oracle against them.

Differences in *form* from the reference (results are the same):
  * functional, explicit device/dtype (the reference leans on set_default_tensor_type);
  * weights come in as a plain dict with the reference's state_dict keys;
  * the SDF input-gradient is an analytic reverse sweep sharing the forward's pre-activations
    (the reference re-runs forward and calls autograd.grad, fields.py:97-108) - it stays
    differentiable, so training-time double backward goes through plain autograd here;
  * the two torch.rand draws (renderer.py:348,355) are injected arguments.

All file:line citations are relative to the reference repository root.
"""
import math
from dataclasses import dataclass, field
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

SQRT2 = math.sqrt(2.0)


# ----------------------------------------------------------------------------
# configuration records (kwarg names of the reference constructors)
# ----------------------------------------------------------------------------

@dataclass
class SDFConf:          # fields.py:10-21
    d_in: int = 3
    d_out: int = 257
    d_hidden: int = 256
    n_layers: int = 8
    skip_in: Sequence[int] = (4,)
    multires: int = 6
    bias: float = 0.5
    scale: float = 1.0


@dataclass
class RenderingConf:    # fields.py:113-122
    d_feature: int = 256
    mode: str = "idr"
    d_in: int = 9
    d_out: int = 3
    d_hidden: int = 256
    n_layers: int = 4
    multires_view: int = 4
    squeeze_out: bool = True


@dataclass
class NeRFConf:         # fields.py:265-277
    D: int = 8
    W: int = 256
    d_in: int = 4
    d_in_view: int = 3
    multires: int = 10
    multires_view: int = 4
    skips: Sequence[int] = (4,)
    rgb_dims: int = 3
    gen_depth_feats: bool = False
    dpt_dim: int = 96


@dataclass
class RendererConf:     # renderer.py:78-88
    n_samples: int = 64
    n_importance: int = 64
    n_outside: int = 32
    up_sample_steps: int = 4
    perturb: float = 1.0


@dataclass
class Nets:
    """Weights (reference state_dict key schema) + configs of one experiment."""
    sdf: dict
    color: dict
    nerf: dict
    variance: torch.Tensor
    vdn: Optional[dict] = None
    sdf_conf: SDFConf = field(default_factory=SDFConf)
    color_conf: RenderingConf = field(default_factory=RenderingConf)
    vdn_conf: RenderingConf = field(default_factory=lambda: RenderingConf(d_out=96))
    nerf_conf: NeRFConf = field(default_factory=NeRFConf)


def nets_from_numpy(states, dtype=torch.float32, requires_grad=False, device="cpu"):
    """Build `Nets` from vdn_train.synth.make_all_states()-style dict of numpy arrays."""
    def cv(d):
        if d is None:
            return None
        return {k: torch.tensor(v, dtype=dtype, device=device).requires_grad_(requires_grad) for k, v in d.items()}
    wdepth = states.get("depth_network_fine") is not None
    return Nets(sdf=cv(states["sdf_network_fine"]), color=cv(states["color_network_fine"]),
                nerf=cv(states["nerf"]),
                variance=torch.tensor(states["variance_network_fine"]["variance"], dtype=dtype,
                                      device=device).requires_grad_(requires_grad),
                vdn=cv(states["depth_network_fine"]),
                nerf_conf=NeRFConf(gen_depth_feats=wdepth))


def all_params(nets: Nets):
    """Parameter order of dpt_runner.py:121-130: nerf, sdf, variance, colour, (vdn)."""
    out = []
    out += [("nerf." + k, v) for k, v in nets.nerf.items()]
    out += [("sdf." + k, v) for k, v in nets.sdf.items()]
    out += [("variance", nets.variance)]
    out += [("color." + k, v) for k, v in nets.color.items()]
    if nets.vdn is not None:
        out += [("vdn." + k, v) for k, v in nets.vdn.items()]
    return out


# ----------------------------------------------------------------------------
# a1: positional encoding  (embedder.py:6-51)
# ----------------------------------------------------------------------------

def embed(x, n_freqs):
    """[x, sin(2^0 x), cos(2^0 x), sin(2^1 x), ...]; embedder.py:16-36 (log-sampled bands are
    exact powers of two, embedder.py:22-23)."""
    if n_freqs <= 0:
        return x
    parts = [x]
    for k in range(n_freqs):
        f = float(2 ** k)
        parts.append(torch.sin(x * f))
        parts.append(torch.cos(x * f))
    return torch.cat(parts, -1)


def weight_norm_eff(g, v):
    """torch.nn.utils.weight_norm, dim=0: w = v * (g / ||v||_row)  (fields.py:65-66)."""
    return v * (g / v.norm(dim=1, keepdim=True))


def _softplus100(a):
    return F.softplus(a, beta=100)          # fields.py:70 (threshold 20 is torch's default)


def _softplus100_grad(a):
    """d softplus_beta(a) / da exactly as ATen's backward forms it: z/(z+1), 1 past the threshold."""
    z = torch.exp(a * 100.0)
    return torch.where(a * 100.0 > 20.0, torch.ones_like(a), z / (z + 1.0))


# ----------------------------------------------------------------------------
# a2-a5: SDF network  (fields.py:9-108)
# ----------------------------------------------------------------------------

def sdf_eff_weights(p, conf: SDFConf):
    nl = conf.n_layers + 1
    return [(weight_norm_eff(p["lin%d.weight_g" % l], p["lin%d.weight_v" % l]), p["lin%d.bias" % l])
            for l in range(nl)]


def sdf_forward(p, x, conf: SDFConf, with_gradient=False, weights=None):
    """fields.py:72-89 -> [P, d_out] (col 0 = sdf / scale). With `with_gradient`, also d sdf / d x
    [P,3] (what fields.py:97-108 returns, before its unsqueeze), by reverse sweep."""
    wb = weights if weights is not None else sdf_eff_weights(p, conf)
    nl = len(wb)
    xin = x * conf.scale                                   # fields.py:73
    h0 = embed(xin, conf.multires)                         # fields.py:74-75
    h = h0
    pre = []
    for l, (w, b) in enumerate(wb):
        if l in conf.skip_in:
            h = torch.cat([h, h0], 1) / SQRT2              # fields.py:82-83
        a = F.linear(h, w, b)                              # fields.py:85
        pre.append(a)
        if l < nl - 1:
            h = _softplus100(a)                            # fields.py:87-88
    out = torch.cat([a[:, :1] / conf.scale, a[:, 1:]], -1)  # fields.py:89
    if not with_gradient:
        return out
    # reverse sweep: u = d out[:,0] / d (input of layer l)
    P = x.shape[0]
    d0 = h0.shape[1]
    u = (wb[nl - 1][0][0:1, :] / conf.scale).expand(P, -1)
    u_pe = torch.zeros(P, d0, dtype=x.dtype, device=x.device)
    for l in range(nl - 2, -1, -1):
        if (l + 1) in conf.skip_in:                        # u is wrt cat([h, h0]) / sqrt2
            u_pe = u_pe + u[:, -d0:] / SQRT2
            u = u[:, :-d0] / SQRT2
        v = u * _softplus100_grad(pre[l])
        u = v @ wb[l][0]
    u = u + u_pe                                           # wrt PE(x * scale)
    g = u[:, :conf.d_in]
    for k in range(conf.multires):
        f = float(2 ** k)
        s0 = conf.d_in * (1 + 2 * k)
        us, uc = u[:, s0:s0 + conf.d_in], u[:, s0 + conf.d_in:s0 + 2 * conf.d_in]
        g = g + f * (torch.cos(xin * f) * us - torch.sin(xin * f) * uc)
    return out, g * conf.scale


def sdf_only(p, x, conf: SDFConf, weights=None):
    """fields.py:91-92."""
    return sdf_forward(p, x, conf, weights=weights)[:, :1]


# ----------------------------------------------------------------------------
# a6/a7: RenderingNetwork (colour head and the 96-channel VDN head), fields.py:112-176
# ----------------------------------------------------------------------------

def rendering_forward(p, points, normals, view_dirs, feature_vectors, conf: RenderingConf):
    v = embed(view_dirs, conf.multires_view)               # fields.py:149-150
    if conf.mode == "idr":
        x = torch.cat([points, v, normals, feature_vectors], -1)     # fields.py:154
    elif conf.mode == "no_view_dir":
        x = torch.cat([points, normals, feature_vectors], -1)
    elif conf.mode == "no_normal":
        x = torch.cat([points, v, feature_vectors], -1)
    else:
        raise ValueError(conf.mode)
    nl = conf.n_layers + 1
    for l in range(nl):
        w = weight_norm_eff(p["lin%d.weight_g" % l], p["lin%d.weight_v" % l])
        x = F.linear(x, w, p["lin%d.bias" % l])            # fields.py:165
        if l < nl - 1:
            x = torch.relu(x)                              # fields.py:167-168
    return torch.sigmoid(x) if conf.squeeze_out else torch.relu(x)   # fields.py:170-175


# ----------------------------------------------------------------------------
# a8: background NeRF  (fields.py:264-355)
# ----------------------------------------------------------------------------

def nerf_forward(p, input_pts, input_views, conf: NeRFConf):
    e = embed(input_pts, conf.multires)                    # fields.py:325-326
    ev = embed(input_views, conf.multires_view)            # fields.py:327-328
    h = e
    for i in range(conf.D):
        h = torch.relu(F.linear(h, p["pts_linears.%d.weight" % i], p["pts_linears.%d.bias" % i]))
        if i in conf.skips:
            h = torch.cat([e, h], -1)                      # fields.py:334-335
    alpha = F.linear(h, p["alpha_linear.weight"], p["alpha_linear.bias"])
    feature = F.linear(h, p["feature_linear.weight"], p["feature_linear.bias"])
    h = torch.cat([feature, ev], -1)                       # fields.py:340
    h = torch.relu(F.linear(h, p["views_linears.0.weight"], p["views_linears.0.bias"]))
    rgb = F.linear(h, p["rgb_linear.weight"], p["rgb_linear.bias"])
    feat = F.linear(h, p["dpt_linear.weight"], p["dpt_linear.bias"]) if conf.gen_depth_feats else None
    return alpha, rgb, feat


def inv_s_from_variance(variance):
    """fields.py:363-364 + renderer.py:262: exp(10 v).clip(1e-6, 1e6)."""
    return torch.exp(variance * 10.0).clip(1e-6, 1e6)


# ----------------------------------------------------------------------------
# a10: inverse-CDF sampling  (renderer.py:44-74), det=True only (renderer.py:190)
# ----------------------------------------------------------------------------

def sample_pdf_det(bins, weights, n_samples):
    weights = weights + 1e-5
    pdf = weights / weights.sum(-1, keepdim=True)
    cdf = torch.cumsum(pdf, -1)
    cdf = torch.cat([torch.zeros_like(cdf[..., :1]), cdf], -1)
    u = torch.linspace(0.5 / n_samples, 1.0 - 0.5 / n_samples, n_samples, dtype=bins.dtype, device=bins.device)
    u = u.expand(list(cdf.shape[:-1]) + [n_samples]).contiguous()
    inds = torch.searchsorted(cdf, u, right=True)
    below = (inds - 1).clamp(min=0)
    above = inds.clamp(max=cdf.shape[-1] - 1)
    cdf_b, cdf_a = torch.gather(cdf, 1, below), torch.gather(cdf, 1, above)
    bin_b, bin_a = torch.gather(bins, 1, below), torch.gather(bins, 1, above)
    denom = cdf_a - cdf_b
    denom = torch.where(denom < 1e-5, torch.ones_like(denom), denom)
    t = (u - cdf_b) / denom
    return bin_b + t * (bin_a - bin_b)


def excl_cumprod_weights(alpha):
    """alpha * exclusive-cumprod(1 - alpha + 1e-7)  (renderer.py:126,187-188,301)."""
    B = alpha.shape[0]
    one = torch.ones(B, 1, dtype=alpha.dtype, device=alpha.device)
    return alpha * torch.cumprod(torch.cat([one, 1.0 - alpha + 1e-7], -1), -1)[:, :-1]


# ----------------------------------------------------------------------------
# a11/a12: hierarchical up-sampling  (renderer.py:147-207)
# ----------------------------------------------------------------------------

def up_sample(rays_o, rays_d, z_vals, sdf, n_importance, inv_s):
    B, M = z_vals.shape
    pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
    radius = torch.linalg.norm(pts, ord=2, dim=-1)
    inside = (radius[:, :-1] < 1.0) | (radius[:, 1:] < 1.0)             # renderer.py:153-154
    prev_sdf, next_sdf = sdf[:, :-1], sdf[:, 1:]
    prev_z, next_z = z_vals[:, :-1], z_vals[:, 1:]
    mid_sdf = (prev_sdf + next_sdf) * 0.5
    cos_val = (next_sdf - prev_sdf) / (next_z - prev_z + 1e-5)          # renderer.py:159
    prev_cos = torch.cat([torch.zeros(B, 1, dtype=z_vals.dtype, device=z_vals.device), cos_val[:, :-1]], -1)
    cos_val = torch.minimum(prev_cos, cos_val)                          # renderer.py:176-178
    cos_val = cos_val.clip(-1e3, 0.0) * inside                          # renderer.py:179
    dist = next_z - prev_z
    prev_cdf = torch.sigmoid((mid_sdf - cos_val * dist * 0.5) * inv_s)
    next_cdf = torch.sigmoid((mid_sdf + cos_val * dist * 0.5) * inv_s)
    alpha = (prev_cdf - next_cdf + 1e-5) / (prev_cdf + 1e-5)            # renderer.py:186
    weights = excl_cumprod_weights(alpha)
    return sample_pdf_det(z_vals, weights, n_importance).detach()


def cat_z_vals(sdf_fn, rays_o, rays_d, z_vals, new_z, sdf, last):
    B, M = z_vals.shape
    z_cat = torch.cat([z_vals, new_z], -1)
    z_sorted, index = torch.sort(z_cat, dim=-1)                         # renderer.py:197-198
    if not last:
        pts = rays_o[:, None, :] + rays_d[:, None, :] * new_z[..., :, None]
        new_sdf = sdf_fn(pts.reshape(-1, 3)).reshape(B, -1)             # renderer.py:201
        sdf = torch.gather(torch.cat([sdf, new_sdf], -1), 1, index)     # renderer.py:202-205
    return z_sorted, sdf


def coarse_and_outside_z(near, far, conf: RendererConf, perturb, t_rand=None, t_rand_out=None):
    """renderer.py:334-359. `t_rand` [B,1], `t_rand_out` [B,n_outside]: the two uniform draws."""
    dt, dev = near.dtype, near.device
    z = torch.linspace(0.0, 1.0, conf.n_samples, dtype=dt, device=dev)
    z = near + (far - near) * z[None, :]
    zo = None
    if conf.n_outside > 0:
        zo = torch.linspace(1e-3, 1.0 - 1.0 / (conf.n_outside + 1.0), conf.n_outside, dtype=dt, device=dev)
    if perturb > 0:
        z = z + (t_rand - 0.5) * 2.0 / conf.n_samples                   # renderer.py:348-349
        if conf.n_outside > 0:
            mids = 0.5 * (zo[1:] + zo[:-1])
            upper = torch.cat([mids, zo[-1:]], -1)
            lower = torch.cat([zo[:1], mids], -1)
            zo = lower[None, :] + (upper - lower)[None, :] * t_rand_out  # renderer.py:352-356
    if conf.n_outside > 0:
        zo = far / torch.flip(zo, dims=[-1]) + 1.0 / conf.n_samples      # renderer.py:359
        if zo.dim() == 1:
            zo = zo[None, :].expand(near.shape[0], -1)
    return z, zo


def hierarchical_z(nets: Nets, rays_o, rays_d, z_vals, conf: RendererConf, record=None):
    """renderer.py:367-386 (no-grad). Returns the N = n_samples + n_importance sorted z per ray."""
    with torch.no_grad():
        B = z_vals.shape[0]
        wb = sdf_eff_weights(nets.sdf, nets.sdf_conf)
        sdf_fn = lambda pts: sdf_only(nets.sdf, pts, nets.sdf_conf, weights=wb)
        pts = rays_o[:, None, :] + rays_d[:, None, :] * z_vals[..., :, None]
        sdf = sdf_fn(pts.reshape(-1, 3)).reshape(B, conf.n_samples)
        if record is not None:
            record["coarse_sdf"] = sdf.clone()
        for i in range(conf.up_sample_steps):
            new_z = up_sample(rays_o, rays_d, z_vals, sdf, conf.n_importance // conf.up_sample_steps, 64 * 2 ** i)
            z_vals, sdf = cat_z_vals(sdf_fn, rays_o, rays_d, z_vals, new_z, sdf,
                                     last=(i + 1 == conf.up_sample_steps))
            if record is not None:
                record["z_round%d" % i] = z_vals.clone()
    return z_vals


# ----------------------------------------------------------------------------
# a13: background pass  (renderer.py:100-145) - only what render() consumes
# ----------------------------------------------------------------------------

def render_core_outside(nets: Nets, rays_o, rays_d, z_vals, sample_dist):
    B, T = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], sample_dist)], -1)   # renderer.py:107-108
    mid_z = z_vals + dists * 0.5
    pts = rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., :, None]
    r = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).clip(1.0, 1e10)
    pts4 = torch.cat([pts / r, 1.0 / r], -1)                                      # renderer.py:114-115
    dirs = rays_d[:, None, :].expand(B, T, 3)
    density, rgb, feat = nerf_forward(nets.nerf, pts4.reshape(-1, 4), dirs.reshape(-1, 3), nets.nerf_conf)
    alpha = 1.0 - torch.exp(-F.softplus(density.reshape(B, T)) * dists)           # renderer.py:124
    return {"alpha": alpha, "sampled_color": rgb.reshape(B, T, -1),
            "sampled_feat": None if feat is None else feat.reshape(B, T, -1), "z_vals": mid_z}


# ----------------------------------------------------------------------------
# a14: render_core  (renderer.py:209-330)
# ----------------------------------------------------------------------------

def render_core(nets: Nets, rays_o, rays_d, z_vals, sample_dist, bg=None, background_rgb=None,
                cos_anneal_ratio=0.0, depth_before_color=False):
    B, N = z_vals.shape
    dists = z_vals[..., 1:] - z_vals[..., :-1]
    dists = torch.cat([dists, torch.full_like(dists[..., :1], sample_dist)], -1)   # renderer.py:228-229
    mid_z = z_vals + dists * 0.5
    pts = (rays_o[:, None, :] + rays_d[:, None, :] * mid_z[..., :, None]).reshape(-1, 3)
    dirs = rays_d[:, None, :].expand(B, N, 3).reshape(-1, 3)

    out, gradients = sdf_forward(nets.sdf, pts, nets.sdf_conf, with_gradient=True)  # renderer.py:239-243
    sdf, feature = out[:, :1], out[:, 1:]
    sampled_feat = None
    if nets.vdn is not None:                                                       # renderer.py:245-249
        sampled_feat = rendering_forward(nets.vdn, pts, gradients, dirs, feature, nets.vdn_conf)
        if depth_before_color:
            feature = torch.cat([feature, sampled_feat], -1)
        sampled_feat = sampled_feat.reshape(B, N, -1)
    sampled_color = rendering_forward(nets.color, pts, gradients, dirs, feature, nets.color_conf).reshape(B, N, -1)

    inv_s = inv_s_from_variance(nets.variance)                                     # renderer.py:262-263
    true_cos = (dirs * gradients).sum(-1, keepdim=True)                            # renderer.py:265
    iter_cos = -(F.relu(-true_cos * 0.5 + 0.5) * (1.0 - cos_anneal_ratio) +
                 F.relu(-true_cos) * cos_anneal_ratio)                             # renderer.py:269-270
    d = dists.reshape(-1, 1)
    est_next = sdf + iter_cos * d * 0.5
    est_prev = sdf - iter_cos * d * 0.5
    prev_cdf = torch.sigmoid(est_prev * inv_s)
    next_cdf = torch.sigmoid(est_next * inv_s)
    p, c = prev_cdf - next_cdf, prev_cdf
    alpha = ((p + 1e-5) / (c + 1e-5)).reshape(B, N).clip(0.0, 1.0)                 # renderer.py:282

    pts_norm = torch.linalg.norm(pts, ord=2, dim=-1, keepdim=True).reshape(B, N)
    inside = (pts_norm < 1.0).to(z_vals.dtype).detach()
    relax = (pts_norm < 1.2).to(z_vals.dtype).detach()

    if bg is not None:                                                             # renderer.py:289-299
        alpha = alpha * inside + bg["alpha"][:, :N] * (1.0 - inside)
        alpha = torch.cat([alpha, bg["alpha"][:, N:]], -1)
        sampled_color = sampled_color * inside[:, :, None] + bg["sampled_color"][:, :N] * (1.0 - inside)[:, :, None]
        sampled_color = torch.cat([sampled_color, bg["sampled_color"][:, N:]], 1)
        if sampled_feat is not None:
            sampled_feat = sampled_feat * inside[:, :, None] + bg["sampled_feat"][:, :N] * (1.0 - inside)[:, :, None]
            sampled_feat = torch.cat([sampled_feat, bg["sampled_feat"][:, N:]], 1)

    weights = excl_cumprod_weights(alpha)                                          # renderer.py:301
    weights_sum = weights.sum(-1, keepdim=True)
    color = (sampled_color * weights[:, :, None]).sum(1)
    d_feats = None if sampled_feat is None else (sampled_feat * weights[:, :, None]).sum(1)
    if background_rgb is not None:
        color = color + background_rgb * (1.0 - weights_sum)                       # renderer.py:309-310

    g3 = gradients.reshape(B, N, 3)
    gerr = (torch.linalg.norm(g3, ord=2, dim=-1) - 1.0) ** 2
    eik_num = (relax * gerr).sum()
    eik_den = relax.sum()
    gradient_error = eik_num / (eik_den + 1e-5)                                    # renderer.py:313-315
    return {"d_feats": d_feats, "color": color, "sdf": sdf, "dists": dists, "gradients": g3,
            "s_val": (1.0 / inv_s).expand(B * N, 1), "mid_z_vals": mid_z, "weights": weights,
            "cdf": c.reshape(B, N), "gradient_error": gradient_error, "inside_sphere": inside,
            "eik_num": eik_num, "eik_den": eik_den, "alpha": alpha,
            "sampled_color": sampled_color, "feature": feature}


# ----------------------------------------------------------------------------
# a15: render  (renderer.py:332-439)
# ----------------------------------------------------------------------------

def render(nets: Nets, rays_o, rays_d, near, far, conf: RendererConf = RendererConf(), perturb_overwrite=-1,
           background_rgb=None, cos_anneal_ratio=0.0, depth_before_color=False,
           t_rand=None, t_rand_out=None, z_vals_inject=None, record=None):
    """Same output dict as the reference. `z_vals_inject` [B,N] skips the sampler (for per-sample
    parity); `record` (dict) collects intermediates."""
    B = rays_o.shape[0]
    sample_dist = 2.0 / conf.n_samples
    perturb = conf.perturb if perturb_overwrite < 0 else perturb_overwrite
    z_vals, z_out = coarse_and_outside_z(near, far, conf, perturb, t_rand, t_rand_out)
    n_samples = conf.n_samples
    if conf.n_importance > 0:
        if z_vals_inject is None:
            z_vals = hierarchical_z(nets, rays_o, rays_d, z_vals, conf, record)
        else:
            z_vals = z_vals_inject
        n_samples = conf.n_samples + conf.n_importance
    bg = None
    if conf.n_outside > 0:
        z_feed, _ = torch.sort(torch.cat([z_vals, z_out], -1), dim=-1)             # renderer.py:390-391
        bg = render_core_outside(nets, rays_o, rays_d, z_feed, sample_dist)
    fine = render_core(nets, rays_o, rays_d, z_vals, sample_dist, bg=bg, background_rgb=background_rgb,
                       cos_anneal_ratio=cos_anneal_ratio, depth_before_color=depth_before_color)
    weights = fine["weights"]
    if record is not None:
        record["z_vals_inside"] = z_vals.detach().clone()
        record["fine"] = fine
        record["bg"] = bg
    return {
        "render_feats": fine["d_feats"],
        "color_fine": fine["color"],
        "s_val": fine["s_val"].reshape(B, n_samples).mean(-1, keepdim=True),
        "cdf_fine": fine["cdf"],
        "weight_sum": weights.sum(-1, keepdim=True),
        "weight_max": torch.max(weights, dim=-1, keepdim=True)[0],
        "gradients": fine["gradients"],
        "weights": weights,
        "z_vals": bg["z_vals"] if bg is not None else fine["mid_z_vals"],        # renderer.py:421-424
        "gradient_error": fine["gradient_error"],
        "inside_sphere": fine["inside_sphere"],
        "eik_num": fine["eik_num"], "eik_den": fine["eik_den"],                    # extras (not in the reference dict)
    }


# ----------------------------------------------------------------------------
# a16: SDF lattice for mesh extraction  (renderer.py:10-30, 441-446), marching cubes excluded
# ----------------------------------------------------------------------------

def extract_fields(nets: Nets, bound_min, bound_max, resolution, block=64):
    X = torch.linspace(bound_min[0], bound_max[0], resolution).split(block)
    Y = torch.linspace(bound_min[1], bound_max[1], resolution).split(block)
    Z = torch.linspace(bound_min[2], bound_max[2], resolution).split(block)
    u = torch.zeros(resolution, resolution, resolution)
    wb = sdf_eff_weights(nets.sdf, nets.sdf_conf)
    with torch.no_grad():
        for xi, xs in enumerate(X):
            for yi, ys in enumerate(Y):
                for zi, zs in enumerate(Z):
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], -1)
                    val = -sdf_only(nets.sdf, pts, nets.sdf_conf, weights=wb)
                    u[xi * block: xi * block + len(xs), yi * block: yi * block + len(ys),
                      zi * block: zi * block + len(zs)] = val.reshape(len(xs), len(ys), len(zs))
    return u


# ----------------------------------------------------------------------------
# a-R: the caller's loss / schedule semantics  (dpt_runner.py:197-257, 304-319)
# ----------------------------------------------------------------------------

def near_far_from_sphere(rays_o, rays_d):
    """dataset.py:111-118."""
    a = torch.sum(rays_d ** 2, dim=-1, keepdim=True)
    b = 2.0 * torch.sum(rays_o * rays_d, dim=-1, keepdim=True)
    mid = 0.5 * (-b) / a
    return mid - 1.0, mid + 1.0


def loss_from_render(out, true_rgb, mask=None, igr_weight=0.1, mask_weight=0.0, gt_feats=None,
                     depth_ramp=None):
    """dpt_runner.py:208-243. `depth_ramp` = depth_iter_weight() when the VDN loss is active."""
    if mask is None:
        mask = torch.ones_like(true_rgb[:, :1])
    mask_sum = mask.sum() + 1e-5
    err = (out["color_fine"] - true_rgb) * mask
    color_loss = err.abs().sum() / mask_sum
    psnr = 20.0 * torch.log10(1.0 / (((out["color_fine"] - true_rgb) ** 2 * mask).sum() / (mask_sum * 3.0)).sqrt())
    loss = color_loss + out["gradient_error"] * igr_weight
    if mask_weight != 0.0:
        loss = loss + F.binary_cross_entropy(out["weight_sum"].clip(1e-3, 1.0 - 1e-3), mask) * mask_weight
    depth_loss = None
    if depth_ramp is not None and gt_feats is not None:
        derr = (out["render_feats"] - gt_feats) * mask
        depth_loss = derr.abs().sum() / mask_sum
        loss = loss + depth_loss * depth_ramp
    return {"loss": loss, "color_loss": color_loss, "psnr": psnr, "depth_loss": depth_loss}


def learning_rate_factor(iter_step, warm_up_end=5000, end_iter=300000, alpha=0.05):
    """dpt_runner.py:310-316."""
    if iter_step < warm_up_end:
        return iter_step / warm_up_end
    progress = (iter_step - warm_up_end) / (end_iter - warm_up_end)
    return (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - alpha) + alpha


def cos_anneal_ratio(iter_step, anneal_end=50000):
    """dpt_runner.py:304-308."""
    return 1.0 if anneal_end == 0 else min(1.0, iter_step / anneal_end)


def depth_iter_weight(depth_iter, total_iter=5000):
    """dpt_runner.py:167-171."""
    return 1.0 / (math.exp(-10 * (depth_iter / total_iter - 0.5)) + 1.0)
