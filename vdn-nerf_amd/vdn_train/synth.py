"""Deterministic synthetic weights, cameras and rays for the NeuS hot path.

Nothing here touches the GPU or the reference. The generator is counter based
(splitmix64 -> uniform -> Box-Muller in float64, rounded once to float32) so the
same (seed, name) always yields the same tensor, on any machine, without
storing multi-megabyte weight fixtures.

Shapes follow the only shape family the reference ships
(confs/womsk_white.conf:41-90, confs/womsk_white_wdepth.conf:46-109):
SDF 8x256 (multires 6, skip 4, d_out 257), colour/VDN heads 4x256
(d_in 9, d_feature 256, multires_view 4), background NeRF 8x256
(d_in 4, multires 10, multires_view 4).
"""
import zlib

import numpy as np

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    x = (x + np.uint64(0x9E3779B97F4A7C15)) & _MASK
    z = x
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _MASK
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _MASK
    return z ^ (z >> np.uint64(31))


def _stream_key(seed, name):
    h = zlib.crc32(name.encode()) & 0xFFFFFFFF
    return np.uint64((int(seed) & 0xFFFFFFFF) << 32 | h)


def uniform(seed, name, shape):
    """float64 uniforms in (0, 1), one per element, keyed by (seed, name, index)."""
    n = int(np.prod(shape)) if len(shape) else 1
    with np.errstate(over="ignore"):
        idx = np.arange(n, dtype=np.uint64)
        key = _splitmix64(np.full(n, _stream_key(seed, name), dtype=np.uint64))
        bits = _splitmix64(key ^ _splitmix64(idx))
    u = ((bits >> np.uint64(11)).astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)
    return u.reshape(shape)


def normal(seed, name, shape):
    """float64 standard normals (Box-Muller over two independent uniform streams)."""
    u1 = uniform(seed, name + "/u1", shape)
    u2 = uniform(seed, name + "/u2", shape)
    return np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2)


# ----------------------------------------------------------------------------
# network weights (state_dict key schema of the reference: SURVEY.md 8b)
# ----------------------------------------------------------------------------

def sdf_layer_dims(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6):
    """(in, out) per layer, as fields.py:24-43 lays them out."""
    d0 = d_in * (1 + 2 * multires) if multires > 0 else d_in
    dims = [d0] + [d_hidden] * n_layers + [d_out]
    out = []
    for l in range(len(dims) - 1):
        o = dims[l + 1] - dims[0] if (l + 1) in skip_in else dims[l + 1]
        out.append((dims[l], o))
    return out


def make_sdf_state(seed=0, d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6,
                   bias=0.5, dense_noise=0.02, dtype=np.float32):
    """Geometric-init-shaped SDF weights (sphere of radius `bias`, fields.py:45-63) plus a
    small dense perturbation so that every weight (PE columns included) is exercised, the way
    a trained network's would be. weight_v holds the direction, weight_g its row norm scaled
    by a per-row factor near 1, so weight-norm is not the identity."""
    layers = sdf_layer_dims(d_in, d_out, d_hidden, n_layers, skip_in, multires)
    d0 = layers[0][0]
    nl = len(layers)
    sd = {}
    for l, (ci, co) in enumerate(layers):
        w = np.zeros((co, ci))
        b = np.zeros((co,))
        tag = "sdf/lin%d" % l
        if l == nl - 1:
            w = np.sqrt(np.pi) / np.sqrt(ci) + 1e-4 * normal(seed, tag + "/w", (co, ci))
            b[:] = -bias
        elif multires > 0 and l == 0:
            w[:, :3] = normal(seed, tag + "/w", (co, 3)) * (np.sqrt(2) / np.sqrt(co))
        elif multires > 0 and l in skip_in:
            w = normal(seed, tag + "/w", (co, ci)) * (np.sqrt(2) / np.sqrt(co))
            w[:, -(d0 - 3):] = 0.0
        else:
            w = normal(seed, tag + "/w", (co, ci)) * (np.sqrt(2) / np.sqrt(co))
        if dense_noise > 0:
            w = w + dense_noise * (np.sqrt(2) / np.sqrt(co)) * normal(seed, tag + "/noise", (co, ci)) * (
                0.25 if l == nl - 1 else 1.0)
            b = b + 0.01 * dense_noise * normal(seed, tag + "/bnoise", (co,))
        gscale = 1.0 + 0.05 * (uniform(seed, tag + "/g", (co, 1)) - 0.5)
        vscale = 0.5 + uniform(seed, tag + "/v", (co, 1))
        rn = np.sqrt((w * w).sum(1, keepdims=True))
        sd["lin%d.bias" % l] = b.astype(dtype)
        sd["lin%d.weight_g" % l] = (rn * gscale).astype(dtype)
        sd["lin%d.weight_v" % l] = (w * vscale).astype(dtype)
    return sd


def make_rendering_state(seed=0, tag="color", d_feature=256, d_in=9, d_out=3, d_hidden=256, n_layers=4,
                         multires_view=4, dtype=np.float32):
    """RenderingNetwork weights (fields.py:127-144): uniform(-1/sqrt(in), 1/sqrt(in)) like nn.Linear."""
    d0 = d_in + d_feature + (3 * 2 * multires_view if multires_view > 0 else 0)
    dims = [d0] + [d_hidden] * n_layers + [d_out]
    sd = {}
    for l in range(len(dims) - 1):
        ci, co = dims[l], dims[l + 1]
        k = 1.0 / np.sqrt(ci)
        t = "%s/lin%d" % (tag, l)
        w = (2 * uniform(seed, t + "/w", (co, ci)) - 1) * k
        b = (2 * uniform(seed, t + "/b", (co,)) - 1) * k
        rn = np.sqrt((w * w).sum(1, keepdims=True))
        gscale = 1.0 + 0.1 * (uniform(seed, t + "/g", (co, 1)) - 0.5)
        sd["lin%d.bias" % l] = b.astype(dtype)
        sd["lin%d.weight_g" % l] = (rn * gscale).astype(dtype)
        sd["lin%d.weight_v" % l] = (w * (0.5 + uniform(seed, t + "/v", (co, 1)))).astype(dtype)
    return sd


def make_nerf_state(seed=0, D=8, W=256, d_in=4, d_in_view=3, multires=10, multires_view=4, skips=(4,),
                    rgb_dims=3, gen_depth_feats=False, dpt_dim=96, dtype=np.float32):
    """Background NeRF weights (fields.py:303-320), nn.Linear-style uniform init."""
    ch = d_in * (1 + 2 * multires) if multires > 0 else d_in
    chv = d_in_view * (1 + 2 * multires_view) if multires_view > 0 else d_in_view
    sd = {}

    def lin(name, ci, co):
        k = 1.0 / np.sqrt(ci)
        sd[name + ".weight"] = ((2 * uniform(seed, "nerf/" + name + "/w", (co, ci)) - 1) * k).astype(dtype)
        sd[name + ".bias"] = ((2 * uniform(seed, "nerf/" + name + "/b", (co,)) - 1) * k).astype(dtype)

    lin("pts_linears.0", ch, W)
    for i in range(D - 1):
        lin("pts_linears.%d" % (i + 1), W + ch if i in skips else W, W)
    lin("views_linears.0", chv + W, W // 2)
    lin("feature_linear", W, W)
    lin("alpha_linear", W, 1)
    lin("rgb_linear", W // 2, rgb_dims)
    if gen_depth_feats:
        lin("dpt_linear", W // 2, dpt_dim)
    return sd


def variant_state(state, mode="idr", weight_norm=True):
    """Re-express a weight-normed state dict (lin{l}.bias / weight_g / weight_v) for the other constructor variants of
    the reference (fields.py:113-176): `mode` drops the first layer's columns of the input the mode leaves out
    ('no_normal': normals, columns 30..32; 'no_view_dir': the 27 encoded view columns 3..29 - that mode only exists with
    multires_view = 0, whose raw 3 view columns are not an input either); weight_norm=False stores the effective matrix
    g * v / |v| under the plain nn.Linear keys (lin{l}.weight, lin{l}.bias). Deterministic: generator and tests share it."""
    out = {}
    layers = sorted({k.split(".")[0] for k in state})
    for name in layers:
        g, v, b = (np.asarray(state[name + "." + k], np.float32) for k in ("weight_g", "weight_v", "bias"))
        if name == "lin0" and mode != "idr":
            keep = [c for c in range(v.shape[1]) if not ((30 <= c < 33) if mode == "no_normal" else (3 <= c < 30))]
            v = np.ascontiguousarray(v[:, keep])
        if weight_norm:
            out[name + ".bias"], out[name + ".weight_g"], out[name + ".weight_v"] = b, g, v
        else:
            out[name + ".weight"] = (g * v / np.linalg.norm(v.astype(np.float64), axis=1, keepdims=True)).astype(np.float32)
            out[name + ".bias"] = b
    return out


def make_all_states(seed=0, wdepth=False, variance=0.3, dense_noise=0.02, depth_before_color=False):
    """All networks of one experiment: keys mirror dpt_runner.py:366-375's checkpoint dict. depth_before_color: the colour
    network takes the 96 VDN channels behind the feature vector (d_feature = 352, renderer.py:247-248)."""
    st = {
        "nerf": make_nerf_state(seed, gen_depth_feats=wdepth),
        "sdf_network_fine": make_sdf_state(seed, dense_noise=dense_noise),
        "variance_network_fine": {"variance": np.asarray(variance, dtype=np.float32)},
        "color_network_fine": make_rendering_state(seed, "color", d_out=3, d_feature=352 if depth_before_color else 256),
        "depth_network_fine": make_rendering_state(seed, "vdn", d_out=96) if wdepth else None,
    }
    return st


# ----------------------------------------------------------------------------
# synthetic 800x800 scene: cameras on a sphere, pinhole rays
# ----------------------------------------------------------------------------

H = W_IMG = 800
FOCAL = 1111.0
N_CAMERAS = 40
CAM_RADIUS = 3.0


def make_cameras(seed=0, n=N_CAMERAS, radius=CAM_RADIUS):
    """c2w [n,4,4] (OpenCV convention: +z forward, +y down), looking at the origin."""
    u = uniform(seed, "cam/u", (n,))
    v = uniform(seed, "cam/v", (n,))
    theta = 2 * np.pi * u
    zc = 0.15 + 0.7 * v              # upper hemisphere band
    r = np.sqrt(1 - zc * zc)
    c = np.stack([r * np.cos(theta), r * np.sin(theta), zc], -1) * radius
    fwd = -c / np.linalg.norm(c, axis=-1, keepdims=True)
    up = np.array([0.0, 0.0, 1.0])
    right = np.cross(fwd, up)
    right /= np.linalg.norm(right, axis=-1, keepdims=True)
    down = np.cross(fwd, right)
    c2w = np.tile(np.eye(4), (n, 1, 1))
    c2w[:, :3, 0] = right
    c2w[:, :3, 1] = down
    c2w[:, :3, 2] = fwd
    c2w[:, :3, 3] = c
    return c2w


def intrinsics_inv(focal=FOCAL, h=H, w=W_IMG):
    K = np.array([[focal, 0, (w - 1) / 2.0], [0, focal, (h - 1) / 2.0], [0, 0, 1.0]])
    return np.linalg.inv(K)


def pixel_rays(c2w, px, py, focal=FOCAL):
    """poses.py:203-208 for fixed (non-learnable) pose: p = K^-1 [x,y,1]; v = normalize(p);
    rays_v = R v; rays_o = t. Returns float32 [B,3],[B,3]."""
    Kinv = intrinsics_inv(focal)
    p = np.stack([px, py, np.ones_like(px)], -1).astype(np.float64) @ Kinv.T
    v = p / np.linalg.norm(p, axis=-1, keepdims=True)
    d = v @ c2w[:3, :3].T
    o = np.broadcast_to(c2w[:3, 3], d.shape)
    return o.astype(np.float32).copy(), d.astype(np.float32).copy()


def random_pixel_batch(seed, step, img_idx, batch, rank=0, cams=None, crop=None, focal=FOCAL):
    """512 pixels uniform over one image (poses.py:193-194), keyed by (seed, step, rank). `crop` = side of a centred
    square window to draw from (object-centric captures: the object fills most of the frame)."""
    cams = make_cameras(seed) if cams is None else cams
    tag = "pix/%d/%d" % (step, rank)
    if crop is None:
        px = np.floor(uniform(seed, tag + "/x", (batch,)) * W_IMG)
        py = np.floor(uniform(seed, tag + "/y", (batch,)) * H)
    else:
        px = np.floor(uniform(seed, tag + "/x", (batch,)) * crop) + (W_IMG - crop) // 2
        py = np.floor(uniform(seed, tag + "/y", (batch,)) * crop) + (H - crop) // 2
    return pixel_rays(cams[img_idx], px, py, focal=focal)


def near_far_from_sphere(rays_o, rays_d):
    """dataset.py:111-118."""
    a = (rays_d * rays_d).sum(-1, keepdims=True)
    b = 2.0 * (rays_o * rays_d).sum(-1, keepdims=True)
    mid = 0.5 * (-b) / a
    return (mid - 1.0).astype(np.float32), (mid + 1.0).astype(np.float32)


def target_colors(rays_o, rays_d, albedo=1.0):
    """Procedural ground truth for PSNR runs: sphere r=0.5 with a view-dependent tint on white. `albedo` < 1 darkens
    the object (with the default bright object an empty white scene is a strong local minimum of the L1 loss)."""
    o = rays_o.astype(np.float64)
    d = rays_d.astype(np.float64)
    b = (o * d).sum(-1)
    c = (o * o).sum(-1) - 0.25
    disc = b * b - c
    hit = disc > 0
    t = -b - np.sqrt(np.where(hit, disc, 0.0))
    p = o + d * t[:, None]
    n = p / 0.5
    base = 0.5 + 0.5 * n
    spec = np.clip(-(n * d).sum(-1), 0, 1)[:, None] ** 4
    col = np.clip(0.8 * base + 0.2 * spec, 0, 1) * albedo
    return np.where(hit[:, None], col, 1.0).astype(np.float32)


def jitter(seed, step, batch, n_outside=32, rank=0):
    """The two torch.rand draws of renderer.py:348,355, as injectable tensors."""
    tag = "jit/%d/%d" % (step, rank)
    return (uniform(seed, tag + "/a", (batch, 1)).astype(np.float32),
            uniform(seed, tag + "/b", (batch, n_outside)).astype(np.float32))
