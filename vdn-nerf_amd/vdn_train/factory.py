"""Build the shipped-config networks + NeuSRenderer (confs/womsk_white*.conf:41-109) and load
synthetic or checkpointed weights. Mirrors dpt_runner.py:117-142."""
import torch

from dpt_models.fields import NeRF, RenderingNetwork, SDFNetwork, SingleVarianceNetwork
from dpt_models.renderer import NeuSRenderer

CONF = {
    "nerf": dict(D=8, d_in=4, d_in_view=3, W=256, multires=10, multires_view=4, output_ch=4, skips=[4], rgb_dims=3,
                 use_viewdirs=True),
    "sdf_network": dict(d_out=257, d_in=3, d_hidden=256, n_layers=8, skip_in=[4], multires=6, bias=0.5, scale=1.0,
                        geometric_init=True, weight_norm=True),
    "variance_network": dict(init_val=0.3),
    "rendering_network": dict(d_feature=256, mode="idr", d_in=9, d_out=3, d_hidden=256, n_layers=4, weight_norm=True,
                              multires_view=4, squeeze_out=True),
    "depth_extract_network": dict(d_feature=256, mode="idr", d_in=9, d_out=96, d_hidden=256, n_layers=4,
                                  weight_norm=True, multires_view=4, squeeze_out=True),
    "neus_renderer": dict(n_samples=64, n_importance=64, n_outside=32, up_sample_steps=4, perturb=1.0),
}


MODE_KW = {"idr": dict(mode="idr", d_in=9, multires_view=4), "no_normal": dict(mode="no_normal", d_in=6, multires_view=4),
           "no_view_dir": dict(mode="no_view_dir", d_in=6, multires_view=0)}


def build_renderer(wdepth=False, device="cuda", states=None, precision="fp32", color_mode="idr", weight_norm=True,
                   depth_before_color=False, **renderer_overrides):
    """-> NeuSRenderer with its five networks on `device`. `states`: vdn_train.synth.make_all_states()-style
    dict of numpy arrays (checkpoint key names of dpt_runner.py:366-375), always in the shipped (idr, weight-normed)
    form: `color_mode` / `weight_norm` select the other constructor variants of fields.py:113-176 / 10-21 and the
    states are re-expressed for them (synth.variant_state)."""
    from vdn_train import synth
    nerf_kw = dict(CONF["nerf"])
    if wdepth:
        nerf_kw.update(gen_depth_feats=True, dpt_dim=96)
    nerf = NeRF(**nerf_kw)
    sdf = SDFNetwork(**dict(CONF["sdf_network"], weight_norm=weight_norm))
    var = SingleVarianceNetwork(**CONF["variance_network"])
    # depth_before_color: render() then feeds the colour network cat([feature_vector, VDN output]) (renderer.py:247-248)
    col = RenderingNetwork(**dict(CONF["rendering_network"], weight_norm=weight_norm, **MODE_KW[color_mode],
                                  **(dict(d_feature=352) if depth_before_color else {})))
    vdn = RenderingNetwork(**dict(CONF["depth_extract_network"], weight_norm=weight_norm, **MODE_KW[color_mode])) if wdepth else None
    if states is not None:
        tt = lambda d: {k: torch.as_tensor(v) for k, v in d.items()}
        nerf.load_state_dict(tt(states["nerf"]))
        sdf.load_state_dict(tt(synth.variant_state(states["sdf_network_fine"], "idr", weight_norm)))
        var.load_state_dict(tt(states["variance_network_fine"]))
        col.load_state_dict(tt(synth.variant_state(states["color_network_fine"], color_mode, weight_norm)))
        if wdepth:
            vdn.load_state_dict(tt(synth.variant_state(states["depth_network_fine"], color_mode, weight_norm)))
    for m in (nerf, sdf, col, vdn):
        if m is not None:
            m.precision = precision          # "fp32" (parity) or "bf16" (throughput)
    mods = [m.to(device) for m in (nerf, sdf, var, col)] + ([vdn.to(device)] if wdepth else [None])
    kw = dict(CONF["neus_renderer"], precision=precision)     # (explicit: the VDN_PRECISION environment default does not apply here)
    kw.update(renderer_overrides)
    return NeuSRenderer(*mods, **kw)
