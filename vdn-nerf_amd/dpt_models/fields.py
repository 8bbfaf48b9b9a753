"""Drop-in for the reference's dpt_models/fields.py on MI355X.

Same classes, constructor kwargs, parameter names / order and state_dict schema as the reference
(SURVEY.md 8b): SDFNetwork (fields.py:9-108), RenderingNetwork (112-176), NeRF (264-355),
SingleVarianceNetwork (358-364). Parameters are ordinary torch Parameters; every forward runs the
hand-written gfx950 kernels of libvdn_render.so through the C ABI (include/vdn_render.h). There is
no eager/CPU implementation here: CPU tensors or a missing library raise.
"""
import numpy as np
import torch
import torch.nn as nn

from vdn_hip import images, layout, lib
from dpt_models.embedder import get_embedder


def _require_gpu(t, what):
    if not t.is_cuda:
        raise RuntimeError("%s: expected a tensor on the MI355X (cuda) device, got %s. This package has no "
                           "CPU path." % (what, t.device))
    if t.dtype != torch.float32:
        raise ValueError("%s: expected float32, got %s" % (what, t.dtype))


_stream = lib.stream_handle          # the HIP handle of torch's current stream


class _WNLinear(nn.Module):
    """Parameter holder with the key schema of nn.utils.weight_norm(nn.Linear): bias, weight_g, weight_v."""

    def __init__(self, lin):
        super().__init__()
        w = lin.weight.detach()
        self.bias = nn.Parameter(lin.bias.detach().clone())
        self.weight_g = nn.Parameter(w.norm(dim=1, keepdim=True).clone())
        self.weight_v = nn.Parameter(w.clone())

    def triple(self):
        return (self.weight_g, self.weight_v, self.bias)


class _PlainLinear(nn.Module):
    def __init__(self, lin):
        super().__init__()
        self.weight = nn.Parameter(lin.weight.detach().clone())
        self.bias = nn.Parameter(lin.bias.detach().clone())

    def triple(self):
        return (None, self.weight, self.bias)


class _HipNet(nn.Module):
    """Lazily built device state (effective weights + MFMA chunk streams) shared by the networks.

    `precision` (not a reference kwarg; set as an attribute or through vdn_train.factory):
      "fp32" - exact-f32 MFMA kernels, the parity path (<= 1e-4 vs the reference);
      "bf16" - bf16-operand / fp32-accumulate MFMA kernels, bf16 activation workspaces: the throughput path.
    """
    precision = "fp32"

    def _sfx(self):
        if self.precision not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (self.precision,))
        return "_f32" if self.precision == "fp32" else "_bf16"

    def _store_dtype(self):
        return torch.float32 if self.precision == "fp32" else torch.bfloat16

    def _images(self, join=True):
        # A vdn_train.Trainer updates this network's parameters and weight images on its side stream (raw pointers: torch's
        # version counters do not move). It leaves a hook here that orders torch's current stream behind that update, so
        # render() / validate_image right after train_step never read half-rebuilt images. (An event wait; free once done.
        # join=False: the Trainer's own forward, which places that wait itself - right in front of the first launch that needs it.)
        if join:
            self._join_trainer()
        st = self._image_state()
        st.refresh(_stream())
        return st

    def _join_trainer(self):
        hook = self.__dict__.get("_stream_join")
        if hook is not None:
            hook = hook()             # (a weakref.WeakMethod: the network does not keep its Trainer alive, and copies / pickles of
            if hook is not None:      # the module carry no Trainer - __getstate__ below)
                hook()

    def _image_state(self):
        """The weight-image object of this network (created on first use; NOT refreshed: _images() does that)."""
        dev = lib.first_param(self).device
        if dev.type != "cuda":
            raise RuntimeError("%s lives on %s; move it to the MI355X with .to('cuda') (no CPU path)"
                               % (type(self).__name__, dev))
        st = self.__dict__.get("_img")
        fmt = images.FMT_F32 if self._sfx() == "_f32" else images.FMT_BF16
        if st is None or st.device != dev or st.fmt != fmt:
            st = images.NetImages(self._matrices(), self._streams(), dev, fmt)
            self.__dict__["_img"] = st
        return st


    def __getstate__(self):
        # device state and Trainer hooks are per-process / per-object: copy.deepcopy and torch.save(module) rebuild them lazily
        st = dict(self.__dict__)
        for k in ("_img", "_stream_join", "_scratch", "_cold_start"):
            st.pop(k, None)
        return st


class SDFNetwork(_HipNet):
    def __init__(self, d_in, d_out, d_hidden, n_layers, skip_in=(4,), multires=0, bias=0.5, scale=1,
                 geometric_init=True, weight_norm=True, inside_outside=False):
        super().__init__()
        dims = [d_in] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embed_fn_fine = None
        if multires > 0:
            self.embed_fn_fine, dims[0] = get_embedder(multires, input_dims=d_in)
        self.num_layers = len(dims)
        self.skip_in = tuple(skip_in)
        self.scale = scale
        self.conf = dict(d_in=d_in, d_out=d_out, d_hidden=d_hidden, n_layers=n_layers, skip_in=tuple(skip_in),
                         multires=multires)
        self.weight_norm = weight_norm
        # Same construction order and RNG draws as the reference (fields.py:37-68): nn.Linear's own
        # init first, then the geometric init of IDR.
        for l in range(self.num_layers - 1):
            out_dim = dims[l + 1] - dims[0] if (l + 1) in self.skip_in else dims[l + 1]
            lin = nn.Linear(dims[l], out_dim)
            if geometric_init:
                if l == self.num_layers - 2:
                    mean = np.sqrt(np.pi) / np.sqrt(dims[l])
                    torch.nn.init.normal_(lin.weight, mean=-mean if inside_outside else mean, std=0.0001)
                    torch.nn.init.constant_(lin.bias, bias if inside_outside else -bias)
                elif multires > 0 and l == 0:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.constant_(lin.weight[:, 3:], 0.0)
                    torch.nn.init.normal_(lin.weight[:, :3], 0.0, np.sqrt(2) / np.sqrt(out_dim))
                elif multires > 0 and l in self.skip_in:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
                    torch.nn.init.constant_(lin.weight[:, -(dims[0] - 3):], 0.0)
                else:
                    torch.nn.init.constant_(lin.bias, 0.0)
                    torch.nn.init.normal_(lin.weight, 0.0, np.sqrt(2) / np.sqrt(out_dim))
            # fields.py:65-68: weight_norm(lin) re-parametrises (bias, weight_g, weight_v); without it the nn.Linear itself
            setattr(self, "lin" + str(l), _WNLinear(lin) if weight_norm else _PlainLinear(lin))

    def _matrices(self):
        return {"lin%d" % l: getattr(self, "lin%d" % l).triple() for l in range(self.num_layers - 1)}

    def _streams(self):
        # the bf16 kernel computes in scaled units (csrc/k_sdf_fwd2.h); the streams are rebuilt when the precision changes
        return images.sdf_streams(scaled=(self.precision == "bf16"), **self.conf)

    # -- kernels ------------------------------------------------------------------------------
    def _run_round(self, rays, sdf_out, merge, upsample):
        """One up-sampling round in one launch (vdn_sdf_merge_upsample_bf16): the sdf-only pass on the round's new samples
        `rays` = (rays_o, rays_d, new_z [B,16]) -> sdf_out [B,16], then `merge` (VdnMergeArgs) and `upsample` (VdnUpsampleArgs) on
        those rays. False = this shape / precision is not covered and nothing was launched (make the two calls)."""
        rays_o, rays_d, z = rays
        if self.precision != "bf16" or z.shape[1] != 16 or not z.is_contiguous() or not sdf_out.is_contiguous():
            return False
        img = self._images()
        a = lib.VdnSdfArgs()
        a.rays_o, a.rays_d, a.z, a.n_per_ray, a.z_ld = rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), 16, 16
        a.P, a.scale, a.sdf, a.sdf_ld = z.shape[0] * 16, float(self.scale), sdf_out.data_ptr(), 16
        a.w8row = img.weff_view("lin8").data_ptr()
        a.blob = img.blobs["sdf"].data_ptr()
        return lib.try_call("vdn_sdf_merge_upsample_bf16", a, merge, upsample, _stream())

    def _run_first(self, rays, sdf_out, upsample):
        """The sampler's first pass + first up-sampling round in one launch (vdn_sdf_upsample_bf16): `rays` = (rays_o, rays_d,
        z [B,64] column slice) -> sdf_out [B,64] column slice, then `upsample` (VdnUpsampleArgs, M = 64). False = not covered."""
        rays_o, rays_d, z = rays
        if self.precision != "bf16" or z.shape[1] != 64 or z.stride(1) != 1 or sdf_out.stride(1) != 1:
            return False
        img = self._images()
        a = lib.VdnSdfArgs()
        a.rays_o, a.rays_d, a.z, a.n_per_ray, a.z_ld = rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), 64, z.stride(0)
        a.P, a.scale, a.sdf, a.sdf_ld = z.shape[0] * 64, float(self.scale), sdf_out.data_ptr(), sdf_out.stride(0)
        a.cold_start = int(self.__dict__.get("_cold_start", False))      # (a Trainer's renderer: the step's first SDF pass finds the stream cold)
        a.w8row = img.weff_view("lin8").data_ptr()
        a.blob = img.blobs["sdf"].data_ptr()
        return lib.try_call("vdn_sdf_upsample_bf16", a, upsample, _stream())

    def _run(self, mode, pts=None, rays=None, workspace=None, sdf_out=None):
        """mode 0 -> sdf [P]; mode 1 -> (sdf [P], feat [P,256], normals [P,3]). `rays` = (rays_o, rays_d, z[B,n])."""
        img = self._images()
        a = lib.VdnSdfArgs()
        if pts is not None:
            _require_gpu(pts, "SDFNetwork input")
            pts = pts.contiguous()
            P, dev = pts.shape[0], pts.device
            a.pts, a.n_per_ray = pts.data_ptr(), 1
        else:
            rays_o, rays_d, z = rays           # z may be a column slice [B, n] of a wider row-major buffer
            P, dev = z.shape[0] * z.shape[1], z.device
            a.rays_o, a.rays_d, a.z, a.n_per_ray = rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), z.shape[1]
            a.z_ld = z.stride(0)
        a.P, a.scale = P, float(self.scale)
        if sdf_out is not None:                # [B, n] column slice of a wider buffer
            sdf, a.sdf_ld = sdf_out, sdf_out.stride(0)
        else:
            sdf = torch.empty(P, dtype=torch.float32, device=dev)
            a.sdf_ld = a.n_per_ray
        a.sdf = sdf.data_ptr()
        a.w8row = img.weff_view("lin8").data_ptr()
        if mode == 0:
            a.blob = img.blobs["sdf"].data_ptr()
            lib.call("vdn_sdf_mlp_fwd" + self._sfx(), 0, a, _stream())
            return sdf
        Pr = layout.rows(P, self.precision)          # bf16 planes are tile-blocked and padded to 32 points
        feat = torch.empty(Pr, 256, dtype=self._store_dtype(), device=dev)
        normals = torch.empty(P, 3, dtype=torch.float32, device=dev)
        a.blob = img.blobs["full"].data_ptr()
        a.feat, a.normals = feat.data_ptr(), normals.data_ptr()
        S = None
        if self.precision == "fp32":                 # the bf16 kernel keeps softplus' on the chip
            S = torch.empty(8, Pr, 256, dtype=self._store_dtype(), device=dev)
            a.S = S.data_ptr()
        lib.call("vdn_sdf_mlp_fwd" + self._sfx(), 1, a, _stream())
        if workspace is not None and S is not None:
            workspace["S"] = S
        return sdf, feat, normals

    def forward(self, inputs):
        if inputs.numel() == 0:
            return inputs.new_zeros(0, self.conf["d_out"])
        sdf, feat, _ = self._run(1, pts=inputs.detach())
        if self.precision == "bf16":
            feat = layout.from_pt32(feat, inputs.shape[0], 256)
        return torch.cat([sdf[:, None], feat], dim=-1)

    def sdf(self, x):
        if x.numel() == 0:
            return x.new_zeros(0, 1)
        return self._run(0, pts=x.detach())[:, None]

    def sdf_hidden_appearance(self, x):
        return self.forward(x)

    def gradient(self, x):
        if x.numel() == 0:
            return x.new_zeros(0, 1, 3)
        _, _, n = self._run(1, pts=x.detach())
        return n.unsqueeze(1)


class RenderingNetwork(_HipNet):
    def __init__(self, d_feature, mode, d_in, d_out, d_hidden, n_layers, weight_norm=True, multires_view=0,
                 squeeze_out=True):
        super().__init__()
        self.mode = mode
        self.squeeze_out = squeeze_out
        dims = [d_in + d_feature] + [d_hidden for _ in range(n_layers)] + [d_out]
        self.embedview_fn = None
        if multires_view > 0:
            self.embedview_fn, input_ch = get_embedder(multires_view)
            dims[0] += (input_ch - 3)
        self.num_layers = len(dims)
        self.conf = dict(d_feature=d_feature, mode=mode, d_in=d_in, d_out=d_out, d_hidden=d_hidden,
                         n_layers=n_layers, multires_view=multires_view)
        for l in range(self.num_layers - 1):                                            # fields.py:137-144
            lin = nn.Linear(dims[l], dims[l + 1])
            setattr(self, "lin" + str(l), _WNLinear(lin) if weight_norm else _PlainLinear(lin))

    def _matrices(self):
        return {"lin%d" % l: getattr(self, "lin%d" % l).triple() for l in range(self.num_layers - 1)}

    def _streams(self):
        return images.rendering_streams(**self.conf)

    def _run(self, normals, feat, pts=None, dirs=None, rays=None, extra=None):
        """`extra` [P,96]: the VDN channels appended to the feature vector (d_feature = 352, renderer.py:247-248)."""
        if (extra is not None) != (self.conf["d_feature"] == 352):
            raise ValueError("a d_feature = %d network %s the 96 appended VDN channels" %
                             (self.conf["d_feature"], "needs" if extra is None else "does not take"))
        img = self._images()
        a = lib.VdnRenderNetArgs()
        if extra is not None:
            a.extra = extra.data_ptr()
        P, dev = normals.shape[0], normals.device
        a.blob = img.blobs["fwd"].data_ptr()
        a.normals, a.feat = normals.data_ptr(), feat.data_ptr()
        if rays is not None:
            rays_o, rays_d, z = rays
            a.rays_o, a.rays_d, a.z, a.n_per_ray = rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), z.shape[1]
        else:
            a.pts, a.dirs, a.n_per_ray = pts.data_ptr(), dirs.data_ptr(), 1
        d_out = self.conf["d_out"]
        out = torch.empty(P, d_out, dtype=torch.float32, device=dev)
        a.out, a.P, a.d_out, a.squeeze_out = out.data_ptr(), P, d_out, int(self.squeeze_out)
        lib.call("vdn_rendernet_fwd" + self._sfx(), a, _stream())
        return out

    def forward(self, points, normals, view_dirs, feature_vectors):
        for t, n in ((points, "points"), (normals, "normals"), (view_dirs, "view_dirs"), (feature_vectors, "feature_vectors")):
            _require_gpu(t, "RenderingNetwork " + n)
        if points.shape[0] == 0:
            return points.new_zeros(0, self.conf["d_out"])
        fv = feature_vectors.detach()
        extra = None
        if self.conf["d_feature"] == 352:
            fv, extra = fv[:, :256], fv[:, 256:].contiguous()
        fv = fv.contiguous()
        if self.precision == "bf16":
            fv = layout.to_pt32(fv)
        return self._run(normals.detach().contiguous(), fv, pts=points.detach().contiguous(), dirs=view_dirs.detach().contiguous(), extra=extra)


class NeRF(_HipNet):
    def __init__(self, D=8, W=256, d_in=3, d_in_view=3, gen_depth_feats=False, dpt_dim=1, multires=0,
                 multires_view=0, output_ch=4, skips=[4], rgb_dims=3, use_viewdirs=False):
        super().__init__()
        self.D, self.W, self.d_in, self.d_in_view = D, W, d_in, d_in_view
        self.input_ch, self.input_ch_view = 3, 3
        self.gen_depth_feats, self.dpt_dim = gen_depth_feats, dpt_dim
        if multires > 0:
            _, self.input_ch = get_embedder(multires, input_dims=d_in)
        if multires_view > 0:
            _, self.input_ch_view = get_embedder(multires_view, input_dims=d_in_view)
        self.skips = list(skips)
        self.use_viewdirs = use_viewdirs
        if not use_viewdirs:
            raise ValueError("NeRF(use_viewdirs=False) asserts in the reference's forward (fields.py:355)")
        self.conf = dict(D=D, W=W, d_in=d_in, d_in_view=d_in_view, multires=multires, multires_view=multires_view,
                         skips=tuple(skips), rgb_dims=rgb_dims, gen_depth_feats=gen_depth_feats, dpt_dim=dpt_dim)
        # construction order of fields.py:303-320 (defines parameters() / Adam state order)
        self.pts_linears = nn.ModuleList(
            [_PlainLinear(nn.Linear(self.input_ch, W))] +
            [_PlainLinear(nn.Linear(W, W) if i not in self.skips else nn.Linear(W + self.input_ch, W)) for i in range(D - 1)])
        self.views_linears = nn.ModuleList([_PlainLinear(nn.Linear(self.input_ch_view + W, W // 2))])
        self.feature_linear = _PlainLinear(nn.Linear(W, W))
        self.alpha_linear = _PlainLinear(nn.Linear(W, 1))
        self.rgb_linear = _PlainLinear(nn.Linear(W // 2, rgb_dims))
        if gen_depth_feats:
            self.dpt_linear = _PlainLinear(nn.Linear(W // 2, dpt_dim))

    def _matrices(self):
        m = {"pts_linears.%d" % i: self.pts_linears[i].triple() for i in range(self.D)}
        m["views_linears.0"] = self.views_linears[0].triple()
        m["feature_linear"] = self.feature_linear.triple()
        m["alpha_linear"] = self.alpha_linear.triple()
        m["rgb_linear"] = self.rgb_linear.triple()
        if self.gen_depth_feats:
            m["dpt_linear"] = self.dpt_linear.triple()
        return m

    def _streams(self):
        return images.nerf_streams(**self.conf)

    def _run(self, pts4=None, dirs=None, rays=None, active=None, scratch=False):
        """`active` = (idx int32 [P], n int32 [1]) from vdn_background_active: only those points are evaluated, the other
        outputs stay zero - or, with `scratch`, finite (render_core multiplies them by zero)."""
        img = self._images()
        a = lib.VdnNerfArgs()
        a.blob = img.blobs["fwd"].data_ptr()
        if rays is not None:
            rays_o, rays_d, z = rays
            P, dev = z.numel(), z.device
            a.rays_o, a.rays_d, a.z, a.n_per_ray = rays_o.data_ptr(), rays_d.data_ptr(), z.data_ptr(), z.shape[1]
        else:
            P, dev = pts4.shape[0], pts4.device
            a.pts4, a.dirs, a.n_per_ray = pts4.data_ptr(), dirs.data_ptr(), 1
        # one allocation for the outputs. With a work list the rows off the list must hold FINITE values (render_core multiplies
        # them by zero): a zero fill per call (a 5-us launch on a 360-us render) - or, for a caller that consumes the outputs
        # before its next call on the same stream (`scratch`: NeuSRenderer.render), one buffer per stream that was zeroed once
        # and afterwards holds zeros or an earlier call's outputs
        nf = 96 if self.gen_depth_feats else 0
        if active is not None and scratch:
            key = (P, nf, str(dev), _stream())
            ent = self.__dict__.setdefault("_scratch", {}).get(key)
            if ent is None:
                self.__dict__["_scratch"].clear()            # (one shape at a time: a new batch size replaces the old buffer)
                ent = self.__dict__["_scratch"][key] = [torch.zeros(P * (4 + nf), dtype=torch.float32, device=dev), img.builds]
            elif ent[1] != img.builds:
                # the parameters changed since the buffer was last zeroed (an optimizer step, a checkpoint reload): rows off the
                # list may hold outputs of the OLD weights - non-finite ones if that state had diverged, and 0 * NaN would poison
                # every later render. One fill per parameter change, none in a render loop
                # (img.builds, not the parameters' version tuple: the Trainer's fused Adam does not move torch's version counters)
                ent[0].zero_()
                ent[1] = img.builds
            buf = ent[0]
        else:
            buf = (torch.empty if active is None else torch.zeros)(P * (4 + nf), dtype=torch.float32, device=dev)
        density, rgb = buf[:P], buf[P:4 * P].view(P, 3)
        feat = buf[4 * P:].view(P, 96) if nf else None
        a.density, a.rgb, a.feat, a.P = density.data_ptr(), rgb.data_ptr(), (feat.data_ptr() if feat is not None else None), P
        if active is not None:
            a.active_idx, a.n_active = active[0].data_ptr(), active[1].data_ptr()
        lib.call("vdn_nerf_mlp_fwd" + self._sfx(), a, _stream())
        return density, rgb, feat

    def forward(self, input_pts, input_views):
        _require_gpu(input_pts, "NeRF input_pts")
        _require_gpu(input_views, "NeRF input_views")
        density, rgb, feat = self._run(pts4=input_pts.detach().contiguous(), dirs=input_views.detach().contiguous())
        return density[:, None], rgb, feat


class SingleVarianceNetwork(nn.Module):
    def __init__(self, init_val):
        super().__init__()
        self.register_parameter("variance", nn.Parameter(torch.tensor(init_val)))

    def forward(self, x):
        # fields.py:363-364 (the ones() is created on the parameter's device, not the default one)
        return torch.ones([len(x), 1], device=self.variance.device) * torch.exp(self.variance * 10.0)
