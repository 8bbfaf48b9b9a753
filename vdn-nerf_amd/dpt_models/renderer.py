"""Drop-in for the reference's dpt_models/renderer.py on MI355X.

NeuSRenderer keeps the reference's constructor and render()/extract_geometry() signatures and the
exact output dict of renderer.py:426-439; the work is done by the gfx950 kernels of
libvdn_render.so: vdn_coarse_z, vdn_upsample_round, vdn_merge_sorted, vdn_sections (sampling,
renderer.py:334-391), the fused MLP kernels (fields.py), and vdn_alpha_composite_fwd
(renderer.py:262-315). No eager/CPU path exists.

Extra, optional keyword arguments (not in the reference): `t_rand` [B,1] and `t_rand_out`
[B,n_outside] inject the two uniform draws of renderer.py:348,355; `z_vals_inject` [B,N] skips the
hierarchical sampler. They exist for parity tests (hard part 6).
"""
import os

import numpy as np
import torch

from vdn_hip import lib
from dpt_models.fields import _require_gpu, _stream


def extract_fields_device(bound_min, bound_max, resolution, query_func, device=None):
    """SDF lattice in 64^3 blocks (renderer.py:10-30), kept on the device; `query_func` maps [P,3] device points -> values."""
    N = 64
    device = device or (bound_min.device if torch.is_tensor(bound_min) and bound_min.is_cuda else torch.device("cuda"))
    lo = [float(v) for v in bound_min]
    hi = [float(v) for v in bound_max]
    X = torch.linspace(lo[0], hi[0], resolution).split(N)
    Y = torch.linspace(lo[1], hi[1], resolution).split(N)
    Z = torch.linspace(lo[2], hi[2], resolution).split(N)
    u = torch.zeros([resolution, resolution, resolution], dtype=torch.float32, device=device)
    with torch.no_grad():
        for xi, xs in enumerate(X):
            for yi, ys in enumerate(Y):
                for zi, zs in enumerate(Z):
                    xx, yy, zz = torch.meshgrid(xs, ys, zs, indexing="ij")
                    pts = torch.stack([xx.reshape(-1), yy.reshape(-1), zz.reshape(-1)], dim=-1).to(device)
                    val = query_func(pts).reshape(len(xs), len(ys), len(zs))
                    u[xi * N: xi * N + len(xs), yi * N: yi * N + len(ys), zi * N: zi * N + len(zs)] = val
    return u


def extract_fields(bound_min, bound_max, resolution, query_func, device=None):
    """renderer.py:10-30: the lattice as a numpy array, like the reference returns it."""
    return extract_fields_device(bound_min, bound_max, resolution, query_func, device).cpu().numpy()


def extract_geometry(bound_min, bound_max, resolution, threshold, query_func, method=None):
    """renderer.py:33-41 -> (vertices [V,3] float64 in world coordinates, triangles [F,3]). The reference triangulates with
    `mcubes.marching_cubes(u, threshold)` (third-party PyMCubes): vdn_hip.mesh.marching_cubes extracts the same mesh on the
    device - the classic 256-case tables with the library's sequential vertex / triangle numbering, float64 vertices (restated
    from its published source in oracle/marching_cubes.py; the package itself is absent here). `method="tets"` (or
    VDN_MESH_METHOD=tets): rounds 3-5's marching tetrahedra - the same level set, a finer, different triangulation."""
    from vdn_hip import mesh
    method = method or os.environ.get("VDN_MESH_METHOD", "cubes")
    if method not in ("cubes", "tets"):
        raise ValueError("method must be 'cubes' or 'tets', got %r" % (method,))
    u = extract_fields_device(bound_min, bound_max, resolution, query_func)
    vertices, triangles = mesh.marching_cubes(u, threshold) if method == "cubes" else mesh.marching_tets(u, threshold)
    b_max_np = np.asarray([float(v) for v in bound_max])
    b_min_np = np.asarray([float(v) for v in bound_min])
    vertices = vertices.cpu().numpy().astype(np.float64) / (resolution - 1.0) * (b_max_np - b_min_np)[None, :] + b_min_np[None, :]
    return vertices, triangles.cpu().numpy()


_FUSE_ROUNDS = __import__("os").environ.get("VDN_FUSE_ROUNDS", "1") != "0"      # A/B switch: vdn_merge_upsample vs the two launches


def fuse_sdf_rounds():
    """VDN_FUSE_SDF_ROUNDS=0: the sampler's rounds as two launches each (SDF pass, merge + up-sample) instead of one."""
    import os
    return os.environ.get("VDN_FUSE_SDF_ROUNDS", "1") != "0"


def bg_compaction():
    """VDN_BG_COMPACT=0 evaluates every background sample as the reference does (A/B switch for the parity tests)."""
    import os
    return os.environ.get("VDN_BG_COMPACT", "1") != "0"


def background_active(rays_o, rays_d, mid_z, T, out=None):
    """The background samples render_core does not multiply by zero (vdn_background_active, include/vdn_render.h):
    -> (idx int32 [B*T], n int32 [1]) on the device. Inside-sphere samples get weight (1 - inside_sphere) = 0 for the
    NeRF++ output (renderer.py:284-299), so the background network skips them - same outputs, same gradients."""
    B, N = mid_z.shape
    dev = mid_z.device
    if out is None:
        out = (torch.empty(B * T, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev),
               torch.empty(B, dtype=torch.int32, device=dev))
    a = lib.VdnBackgroundActiveArgs()
    a.rays_o, a.rays_d, a.mid_z = rays_o.data_ptr(), rays_d.data_ptr(), mid_z.data_ptr()
    a.B, a.N, a.T = B, N, T
    a.active_idx, a.n_active, a.ray_counts = out[0].data_ptr(), out[1].data_ptr(), out[2].data_ptr()
    lib.call("vdn_background_active", a, _stream())
    return out[0], out[1]


class _RenderCoreFn(torch.autograd.Function):
    """Differentiable part of render() (render_core_outside + render_core at detached z) as one autograd
    node: forward and backward are the hand-written kernels driven by vdn_hip.train.TrainEngine."""

    @staticmethod
    def forward(ctx, engine, rays_o, rays_d, z, z_out, bgc, car, *params):
        # differentiable rays (learnable poses, poses.py:198-208): the backward then also returns d loss / d rays_o, rays_d
        # and d loss / d z, d z_out (render() chains the latter to near / far)
        ctx.ray_grads = any(ctx.needs_input_grad[1:5])
        ctx.set_materialize_grads(False)          # unused outputs (cdf_fine, gradients ...) arrive as None, not as zero tensors
        # Without ray gradients the step takes the Trainer's foreground work list (vdn_hip/train.py: skip_far): inside samples
        # beyond the relaxed sphere enter every output but `gradients` / `cdf_fine` through exact zeros, so the training launch,
        # the heads, the backward and the weight-gradient GEMM cover the listed samples only, and an inference launch on the
        # list's complement delivers `sdf` / `normals` there (rest_normals). Same outputs bit for bit; parameter gradients differ
        # by the dW summation order. backward() below re-runs the forward on every sample in the one case that needs the saves
        # everywhere: a loss on `gradients` or `cdf_fine` (the reference's own loss uses neither, dpt_runner.py:215-243).
        ctx.skipped = not ctx.ray_grads and os.environ.get("VDN_RENDER_FG_COMPACT", "1") != "0"
        ctx.fwd_args = (bgc, car)
        # a _TrainPlan (below): sampler + this forward were captured once as a HIP graph on the plan's fixed input buffers
        # (rays_o ... z_out ARE those buffers); one replay instead of ~20 launches issued from Python
        ctx.plan = plan = engine.__dict__.pop("_plan_call", None)
        if plan is not None:
            plan.replay_forward(engine)
        else:
            engine.forward(rays_o, rays_d, z, z_out, bgc, car, ray_grads=ctx.ray_grads, skip_far=ctx.skipped, rest_normals=ctx.skipped)
        ctx.skipped = ctx.skipped and engine._fg_compact
        ctx.engine, ctx.generation, ctx.n_params = engine, engine.generation, len(params)
        c = engine.outputs_clone()                # one copy: the engine's buffers are rewritten by the next forward
        color, weights, eik = c["color"], c["weights"], c["eik"][0]
        feats = c["feat_out"] if "feat_out" in c else color.new_zeros(0)
        # cdf_fine and gradients stay attached as in the reference (renderer.py:426-439; its own loss never uses them)
        cdf, normals = c["cdf"], c["normals"].view(engine.B, engine.N, 3).detach()       # (detach: a tensor of its own, not a view)
        aux = (c["inside"], c["bg_mid"] if engine.r.n_outside > 0 else c["mid_z"], c["eik"][1:3].detach())
        ctx.mark_non_differentiable(*aux)
        # s_val = 1 / inv_s for every ray (renderer.py:324, 420: a function of the variance alone) and the two reductions of
        # `weights` (renderer.py:309, 431) are outputs of this node as well: their adjoints are folded in by backward() below
        # instead of nine small torch launches per forward
        # The outputs share ONE cloned arena (one copy launch instead of twelve) as tensors of their own (TrainEngine.outputs_clone:
        # not views, separate version counters): in-place ops on them are allowed (color_fine.clamp_() ...), and a
        # saved output that is modified before backward() is still caught. (What remains of the sharing: a retained output keeps
        # the whole ~2 MB arena of its batch alive - INTEGRATION.md.)
        outs = (color, feats, weights, eik.detach(), cdf, normals, c["s_val"], c["wsum"], c["wmax"])
        ctx.save_for_backward(weights, rays_o, rays_d, z, z_out)
        return outs + aux

    @staticmethod
    def backward(ctx, g_color, g_feats, g_weights, g_eik, g_cdf, g_normals, g_sval, g_wsum, g_wmax, *unused):
        eng = ctx.engine
        if eng.generation != ctx.generation:
            raise RuntimeError("NeuSRenderer.render was called again before this result's backward(): the training "
                               "engine keeps the activations of the latest forward only")
        weights, rays_o, rays_d, z, z_out = ctx.saved_tensors
        if ctx.skipped and (g_cdf is not None or g_normals is not None):
            # adjoints on `cdf_fine` / `gradients` reach the samples the work list skipped: the saves are needed everywhere
            eng.forward(rays_o, rays_d, z, z_out, ctx.fwd_args[0], ctx.fwd_args[1])
            eng.generation = ctx.generation
        if g_wsum is not None:                    # weight_sum = weights.sum(-1, keepdim=True)
            g_weights = g_wsum.expand_as(weights) if g_weights is None else g_weights + g_wsum
        if g_wmax is not None:                    # weight_max = weights.max(-1, keepdim=True)[0]: to the (first) arg max
            hot = torch.zeros_like(weights).scatter_(1, weights.argmax(dim=-1, keepdim=True), g_wmax)
            g_weights = hot if g_weights is None else g_weights + hot
        g_feats = g_feats if (g_feats is not None and g_feats.numel() > 0) else None
        plan = ctx.plan
        if plan is not None and g_cdf is None and g_normals is None and not ctx.ray_grads and plan.replay_backward(eng, g_color, g_feats, g_weights, g_eik):
            pass                                  # (the captured launch sequence of eng.backward on staged adjoints)
        else:
            if g_weights is not None:
                g_weights = g_weights.contiguous()
            eng.backward(g_color, g_feats, g_weights, g_eik, g_cdf=g_cdf, g_gradients=g_normals)
        flat = eng.param_grads(clone=True)        # clones: the engine's buffers are reused by the next step
        assert len(flat) == ctx.n_params
        if g_sval is not None:
            # s_val[b] = 1 / clip(exp(10 var), 1e-6, 1e6): d / d var = -10 s_val where the clip passes
            var = eng.r.deviation_network.variance
            i = next(k for k, p in enumerate(eng.params) if p is var)
            inv_s = torch.exp(var.detach() * 10.0)
            passes = ((inv_s > 1e-6) & (inv_s < 1e6)).to(inv_s.dtype)
            flat[i] = flat[i] + (g_sval.sum() * (-10.0) * passes / inv_s).reshape(flat[i].shape)
        rays = (None,) * 4
        if ctx.ray_grads:
            w = eng.w
            rays = (w["d_rays_o"].clone(), w["d_rays_d"].clone(), w["d_z"].clone(),
                    w["d_z_out"].clone() if "d_z_out" in w else None)
            rays = tuple(g if need else None for g, need in zip(rays, ctx.needs_input_grad[1:5]))
        return (None,) + rays + (None, None) + tuple(flat)


# environment switches that select launches inside the captured region: a plan is valid for one setting of them
_PLAN_ENV = ("VDN_RENDER_FG_COMPACT", "VDN_FG_COMPACT", "VDN_BG_COMPACT", "VDN_FUSED_PREP", "VDN_TRAIN_COLOR_FUSED", "VDN_SDF_TAIL",
             "VDN_SDF_TAIL_ROW0", "VDN_SDF_TAIL_MAX", "VDN_BWD_SPLIT_DW", "VDN_SDF_BWD_SPLIT", "VDN_FUSE_ROUNDS", "VDN_FUSE_SDF_ROUNDS")
# engine attributes forward() leaves for backward(): a replayed forward restores them as the capture left them
_PLAN_STATE = ("_ctx", "_fg_compact", "_bg_compact", "_composite_bwd_done", "_bwd_train", "_pending", "_ray_grads", "_color_fused",
               "_car_dev", "_fwd_rays", "_fused_keep")


class _TrainPlan:
    """render() under grad of one batch size as HIP graphs (round 6; what RenderPlan is for the inference path). An unchanged
    dpt_runner.py spends more host time per step than the device needs: ~0.5 ms of Python in render() and ~0.25 ms in
    this package's share of loss.backward(), most of it ~45 ctypes launches whose argument blocks are refilled field by field.
    The plan captures, once per (renderer, batch size, configuration):
      * forward: the sampler (renderer.py:334-386, its jitter drawn by a generator call INSIDE the graph) + TrainEngine.forward on
        fixed input buffers; the annealing ratio comes from a device scalar (VdnCompositeArgs.cos_anneal_dev), so the runner's
        schedule (dpt_runner.py:304-308) needs no re-capture;
      * backward, per pattern of present adjoints: TrainEngine.backward on staged adjoints, side-stream work included
        (fork / join events become graph dependencies).
    A step is then: one multi-tensor copy of the inputs, one replay, one copy of the outputs' arena; in backward one copy per
    adjoint, one replay, one multi-tensor copy of the gradients. Row counts of the work lists are device-side and every grid is
    sized for the full batch, so the graphs are independent of the data. Built by NeuSRenderer at the SECOND call with a key (the
    first ran eagerly: lazy workspaces, kernel attributes and weight images exist); `VDN_RENDER_GRAPHS=0` turns it off."""

    def __init__(self, rend, eng, has_bg, perturb, inject):
        B, dev, O = eng.B, eng.dev, rend.n_outside
        f = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=dev)
        self.rays_o, self.rays_d, self.near, self.far = f(B, 3), f(B, 3), f(B), f(B)
        self.rays_d[:, 2] = 1.0
        self.far.fill_(1.0)
        self.bgc = f(3) if has_bg else None
        self.car, self._car_host = f(1), 0.0
        self.t_rand, self.t_rand_out = (f(B, 1), f(B, O) if O > 0 else None) if inject else (None, None)
        self.perturb = perturb
        self._dst = [self.rays_o, self.rays_d, self.near, self.far] + ([self.bgc] if has_bg else []) + \
                    ([self.t_rand] + ([self.t_rand_out] if O > 0 else []) if inject else [])
        self.bwd = {}
        skip = os.environ.get("VDN_RENDER_FG_COMPACT", "1") != "0"
        g = torch.cuda.CUDAGraph()
        with torch.no_grad(), torch.cuda.graph(g):
            z, z_out = rend._sample(self.rays_o, self.rays_d, self.near, self.far, perturb, self.t_rand, self.t_rand_out, None)
            z = z.contiguous()
            eng.forward(self.rays_o, self.rays_d, z, z_out, self.bgc, 0.0, skip_far=skip, rest_normals=skip, cos_anneal_dev=self.car)
        self.fwd, self.z, self.z_out = g, z, z_out
        self.state = {k: eng.__dict__.get(k) for k in _PLAN_STATE}

    def stage(self, rays_o, rays_d, near, far, background_rgb, car, t_rand, t_rand_out):
        src = [rays_o, rays_d, near, far]
        if self.bgc is not None:
            src.append(background_rgb.detach().reshape(-1))
        if self.t_rand is not None:
            src.append(t_rand.detach().reshape(self.t_rand.shape))
            if self.t_rand_out is not None:
                src.append(t_rand_out.detach().reshape(self.t_rand_out.shape))
        torch._foreach_copy_(self._dst, src)
        car = float(car)
        if car != self._car_host:
            self.car.fill_(car)
            self._car_host = car

    def replay_forward(self, eng):
        self.fwd.replay()
        eng.__dict__.update(self.state)
        eng._ctx = self.state["_ctx"][:3] + (self._car_host,) + self.state["_ctx"][4:]
        eng.generation = getattr(eng, "generation", 0) + 1

    def replay_backward(self, eng, g_color, g_feats, g_weights, g_eik):
        """-> False when this call has to run eagerly (the engine has not run a backward yet)."""
        pat = (g_color is not None, g_feats is not None, g_weights is not None, g_eik is not None)
        ent = self.bwd.get(pat)
        if ent is None or ent is False:
            if not getattr(eng, "_bwd_warm", False) or ent is False:
                return False
            B, dev = eng.B, eng.dev
            f = lambda *sh: torch.zeros(*sh, dtype=torch.float32, device=dev)
            bufs = (f(B, 3) if pat[0] else None, f(B, 96) if pat[1] else None, f(B, eng.T) if pat[2] else None, f(1) if pat[3] else None)
            g = torch.cuda.CUDAGraph()
            try:
                with torch.no_grad(), torch.cuda.graph(g):
                    eng.backward(bufs[0], bufs[1], bufs[2], bufs[3])
            except Exception as e:                           # (as for the forward: report once, this pattern stays eager)
                import warnings
                self.bwd[pat] = False
                eng.__dict__.update(self.state)
                warnings.warn("NeuSRenderer: capturing the backward of render() as a HIP graph failed (%s: %s); eager launches" % (type(e).__name__, e))
                return False
            eng.__dict__.update(self.state)
            ent = self.bwd[pat] = (g, bufs)
        g, bufs = ent
        for dst, src in zip(bufs, (g_color, g_feats, g_weights, g_eik)):
            if dst is not None:
                dst.copy_(src.reshape(dst.shape) if src.numel() == dst.numel() else src)       # (g_weights may be an expanded view)
        g.replay()
        return True


class NeuSRenderer:
    def __init__(self, nerf, sdf_network, deviation_network, color_network, depth_network, n_samples, n_importance,
                 n_outside, up_sample_steps, perturb, precision=None):
        """Reference signature (renderer.py:78-88) plus ONE optional keyword the reference does not have:
        precision = "fp32" | "bf16" | None - the kernels every network of this renderer runs on. "fp32": exact-f32 MFMA kernels, the
        path the 1e-4 parity tests hold on (the default). "bf16": bf16-operand / fp32-accumulate kernels, the throughput path
        (tests/test_gpu_bf16.py states its bounds). None: the environment variable VDN_PRECISION, read once here (default fp32) -
        so an UNCHANGED dpt_runner.py runs on the bf16 kernels with `VDN_PRECISION=bf16 python dpt_runner.py ...`, or with the line
        `precision = bf16` in the conf's `model.neus_renderer` block (the runner splats that block into this constructor)."""
        if precision is None:
            precision = os.environ.get("VDN_PRECISION", "").strip().lower() or None
        if precision is not None:
            if precision not in ("fp32", "bf16"):
                raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (precision,))
            for m in (nerf, sdf_network, color_network, depth_network):
                if m is not None:
                    m.precision = precision
        self.nerf = nerf
        self.sdf_network = sdf_network
        self.deviation_network = deviation_network
        self.color_network = color_network
        self.depth_network = depth_network
        self.n_samples = n_samples
        self.n_importance = n_importance
        self.n_outside = n_outside
        self.up_sample_steps = up_sample_steps
        self.perturb = perturb
        if n_importance > 0 and (n_importance % up_sample_steps != 0 or n_importance // up_sample_steps > 64):
            raise ValueError("n_importance must be a multiple of up_sample_steps with at most 64 samples per round")
        if n_samples + n_importance + n_outside > 256 or n_outside > 64:
            raise ValueError("the per-ray kernels handle up to 256 samples per ray and n_outside <= 64")
        if depth_network is not None and n_outside > 0 and not getattr(nerf, "gen_depth_feats", False):
            # renderer.py:295-299 blends the VDN features with the background's only when the NeRF produces them; the
            # kernels implement the shipped pairing (womsk_white_wdepth: both on)
            raise ValueError("a depth_network needs a background NeRF with gen_depth_feats=True (the shipped wdepth configuration)")
        self._const_cache = {}

    @property
    def precision(self):
        """The kernels' precision when all networks agree, else "mixed". Assigning sets every network (one line instead of four)."""
        ps = {m.precision for m in (self.nerf, self.sdf_network, self.color_network, self.depth_network) if m is not None}
        return ps.pop() if len(ps) == 1 else "mixed"

    @precision.setter
    def precision(self, value):
        if value not in ("fp32", "bf16"):
            raise ValueError("precision must be 'fp32' or 'bf16', got %r" % (value,))
        for m in (self.nerf, self.sdf_network, self.color_network, self.depth_network):
            if m is not None:
                m.precision = value

    # constant vectors whose rounding must be torch.linspace's (renderer.py:335,340,352-354; 53)
    def _consts(self, dev):
        c = self._const_cache.get(dev)
        if c is None:
            c = {"lin_samples": torch.linspace(0.0, 1.0, self.n_samples).to(dev)}
            if self.n_outside > 0:
                zo = torch.linspace(1e-3, 1.0 - 1.0 / (self.n_outside + 1.0), self.n_outside)
                mids = .5 * (zo[1:] + zo[:-1])
                c["lin_outside"] = zo.to(dev)
                c["out_upper"] = torch.cat([mids, zo[-1:]], -1).to(dev)
                c["out_lower"] = torch.cat([zo[:1], mids], -1).to(dev)
            if self.n_importance > 0:
                n = self.n_importance // self.up_sample_steps
                c["u"] = torch.linspace(0.5 / n, 1.0 - 0.5 / n, steps=n).to(dev)
            self._const_cache[dev] = c
        return c

    # -------------------------------------------------------------------------------------
    def _sample(self, rays_o, rays_d, near, far, perturb, t_rand, t_rand_out, z_vals_inject, defer_last_merge=False):
        """renderer.py:334-386 -> z [B,N] (sorted inside samples), z_out [B,O] or None.
        defer_last_merge (the training engine's fused preparation, vdn_train_prep): the final cat_z_vals - which evaluates
        no sdf (renderer.py:379-385) - is left to the caller: self._pending_merge = (new_z [B,n_imp], M_old) and the first
        M_old columns of z are the sorted row so far."""
        self._pending_merge = None
        B, dev = rays_o.shape[0], rays_o.device
        S, I, O = self.n_samples, self.n_importance, self.n_outside
        N = S + I
        c = self._consts(dev)
        st = _stream()
        z = torch.empty(B, N, dtype=torch.float32, device=dev)
        z_out = torch.empty(B, O, dtype=torch.float32, device=dev) if O > 0 else None
        a = lib.VdnCoarseArgs()
        a.near, a.far, a.lin_samples = near.data_ptr(), far.data_ptr(), c["lin_samples"].data_ptr()
        a.z, a.B, a.n_samples, a.n_outside, a.z_ld = z.data_ptr(), B, S, O, N
        if O > 0:
            a.lin_outside, a.out_lower, a.out_upper = (c[k].data_ptr() for k in ("lin_outside", "out_lower", "out_upper"))
            a.z_out = z_out.data_ptr()
        if perturb > 0:
            if t_rand is None and t_rand_out is None and O > 0:
                # the two uniform draws of renderer.py:348,355 from one generator call, which covers the next 16 batches of this
                # size (a 5-us launch in front of the sampler, on a 360-us render)
                # - as long as nobody else touched the generator in between: re-seeding it, or drawing from it, starts a new block
                # (a RenderPlan's capture draws its own inside the graph: every replay then gets fresh jitter)
                if torch.cuda.is_current_stream_capturing():
                    u = torch.rand(B * (1 + O), device=dev)
                else:
                    gen = torch.cuda.default_generators[dev.index if dev.index is not None else torch.cuda.current_device()]
                    state = (gen.initial_seed(), gen.get_offset())
                    jit = self.__dict__.get("_jitter")
                    if (jit is None or jit[0].shape[1] != B * (1 + O) or jit[0].device != dev or jit[1] >= jit[0].shape[0]
                            or jit[2] != state):
                        block = torch.rand(16, B * (1 + O), device=dev)
                        jit = self.__dict__["_jitter"] = [block, 0, (gen.initial_seed(), gen.get_offset())]
                    u = jit[0][jit[1]]
                    jit[1] += 1
                t_rand, t_rand_out = u[:B].view(B, 1), u[B:].view(B, O)
            if t_rand is None:
                t_rand = torch.rand([B, 1], device=dev)                  # renderer.py:348
            a.t_rand = t_rand.data_ptr()
            if O > 0:
                if t_rand_out is None:
                    t_rand_out = torch.rand([B, O], device=dev)          # renderer.py:355
                a.t_rand_out = t_rand_out.data_ptr()
        lib.call("vdn_coarse_z", a, st)
        if I > 0:
            if z_vals_inject is not None:
                z = z_vals_inject.contiguous()
            else:
                sdf = torch.empty(B, N, dtype=torch.float32, device=dev)
                n_imp = I // self.up_sample_steps
                new_z = torch.empty(B, n_imp, dtype=torch.float32, device=dev)
                new_sdf = torch.empty(B, n_imp, dtype=torch.float32, device=dev)
                M = S

                def upsample_args(i, M):
                    u = lib.VdnUpsampleArgs()
                    u.rays_o, u.rays_d, u.z, u.sdf, u.u = (t.data_ptr() for t in (rays_o, rays_d, z, sdf, c["u"]))
                    u.new_z, u.inv_s, u.B, u.M, u.ld, u.n_imp = new_z.data_ptr(), float(64 * 2 ** i), B, M, N, n_imp
                    return u
                # first pass (renderer.py:369-370) and first round, in one launch where the kernel covers the shape
                if not (_FUSE_ROUNDS and fuse_sdf_rounds() and self.sdf_network._run_first((rays_o, rays_d, z[:, :S]), sdf[:, :S], upsample_args(0, M))):
                    self.sdf_network._run(0, rays=(rays_o, rays_d, z[:, :S]), sdf_out=sdf[:, :S])
                    lib.call("vdn_upsample_round", upsample_args(0, M), st)
                for i in range(self.up_sample_steps):
                    last = (i + 1 == self.up_sample_steps)
                    m = lib.VdnMergeArgs()
                    m.z, m.new_z, m.z_out = z.data_ptr(), new_z.data_ptr(), z.data_ptr()
                    m.B, m.M, m.K, m.ld, m.ld_out = B, M, n_imp, N, N
                    if not last:
                        m.sdf, m.new_sdf, m.sdf_out = sdf.data_ptr(), new_sdf.data_ptr(), sdf.data_ptr()
                        # the whole round in one launch where the kernel covers the shape (bf16, 16 new samples per ray)
                        if _FUSE_ROUNDS and fuse_sdf_rounds() and self.sdf_network._run_round(
                                (rays_o, rays_d, new_z), new_sdf, m, upsample_args(i + 1, M + n_imp)):
                            M += n_imp
                            continue
                        self.sdf_network._run(0, rays=(rays_o, rays_d, new_z), sdf_out=new_sdf)         # renderer.py:201
                        # cat_z_vals of this round + up_sample of the next one (renderer.py:372-386) in one launch: the new
                        # samples of round i are read before those of round i+1 are written over them
                        if _FUSE_ROUNDS:
                            lib.call("vdn_merge_upsample", m, upsample_args(i + 1, M + n_imp), st)
                        else:
                            lib.call("vdn_merge_sorted", m, st)
                            lib.call("vdn_upsample_round", upsample_args(i + 1, M + n_imp), st)
                    elif defer_last_merge:
                        self._pending_merge = (new_z, M)
                    else:
                        lib.call("vdn_merge_sorted", m, st)
                    M += n_imp
        else:
            z = z[:, :S] if N == S else z
        return z, z_out

    def _sections(self, z, n, sample_dist):
        B, dev = z.shape[0], z.device
        dists = torch.empty(B, n, dtype=torch.float32, device=dev)
        mid = torch.empty(B, n, dtype=torch.float32, device=dev)
        a = lib.VdnSectionArgs()
        a.z, a.dists, a.mid_z, a.sample_dist, a.B, a.n, a.ld = z.data_ptr(), dists.data_ptr(), mid.data_ptr(), sample_dist, B, n, z.stride(0)
        lib.call("vdn_sections", a, _stream())
        return dists, mid

    def render(self, rays_o, rays_d, near, far, perturb_overwrite=-1, background_rgb=None, cos_anneal_ratio=0.0,
               depth_before_color=False, t_rand=None, t_rand_out=None, z_vals_inject=None):
        for t, n in ((rays_o, "rays_o"), (rays_d, "rays_d"), (near, "near"), (far, "far")):
            _require_gpu(t, "NeuSRenderer.render " + n)
        if depth_before_color and self.depth_network is None:
            depth_before_color = False               # renderer.py:245-248: only inside `if self.depth_network is not None`
        if bool(depth_before_color) != (self.color_network.conf["d_feature"] == 352):
            raise ValueError("depth_before_color=%s needs a colour network with d_feature = %d (renderer.py:247-248: it is fed "
                             "cat([feature_vector, VDN output]))" % (bool(depth_before_color), 352 if depth_before_color else 256))
        if rays_o.dim() != 2 or rays_o.shape[1] != 3 or rays_d.shape != rays_o.shape:
            raise ValueError("rays_o / rays_d must be [B,3]")
        B, dev = rays_o.shape[0], rays_o.device
        if near.numel() != B or far.numel() != B:
            raise ValueError("near / far must hold one value per ray")
        # learnable poses (poses.py:198-208): rays, near and far may carry a graph; the sampler works on detached copies
        attached = (rays_o, rays_d, near, far) if (torch.is_grad_enabled() and any(t.requires_grad for t in (rays_o, rays_d, near, far))) else None
        rays_o = rays_o.detach().contiguous()
        rays_d = rays_d.detach().contiguous()
        near = near.detach().reshape(B).contiguous()
        far = far.detach().reshape(B).contiguous()
        S, I, O = self.n_samples, self.n_importance, self.n_outside
        N = S + I
        T = N + O
        sample_dist = 2.0 / S                                                    # renderer.py:334
        perturb = self.perturb if perturb_overwrite < 0 else perturb_overwrite
        if B == 0:
            raise ValueError("empty ray batch")
        st = _stream()

        params = self._all_parameters()
        differentiable = torch.is_grad_enabled() and (attached is not None or any(p.requires_grad for p in params))
        # inference with the background work list: the last round's merge rides in vdn_train_prep (as in the training engine)
        defer = (not differentiable) and O > 0 and bg_compaction()
        if differentiable:
            # an optimizer step changed every network at once: ONE weight-norm launch and ONE image-build launch for all of
            # them here instead of two per network as each is first used (vdn_hip/images.py::refresh_together)
            nets = [m for m in (self.nerf, self.sdf_network, self.color_network, self.depth_network) if m is not None]
            for m in nets:
                m._join_trainer()
            from vdn_hip import images as _images
            ims = [m._image_state() for m in nets]
            stale = [im for im in ims if im.stale()]
            if len(stale) > 1:
                _images.refresh_together(stale, st, self.__dict__.setdefault("_img_tables", {}))
            elif stale:
                stale[0].refresh(st)
            # ... and nothing changes a parameter before this call returns: the dozen _images() calls below (each sampler pass,
            # each network of the engine) skip their own staleness test (a walk over the parameters' versions and addresses)
            _images.trust(ims)
            try:
                if attached is None and z_vals_inject is None:
                    # the second and later calls of a configuration: sampler + forward as one HIP-graph replay (_TrainPlan)
                    eng = self._engine_for(B, dev, params)
                    plan = self._plan_for(eng, background_rgb, perturb, t_rand, t_rand_out)
                    if plan is not None:
                        plan.stage(rays_o, rays_d, near, far, background_rgb, cos_anneal_ratio, t_rand, t_rand_out)
                        return self._render_train(plan.rays_o, plan.rays_d, plan.z, plan.z_out, plan.bgc, cos_anneal_ratio, params, plan=plan, eng=eng)
                z, z_out = self._sample(rays_o, rays_d, near, far, perturb, t_rand, t_rand_out, z_vals_inject, defer_last_merge=defer)
                if attached is not None:
                    rays_o, rays_d, z, z_out = self._attach_rays(attached, near, far, z, z_out)
                return self._render_train(rays_o, rays_d, z, z_out, background_rgb, cos_anneal_ratio, params)
            finally:
                _images.trust(None)
        z, z_out = self._sample(rays_o, rays_d, near, far, perturb, t_rand, t_rand_out, z_vals_inject, defer_last_merge=defer)
        bg_density = bg_rgb = bg_feat = bg_dists = bg_mid = None
        if O > 0 and bg_compaction():
            # z_feed (renderer.py:390-391), both section sets and the background work list in two launches (vdn_train_prep)
            f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
            z = z.contiguous()
            z_feed, dists, mid_z, bg_dists, bg_mid = f32(B, T), f32(B, N), f32(B, N), f32(B, T), f32(B, T)
            active3 = (torch.empty(B * T, dtype=torch.int32, device=dev), torch.empty(1, dtype=torch.int32, device=dev),
                       torch.empty(B, dtype=torch.int32, device=dev))
            tp = lib.VdnTrainPrepArgs()
            tp.rays_o, tp.rays_d, tp.z, tp.z_out, tp.z_feed = (t.data_ptr() for t in (rays_o, rays_d, z, z_out, z_feed))
            tp.B, tp.N, tp.T, tp.z_ld, tp.sample_dist, tp.fg_radius = B, N, T, z.stride(0), sample_dist, 1.2
            tp.dists, tp.mid_z, tp.bg_dists, tp.bg_mid = (t.data_ptr() for t in (dists, mid_z, bg_dists, bg_mid))
            tp.bg_active_idx, tp.bg_n_active, tp.bg_ray_counts = (t.data_ptr() for t in active3)
            if self._pending_merge is not None:
                tp.new_z, tp.M_old = self._pending_merge[0].data_ptr(), self._pending_merge[1]
            lib.call("vdn_train_prep", tp, st)
            bg_density, bg_rgb, bg_feat = self.nerf._run(rays=(rays_o, rays_d, bg_mid), active=active3[:2],
                                                         scratch=not torch.cuda.is_current_stream_capturing())
        else:
            dists, mid_z = self._sections(z, N, sample_dist)
            if O > 0:                                                            # renderer.py:389-397
                z_feed = torch.empty(B, T, dtype=torch.float32, device=dev)
                m = lib.VdnMergeArgs()
                m.z, m.new_z, m.z_out = z.data_ptr(), z_out.data_ptr(), z_feed.data_ptr()
                m.B, m.M, m.K, m.ld, m.ld_out = B, N, O, z.stride(0), T
                lib.call("vdn_merge_sorted", m, st)
                bg_dists, bg_mid = self._sections(z_feed, T, sample_dist)
                bg_density, bg_rgb, bg_feat = self.nerf._run(rays=(rays_o, rays_d, bg_mid), active=None)

        return self._shade(rays_o, rays_d, dists, mid_z, (bg_density, bg_rgb, bg_feat, bg_dists, bg_mid) if O > 0 else None,
                           background_rgb, cos_anneal_ratio, depth_before_color)

    def _shade(self, rays_o, rays_d, dists, mid_z, bg, background_rgb, cos_anneal_ratio, depth_before_color=False):
        """render_core of the reference on sampled rays (renderer.py:239-315; SURVEY.md 8d "C2": PE + SDF MLP + gradient sweep,
        colour / VDN heads, NeuS alpha + compositing) on B x N points: ONE launch where vdn_shade_fused_bf16 covers the shape (bf16,
        128 samples per ray; three with the VDN head), else the SDF kernel, the heads, the compositor and the eikonal reduce.
        `bg`: the background pass' (density, rgb, feat, dists, mid_z) over all N + n_outside sections, or None."""
        B, N = mid_z.shape
        dev, st = mid_z.device, _stream()
        bg_density, bg_rgb, bg_feat, bg_dists, bg_mid = bg if bg is not None else (None,) * 5
        O = 0 if bg is None else bg_dists.shape[1] - N
        T = N + O
        fused = self._fused_shading(N, depth_before_color)
        sampled_feat = None
        if not fused:
            sdf, feat, normals = self.sdf_network._run(1, rays=(rays_o, rays_d, mid_z))          # renderer.py:239-243
            if self.depth_network is not None:                                       # renderer.py:245-249
                sampled_feat = self.depth_network._run(normals, feat, rays=(rays_o, rays_d, mid_z))
            sampled_color = self.color_network._run(normals, feat, rays=(rays_o, rays_d, mid_z),        # renderer.py:247-251
                                                    extra=sampled_feat if depth_before_color else None)

        a = lib.VdnCompositeArgs()
        f32 = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        weights, alpha, cdf, inside = f32(B, T), f32(B, T), f32(B, N), f32(B, N)
        color, wsum, wmax, s_val = f32(B, 3), f32(B, 1), f32(B, 1), f32(B, 1)
        eik_partial, eik = f32(B, 2), f32(3)
        feat_out = f32(B, 96) if sampled_feat is not None else None
        a.rays_o, a.rays_d = rays_o.data_ptr(), rays_d.data_ptr()
        a.dists, a.mid_z = dists.data_ptr(), mid_z.data_ptr()
        if not fused:
            a.sdf, a.normals, a.color = sdf.data_ptr(), normals.data_ptr(), sampled_color.data_ptr()
        a.variance = self.deviation_network.variance.data_ptr()
        if sampled_feat is not None:
            a.feat, a.feat_out, a.feat_ch = sampled_feat.data_ptr(), feat_out.data_ptr(), 96
        if O > 0:
            a.bg_density, a.bg_rgb, a.bg_dists = bg_density.data_ptr(), bg_rgb.data_ptr(), bg_dists.data_ptr()
            if sampled_feat is not None:
                if bg_feat is None:
                    raise ValueError("depth_network is set but the NeRF was built with gen_depth_feats=False")
                a.bg_feat = bg_feat.data_ptr()
        bgc = None
        if background_rgb is not None:
            bgc = background_rgb.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
            if bgc.numel() != 3:
                raise ValueError("background_rgb must have 3 values")
            a.background_rgb = bgc.data_ptr()
        a.cos_anneal_ratio = float(cos_anneal_ratio)
        a.B, a.N, a.T = B, N, T
        a.weights, a.alpha_out, a.cdf, a.inside_sphere = weights.data_ptr(), alpha.data_ptr(), cdf.data_ptr(), inside.data_ptr()
        a.color_out, a.weight_sum, a.weight_max, a.s_val = color.data_ptr(), wsum.data_ptr(), wmax.data_ptr(), s_val.data_ptr()
        a.eik_partial, a.eik_out = eik_partial.data_ptr(), eik.data_ptr()
        if fused:
            # renderer.py:239-315 in one launch (csrc/k_sdf_fwd2.h MODE 2): a 128-point workgroup is one ray - SDF network +
            # gradient sweep, the colour head on the feature vector in registers, the ray's compositing from LDS, the eikonal sums
            # by the last ray
            sn = self.sdf_network
            sa = lib.VdnSdfArgs()
            sa.blob = sn._images().blobs["full"].data_ptr()
            sa.rays_o, sa.rays_d, sa.z = rays_o.data_ptr(), rays_d.data_ptr(), mid_z.data_ptr()
            sa.n_per_ray, sa.z_ld, sa.sdf_ld, sa.P, sa.scale = N, mid_z.stride(0), N, B * N, float(sn.scale)
            sdf, normals = f32(B * N), f32(B * N, 3)
            sa.sdf, sa.normals = sdf.data_ptr(), normals.data_ptr()
            feat = None
            ticket = self.__dict__.get("_shade_ticket")
            if ticket is None or ticket.device != dev:
                ticket = self.__dict__["_shade_ticket"] = torch.zeros(1, dtype=torch.int32, device=dev)
            cn = self.color_network
            if sn.precision == "bf16":
                if self.depth_network is not None:       # the VDN head reads the feature plane: the fused kernel also writes it out
                    feat = torch.empty(B * N, 256, dtype=torch.bfloat16, device=dev)
                    sa.feat = feat.data_ptr()
                lib.call("vdn_shade_fused_bf16", sa, lib.ptr(cn._images().blobs["c2"]), int(cn.squeeze_out), a, lib.ptr(ticket), st)
            else:
                # the exact-fp32 kernels (csrc/k_shade_f32.h): the same three stages back to back in the ray's workgroup, handing the
                # feature vector, the normal and the colour over through these buffers
                sa.w8row = sn._images().weff_view("lin8").data_ptr()
                feat, S = f32(B * N, 256), f32(8, B * N, 256)
                sa.feat, sa.S = feat.data_ptr(), S.data_ptr()
                sampled_color = f32(B * N, 3)
                ca = lib.VdnRenderNetArgs()
                ca.blob = cn._images().blobs["fwd"].data_ptr()
                ca.rays_o, ca.rays_d, ca.z, ca.n_per_ray = sa.rays_o, sa.rays_d, sa.z, N
                ca.normals, ca.feat, ca.out = sa.normals, sa.feat, sampled_color.data_ptr()
                ca.P, ca.d_out, ca.squeeze_out = B * N, 3, int(cn.squeeze_out)
                a.sdf, a.normals, a.color = sa.sdf, sa.normals, ca.out
                lib.call("vdn_shade_fused_f32", sa, ca, a, lib.ptr(ticket), st)
            if self.depth_network is not None:
                # renderer.py:245-249, 306-308: the 96 VDN channels keep their own launches - the head on the feature plane and the
                # normals the fused kernel wrote, then their weighted sums with the weights it composited
                if bg is not None and bg_feat is None:
                    raise ValueError("depth_network is set but the NeRF was built with gen_depth_feats=False")
                sampled_feat = self.depth_network._run(normals, feat, rays=(rays_o, rays_d, mid_z))
                feat_out = f32(B, 96)
                a.feat, a.feat_out, a.feat_ch = sampled_feat.data_ptr(), feat_out.data_ptr(), 96
                if O > 0:
                    a.bg_feat = bg_feat.data_ptr()
                lib.call("vdn_feat_composite", a, st)
        else:
            lib.call("vdn_alpha_composite_fwd", a, st)

        self.last_eikonal_terms = eik[1:3]      # (numerator, denominator) for the data-parallel reduction
        return {
            "render_feats": feat_out,
            "color_fine": color,
            "s_val": s_val,
            "cdf_fine": cdf,
            "weight_sum": wsum,
            "weight_max": wmax,
            "gradients": normals.view(B, N, 3),
            "weights": weights,
            "z_vals": bg_mid if O > 0 else mid_z,                                # renderer.py:421-424
            "gradient_error": eik[0],
            "inside_sphere": inside,
        }

    def _fused_shading(self, N, depth_before_color=False):
        """True where vdn_shade_fused_bf16 / vdn_shade_fused_f32 cover the configuration: rays of exactly 128 inside samples (one
        workgroup per ray), the shipped 'idr' colour head (d_feature 256, d_out 3), both networks in one precision; a VDN head's 96
        channels keep their own two launches behind it. VDN_SHADE_FUSED=0 forces the separate launches (SDF / colour / compositor / eikonal reduce)."""
        cn, sn = self.color_network, self.sdf_network
        return (os.environ.get("VDN_SHADE_FUSED", "1") != "0" and N == 128 and not depth_before_color
                and sn.precision == cn.precision and sn.precision in ("bf16", "fp32") and cn.conf.get("mode") == "idr"
                and cn.conf.get("d_feature") == 256 and cn.conf.get("d_out") == 3)

    def shade_launches(self):
        """Launches of _shade (renderer.py:239-315) on the current configuration, for the bench's C2 leg."""
        vdn = 2 if self.depth_network is not None else 0               # the VDN head and the weighted sums of its 96 channels
        if self._fused_shading(self.n_samples + self.n_importance):
            return 1 + vdn
        return 4 + vdn      # SDF, colour, compositor, eikonal reduce

    def plan(self, batch, **render_kwargs):
        """A replayable whole-batch render() for `batch` rays with fixed keyword arguments -> RenderPlan."""
        return RenderPlan(self, batch, **render_kwargs)

    def _all_parameters(self):
        """dpt_runner.py:121-130 order: nerf, sdf, variance, colour, (vdn)."""
        ps = []
        for m in (self.nerf, self.sdf_network, self.deviation_network, self.color_network, self.depth_network):
            if m is not None:
                lib.module_params(m, ps)
        return ps

    def _attach_rays(self, attached, near_d, far_d, z, z_out):
        """Re-attach the sampler's detached depths to near / far exactly where the reference's graph has them:
        * z_vals = near + (far - near) * linspace (+ a jitter independent of them) is built with autograd on
          (renderer.py:335-349), but with importance sampling the whole up-sampling loop - including cat_z_vals, which
          REPLACES z_vals by the sorted concatenation - runs under torch.no_grad() (367-386): the inside depths that reach
          render_core then carry no graph at all. Only with n_importance = 0 do they depend on (near, far), with weights
          (1 - t_i, t_i);
        * z_vals_outside = far / flip(...) + 1 / n_samples (359) is outside the no_grad block: far * c_k + const.
        The values stay the kernels' own: only zero-valued differences carrying the graph are added."""
        rays_o, rays_d, near, far = attached
        B, dev = z.shape[0], z.device
        S = self.n_samples
        near, far = near.reshape(B, 1), far.reshape(B, 1)
        z_att = z
        if self.n_importance == 0:
            zc = near + (far - near) * self._consts(dev)["lin_samples"][None, :]   # renderer.py:335-336 (graph only)
            z_att = z + (zc - zc.detach())
        z_out_att = z_out
        if z_out is not None:
            fd = far_d.reshape(B, 1)
            # z_out = far * c + 1 / n_samples (renderer.py:359); a ray with far == 0 has z_out = 1 / n_samples for every c: its
            # (zero-valued) graph term gets c = 0 instead of 0 / 0
            c = torch.where(fd != 0, (z_out - 1.0 / S) / torch.where(fd != 0, fd, torch.ones_like(fd)), torch.zeros_like(z_out))
            z_out_att = z_out + (far - far.detach()) * c
        return rays_o.contiguous(), rays_d.contiguous(), z_att, z_out_att

    def _engine_for(self, B, dev, params):
        from vdn_hip.train import TrainEngine
        engines = self.__dict__.setdefault("_engines", {})
        pkey = (dev, tuple(p.data_ptr() for p in params), self.precision)
        key = (B,) + pkey
        eng = engines.pop(key, None)
        if eng is None:
            # An engine is built for one batch size (its workspaces are a few GB at B = 512, of 288). The runner alternates between
            # its training batch and the chunks of its image loops, whose last chunk per image is ragged (dpt_runner.py:439-445):
            # the three most recently used sizes stay alive instead of being rebuilt twice per image. Engines of re-allocated
            # parameters go at once.
            for k in [k for k in engines if k[1:] != pkey]:
                del engines[k]
            # VDN_RENDER_ENGINES=1..3 (default 3) bounds how many stay alive: on a smaller or shared device one engine per renderer
            # is the round-4 memory footprint, at the price of rebuilding when the batch size changes
            keep = min(3, max(1, int(os.environ.get("VDN_RENDER_ENGINES", "3") or 3)))
            while len(engines) >= keep:
                del engines[next(iter(engines))]             # (dicts keep insertion order: the least recently used one)
            eng = TrainEngine(self, B, dev)
        engines[key] = eng                                   # (re-inserted: most recently used)
        return eng

    def _plan_for(self, eng, background_rgb, perturb, t_rand, t_rand_out):
        """The _TrainPlan of this call's configuration, or None (first call with the configuration, graphs off, jitter injected
        only in part, a capture already in progress)."""
        if os.environ.get("VDN_RENDER_GRAPHS", "1") == "0" or torch.cuda.is_current_stream_capturing():
            return None
        O = self.n_outside
        inject = t_rand is not None or t_rand_out is not None
        if inject and (t_rand is None or (O > 0 and t_rand_out is None) or perturb <= 0):
            return None
        key = (background_rgb is not None, float(perturb), inject, tuple(os.environ.get(k) for k in _PLAN_ENV))
        plans = eng.__dict__.setdefault("_plans", {})
        plan = plans.get(key)
        if plan is None:
            seen = eng.__dict__.setdefault("_plan_seen", {})
            seen[key] = seen.get(key, 0) + 1
            if seen[key] < 2:
                return None                                  # the first call of a configuration runs eagerly (and warms everything up)
            if seen[key] < 0:
                return None                                  # (a capture of this configuration failed before: eager from then on)
            while len(plans) >= 4:
                del plans[next(iter(plans))]
            try:
                plan = plans[key] = _TrainPlan(self, eng, key[0], float(perturb), inject)
            except Exception as e:                           # never let the optimisation break a render(): report once, stay eager
                import warnings
                seen[key] = -(1 << 30)
                warnings.warn("NeuSRenderer: capturing render() as a HIP graph failed (%s: %s); this configuration keeps the eager "
                              "launches (VDN_RENDER_GRAPHS=0 silences the attempt)" % (type(e).__name__, e))
                return None
        return plan

    def _render_train(self, rays_o, rays_d, z, z_out, background_rgb, cos_anneal_ratio, params, plan=None, eng=None):
        """Training path: same outputs, differentiable wrt every network parameter (dpt_runner.py:253).
        plan: a _TrainPlan whose input buffers hold this call's rays (rays_o ... z_out are then the plan's own tensors)."""
        B, dev = rays_o.shape[0], rays_o.device
        if eng is None:
            eng = self._engine_for(B, dev, params)
        if plan is not None:
            eng._plan_call = plan
        bgc = None
        if background_rgb is not None:
            bgc = background_rgb.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous()
            if bgc.numel() != 3:
                raise ValueError("background_rgb must have 3 values")
        z = z.contiguous()
        color, feats, weights, eik, cdf, normals, s_val, wsum, wmax, inside, z_ret, eik_terms = _RenderCoreFn.apply(
            eng, rays_o, rays_d, z, z_out, bgc, float(cos_anneal_ratio), *params)
        self.last_eikonal_terms = eik_terms
        return {
            "render_feats": feats if self.depth_network is not None else None,
            "color_fine": color,
            "s_val": s_val,
            "cdf_fine": cdf,
            "weight_sum": wsum,
            "weight_max": wmax,
            "gradients": normals,
            "weights": weights,
            "z_vals": z_ret,
            "gradient_error": eik,
            "inside_sphere": inside,
        }

    def extract_fields(self, bound_min, bound_max, resolution):
        return extract_fields(bound_min, bound_max, resolution, lambda pts: -self.sdf_network.sdf(pts))

    def extract_geometry(self, bound_min, bound_max, resolution, threshold=0.0, method=None):
        """renderer.py:441-446. `method` (not a reference argument): "cubes" (default, PyMCubes' mesh) or "tets"."""
        return extract_geometry(bound_min, bound_max, resolution=resolution, threshold=threshold,
                                query_func=lambda pts: -self.sdf_network.sdf(pts), method=method)


class RenderPlan:
    """render() of one fixed batch size captured once as a HIP graph and replayed per batch.

    The inference render() is ~20 launches of 5-130 us plus ~25 buffer allocations; issued one by one from Python the host
    needs ~350 us per 512-ray batch against ~400 us of device time (measured, tools/dev/fwd_probe.py): the device is still the
    bound, but barely. The plan runs render() once under stream capture on fixed input buffers - every buffer render()
    allocates lands in the graph's private pool - and each call copies the rays in, refreshes the weight images if a
    parameter changed (the launches read them through fixed pointers) and replays: ~75 us of host time per batch. The replay
    itself is ~3% slower on the device than the eager launches (gaps between graph nodes), so the image loops
    (vdn_train/validate.py) use it only on request (VDN_RENDER_GRAPH=1): it is for hosts that are busy with something else.

    The returned dict is the one render() returned at capture: its tensors are overwritten by the next call. Keyword
    arguments (background_rgb, cos_anneal_ratio, perturb_overwrite, depth_before_color) are fixed at construction; the
    jitter, when perturb is on, is drawn inside the graph from torch's generator and differs per replay.
    """

    def __init__(self, renderer, batch, **render_kwargs):
        if torch.is_grad_enabled() and any(p.requires_grad for p in renderer._all_parameters()):
            raise RuntimeError("RenderPlan is the inference path: build and call it under torch.no_grad()")
        dev = next(renderer.sdf_network.parameters()).device
        self.renderer, self.batch, self.kw = renderer, int(batch), dict(render_kwargs)
        bg = self.kw.get("background_rgb")
        if bg is not None:
            self.kw["background_rgb"] = bg.detach().to(device=dev, dtype=torch.float32).reshape(-1).contiguous().clone()
        self.o, self.d = torch.zeros(batch, 3, device=dev), torch.zeros(batch, 3, device=dev)
        self.d[:, 2] = 1.0
        self.near, self.far = torch.zeros(batch, 1, device=dev), torch.ones(batch, 1, device=dev)
        side = torch.cuda.Stream(device=dev)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                         # warm-up outside capture: weight images, lazy workspaces
            renderer.render(self.o, self.d, self.near, self.far, **self.kw)
        torch.cuda.current_stream().wait_stream(side)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self.out = renderer.render(self.o, self.d, self.near, self.far, **self.kw)

    def _nets(self):
        r = self.renderer
        return [m for m in (r.nerf, r.sdf_network, r.color_network, r.depth_network) if m is not None]

    def __call__(self, rays_o, rays_d, near, far):
        if rays_o.shape[0] != self.batch:
            raise ValueError("this plan renders batches of %d rays, got %d" % (self.batch, rays_o.shape[0]))
        for m in self._nets():
            m._images()                                       # re-materialise the weight images if a parameter changed
        self.o.copy_(rays_o.detach())
        self.d.copy_(rays_d.detach())
        self.near.copy_(near.detach().reshape(-1, 1))
        self.far.copy_(far.detach().reshape(-1, 1))
        self.graph.replay()
        return self.out
