"""Training loop semantics of the reference's Runner.train (dpt_runner.py:173-299, 304-323, 350-381)
on the MI355X engine: loss, Adam, warm-up + cosine learning rate, cos-anneal, the VDN depth-loss
ramp, checkpoints in the reference's key schema, and ray-sharded data parallelism.

The hot loop does not go through autograd: TrainEngine.forward -> fused loss kernel ->
TrainEngine.backward -> (all-reduce of the gradient slices) -> fused Adam on flat buffers.

Schedule of one step (overlap=True, the default): the next step begins with the sampler - four dependent SDF-only passes
that need only the SDF weights - so the step ends in three pieces:
  main stream   ... SDF backward -> SDF weight-gradient GEMM -> [all-reduce SDF slice] -> Adam(SDF, variance) -> SDF images
                -> next step's sampler ...
  side stream   background backward -> background weight-gradient GEMM -> [all-reduce] -> Adam -> images   (beside the SDF
                backward) -> (SDF update done) -> colour / VDN heads' weight-gradient GEMM -> [all-reduce] -> Adam -> images
                (beside the next step's sampler; joined before the next forward)
The heads' HBM-bound GEMM and the collectives run beside the latency-bound sampler; the small launches of the SDF update
have no GEMM beside them. Results are identical to the in-order
schedule (overlap=False) bit for bit: the same launches on the same data, only on two streams.
"""
import ctypes
import math
import os
import weakref

import numpy as np
import torch

from vdn_hip import images, lib
from vdn_hip.train import TrainEngine
from vdn_train import dp

DEFAULT_TRAIN_CONF = dict(learning_rate=5e-4, learning_rate_alpha=0.05, end_iter=300000, warm_up_end=5000, anneal_end=50000,
                          igr_weight=0.1, mask_weight=0.0, use_white_bkgd=True, extract_depth=False, depth_start_iter=5000)


_stream = lib.stream_handle          # the HIP handle of torch's current stream


class Trainer:
    def __init__(self, renderer, batch_size, device, conf=None, world_size=1, rank=0, collectives=None, overlap=None):
        """collectives: None = on when world_size > 1; True runs the collective calls also in a one-rank group (the RCCL path
        on a single GPU). overlap: None = on (VDN_OVERLAP=0 turns it off) - see the module docstring; after train_step the
        colour / VDN / background parameters and gradients are then complete on the side stream: join() before touching them
        (state_dict, load_checkpoint, TrainEngine.param_grads and the networks' own weight-image accessor do: render(),
        validate_image ... right after train_step are ordered behind the update without an explicit join())."""
        self.r, self.B, self.dev = renderer, batch_size, torch.device(device)
        self.conf = dict(DEFAULT_TRAIN_CONF)
        self.conf.update(conf or {})
        self.world, self.rank = world_size, rank
        self.iter_step, self.depth_iter = 0, 0
        self._img_cache = {}
        # flatten: every Parameter becomes a view of one buffer (names / state_dict unchanged), so Adam is one
        # launch and the gradient all-reduce one message
        self._params = renderer._all_parameters()
        total = sum(p.numel() for p in self._params)
        self._param_flat = torch.empty(total, dtype=torch.float32, device=self.dev)
        off = 0
        with torch.no_grad():
            for p in self._params:
                n = p.numel()
                self._param_flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self._param_flat[off:off + n].view(p.shape)
                off += n
        self._exp_avg = torch.zeros_like(self._param_flat)
        self._exp_avg_sq = torch.zeros_like(self._param_flat)
        # torch.optim.Adam keeps a step count per parameter and skips parameters whose .grad is None. In the reference that
        # is the VDN head and the background network's dpt_linear until the depth-feature loss first enters the loss
        # (dpt_runner.py:239-243): their moments and bias correction start THEN. Two groups of flat-buffer ranges:
        self._depth_idx = set()
        ranges, off = [], 0
        crit_ids = set(id(p) for m in (renderer.sdf_network, renderer.deviation_network) for p in m.parameters())
        dn = renderer.depth_network
        # render(depth_before_color=True) feeds the VDN head's output to the colour network (renderer.py:247-248): the head then
        # has a colour-loss gradient from iteration 0 and steps with everything else; only dpt_linear waits for the depth loss
        dbc = dn is not None and renderer.color_network.conf["d_feature"] == 352
        depth_ids = set(id(p) for p in dn.parameters()) if (dn is not None and not dbc) else set()
        dpt = getattr(renderer.nerf, "dpt_linear", None) if renderer.nerf is not None else None
        if dpt is not None:
            depth_ids |= set(id(p) for p in dpt.parameters())
        for i, p in enumerate(self._params):
            kind = "depth" if id(p) in depth_ids else ("sdf" if id(p) in crit_ids else "rest")
            if kind == "depth":
                self._depth_idx.add(i)
            if ranges and ranges[-1][0] == kind:
                ranges[-1][2] = off + p.numel()
            else:
                ranges.append([kind, off, off + p.numel()])
            off += p.numel()
        # step groups of the fused Adam (two element ranges per launch): the SDF network + variance (critical path), the other
        # parameters that step from iteration 0, and the late group
        self._sdf_ranges = [(b, e) for k, b, e in ranges if k == "sdf"]
        self._rest_ranges = [(b, e) for k, b, e in ranges if k == "rest"]
        self._depth_ranges = [(b, e) for k, b, e in ranges if k == "depth"]
        if len(self._sdf_ranges) != 1 or len(self._rest_ranges) > 2 or len(self._depth_ranges) > 2:
            raise ValueError("unexpected parameter order: the fused Adam handles two ranges per step group")
        # all-reduce slices of the flat gradient: the SDF slice, and whatever lies before / behind it (dpt_runner.py:121-130 order)
        (sb, se), total = self._sdf_ranges[0], off
        self._slices_rest = [(b, e) for b, e in ((0, sb), (se, total)) if e > b]
        self._depth_adam_steps = 0          # optimizer steps the depth group has taken
        self.coll = dp.Collectives(world_size, force=bool(collectives))
        if self.coll.enabled:
            self.coll.broadcast(self._param_flat, 0)     # replicas must start from rank 0's parameters (e.g. per-process random init)
        self.engine = TrainEngine(renderer, batch_size, self.dev)
        self.overlap = (os.environ.get("VDN_OVERLAP", "1") != "0") if overlap is None else bool(overlap)
        self._ev_gemm, self._ev_rest, self._ev_tail = (torch.cuda.Event() for _ in range(3))
        self._ev_comp, self._ev_log, self._log_stream = torch.cuda.Event(), torch.cuda.Event(), None
        self._ev_ws, self._ws_pending = torch.cuda.Event(), False      # the side stream's last reader of the forward's workspaces
        self._rest_pending, self._rest_gen, self._joined = False, 0, {}
        self._jitter, self._jitter_next = None, 0
        self.engine.join_hook = self.join
        # the networks this Trainer updates on its side stream wait for that update whenever their weight images are used
        # (dpt_models/fields.py::_HipNet._images): rendering with the renderer right after train_step needs no explicit join().
        # The SDF network and the variance are updated on the caller's stream and need none (the next step's sampler runs on
        # them beside the side-stream half).
        for m in (renderer.nerf, renderer.color_network, renderer.depth_network):
            if m is not None:
                m.__dict__["_stream_join"] = weakref.WeakMethod(self.join)
        # the sampler's first SDF pass of a step starts on caches full of the previous step's planes: it warms its weight stream
        # (VdnSdfArgs.cold_start; the kernels that save activations do so on their own)
        renderer.sdf_network.__dict__["_cold_start"] = True
        self._eik_global = torch.zeros(3, dtype=torch.float32, device=self.dev)
        self._eik_partial = torch.zeros(batch_size, 2, dtype=torch.float32, device=self.dev)
        self._eik_handles = []
        B, T = batch_size, self.engine.T
        f = lambda *s: torch.empty(*s, dtype=torch.float32, device=self.dev)
        self.g_color, self.g_weights, self.g_eik, self.scalars = f(B, 3), f(B, T), f(1), f(6)
        self.g_feats = f(B, 96) if self.engine.wdepth else None
        self.bg = torch.ones(3, device=self.dev) if self.conf["use_white_bkgd"] else None

    # ---- schedules (dpt_runner.py:304-319, 167-171)
    def learning_rate(self):
        c = self.conf
        if self.iter_step < c["warm_up_end"]:
            factor = self.iter_step / c["warm_up_end"]
        else:
            progress = (self.iter_step - c["warm_up_end"]) / (c["end_iter"] - c["warm_up_end"])
            factor = (math.cos(math.pi * progress) + 1.0) * 0.5 * (1 - c["learning_rate_alpha"]) + c["learning_rate_alpha"]
        return c["learning_rate"] * factor

    def cos_anneal_ratio(self):
        return 1.0 if self.conf["anneal_end"] == 0 else min(1.0, self.iter_step / self.conf["anneal_end"])

    def depth_iter_weight(self, total_iter=5000):
        return 1.0 / (math.exp(-10 * (self.depth_iter / total_iter - 0.5)) + 1.0)

    # ---- one iteration of dpt_runner.py:197-259
    def train_step(self, rays_o, rays_d, near, far, true_rgb, gt_feats=None, mask=None, t_rand=None, t_rand_out=None,
                   z_vals_inject=None):
        """One iteration (dpt_runner.py:197-259) -> device tensor [loss, color_loss, psnr, eikonal, depth_loss, mask_loss]."""
        r, eng, st = self.r, self.engine, _stream()
        B = self.B
        # the kernels take raw pointers: packed float32 rows on this device, exactly B of them. The reference's flow slices
        # one [B, 10+C] row (dpt_runner.py:201): such column views are strided and are packed here.
        def packed(t, shape, what):
            if t is None:
                return None
            if not (torch.is_tensor(t) and t.device == self.dev and t.dtype == torch.float32):
                raise ValueError("train_step: %s must be a float32 tensor on %s" % (what, self.dev))
            if t.numel() != int(np.prod(shape)) or t.shape[0] != shape[0]:
                raise ValueError("train_step: %s has shape %s, expected %s (batch size is fixed at construction)" % (what, tuple(t.shape), shape))
            return t.reshape(shape).contiguous()
        rays_o, rays_d = packed(rays_o, (B, 3), "rays_o"), packed(rays_d, (B, 3), "rays_d")
        near, far = packed(near, (B, 1), "near"), packed(far, (B, 1), "far")
        true_rgb = packed(true_rgb, (B, 3), "true_rgb")
        gt_feats, mask = packed(gt_feats, (B, 96), "gt_feats"), packed(mask, (B, 1), "mask")
        if r.perturb > 0 and t_rand is None and t_rand_out is None and z_vals_inject is None and r.n_outside > 0:
            # the two uniform draws of renderer.py:348,355: one generator call covers the next 32 steps (the 5-us launch of a
            # per-step draw sits on the critical path in front of the sampler)
            if self._jitter is None or self._jitter_next >= self._jitter.shape[0]:
                self._jitter, self._jitter_next = torch.rand(32, B * (1 + r.n_outside), device=self.dev), 0
            u = self._jitter[self._jitter_next]
            self._jitter_next += 1
            t_rand, t_rand_out = u[:B].view(B, 1), u[B:].view(B, r.n_outside)
        with torch.no_grad():
            # (the sampler only reads the SDF weight images: it runs beside the previous step's side-stream half)
            z, z_out = r._sample(rays_o, rays_d, near.reshape(B), far.reshape(B), r.perturb, t_rand, t_rand_out, z_vals_inject,
                                 defer_last_merge=True)
        # The side-stream half of the last step is joined in two places. (1) Here, in front of the step preparation, which rewrites
        # the work lists' device-side row counts, and of the forward kernels, which rewrite the saved planes: the side stream's LAST
        # READER of those workspaces - the heads' weight-gradient GEMM (it reads the feature plane, the heads' saves and deltas and
        # fg_active[1]; the background network's backward and GEMM precede it on that stream) - has an event right behind its launch
        # (write-after-read: ADVICE round 4; the first version of the late join left this ordering to timing). (2) In front of
        # the heads' forward launches, the first ones on this stream that read what the side stream UPDATES (parameters, weight
        # images): eng.forward(before_heads=...) - the rest of that stream's work (finalize, all-reduce, Adam, image build) stays
        # off this stream's critical path until then. Without a side stream everything runs on this stream.
        late_join = eng._side is not None
        if not late_join:
            self.join()
        elif self._ws_pending:
            lib.current_stream().wait_event(self._ev_ws)       # (same-box A/B against the unordered arm: + 5 us per step)
            self._ws_pending = False
        depth_on = self.conf["extract_depth"] and self.iter_step > self.conf["depth_start_iter"] and gt_feats is not None
        # Plain configuration (one rank, no mask, no mask loss, no VDN head): compositor, colour-term gradient and the compositor's
        # adjoint are ONE launch (vdn_composite_train) - nothing global sits between them but the eikonal denominator, which is
        # the foreground list's length - and the loss SCALARS (logging only) are reduced on a stream of their own, off the
        # critical path: 4 launches of 5 - 15 us each (compositor, eikonal reduce, loss, adjoint) become one. Same device
        # functions, same expressions: gradients bit-identical (tests/test_gpu_train_parity.py). VDN_FUSED_COMPOSITE=0: off.
        # With more than one rank the denominator is the sum of the ranks' list lengths: an all-reduce of ONE int32, started right behind
        # the step preparation and long done when the compositor needs it; the global numerator is only needed for the scalars, on the
        # logging stream. (VDN_DP_FUSED=0: compositor, early eikonal all-reduce, loss kernel and adjoint as separate launches.)
        dp = self.coll.enabled
        fusable = (mask is None and self.conf["mask_weight"] == 0.0 and os.environ.get("VDN_FUSED_COMPOSITE", "1") != "0"
                   and (not dp or (os.environ.get("VDN_DP_FUSED", "1") != "0" and r.n_outside > 0
                                   and os.environ.get("VDN_FG_COMPACT", "1") != "0")))
        fuse = fusable and not eng.wdepth
        fl = dict(true_rgb=true_rgb, g_color=self.g_color, igr_weight=self.conf["igr_weight"], grad_scale=1.0 / self.world) if fuse else None
        # ... and with the VDN head in the loss (womsk_white_wdepth): the 96 feature channels keep their streaming launches, but the
        # loss gradients are made inside the compositor's launches (vdn_composite_fwd_train / vdn_composite_bwd_train) and the
        # eikonal reduce + loss kernel leave the critical path the same way
        depth_w = self.depth_iter_weight() if depth_on else 0.0
        fuse_wd = fusable and eng.wdepth and depth_on
        if fuse_wd:
            fl = dict(true_rgb=true_rgb, g_color=self.g_color, igr_weight=self.conf["igr_weight"], grad_scale=1.0 / self.world,
                      gt_feats=gt_feats, g_feats=self.g_feats, depth_weight=depth_w)
        if dp and fl is not None:
            if self.__dict__.get("_fg_cnt_g") is None:
                self._fg_cnt_g = torch.zeros(1, dtype=torch.int32, device=self.dev)
            fl.update(fg_count=self._fg_cnt_g, after_prep=self._count_begin,
                      before_composite=lambda: self.coll.finish(self._cnt_handles, tag="fg_count"))
        after_sdf = self._eikonal_begin if (dp and fl is None) else None
        w = eng.forward(rays_o, rays_d, z.contiguous(), z_out, self.bg, self.cos_anneal_ratio(), skip_far=True,
                        pending_merge=r._pending_merge, after_sdf=after_sdf, fuse_loss=fl, before_heads=self.join if late_join else None)
        fused_wd = getattr(eng, "_bwd_train", None) is not None
        fused = bool(getattr(eng, "_composite_bwd_done", False)) or fused_wd
        if dp and fl is not None and not fused:
            raise RuntimeError("the engine did not take the fused compositor path the data-parallel step was set up for")
        if dp and not fused:
            # the eikonal term is a ratio of sums over the GLOBAL batch (renderer.py:313-315): its two sums were
            # on their way since the SDF kernel finished; the loss kernel and the compositor's adjoint read w["eik"]
            self.coll.finish(self._eik_handles, tag="eikonal")
            # (ratio, num, den) from the reduced pair in one launch: vdn_eikonal_reduce over ONE "ray" whose partial sums are the
            # global ones - the same f32 expression as three tiny torch launches on the critical path
            eg = self._eik_global
            lib.call("vdn_eikonal_reduce", ctypes.c_void_p(eg.data_ptr() + 4), 1, lib.ptr(w["eik"]), st)
        a = lib.VdnLossArgs()
        a.color, a.true_rgb, a.weights, a.eik = w["color"].data_ptr(), true_rgb.data_ptr(), w["weights"].data_ptr(), w["eik"].data_ptr()
        a.mask = mask.data_ptr() if mask is not None else None
        a.igr_weight, a.mask_weight = self.conf["igr_weight"], self.conf["mask_weight"]
        a.grad_scale = 1.0 / self.world
        a.B, a.T, a.C = B, eng.T, 96
        a.g_color, a.g_eik, a.out_scalars = self.g_color.data_ptr(), self.g_eik.data_ptr(), self.scalars.data_ptr()
        if self.conf["mask_weight"] != 0.0:
            a.g_weights = self.g_weights.data_ptr()
        if depth_on:
            a.feats, a.gt_feats, a.g_feats = w["feat_out"].data_ptr(), gt_feats.data_ptr(), self.g_feats.data_ptr()
            a.depth_weight = depth_w
            self.depth_iter += 1
        if fused:
            if self._log_stream is None:
                from vdn_hip.train import shared_stream
                self._log_stream = shared_stream(self.dev, "log")
            ls = self._log_stream
            self._ev_comp.record(lib.current_stream())
            ls.wait_event(self._ev_comp)
            if fused_wd:
                # (the kernel rewrites the gradients with the values the compositor's launches made: into copies, since the
                # adjoint's feature launches read g_feats on the main stream meanwhile)
                if self.__dict__.get("_g_log") is None:
                    self._g_log = (torch.empty_like(self.g_color), torch.empty_like(self.g_feats))
                a.g_color, a.g_feats = self._g_log[0].data_ptr(), self._g_log[1].data_ptr()
            lib.call("vdn_eikonal_reduce", lib.ptr(w["eik_partial"]), B, lib.ptr(w["eik"]), ls.cuda_stream)
            if dp:
                # the reported eikonal term is the GLOBAL ratio: this rank's (num, den) summed over the ranks in place, then the
                # ratio of the pair by the same kernel over one "ray" (it reads the two sums before it writes the triple)
                with torch.cuda.stream(ls):
                    self.coll.finish(self.coll.begin([w["eik"][1:3]], side=True), tag="eikonal")
                lib.call("vdn_eikonal_reduce", ctypes.c_void_p(w["eik"].data_ptr() + 4), 1, lib.ptr(w["eik"]), ls.cuda_stream)
            lib.call("vdn_loss_fwd_bwd", a, ls.cuda_stream)      # the scalars (it rewrites g_color / g_eik with the values already used)
            self._ev_log.record(ls)
        else:
            lib.call("vdn_loss_fwd_bwd", a, st)
        g_feats = self.g_feats if depth_on else None
        g_weights = self.g_weights if self.conf["mask_weight"] != 0.0 else None
        lr, main_step = self.learning_rate(), self.iter_step + 1 - self._step0()

        def adam(ranges, step, stream):
            (b0, e0), (b1, e1) = ranges[0], (ranges[1] if len(ranges) > 1 else (0, 0))
            lib.call("vdn_adam_step_ranges", lib.ptr(self._param_flat), lib.ptr(grad), lib.ptr(self._exp_avg), lib.ptr(self._exp_avg_sq),
                     b0, e0, b1, e1, lr, 0.9, 0.999, 1e-8, step, stream)

        rest_nets = [k for k in eng.nets if k != "sdf"]
        if depth_on and self._depth_ranges:
            # only when the depth loss is in this step's loss: torch.optim.Adam skips parameters whose .grad is None (the
            # runner's zero_grad() resets them every iteration), it does not step them on zero gradients
            self._depth_adam_steps += 1
        depth_step = self._depth_adam_steps if (depth_on and self._depth_ranges) else 0

        sdf_begin = self._sdf_ranges[0][0]

        def update_rest(stream, part="rest"):
            """all-reduce, Adam and weight images of the parameters outside the SDF group: all of them ("rest"), or only those
            in front of the SDF network in the flat buffer ("nerf": the background network) / behind it ("heads")."""
            keep = {"rest": lambda b: True, "nerf": lambda b: b < sdf_begin, "heads": lambda b: b >= sdf_begin}[part]
            nets = [k for k in rest_nets if part == "rest" or (k == "nerf") == (part == "nerf")]
            on_side = stream != st            # (without a side stream these slices share the caller's stream and the main group)
            self.coll.sum_now([grad[b:e] for b, e in self._slices_rest if keep(b)], side=on_side, tag="grad_" + part)
            rr = [r_ for r_ in self._rest_ranges if keep(r_[0])]
            if rr:
                adam(rr, main_step, stream)
            dr = [r_ for r_ in self._depth_ranges if keep(r_[0])] if depth_step else []
            if dr:
                adam(dr, depth_step, stream)
            if nets:
                images.refresh_together([eng.nets[k].img for k in nets], stream, self._img_cache.setdefault(part, {}))

        def update_sdf(stream):
            self.coll.sum_now([grad[b:e] for b, e in self._sdf_ranges], tag="grad_sdf")
            adam(self._sdf_ranges, main_step, stream)
            images.refresh_together([eng.nets["sdf"].img], stream, self._img_cache.setdefault("sdf", {}))

        grad = eng._grad_flat
        split = self.overlap and os.environ.get("VDN_SPLIT_REST", "1") != "0" and "nerf" in eng.dw_groups and "heads" in eng.dw_groups
        # (events on the critical chain are marker packets, 3 - 4 us each: the side stream's fork reuses _ev_comp - nothing was
        # launched on this stream since; not with the VDN head, where the compositor's adjoint is still to come in backward() - the SDF GEMM's event is only recorded for the schedule that waits for it, and the heads'
        # event is covered by the `after` events below, which are recorded later on this stream)
        trim = os.environ.get("VDN_EVENT_TRIM", "1") != "0"           # (0: the A/B arm that records them all)
        eng.backward(self.g_color, g_feats, g_weights, self.g_eik, defer_rest=True,
                     gemm_event=self._ev_gemm if (self.overlap and not (split and trim)) else None,
                     fork_event=self._ev_comp if (fused and not fused_wd and trim) else None, heads_event=not trim)
        side = None
        if split:
            # the background network's half right behind its backward, beside the SDF backward on the main stream ...
            side = eng.side_weight_grads("nerf")
            if side is not None:
                with torch.cuda.stream(side):
                    update_rest(side.cuda_stream, "nerf")
        update_sdf(st)
        if split and side is not None:
            # ... and the heads' half behind the SDF group's update: their GEMM runs beside the next step's sampler (which leaves
            # most of the chip idle) instead of beside the SDF update's small launches, which it starved: 200 us for launches that
            # take 55 us alone. Measured (same box, 3 alternating runs each): 1.410 ms / step against 1.427 with one GEMM for both
            # halves behind the SDF GEMM; deferring the background half too (its next forward has slack) costs 1.48 - its GEMM
            # then starves the sampler's 256-workgroup SDF passes (150 us for an 18-us pass)
            self._ev_tail.record(lib.current_stream())
            eng.side_weight_grads("heads", after=self._ev_tail, gemm_event=self._ev_ws)
            self._ws_pending = True
            with torch.cuda.stream(side):
                update_rest(side.cuda_stream, "heads")
                self._ev_rest.record(side)
                self._rest_gen += 1
            self._rest_pending = True
        else:
            if split:                            # no side stream: both halves on the caller's stream
                update_rest(st, "nerf")
                eng.weight_grads("heads", st)
                update_rest(st, "heads")
            else:
                side = eng.rest_weight_grads(after=self._ev_gemm, gemm_event=self._ev_ws) if self.overlap else None
                self._ws_pending = side is not None
                if side is None:
                    if not self.overlap:
                        eng._join()                     # the background network's backward (side stream) feeds the rest group
                        eng.weight_grads("rest", st)
                    update_rest(st)
                else:
                    with torch.cuda.stream(side):
                        update_rest(side.cuda_stream)
                        self._ev_rest.record(side)
                        self._rest_gen += 1
                    self._rest_pending = True
        if fused:
            lib.current_stream().wait_event(self._ev_log)      # the scalars (long done) are ordered in front of what follows
        self.iter_step += 1
        return self.scalars        # device tensor [loss, color_loss, psnr, eikonal, depth_loss, mask_loss]; no host sync here

    # the flat buffers / the parameter list as seen from outside: complete on torch's current stream
    params = property(lambda self: (self.join(), self._params)[1])
    param_flat = property(lambda self: (self.join(), self._param_flat)[1])
    exp_avg = property(lambda self: (self.join(), self._exp_avg)[1])
    exp_avg_sq = property(lambda self: (self.join(), self._exp_avg_sq)[1])

    def join(self):
        """Order torch's current stream behind the side-stream half of the last step (colour / VDN / background gradients,
        their Adam step and weight images). train_step calls it before its forward; call it before reading those parameters
        or rendering with the renderer outside the Trainer."""
        # `_rest_pending` = a side-stream half has ever been issued. One wait per stream and per recording of the event: a
        # wait the host cannot see to be complete is a barrier packet in the stream's queue (3 - 7 us each on the step's critical
        # path, and the forward asks once per network: tools/dev/gap_probe.py), while a single flag cleared by a wait on one
        # stream would leave a step issued on another stream unordered.
        if self._rest_pending:
            cur = lib.current_stream()
            if self._joined.get(cur.cuda_stream) != self._rest_gen or os.environ.get("VDN_EVENT_TRIM", "1") == "0":
                cur.wait_event(self._ev_rest)
                self._joined[cur.cuda_stream] = self._rest_gen

    def _count_begin(self, eng):
        """Right behind the step preparation: the length of this rank's foreground work list - the eikonal term's denominator
        (the compositor's own norm test made the list) - on its way to the sum over the ranks, under the SDF kernel."""
        self._fg_cnt_g.copy_(eng.w["fg_active"][1])
        self._cnt_handles = self.coll.begin([self._fg_cnt_g])

    def _eikonal_begin(self, eng):
        """Right behind the fused SDF kernel: this rank's eikonal sums (vdn_eikonal_terms: the compositor's own expressions) and
        the start of their all-reduce, which then runs under the colour head, the background network and the compositor."""
        rays_o, rays_d = eng._fwd_rays
        a = lib.VdnEikonalArgs()
        a.rays_o, a.rays_d, a.mid_z, a.normals = rays_o.data_ptr(), rays_d.data_ptr(), eng.w["mid_z"].data_ptr(), eng.w["normals"].data_ptr()
        a.B, a.N = eng.B, eng.N
        a.eik_partial, a.eik_out = self._eik_partial.data_ptr(), self._eik_global.data_ptr()
        lib.call("vdn_eikonal_terms", a, _stream())
        self._eik_handles = self.coll.begin([self._eik_global[1:3]])

    def _step0(self):
        return getattr(self, "_adam_step_offset", 0)

    # ---- checkpoints in the reference's schema (dpt_runner.py:366-381, 350-359)
    def state_dict(self):
        self.join()
        r = self.r
        state, off = {}, 0
        for i, p in enumerate(self._params):
            n = p.numel()
            steps = self._depth_adam_steps if i in self._depth_idx else self.iter_step - self._step0()
            if steps > 0:          # torch.optim.Adam has no state for a parameter that never had a gradient
                state[i] = {"step": torch.tensor(float(steps)),
                            "exp_avg": self._exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self._exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        opt = {"state": state,
               "param_groups": [{"lr": self.learning_rate(), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                                 "params": list(range(len(self._params)))}]}
        cl = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
        return {"nerf": cl(r.nerf), "sdf_network_fine": cl(r.sdf_network), "variance_network_fine": cl(r.deviation_network),
                "color_network_fine": cl(r.color_network),
                "depth_network_fine": cl(r.depth_network) if r.depth_network is not None else None,
                "optimizer": opt, "iter_step": self.iter_step}

    def save_checkpoint(self, path):
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path_or_dict):
        ck = torch.load(path_or_dict, map_location=self.dev) if isinstance(path_or_dict, str) else path_or_dict
        self.join()
        r = self.r
        with torch.no_grad():
            def load(mod, sd, strict=True):
                own = dict(mod.named_parameters())
                for k, v in sd.items():
                    if k in own:
                        own[k].copy_(v.to(self.dev))           # in place: parameters stay views of the flat buffer
                    elif strict:
                        raise KeyError(k)
            load(r.nerf, ck["nerf"], strict=False)                                   # dpt_runner.py:352
            load(r.sdf_network, ck["sdf_network_fine"])
            load(r.deviation_network, ck["variance_network_fine"])
            load(r.color_network, ck["color_network_fine"])
            self.iter_step = int(ck["iter_step"])
            if (r.depth_network is not None and ck.get("depth_network_fine") is not None
                    and self.iter_step > self.conf["depth_start_iter"]):             # dpt_runner.py:358-359
                load(r.depth_network, ck["depth_network_fine"])
            st = ck["optimizer"]["state"]
            off = 0
            steps = {False: 0, True: 0}           # per step group: (main, depth)
            self._exp_avg.zero_()
            self._exp_avg_sq.zero_()
            for i, p in enumerate(self._params):
                n = p.numel()
                if i in st:
                    self._exp_avg[off:off + n].copy_(st[i]["exp_avg"].reshape(-1).to(self.dev))
                    self._exp_avg_sq[off:off + n].copy_(st[i]["exp_avg_sq"].reshape(-1).to(self.dev))
                    g = i in self._depth_idx
                    steps[g] = max(steps[g], int(st[i]["step"]))
                off += n
            self._adam_step_offset = self.iter_step - steps[False]
            self._depth_adam_steps = steps[True]
