"""Training loop semantics of the reference's Runner.train (dpt_runner.py:173-299, 304-323, 350-381)
on the MI355X engine: loss, Adam, warm-up + cosine learning rate, cos-anneal, the VDN depth-loss
ramp, checkpoints in the reference's key schema, and ray-sharded data parallelism.

The hot loop does not go through autograd: TrainEngine.forward -> fused loss kernel ->
TrainEngine.backward -> (one all-reduce of the flat gradient) -> fused Adam on flat buffers.
"""
import math

import numpy as np
import torch

from vdn_hip import images, lib
from vdn_hip.train import TrainEngine
from vdn_train import dp

DEFAULT_TRAIN_CONF = dict(learning_rate=5e-4, learning_rate_alpha=0.05, end_iter=300000, warm_up_end=5000, anneal_end=50000,
                          igr_weight=0.1, mask_weight=0.0, use_white_bkgd=True, extract_depth=False, depth_start_iter=5000)


def _stream():
    return torch.cuda.current_stream().cuda_stream


class Trainer:
    def __init__(self, renderer, batch_size, device, conf=None, world_size=1, rank=0):
        self.r, self.B, self.dev = renderer, batch_size, torch.device(device)
        self.conf = dict(DEFAULT_TRAIN_CONF)
        self.conf.update(conf or {})
        self.world, self.rank = world_size, rank
        self.iter_step, self.depth_iter = 0, 0
        self._img_cache = {}
        # flatten: every Parameter becomes a view of one buffer (names / state_dict unchanged), so Adam is one
        # launch and the gradient all-reduce one message
        self.params = renderer._all_parameters()
        total = sum(p.numel() for p in self.params)
        self.param_flat = torch.empty(total, dtype=torch.float32, device=self.dev)
        off = 0
        with torch.no_grad():
            for p in self.params:
                n = p.numel()
                self.param_flat[off:off + n].copy_(p.detach().reshape(-1))
                p.data = self.param_flat[off:off + n].view(p.shape)
                off += n
        self.exp_avg = torch.zeros_like(self.param_flat)
        self.exp_avg_sq = torch.zeros_like(self.param_flat)
        # torch.optim.Adam keeps a step count per parameter and skips parameters whose .grad is None. In the reference that
        # is the VDN head and the background network's dpt_linear until the depth-feature loss first enters the loss
        # (dpt_runner.py:239-243): their moments and bias correction start THEN. Two groups of flat-buffer ranges:
        self._depth_idx = set()
        ranges, off = [], 0
        dn = renderer.depth_network
        # render(depth_before_color=True) feeds the VDN head's output to the colour network (renderer.py:247-248): the head then
        # has a colour-loss gradient from iteration 0 and steps with everything else; only dpt_linear waits for the depth loss
        dbc = dn is not None and renderer.color_network.conf["d_feature"] == 352
        depth_ids = set(id(p) for p in dn.parameters()) if (dn is not None and not dbc) else set()
        dpt = getattr(renderer.nerf, "dpt_linear", None) if renderer.nerf is not None else None
        if dpt is not None:
            depth_ids |= set(id(p) for p in dpt.parameters())
        for i, p in enumerate(self.params):
            is_d = id(p) in depth_ids
            if is_d:
                self._depth_idx.add(i)
            if ranges and ranges[-1][0] == is_d:
                ranges[-1][2] = off + p.numel()
            else:
                ranges.append([is_d, off, off + p.numel()])
            off += p.numel()
        self._main_ranges = [(b, e) for d, b, e in ranges if not d]
        self._depth_ranges = [(b, e) for d, b, e in ranges if d]
        if len(self._main_ranges) > 2 or len(self._depth_ranges) > 2:
            raise ValueError("unexpected parameter order: the fused Adam handles two ranges per step group")
        self._depth_adam_steps = 0          # optimizer steps the depth group has taken
        if world_size > 1:
            import torch.distributed as dist
            dist.broadcast(self.param_flat, 0)     # replicas must start from rank 0's parameters (e.g. per-process random init)
        self.engine = TrainEngine(renderer, batch_size, self.dev)
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
            # the two uniform draws of renderer.py:348,355 from one generator call (a launch less per step; render() itself
            # keeps the reference's two calls)
            u = torch.rand(B * (1 + r.n_outside), device=self.dev)
            t_rand, t_rand_out = u[:B].view(B, 1), u[B:].view(B, r.n_outside)
        with torch.no_grad():
            z, z_out = r._sample(rays_o, rays_d, near.reshape(B), far.reshape(B), r.perturb, t_rand, t_rand_out, z_vals_inject,
                                 defer_last_merge=True)
        w = eng.forward(rays_o, rays_d, z.contiguous(), z_out, self.bg, self.cos_anneal_ratio(), skip_far=True,
                        pending_merge=r._pending_merge)
        if self.world > 1:
            # the eikonal term is a ratio of sums over the GLOBAL batch (renderer.py:313-315)
            w["eik"][0:1].copy_(dp.global_eikonal(w["eik"][1:3]).reshape(1))
        depth_on = self.conf["extract_depth"] and self.iter_step > self.conf["depth_start_iter"] and gt_feats is not None
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
            a.depth_weight = self.depth_iter_weight()
            self.depth_iter += 1
        lib.call("vdn_loss_fwd_bwd", a, st)
        grad = eng.backward(self.g_color, self.g_feats if depth_on else None,
                            self.g_weights if self.conf["mask_weight"] != 0.0 else None, self.g_eik)
        if self.world > 1:
            dp.allreduce_flat(grad)                # one flat message: all gradients of all networks
        def adam(ranges, step):
            (b0, e0), (b1, e1) = ranges[0], (ranges[1] if len(ranges) > 1 else (0, 0))
            lib.call("vdn_adam_step_ranges", lib.ptr(self.param_flat), lib.ptr(grad), lib.ptr(self.exp_avg), lib.ptr(self.exp_avg_sq),
                     b0, e0, b1, e1, self.learning_rate(), 0.9, 0.999, 1e-8, step, st)
        adam(self._main_ranges, self.iter_step + 1 - self._step0())
        if self._depth_ranges and depth_on:
            # only when the depth loss is in this step's loss: torch.optim.Adam skips parameters whose .grad is None (the
            # runner's zero_grad() resets them every iteration), it does not step them on zero gradients
            self._depth_adam_steps += 1
            adam(self._depth_ranges, self._depth_adam_steps)
        # weights changed behind torch's version counters: rebuild every network's images now, in two launches
        images.refresh_together([net.img for net in eng.nets.values()], st, self._img_cache)
        self.iter_step += 1
        return self.scalars        # device tensor [loss, color_loss, psnr, eikonal, depth_loss, mask_loss]; no host sync here

    def _step0(self):
        return getattr(self, "_adam_step_offset", 0)

    # ---- checkpoints in the reference's schema (dpt_runner.py:366-381, 350-359)
    def state_dict(self):
        r = self.r
        state, off = {}, 0
        for i, p in enumerate(self.params):
            n = p.numel()
            steps = self._depth_adam_steps if i in self._depth_idx else self.iter_step - self._step0()
            if steps > 0:          # torch.optim.Adam has no state for a parameter that never had a gradient
                state[i] = {"step": torch.tensor(float(steps)),
                            "exp_avg": self.exp_avg[off:off + n].view(p.shape).clone(),
                            "exp_avg_sq": self.exp_avg_sq[off:off + n].view(p.shape).clone()}
            off += n
        opt = {"state": state,
               "param_groups": [{"lr": self.learning_rate(), "betas": (0.9, 0.999), "eps": 1e-8, "weight_decay": 0, "amsgrad": False,
                                 "params": list(range(len(self.params)))}]}
        cl = lambda m: {k: v.detach().clone() for k, v in m.state_dict().items()}
        return {"nerf": cl(r.nerf), "sdf_network_fine": cl(r.sdf_network), "variance_network_fine": cl(r.deviation_network),
                "color_network_fine": cl(r.color_network),
                "depth_network_fine": cl(r.depth_network) if r.depth_network is not None else None,
                "optimizer": opt, "iter_step": self.iter_step}

    def save_checkpoint(self, path):
        torch.save(self.state_dict(), path)

    def load_checkpoint(self, path_or_dict):
        ck = torch.load(path_or_dict, map_location=self.dev) if isinstance(path_or_dict, str) else path_or_dict
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
            self.exp_avg.zero_()
            self.exp_avg_sq.zero_()
            for i, p in enumerate(self.params):
                n = p.numel()
                if i in st:
                    self.exp_avg[off:off + n].copy_(st[i]["exp_avg"].reshape(-1).to(self.dev))
                    self.exp_avg_sq[off:off + n].copy_(st[i]["exp_avg_sq"].reshape(-1).to(self.dev))
                    g = i in self._depth_idx
                    steps[g] = max(steps[g], int(st[i]["step"]))
                off += n
            self._adam_step_offset = self.iter_step - steps[False]
            self._depth_adam_steps = steps[True]
