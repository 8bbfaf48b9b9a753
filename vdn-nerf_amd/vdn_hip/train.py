"""Training-step engine: persistent workspaces + descriptor tables that drive the forward-with-saves
and the hand-derived backward of NeuSRenderer.render on the MI355X kernels.

One TrainEngine per (renderer, batch size). forward() is the differentiable part of render() -
render_core_outside + render_core (renderer.py:100-145, 209-330) at given, detached z - with every
activation the adjoint needs kept in HBM; backward() runs
  composite_bwd -> RenderingNetwork bwd (colour, VDN) -> SDF rbar/fbar -> NeRF bwd
  -> one batched weight-gradient GEMM -> finalize/scatter -> weight-norm backward
and returns d loss / d parameter for every parameter, in module.parameters() order.
"""
import os

import numpy as np
import torch

from . import images, layout, lib

PTS_PER_SPLIT = int(os.environ.get("VDN_DW_SPLIT_PTS", "4096"))        # rows one workgroup of the weight-gradient GEMM contracts


def _pts_per_split(net, precision="bf16"):
    """Rows per K split of the weight-gradient GEMM, per launch group (SDF network / the rest): read when an engine is built.
    Measured at the bench's steady-state lists (tools/dev/dw_split_sweep.py, all settings interleaved in one process): the SDF
    group at 6144 rows per split is 231 workgroups - ONE round of the 256 CUs (this kernel runs one 128-KiB-LDS workgroup per
    CU) - and the step is 3.1 % shorter than at 4096 (336 workgroups = 1.3 rounds); 8192 (168 workgroups) gives part of that
    back, 3072 (462) is in between. The rest group is best at 4096 (376 workgroups; 6144: +1.4 %, 3072 = 510 workgroups,
    two full rounds: +17 %)."""
    if precision == "fp32":
        # the fp32 GEMM is MFMA-bound (128 x 128 tiles), not HBM-bound: more, shorter splits balance better (2048 / 2048: 8.27 ms per
        # step against 8.60 at 4096 and 8.77 at 6144 / 4096; 1024: 8.32; tools/dev/step_wall.py ... fp32, same box)
        dflt = "2048"
        return int(os.environ.get("VDN_DW_SPLIT_PTS_SDF" if net == "sdf" else "VDN_DW_SPLIT_PTS_REST", dflt))
    if net == "sdf":
        return int(os.environ.get("VDN_DW_SPLIT_PTS_SDF", os.environ.get("VDN_DW_SPLIT_PTS", "6144")))
    # the rest group runs as two launches on the default schedule: the background network's entries and the heads'
    rest = os.environ.get("VDN_DW_SPLIT_PTS_REST", str(PTS_PER_SPLIT))
    return int(os.environ.get("VDN_DW_SPLIT_PTS_NERF" if net == "nerf" else "VDN_DW_SPLIT_PTS_HEADS", rest))


_stream = lib.stream_handle          # the HIP handle of torch's current stream


_SHARED_STREAMS = {}


def shared_stream(dev, role):
    """One stream per (device, role) and process, shared by every engine / Trainer built in it. HIP maps streams onto a handful of
    hardware queues in creation order; a process that builds many engines (bench.py's legs, a test session) and gives each its own
    side streams ends up with an engine whose two streams share ONE queue - its halves then run one after the other (measured: a
    step of 1.46 ms instead of 1.13 on the sixth Trainer of a process). Engines use their side streams one step at a time and
    order every hand-over with events, so sharing them is safe; two Trainers stepped concurrently from two host threads would
    serialise on them (not a supported use)."""
    key = (torch.device(dev).index if torch.device(dev).index is not None else torch.cuda.current_device(), role)
    st = _SHARED_STREAMS.get(key)
    if st is None:
        st = _SHARED_STREAMS[key] = torch.cuda.Stream(device=dev)
    return st


class _Net:
    """Gradient-side state of one network: d W_eff flat buffer, per-parameter gradient views."""

    def __init__(self, module, dev, grad_views):
        self.module = module
        self.img = module._images()
        self.dweff = torch.zeros_like(self.img.weff)
        self.grads = {}          # id(param) -> view into the engine's flat gradient buffer
        for name, (g, v, b) in self.img.matrices.items():
            for t in (g, v, b):
                if t is not None:
                    self.grads[id(t)] = grad_views[id(t)]

    def dweff_view(self, name):
        v = self.img.matrices[name][1]
        o = self.img.w_off[name]
        return self.dweff[o:o + v.numel()].view(v.shape)

    def dw_target(self, name):
        """Where the finalize pass writes d W for matrix `name`: d W_eff (weight-normed) or the .grad buffer itself."""
        g, v, b = self.img.matrices[name]
        return self.dweff_view(name) if g is not None else self.grads[id(v)]

    def bias_target(self, name):
        b = self.img.matrices[name][2]
        return None if b is None else self.grads[id(b)]


class TrainEngine:
    def __init__(self, renderer, B, dev):
        self.r = renderer
        self.B, self.dev = B, dev
        S, I, O = renderer.n_samples, renderer.n_importance, renderer.n_outside
        self.N = S + I
        self.T = self.N + O
        self.P = B * self.N
        self.Q = B * self.T
        self.wdepth = renderer.depth_network is not None
        # render(depth_before_color=True): the colour network is a d_feature = 352 one, fed the VDN head's output too
        self.dbc = renderer.color_network.conf["d_feature"] == 352
        if self.dbc and not self.wdepth:
            raise ValueError("a d_feature = 352 colour network needs a depth_network (renderer.py:245-248)")
        # The background network runs on a side stream beside the SDF kernels (VDN_SIDE_STREAM=0 puts everything on the caller's
        # stream, e.g. for a rocprof kernel trace: concurrent kernels inflate each other's durations there). It pays since the
        # bf16 SDF kernel runs one 128-point workgroup per CU: a work list of ~50 K rows is 1.5 rounds of workgroups, and the
        # half-empty round's CUs take the background network's workgroups (measured: 1.78 -> 1.65 ms per step).
        use_side = os.environ.get("VDN_SIDE_STREAM", "1") == "1" and torch.device(dev).type == "cuda"
        self._side = shared_stream(dev, "side") if use_side else None
        self._ev_fork = torch.cuda.Event() if use_side else None
        self._ev_join = torch.cuda.Event() if use_side else None
        # a second side stream for the VDN head's forward: it and the colour head read the same inputs and each fills only
        # ~55 % of the workgroup slots (285 tiles on 512), so side by side they take little more than one of them alone
        self._side2 = shared_stream(dev, "side2") if (use_side and renderer.depth_network is not None) else None
        self._ev_fork2 = torch.cuda.Event() if self._side2 is not None else None
        self._ev_join2 = torch.cuda.Event() if self._side2 is not None else None
        self._ev_heads = torch.cuda.Event() if use_side else None
        self._pending = False
        self._fused_keep = None
        self._fg_compact = self._bg_compact = False
        precs = {m.precision for m in (renderer.nerf, renderer.sdf_network, renderer.color_network, renderer.depth_network) if m is not None}
        if len(precs) != 1:
            raise ValueError("all networks of a renderer must share one precision, got %s" % sorted(precs))
        self.precision = precs.pop()
        self.sfx = "_f32" if self.precision == "fp32" else "_bf16"
        sdt = torch.float32 if self.precision == "fp32" else torch.bfloat16     # MLP-internal workspaces
        f = lambda *shape: torch.empty(*shape, dtype=torch.float32, device=dev)
        fs = lambda *shape: torch.empty(*shape, dtype=sdt, device=dev)
        P, Q, N, T = self.P, self.Q, self.N, self.T
        # MLP-internal planes: row-major [P, ld] in fp32, tile-blocked + padded to 32 points in bf16 (vdn_hip/layout.py).
        # `Pp` / `Qp` are the plane row counts; API-visible tensors (sdf, normals, colours ...) keep exact P / Q rows.
        Pp, Qp = layout.rows(P, self.precision), layout.rows(Q, self.precision)
        self.Pp, self.Qp = Pp, Qp
        w = self.w = {}
        # ---- forward saves
        w["dists"] = f(B, N)
        # dense per-point outputs are zero-initialised: points a work list skips keep finite values, and the compositor only
        # ever multiplies those by exact zeros (include/vdn_render.h: vdn_foreground_active / vdn_background_active)
        fz = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
        # what render()'s autograd node hands out (it must copy: these buffers are rewritten by the next forward) lies in ONE
        # allocation, so that copy is one launch instead of ten (outputs_clone)
        slots, n_out = {}, 0
        for name, shape in (("color", (B, 3)), ("weights", (B, T)), ("eik", (3,)), ("feat_out", (B, 96) if self.wdepth else None),
                            ("cdf", (B, N)), ("normals", (P, 3)), ("inside", (B, N)), ("s_val", (B, 1)),
                            ("bg_mid", (B, T) if O > 0 else None), ("mid_z", (B, N)), ("wsum", (B, 1)), ("wmax", (B, 1))):
            if shape is not None:
                slots[name] = (n_out, shape)
                n_out += (int(np.prod(shape)) + 63) // 64 * 64
        self._out_arena, self._out_slots = fz(n_out), slots
        out = lambda name: self._out_arena[slots[name][0]:slots[name][0] + int(np.prod(slots[name][1]))].view(slots[name][1])
        w["sdf"], w["feat"], w["normals"], w["mid_z"] = fz(P), fs(Pp, 256), out("normals"), out("mid_z")
        w["fg_active"] = (torch.zeros(P, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
                          torch.zeros(B, dtype=torch.int32, device=dev))
        w["H"], w["V"], w["PE"] = fs(8, Pp, 256), fs(8, Pp, 256), fs(Pp, 64)
        if self.precision == "fp32":          # the bf16 SDF kernel keeps softplus' on the chip (csrc/k_sdf_fwd2.h)
            w["S"] = fs(8, Pp, 256)
        w["col_out"], w["col_h"], w["col_small"] = fz(P, 3), fs(4, Pp, 256), fs(Pp, 64)
        if self.wdepth:
            w["vdn_out"], w["vdn_h"], w["vdn_small"] = fz(P, 96), fs(4, Pp, 256), fs(Pp, 64)
        if self.dbc:
            w["col_extra"] = fs(Pp, 96)
        if O > 0:
            w["z_feed"], w["bg_dists"], w["bg_mid"] = f(B, T), f(B, T), out("bg_mid")
            # zero-initialised: points the active list skips keep finite values (the compositor multiplies them by zero)
            w["bg_density"], w["bg_rgb"] = torch.zeros(Q, device=dev), torch.zeros(Q, 3, device=dev)
            w["bg_feat"] = torch.zeros(Q, 96, device=dev) if self.wdepth else None
            w["bg_active"] = (torch.zeros(Q, dtype=torch.int32, device=dev), torch.zeros(1, dtype=torch.int32, device=dev),
                              torch.zeros(B, dtype=torch.int32, device=dev))
            w["nf_h"], w["nf_pe"], w["nf_feature"], w["nf_vpe"], w["nf_hv"] = fs(8, Qp, 256), fs(Qp, 96), fs(Qp, 256), fs(Qp, 32), fs(Qp, 128)
        w["weights"], w["alpha"], w["cdf"], w["inside"] = out("weights"), f(B, T), out("cdf"), out("inside")
        w["color"], w["wsum"], w["wmax"], w["s_val"] = out("color"), out("wsum"), out("wmax"), out("s_val")
        w["eik_partial"], w["eik"] = f(B, 2), out("eik")
        w["feat_out"] = out("feat_out") if self.wdepth else None
        # ---- backward intermediates
        w["d_sdf"], w["d_normals"], w["d_color"], w["d_featvec"] = f(P), f(P, 3), f(P, 3), fs(Pp, 256)
        w["d_vdn"] = f(P, 96) if self.wdepth else None
        w["feat_scratch"] = f(B * (2 * T + N)) if self.wdepth else None        # VdnCompositeBwdArgs.feat_scratch
        w["d_var_partial"], w["d_variance"] = f(B), f(1)
        w["col_dout"], w["col_dh"] = fs(Pp, 32), fs(4, Pp, 256)
        if self.wdepth:
            w["vdn_dout"], w["vdn_dh"] = fs(Pp, 96), fs(4, Pp, 256)
        w["UB"], w["EX"], w["AB"] = fs(Pp * 2144), fs(8, Pp, 256), fs(Pp * 2336)
        if O > 0:
            w["d_bg_density"], w["d_bg_rgb"] = f(Q), f(Q, 3)
            w["d_bg_feat"] = f(Q, 96) if self.wdepth else None
            w["nf_do"], w["nf_dv"], w["nf_dhead"], w["nf_dh"] = fs(Qp, 128 if self.wdepth else 32), fs(Qp, 128), fs(Qp, 288), fs(8, Qp, 256)
        # one flat gradient buffer over all parameters, in dpt_runner.py:121-130 order (nerf, sdf, variance,
        # colour, vdn): the single message of the data-parallel all-reduce
        self.params = renderer._all_parameters()
        total = sum(p.numel() for p in self.params)
        self._grad_flat = torch.zeros(total, dtype=torch.float32, device=dev)
        views, off = {}, 0
        for p in self.params:
            views[id(p)] = self._grad_flat[off:off + p.numel()].view(p.shape)
            off += p.numel()
        self.grad_views = views
        self.nets = {"sdf": _Net(renderer.sdf_network, dev, views), "color": _Net(renderer.color_network, dev, views)}
        if self.wdepth:
            self.nets["vdn"] = _Net(renderer.depth_network, dev, views)
        if O > 0:
            self.nets["nerf"] = _Net(renderer.nerf, dev, views)
        self._build_dw_plan()

    # ------------------------------------------------------------------------------------------
    # weight-gradient plan
    # ------------------------------------------------------------------------------------------
    def _build_dw_plan(self):
        w, P, Q = self.w, self.P, self.Q
        ent = []    # dict(net, name, rmap, cmap, scale, A, A2, B, B2, bias(bool), Pn, extra)
        # operand spec = (tensor, element offset of the plane inside the tensor, first column, ld of the plane)
        Pp, Qp, prec = self.Pp, self.Qp, self.precision
        ub_off, off = {}, 0
        for l, cols in enumerate((64, 256, 256, 256, 288, 256, 256, 256, 256)):
            ub_off[l] = (off, cols)
            off += Pp * cols
        UB = lambda l, c0=0: (w["UB"], ub_off[l][0], c0, ub_off[l][1])
        AB = lambda l: (w["AB"], 0, 0, 288) if l == 8 else (w["AB"], Pp * 288 + (7 - l) * Pp * 256, 0, 256)
        sl = lambda t, l, ld=256: (t, l * t.shape[1] * t.shape[2], 0, ld)          # layer plane of a [L, rows, ld] tensor
        whole = lambda t, c0=0: (t, 0, c0, t.shape[1])
        maps = images.sdf_layer_maps()
        # bf16: the SDF forward saves H, V and PE in units of 1/(100 log2 e) (include/vdn_render.h: VdnSdfArgs), so both
        # segments of every entry below come out 100 log2(e) too large: the finalize scale takes it out
        unit = 1.0 / images.SDF_UNIT if prec == "bf16" else 1.0
        maps = [(name, km, nm, sc * unit) for (name, km, nm, sc) in maps]
        for l, (name, km, nm, sc) in enumerate(maps):
            if l == 8:
                ent.append(dict(net="sdf", name=name, rmap=nm, cmap=km, scale=sc, A=AB(8), B=sl(w["H"], 7), bias=True, Pn=P))
                continue
            A, A2 = AB(l), sl(w["V"], l)
            if l == 0:
                ent.append(dict(net="sdf", name=name, rmap=nm, cmap=km, scale=sc, A=A, B=whole(w["PE"]), A2=A2, B2=UB(0), bias=True, Pn=P))
            elif l == 4:
                ent.append(dict(net="sdf", name=name, rmap=nm, cmap=km[:224], scale=sc, A=A, B=sl(w["H"], 3), A2=A2, B2=UB(4), bias=True, Pn=P))
                ent.append(dict(net="sdf", name=name, rmap=nm, cmap=km[224:], scale=sc, A=A, B=whole(w["PE"]), A2=A2, B2=UB(4, 224), bias=False, Pn=P))
            else:
                ent.append(dict(net="sdf", name=name, rmap=nm, cmap=km, scale=sc, A=A, B=sl(w["H"], l - 1), A2=A2, B2=UB(l), bias=True, Pn=P))
        # d W8[row 0, :] += colsum(ub_8) / scale   (u_8 = W8[0,:] / scale)
        ent.append(dict(net="sdf", name="lin8", rmap=images.ident_map(256), cmap=None, scale=1.0, A=UB(8), B=None, bias=False, Pn=P,
                        extra_row0=True))

        def rendering(net, dh, dout, save_h, small, d_out):
            km0 = self.nets[net].img.streams["fwd"][0].kmap
            pad = 96 if d_out == 96 else 32
            ent.append(dict(net=net, name="lin0", rmap=images.ident_map(256), cmap=km0[:256], scale=1.0, A=sl(dh, 0), B=whole(w["feat"]), bias=True, Pn=P))
            ent.append(dict(net=net, name="lin0", rmap=images.ident_map(256), cmap=km0[256:320], scale=1.0, A=sl(dh, 0), B=whole(small), bias=False, Pn=P))
            if len(km0) > 320:        # d_feature = 352: the appended VDN channels
                ent.append(dict(net=net, name="lin0", rmap=images.ident_map(256), cmap=km0[320:], scale=1.0, A=sl(dh, 0), B=whole(w["col_extra"]), bias=False, Pn=P))
            for l in (1, 2, 3):
                ent.append(dict(net=net, name="lin%d" % l, rmap=images.ident_map(256), cmap=images.ident_map(256), scale=1.0,
                                A=sl(dh, l), B=sl(save_h, l - 1), bias=True, Pn=P))
            ent.append(dict(net=net, name="lin4", rmap=images.ident_map(d_out, pad), cmap=images.ident_map(256), scale=1.0,
                            A=whole(dout), B=sl(save_h, 3), bias=True, Pn=P))
        rendering("color", w["col_dh"], w["col_dout"], w["col_h"], w["col_small"], 3)
        if self.wdepth:
            rendering("vdn", w["vdn_dh"], w["vdn_dout"], w["vdn_h"], w["vdn_small"], 96)
        if "nerf" in self.nets:
            st = self.nets["nerf"].img.streams
            km5, kmv = st["_km5"], st["_kmv"]
            I256 = images.ident_map(256)
            dh, h = w["nf_dh"], w["nf_h"]
            ent.append(dict(net="nerf", name="pts_linears.0", rmap=I256, cmap=images.ident_map(84, 96), scale=1.0, A=sl(dh, 0), B=whole(w["nf_pe"]), bias=True, Pn=Q))
            for i in (1, 2, 3, 4, 6, 7):
                ent.append(dict(net="nerf", name="pts_linears.%d" % i, rmap=I256, cmap=I256, scale=1.0, A=sl(dh, i), B=sl(h, i - 1), bias=True, Pn=Q))
            ent.append(dict(net="nerf", name="pts_linears.5", rmap=I256, cmap=km5[:96], scale=1.0, A=sl(dh, 5), B=whole(w["nf_pe"]), bias=True, Pn=Q))
            ent.append(dict(net="nerf", name="pts_linears.5", rmap=I256, cmap=km5[96:], scale=1.0, A=sl(dh, 5), B=sl(h, 4), bias=False, Pn=Q))
            ent.append(dict(net="nerf", name="feature_linear", rmap=I256, cmap=I256, scale=1.0, A=(w["nf_dhead"], 0, 0, 288), B=sl(h, 7), bias=True, Pn=Q))
            ent.append(dict(net="nerf", name="alpha_linear", rmap=images.ident_map(1, 32), cmap=I256, scale=1.0, A=(w["nf_dhead"], 0, 256, 288), B=sl(h, 7), bias=True, Pn=Q))
            ent.append(dict(net="nerf", name="views_linears.0", rmap=images.ident_map(128), cmap=kmv[:256], scale=1.0, A=whole(w["nf_dv"]), B=whole(w["nf_feature"]), bias=True, Pn=Q))
            ent.append(dict(net="nerf", name="views_linears.0", rmap=images.ident_map(128), cmap=kmv[256:], scale=1.0, A=whole(w["nf_dv"]), B=whole(w["nf_vpe"]), bias=False, Pn=Q))
            ldo = w["nf_do"].shape[1]
            ent.append(dict(net="nerf", name="rgb_linear", rmap=images.ident_map(3, 32), cmap=images.ident_map(128), scale=1.0, A=(w["nf_do"], 0, 0, ldo), B=whole(w["nf_hv"]), bias=True, Pn=Q))
            if self.wdepth:
                ent.append(dict(net="nerf", name="dpt_linear", rmap=images.ident_map(96), cmap=images.ident_map(128), scale=1.0, A=(w["nf_do"], 0, 32, ldo), B=whole(w["nf_hv"]), bias=True, Pn=Q))

        # ---- tables
        dev = self.dev
        all_maps, moff = [], 0
        dw = np.zeros(len(ent), dtype=lib.struct_dtype("VdnDwDesc"))
        fin = np.zeros(len(ent), dtype=lib.struct_dtype("VdnDwFinalizeDesc"))
        slab_elems, cs_elems, wg = 0, 0, 0
        lay = []
        # Two launch groups: the SDF network's entries (the critical path into the next step: its sampler only needs the SDF
        # weights) and the rest (colour / VDN heads, background network). `ent` lists the SDF entries first, so the group
        # tables are a prefix and a suffix of the full table; the suffix copy numbers its workgroups from its own zero.
        n_sdf = sum(1 for e in ent if e["net"] == "sdf")
        assert all(e["net"] == "sdf" for e in ent[:n_sdf]) and all(e["net"] != "sdf" for e in ent[n_sdf:])
        for i, e in enumerate(ent):
            mt = len(e["rmap"]) // 32
            nt = 0 if e["cmap"] is None else len(e["cmap"]) // 32
            # K splits are sized on the static row counts (the work lists shrink them at run time: ~2.3 K foreground and
            # ~3.4 K background rows per workgroup on the bench scene). Measured alternatives: equal BYTES per workgroup
            # (long row ranges for narrow operands) is 15-40 % slower - a stage of a narrow entry is latency-, not
            # bandwidth-bound - and 2048 / 8192 rows per split are 10 % slower (more slab traffic / a coarser tail).
            segs = 2 if e.get("A2") is not None else 1
            K = e["Pn"] * segs
            pps = _pts_per_split(e["net"], prec)
            splits = max(1, (K + pps - 1) // pps)
            if segs == 2:
                splits += splits % 2          # first half of the splits = segment 1, second half = segment 2
            lay.append((mt, nt, splits, slab_elems, cs_elems, moff, wg))
            all_maps.append(np.asarray(e["rmap"], np.int32))
            moff_r = moff
            moff += mt * 32
            if nt:
                all_maps.append(np.asarray(e["cmap"], np.int32))
                moff += nt * 32
            slab_elems += splits * mt * 32 * nt * 32
            cs_elems += splits * mt * 32
            wg += lib.call_value("vdn_dw_entry_wgs" + self.sfx, mt, nt, splits)
        self.dw_total_wgs = wg
        self.maps = torch.from_numpy(np.concatenate(all_maps)).to(dev)
        self.slab = torch.empty(max(slab_elems, 1), dtype=torch.float32, device=dev)
        self.colsum = torch.empty(max(cs_elems, 1), dtype=torch.float32, device=dev)
        P4 = lambda spec: spec[0].data_ptr() + spec[0].element_size() * (spec[1] + layout.col_offset_elems(spec[2], prec))
        self.dw_list_kind = []
        for i, e in enumerate(ent):
            mt, nt, splits, so, co, mo, wg0 = lay[i]
            d = dw[i]
            d["A1"], d["lda1"] = P4(e["A"]), e["A"][3]
            if e["B"] is not None:
                d["B1"], d["ldb1"] = P4(e["B"]), e["B"][3]
            if e.get("A2") is not None:
                d["A2"], d["lda2"], d["B2"], d["ldb2"] = P4(e["A2"]), e["A2"][3], P4(e["B2"]), e["B2"][3]
            d["P"], d["m_tiles"], d["n_tiles"], d["splits"], d["wg_begin"] = e["Pn"], mt, nt, splits, wg0
            # rows of the compact work lists (device scalars written by the forward)
            d["P_dev"] = (w["bg_active"] if e["net"] == "nerf" else w["fg_active"])[1].data_ptr()
            self.dw_list_kind.append("bg" if e["net"] == "nerf" else "fg")
            d["slab"] = self.slab.data_ptr() + 4 * so
            want_cs = e["bias"] or e.get("extra_row0")
            d["colsum"] = self.colsum.data_ptr() + 4 * co if want_cs else 0
            net = self.nets[e["net"]]
            fd = fin[i]
            fd["slab"], fd["colsum"] = d["slab"], d["colsum"]
            fd["rmap"] = self.maps.data_ptr() + 4 * mo
            fd["cmap"] = self.maps.data_ptr() + 4 * (mo + mt * 32)
            fd["splits"], fd["M"], fd["N"] = splits, mt * 32, nt * 32
            fd["scale"] = e["scale"]
            if e.get("extra_row0"):
                # colsum(ub_8) / scale joins row 0 of d W8 inside the lin8 entry's own finalize (VdnDwFinalizeDesc.xsum): no
                # '+=' descriptor, no second finalize launch. (The sums are indexed by the image row = the target column.)
                main = next(k for k, x in enumerate(ent) if x["net"] == e["net"] and x["name"] == e["name"] and not x.get("extra_row0"))
                fm = fin[main]
                fm["xsum"], fm["xsplits"], fm["xM"], fm["xrow"] = d["colsum"], splits, mt * 32, 0
                fm["xscale"] = 1.0 / float(self.r.sdf_network.scale)
                fd["slab"], fd["colsum"] = 0, 0
            else:
                tgt = net.dw_target(e["name"])
                fd["target"], fd["t_stride"] = tgt.data_ptr(), tgt.shape[1]
                if e["bias"]:
                    bt = net.bias_target(e["name"])
                    fd["btarget"], fd["bscale"] = bt.data_ptr(), 1.0
        self.dw_table = torch.from_numpy(dw.view(np.uint8)).to(dev)
        wg_sdf = int(dw[n_sdf]["wg_begin"]) if n_sdf < len(ent) else wg
        rest = dw[n_sdf:].copy()
        rest["wg_begin"] -= wg_sdf
        self.dw_groups = {"sdf": (self.dw_table, n_sdf, wg_sdf)}
        if len(rest):
            self.dw_groups["rest"] = (torch.from_numpy(rest.view(np.uint8)).to(dev), len(rest), wg - wg_sdf)
        # the rest group once more as its two halves (the Trainer's schedule launches them at different times: DESIGN.md 3d):
        # "heads" = colour / VDN heads, "nerf" = the background network. `ent` lists sdf, heads, nerf in this order.
        n_heads_end = n_sdf + sum(1 for e in ent if e["net"] not in ("sdf", "nerf"))
        assert all(e["net"] == "nerf" for e in ent[n_heads_end:])
        sub_ranges = {"heads": (n_sdf, n_heads_end), "nerf": (n_heads_end, len(ent))}
        for gname, (lo, hi) in sub_ranges.items():
            if hi > lo:
                sub = dw[lo:hi].copy()
                wg_lo = int(dw[lo]["wg_begin"])
                wg_hi = int(dw[hi]["wg_begin"]) if hi < len(ent) else wg
                sub["wg_begin"] -= wg_lo
                self.dw_groups[gname] = (torch.from_numpy(sub.view(np.uint8)).to(dev), hi - lo, wg_hi - wg_lo)
        # one more finalize descriptor: d loss / d variance = sum over rays of the compositor's per-ray partials (a [B,1] "column
        # sum" with one row): the reduction rides in the finalize launch instead of a launch of its own
        vfin = np.zeros(1, dtype=fin.dtype)
        self._var_map = torch.zeros(1, dtype=torch.int32, device=dev)
        vfin[0]["colsum"], vfin[0]["rmap"] = w["d_var_partial"].data_ptr(), self._var_map.data_ptr()
        vfin[0]["btarget"], vfin[0]["bscale"] = self.grad_views[id(self.r.deviation_network.variance)].data_ptr(), 1.0
        vfin[0]["splits"], vfin[0]["M"], vfin[0]["N"] = self.B, 1, 0
        fin_all = np.concatenate([fin, vfin])
        self.fin_table = torch.from_numpy(fin_all.view(np.uint8)).to(dev)
        self.fin_has_phase1 = bool((fin_all["accumulate"] != 0).any())
        self.n_fin = len(fin_all)
        self.n_dw = len(ent)
        self.fin_max_M = int(max(len(e["rmap"]) for e in ent))
        # per-group finalize tables (the variance's reduction rides with the SDF group)
        fg = {"sdf": np.concatenate([fin[:n_sdf], vfin]), "rest": fin[n_sdf:]}
        mm = {"sdf": ent[:n_sdf], "rest": ent[n_sdf:]}
        for gname, (lo, hi) in sub_ranges.items():
            fg[gname], mm[gname] = fin[lo:hi], ent[lo:hi]
        self.fin_groups = {k: (torch.from_numpy(v.view(np.uint8).copy()).to(dev), len(v), int(max(len(e["rmap"]) for e in mm[k])),
                               bool((v["accumulate"] != 0).any())) for k, v in fg.items() if len(v) and len(mm[k])}
        # weight-norm backward table
        rows, row_group = [], []
        for key, net in self.nets.items():          # "sdf" first (dict order of self.nets)
            for name, (g, v, b) in net.img.matrices.items():
                if g is not None:
                    rows.append((g, v, net.img.inv_norm[net.img.r_off[name]:], net.dweff_view(name), net.grads[id(g)], net.grads[id(v)]))
                    row_group.append(("sdf",) if key == "sdf" else ("rest", "nerf" if key == "nerf" else "heads"))
        wn = np.zeros(len(rows), dtype=lib.struct_dtype("VdnWeightNormBwdDesc"))
        for i, (g, v, inv, dwe, dg, dv) in enumerate(rows):
            wn[i]["g"], wn[i]["v"], wn[i]["inv_norm"], wn[i]["dw_eff"] = g.data_ptr(), v.data_ptr(), inv.data_ptr(), dwe.data_ptr()
            wn[i]["dg"], wn[i]["dv"], wn[i]["rows"], wn[i]["cols"] = dg.data_ptr(), dv.data_ptr(), v.shape[0], v.shape[1]
        self.wn_table = torch.from_numpy(wn.view(np.uint8).copy() if len(rows) else np.zeros(8, np.uint8)).to(dev)
        self.n_wn = len(rows)
        self.wn_max_rows = max([r[1].shape[0] for r in rows] + [1])
        self.wn_groups = {}
        for k in ("sdf", "rest", "heads", "nerf"):
            idx = [i for i, gk in enumerate(row_group) if k in gk]
            if idx:
                self.wn_groups[k] = (torch.from_numpy(wn[idx].view(np.uint8).copy()).to(dev), len(idx), max(rows[i][1].shape[0] for i in idx))
        self._param_ptrs = self._ptr_key()

    @property
    def grad_flat(self):
        """The flat gradient buffer (dpt_runner.py:121-130 parameter order), complete on torch's current stream."""
        if getattr(self, "join_hook", None) is not None:     # the Trainer's deferred half of the backward (side stream)
            self.join_hook()
        return self._grad_flat

    def _ptr_key(self):
        return tuple(p.data_ptr() for net in self.nets.values() for p in lib.module_params(net.module))

    # ------------------------------------------------------------------------------------------
    def _ray_workspaces(self):
        """Buffers of the ray adjoint (differentiable rays_o / rays_d / z, poses.py:198-208), allocated on first use."""
        w, B, N, T, P, Q, dev = self.w, self.B, self.N, self.T, self.P, self.Q, self.dev
        if "U_pe" not in w:
            f = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
            w["U_pe"], w["d_pts"], w["d_dirs"] = f(P, 39), f(P, 3), f(P, 3)
            w["d_dists"], w["d_dir_cos"] = f(B, N), f(B, 3)
            w["d_rays_o"], w["d_rays_d"], w["d_z"] = f(B, 3), f(B, 3), f(B, N)
            if self.r.n_outside > 0:
                w["d_bg_pts"], w["d_bg_dirs"], w["d_bg_dists"], w["d_z_out"] = f(Q, 3), f(Q, 3), f(B, T), f(B, T - N)
        return w

    def forward(self, rays_o, rays_d, z, z_out, background_rgb, cos_anneal_ratio, skip_far=False, ray_grads=False, pending_merge=None,
                after_sdf=None, fuse_loss=None, before_heads=None, rest_normals=False, cos_anneal_dev=None):
        """Differentiable part of render() at detached z [B,N] (+ z_out [B,O]); returns the output tensors.
        skip_far (the Trainer's hot loop): inside samples beyond the relaxed sphere (|p| >= 1.2: inside_sphere = 0 and
        relax_inside_sphere = 0, renderer.py:284-286) enter the loss only through exact zeros, so the SDF / colour / VDN
        networks skip them; `normals` / `sdf` are then only valid at the listed points (render() never sets it: it returns
        `gradients` for every sample).
        rest_normals (render() under grad, with skip_far): the samples the work list skips still get their `sdf` and `normals`
        (render() returns `gradients` for every sample, and `cdf_fine` is made from them) from an inference launch of the SDF
        network on the list's complement - no saves, no colour / VDN head, no backward there.
        fuse_loss (the Trainer's plain configuration): dict(true_rgb, g_color, igr_weight, grad_scale) - the compositor, the
        colour term's gradient and the compositor's adjoint run as ONE launch (vdn_composite_train); backward() then starts at
        the heads. With the VDN head and the dict's gt_feats / g_feats / depth_weight: vdn_composite_fwd_train here and
        vdn_composite_bwd_train in backward() (the loss gradients are made inside them). Needs skip_far (the eikonal denominator
        is the foreground list's length).
        cos_anneal_dev: a [1] device tensor the compositor (and its adjoint in backward()) reads cos_anneal_ratio from instead of the
        by-value argument (VdnCompositeArgs.cos_anneal_dev): launches captured in a HIP graph follow a changing ratio."""
        r, w, B, N, T = self.r, self.w, self.B, self.N, self.T
        st = _stream()
        if ray_grads and skip_far:
            raise ValueError("ray gradients need every inside sample evaluated (skip_far=False)")
        self._ray_grads = bool(ray_grads)
        self._car_dev = cos_anneal_dev
        self._fwd_rays = (rays_o, rays_d)
        if ray_grads:
            self._ray_workspaces()
        # before_heads (the Trainer): called right in front of the colour / VDN heads' launches - the first ones on this stream that
        # read what its side stream updates (the wait for that update then sits behind the SDF kernel instead of in front of the
        # step preparation, where the update has only just finished; the background network runs on the side stream itself)
        for net in self.nets.values():
            net.img = net.module._images(join=before_heads is None)           # refresh weight images if parameters changed
        if self._ptr_key() != self._param_ptrs:
            raise RuntimeError("parameters were re-allocated after the training engine was built; rebuild it")
        sample_dist = 2.0 / r.n_samples
        O = r.n_outside
        # without a background pass (n_outside = 0) render_core does not blend with inside_sphere (renderer.py:289): every
        # foreground sample counts, nothing may be skipped
        self._fg_compact = bool(skip_far) and O > 0 and os.environ.get("VDN_FG_COMPACT", "1") != "0"
        # VDN_TRAIN_COLOR_FUSED=1 (bf16; OFF by default): the colour head rides in the SDF kernel's launch (csrc/k_sdf_fwd2.h MODE 3: the
        # feature vector stays in registers, the head's saved planes are written from there) - one launch and one trip of the
        # feature plane less, and measured SLOWER in the step: same-box A/B, three alternating runs, 1 141 / 1 138 / 1 121 us against
        # 1 113 / 1 123 / 1 122 with the two launches. The 128-row workgroups of the SDF kernel run whole rounds (two on the bench
        # scene's ~36 K rows), each 11 us longer with the head's 33 chunk steps, while rendernet_fwd's 288 workgroups share the
        # chip with the background network at no such price; and the join with the side stream's update of the head's weights
        # has to move in front of the SDF kernel. Not with ray gradients (U_pe), the depth_before_color input, or the tail split.
        self._color_fused = (self.precision == "bf16" and not self.dbc and not ray_grads and "c2" in self.nets["color"].img.blobs
                             and r.color_network.conf["d_out"] == 3 and os.environ.get("VDN_TRAIN_COLOR_FUSED", "0") == "1"
                             and not self._tail_wanted())
        from dpt_models.renderer import background_active, bg_compaction
        self._bg_compact = O > 0 and bg_compaction()
        fused_prep = self._bg_compact and os.environ.get("VDN_FUSED_PREP", "1") != "0"
        if pending_merge is not None and not fused_prep:
            # NeuSRenderer._sample(defer_last_merge=True) left the last round's samples unmerged: complete z here
            new_z, M_old = pending_merge
            m = lib.VdnMergeArgs()
            m.z, m.new_z, m.z_out = z.data_ptr(), new_z.data_ptr(), z.data_ptr()
            m.B, m.M, m.K, m.ld, m.ld_out = B, M_old, N - M_old, z.stride(0), z.stride(0)
            lib.call("vdn_merge_sorted", m, st)
            pending_merge = None
        if O > 0 and not fused_prep:
            m = lib.VdnMergeArgs()
            m.z, m.new_z, m.z_out = z.data_ptr(), z_out.data_ptr(), w["z_feed"].data_ptr()
            m.B, m.M, m.K, m.ld, m.ld_out = B, N, O, z.stride(0), T
            lib.call("vdn_merge_sorted", m, st)
        if fused_prep:
            # z_feed, the sections of both depth sets and both work lists in two launches (vdn_train_prep) instead of seven
            tp = lib.VdnTrainPrepArgs()
            tp.rays_o, tp.rays_d, tp.z, tp.z_out, tp.z_feed = (t.data_ptr() for t in (rays_o, rays_d, z, z_out, w["z_feed"]))
            tp.B, tp.N, tp.T, tp.z_ld, tp.sample_dist, tp.fg_radius = B, N, T, z.stride(0), sample_dist, 1.2
            if pending_merge is not None:
                tp.new_z, tp.M_old = pending_merge[0].data_ptr(), pending_merge[1]
            tp.dists, tp.mid_z, tp.bg_dists, tp.bg_mid = (w[k].data_ptr() for k in ("dists", "mid_z", "bg_dists", "bg_mid"))
            if self._fg_compact:
                tp.fg_active_idx, tp.fg_n_active, tp.fg_ray_counts = (t.data_ptr() for t in w["fg_active"])
            tp.bg_active_idx, tp.bg_n_active, tp.bg_ray_counts = (t.data_ptr() for t in w["bg_active"])
            lib.call("vdn_train_prep", tp, st)
            if fuse_loss is not None and fuse_loss.get("after_prep") is not None:
                fuse_loss["after_prep"](self)            # (the data-parallel Trainer: the list's length starts its all-reduce here)
        else:
            a = lib.VdnSectionArgs()
            a.z, a.dists, a.mid_z, a.sample_dist, a.B, a.n, a.ld = z.data_ptr(), w["dists"].data_ptr(), w["mid_z"].data_ptr(), sample_dist, B, N, z.stride(0)
            lib.call("vdn_sections", a, st)
            if self._fg_compact:
                fa = lib.VdnForegroundActiveArgs()
                fa.rays_o, fa.rays_d, fa.mid_z, fa.B, fa.N, fa.radius = rays_o.data_ptr(), rays_d.data_ptr(), w["mid_z"].data_ptr(), B, N, 1.2
                fa.active_idx, fa.n_active, fa.ray_counts = (t.data_ptr() for t in w["fg_active"])
                lib.call("vdn_foreground_active", fa, st)
        if not self._fg_compact:
            w["fg_active"][1].fill_(self.P)               # the dW GEMM's device-side row count
        if O > 0:
            if not fused_prep:
                a = lib.VdnSectionArgs()
                a.z, a.dists, a.mid_z, a.sample_dist, a.B, a.n, a.ld = w["z_feed"].data_ptr(), w["bg_dists"].data_ptr(), w["bg_mid"].data_ptr(), sample_dist, B, T, T
                lib.call("vdn_sections", a, st)
            n = lib.VdnNerfArgs()
            n.blob = self.nets["nerf"].img.blobs["fwd"].data_ptr()
            n.rays_o, n.rays_d, n.z, n.n_per_ray, n.P = rays_o.data_ptr(), rays_d.data_ptr(), w["bg_mid"].data_ptr(), T, self.Q
            n.density, n.rgb = w["bg_density"].data_ptr(), w["bg_rgb"].data_ptr()
            n.feat = w["bg_feat"].data_ptr() if w["bg_feat"] is not None else None
            n.save_h, n.save_pe, n.save_feature, n.save_vpe, n.save_hv = (w[k].data_ptr() for k in ("nf_h", "nf_pe", "nf_feature", "nf_vpe", "nf_hv"))
            # only the background samples the compositor does not multiply by zero (saves are in compact order)
            if self._bg_compact:
                if not fused_prep:
                    background_active(rays_o, rays_d, w["mid_z"], T, out=w["bg_active"])
                n.active_idx, n.n_active = w["bg_active"][0].data_ptr(), w["bg_active"][1].data_ptr()
            else:
                w["bg_active"][1].fill_(self.Q)           # the dW GEMM's device-side row count
            # the NeRF++ background is independent of the SDF / colour path until compositing: it may run on a side stream
            # (81 920 background points = 1.25 rounds of the CUs; the tail round could overlap the SDF kernels)
            self._fork()
            lib.call("vdn_nerf_mlp_fwd" + self.sfx, n, self._side_handle(st))
            if rest_normals and self._fg_compact:
                self._sdf_rest(rays_o, rays_d, self._side_handle(st))     # (beside the training launch on the listed samples)
            self._side_done()
        if self._color_fused and before_heads is not None:
            before_heads()                  # (the fused launch reads the colour head's weight image, which the side stream updates)
            before_heads = None
        self._sdf_forward(rays_o, rays_d)
        if after_sdf is not None:       # (the data-parallel Trainer reduces the eikonal sums over the ranks from here on)
            after_sdf(self)

        def rnet(net, out, save_h, small, d_out, module, stream=None):
            c = lib.VdnRenderNetArgs()
            c.blob = self.nets[net].img.blobs["fwd"].data_ptr()
            c.rays_o, c.rays_d, c.z, c.n_per_ray = rays_o.data_ptr(), rays_d.data_ptr(), w["mid_z"].data_ptr(), N
            c.normals, c.feat, c.out = w["normals"].data_ptr(), w["feat"].data_ptr(), out.data_ptr()
            c.save_h, c.save_small = save_h.data_ptr(), small.data_ptr()
            c.P, c.d_out, c.squeeze_out = self.P, d_out, int(module.squeeze_out)
            if net == "color" and self.dbc:          # renderer.py:247-248
                c.extra, c.save_extra = w["vdn_out"].data_ptr(), w["col_extra"].data_ptr()
            lib.call("vdn_rendernet_fwd" + self.sfx, self._fg(c), st if stream is None else stream)
        if before_heads is not None:
            before_heads()
        vdn_beside = self.wdepth and self._side2 is not None and not self.dbc     # (depth_before_color: the colour head reads the VDN output)
        if self.wdepth:
            if vdn_beside:
                self._ev_fork2.record(lib.current_stream())
                self._side2.wait_event(self._ev_fork2)
                rnet("vdn", w["vdn_out"], w["vdn_h"], w["vdn_small"], 96, r.depth_network, stream=self._side2.cuda_stream)
                self._ev_join2.record(self._side2)
            else:
                rnet("vdn", w["vdn_out"], w["vdn_h"], w["vdn_small"], 96, r.depth_network)
        if not self._color_fused:
            rnet("color", w["col_out"], w["col_h"], w["col_small"], 3, r.color_network)
        if vdn_beside:
            lib.current_stream().wait_event(self._ev_join2)

        c = self._composite_common(lib.VdnCompositeArgs(), rays_o, rays_d, background_rgb, cos_anneal_ratio)
        c.weights, c.alpha_out, c.cdf, c.inside_sphere = w["weights"].data_ptr(), w["alpha"].data_ptr(), w["cdf"].data_ptr(), w["inside"].data_ptr()
        c.color_out, c.weight_sum, c.weight_max, c.s_val = w["color"].data_ptr(), w["wsum"].data_ptr(), w["wmax"].data_ptr(), w["s_val"].data_ptr()
        c.eik_partial, c.eik_out = w["eik_partial"].data_ptr(), w["eik"].data_ptr()
        if self.wdepth:
            c.feat_out = w["feat_out"].data_ptr()
        self._join()
        self._ctx = (rays_o, rays_d, background_rgb, cos_anneal_ratio, z)
        self._composite_bwd_done = False
        self._bwd_train = None
        if fuse_loss is not None and fuse_loss.get("before_composite") is not None:
            if not (fused_prep and self._fg_compact):
                raise RuntimeError("fuse_loss with a global foreground count needs the fused step preparation and the work lists")
            fuse_loss["before_composite"]()              # (... and has arrived: vdn_composite_train / _bwd_train read the global count)
        fg_count = lib.ptr(fuse_loss["fg_count"] if (fuse_loss is not None and fuse_loss.get("fg_count") is not None) else w["fg_active"][1])
        if fuse_loss is not None and self._fg_compact and self.wdepth and fuse_loss.get("gt_feats") is not None and not ray_grads:
            # with the VDN head: the per-ray kernel and the features' weighted sums, which write d loss / d render_feats on the spot;
            # backward() then makes the colour term's gradient inside the compositor's adjoint (vdn_composite_bwd_train)
            lib.call("vdn_composite_fwd_train", c, lib.ptr(fuse_loss["gt_feats"]), lib.ptr(fuse_loss["g_feats"]),
                     float(fuse_loss["depth_weight"]), float(fuse_loss["grad_scale"]), st)
            self._bwd_train = fuse_loss
        elif fuse_loss is not None and self._fg_compact and not self.wdepth and not ray_grads:
            cb = self._composite_bwd_args(None, None, None, None, None)
            self._fused_keep = (c, cb, fuse_loss)
            lib.call("vdn_composite_train", c, cb, lib.ptr(fuse_loss["true_rgb"]), lib.ptr(fuse_loss["g_color"]), fg_count,
                     float(fuse_loss["igr_weight"]), float(fuse_loss["grad_scale"]), st)
            self._composite_bwd_done = True
        else:
            lib.call("vdn_alpha_composite_fwd", c, st)
        self.generation = getattr(self, "generation", 0) + 1
        return w

    def _fg(self, args):
        """Attach the foreground work list of the current forward to a kernel argument block."""
        if self._fg_compact:
            args.active_idx, args.n_active = self.w["fg_active"][0].data_ptr(), self.w["fg_active"][1].data_ptr()
        return args

    # ---- optional side stream for the background network (VDN_SIDE_STREAM=1; default: everything on the caller's stream)
    def _fork(self):
        if self._side is not None:
            self._ev_fork.record(lib.current_stream())
            self._side.wait_event(self._ev_fork)

    def _side_handle(self, main_handle):
        return main_handle if self._side is None else self._side.cuda_stream

    def _side_done(self):
        if self._side is not None:
            self._ev_join.record(self._side)
            self._pending = True

    def _join(self):
        if self._side is not None and self._pending:
            lib.current_stream().wait_event(self._ev_join)
            self._pending = False

    def _tail_wanted(self):
        """The tail split of the fused SDF forward (csrc/k_sdf_fwd1_split.h): on where there is no side stream, off with one."""
        tail_default = "1" if self._side is None else "0"
        return (self.precision == "bf16" and self._fg_compact and os.environ.get("VDN_SDF_TAIL", tail_default) != "0"
                and not getattr(self, "_ray_grads", False))

    def _sdf_rest(self, rays_o, rays_d, stream):
        """`sdf` and `normals` of the inside samples the foreground work list skips: the list's complement (vdn_foreground_active,
        complement = 1) through the SDF kernel's inference form (mode 1 without saves; its feature plane goes to a backward
        intermediate that is free during the forward)."""
        w, N = self.w, self.N
        if "fg_rest" not in w:
            w["fg_rest"] = (torch.zeros(self.P, dtype=torch.int32, device=self.dev), torch.zeros(1, dtype=torch.int32, device=self.dev),
                            torch.zeros(self.B, dtype=torch.int32, device=self.dev))
        fa = lib.VdnForegroundActiveArgs()
        fa.rays_o, fa.rays_d, fa.mid_z, fa.B, fa.N, fa.radius = rays_o.data_ptr(), rays_d.data_ptr(), w["mid_z"].data_ptr(), self.B, N, 1.2
        fa.complement = 1
        fa.active_idx, fa.n_active, fa.ray_counts = (t.data_ptr() for t in w["fg_rest"])
        lib.call("vdn_foreground_active", fa, stream)
        s = lib.VdnSdfArgs()
        img = self.nets["sdf"].img
        s.blob = img.blobs["full"].data_ptr()
        s.rays_o, s.rays_d, s.z, s.n_per_ray, s.z_ld, s.sdf_ld = rays_o.data_ptr(), rays_d.data_ptr(), w["mid_z"].data_ptr(), N, N, N
        s.P, s.scale = self.P, float(self.r.sdf_network.scale)
        s.sdf, s.feat, s.normals = w["sdf"].data_ptr(), w["d_featvec"].data_ptr(), w["normals"].data_ptr()
        s.w8row = img.weff_view("lin8").data_ptr()
        if "S" in w:              # (the fp32 kernel keeps softplus' in a plane between its forward pass and its sweep)
            if "S_rest" not in w:
                w["S_rest"] = torch.empty_like(w["S"])
            s.S = w["S_rest"].data_ptr()
        s.active_idx, s.n_active = w["fg_rest"][0].data_ptr(), w["fg_rest"][1].data_ptr()
        lib.call("vdn_sdf_mlp_fwd" + self.sfx, 1, s, stream)

    def _sdf_forward(self, rays_o, rays_d):
        """The fused SDF kernel (PE -> 9 layers -> sdf / feature + gradient sweep) with the training-mode saves, on the
        section mid-points currently in the workspace. Separate so that bench.py can time exactly this launch."""
        r, w, N = self.r, self.w, self.N
        s = lib.VdnSdfArgs()
        img = self.nets["sdf"].img
        s.blob = img.blobs["full"].data_ptr()
        s.rays_o, s.rays_d, s.z, s.n_per_ray, s.z_ld, s.sdf_ld = rays_o.data_ptr(), rays_d.data_ptr(), w["mid_z"].data_ptr(), N, N, N
        s.P, s.scale = self.P, float(r.sdf_network.scale)
        s.sdf, s.feat, s.normals = w["sdf"].data_ptr(), w["feat"].data_ptr(), w["normals"].data_ptr()
        if "S" in w:
            s.S = w["S"].data_ptr()
        s.w8row = img.weff_view("lin8").data_ptr()
        s.H, s.V, s.PE = w["H"].data_ptr(), w["V"].data_ptr(), w["PE"].data_ptr()
        if getattr(self, "_ray_grads", False):
            s.U_pe = w["U_pe"].data_ptr()
        # The tail of the work list (bf16): the 128-row kernel runs whole rounds of one workgroup per CU, so ~37 K rows cost two
        # rounds; when the list ends within VDN_SDF_TAIL_MAX rows behind the last full round, those rows go to the 32-row
        # feature-split kernel instead (csrc/k_sdf_fwd1_split.h: same planes, bit for bit; both launches decide on the device-side
        # row count). VDN_SDF_TAIL=0: off. (Not with ray gradients: the tail kernel does not write U_pe.)
        # Default: only on the one-stream schedule. The tail kernel buys latency with CU time (32 rows per 45 us against 128 per
        # 80): it shortens the launch when the second round's CUs would idle (-37 us per step on one stream), but on the default
        # two-stream schedule those CUs run the background network and the extra CU time costs +7 .. +20 us (same-box A/B).
        tail = self._tail_wanted()
        fused = getattr(self, "_color_fused", False) and not tail
        if tail:
            row0 = int(os.environ.get("VDN_SDF_TAIL_ROW0", str(128 * torch.cuda.get_device_properties(self.dev).multi_processor_count)))
            tail = self.P > row0
            if tail:
                s.tail_row0, s.tail_max_rows = row0, int(os.environ.get("VDN_SDF_TAIL_MAX", "8192"))

        def launch():
            if fused:
                cimg = self.nets["color"].img
                lib.call("vdn_sdf_color_train_bf16", self._fg(s), lib.ptr(cimg.blobs["c2"]), int(r.color_network.squeeze_out),
                         lib.ptr(w["col_h"]), lib.ptr(w["col_small"]), lib.ptr(w["col_out"]), _stream())
                return
            lib.call("vdn_sdf_mlp_fwd" + self.sfx, 1, self._fg(s), _stream())
            if tail:
                lib.call("vdn_sdf_fwd_tail_bf16", s, _stream())
        probe = getattr(self, "sdf_probe", None)
        if probe is None:
            launch()
            return
        # bench.py's in-situ timing of the north-star kernel: HIP events on the launch stream right around THIS step's launch,
        # with the step's row count (read back after the run)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        launch()
        e1.record()
        probe.append((e0, e1, w["fg_active"][1].clone() if self._fg_compact else None))

    def _dw_rows(self):
        """Rows each weight-gradient entry contracts over in the current step (the work lists of the last forward)."""
        n = {"bg": int(self.w["bg_active"][1].item()) if "bg_active" in self.w else 0, "fg": int(self.w["fg_active"][1].item())}
        tab = np.frombuffer(self.dw_table.cpu().numpy().tobytes(), dtype=lib.struct_dtype("VdnDwDesc"))
        return tab, [min(int(d["P"]), n[k]) for d, k in zip(tab, self.dw_list_kind)]

    def dw_bytes(self):
        """Algorithmic HBM bytes of one weight-gradient GEMM launch: every operand plane read once + the slabs written."""
        esz = 4 if self.precision == "fp32" else 2
        total = 0
        tab, rows = self._dw_rows()
        for d, r in zip(tab, rows):
            segs = 2 if d["A2"] else 1
            total += segs * r * 32 * (int(d["m_tiles"]) + int(d["n_tiles"])) * esz
            total += int(d["splits"]) * int(d["m_tiles"]) * 32 * int(d["n_tiles"]) * 32 * 4
        return total

    def dw_flops(self):
        tab, rows = self._dw_rows()
        return sum(2.0 * (2 if d["A2"] else 1) * r * int(d["m_tiles"]) * 32 * int(d["n_tiles"]) * 32 for d, r in zip(tab, rows))

    def _launch_dw(self):
        lib.call("vdn_dw_gemm" + self.sfx, lib.ptr(self.dw_table), self.n_dw, self.dw_total_wgs, _stream())

    def _launch_dw_groups(self):
        """The weight-gradient GEMM as the Trainer's step launches it: one launch per group (bench.py times this)."""
        for group in ("sdf", "rest"):
            if group in self.dw_groups:
                tab, n, wgs = self.dw_groups[group]
                lib.call("vdn_dw_gemm" + self.sfx, lib.ptr(tab), n, wgs, _stream())

    def side_weight_grads(self, group, after=None, gemm_event=None):
        """One half of the rest group on the side stream (backward(defer_rest=True) came first): "nerf" = the background
        network's weight gradients, behind its backward on that stream; "heads" = the colour / VDN heads', behind the heads'
        backward on the caller's stream. `after`: one more event to wait for. gemm_event: recorded on the side stream right
        behind the GEMM launch - the last reader of the forward's workspaces on that stream (saved planes, deltas, the device-side
        row counts of the work lists): the next step's preparation, which rewrites them, waits for it. Returns the torch
        stream (None without a side stream: the work was issued on the caller's)."""
        if self._side is None:
            self.weight_grads(group, _stream())
            return None
        if after is not None:           # (recorded on the caller's stream behind the heads' backward: covers it)
            self._side.wait_event(after)
        elif group != "nerf":
            self._side.wait_event(self._ev_heads)
        self.weight_grads(group, self._side.cuda_stream, gemm_event, event_stream=self._side)
        self._pending = False            # joined by the caller's own event, not by _join()
        return self._side

    def rest_weight_grads(self, after=None, gemm_event=None):
        """Second half of backward(defer_rest=True): the colour / VDN / background networks' weight gradients, issued on the side
        stream behind the background network's backward (or, without a side stream, on the caller's stream). `after`: an event
        the work should also wait for (the Trainer passes the SDF group's GEMM, so that the two HBM-bound GEMMs do not share the
        chip and this one runs beside the next step's sampler instead). Returns the torch stream it was issued on (None = the
        caller's): the caller orders its next use of those gradients / planes behind it."""
        if self._side is None:
            self.weight_grads("rest", _stream())
            return None
        if after is not None:           # (recorded on the caller's stream behind the heads' backward: covers it)
            self._side.wait_event(after)
        else:
            self._side.wait_event(self._ev_heads)
        self.weight_grads("rest", self._side.cuda_stream, gemm_event, event_stream=self._side)
        self._pending = False            # joined by the caller's own event, not by _join()
        return self._side

    def weight_grads(self, group, stream, gemm_event=None, event_stream=None):
        """Weight-gradient GEMM + finalize + weight-norm backward of one launch group ("sdf": the SDF network and the variance;
        "rest": colour / VDN heads and the background network) on `stream` (a raw handle). The two groups touch disjoint
        slabs and disjoint ranges of the flat gradient buffer. gemm_event: recorded right behind the GEMM launch on
        `event_stream` (the torch stream object of `stream`; default: torch's current stream, which must then be `stream`)."""
        if group in self.dw_groups:
            tab, n, wgs = self.dw_groups[group]
            lib.call("vdn_dw_gemm" + self.sfx, lib.ptr(tab), n, wgs, stream)
        if gemm_event is not None:
            gemm_event.record(event_stream if event_stream is not None else lib.current_stream())
        if group in self.fin_groups:
            tab, n, max_m, phase1 = self.fin_groups[group]
            lib.call("vdn_dw_finalize", lib.ptr(tab), n, max_m, 0, stream)
            if phase1:
                lib.call("vdn_dw_finalize", lib.ptr(tab), n, max_m, 1, stream)
        if group in self.wn_groups:
            tab, n, max_rows = self.wn_groups[group]
            lib.call("vdn_weightnorm_bwd", lib.ptr(tab), n, max_rows, stream)

    def _composite_common(self, c, rays_o, rays_d, background_rgb, cos_anneal_ratio):
        w, r = self.w, self.r
        c.rays_o, c.rays_d, c.sdf, c.normals = rays_o.data_ptr(), rays_d.data_ptr(), w["sdf"].data_ptr(), w["normals"].data_ptr()
        c.dists, c.mid_z, c.color = w["dists"].data_ptr(), w["mid_z"].data_ptr(), w["col_out"].data_ptr()
        c.variance = r.deviation_network.variance.data_ptr()
        if self.wdepth:
            c.feat, c.feat_ch = w["vdn_out"].data_ptr(), 96
        if r.n_outside > 0:
            c.bg_density, c.bg_rgb, c.bg_dists = w["bg_density"].data_ptr(), w["bg_rgb"].data_ptr(), w["bg_dists"].data_ptr()
            if self.wdepth:
                c.bg_feat = w["bg_feat"].data_ptr()
        if background_rgb is not None:
            c.background_rgb = background_rgb.data_ptr()
        c.cos_anneal_ratio = float(cos_anneal_ratio)
        if getattr(self, "_car_dev", None) is not None:
            c.cos_anneal_dev = self._car_dev.data_ptr()
        c.B, c.N, c.T = self.B, self.N, self.T
        return c

    def _composite_bwd_args(self, g_color, g_feat, g_weights, g_eik, g_cdf):
        """Argument block of the compositor's adjoint (vdn_alpha_composite_bwd / vdn_composite_train) for the last forward."""
        r, w = self.r, self.w
        rays_o, rays_d, background_rgb, car, z = self._ctx
        c = self._composite_common(lib.VdnCompositeBwdArgs(), rays_o, rays_d, background_rgb, car)
        c.alpha, c.weights, c.eik = w["alpha"].data_ptr(), w["weights"].data_ptr(), w["eik"].data_ptr()
        c.g_cdf = g_cdf.data_ptr() if g_cdf is not None else None
        c.g_color = g_color.data_ptr() if g_color is not None else None
        c.g_feat = g_feat.data_ptr() if (g_feat is not None and self.wdepth) else None
        c.g_weights = g_weights.data_ptr() if g_weights is not None else None
        c.g_eik = g_eik.data_ptr() if g_eik is not None else None
        c.d_sdf, c.d_normals, c.d_color = w["d_sdf"].data_ptr(), w["d_normals"].data_ptr(), w["d_color"].data_ptr()
        if self.wdepth:
            c.d_feat = w["d_vdn"].data_ptr()
            c.feat_scratch = w["feat_scratch"].data_ptr()
            if g_feat is None:
                w["d_vdn"].zero_()
        if r.n_outside > 0:
            c.d_bg_density, c.d_bg_rgb = w["d_bg_density"].data_ptr(), w["d_bg_rgb"].data_ptr()
            if self.wdepth:
                c.d_bg_feat = w["d_bg_feat"].data_ptr()
                if g_feat is None:
                    w["d_bg_feat"].zero_()
        c.d_var_partial = w["d_var_partial"].data_ptr()       # summed into the variance's gradient by the finalize launch (_build_dw_plan)
        if getattr(self, "_ray_grads", False):
            c.d_dists, c.d_dir_cos = w["d_dists"].data_ptr(), w["d_dir_cos"].data_ptr()
            if r.n_outside > 0:
                c.d_bg_dists = w["d_bg_dists"].data_ptr()
        self._bwd_args_keep = (g_color, g_feat, g_weights, g_eik, g_cdf)
        return c

    # ------------------------------------------------------------------------------------------
    def backward(self, g_color, g_feat, g_weights, g_eik, g_cdf=None, g_gradients=None, defer_rest=False, gemm_event=None,
                 fork_event=None, heads_event=True):
        """Upstream grads (any may be None) -> list of parameter grads (clones) per network. g_cdf [B,N] / g_gradients
        [B,N,3]: adjoints of the `cdf_fine` / `gradients` outputs (the reference returns them attached, renderer.py:426-439).
        defer_rest (the Trainer's hot loop): only the SDF network's and the variance's gradients are complete on return (on the
        caller's stream); the caller finishes the others with rest_weight_grads() - on the side stream, behind the background
        network's backward - so that they overlap whatever follows on the main stream.
        Every event recorded on the caller's stream is a marker packet in front of the next kernel of the step's critical chain
        (3 - 4 us each: tools/dev/event_probe.py), so the Trainer passes what it already has: fork_event = an event it recorded
        on the current stream since the forward's last launch (the side stream waits for that one instead of a new one);
        heads_event=False when it will hand side_weight_grads / rest_weight_grads an `after` event recorded later on this stream
        (which then also covers the heads' backward)."""
        r, w, st = self.r, self.w, _stream()
        self._bwd_warm = True           # (render()'s graph plans capture a backward only after one ran eagerly: lazy one-time set-up is done)
        rays_o, rays_d, background_rgb, car, z = self._ctx
        keep = [t.contiguous() if t is not None else None for t in (g_color, g_feat, g_weights, g_eik, g_cdf, g_gradients)]
        g_color, g_feat, g_weights, g_eik, g_cdf, g_gradients = keep
        if (g_cdf is not None or g_gradients is not None) and self._fg_compact:
            raise ValueError("adjoints of cdf_fine / gradients need every inside sample evaluated (skip_far=False)")
        use_vdn = self.wdepth and g_feat is not None
        rg = getattr(self, "_ray_grads", False)
        if getattr(self, "_composite_bwd_done", False):
            self._composite_bwd_done = False          # (forward(fuse_loss=...) ran the compositor's adjoint already)
        elif getattr(self, "_bwd_train", None) is not None:
            fl, self._bwd_train = self._bwd_train, None
            c = self._composite_bwd_args(None, g_feat, None, None, None)
            lib.call("vdn_composite_bwd_train", c, lib.ptr(w["color"]), lib.ptr(fl["true_rgb"]), lib.ptr(fl["g_color"]),
                     lib.ptr(fl["fg_count"] if fl.get("fg_count") is not None else w["fg_active"][1]),
                     float(fl["igr_weight"]), float(fl["grad_scale"]), st)
        else:
            c = self._composite_bwd_args(g_color, g_feat, g_weights, g_eik, g_cdf)
            lib.call("vdn_alpha_composite_bwd", c, st)
        if g_gradients is not None:              # `gradients` is the SDF normal itself: its adjoint joins the alpha / eikonal parts
            w["d_normals"].add_(g_gradients.reshape(self.P, 3))
        if r.n_outside > 0:                      # NeRF backward on the side stream, beside the heads' and the SDF backward
            nb = lib.VdnNerfBwdArgs()
            nb.blob = self.nets["nerf"].img.blobs["bwd"].data_ptr()
            nb.g_density, nb.g_rgb = w["d_bg_density"].data_ptr(), w["d_bg_rgb"].data_ptr()
            nb.g_feat = w["d_bg_feat"].data_ptr() if self.wdepth else None
            nb.save_h, nb.save_hv = w["nf_h"].data_ptr(), w["nf_hv"].data_ptr()
            nb.delta_o, nb.delta_v, nb.delta_head, nb.delta_h = (w[k].data_ptr() for k in ("nf_do", "nf_dv", "nf_dhead", "nf_dh"))
            nb.P = self.Q
            if self._bg_compact:
                nb.active_idx, nb.n_active = w["bg_active"][0].data_ptr(), w["bg_active"][1].data_ptr()
            if rg:
                # samples off the work list enter the loss through exact zeros: their adjoints stay zero
                w["d_bg_pts"].zero_()
                w["d_bg_dirs"].zero_()
                nb.rays_o, nb.rays_d, nb.z, nb.n_per_ray = rays_o.data_ptr(), rays_d.data_ptr(), w["bg_mid"].data_ptr(), self.T
                nb.d_pts, nb.d_dirs = w["d_bg_pts"].data_ptr(), w["d_bg_dirs"].data_ptr()
            if fork_event is not None and self._side is not None:
                self._side.wait_event(fork_event)
            else:
                self._fork()
            lib.call("vdn_nerf_mlp_bwd" + self.sfx, nb, self._side_handle(st))
            # render()'s autograd node (no deferred half): the background network's weight gradients follow its backward on the side
            # stream, beside the heads' and the SDF network's backward on the caller's, instead of waiting in one GEMM launch for
            # all groups behind everything else (the Trainer's schedule, DESIGN.md 3d, minus its deferral past the step's end)
            split_dw = (not defer_rest and self._side is not None and not rg and os.environ.get("VDN_BWD_SPLIT_DW", "1") != "0"
                        and all(k in self.dw_groups for k in ("sdf", "heads", "nerf")))
            if split_dw:
                self.weight_grads("nerf", self._side.cuda_stream)
            self._side_done()

        def rnet_bwd(net, g_out, out, save_h, dout, dh, d_out, module, accumulate):
            b = lib.VdnRenderNetBwdArgs()
            b.blob = self.nets[net].img.blobs["bwd"].data_ptr()
            b.g_out, b.out, b.save_h = g_out.data_ptr(), out.data_ptr(), save_h.data_ptr()
            b.delta_out, b.delta_h = dout.data_ptr(), dh.data_ptr()
            b.d_feat, b.d_normals = w["d_featvec"].data_ptr(), w["d_normals"].data_ptr()
            b.acc_feat, b.acc_normals = int(accumulate), 1
            b.P, b.d_out, b.squeeze_out = self.P, d_out, int(module.squeeze_out)
            if net == "color" and self.dbc:          # d loss / d (VDN output) through the colour network joins the compositor's
                b.d_extra = w["d_vdn"].data_ptr()
            if rg:
                b.rays_d, b.n_per_ray, b.acc_pts = rays_d.data_ptr(), self.N, int(accumulate)
                b.d_pts, b.d_dirs = w["d_pts"].data_ptr(), w["d_dirs"].data_ptr()
            lib.call("vdn_rendernet_bwd" + self.sfx, self._fg(b), st)
        # d_normals already holds the alpha + eikonal parts: the heads add their input gradients into it;
        # d_featvec is overwritten by the first head and accumulated by the second
        rnet_bwd("color", w["d_color"], w["col_out"], w["col_h"], w["col_dout"], w["col_dh"], 3, r.color_network, False)
        # (accumulate flag covers both d_feat and d_normals; d_normals must always accumulate)
        if self.wdepth:
            rnet_bwd("vdn", w["d_vdn"], w["vdn_out"], w["vdn_h"], w["vdn_dout"], w["vdn_dh"], 96, r.depth_network, True)
        if defer_rest and self._side is not None and heads_event:
            self._ev_heads.record(lib.current_stream())      # the heads' deltas (operands of the rest group) are complete

        rb = lib.VdnSdfRbarArgs()
        img = self.nets["sdf"].img
        rb.blob = img.blobs["full"].data_ptr()
        rb.rays_o, rb.rays_d, rb.z, rb.n_per_ray, rb.z_ld = rays_o.data_ptr(), rays_d.data_ptr(), w["mid_z"].data_ptr(), self.N, self.N
        rb.P, rb.scale = self.P, float(r.sdf_network.scale)
        # bf16: the forward did not store softplus'; the chains re-derive it from the saved activations, which (like V) are
        # in units of 1/(100 log2 e): s_from_h = 2 (include/vdn_render.h)
        s_from_h = 2 if self.precision == "bf16" else 0
        s_planes = w["H"] if s_from_h else w["S"]
        rb.g_normals, rb.S, rb.V, rb.UB, rb.EX = w["d_normals"].data_ptr(), s_planes.data_ptr(), w["V"].data_ptr(), w["UB"].data_ptr(), w["EX"].data_ptr()
        rb.s_from_h = s_from_h
        # bf16 without ray gradients: both chains in one feature-split launch (csrc/k_sdf_bwd_split.h: ex_l never leaves the chip;
        # 218 us against 165 + 140 at the steady-state lists, the step -43 us). VDN_SDF_BWD_SPLIT=0: the two kernels
        one_launch = self.precision == "bf16" and not rg and os.environ.get("VDN_SDF_BWD_SPLIT", "1") != "0"
        fb = lib.VdnSdfFbarArgs()
        fb.blob = img.blobs["fbar"].data_ptr()
        fb.g_sdf, fb.g_feat, fb.S, fb.EX, fb.AB = w["d_sdf"].data_ptr(), w["d_featvec"].data_ptr(), s_planes.data_ptr(), w["EX"].data_ptr(), w["AB"].data_ptr()
        fb.P, fb.scale, fb.s_from_h = self.P, float(r.sdf_network.scale), s_from_h
        if rg:
            fb.rays_o, fb.rays_d, fb.z, fb.n_per_ray, fb.z_ld = rays_o.data_ptr(), rays_d.data_ptr(), w["mid_z"].data_ptr(), self.N, self.N
            fb.g_normals, fb.U_pe, fb.acc_pts, fb.d_pts = w["d_normals"].data_ptr(), w["U_pe"].data_ptr(), 1, w["d_pts"].data_ptr()
        # (the one-launch kernel declines batches whose planes exceed its 32-bit buffer offsets: status -10 -> the two kernels)
        if not (one_launch and lib.try_call("vdn_sdf_bwd_split_bf16", self._fg(rb), self._fg(fb), st)):
            lib.call("vdn_sdf_bwd_rbar" + self.sfx, self._fg(rb), st)
            lib.call("vdn_sdf_bwd_fbar" + self.sfx, self._fg(fb), st)

        if defer_rest:
            if rg:
                raise ValueError("defer_rest is the Trainer's path: no ray gradients there")
            self.weight_grads("sdf", st, gemm_event)
            return self._grad_flat
        if r.n_outside > 0 and split_dw:
            self.weight_grads("sdf", st)
            self.weight_grads("heads", st)
            self._join()
        else:
            self._join()
            self._launch_dw()
            lib.call("vdn_dw_finalize", lib.ptr(self.fin_table), self.n_fin, self.fin_max_M, 0, st)
            if self.fin_has_phase1:
                lib.call("vdn_dw_finalize", lib.ptr(self.fin_table), self.n_fin, self.fin_max_M, 1, st)
            if self.n_wn:
                lib.call("vdn_weightnorm_bwd", lib.ptr(self.wn_table), self.n_wn, self.wn_max_rows, st)
        if rg:
            ra = lib.VdnRayAdjointArgs()
            ra.rays_d, ra.mid_z = rays_d.data_ptr(), w["mid_z"].data_ptr()
            ra.d_pts, ra.d_dirs, ra.d_dists, ra.d_dir_cos = (w[k].data_ptr() for k in ("d_pts", "d_dirs", "d_dists", "d_dir_cos"))
            ra.B, ra.N, ra.T = self.B, self.N, self.T
            if r.n_outside > 0:
                ra.bg_mid = w["bg_mid"].data_ptr()
                ra.d_bg_pts, ra.d_bg_dirs, ra.d_bg_dists, ra.d_z_out = (w[k].data_ptr() for k in ("d_bg_pts", "d_bg_dirs", "d_bg_dists", "d_z_out"))
            ra.d_rays_o, ra.d_rays_d, ra.d_z = w["d_rays_o"].data_ptr(), w["d_rays_d"].data_ptr(), w["d_z"].data_ptr()
            lib.call("vdn_ray_adjoint", ra, st)
        # gradients now sit in self._grad_flat (views per parameter in self.grad_views)
        return self._grad_flat

    def outputs_clone(self):
        """A copy of everything the latest forward hands to the caller (one device copy) -> {name: tensor}."""
        a = self._out_arena.clone()
        if self.__dict__.get("_out_strided") is None:
            self._out_strided = [(k, tuple(sh), tuple(torch.empty(sh, device="meta").stride()), o) for k, (o, sh) in self._out_slots.items()]
        # Tensor.set_ on the copy's storage, not as_strided / detach: each output is then a tensor of its own - not a view (autograd
        # refuses in-place ops on the views a multi-output node returns: ADVICE round 5) and with its OWN version counter (views and
        # detach() aliases share their base's: an in-place op on one output would read as a modification of every other one that
        # somebody saved for backward)
        st, base = a.untyped_storage(), a.storage_offset()
        return {k: torch.empty(0, dtype=torch.float32, device=a.device).set_(st, base + o, sh, sd) for k, sh, sd, o in self._out_strided}

    def param_grads(self, clone=True):
        """Per-parameter gradients in renderer._all_parameters() order."""
        if getattr(self, "join_hook", None) is not None:     # the Trainer's deferred half of the backward (side stream)
            self.join_hook()
        if not clone:
            return [self.grad_views[id(p)] for p in self.params]
        # ONE multi-tensor copy into fresh views of one new allocation (a clone per parameter is 67 - 77 copy launches of 3 us
        # each on the step's stream: 250 us of the unchanged runner's 2.1 ms step). The views are new tensors nothing else
        # refers to, so autograd's AccumulateGrad adopts them as .grad without another copy; their allocation belongs to this
        # backward alone - the engine's own buffer is reused by the next step. Every view starts on a 256-byte boundary:
        # torch.optim's multi-tensor kernels fall back to scalar loads on lists with a misaligned member (packed offsets
        # cost the runner's Adam step + 60 us).
        if self.__dict__.get("_clone_slots") is None:
            slots, off = [], 0
            for p in self.params:
                slots.append((tuple(p.shape), tuple(self.grad_views[id(p)].stride()), off))
                off += (p.numel() + 63) // 64 * 64
            self._clone_slots, self._clone_total = slots, off
            self._grad_list = [self.grad_views[id(p)] for p in self.params]
        flat = torch.empty(self._clone_total, dtype=torch.float32, device=self._grad_flat.device)
        out = [flat.as_strided(sh, st, o) for sh, st, o in self._clone_slots]       # (one op per view: half the host time of slice + view)
        torch._foreach_copy_(out, self._grad_list)
        return out
