"""Weight-image plans: how each network's effective weights are re-tiled into the MFMA chunk
streams the kernels consume (csrc/mlp_engine.h), and the device tables that drive
vdn_weightnorm_materialize / vdn_build_images.

A *stream* is the exact sequence of 32-row chunks one kernel walks through; a layer of a stream is
described by a padded input map (kmap: padded k -> source column, -1 = zero), and per chunk a
source matrix, a bias vector and a 32-entry output-row map. Transposed layers (reverse sweeps,
backward chains) swap the strides, nothing else.
"""
import math

import numpy as np
import torch

from . import lib

FMT_F32, FMT_BF16 = 0, 1


def chunk_bytes(kt, fmt=FMT_F32):
    return kt * 4096 + 1024 if fmt == FMT_F32 else kt * 2048 + 1024


def chunk_stride(kt_max, fmt=FMT_F32):
    """Uniform distance between consecutive chunks of a stream (csrc/mlp_engine.h: P::stride): the largest
    chunk rounded up to 4 KiB, so that every wave issues the same number of 1-KiB global_load_lds per chunk."""
    return (chunk_bytes(kt_max, fmt) + 4095) // 4096 * 4096


# largest contraction width (in 32-wide tiles) each kernel's stream is laid out for (must match the kernels)
STREAM_KT_MAX = {"sdf": 9, "full": 9, "fbar": 9, "c2": 9}


def _pad32(n):
    return (n + 31) // 32 * 32


def ident_map(n, pad=None):
    pad = _pad32(n) if pad is None else pad
    m = np.full(pad, -1, np.int32)
    m[:n] = np.arange(n)
    return m


class Layer:
    """One GEMM layer of a stream. chunks: list of chunks; a chunk is either one piece
    (matrix_name, use_bias, rowmap32) covering the whole contraction width, or a list of pieces
    (matrix_name, use_bias, rowmap32, kt_begin, kt_count) when its k-tiles come from several matrices."""

    def __init__(self, kmap, chunks, scale=1.0, transposed=False, bias_scale=1.0, tail=None):
        self.kmap = np.asarray(kmap, np.int32)
        assert len(self.kmap) % 32 == 0
        self.chunks = chunks
        self.scale = float(scale)
        self.bias_scale = float(bias_scale)
        self.transposed = transposed
        # (matrix name, byte offset inside the chunk, floats[, first element, stride]): floats of that effective weight - by
        # default its leading ones (row 0); with (first, stride) every stride-th from `first` on (stride = row length: a column)
        self.tail = tail

    @property
    def kt(self):
        return len(self.kmap) // 32


def dense_layer(name, kmap, nmap, bias=True, scale=1.0, bias_scale=1.0):
    """Forward layer reading matrix `name` [rows, cols]: padded out row n <- source row nmap[n]."""
    nmap = np.asarray(nmap, np.int32)
    assert len(nmap) % 32 == 0
    return Layer(kmap, [(name, bias, nmap[i:i + 32]) for i in range(0, len(nmap), 32)], scale, bias_scale=bias_scale)


def transposed_layer_multi(parts, fwd_kmap, scale=1.0):
    """W^T layer whose contraction (the forward layer's padded OUTPUT order) spans several matrices:
    parts = [(matrix_name, fwd_nmap_part)], concatenated along k in that order."""
    fwd_kmap = np.asarray(fwd_kmap, np.int32)
    kfull = np.concatenate([np.asarray(nm, np.int32) for _, nm in parts])
    chunks = []
    for i in range(0, len(fwd_kmap), 32):
        pieces, kt0 = [], 0
        for name, nm in parts:
            ktc = len(nm) // 32
            pieces.append((name, False, fwd_kmap[i:i + 32], kt0, ktc))
            kt0 += ktc
        chunks.append(pieces)
    return Layer(kfull, chunks, scale, transposed=True)


def transposed_layer(name, fwd_kmap, fwd_nmap, scale=1.0):
    """W^T layer of a forward layer (kmap, nmap): image rows follow the forward layer's padded INPUT
    order, the contraction runs over its padded OUTPUT order."""
    fwd_kmap = np.asarray(fwd_kmap, np.int32)
    return Layer(np.asarray(fwd_nmap, np.int32),
                 [(name, False, fwd_kmap[i:i + 32]) for i in range(0, len(fwd_kmap), 32)], scale, transposed=True)


class NetImages:
    """Device-side state of one network: effective weights + chunk-stream blobs.

    matrices: ordered {name: (g or None, v, bias or None)} of torch Parameters.
    streams:  {stream_name: [Layer, ...]}.
    """

    def __init__(self, matrices, streams, device, fmt=FMT_F32):
        self.device = device
        self.fmt = fmt
        self.matrices = matrices
        self.streams = streams
        self._key = None
        # counts every (re)build and every invalidation of the images: what "the weights changed" means to consumers that cache
        # something derived from them (fields.py: the NeRF scratch buffer). torch's version counters are not enough - the Trainer's
        # fused Adam writes parameters through raw pointers and refresh_together() then stores the SAME version tuple again.
        self.builds = 0
        names = list(matrices)
        self.names = names
        sizes = [matrices[n][1].numel() for n in names]
        rows = [matrices[n][1].shape[0] for n in names]
        self.w_off = dict(zip(names, np.concatenate([[0], np.cumsum(sizes)[:-1]]).tolist()))
        self.r_off = dict(zip(names, np.concatenate([[0], np.cumsum(rows)[:-1]]).tolist()))
        self.weff = torch.empty(int(sum(sizes)), dtype=torch.float32, device=device)
        self.inv_norm = torch.empty(int(sum(rows)), dtype=torch.float32, device=device)
        self.max_rows = max(rows)
        # blobs
        self.blobs, self.blob_off = {}, {}
        maps, map_off = [], 0
        chunk_rows = []
        for sname, layers in streams.items():
            if sname.startswith("_") or (sname == "c2" and fmt != FMT_BF16):     # ("c2" is the bf16 fused kernel's format)
                continue
            kt_max = max(L.kt for L in layers)
            kt_max = max(kt_max, STREAM_KT_MAX.get(sname, 0))
            stride = chunk_stride(kt_max, fmt)
            total = stride * sum(len(L.chunks) for L in layers)
            self.blobs[sname] = torch.zeros(total, dtype=torch.uint8, device=device)
            off = 0
            for L in layers:
                k_off = map_off
                maps.append(L.kmap)
                map_off += len(L.kmap)
                for chunk in L.chunks:
                    pieces = chunk if isinstance(chunk, list) else [chunk + (0, 0)]
                    for pi, (mname, use_bias, rowmap, kt0, ktc) in enumerate(pieces):
                        n_off = map_off
                        maps.append(np.asarray(rowmap, np.int32))
                        map_off += 32
                        chunk_rows.append((sname, off, mname, use_bias, k_off + 32 * kt0, n_off, L, kt0, ktc, pi == 0))
                    off += stride
        self.maps = torch.from_numpy(np.concatenate(maps)).to(device)
        self._chunk_rows = chunk_rows
        self._tables_key = None
        self.wn_table = None
        self.chunk_table = None

    def weff_view(self, name):
        v = self.matrices[name][1]
        o = self.w_off[name]
        return self.weff[o:o + v.numel()].view(v.shape)

    def _build_tables(self):
        wn = np.zeros(len(self.names), dtype=lib.struct_dtype("VdnWeightNormDesc"))
        for i, n in enumerate(self.names):
            g, v, b = self.matrices[n]
            wn[i]["g"] = 0 if g is None else g.data_ptr()
            wn[i]["v"] = v.data_ptr()
            wn[i]["w_eff"] = self.weff.data_ptr() + 4 * self.w_off[n]
            wn[i]["inv_norm"] = self.inv_norm.data_ptr() + 4 * self.r_off[n]
            wn[i]["rows"], wn[i]["cols"] = v.shape
        ch = np.zeros(len(self._chunk_rows), dtype=lib.struct_dtype("VdnChunkDesc"))
        mp = self.maps.data_ptr()
        for i, (sname, off, mname, use_bias, k_off, n_off, L, kt0, ktc, first) in enumerate(self._chunk_rows):
            g, v, b = self.matrices[mname]
            rows, cols = v.shape
            ch[i]["src"] = self.weff.data_ptr() + 4 * self.w_off[mname]
            ch[i]["bias"] = b.data_ptr() if (use_bias and b is not None) else 0
            ch[i]["kmap"] = mp + 4 * k_off
            ch[i]["nmap"] = mp + 4 * n_off
            ch[i]["dst"] = self.blobs[sname].data_ptr() + off
            if L.transposed:      # image row index -> source COLUMN, k index -> source ROW
                ch[i]["row_stride"], ch[i]["col_stride"] = 1, cols
            else:
                ch[i]["row_stride"], ch[i]["col_stride"] = cols, 1
            ch[i]["n0"] = 0
            ch[i]["k_pad"] = len(L.kmap)
            ch[i]["scale"] = L.scale
            ch[i]["bias_scale"] = L.bias_scale
            if L.tail is not None:
                tname, toff, tn = L.tail[:3]
                tfirst, tstride = (L.tail[3], L.tail[4]) if len(L.tail) > 3 else (0, 1)
                ch[i]["tail"], ch[i]["tail_off"], ch[i]["tail_n"] = self.weff.data_ptr() + 4 * (self.w_off[tname] + tfirst), toff, tn
                ch[i]["tail_stride"] = tstride
            ch[i]["fmt"] = self.fmt
            ch[i]["kt_begin"], ch[i]["kt_count"], ch[i]["write_bias"] = kt0, ktc, int(first)
        self.wn_table = torch.from_numpy(wn.view(np.uint8)).to(self.device)
        self.chunk_table = torch.from_numpy(ch.view(np.uint8)).to(self.device)
        self._n_wn, self._n_ch = len(wn), len(ch)

    def invalidate(self):
        """Parameters were written outside torch's version tracking (fused Adam through raw pointers)."""
        self._key = None
        self.builds += 1

    def _params(self):
        return [t for n in self.names for t in self.matrices[n] if t is not None]

    def _ensure_tables(self):
        ptr_key = tuple(t.data_ptr() for t in self._params())
        if ptr_key != self._tables_key:
            self._build_tables()
            self._tables_key = ptr_key
            self._key = None

    def stale(self):
        self._ensure_tables()
        return tuple(t._version for t in self._params()) != self._key

    def refresh(self, stream):
        """Re-materialise W_eff and all chunk images if any parameter changed (in place or rebound)."""
        if _TRUSTED is not None and id(self) in _TRUSTED:
            return False
        self._ensure_tables()
        key = tuple(t._version for t in self._params())
        if key == self._key:
            return False
        lib.call("vdn_weightnorm_materialize", lib.ptr(self.wn_table), self._n_wn, self.max_rows, stream)
        lib.call("vdn_build_images", lib.ptr(self.chunk_table), self._n_ch, stream)
        self._key = key
        self.builds += 1
        return True


_TRUSTED = None


def trust(images):
    """render() under grad has just tested / refreshed these image sets and changes no parameter until it returns: their
    refresh() is a no-op until trust(None)."""
    global _TRUSTED
    _TRUSTED = None if images is None else {id(im) for im in images}


def refresh_together(images, stream, cache):
    """Rebuild the weight images of several networks with ONE weight-norm launch and ONE image-build launch (the
    descriptor tables are self-contained, so they concatenate). Used after the fused Adam step, which changes every
    network at once: 2 launches instead of 2 per network. `cache` (a dict) keeps the concatenated tables."""
    for im in images:
        im._ensure_tables()
    key = tuple(im.wn_table.data_ptr() for im in images)
    if cache.get("key") != key:
        cache["key"] = key
        cache["wn"] = torch.cat([im.wn_table for im in images])
        cache["ch"] = torch.cat([im.chunk_table for im in images])
        cache["n_wn"], cache["n_ch"] = sum(im._n_wn for im in images), sum(im._n_ch for im in images)
        cache["max_rows"] = max(im.max_rows for im in images)
    lib.call("vdn_weightnorm_materialize", lib.ptr(cache["wn"]), cache["n_wn"], cache["max_rows"], stream)
    lib.call("vdn_build_images", lib.ptr(cache["ch"]), cache["n_ch"], stream)
    for im in images:
        im._key = tuple(t._version for t in im._params())
        im.builds += 1


# ---------------------------------------------------------------------------------------------
# plans for the shipped shape family
# ---------------------------------------------------------------------------------------------

SDF_UNIT = 100.0 * math.log2(math.e)      # the bf16 SDF kernel computes in units of 1 / (100 log2 e)  (csrc/k_sdf_fwd2.h)
SDF_TAIL_OFF = 9 * 2048 + 1024            # first byte behind the widest (9 k-tile) chunk's bias block: 1 KiB free in the 20-KiB stride


def sdf_streams(d_in, d_out, d_hidden, n_layers, skip_in, multires, scaled=False):
    """Streams of the SDF kernels: 'sdf' (mode 0), 'full' (mode 1: forward + reverse sweep), 'fbar' (adjoint chain).
    scaled=True (bf16 path, csrc/k_sdf_fwd2.h): activations are carried as g = 100 log2(e) h, so hidden-layer weights
    stay as they are while their biases are scaled by 100 log2 e, the last layer's weights by 1 / (100 log2 e), and the
    sweep's transposed weights by 1/255 (it multiplies by 255 sigma). 'fbar' is never scaled."""
    if not (d_in == 3 and d_hidden == 256 and n_layers == 8 and tuple(skip_in) == (4,) and multires == 6 and d_out == 257):
        raise ValueError("SDFNetwork: the HIP kernels implement the shipped shape family only "
                         "(d_in=3, d_out=257, d_hidden=256, n_layers=8, skip_in=(4,), multires=6); got "
                         "d_in=%d d_out=%d d_hidden=%d n_layers=%d skip_in=%s multires=%d"
                         % (d_in, d_out, d_hidden, n_layers, tuple(skip_in), multires))
    d0 = 39
    fwd = []   # (name, kmap, nmap, scale) per layer 0..7
    for l in range(8):
        name = "lin%d" % l
        if l == 0:
            km, nm, sc = ident_map(d0, 64), ident_map(256), 1.0
        elif l == 3:
            km, nm, sc = ident_map(256), ident_map(256 - d0, 224), 1.0
        elif l == 4:
            km = np.full(288, -1, np.int32)
            km[:256 - d0] = np.arange(256 - d0)                    # h4 (217)
            km[224:224 + d0] = (256 - d0) + np.arange(d0)          # PE (39)
            nm, sc = ident_map(256), 1.0 / math.sqrt(2.0)
        else:
            km, nm, sc = ident_map(256), ident_map(256), 1.0
        fwd.append((name, km, nm, sc))
    if scaled:
        # the 25 spare contraction slots behind the 39 encoded inputs (layer 0: 39..63; layer 4: 263..287) read the weight
        # columns of the first 25 encoded values once more: the kernel feeds them the bf16 rounding residue of those inputs
        n0, km0, nm0, sc0 = fwd[0]
        km0 = km0.copy()
        km0[d0:d0 + 25] = np.arange(25)
        fwd[0] = (n0, km0, nm0, sc0)
        n4, km4, nm4, sc4 = fwd[4]
        km4 = km4.copy()
        km4[224 + d0:224 + d0 + 25] = (256 - d0) + np.arange(25)
        fwd[4] = (n4, km4, nm4, sc4)
    bsc = SDF_UNIT if scaled else 1.0
    wsc8 = 1.0 / SDF_UNIT if scaled else 1.0
    swsc = 1.0 / 255.0 if scaled else 1.0
    hidden = [dense_layer(n, km, nm, True, sc, bias_scale=bsc) for (n, km, nm, sc) in fwd]
    last_sdf = dense_layer("lin8", ident_map(256), ident_map(1, 32), True, wsc8)
    nm8 = np.full(288, -1, np.int32)
    nm8[:256] = 1 + np.arange(256)      # feature rows
    nm8[256] = 0                        # sdf row
    last_full = dense_layer("lin8", ident_map(256), nm8, True, wsc8)
    sweep = [transposed_layer(n, km, nm, sc * swsc) for (n, km, nm, sc) in reversed(fwd[0:8])]
    # adjoint of the forward pass: W8^T, W7^T .. W1^T
    # (+ W0^T at the end: the adjoint of the encoded input, only read when the rays are differentiable; unscaled kmap)
    fbar = [transposed_layer("lin8", ident_map(256), nm8)] + [transposed_layer(n, km, nm, sc) for (n, km, nm, sc) in reversed(fwd[1:8])]
    fbar.append(transposed_layer("lin0", ident_map(d0, 64), ident_map(256), 1.0))
    if scaled:
        # row 0 of W8 (f32, 256 values) rides in every chunk's unused tail, behind the widest chunk's bias block: the kernel
        # reads it from whichever slot is current (the f32 sdf row, the sweep's first operand)
        for L in hidden + [last_sdf, last_full] + sweep:
            L.tail = ("lin8", SDF_TAIL_OFF, 256)
    return {"sdf": hidden + [last_sdf], "full": hidden + [last_full] + sweep, "fbar": fbar}


def sdf_layer_maps():
    """(name, kmap, nmap, scale) of the 9 SDF layers in image coordinates (for the weight-gradient scatter)."""
    st = sdf_streams(3, 257, 256, 8, (4,), 6)
    out = []
    for l, L in enumerate(st["full"][:9]):
        nmap = np.concatenate([c[2] for c in L.chunks])
        out.append(("lin%d" % l, L.kmap, nmap, L.scale))
    return out


def rendering_streams(d_feature, mode, d_in, d_out, d_hidden, n_layers, multires_view):
    """The kernels assemble [feature(256) | points(3), PE4(view)(27), normals(3)] (slots 0..288). The reference's three
    modes (fields.py:154-158) are column selections of that input: a slot the mode leaves out maps to no weight column
    (-1: a zero row of the image, no weight gradient, a zero input adjoint) - 'no_normal' = [points, view, feature] needs
    multires_view = 4; 'no_view_dir' = [points, normals, feature] only exists with multires_view = 0 in the reference
    (with an encoder its first layer is sized for view columns it is never given, fields.py:132-135 vs 156)."""
    family = {("idr", 9, 4), ("no_normal", 6, 4), ("no_view_dir", 6, 0)}
    # d_feature = 352: the colour network of render(depth_before_color=True), fed cat([feature_vector, VDN output])
    # (renderer.py:247-248): 3 more input tiles behind the standard 10, weight columns behind the 256 feature columns
    if not (d_feature in (256, 352) and (mode, d_in, multires_view) in family and d_hidden == 256 and n_layers == 4 and
            (d_out == 96 or 1 <= d_out <= 4) and not (d_feature == 352 and d_out == 96)):
        raise ValueError("RenderingNetwork: the HIP kernels implement d_feature in {256, 352 (d_out <= 4)}, d_hidden=256, n_layers=4, "
                         "d_out in {1..4, 96} with (mode, d_in, multires_view) in %s only" % sorted(family))
    km0 = np.full(320 if d_feature == 256 else 416, -1, np.int32)
    if d_feature == 352:
        km0[320:416] = ((d_in - 3 + 27) if multires_view else d_in) + 256 + np.arange(96)
    if mode == "idr":
        km0[:256] = 33 + np.arange(256)      # feature vector columns
        km0[256:289] = np.arange(33)         # points(3), PE(view)(27), normals(3)
    elif mode == "no_normal":
        km0[:256] = 30 + np.arange(256)
        km0[256:286] = np.arange(30)         # points(3), PE(view)(27); the normals' slots read nothing
    else:
        km0[:256] = 6 + np.arange(256)
        km0[256:259] = np.arange(3)          # points(3)
        km0[286:289] = 3 + np.arange(3)      # normals(3); the encoded view's slots read nothing
    layers = [dense_layer("lin0", km0, ident_map(256))]
    for l in (1, 2, 3):
        layers.append(dense_layer("lin%d" % l, ident_map(256), ident_map(256)))
    nm4 = ident_map(d_out, 96 if d_out == 96 else 32)
    layers.append(dense_layer("lin4", ident_map(256), nm4))
    bwd = [transposed_layer("lin4", ident_map(256), nm4)]
    bwd += [transposed_layer("lin%d" % l, ident_map(256), ident_map(256)) for l in (3, 2, 1)]
    bwd.append(transposed_layer("lin0", km0, ident_map(256)))
    st = {"fwd": layers, "bwd": bwd}
    if d_feature == 256 and mode == "idr" and 1 <= d_out <= 4:
        # "c2": the colour head behind the fused SDF kernel (csrc/k_sdf_fwd2.h MODE 2), in that kernel's chunk format (9 k-tiles,
        # 20-KiB stride): the first layer contracts [feature (256) | points, PE(view), normal x, y (32)]; the 33rd small input - the
        # normal's z component - is an f32 rank-1 term in the kernel's epilogue, its weight column (source column 32 of lin0) rides
        # in every chunk's tail like row 0 of W8 does in the SDF streams
        c2 = [dense_layer("lin0", np.concatenate([km0[:256], km0[256:288]]), ident_map(256))]
        c2 += [dense_layer("lin%d" % l, ident_map(256), ident_map(256)) for l in (1, 2, 3)]
        c2.append(dense_layer("lin4", ident_map(256), nm4))
        n_in = 33 + 256                       # row length of lin0's effective weight
        for L in c2:
            L.tail = ("lin0", SDF_TAIL_OFF, 256, int(km0[288]), n_in)
        st["c2"] = c2
    return st


def nerf_streams(D, W, d_in, d_in_view, multires, multires_view, skips, rgb_dims, gen_depth_feats, dpt_dim):
    if not (D == 8 and W == 256 and d_in == 4 and d_in_view == 3 and multires == 10 and multires_view == 4 and
            tuple(skips) == (4,) and rgb_dims == 3 and (not gen_depth_feats or dpt_dim == 96)):
        raise ValueError("NeRF: the HIP kernels implement D=8, W=256, d_in=4, d_in_view=3, multires=10, "
                         "multires_view=4, skips=[4], rgb_dims=3, dpt_dim=96 only")
    ch = 84
    layers = [dense_layer("pts_linears.0", ident_map(ch, 96), ident_map(256))]
    for i in (1, 2, 3, 4):
        layers.append(dense_layer("pts_linears.%d" % i, ident_map(256), ident_map(256)))
    km5 = np.full(352, -1, np.int32)
    km5[:ch] = np.arange(ch)
    km5[96:352] = ch + np.arange(256)
    layers.append(dense_layer("pts_linears.5", km5, ident_map(256)))
    for i in (6, 7):
        layers.append(dense_layer("pts_linears.%d" % i, ident_map(256), ident_map(256)))
    head = dense_layer("feature_linear", ident_map(256), ident_map(256))
    head.chunks.append(("alpha_linear", True, ident_map(1, 32)))
    layers.append(head)
    kmv = np.full(288, -1, np.int32)
    kmv[:256] = np.arange(256)
    kmv[256:283] = 256 + np.arange(27)
    layers.append(dense_layer("views_linears.0", kmv, ident_map(128)))
    out = dense_layer("rgb_linear", ident_map(128), ident_map(3, 32))
    if gen_depth_feats:
        nm = ident_map(96)
        out.chunks += [("dpt_linear", True, nm[i:i + 32]) for i in (0, 32, 64)]
    layers.append(out)
    # adjoint chain: Wout^T, Wviews^T, Whead^T, W7^T .. W1^T
    out_parts = [("rgb_linear", ident_map(3, 32))] + ([("dpt_linear", ident_map(96))] if gen_depth_feats else [])
    bwd = [transposed_layer_multi(out_parts, ident_map(128))]
    bwd.append(transposed_layer("views_linears.0", kmv, ident_map(128)))
    bwd.append(transposed_layer_multi([("feature_linear", ident_map(256)), ("alpha_linear", ident_map(1, 32))], ident_map(256)))
    bwd += [transposed_layer("pts_linears.%d" % i, ident_map(256), ident_map(256)) for i in (7, 6)]
    bwd.append(transposed_layer("pts_linears.5", km5, ident_map(256)))
    bwd += [transposed_layer("pts_linears.%d" % i, ident_map(256), ident_map(256)) for i in (4, 3, 2, 1)]
    # W0^T: the adjoint of the encoded point, only read when the rays are differentiable (vdn_nerf_mlp_bwd with d_pts)
    bwd.append(transposed_layer("pts_linears.0", ident_map(ch, 96), ident_map(256)))
    return {"fwd": layers, "bwd": bwd, "_km5": km5, "_kmv": kmv}
