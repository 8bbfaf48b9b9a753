"""The weight-gradient GEMM of the bf16 path, alone, through the C ABI (include/vdn_render.h: VdnDwDesc): planes in the
tile-blocked bf16 layout, one or two segments, ragged row counts (partial 32-row blocks, device-side row limits, splits that
get no rows), narrow and over-wide operands - against an fp32 matmul of the same bf16-rounded operands."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "vdn-nerf_amd"))
pytestmark = pytest.mark.gpu


def _plane(torch, layout, P, ld, seed, dev, garbage_pad=True):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(P, ld, generator=g).to(dev)
    buf = layout.to_pt32(x)
    if garbage_pad and P % 32:
        # rows beyond P are padding the producers leave undefined: fill them with huge values so a kernel that
        # contracts over them cannot pass
        full = layout.from_pt32(buf, layout.pad32(P), ld)
        full[P:] = 3.0e38
        buf = layout.to_pt32(full)
    return x.to(torch.bfloat16).float(), buf


CASES = [
    # (P, P_dev, [(m_cols, n_cols, two_segments, splits)])
    (4096, None, [(256, 256, False, 1)]),
    (5000, None, [(256, 256, True, 4), (256, 64, False, 2), (32, 256, False, 2), (288, 256, False, 2)]),
    (8192 + 77, 6000 + 13, [(256, 256, True, 6), (128, 288, False, 3), (256, 0, False, 3), (96, 128, False, 3)]),
    (300, 31, [(256, 256, False, 2), (64, 96, True, 2)]),
]


@pytest.mark.parametrize("case", range(len(CASES)))
def test_dw_gemm_bf16_vs_matmul(case):
    import torch
    from vdn_hip import layout, lib
    dev = torch.device("cuda:0")
    P, P_dev, entries = CASES[case]
    rows = P if P_dev is None else min(P, P_dev)
    pdev = None if P_dev is None else torch.tensor([P_dev], dtype=torch.int32, device=dev)
    dw = np.zeros(len(entries), dtype=lib.struct_dtype("VdnDwDesc"))
    keep, expect, wg = [], [], 0
    for i, (M, N, two, splits) in enumerate(entries):
        mt, nt = M // 32, N // 32
        d = dw[i]
        segs = []
        for s in range(2 if two else 1):
            a, abuf = _plane(torch, layout, P, M, 100 * case + 10 * i + s, dev)
            d["A%d" % (s + 1)], d["lda%d" % (s + 1)] = abuf.data_ptr(), M
            keep.append(abuf)
            if nt:
                b, bbuf = _plane(torch, layout, P, N, 100 * case + 10 * i + s + 5, dev)
                d["B%d" % (s + 1)], d["ldb%d" % (s + 1)] = bbuf.data_ptr(), N
                keep.append(bbuf)
            else:
                b = None
            segs.append((a, b))
        slab = torch.full((splits, M, max(N, 1)), float("nan"), device=dev)
        colsum = torch.full((splits, M), float("nan"), device=dev)
        keep += [slab, colsum]
        d["P"], d["m_tiles"], d["n_tiles"], d["splits"], d["wg_begin"] = P, mt, nt, splits, wg
        d["slab"], d["colsum"] = slab.data_ptr(), colsum.data_ptr()
        if pdev is not None:
            d["P_dev"] = pdev.data_ptr()
        wg += lib.call_value("vdn_dw_entry_wgs_bf16", mt, nt, splits)
        want = sum(a[:rows].T @ b[:rows] for a, b in segs) if nt else None
        expect.append((slab, colsum, want, segs[0][0][:rows].sum(0), nt))
    table = torch.from_numpy(dw.view(np.uint8)).to(dev)
    lib.call("vdn_dw_gemm_bf16", lib.ptr(table), len(entries), wg, torch.cuda.current_stream().cuda_stream)
    torch.cuda.synchronize()
    for slab, colsum, want, want_cs, nt in expect:
        if nt:
            got = slab.sum(0)
            assert torch.isfinite(got).all()
            err = (got - want).abs().max().item()
            assert err <= 2e-3 * max(1.0, want.abs().max().item()), err
        cs = colsum.sum(0)
        assert torch.isfinite(cs).all()
        assert (cs - want_cs).abs().max().item() <= 2e-3 * max(1.0, want_cs.abs().max().item())
