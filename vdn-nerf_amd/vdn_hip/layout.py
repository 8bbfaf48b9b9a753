"""Activation-plane layouts (see csrc/mlp_engine.h).

fp32 path: row-major [P, ld].  bf16 path: tile-blocked "PT32": points padded to a multiple of 32;
element (p, f) at (p>>5)*(32*ld) + (f>>5)*1024 + ((f&31)>>3)*256 + ((f&7)>>2)*128 + (p&31)*4 + (f&3).
These converters serve the stand-alone module API (SDFNetwork.forward / RenderingNetwork.forward), where
the caller hands over / expects row-major tensors; the render path never converts."""
import torch


def pad32(P):
    return (P + 31) // 32 * 32


def rows(P, precision):
    return P if precision == "fp32" else pad32(P)


def to_pt32(x):
    """[P, ld] float -> bf16 PT32 buffer [pad32(P) * ld]."""
    P, ld = x.shape
    Pp = pad32(P)
    buf = torch.zeros(Pp, ld, dtype=torch.bfloat16, device=x.device)
    buf[:P] = x.to(torch.bfloat16)
    # [blk, c, nt, q, hh, e] -> [blk, nt, q, hh, c, e]
    return buf.view(Pp // 32, 32, ld // 32, 4, 2, 4).permute(0, 2, 3, 4, 1, 5).contiguous().view(-1)


def from_pt32(buf, P, ld):
    """bf16 PT32 buffer -> [P, ld] float32."""
    Pp = pad32(P)
    x = buf.view(-1)[:Pp * ld].view(Pp // 32, ld // 32, 4, 2, 32, 4).permute(0, 4, 1, 2, 3, 5).contiguous().view(Pp, ld)
    return x[:P].float()


def col_offset_elems(col0, precision):
    """Element offset of column col0 (a multiple of 32) inside a plane."""
    if precision == "fp32":
        return col0
    assert col0 % 32 == 0
    return (col0 // 32) * 1024
