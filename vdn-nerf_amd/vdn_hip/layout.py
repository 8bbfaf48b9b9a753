"""Activation-plane layouts (see csrc/mlp_engine.h).

fp32 path: row-major [P, ld].  bf16 path: tile-blocked "PT32": points padded to a multiple of 32; per block and 32-feature
tile [k(2)][h(2)][point(32)][8 features], the 16-byte unit (k, h, point) holding features 16k + 4h + {0..3} and
16k + 8 + 4h + {0..3}: element (p, f) at
(p>>5)*(32*ld) + (f>>5)*1024 + ((f>>4)&1)*512 + ((f>>2)&1)*256 + (p&31)*8 + ((f>>3)&1)*4 + (f&3).
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
    # feature within a tile = 16 k + 8 j + 4 hh + e:  [blk, c, nt, k, j, hh, e] -> [blk, nt, k, hh, c, j, e]
    return buf.view(Pp // 32, 32, ld // 32, 2, 2, 2, 4).permute(0, 2, 3, 5, 1, 4, 6).contiguous().view(-1)


def from_pt32(buf, P, ld):
    """bf16 PT32 buffer -> [P, ld] float32."""
    Pp = pad32(P)
    # [blk, nt, k, hh, c, j, e] -> [blk, c, nt, k, j, hh, e]
    x = buf.view(-1)[:Pp * ld].view(Pp // 32, ld // 32, 2, 2, 32, 2, 4).permute(0, 4, 1, 2, 5, 3, 6).contiguous().view(Pp, ld)
    return x[:P].float()


def col_offset_elems(col0, precision):
    """Element offset of column col0 (a multiple of 32) inside a plane."""
    if precision == "fp32":
        return col0
    assert col0 % 32 == 0
    return (col0 // 32) * 1024
