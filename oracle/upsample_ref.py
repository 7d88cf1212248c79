"""Bilinear align_corners=True oracle (numpy, float64).  Test infrastructure only.

Restates what the reference's F.interpolate(..., mode='bilinear',
align_corners=True) call sites compute (bisenet network.py:82-84,93-94,164-166):
src = dst*(in-1)/(out-1) (0 when out==1); i0 = floor(src); i1 = min(i0+1, in-1);
lambda = src - i0; 4-tap lerp.  The operator is linear, so forward and backward
are the dense matrices W_y (OH x IH), W_x (OW x IW) and their transposes.
Pinned against torch's CPU F.interpolate in tests/test_oracles_cpu.py."""
import numpy as np


def interp_matrix(n_in, n_out):
    """W [n_out, n_in] with W @ v == 1-D linear resize of v (align_corners=True).
    The scale is formed in float32 like ATen's area_pixel_compute_scale."""
    W = np.zeros((n_out, n_in), dtype=np.float64)
    scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
    for o in range(n_out):
        r = np.float32(scale * np.float32(o))
        i0 = min(int(r), n_in - 1)
        i1 = i0 + (1 if i0 < n_in - 1 else 0)
        l1 = float(np.float32(r - np.float32(i0)))
        W[o, i0] += 1.0 - l1
        W[o, i1] += l1
    return W


def upsample_bilinear_ac(x, OH, OW):
    """x [..., IH, IW] -> [..., OH, OW]"""
    Wy, Wx = interp_matrix(x.shape[-2], OH), interp_matrix(x.shape[-1], OW)
    return np.einsum("oi,...ij,pj->...op", Wy, x.astype(np.float64), Wx)


def upsample_bilinear_ac_backward(dy, IH, IW):
    """dy [..., OH, OW] -> dx [..., IH, IW] (transposed operator)"""
    Wy, Wx = interp_matrix(IH, dy.shape[-2]), interp_matrix(IW, dy.shape[-1])
    return np.einsum("oi,...op,pj->...ij", Wy, dy.astype(np.float64), Wx)


def upsample_nearest(x, OH, OW):
    """torch 'nearest': src = floor(dst * in/out) (float32 scale)"""
    IH, IW = x.shape[-2], x.shape[-1]
    iy = np.minimum((np.arange(OH, dtype=np.float32) * (np.float32(IH) / np.float32(OH))).astype(np.int64), IH - 1)
    ix = np.minimum((np.arange(OW, dtype=np.float32) * (np.float32(IW) / np.float32(OW))).astype(np.int64), IW - 1)
    return x[..., iy[:, None], ix[None, :]]
