"""Oracle (test infrastructure): dense nearest-neighbour correspondence.

The reference delegates this to the `KNN` layer of NEC-Labs' Caffe-UCN fork (not vendored,
no version pin; call site ucn_code/nets/train.prototxt:2116-2141, consumer
ucn_code/lib/layers/hard_negative_processing_layer.py:175-192).  Nothing in the reference
holds known answers for it, so PARITY IS UNPINNED; the contract restated here is:

  * query  (C, Nq) f32 and ref (C, Nr) f32 are one image's slice of NCHW feature maps with the
    spatial axes flattened row-major (flat index = y*W + x);
  * squared L2 distance d2 = (|q|^2 + |r|^2) - 2 q.r, every sum accumulated over the channel
    axis in increasing c in float64 (products of two float32 are exact in float64);
  * the k nearest refs per query ordered by (d2, index) ascending, ties to the lowest index;
  * indices are 1-based by default (the consumer subtracts 1), outputs shaped (k, Nq).

`nn_exact_numpy` is the plain restatement; oracle.cnative.nn_exact is the same thing in C
(bit-identical, checked by the CPU tests) for sizes where numpy is too slow.
`nn_sgemm` is the float32 BLAS formulation that a numpy user of the reference would run;
it is only used as the timed CPU baseline.
"""
import numpy as np


def nn_exact_numpy(q, r, k=1, index_base=1):
    q64 = np.asarray(q, dtype=np.float32).astype(np.float64)
    r64 = np.asarray(r, dtype=np.float32).astype(np.float64)
    c, nq = q64.shape
    nr = r64.shape[1]
    dot = np.zeros((nq, nr))
    qq = np.zeros(nq)
    rr = np.zeros(nr)
    for ch in range(c):                       # fixed accumulation order over channels
        dot += np.multiply.outer(q64[ch], r64[ch])
        qq += q64[ch] * q64[ch]
        rr += r64[ch] * r64[ch]
    d2 = (qq[:, None] + rr[None, :]) - 2.0 * dot
    order = np.argsort(d2, axis=1, kind="stable")[:, :k]      # stable: ties -> lowest index
    dist = np.take_along_axis(d2, order, axis=1)
    return (order.T + index_base).astype(np.int32), np.ascontiguousarray(dist.T)


def nn_sgemm(q, r, k=1, index_base=1):
    """float32 |a|^2+|b|^2-2ab through BLAS sgemm + argmin/argpartition (CPU baseline only)."""
    q = np.asarray(q, dtype=np.float32)
    r = np.asarray(r, dtype=np.float32)
    d2 = (q * q).sum(0)[:, None] + (r * r).sum(0)[None, :] - 2.0 * (q.T @ r)
    if k == 1:
        idx = np.argmin(d2, axis=1)[None, :]
    else:
        part = np.argpartition(d2, k - 1, axis=1)[:, :k]
        sub = np.take_along_axis(d2, part, axis=1)
        idx = np.take_along_axis(part, np.argsort(sub, axis=1, kind="stable"), axis=1).T
    dist = np.take_along_axis(d2, idx.T, axis=1).T
    return (idx + index_base).astype(np.int32), dist


def extract_feature(feature, coords, downsampling=4.0, coord_start_ind=0, mode="nearest"):
    """`ExtractFeature` layer contract (ucn_code/nets/train.prototxt:2116-2127; external Caffe-UCN
    layer, PARITY UNPINNED): sample feature (B,C,H,W) f32 at the image coordinates held in columns
    coord_start_ind, coord_start_ind+1 (x, y) of coords (B,n,ncol), divided by the downsampling
    factor as hard_negative_processing_layer.py:33-37 does.  mode "nearest": cell (floor(x), floor(y))
    clamped to the map; "bilinear": float32 interpolation of the four surrounding cells (clamped),
    value = (1-wy)*((1-wx)*f00 + wx*f01) + wy*((1-wx)*f10 + wx*f11).  Returns (B,C,n) f32."""
    f = np.asarray(feature, dtype=np.float32)
    c = np.asarray(coords, dtype=np.float32)
    b, ch, h, w = f.shape
    ds = np.float32(downsampling)
    fx = c[:, :, coord_start_ind] / ds
    fy = c[:, :, coord_start_ind + 1] / ds
    x0f, y0f = np.floor(fx), np.floor(fy)
    x0 = np.clip(x0f, 0, w - 1).astype(np.int64)
    y0 = np.clip(y0f, 0, h - 1).astype(np.int64)
    bi = np.arange(b)[:, None]
    if mode == "nearest":
        return np.ascontiguousarray(f[bi, :, y0, x0].transpose(0, 2, 1))
    x1 = np.clip(x0f + 1, 0, w - 1).astype(np.int64)
    y1 = np.clip(y0f + 1, 0, h - 1).astype(np.int64)
    wx = (fx - x0f).astype(np.float32)[:, :, None]
    wy = (fy - y0f).astype(np.float32)[:, :, None]
    one = np.float32(1.0)
    f00, f01 = f[bi, :, y0, x0], f[bi, :, y0, x1]
    f10, f11 = f[bi, :, y1, x0], f[bi, :, y1, x1]
    top = (one - wx) * f00 + wx * f01
    bot = (one - wx) * f10 + wx * f11
    out = (one - wy) * top + wy * bot
    return np.ascontiguousarray(out.astype(np.float32).transpose(0, 2, 1))
