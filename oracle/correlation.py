"""Oracle (test infrastructure): correlation maps and feature normalisation.

Restates
  classifier_code/train_test_scripts/arch/eccv_net/eccv_net.py:181-231   (pinned by golden
      vectors produced by the imported reference layer on CPU)
  classifier_code/train_test_scripts/arch/acmm_net/net_final_old.py:33-61 (restated; its
      forward calls .cuda(), so it cannot run here; cross-checked against the eccv oracle,
      which shares the contraction with the operands swapped)
  ucn_code/lib/layers/l2norm_layer.py:18-29
All arithmetic in float32 like the reference; summation order differs from torch's, so
comparisons carry a 1e-5 relative tolerance.
"""
import numpy as np


def l2norm(x):
    """l2norm_layer.py:18-29: x / sqrt(sum_c x^2) over axis 1, no epsilon."""
    x = np.asarray(x)
    return x / np.sqrt(np.sum(x ** 2, axis=1, keepdims=True))


def bilinear_align_corners(x, out_h, out_w):
    """torch.nn.UpsamplingBilinear2d(size=(out_h,out_w)) == bilinear, align_corners=True.

    x (B,C,H,W) float32.  Source coordinate = dst * (in-1)/(out-1) in float32, lower index
    by truncation, weights (1-l, l), rows combined after columns -- the order torch's CPU
    kernel uses.
    """
    x = np.asarray(x, dtype=np.float32)
    _, _, h, w = x.shape

    def axis(n_in, n_out):
        scale = np.float32(n_in - 1) / np.float32(n_out - 1) if n_out > 1 else np.float32(0)
        src = scale * np.arange(n_out, dtype=np.float32)
        i0 = src.astype(np.int64)
        i1 = np.minimum(i0 + 1, n_in - 1)
        l1 = (src - i0.astype(np.float32)).astype(np.float32)
        return i0, i1, np.float32(1) - l1, l1
    y0, y1, hy0, hy1 = axis(h, out_h)
    x0, x1, wx0, wx1 = axis(w, out_w)
    top = x[:, :, y0][:, :, :, x0] * wx0 + x[:, :, y0][:, :, :, x1] * wx1
    bot = x[:, :, y1][:, :, :, x0] * wx0 + x[:, :, y1][:, :, :, x1] * wx1
    return (hy0[:, None] * top + hy1[:, None] * bot).astype(np.float32)


def correlation_map_eccv(feature1, feature2, knn_inds, grid=28):
    """eccv_net.py:181-231.

    feature1 = real image features, feature2 = template features, both (B,C,H,W) f32;
    knn_inds (K,2) int64 [x, y] on the grid.  corr[k,b,y,x] = relu(sum_c f2[b,c,ky,kx] *
    f1[b,c,y,x]); out[b,k] = corr / sum_{y,x} exp(corr) * 10.  Returns (B,K,grid,grid) f32.
    """
    f1 = bilinear_align_corners(feature1, grid, grid)
    f2 = bilinear_align_corners(feature2, grid, grid)
    knn = np.asarray(knn_inds, dtype=np.int64)
    tmpl = f2[:, :, knn[:, 1], knn[:, 0]]                       # (B,C,K)
    corr = np.einsum("bck,bcyx->bkyx", tmpl, f1, optimize=False).astype(np.float32)
    corr = np.maximum(corr, np.float32(0))
    denom = np.sum(np.exp(corr), axis=(2, 3), keepdims=True, dtype=np.float32)
    return (corr / denom * np.float32(10)).astype(np.float32)


def correlation_map_acmm(feature1, feature2, mask, grid=28):
    """net_final_old.py:33-61.

    corr[b,p,y,x] = relu(sum_c f1[b,c,p] * f2[b,c,y,x]) with p = py*grid+px over feature1's
    cells; divided by sqrt(sum_p corr^2) (no epsilon: 0/0 = NaN as in the reference);
    multiplied by max_pool2d(mask, 2, 2, ceil_mode=True).  Returns (B, grid*grid, grid, grid).
    """
    f1 = bilinear_align_corners(feature1, grid, grid)
    f2 = bilinear_align_corners(feature2, grid, grid)
    b, c = f1.shape[:2]
    corr = np.einsum("bcp,bcyx->bpyx", f1.reshape(b, c, grid * grid), f2,
                     optimize=False).astype(np.float32)
    corr = np.maximum(corr, np.float32(0))
    norm = np.sqrt(np.sum(corr * corr, axis=1, keepdims=True, dtype=np.float32))
    with np.errstate(invalid="ignore", divide="ignore"):
        corr = corr / norm
    return (corr * max_pool2x2_ceil(mask)).astype(np.float32)


def max_pool2x2_ceil(mask):
    """F.max_pool2d(kernel_size=2, stride=2, ceil_mode=True) on (B,1,H,W)."""
    m = np.asarray(mask, dtype=np.float32)
    b, ch, h, w = m.shape
    oh, ow = -(-h // 2), -(-w // 2)
    pad = np.full((b, ch, oh * 2, ow * 2), -np.inf, dtype=np.float32)
    pad[:, :, :h, :w] = m
    return pad.reshape(b, ch, oh, 2, ow, 2).max(axis=(3, 5))
