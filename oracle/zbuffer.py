"""Oracle (test infrastructure): point-cloud z-buffer (packed depth|index minimum).

NOT IN THE REFERENCE -- the reference tests visibility against a Blender-rendered depth
map (run_proj.py:33-46) and its only point-to-depth code is the last-writer-wins splat of
io_utils.py:119-124 (restated in oracle.geometry.get_pt_depth_map).  PARITY UNPINNED; the
contract of this added mode is:

  * point i projects to pixel (x, y) = (int(u_i), int(v_i)) of an (H, W) image; it takes part
    iff 0 <= x < W, 0 <= y < H and z_i is finite and > 0;
  * key_i = (bits(float32(z_i)) << 32) | i  (float32 conversion rounds to nearest even;
    positive floats order like their bit patterns);
  * every pixel of the (2*radius+1)^2 block around (x, y), clipped to the image, keeps the
    minimum key; winner = low 32 bits, depth = high 32 bits as float32; pixels nobody wrote
    have winner -1 and depth +inf;
  * zbuffer_visibility: point i is visible iff it takes part and
    float64(z_i) - float64(depth[y, x]) <= relax.
"""
import numpy as np

EMPTY_KEY = np.uint64(0xFFFFFFFFFFFFFFFF)


def _pixels(pt_render, h, w):
    with np.errstate(invalid="ignore"):
        ok = np.isfinite(pt_render[0]) & np.isfinite(pt_render[1]) & np.isfinite(pt_render[2])
        x = np.where(ok, np.trunc(pt_render[0]), -1).astype(np.int64)
        y = np.where(ok, np.trunc(pt_render[1]), -1).astype(np.int64)
        ok &= (pt_render[2] > 0) & (x >= 0) & (x < w) & (y >= 0) & (y < h)
    return x, y, ok


def zbuffer(pt_render, h=540, w=960, radius=0):
    x, y, ok = _pixels(pt_render, h, w)
    zbits = pt_render[2].astype(np.float32).view(np.uint32).astype(np.uint64)
    keys = (zbits << np.uint64(32)) | np.arange(pt_render.shape[1], dtype=np.uint64)
    img = np.full((h, w), EMPTY_KEY, dtype=np.uint64)
    for dy in range(-radius, radius + 1):
        for dx in range(-radius, radius + 1):
            yy, xx = y + dy, x + dx
            m = ok & (yy >= 0) & (yy < h) & (xx >= 0) & (xx < w)
            np.minimum.at(img, (yy[m], xx[m]), keys[m])
    empty = img == EMPTY_KEY
    winner = np.where(empty, -1, (img & np.uint64(0xFFFFFFFF)).astype(np.int64)).astype(np.int32)
    depth = (img >> np.uint64(32)).astype(np.uint32).view(np.float32).copy()
    depth[empty] = np.inf
    return img, winner, depth


def zbuffer_visibility(pt_render, depth, relax=0.0):
    h, w = depth.shape
    x, y, ok = _pixels(pt_render, h, w)
    d = depth[np.where(ok, y, 0), np.where(ok, x, 0)].astype(np.float64)
    with np.errstate(invalid="ignore"):
        return ok & (pt_render[2] - d <= relax)
