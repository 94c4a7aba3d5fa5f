"""Dense nearest-neighbour correspondence and keypoint correlation maps.

  dense_nn                 the `KNN` layer contract of ucn_code/nets/train.prototxt:2116-2141
  l2norm                   ucn_code/lib/layers/l2norm_layer.py:18-29
  ind2coord, pck           ucn_code/lib/layers/hard_negative_processing_layer.py:44-58,188-190,218-234
  hard_negatives           ucn_code/lib/layers/hard_negative_processing_layer.py:188-192,246-250
  CorrelationMapLayer      classifier_code/train_test_scripts/arch/eccv_net/eccv_net.py:161-231
  correlation_map_acmm     classifier_code/train_test_scripts/arch/acmm_net/net_final_old.py:33-61
  get_pt, knn_inds_to_grid classifier_code/train_test_scripts/utils/proxy_renderer.py:50-58 and
                           multi_view_train.py:164,177 + eccv_net.py:282
"""
import numpy as np
import torch

from . import _abi, _device as dv, io_utils

_workspaces = {}
_ws_bytes = {}


def _workspace(nbytes):
    """Per-device scratch owned by this module (grown on demand, reused across calls)."""
    key = torch.cuda.current_device()
    ws = _workspaces.get(key)
    if ws is None or ws.numel() < nbytes:
        ws = torch.empty(int(nbytes), dtype=torch.uint8, device=dv.device())
        _workspaces[key] = ws
    return ws


def dense_nn(query, ref, k=1, index_base=1, return_stats=False, filter="native", num_query=None, normalize=False,
             pad=True):
    """k nearest reference cells (squared L2) for every query column.

    query (B,C,Hq,Wq) or (B,C,n); ref (B,C,H,W) or (B,C,nr); float32, or bfloat16 torch tensors.
    Returns dist (B,k,n) float32 and ind (B,k,n) int32 with ind = y*W + x + index_base (the
    reference's layer is 1-based), neighbours ordered by (distance, index).  The index order is
    exact with respect to float64 arithmetic on the given inputs.

    filter="split" (float32 inputs, k=1): the filter pass contracts a bfloat16 hi/lo split of the
    inputs on bfloat16 MFMAs instead of float32 MFMAs; results are the same exact ones.  Calls too small
    for that to pay (under 65536 queries) run the float32 filter; filter="split-always" keeps the split
    filter for them too (tests, timing).

    normalize=True (float32 inputs): both maps pass through L2NormLayer.forward (l2norm_layer.py:18-29)
    inside the call, as the reference's net does in front of the KNN layer (train.prototxt:2096-2115); the
    result is that of dense_nn(l2norm(query), l2norm(ref), ...), bit for bit, without the extra passes.

    pad=True: the library brings shapes off the grid of its fast kernels (channels not a multiple of 32, cell
    counts not multiples of 8) onto it inside the call, in ways that cannot change a result (zero channels,
    repeated query columns whose results are dropped, reference cells no query can select); pad=False
    (ISPA_NN_NO_PAD) makes it take the arrays as they are (guarded loaders: same results, slower).

    num_query (B,) or (B,1): the layer's `num_coord` bottom with `use_all_query: false`
    (train.prototxt:2128-2141) - only the first num_query[b] query columns of item b are real; the
    others come back as index -1 and distance +inf."""
    if filter not in ("native", "split", "split-always"):
        raise ValueError("filter must be 'native', 'split' or 'split-always'")
    as_tensor = dv.is_tensor(query)
    bf16 = as_tensor and query.dtype == torch.bfloat16
    tdt = torch.bfloat16 if bf16 else torch.float32
    q = dv.to_dev(query, tdt)
    r = dv.to_dev(ref, tdt)
    if q.dim() < 3 or r.dim() < 3 or q.shape[0] != r.shape[0] or q.shape[1] != r.shape[1]:
        raise ValueError(f"query {tuple(q.shape)} and ref {tuple(r.shape)} must share batch and channels")
    b, c = q.shape[:2]
    q = q.reshape(b, c, -1)
    r = r.reshape(b, c, -1)
    nq, nr = q.shape[2], r.shape[2]
    if not 1 <= k <= 8:
        raise ValueError("k must be in [1, 8]")
    dist = dv.empty((b, k, nq), torch.float32)
    ind = dv.empty((b, k, nq), torch.int32)
    stats = torch.zeros(4, dtype=torch.int32, device=dv.device()) if return_stats else None
    code = _abi.BF16 if bf16 else {"native": _abi.F32, "split": _abi.F32_SPLIT, "split-always": _abi.F32_SPLIT_ALWAYS}[filter]
    if normalize:
        if bf16:
            raise ValueError("normalize=True takes float32 features")
        code |= _abi.NN_NORMALIZE
    if not pad:
        code |= _abi.NN_NO_PAD
    key = (code, b, c, nq, nr, k)
    need = _ws_bytes.get(key)
    if need is None:
        need = _ws_bytes[key] = _abi.lib.ispa_nn_workspace_bytes_for(code, b, c, nq, nr, k)
    ws = _workspace(max(need, 256))
    _abi.check(_abi.lib.ispa_nn_topk(dv.ptr(q), dv.ptr(r), code, b, c, nq, nr, k,
                                     int(index_base), dv.ptr(dist), dv.ptr(ind), dv.ptr(ws), ws.numel(),
                                     dv.ptr(stats), dv.stream_ptr()))
    if num_query is not None:
        nv = dv.to_dev(np.asarray(num_query).reshape(-1) if not dv.is_tensor(num_query) else num_query.reshape(-1),
                       torch.int64)
        if nv.numel() != b:
            raise ValueError("num_query must hold one count per batch item")
        unreal = torch.arange(nq, device=nv.device)[None, None, :] >= nv[:, None, None]
        ind.masked_fill_(unreal, -1)
        dist.masked_fill_(unreal, float("inf"))
    out = (dv.back(dist, as_tensor), dv.back(ind, as_tensor))
    if return_stats:
        st = stats.cpu().tolist()
        out += ({"exact_path_queries": st[0] + st[1], "full_rescans": st[0], "candidate_rescores": st[1]},)
    return out


def extract_feature(feature, correspondences, downsampling_factor=4.0, coord_start_ind=0, mode="nearest"):
    """Query features for dense_nn, as the `ExtractFeature` layer in front of `KNN` produces them
    (ucn_code/nets/train.prototxt:2116-2127): feature (B,C,H,W) sampled at image coordinates
    correspondences[:, :, coord_start_ind : coord_start_ind + 2] / downsampling_factor -> (B,C,n) f32.
    mode "nearest" (cell floor(x), floor(y)) or "bilinear"; the layer's own source is not in the
    reference, so the sampling rule is this library's contract."""
    if mode not in ("nearest", "bilinear"):
        raise ValueError("mode must be 'nearest' or 'bilinear'")
    as_tensor = dv.is_tensor(feature)
    f = dv.to_dev(feature, torch.float32)
    c = dv.to_dev(correspondences, torch.float32)
    if f.dim() != 4 or c.dim() != 3 or c.shape[0] != f.shape[0] or c.shape[2] < coord_start_ind + 2:
        raise ValueError("feature must be (B,C,H,W) and correspondences (B,n,>=coord_start_ind+2)")
    b, ch, h, w = f.shape
    n = c.shape[1]
    out = dv.empty((b, ch, n), torch.float32)
    _abi.check(_abi.lib.ispa_extract_feature(dv.ptr(f), b, ch, h, w, dv.ptr(c), n, c.shape[2], int(coord_start_ind),
                                             float(downsampling_factor), int(mode == "bilinear"), dv.ptr(out),
                                             dv.stream_ptr()))
    return dv.back(out, as_tensor)


def l2norm(x):
    """x (B,C,...) float32 -> x / sqrt(sum_c x^2); no epsilon, like the reference layer."""
    as_tensor = dv.is_tensor(x)
    t = dv.to_dev(x, torch.float32)
    b, c = t.shape[:2]
    flat = t.reshape(b, c, -1)
    out = torch.empty_like(flat)
    _abi.check(_abi.lib.ispa_l2norm(dv.ptr(flat), b, c, flat.shape[2], dv.ptr(out), dv.stream_ptr()))
    return dv.back(out.reshape(t.shape), as_tensor)


def ind2coord(ind, width, downsampling_factor=4, index_base=1):
    """Flat KNN index -> (x, y) image coordinates: x = i % W, y = floor(i / W), times the
    feature stride (hard_negative_processing_layer.py:44-58 after the `- 1` of line 188)."""
    i = np.asarray(ind) - index_base
    return np.stack([i % width, np.floor(i / width)], axis=-1) * downsampling_factor


def pck(ind, correspondences, width, radii, downsampling_factor=4, index_base=1):
    """Hit counts of the 1-NN matches, as HardNegativeProcessingLayer.forward computes them
    (hard_negative_processing_layer.py:188-190,218-234), on the device (ispa_pck_counts).

    ind (B,k,n) from dense_nn (array or device tensor); correspondences (B,n,5) rows
    [x1, y1, x2, y2, flag]; a row takes part iff flag == 1; a hit is an image-space distance below
    the radius.  Returns (len(radii), B) float64 like the layer's `pcks` array."""
    radii = [float(r) for r in radii]
    if len(radii) > _abi.PCK_MAX_RADII:
        raise ValueError(f"at most {_abi.PCK_MAX_RADII} radii per call")
    as_tensor = dv.is_tensor(ind)
    it = dv.to_dev(ind, torch.int32)
    ct = dv.to_dev(correspondences, torch.float64)
    if it.dim() != 3 or ct.dim() != 3 or ct.shape[2] != 5 or ct.shape[:2] != (it.shape[0], it.shape[2]):
        raise ValueError("ind must be (B,k,n) and correspondences (B,n,5)")
    b, k, n = it.shape
    rt = dv.to_dev(np.asarray(radii, dtype=np.float64), torch.float64)
    counts = dv.empty((len(radii), b), torch.int32)
    _abi.check(_abi.lib.ispa_pck_counts(dv.ptr(it), b, k, n, dv.ptr(ct), int(width), float(downsampling_factor),
                                        int(index_base), dv.ptr(rt), len(radii), dv.ptr(counts), dv.stream_ptr()))
    return dv.back(counts.to(torch.float64), as_tensor)


NEGATIVE_MINING_PIXEL_THRESH = 4      # ucn_code/lib/config.py:80


def hard_negatives(ind, correspondences, width, thresh=NEGATIVE_MINING_PIXEL_THRESH, downsampling_factor=4,
                   index_base=1):
    """Which ground-truth correspondences the 1-NN search got wrong by more than `thresh` feature cells, as
    HardNegativeProcessingLayer.forward decides it (hard_negative_processing_layer.py:188-192,246-250), on the
    device (ispa_hard_negatives).

    ind (B,k,n) from dense_nn; correspondences (B,n,5) rows [x1, y1, x2, y2, flag].  Returns (mask (B,n) bool,
    knn_xy (B,n,2) float64): mask[b,i] is True where flag == 1 and the nearest cell is farther than thresh from
    (x2, y2) / downsampling_factor; knn_xy is the nearest cell's image coordinate, what the layer writes into
    columns 2:4 of the negative pairs it overwrites (line 265; the random choice of those pairs, lines 254-266,
    is the caller's).  The layer's `hard_neg_inds` is mask[b][flag == 1]."""
    as_tensor = dv.is_tensor(ind)
    it = dv.to_dev(ind, torch.int32)
    ct = dv.to_dev(correspondences, torch.float64)
    if it.dim() != 3 or ct.dim() != 3 or ct.shape[2] != 5 or ct.shape[:2] != (it.shape[0], it.shape[2]):
        raise ValueError("ind must be (B,k,n) and correspondences (B,n,5)")
    b, k, n = it.shape
    mask = dv.empty((b, n), torch.uint8)
    xy = dv.empty((b, n, 2), torch.float64)
    _abi.check(_abi.lib.ispa_hard_negatives(dv.ptr(it), b, k, n, dv.ptr(ct), int(width), float(downsampling_factor),
                                            int(index_base), float(thresh), dv.ptr(mask), dv.ptr(xy), dv.stream_ptr()))
    return dv.back(mask.bool(), as_tensor), dv.back(xy, as_tensor)


def resize_bilinear(x, size):
    as_tensor = dv.is_tensor(x)
    t = dv.to_dev(x, torch.float32)
    b, c, h, w = t.shape
    out = dv.empty((b, c, size[0], size[1]), torch.float32)
    _abi.check(_abi.lib.ispa_resize_bilinear(dv.ptr(t), b, c, h, w, size[0], size[1], dv.ptr(out),
                                             dv.stream_ptr()))
    return dv.back(out, as_tensor)


class CorrelationMapLayer:
    """eccv_net.CorrelationMapLayer: forward(feature1, feature2, knn_inds) -> (B,K,28,28).

    feature1 = real-image features, feature2 = template features, (B,C,H,W) float32;
    knn_inds (K,2) integer [x, y] template cells on the 28 x 28 grid."""

    def __init__(self, grid=28):
        self.grid = grid
        self._knn_cache = {}

    def forward(self, feature1, feature2, knn_inds):
        as_tensor = dv.is_tensor(feature1)
        f1 = dv.to_dev(feature1, torch.float32)
        f2 = dv.to_dev(feature2, torch.float32)
        if f1.shape != f2.shape or f1.dim() != 4:
            raise ValueError("feature maps must be two (B,C,H,W) arrays of the same shape")
        knn = self._knn(knn_inds).reshape(-1, 2)
        b, c, h, w = f1.shape
        k = knn.shape[0]
        out = dv.empty((b, k, self.grid, self.grid), torch.float32)
        _abi.check(_abi.lib.ispa_corr_map_eccv(dv.ptr(f1), dv.ptr(f2), dv.ptr(knn), b, c, h, w, k, self.grid,
                                               dv.ptr(out), None, dv.stream_ptr()))
        return dv.back(out, as_tensor)

    __call__ = forward

    def _knn(self, knn_inds):
        # host lists / arrays (what the reference passes, eccv_net.py:279) are range-checked without
        # touching the device; a device tensor costs one synchronising read
        if dv.is_tensor(knn_inds):
            host = knn_inds.cpu().numpy()
            if host.size and (host.min() < 0 or host.max() >= self.grid):
                raise IndexError(f"knn_inds outside the {self.grid} x {self.grid} grid")
            return dv.to_dev(knn_inds, torch.int64)
        # The keypoint lists of a class are the same call after call (the reference builds them once per
        # class and view): the checked device copy of a host list is kept, keyed by its contents, so a
        # repeated call uploads nothing.
        host = np.ascontiguousarray(np.asarray(knn_inds), dtype=np.int64)
        key = (self.grid, host.shape, host.tobytes(), str(dv.device()))
        hit = self._knn_cache.get(key)
        if hit is None:
            if host.size and (host.min() < 0 or host.max() >= self.grid):
                raise IndexError(f"knn_inds outside the {self.grid} x {self.grid} grid")
            if len(self._knn_cache) >= 64:
                self._knn_cache.pop(next(iter(self._knn_cache)))
            hit = self._knn_cache[key] = dv.to_dev(host, torch.int64)
        return hit

    def forward_views(self, feature_i, feature_j, knn_inds):
        """The per-view loop of eccv_net.Net.forward (eccv_net.py:282-289) in one call:
        feature_i, feature_j (B,V,C,H,W), knn_inds (V,K,2) -> (B,V,K,grid,grid), the reference's
        `embedding` after stack + permute."""
        as_tensor = dv.is_tensor(feature_i)
        f1 = dv.to_dev(feature_i, torch.float32)
        f2 = dv.to_dev(feature_j, torch.float32)
        if f1.shape != f2.shape or f1.dim() != 5:
            raise ValueError("feature maps must be two (B,V,C,H,W) arrays of the same shape")
        knn = self._knn(knn_inds)
        b, v, c, h, w = f1.shape
        if knn.dim() != 3 or knn.shape[0] != v or knn.shape[2] != 2:
            raise ValueError("knn_inds must be (V,K,2)")
        k = knn.shape[1]
        out = dv.empty((b, v, k, self.grid, self.grid), torch.float32)
        _abi.check(_abi.lib.ispa_corr_map_eccv_views(dv.ptr(f1), dv.ptr(f2), dv.ptr(knn), b, v, c, h, w, k, self.grid,
                                                     dv.ptr(out), None, dv.stream_ptr()))
        return dv.back(out, as_tensor)


def correlation_map_acmm(feature1, feature2, mask, grid=28):
    """net_final_old.py:33-61 -> (B, grid*grid, grid, grid) float32; mask (B,1,H,W)."""
    as_tensor = dv.is_tensor(feature1)
    f1 = dv.to_dev(feature1, torch.float32)
    f2 = dv.to_dev(feature2, torch.float32)
    m = dv.to_dev(mask, torch.float32)
    b, c, h, w = f1.shape
    if m.shape != (b, 1, h, w):
        raise ValueError(f"mask must be ({b},1,{h},{w})")
    out = dv.empty((b, grid * grid, grid, grid), torch.float32)
    ws = _workspace(4 * _abi.lib.ispa_corr_map_acmm_workspace_floats(b, c, grid))
    _abi.check(_abi.lib.ispa_corr_map_acmm(dv.ptr(f1), dv.ptr(f2), dv.ptr(m), b, c, h, w, grid, dv.ptr(out),
                                           dv.ptr(ws), dv.stream_ptr()))
    return dv.back(out, as_tensor)


def get_pt(anchors, file_name, depth):
    """proxy_renderer.get_pt: anchors (K,3) -> (1,K,3) = [u224, v224, visible] for the render
    named file_name, through the variant-B pruning.  depth: directory or (H,W) array."""
    anchors = np.asarray(anchors, dtype=np.float64)
    truth, crop_pts = io_utils.get_render_array(io_utils.get_params_from_file(file_name), anchors, depth)
    out = np.zeros((1, anchors.shape[0], 3))
    out[0, :, :2] = crop_pts[:2, :].T
    out[0, :, 2] = truth
    return out


def knn_inds_to_grid(knn_xy, grid=28, image=224):
    """clip to [0, image-1], divide by image/grid, truncate -- the path from get_pt's pixel
    coordinates to CorrelationMapLayer's integer cells."""
    return (np.clip(np.asarray(knn_xy, dtype=np.float64), 0, image - 1) / (float(image) / grid)).astype(np.int64)


def nearest_view(view_params, param_np, return_distance=False):
    """proxy_renderer.proxy_render's lookup (lines 101-106): index of the stored pose closest to
    view_params, angle differences wrapped, elevation and tilt weighted 1/8 (ispa_nearest_view).

    view_params (3,) -> int, or (n,3) -> (n,) int32 array; param_np (P,3) stored [az, el, tilt]
    (array or device tensor: a table kept on the device is not copied again)."""
    v = np.asarray(view_params, dtype=np.float64) if not dv.is_tensor(view_params) else view_params
    single = v.ndim == 1
    vt = dv.to_dev(v, torch.float64).reshape(-1, 3)
    st = dv.to_dev(param_np, torch.float64)
    if st.dim() != 2 or st.shape[1] != 3:
        raise ValueError("param_np must be (P,3)")
    idx = dv.empty((vt.shape[0],), torch.int32)
    dist = dv.empty((vt.shape[0],), torch.float64) if return_distance else None
    _abi.check(_abi.lib.ispa_nearest_view(dv.ptr(vt), vt.shape[0], dv.ptr(st), st.shape[0], dv.ptr(idx),
                                          dv.ptr(dist), dv.stream_ptr()))
    idx = idx.cpu().numpy()
    out = int(idx[0]) if single else idx
    if return_distance:
        d = dist.cpu().numpy()
        return out, (float(d[0]) if single else d)
    return out
