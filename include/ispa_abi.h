/*
 * ispa_abi.h -- C ABI of libispa_hip.so, the MI355X (gfx950) implementation of the dense
 * correspondence / disparity path of val-iisc/pose_estimation.
 *
 * The reference is pure Python; it has no FFI layer.  Each entry point below replaces the
 * Python function (or external Caffe layer) cited beside it, and is what a ctypes stub in the
 * reference would bind.
 *
 * Conventions
 *   - every pointer is a DEVICE pointer unless the parameter is called host_*;
 *   - the caller allocates and owns every buffer, including workspaces; the library keeps no
 *     reference to them after the call returns and never allocates or frees device memory;
 *   - all work is enqueued on `stream` (a hipStream_t passed as void*, NULL = default stream)
 *     and is asynchronous with respect to the host;
 *   - return value 0 = ok, negative = ispa_status; ispa_last_error() returns a thread-local
 *     message for the last failing call on the calling thread;
 *   - one GPU per process: the library holds no communicator and exports no collective.  Shards of
 *     pairs, maps or poses are independent; the exchange of result records between ranks is the
 *     caller's (the Python layer uses torch.distributed all_gather_into_tensor over RCCL), because the
 *     process that owns the ranks also owns their bootstrap and stream ordering;
 *   - threads: one host thread per GPU at a time.  Calls from several host threads on ONE GPU are not
 *     supported: the library keeps per-process state that is set on first use without locks (per-kernel
 *     launch attributes, the timing hooks' event slots, the scratch array of ispa_nn_topk_noalloc), and two
 *     calls that share a workspace race on the device.  Calls on different GPUs from different threads are
 *     fine once each GPU has made its first call of a given kind; the intended deployment is one process
 *     per GPU;
 *   - arrays are C-ordered; "pt_render" always means the reference's (3, N) float64 layout with
 *     rows [u, v, z] (pt2render.py:6-13), batched as (n_pose, 3, N).
 */
#ifndef ISPA_ABI_H
#define ISPA_ABI_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define ISPA_ABI_VERSION 1

typedef enum ispa_status {
    ISPA_OK = 0,
    ISPA_EINVAL = -1,       /* bad argument (shape, null pointer, unsupported value) */
    ISPA_EHIP = -2,         /* a HIP runtime call or kernel launch failed             */
    ISPA_EWORKSPACE = -3,   /* workspace too small                                    */
    ISPA_EUNSUPPORTED = -4  /* valid request this build has no kernel for             */
} ispa_status;

enum { ISPA_VIS_A = 0, ISPA_VIS_B = 1 };          /* depth-visibility variants         */
/* feature dtypes for ispa_nn_topk; ISPA_F32_SPLIT = float32 inputs, split-bfloat16 MFMA filter where it is
 * the faster one; ISPA_F32_SPLIT_ALWAYS = the same, also for calls too small for it to pay (tests, timing) */
enum { ISPA_F32 = 0, ISPA_BF16 = 1, ISPA_F32_SPLIT = 2, ISPA_F32_SPLIT_ALWAYS = 3 };
/* OR'ed into the dtype of ispa_nn_topk (float32 features only): L2NormLayer.forward
 * (ucn_code/lib/layers/l2norm_layer.py:18-29) is applied to both maps inside the norms pass, i.e. the call
 * returns what ispa_l2norm on q and r followed by ispa_nn_topk returns, bit for bit; the normalised copies
 * live in the workspace (ask ispa_nn_workspace_bytes_for with the flag set). */
enum { ISPA_NN_NORMALIZE = 256 };
/* OR'ed into the dtype of ispa_nn_topk: hand the arrays to the kernels as they are.  Without it a call whose
 * channel count is not a multiple of 32 (bfloat16 above 256 channels: of 64, up to 512) or whose cell counts
 * are not multiples of 8 (the grid of the MFMA filters: 16-byte rows) is brought onto that grid inside the call, in the workspace - zero channels, copies
 * of the last query column whose results are dropped, reference cells that no query can select, for any k -
 * with the same results as the guarded loaders this flag selects (tests, timing). */
enum { ISPA_NN_NO_PAD = 512 };

int ispa_abi_version(void);
const char *ispa_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Geometry: projection, visibility, crop remap
 * ---------------------------------------------------------------------------------------- */

/* get_render_location(pt_array, camera_matrix)
 *   data_creator/render_wt_pt_proj/pt2render.py:6-13 (same code at
 *   classifier_code/train_test_scripts/utils/pt2render.py:6-13), batched over camera matrices.
 * pts (n,3) f64; cams (n_pose,3,4) f64; out (n_pose,3,n) f64 = [rint(x/z), rint(y/z), z].
 * x,y,z are the k-sequential fma chain fma(P[r][3],1,fma(P[r][2],pz,fma(P[r][1],py,P[r][0]*px))). */
int ispa_project(const double *pts, int64_t n, const double *cams, int n_pose, double *out,
                 void *stream);

/* depth_visibility_array(pt_render, depth_map, error_relaxation)
 *   variant ISPA_VIS_A: data_creator/render_wt_pt_proj/pt2render.py:43-55
 *       truth = gate && |z - depth[y][x]| <= relax
 *   variant ISPA_VIS_B: classifier_code/train_test_scripts/utils/pt2render.py:43-53
 *       truth = gate && z <= min_{3x3, clipped}(depth + (float)relax)   (sum and min in float)
 *   gate: y = (int)v, x = (int)u, 0 < y < ylim, 0 < x < xlim   (reference: 540, 950).
 * depth is (H,W) f32 shared by all poses (depth_per_pose = 0) or (n_pose,H,W).
 * truth (n_pose, n) bytes 0/1.  pt_render is not modified (the pruned copy variant B returns
 * is a host-side selection by truth). */
int ispa_depth_visibility(const double *pt_render, int64_t n, int n_pose, const float *depth,
                          int depth_per_pose, int H, int W, int ylim, int xlim, double relax,
                          int variant, uint8_t *truth, void *stream);

/* crop_visibility_array(pt_render, crop_params, resize)
 *   data_creator/render_wt_pt_proj/pt2render.py:18-38.  IN PLACE on rows 0,1 of pt_render.
 * crop (n_pose,4) int32 [top,bottom,left,right], or one row shared (crop_per_pose = 0). */
int ispa_crop_remap(double *pt_render, int64_t n, int n_pose, const int32_t *crop,
                    int crop_per_pose, int resize, uint8_t *truth, void *stream);

/* get_3d_to_2d + prune_pts            data_creator/render_wt_pt_proj/run_proj.py:24-46
 * get_render_array                    classifier_code/train_test_scripts/utils/io_utils.py:97-117
 * One pass: project, depth-visibility (variant), optional crop test + remap.
 * crop may be NULL (no crop: coordinates stay in render pixels, truth = depth truth).
 * out (n_pose,3,n) f64; truth (n_pose,n) = crop_truth * depth_truth;
 * truth_depth / truth_crop (n_pose,n) optional (NULL to skip). */
int ispa_project_prune(const double *pts, int64_t n, const double *cams, int n_pose,
                       const float *depth, int depth_per_pose, int H, int W, int ylim, int xlim,
                       double relax, int variant, const int32_t *crop, int crop_per_pose,
                       double *out, uint8_t *truth, uint8_t *truth_depth, uint8_t *truth_crop,
                       void *stream);

/* ------------------------------------------------------------------------------------------
 * Splat writers
 * ---------------------------------------------------------------------------------------- */

/* Flow splat of make_disparity      data_creator/make_disparity_syn.py:55-68
 * uv1, uv2 (n_map,3,n) f64 crop coordinates of pose 1 / pose 2; truth (n_map,n) = truth_1.
 * For every kept point in index order: flow[v-s:v+s, u-s:u+s, :] = (uv2-uv1)[:2] with Python
 * slice semantics, i.e. per pixel the kept point with the HIGHEST index wins.
 * owner_ws (n_map, side*side) int32 scratch, needed (and touched) only for maps of more than 200704 pixels
 * (side > 448), may be NULL otherwise; flow (n_map, side, side, 2) f64. */
int ispa_flow_splat(const double *uv1, const double *uv2, const uint8_t *truth, int64_t n,
                    int n_map, int spread, int side, int32_t *owner_ws, double *flow,
                    void *stream);

/* The same splat on (H, W) maps, as the flow demo writes it at the render's own resolution
 *   data_creator/render_wt_pt_proj/get_of.py:52-60 (flow = zeros((540,960,2)); the caller masks by
 *   truth_1 & truth_2, get_of.py:48, and passes un-cropped render coordinates).
 * flow (n_map, H, W, 2) f64.  owner_ws (n_map, H*W) int32 scratch is needed unless H == W <= 448. */
int ispa_flow_splat_hw(const double *uv1, const double *uv2, const uint8_t *truth, int64_t n,
                       int n_map, int spread, int H, int W, int32_t *owner_ws, double *flow,
                       void *stream);

/* Body of make_disparity for a batch of renders of one model, fused
 *   data_creator/make_disparity_syn.py:44-68 (get_3d_to_2d + prune_pts at the render's pose and at
 *   the pose turned by (10,10), masking by the first pose's truth, flow splat).
 * pts (n,3) f64; cams1, cams2 (n_map,3,4) f64; depth (n_map,H,W) f32 or one shared (H,W);
 * crop (n_map,4) int32 or one shared row; flow (n_map,224,224,2) f64.  Same results as
 * ispa_project_prune x 2 + ispa_flow_splat without materialising the projected coordinates. */
int ispa_make_disparity(const double *pts, int64_t n, const double *cams1, const double *cams2,
                        int n_map, const float *depth, int depth_per_map, int H, int W, int ylim,
                        int xlim, double relax, const int32_t *crop, int crop_per_map, int spread,
                        double *flow, void *stream);

/* The same body with a workspace of ispa_make_disparity_workspace_bytes(n, n_map), about 16 bytes per
 * (map, point) and 8-byte aligned: projection, crop and visibility run once per (map, point) instead of once
 * per half-map band, and leave lists of the visible points for the bands to read (room for the worst case; a
 * map's visible points are what is written and read back).  A call whose maps fill the chip (at least 85 % of
 * the compute units in every round of n_map / compute units) runs as one launch with a workgroup per map;
 * other calls run a launch over (map, point) and a launch over (map, band).  Same flow maps, bit for bit; about
 * a third faster at 100 000 points.  The workspace needs no initialisation and holds nothing between calls. */
size_t ispa_make_disparity_workspace_bytes(int64_t n, int n_map);
int ispa_make_disparity_ws(const double *pts, int64_t n, const double *cams1, const double *cams2,
                           int n_map, const float *depth, int depth_per_map, int H, int W, int ylim,
                           int xlim, double relax, const int32_t *crop, int crop_per_map, int spread,
                           double *flow, void *workspace, size_t workspace_bytes, void *stream);

/* The same body for points given in a processing order of the caller's choice.  make_disparity's result depends on
 * the order of the points (the last point written wins a pixel, make_disparity_syn.py:65-68); this entry point
 * separates the order the points are PROCESSED in from the order that decides who wins: pts (n,3) is the model in
 * processing order (e.g. along a space-filling curve, so that neighbouring lanes gather neighbouring depth texels),
 * rank (n) int32 gives each of them its index in the reference's order, inverse (n) int32 is the inverse
 * permutation (inverse[rank[i]] = i).  Same flow maps, bit for bit, as ispa_make_disparity_ws on the points in the
 * reference's order.  rank = inverse = NULL: array order, i.e. ispa_make_disparity_ws.  A rank outside [0, n) drops
 * its point; nothing outside the arrays is read. */
int ispa_make_disparity_ordered(const double *pts, const int32_t *rank, const int32_t *inverse, int64_t n,
                                const double *cams1, const double *cams2, int n_map, const float *depth,
                                int depth_per_map, int H, int W, int ylim, int xlim, double relax,
                                const int32_t *crop, int crop_per_map, int spread, double *flow,
                                void *workspace, size_t workspace_bytes, void *stream);

/* get_pt_depth_map(render_array, size, radius)
 *   classifier_code/train_test_scripts/utils/io_utils.py:119-124: z splat, last writer wins.
 * owner_ws (H*W) int32 scratch; depth_out (H,W) f64. */
int ispa_point_depth_splat(const double *pt_render, int64_t n, int H, int W, int radius,
                           int32_t *owner_ws, double *depth_out, void *stream);

/* Z-buffer of the projected points (no counterpart in the reference, which reads a rendered
 * depth image): per pixel the minimum of (bits(float(z)) << 32 | index) over the points whose
 * (2*radius+1)^2 block covers it.  packed (n_pose,H,W) u64 scratch/out; winner (n_pose,H,W)
 * int32 (-1 empty) and depth (n_pose,H,W) f32 (+inf empty) are optional outputs. */
int ispa_zbuffer(const double *pt_render, int64_t n, int n_pose, int H, int W, int radius,
                 uint64_t *packed, int32_t *winner, float *depth, void *stream);

/* visible = takes part && (double)z - (double)depth[y][x] <= relax. */
int ispa_zbuffer_visibility(const double *pt_render, int64_t n, int n_pose, const float *depth,
                            int H, int W, double relax, uint8_t *truth, void *stream);
/* The same test with the depth taken from the packed keys ispa_zbuffer leaves in `packed` (no
 * winner / depth decode needed in between). */
int ispa_zbuffer_visibility_packed(const double *pt_render, int64_t n, int n_pose, const uint64_t *packed,
                                   int H, int W, double relax, uint8_t *truth, void *stream);

/* Projection + z-buffer + visibility of a point set at n_pose cameras in one pass, screen-tiled: the keys
 * of a tile of pixels live in LDS, so there are no global atomics and the projected coordinates are never
 * written (no counterpart in the reference beyond the projection itself, pt2render.py:6-13).
 * pts (n,3) f64; cams (n_pose,3,4) f64.  Outputs, each optional (NULL to skip), bit-identical to
 * ispa_project -> ispa_zbuffer -> ispa_zbuffer_visibility_packed:
 *   packed (n_pose,H,W) u64 keys, winner (n_pose,H,W) int32, depth (n_pose,H,W) f32, truth (n_pose,n) u8.
 * workspace: ispa_project_zbuffer_workspace_bytes(n, n_pose, H, W) bytes, 256-byte aligned (bounding boxes
 * of chunks of 64 consecutive points per pose).  Results do not depend on the order of the points; speed
 * does: with the model's points in a spatially coherent order (e.g. along a Hilbert curve, once per model:
 * run_proj.spatial_order in the Python layer) a tile visits only the chunks that project near it, with
 * points in random order every tile visits every chunk. */
size_t ispa_project_zbuffer_workspace_bytes(int64_t n, int n_pose, int H, int W);
int ispa_project_zbuffer(const double *pts, int64_t n, const double *cams, int n_pose, int H, int W,
                         int radius, double relax, uint64_t *packed, int32_t *winner, float *depth,
                         uint8_t *truth, void *workspace, size_t workspace_bytes, void *stream);
/* The same pass over already projected coordinates (pt_render as ispa_project writes it); same workspace size. */
int ispa_zbuffer_tiled(const double *pt_render, int64_t n, int n_pose, int H, int W, int radius, double relax,
                       uint64_t *packed, int32_t *winner, float *depth, uint8_t *truth, void *workspace,
                       size_t workspace_bytes, void *stream);

/* ------------------------------------------------------------------------------------------
 * Dense nearest-neighbour correspondence
 * ---------------------------------------------------------------------------------------- */

/* `KNN` layer of the Caffe-UCN fork as used at ucn_code/nets/train.prototxt:2116-2141
 * (external to the reference; consumer hard_negative_processing_layer.py:175-192).
 * q (B,C,nq), r (B,C,nr): NCHW feature maps with the spatial axes flattened; dtype ISPA_F32
 * (float) or ISPA_BF16 (bfloat16 bits).  For every query column the k references with the
 * smallest squared L2 distance, ordered by (distance, index); the index order is EXACT with
 * respect to float64 arithmetic on the given inputs (MFMA filter pass + float64 re-scoring of
 * every query whose filter margin is inside the rounding bound).  Features must be finite: a NaN or
 * infinity in a column makes the results for that column (and, in a reference column, possibly for
 * every query of the pair) unspecified; no memory outside the outputs is touched.
 * idx (B,k,nq) int32 = flat reference index + index_base (the reference's layer is 1-based);
 * dist (B,k,nq) f32 squared distances, within 1e-4 relative of the float64 value with no absolute floor
 * (a distance the filter score cannot give to that accuracy is recomputed in float64: an exact match
 * returns 0.0); stats (optional, 4 int32): [0] = queries re-scored
 * against every reference in float64, [1] = queries settled by float64 re-scoring of the few
 * references that the filter could not separate.
 * ISPA_F32_SPLIT takes the same float32 arrays as ISPA_F32 and returns the same exact index order;
 * its filter pass runs the contraction on bfloat16 MFMAs over a hi/lo split of the inputs (top-1,
 * C a multiple of 32 up to 256, cell counts multiples of 8; other requests, and calls of less than
 * 65536 queries (B nq), which the float32 filter finishes sooner, run as
 * ISPA_F32) and needs a larger workspace: ask ispa_nn_workspace_bytes_for. */
size_t ispa_nn_workspace_bytes(int B, int C, int64_t nq, int64_t nr, int k);     /* ISPA_F32 / ISPA_BF16 */
size_t ispa_nn_workspace_bytes_for(int dtype, int B, int C, int64_t nq, int64_t nr, int k);
int ispa_nn_topk(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr,
                 int k, int index_base, float *dist, int32_t *idx, void *workspace,
                 size_t workspace_bytes, int32_t *stats, void *stream);

/* ispa_nn_topk without a workspace argument, for callers that cannot allocate device memory (code under
 * HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).  No counterpart in the
 * reference.  Same arguments, results and stream semantics as ispa_nn_topk; the workspace is a static array
 * of ISPA_NN_NOALLOC_SCRATCH_BYTES inside the library's code object, one copy per GPU, held until the
 * process ends.  The call allocates and frees nothing.  Problems whose ispa_nn_workspace_bytes_for() exceeds
 * the array are rejected with ISPA_EWORKSPACE (one 64x64x128 pair needs 4 to 8 MiB, the three 128x128x256 top-5
 * views of one multi-view query 38 MiB).  All calls on one GPU share that GPU's array: calls that overlap in time on one GPU
 * (two streams, two host threads) race; calls on one stream are ordered and safe.
 * Under stream capture: the FIRST call on a device looks up the array's address (hipGetSymbolAddress) and the
 * first call of each kernel variant (dtype, channel count, k) raises that kernel's dynamic-LDS limit
 * (hipFuncSetAttribute); neither is a stream operation, so make one un-captured call with the same dtype, C and k
 * on the device before capturing.  Every later call only enqueues kernels and memsets on `stream`. */
#define ISPA_NN_NOALLOC_SCRATCH_BYTES (64u << 20)
size_t ispa_nn_noalloc_scratch_bytes(void);
int ispa_nn_topk_noalloc(const void *q, const void *r, int dtype, int B, int C, int64_t nq, int64_t nr,
                         int k, int index_base, float *dist, int32_t *idx, int32_t *stats, void *stream);

/* Measurement hooks (no counterpart in the reference).  After ispa_timing_enable(n), each of the
 * next n calls of ispa_nn_topk (its filter kernel, the launch that does the MFMA contraction),
 * ispa_project, ispa_project_prune, ispa_flow_splat (square maps up to 448 pixels a side), ispa_make_disparity /
 * _ws / _ordered (both launches of the latter two as one record), ispa_zbuffer, ispa_project_zbuffer or
 * ispa_corr_map_eccv / ispa_corr_map_acmm brackets its dominant kernel with HIP events on the call's stream.  ispa_timing_collect waits for
 * them, writes the elapsed milliseconds of the recorded calls to a HOST array and re-arms the
 * slots.  ispa_timing_enable(0) turns it off and releases the events.  Not thread-safe. */
int ispa_timing_enable(int max_records);
int ispa_timing_collect(float *host_ms, int capacity, int *host_count);

/* L2NormLayer.forward                 ucn_code/lib/layers/l2norm_layer.py:18-29
 * x, out (B,C,hw) f32: out = x / sqrt(sum_c x^2), no epsilon. */
int ispa_l2norm(const float *x, int B, int C, int64_t hw, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Correlation maps
 * ---------------------------------------------------------------------------------------- */

/* torch.nn.UpsamplingBilinear2d(size) (align_corners=True) as called at
 * eccv_net.py:184-185 and net_final_old.py:33-34.  in (B,C,H,W) -> out (B,C,oh,ow) f32. */
int ispa_resize_bilinear(const float *in, int B, int C, int H, int W, int oh, int ow, float *out,
                         void *stream);

/* eccv_net.CorrelationMapLayer.forward(feature1, feature2, knn_inds)
 *   classifier_code/train_test_scripts/arch/eccv_net/eccv_net.py:181-231
 * f1 (real), f2 (template): (B,C,H,W) f32; knn (K,2) int64 [x,y] on the grid x grid lattice.
 * out (B,K,grid,grid) f32 = relu(corr) / sum_yx exp(relu(corr)) * 10.  The 56 -> 28 bilinear resize
 * (eccv_net.py:184-185) happens inside the kernels' loaders: no resized copy is made.  A keypoint off the
 * lattice correlates as an all-zero template (the reference raises IndexError; so does the Python layer).
 * workspace: unused, may be NULL (kept in the signature for ABI version 1). */
int ispa_corr_map_eccv(const float *f1, const float *f2, const int64_t *knn, int B, int C, int H,
                       int W, int K, int grid, float *out, float *workspace, void *stream);

/* The per-view loop of eccv_net.Net.forward around that layer, in one call
 *   classifier_code/train_test_scripts/arch/eccv_net/eccv_net.py:282-289 (loop, stack, permute)
 * f1, f2 (B,V,C,H,W) f32; knn (V,K,2) int64, one keypoint list per view; out (B,V,K,grid,grid) f32,
 * which is the reference's `embedding` before its .view().  workspace: unused, may be NULL. */
int ispa_corr_map_eccv_views(const float *f1, const float *f2, const int64_t *knn, int B, int V, int C, int H,
                             int W, int K, int grid, float *out, float *workspace, void *stream);

/* acmm_net CorrelationMapLayer.forward, correlation part
 *   classifier_code/train_test_scripts/arch/acmm_net/net_final_old.py:33-61
 * mask (B,1,H,W) f32 (max-pooled 2x2 ceil inside).  out (B, grid*grid, grid, grid) f32.
 * workspace: ispa_corr_map_acmm_workspace_floats(B, C, grid) floats (both maps at the grid's size, written by
 * one launch, and the column partials of the normalisation: 2*B*C*cells + 2*B*ceil(cells/64)*cells with
 * cells = grid*grid in this build; ask the function, the tiling may change). */
size_t ispa_corr_map_acmm_workspace_floats(int B, int C, int grid);
int ispa_corr_map_acmm(const float *f1, const float *f2, const float *mask, int B, int C, int H,
                       int W, int grid, float *out, float *workspace, void *stream);

/* `ExtractFeature` layer of the Caffe-UCN fork as the reference configures it (the producer of the KNN queries)
 *   ucn_code/nets/train.prototxt:2116-2127 (num_coord_elements 5, coord_start_ind 0, downsampling_factor 4.0)
 * feature (B,C,H,W) f32; coords (B,n,num_coord_elements) f32 rows with image (x, y) in columns
 * coord_start_ind, +1; out (B,C,n) f32 = feature at (x, y) / downsampling: the cell (floor x, floor y) clamped
 * to the map, or (bilinear != 0) the float32 interpolation of the four surrounding cells.  The layer's
 * source is not part of the reference: the sampling rule is this library's contract. */
int ispa_extract_feature(const float *feature, int B, int C, int H, int W, const float *coords, int64_t n,
                         int num_coord_elements, int coord_start_ind, float downsampling, int bilinear,
                         float *out, void *stream);

/* proxy_renderer.proxy_render's lookup, batched
 *   classifier_code/train_test_scripts/utils/proxy_renderer.py:100-106 (argmin) and :138-151 (normalize)
 * view_params (n_view,3), stored (n_stored,3) f64 rows [azimuth, elevation, tilt] in degrees.
 * index[v] = first j minimising || wrap(stored[j] - view[v]) * [1, 1/8, 1/8] ||_2, differences wrapped
 * to [-180,180) / [-90,90) / [-180,180) as normalize does.  distance (optional, may be NULL) = that norm. */
int ispa_nearest_view(const double *view_params, int n_view, const double *stored, int n_stored,
                      int32_t *index, double *distance, void *stream);

/* HardNegativeProcessingLayer.forward, index decode + PCK hit counts
 *   ucn_code/lib/layers/hard_negative_processing_layer.py:44-58 (_ind2coord), :188-190, :218-234
 * ind (B,k,n) i32 as ispa_nn_topk writes it (row 0 = nearest); correspondences (B,n,5) f64 rows
 * [x1, y1, x2, y2, flag]; a row counts iff flag == 1.  Image coordinate of index j (after subtracting
 * index_base): ((j mod width) * downsampling, floor(j / width) * downsampling).
 * counts (n_radii,B) i32: rows whose coordinate is closer than radii[r] to (x2, y2). */
#define ISPA_PCK_MAX_RADII 8
int ispa_pck_counts(const int32_t *ind, int B, int k, int64_t n, const double *correspondences, int width,
                    double downsampling, int index_base, const double *radii, int n_radii,
                    int32_t *counts, void *stream);

/* HardNegativeProcessingLayer.forward, which correspondences are hard negatives
 *   ucn_code/lib/layers/hard_negative_processing_layer.py:188-192 (coordinates), :246-250 (the test);
 *   threshold cfg.NEGATIVE_MINING_PIXEL_THRESH = 4 feature cells (ucn_code/lib/config.py:80)
 * ind, correspondences, width, downsampling, index_base as for ispa_pck_counts.
 * mask (B,n) u8 = 1 where the row is valid (flag == 1) and its nearest cell (x, y) lies farther than thresh from
 * the ground truth in FEATURE coordinates, (x2, y2) / downsampling; knn_xy (optional, may be NULL) (B,n,2) f64 =
 * the nearest cell's image coordinate (x, y) * downsampling, which the layer writes over a negative pair's
 * columns 2:4 (:265).  Which random negatives get overwritten (:254-266) draws from numpy's global random
 * stream and stays with the caller. */
int ispa_hard_negatives(const int32_t *ind, int B, int k, int64_t n, const double *correspondences, int width,
                        double downsampling, int index_base, double thresh, uint8_t *mask, double *knn_xy,
                        void *stream);

/* Pruning of dense 2-D skeleton keypoints on annotated real images, batched over images
 *   data_creator/utils.py:274-290 (surface_pruned), :292-325 (dir_pruned), :327-348 (surface_dir_prune);
 *   caller data_creator/make_dense_pointset.py:91-95
 * coords (B,2,n) f64 rows [x; y], edited IN PLACE: a pruned point becomes (0, 0), as the reference zeroes
 * the column.  flags (B,n) u8 says which tests a point takes (OR of the ISPA_PRUNE_* bits):
 *   POLYGON  pruned if inside or on the closed polygon (B,n_vert,2) i32 [x, y] (the seat outline; the
 *            reference rasterises it, traces the contour and calls cv2.pointPolygonTest(...) > -1);
 *   BAND     pruned if the pixel (trunc x, trunc y) lies within thickness/2 of one of the image's first
 *            n_seg[b] segments (B,max_seg,4) i32 [x0, y0, x1, y1] (the reference draws them with
 *            cv2.line(..., 255, thickness) and reads src[y, x]; a negative pixel index wraps once);
 *   DROP     pruned unconditionally (bones surface_dir_prune removes by facing direction).
 * pruned (optional, may be NULL) (B,n) u8 = 1 where the point was zeroed.  Corner and end point
 * coordinates are taken clamped to [-8192, 8192]; H, W <= 8192.  The closed-form tests stand in for
 * OpenCV's rasterisers, which are not part of the reference: points within about a pixel of an outline
 * or band rim may be decided differently than by OpenCV. */
#define ISPA_PRUNE_POLYGON 1
#define ISPA_PRUNE_BAND 2
#define ISPA_PRUNE_DROP 4
#define ISPA_PRUNE_MAX_VERTICES 64
#define ISPA_PRUNE_MAX_SEGMENTS 256
int ispa_prune_points_2d(double *coords, int B, int64_t n, const uint8_t *flags, const int32_t *polygon,
                         int n_vert, const int32_t *segments, const int32_t *n_seg, int max_seg, int thickness,
                         int H, int W, uint8_t *pruned, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* ISPA_ABI_H */
