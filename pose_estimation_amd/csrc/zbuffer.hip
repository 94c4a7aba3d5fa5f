// Screen-tiled z-buffer of a projected point set: projection, packed (depth | index) minimum per pixel and
// visibility flags in one pass per pose, without global atomics and without writing the projected
// coordinates.  (The reference has no point z-buffer: it reads a rendered depth image, run_proj.py:33-46;
// its only point-to-depth code is the overwrite splat of classifier_code/.../io_utils.py:119-124.  The
// projection is pt2render.py:6-13, arithmetic as in geom_device.h.)
//
// The image is cut into tiles of kTileH x kTileW pixels whose u64 keys live in LDS (64 KiB).  The points
// are taken in chunks of kChunk consecutive points (a wave's worth); a chunk's pixel bounding box per pose comes from
// projecting the 8 corners of its 3-D bounding box (rigorous for a projective map while all corners are in
// front of the camera; otherwise the chunk counts as covering the whole image).  A workgroup owns one
// (pose, tile): it visits the chunks whose box touches the tile, projects their points and takes an LDS
// atomicMin per covered pixel; then visits them again to test each point that lands in the tile against
// the finished tile (visibility; pass 1 leaves the tile offsets of the points it placed in LDS, so that
// pass 2 recomputes only the depth of those points); then streams the tile to HBM once.  How many chunks a tile must visit
// depends on the order of the points: correctness never does.  With a model's points along a Hilbert curve
// (run_proj.spatial_order, once per model) a chunk is a small blob on the screen and touches 1-2 tiles.
#include "common.h"
#include "geom_device.h"

namespace {

constexpr int kChunk = 64;                  // points per chunk: a wave takes a chunk visit
#ifndef ZB_THREADS
#define ZB_THREADS 512
#define ZB_TILE_H 64
#define ZB_CACHE 94
#define ZB_WG_PER_CU 2
#endif
constexpr int kZThreads = ZB_THREADS;       // two workgroups per CU
constexpr int kWgPerCu = ZB_WG_PER_CU;
constexpr int kZWaves = kZThreads / 64;
#ifndef ZB_BATCH
#define ZB_BATCH 4
#endif
constexpr int kBatch = ZB_BATCH;                   // chunk visits whose loads a wave has in flight together
constexpr int kTileW = 128, kTileH = ZB_TILE_H;   // 8192 keys = 64 KiB of LDS
constexpr int kListMax = 2 * kZThreads;     // visiting chunks listed at a time (4 KiB)
constexpr int kRounds = 4;                  // rounds of kZThreads boxes a list segment tries to take
constexpr int kCacheVisits = ZB_CACHE;            // chunk visits whose tile offsets pass 1 leaves in LDS for pass 2 (12 KiB)
constexpr int kGroup = 4;                   // consecutive items taken off the counter at a time
constexpr unsigned long long kEmptyKey = ~0ull;
typedef unsigned long long u64x2 __attribute__((ext_vector_type(2)));
typedef int i32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
constexpr int kBoxWhole = 32000;            // |bound| of a box that covers every image (H, W <= 16384)
constexpr unsigned kNoPixel = 0xffffu;

// ---- per model: 3-D bounding box of every chunk (NaN coordinates are skipped: such points never take part)
__global__ __launch_bounds__(256) void zb_chunk_aabb_kernel(const double *__restrict__ pts, int64_t n, int nchunks,
                                                            double *__restrict__ aabb) {
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int64_t i = (int64_t)c * kChunk + (threadIdx.x & 63);
    double lo[3], hi[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double v = i < n ? pts[3 * i + a] : __builtin_nan("");
        lo[a] = v == v ? v : __builtin_inf();
        hi[a] = v == v ? v : -__builtin_inf();
    }
#pragma unroll
    for (int a = 0; a < 3; ++a)
        for (int off = 32; off > 0; off >>= 1) {
            lo[a] = fmin(lo[a], __shfl_xor(lo[a], off));
            hi[a] = fmax(hi[a], __shfl_xor(hi[a], off));
        }
    if ((threadIdx.x & 63) == 0)
#pragma unroll
        for (int a = 0; a < 3; ++a) {
            aabb[(int64_t)c * 6 + a] = lo[a];
            aabb[(int64_t)c * 6 + 3 + a] = hi[a];
        }
}

// ---- per (pose, chunk): pixel box [x0, x1] x [y0, y1] of everything the chunk's points can touch
// A pose's own box = union of its chunks' boxes, kept as four minima (x0, -x1, y0, -y1).
__device__ __forceinline__ void zb_pose_box_update(int4 *pose_box_p, bool valid, const short4 o) {
    int v[4] = {valid ? (int)o.x : kBoxWhole, valid ? -(int)o.y : kBoxWhole, valid ? (int)o.z : kBoxWhole,
                valid ? -(int)o.w : kBoxWhole};
#pragma unroll
    for (int a = 0; a < 4; ++a)
        for (int off = 32; off > 0; off >>= 1) v[a] = min(v[a], __shfl_xor(v[a], off));
    if ((threadIdx.x & 63) == 0) {
        atomicMin(&pose_box_p->x, v[0]);
        atomicMin(&pose_box_p->y, v[1]);
        atomicMin(&pose_box_p->z, v[2]);
        atomicMin(&pose_box_p->w, v[3]);
    }
}

__global__ __launch_bounds__(256) void zb_chunk_box_kernel(const double *__restrict__ aabb, int nchunks,
                                                           const double *__restrict__ cams, int radius,
                                                           short4 *__restrict__ box, int4 *__restrict__ pose_box) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    const int p = blockIdx.y;
    const bool valid = c < nchunks;
    const double *P = cams + (int64_t)p * 12;
    const double *b = aabb + (int64_t)(valid ? c : 0) * 6;
    double umin = __builtin_inf(), umax = -__builtin_inf(), vmin = __builtin_inf(), vmax = -__builtin_inf();
    bool whole = false;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
        const double x = b[(corner & 1) ? 3 : 0], y = b[(corner & 2) ? 4 : 1], z = b[(corner & 4) ? 5 : 2];
        double a[3];
#pragma unroll
        for (int r = 0; r < 3; ++r) a[r] = fma(P[4 * r + 3], 1.0, fma(P[4 * r + 2], z, fma(P[4 * r + 1], y, P[4 * r] * x)));
        const double u = a[0] / a[2], v = a[1] / a[2];
        // behind or on the camera plane, or not finite: no bound can be given
        if (!(a[2] > 0.0) || !(fabs(u) < 1.0e6) || !(fabs(v) < 1.0e6)) whole = true;
        umin = fmin(umin, u);
        umax = fmax(umax, u);
        vmin = fmin(vmin, v);
        vmax = fmax(vmax, v);
    }
    short4 o;
    if (whole) {
        o.x = o.z = (short)-kBoxWhole;
        o.y = o.w = (short)kBoxWhole;
    } else {
        // the pixel is the rounded quotient: one pixel of slack covers rounding, `radius` the splat
        const double m = (double)(radius + 1);
        o.x = (short)fmax(floor(umin) - m, (double)-kBoxWhole);
        o.y = (short)fmin(ceil(umax) + m, (double)kBoxWhole);
        o.z = (short)fmax(floor(vmin) - m, (double)-kBoxWhole);
        o.w = (short)fmin(ceil(vmax) + m, (double)kBoxWhole);
    }
    if (valid) box[(int64_t)p * nchunks + c] = o;
    zb_pose_box_update(pose_box + p, valid, o);
}

// Same table from projected coordinates (pt_render rows u, v): exact min / max of the chunk's pixels.
__global__ __launch_bounds__(256) void zb_chunk_box_render_kernel(const double *__restrict__ pt_render, int64_t n,
                                                                  int nchunks, int radius, short4 *__restrict__ box,
                                                                  int4 *__restrict__ pose_box) {
    const int p = blockIdx.y;
    const int c = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (c >= nchunks) return;
    const int64_t i = (int64_t)c * kChunk + (threadIdx.x & 63);
    const double *r = pt_render + (int64_t)p * 3 * n;
    const double u = i < n ? r[i] : __builtin_nan(""), w = i < n ? r[n + i] : __builtin_nan("");
    const bool ok = fabs(u) < 1.0e9 && fabs(w) < 1.0e9;
    const int x = ok ? (int)u : 0, y = ok ? (int)w : 0;
    int v[4] = {ok ? max(x, -kBoxWhole) : kBoxWhole, ok ? -min(x, kBoxWhole) : kBoxWhole,
                ok ? max(y, -kBoxWhole) : kBoxWhole, ok ? -min(y, kBoxWhole) : kBoxWhole};   // four minima
#pragma unroll
    for (int a = 0; a < 4; ++a)
        for (int off = 32; off > 0; off >>= 1) v[a] = min(v[a], __shfl_xor(v[a], off));
    if ((threadIdx.x & 63) == 0) {
        short4 o;                                     // an empty chunk gets an empty box (x0 > x1)
        o.x = (short)max(v[0] - radius, -kBoxWhole);
        o.y = (short)min(-v[1] + radius, kBoxWhole);
        o.z = (short)max(v[2] - radius, -kBoxWhole);
        o.w = (short)min(-v[3] + radius, kBoxWhole);
        box[(int64_t)p * nchunks + c] = o;
        if (o.x <= o.y) {                             // not an empty chunk
            atomicMin(&pose_box[p].x, (int)o.x);
            atomicMin(&pose_box[p].y, -(int)o.y);
            atomicMin(&pose_box[p].z, (int)o.z);
            atomicMin(&pose_box[p].w, -(int)o.w);
        }
    }
}

// One point of the current pose in two halves, so that the loads of several chunk visits can be in flight
// before the first is needed: zb_load fetches the three numbers (model coordinates, or u, v, z of pt_render),
// zb_pixel gives the pixel (int() of the coordinates), the depth and whether the point can take part at all
// (finite pixel, finite positive depth: the rule of zbuf_pixel in splat.hip without the image bounds).
template <bool FROM_RENDER>
__device__ __forceinline__ void zb_load(const double *__restrict__ src, int64_t n, uint32_t i, double (&a)[3]) {
    if (FROM_RENDER) {
        a[0] = src[i];
        a[1] = src[n + i];
        a[2] = src[2 * n + i];
    } else {
        const double *q = src + (size_t)i * 3;
        a[0] = q[0];
        a[1] = q[1];
        a[2] = q[2];
    }
}
template <bool FROM_RENDER>
__device__ __forceinline__ bool zb_pixel(const double (&a)[3], const double (&P)[12], int &x, int &y, double &z) {
    double u, v;
    if (FROM_RENDER) {
        u = a[0];
        v = a[1];
        z = a[2];
    } else {
        ispa_geom::project_point(P, a[0], a[1], a[2], u, v, z);
    }
    // int() truncates; |trunc(u)| < 1e9 <=> |u| < 1e9; NaN fails every comparison
    const bool ok = fabs(u) < 1.0e9 && fabs(v) < 1.0e9 && z > 0.0 && z < __builtin_inf();
    x = ispa_geom::cvt_i32_sat(u);                    // defined for the lanes that `ok` rejects, too
    y = ispa_geom::cvt_i32_sat(v);
    return ok;
}
// depth alone (pass 2 on a cached pixel): the third row of the same fma chain
template <bool FROM_RENDER>
__device__ __forceinline__ double zb_depth(const double (&a)[3], const double (&P)[12]) {
    if (FROM_RENDER) return a[2];
    double acc = P[8] * a[0];
    acc = fma(P[9], a[1], acc);
    acc = fma(P[10], a[2], acc);
    return fma(P[11], 1.0, acc);
}

__device__ __forceinline__ bool zb_hit(const short4 b, int tx0, int ty0, int tx1, int ty1) {
    return b.x <= tx1 && b.y >= tx0 && b.z <= ty1 && b.w >= ty0;
}

// The next segment of the visiting-chunk list: up to kRounds rounds of kZThreads boxes, as many as fit into
// `list` (first counted, then written; one round always fits).  `pre` holds the calling thread's boxes of the
// first kRounds rounds from c0 = 0 when `use_pre` is set (fetched while the previous item was worked on).  A
// wave reserves room for its hits with one LDS atomic, so the order of the list depends on wave timing (the
// results never do: minima commute).  cnt[0..1] must be 0 on entry and are 0 again on exit; ends with a
// barrier: the list is readable by every wave.
__device__ __forceinline__ int zb_build_list(const short4 *__restrict__ box, int nchunks, int &c0, int tx0, int ty0,
                                             int tx1, int ty1, int *list, int *cnt, const short4 (&pre)[kRounds],
                                             bool use_pre) {
    const int lane = threadIdx.x & 63;
    unsigned long long m[kRounds];
#pragma unroll
    for (int r = 0; r < kRounds; ++r) {
        const int c = c0 + r * kZThreads + threadIdx.x;
        const short4 b = use_pre ? pre[r] : box[min(c, nchunks - 1)];
        m[r] = __builtin_amdgcn_ballot_w64(c < nchunks && zb_hit(b, tx0, ty0, tx1, ty1));
    }
    int rounds = kRounds;
    for (;;) {                                        // uniform: at most log2(kRounds) + 1 turns
        int mine = 0;
#pragma unroll
        for (int r = 0; r < kRounds; ++r) mine += r < rounds ? __popcll(m[r]) : 0;
        if (lane == 0 && mine) atomicAdd(&cnt[0], mine);
        __syncthreads();
        const int total = cnt[0];
        __syncthreads();
        if (threadIdx.x == 0) cnt[0] = 0;             // every wave has read the sum; the next add is behind a barrier
        if (total <= kListMax) {
            int at = 0;
            if (lane == 0 && mine) at = atomicAdd(&cnt[1], mine);
            at = __shfl(at, 0);
#pragma unroll
            for (int r = 0; r < kRounds; ++r) {
                if (r < rounds && ((m[r] >> lane) & 1ull))
                    list[at + __popcll(m[r] & ((1ull << lane) - 1ull))] = c0 + r * kZThreads + threadIdx.x;
                at += r < rounds ? __popcll(m[r]) : 0;
            }
            c0 += rounds * kZThreads;
            __syncthreads();
            if (threadIdx.x == 0) cnt[1] = 0;
            return total;
        }
        rounds >>= 1;                                 // too many hits for the list: take fewer rounds
        __syncthreads();                              // the reset above must land before any wave counts again
    }
}

// One segment of chunk visits.  PASS 1: every point of every visiting chunk is projected; those that reach
// into the tile take an LDS atomicMin per covered pixel; the tile offset of a point that lands IN the tile is
// left in `pixcache` for pass 2 (the first kCacheVisits visits of an item).  PASS 2: a point is visible iff it
// lies within `relax` behind the nearest point of its own pixel; cached visits recompute only the depth.
template <bool FROM_RENDER, int PASS>
__device__ __forceinline__ void zb_visit(const double *__restrict__ psrc, int64_t n, const double (&P)[12], int H, int W,
                                         int radius, double relax, int tx0, int ty0, int tx1, int ty1,
                                         const int *list, int nl, int vbase, unsigned long long *tile,
                                         unsigned short *pixcache, uint8_t *__restrict__ truth_p) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    for (int e0 = wave * kBatch; e0 < nl; e0 += kZWaves * kBatch) {
        // every lane loads for every slot (a slot past the list, or a point past the end, re-reads point 0 and
        // is discarded): loads under a lane mask would make the compiler wait for each one before it issues
        // the next
        double a[kBatch][3];
        uint32_t idx[kBatch];
        bool live[kBatch];
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            idx[b] = (uint32_t)list[min(e0 + b, nl - 1)] * kChunk + lane;
            live[b] = e0 + b < nl && (int64_t)idx[b] < n;
            zb_load<FROM_RENDER>(psrc, n, live[b] ? idx[b] : 0u, a[b]);
        }
#pragma unroll
        for (int b = 0; b < kBatch; ++b) {
            const int visit = vbase + e0 + b;
            int x = 0, y = 0;
            double z;
            unsigned pix;
            bool ok = false;
            if (PASS == 2 && visit < kCacheVisits) {              // wave-uniform
                pix = live[b] ? pixcache[visit * kChunk + lane] : kNoPixel;
                z = zb_depth<FROM_RENDER>(a[b], P);
            } else {
                ok = zb_pixel<FROM_RENDER>(a[b], P, x, y, z) && live[b];
                const bool own = ok && x >= tx0 && x <= tx1 && y >= ty0 && y <= ty1;
                pix = own ? (unsigned)((y - ty0) * kTileW + (x - tx0)) : kNoPixel;
            }
            if (PASS == 1) {
                const unsigned long long key =
                    ((unsigned long long)__float_as_uint((float)z) << 32) | (unsigned long long)idx[b];
                if (radius == 0) {
                    if (pix != kNoPixel) atomicMin(&tile[pix], key);
                } else if (ok && x >= 0 && x < W && y >= 0 && y < H) {
                    const int y0 = max(y - radius, ty0), y1 = min(y + radius, ty1);
                    const int x0 = max(x - radius, tx0), x1 = min(x + radius, tx1);
                    for (int yy = y0; yy <= y1; ++yy)
                        for (int xx = x0; xx <= x1; ++xx) atomicMin(&tile[(yy - ty0) * kTileW + (xx - tx0)], key);
                }
                if (e0 + b < nl && visit < kCacheVisits) pixcache[visit * kChunk + lane] = (unsigned short)pix;
            } else if (pix != kNoPixel) {
                const float d = __uint_as_float((uint32_t)(tile[pix] >> 32));
                truth_p[idx[b]] = (z - (double)d) <= relax;
            }
        }
    }
}

#ifdef ZB_STAMP   // development build: shader clocks per phase of the tile kernel, summed over work items
__device__ unsigned long long zb_stamps[16];
#define ZB_T(k) do { if (threadIdx.x == 0) { const unsigned long long t_ = __builtin_amdgcn_s_memtime(); atomicAdd(&zb_stamps[k], t_ - t_last); t_last = t_; } } while (0)
#else
#define ZB_T(k) do { } while (0)
#endif

// Persistent workgroups: each takes (pose, tile) items off a counter until none is left, so that no CU waits
// for a workgroup launch between two tiles and two resident workgroups overlap each other's phases.
template <bool FROM_RENDER>
__global__ __launch_bounds__(kZThreads, kWgPerCu * kZThreads / 256) void zb_tile_kernel(
    const double *__restrict__ src, int64_t n, const double *__restrict__ cams, int n_pose, int H, int W, int radius,
    double relax, const short4 *__restrict__ box, const int4 *__restrict__ pose_box, int nchunks, int tiles_x,
    int tiles_y, int *__restrict__ counter, unsigned long long *__restrict__ packed, int32_t *__restrict__ winner,
    float *__restrict__ depth, uint8_t *__restrict__ truth, int vec_ok) {
    __shared__ __attribute__((aligned(16))) unsigned long long tile[kTileH * kTileW];   // read as 16-byte pairs
    __shared__ int list[kListMax];
    __shared__ int cnt[2];
    __shared__ unsigned short pixcache[kCacheVisits * kChunk];
    const int ntiles = tiles_x * tiles_y;
    const int64_t total_items = (int64_t)n_pose * ntiles;
#ifdef ZB_STAMP
    unsigned long long t_last = __builtin_amdgcn_s_memtime();
#endif
    // Items come off the counter in groups of kGroup consecutive ones, two groups ahead: the counter is asked
    // for the group after next while a group is worked on, so the item after the current one is always known
    // and its pose box and chunk boxes can be fetched a whole item early (under the stores of this kernel a
    // load takes microseconds).  The last items (about kGroup per workgroup) are handed out one by one, so that
    // no workgroup ends the launch with a whole group of expensive tiles while the others are idle: the counter
    // counts groups, the first n_full of them kGroup items long, the rest single items.
    const int64_t tail_items = min(total_items, (int64_t)kGroup * gridDim.x);
    const int n_full = (int)((total_items - tail_items) / kGroup);
    const int n_groups = n_full + (int)(total_items - (int64_t)n_full * kGroup);
    auto group_first = [&](int g) { return g < n_full ? g * kGroup : n_full * kGroup + (g - n_full); };
    auto group_size = [&](int g) { return g < n_full ? kGroup : 1; };
    __shared__ int ahead[2];
    if (threadIdx.x == 0) {
        cnt[0] = cnt[1] = 0;
        ahead[0] = atomicAdd(counter, 1);
        ahead[1] = atomicAdd(counter, 1);
    }
    __syncthreads();
    int group = ahead[0], group_next = ahead[1], in_group = 0;
    __syncthreads();
    int fetched = 0;
    int4 pb_next = make_int4(0, 0, 0, 0);
    short4 pre[kRounds];
    auto prefetch = [&](int it) {                     // pose box and this thread's boxes of item `it`
        const int pp = it < total_items ? it / ntiles : 0;
        pb_next = pose_box[pp];
#pragma unroll
        for (int r = 0; r < kRounds; ++r)
            pre[r] = nchunks > 0 ? box[(int64_t)pp * nchunks + min(r * kZThreads + (int)threadIdx.x, nchunks - 1)]
                                 : make_short4(0, 0, 0, 0);
    };
    prefetch(group < n_groups ? group_first(group) : 0);
    for (;;) {
        if (group >= n_groups) break;                 // group numbers only grow: nothing is left
        const int item = group_first(group) + in_group;
        if (threadIdx.x == 0 && in_group == 0) fetched = atomicAdd(counter, 1);   // the group after next
        const int item_next = in_group + 1 < group_size(group) ? item + 1
                                                               : (group_next < n_groups ? group_first(group_next) : 0);
        const int p = item / ntiles, tl = item - p * ntiles;
        const int tx0 = (tl % tiles_x) * kTileW, ty0 = (tl / tiles_x) * kTileH;
        const int tx1 = min(tx0 + kTileW, W) - 1, ty1 = min(ty0 + kTileH, H) - 1;  // inclusive
        const short4 *pbox = box + (int64_t)p * nchunks;
        const double *psrc = FROM_RENDER ? src + (int64_t)p * 3 * n : src;
        uint8_t *truth_p = truth ? truth + (int64_t)p * n : nullptr;
        // most tiles of a pose are background: the pose's own box (x0, -x1, y0, -y1) settles those at once
        const int4 pb = pb_next;
        bool any = nchunks > 0 && pb.x <= tx1 && -pb.y >= tx0 && pb.z <= ty1 && -pb.w >= ty0;
        int c0 = 0, nl = 0;
        if (any) {
            nl = zb_build_list(pbox, nchunks, c0, tx0, ty0, tx1, ty1, list, cnt, pre, true);
            any = nl > 0 || c0 < nchunks;
        }
        prefetch(item_next);
        ZB_T(any ? 0 : 8);
        if (any) {
            // the camera in vector registers: left to the compiler it is re-fetched by scalar loads (and
            // waited for) in front of every projection
            double P[12];
#pragma unroll
            for (int k = 0; k < 12; ++k) {
                P[k] = FROM_RENDER ? 0.0 : cams[(int64_t)p * 12 + k];
                asm volatile("" : "+v"(P[k]));
            }
            {
                const u64x2 empty = {kEmptyKey, kEmptyKey};
                for (int e = threadIdx.x; e < kTileH * kTileW / 2; e += kZThreads) reinterpret_cast<u64x2 *>(tile)[e] = empty;
            }
            __syncthreads();
            ZB_T(1);
            int vbase = 0, segments = 0;
            for (;;) {
                zb_visit<FROM_RENDER, 1>(psrc, n, P, H, W, radius, relax, tx0, ty0, tx1, ty1, list, nl, vbase, tile,
                                         pixcache, truth_p);
                vbase += nl;
                ++segments;
                if (c0 >= nchunks) break;
                nl = zb_build_list(pbox, nchunks, c0, tx0, ty0, tx1, ty1, list, cnt, pre, false);
            }
            __syncthreads();
            ZB_T(2);
#ifdef ZB_STAMP
            if (threadIdx.x == 0) {
                atomicAdd(&zb_stamps[6], 1ull);
                atomicAdd(&zb_stamps[7], (unsigned long long)vbase);
            }
#endif
            if (truth) {
                if (segments == 1) {                  // the list of pass 1 is still there
                    zb_visit<FROM_RENDER, 2>(psrc, n, P, H, W, radius, relax, tx0, ty0, tx1, ty1, list, nl, 0, tile,
                                             pixcache, truth_p);
                } else {                              // lists are rebuilt, maybe in another order: no cache
                    c0 = 0;
                    vbase = kCacheVisits;
                    while (c0 < nchunks) {
                        nl = zb_build_list(pbox, nchunks, c0, tx0, ty0, tx1, ty1, list, cnt, pre, false);
                        zb_visit<FROM_RENDER, 2>(psrc, n, P, H, W, radius, relax, tx0, ty0, tx1, ty1, list, nl,
                                                 vbase, tile, pixcache, truth_p);
                        vbase += nl;
                    }
                }
            }
            ZB_T(3);
        }
        // the tile goes to HBM once (pairs of pixels: 16-byte stores of keys, 8-byte of winners / depths)
        const int64_t img = (int64_t)p * H * W;
        const int tw = tx1 - tx0 + 1, th = ty1 - ty0 + 1;
        if (vec_ok && kZThreads % (tw >> 1) == 0) {
            // A thread owns one pair of pixel columns and every (kZThreads / pairs)-th row: the LDS offset and the
            // image offset advance by constants, two keys leave LDS as one 16-byte read (W is even here, so tw
            // is; tiles are 128 or, at the right border, W % 128 pixels wide: a power of two for the usual sizes).
            const int pairs = tw >> 1, rows_per = kZThreads / pairs;
            const int lx = ((int)threadIdx.x % pairs) * 2, ly0 = (int)threadIdx.x / pairs;
            const unsigned long long *src = tile + ly0 * kTileW + lx;
            int64_t o = img + (int64_t)(ty0 + ly0) * W + tx0 + lx;
            const int64_t o_step = (int64_t)rows_per * W;
            const int s_step = rows_per * kTileW;
            const bool only_packed = packed && !winner && !depth;
            if (only_packed) {                                   // the common case, nothing else in the loop
                const u64x2 empty = {kEmptyKey, kEmptyKey};
                for (int ly = ly0; ly < th; ly += rows_per, src += s_step, o += o_step) {
                    const u64x2 kk = any ? *reinterpret_cast<const u64x2 *>(src) : empty;
                    __builtin_nontemporal_store(kk, reinterpret_cast<u64x2 *>(packed + o));
                }
            } else {
                for (int ly = ly0; ly < th; ly += rows_per, src += s_step, o += o_step) {
                    const unsigned long long k0 = any ? src[0] : kEmptyKey, k1 = any ? src[1] : kEmptyKey;
                    if (packed) {
                        const u64x2 kk = {k0, k1};
                        __builtin_nontemporal_store(kk, reinterpret_cast<u64x2 *>(packed + o));
                    }
                    if (winner) {
                        const i32x2 ww = {k0 == kEmptyKey ? -1 : (int)(uint32_t)k0, k1 == kEmptyKey ? -1 : (int)(uint32_t)k1};
                        __builtin_nontemporal_store(ww, reinterpret_cast<i32x2 *>(winner + o));
                    }
                    if (depth) {
                        const f32x2 dd = {k0 == kEmptyKey ? __builtin_inff() : __uint_as_float((uint32_t)(k0 >> 32)),
                                          k1 == kEmptyKey ? __builtin_inff() : __uint_as_float((uint32_t)(k1 >> 32))};
                        __builtin_nontemporal_store(dd, reinterpret_cast<f32x2 *>(depth + o));
                    }
                }
            }
        } else {
            for (int e = threadIdx.x; e < th * tw; e += kZThreads) {
                const int ly = e / tw, lx = e - ly * tw;
                const unsigned long long k0 = any ? tile[ly * kTileW + lx] : kEmptyKey;
                const int64_t o = img + (int64_t)(ty0 + ly) * W + tx0 + lx;
                if (packed) packed[o] = k0;
                if (winner) winner[o] = k0 == kEmptyKey ? -1 : (int)(uint32_t)k0;
                if (depth) depth[o] = k0 == kEmptyKey ? __builtin_inff() : __uint_as_float((uint32_t)(k0 >> 32));
            }
        }
        ZB_T(any ? 4 : 9);
        __syncthreads();                              // the tile has been read: the next item may reset it
        if (++in_group == group_size(group)) {
            if (threadIdx.x == 0) ahead[0] = fetched;
            __syncthreads();
            group = group_next;
            group_next = ahead[0];
            in_group = 0;
        }
    }
}

struct ZbPlan {
    int nchunks, tiles_x, tiles_y;
    size_t off_aabb, off_box, off_pose, off_counter, total;
};

ZbPlan zb_plan(int64_t n, int n_pose, int H, int W) {
    ZbPlan p;
    p.nchunks = (int)((n + kChunk - 1) / kChunk);
    p.tiles_x = (W + kTileW - 1) / kTileW;
    p.tiles_y = (H + kTileH - 1) / kTileH;
    auto up = [](size_t x) { return (x + 255) / 256 * 256; };
    p.off_aabb = 0;
    p.off_box = up((size_t)p.nchunks * 6 * sizeof(double));
    p.off_pose = p.off_box + up((size_t)n_pose * p.nchunks * sizeof(short4));
    p.off_counter = p.off_pose + up((size_t)n_pose * sizeof(int4));
    p.total = p.off_counter + 256;
    return p;
}

int zb_check(int64_t n, int n_pose, int H, int W, int radius) {
    ISPA_REQUIRE(n >= 0 && n < (1ll << 32) && n_pose >= 0, "bad point count");
    ISPA_REQUIRE(H > 0 && W > 0 && H <= 16384 && W <= 16384 && radius >= 0 && radius <= 64, "bad shape/radius");
    ISPA_REQUIRE(n_pose <= 65535, "too many poses per call (%d)", n_pose);
    return ISPA_OK;
}

// Workspace words that must start out as "no box yet" / zero: done in front of the box kernels.
int zb_prepare(char *ws, const ZbPlan &pl, int n_pose, hipStream_t s) {
    ISPA_HIP(hipMemsetAsync(ws + pl.off_pose, 0x7f, (size_t)n_pose * sizeof(int4), s));
    ISPA_HIP(hipMemsetAsync(ws + pl.off_counter, 0, 256, s));                       // the item counter
    return ISPA_OK;
}

template <bool FROM_RENDER>
int zb_launch(const double *src, int64_t n, const double *cams, int n_pose, int H, int W, int radius, double relax,
              const ZbPlan &pl, char *ws, uint64_t *packed, int32_t *winner, float *depth, uint8_t *truth,
              hipStream_t s) {
    // 16-byte key stores (8-byte winner / depth stores) need an even width and aligned images
    const int vec_ok = W % 2 == 0 && (uintptr_t)packed % 16 == 0 && (uintptr_t)winner % 8 == 0 && (uintptr_t)depth % 8 == 0;
    const int64_t items = (int64_t)n_pose * pl.tiles_x * pl.tiles_y;
    ISPA_REQUIRE(items < (1ll << 31) - 4096, "too many (pose, tile) items");
    int dev = 0, cus = 0;
    ISPA_HIP(hipGetDevice(&dev));
    ISPA_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    const int64_t resident = (int64_t)cus * kWgPerCu;                                     // two workgroups per CU (LDS)
    const unsigned grid = (unsigned)(items < resident ? items : resident);
    const ispa::TimingSlot timed = ispa::timing_begin(s);
    hipLaunchKernelGGL(zb_tile_kernel<FROM_RENDER>, dim3(grid), dim3(kZThreads), 0, s, src, n, cams, n_pose, H, W, radius,
                       relax, (const short4 *)(ws + pl.off_box), (const int4 *)(ws + pl.off_pose), pl.nchunks,
                       pl.tiles_x, pl.tiles_y, (int *)(ws + pl.off_counter), (unsigned long long *)packed, winner,
                       depth, truth, vec_ok);
    ispa::timing_end(timed, s);
    return ispa::check_launch("zb_tile_kernel");
}

}  // namespace

extern "C" size_t ispa_project_zbuffer_workspace_bytes(int64_t n, int n_pose, int H, int W) {
    if (n < 0 || n_pose < 0 || H <= 0 || W <= 0) return 0;
    return zb_plan(n, n_pose, H, W).total;
}

extern "C" int ispa_project_zbuffer(const double *pts, int64_t n, const double *cams, int n_pose, int H, int W,
                                    int radius, double relax, uint64_t *packed, int32_t *winner, float *depth,
                                    uint8_t *truth, void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = zb_check(n, n_pose, H, W, radius)) return rc;
    if (n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(cams && (n == 0 || pts), "null pointer");
    ISPA_REQUIRE(packed || winner || depth || truth, "no output requested");
    const ZbPlan pl = zb_plan(n, n_pose, H, W);
    if (workspace_bytes < pl.total)
        return ispa::fail(ISPA_EWORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, pl.total);
    ISPA_REQUIRE(workspace && (uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    double *aabb = (double *)(ws + pl.off_aabb);
    if (truth && n > 0) ISPA_HIP(hipMemsetAsync(truth, 0, (size_t)n_pose * n, s));   // points off every tile
    if (int rc = zb_prepare(ws, pl, n_pose, s)) return rc;
    if (pl.nchunks > 0) {
        hipLaunchKernelGGL(zb_chunk_aabb_kernel, dim3(ispa::blocks_for(pl.nchunks, 4)), dim3(256), 0, s, pts, n,
                           pl.nchunks, aabb);
        hipLaunchKernelGGL(zb_chunk_box_kernel, dim3(ispa::blocks_for(pl.nchunks, 256), n_pose), dim3(256), 0, s, aabb,
                           pl.nchunks, cams, radius, (short4 *)(ws + pl.off_box), (int4 *)(ws + pl.off_pose));
        if (int rc = ispa::check_launch("zb_chunk_box_kernel")) return rc;
    }
    return zb_launch<false>(pts, n, cams, n_pose, H, W, radius, relax, pl, ws, packed, winner, depth, truth, s);
}

extern "C" int ispa_zbuffer_tiled(const double *pt_render, int64_t n, int n_pose, int H, int W, int radius,
                                  double relax, uint64_t *packed, int32_t *winner, float *depth, uint8_t *truth,
                                  void *workspace, size_t workspace_bytes, void *stream) {
    if (int rc = zb_check(n, n_pose, H, W, radius)) return rc;
    if (n_pose == 0) return ISPA_OK;
    ISPA_REQUIRE(n == 0 || pt_render, "null pointer");
    ISPA_REQUIRE(packed || winner || depth || truth, "no output requested");
    const ZbPlan pl = zb_plan(n, n_pose, H, W);
    if (workspace_bytes < pl.total)
        return ispa::fail(ISPA_EWORKSPACE, "workspace of %zu bytes given, %zu needed", workspace_bytes, pl.total);
    ISPA_REQUIRE(workspace && (uintptr_t)workspace % 256 == 0, "workspace must be 256-byte aligned");
    hipStream_t s = (hipStream_t)stream;
    char *ws = (char *)workspace;
    if (truth && n > 0) ISPA_HIP(hipMemsetAsync(truth, 0, (size_t)n_pose * n, s));
    if (int rc = zb_prepare(ws, pl, n_pose, s)) return rc;
    if (pl.nchunks > 0) {
        hipLaunchKernelGGL(zb_chunk_box_render_kernel, dim3(ispa::blocks_for(pl.nchunks, 4), n_pose), dim3(256), 0, s,
                           pt_render, n, pl.nchunks, radius, (short4 *)(ws + pl.off_box), (int4 *)(ws + pl.off_pose));
        if (int rc = ispa::check_launch("zb_chunk_box_render_kernel")) return rc;
    }
    return zb_launch<true>(pt_render, n, nullptr, n_pose, H, W, radius, relax, pl, ws, packed, winner, depth, truth, s);
}

#ifdef ZB_STAMP
extern "C" int ispa_dev_zb_stamps(unsigned long long *host16, int reset) {
    ISPA_HIP(hipMemcpyFromSymbol(host16, HIP_SYMBOL(zb_stamps), 16 * sizeof(unsigned long long)));
    if (reset) {
        unsigned long long z[16] = {0};
        ISPA_HIP(hipMemcpyToSymbol(HIP_SYMBOL(zb_stamps), z, sizeof(z)));
    }
    return ISPA_OK;
}
#endif
