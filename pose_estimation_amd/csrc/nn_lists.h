// Dense nearest neighbour: the running top lists the filter kernels keep in registers (included by nn.hip).
#pragma once

namespace {

// Running top list of one query stream.  Interface shared by both implementations:
//   begin_half(id)  start of a half-step (32 reference rows = 16 accumulator registers per lane)
//   push(x, reg)    score of accumulator register `reg` (compile-time after unrolling)
//   end_half()
//   value(s), code(s)  s-th best score and its code = half_step_id * 16 + reg (-1: none)
template <int K>
struct TopList {            // K+1 best scores, descending, each with its code
    float v[K + 1];
    int c[K + 1];
    int base;
    __device__ __forceinline__ void init() {
#pragma unroll
        for (int s = 0; s <= K; ++s) {
            v[s] = kNegInf;
            c[s] = -1;
        }
        base = 0;
    }
    __device__ __forceinline__ void begin_half(int id) { base = id * 16; }
    __device__ __forceinline__ void end_half() {}
    template <typename Acc>
    __device__ __forceinline__ void end_half(const Acc &) {}
    __device__ __forceinline__ void push(float x, int reg) {
        const int code = base + reg;
        if (K == 1) {
            v[1] = __builtin_amdgcn_fmed3f(v[0], v[1], x);
            const bool better = x > v[0];
            c[0] = better ? code : c[0];
            v[0] = better ? x : v[0];
        } else {
            // Most scores beat nobody's list: skip the update when no lane of the wave has a
            // candidate (the chance falls like 64 (K+1) / t with the position t in the stream).
            if (__builtin_amdgcn_ballot_w64(x > v[K]) == 0) return;
            // sorted insert, branch-free: slot s takes its upper neighbour if x goes above that one,
            // x if x goes exactly here, else keeps its entry (x after equal scores: lower code first).
            // Scores: one v_med3 per slot; codes: two selects per slot on the K+1 comparisons.
            bool g[K + 1];
#pragma unroll
            for (int s = 0; s <= K; ++s) g[s] = x > v[s];
            const float xs = g[K] ? x : kNegInf;         // a score that stays out (or a NaN) changes nothing
#pragma unroll
            for (int s = K; s > 0; --s) {
                c[s] = g[s - 1] ? c[s - 1] : (g[s] ? code : c[s]);
                v[s] = __builtin_amdgcn_fmed3f(xs, v[s - 1], v[s]);
            }
            c[0] = g[0] ? code : c[0];
            v[0] = fmaxf(xs, v[0]);
        }
    }
    __device__ __forceinline__ void flush() {}
    __device__ __forceinline__ float value(int s) const { return v[s]; }
    __device__ __forceinline__ int code(int s) const { return v[s] > kAbsent ? c[s] : -1; }
};

// Top-1 (plus the runner-up's score) at three vector instructions per score: the accumulator
// register number replaces the 4 low mantissa bits of the score (a perturbation below 2^-19
// relative, accounted for in the margin test), so the best score carries its own position and
// v_max / v_med3 do all the work; the half-step that produced it is tracked once per 16 scores.
struct TagTop1 {
    float a, b, a_before, pinf;
    int half, cur;
    __device__ __forceinline__ void init() {
        a = b = a_before = kNegInf;
        half = -1;
        cur = -1;
        // +inf in a register the optimiser cannot see through: med3(a, t, +inf) then stays ONE
        // v_med3_f32 instead of a canonicalising v_max pair
        pinf = __builtin_inff();
        asm volatile("" : "+v"(pinf));
    }
    __device__ __forceinline__ void begin_half(int id) {
        cur = id;
        a_before = a;
    }
    __device__ __forceinline__ void push(float x, int reg) {
        const float t = __uint_as_float((__float_as_uint(x) & 0xfffffff0u) | (unsigned)reg);
        b = __builtin_amdgcn_fmed3f(a, b, t);
        // max(a, t) as ONE two-operand instruction (the compiler would add a canonicalising
        // v_max in front of fmaxf because t comes out of integer arithmetic)
        asm("v_max_f32 %0, %1, %2" : "=v"(a) : "v"(a), "v"(t));
    }
    // Two scores at once in five instructions: with m = med3(a, t1, t2) the new best is
    // max3(a, t1, t2) and the new runner-up max(b, m) (b <= a, so whichever of a, t1, t2 is the
    // middle one is the second largest of the three, and only b can still beat it).  Split in two
    // halves so that a caller can place them between different MFMAs.
    float t1, t2, mid;
    __device__ __forceinline__ void push2_begin(float x1, int reg1, float x2, int reg2) {
        t1 = __uint_as_float((__float_as_uint(x1) & 0xfffffff0u) | (unsigned)reg1);
        t2 = __uint_as_float((__float_as_uint(x2) & 0xfffffff0u) | (unsigned)reg2);
        asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid) : "v"(a), "v"(t1), "v"(t2));
    }
    __device__ __forceinline__ void push2_end() {
        asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(t1), "v"(t2));
        asm("v_max_f32 %0, %1, %2" : "=v"(b) : "v"(b), "v"(mid));
    }
    // Four scores in nine instructions, in four slices a caller can place between different MFMAs:
    // with m1 = med3(a, t1, t2), a' = max3(a, t1, t2), m2 = med3(a', t3, t4), a'' = max3(a', t3, t4) the
    // new runner-up is max3(b, m1, m2).
    float mid2;
    __device__ __forceinline__ float tagged(float x, int tag) const {
        return __uint_as_float((__float_as_uint(x) & 0xfffffff0u) | (unsigned)tag);
    }
    template <typename V>
    __device__ __forceinline__ void push4_step(int step, const V &x, int tag0, int off = 0) {
        if (step == 0) {
            t1 = tagged(x[off], tag0);
            t2 = tagged(x[off + 1], tag0 + 1);
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid) : "v"(a), "v"(t1), "v"(t2));
        } else if (step == 1) {
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(t1), "v"(t2));
            t1 = tagged(x[off + 2], tag0 + 2);
        } else if (step == 2) {
            t2 = tagged(x[off + 3], tag0 + 3);
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(mid2) : "v"(a), "v"(t1), "v"(t2));
        } else {
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(a) : "v"(a), "v"(t1), "v"(t2));
            asm("v_max3_f32 %0, %1, %2, %3" : "=v"(b) : "v"(b), "v"(mid), "v"(mid2));
        }
    }
    // x[off .. off+3] with tags tag0 .. tag0+3
    template <typename V>
    __device__ __forceinline__ void push4(const V &x, int tag0, int off = 0) {
#pragma unroll
        for (int step = 0; step < 4; ++step) push4_step(step, x, tag0, off);
    }
    __device__ __forceinline__ void end_half() { half = a != a_before ? cur : half; }
    template <typename Acc>
    __device__ __forceinline__ void end_half(const Acc &) { end_half(); }
    __device__ __forceinline__ void flush() {}
    __device__ __forceinline__ float value(int s) const { return s == 0 ? a : b; }
    __device__ __forceinline__ int code(int s) const {
        return (s == 0 && a > kAbsent) ? half * 16 + (int)(__float_as_uint(a) & 15u) : -1;
    }
};

// Top-k for the bfloat16 filter without a branch per score.  TopList above gates its 24-operation insert
// with a wave ballot; along a stream of a few thousand scores some lane of the wave has a candidate at every
// second score, so the wave pays for the insert most of the time and the branch keeps the vector work
// from overlapping with the MFMAs.  Here the low kTagBits mantissa bits of a score are replaced by its
// position inside an epoch of kEpoch half-steps (half-step within the epoch, accumulator register), so a
// value carries its own code and lists of VALUES only (one v_med3 per slot) do the work:
//   per score    the best three of the current half-step's 16 scores (g1 >= g2 >= g3): 4 operations;
//   per half-step  if no lane's g3 beats what its lists already hold as (K+1)-th best, no lane can have more
//                than two scores of this half-step in its final list, and g1, g2 are inserted into the epoch
//                list (6 operations each); otherwise (the first few hundred scores of a stream, rarely later)
//                all 16 scores are inserted;
//   per epoch    the entries of the epoch list that still matter are merged, with their codes, into the
//                stream's running list (the gated insert, at most K + 1 times), and the epoch list starts
//                empty again.
// The perturbation (below 2^-(23 - kTagBits) relative) is part of the margin the finalize pass tests.
constexpr int kEpoch = 8, kTagBits = 7;                   // 8 half-steps x 16 registers = 128 positions
#ifndef ISPA_TOPK_GROUP
#define ISPA_TOPK_GROUP 1
#endif
template <int K>
struct TagTopK {
    float v[K + 1];                                       // this epoch's best, tagged, descending
    float g1, g2, g3;                                     // this half-step's best three, tagged
    TopList<K> run;                                       // best of the finished epochs, with codes
    int tag_base, cur;
    unsigned keep;
    __device__ __forceinline__ void clear() {
#pragma unroll
        for (int s = 0; s <= K; ++s) v[s] = kNegInf;
    }
    __device__ __forceinline__ void init() {
        clear();
        run.init();
        tag_base = 0;
        cur = 0;
        g1 = g2 = g3 = kNegInf;
        keep = ~((1u << kTagBits) - 1u);
        // the mask lives in a vector register: v_and_or_b32 may read one scalar operand, and that is the tag
        asm volatile("" : "+v"(keep));
    }
    // merges the epoch whose first half-step is `first` into the running list
    __device__ __forceinline__ void commit(int first) {
#pragma unroll
        for (int s = 0; s <= K; ++s) {
            if (__builtin_amdgcn_ballot_w64(v[s] > run.v[K]) == 0) break;   // sorted: the rest is smaller still
            const int tag = (int)(__float_as_uint(v[s]) & ((1u << kTagBits) - 1u));
            run.begin_half(first + (tag >> 4));
            run.push(v[s], tag & 15);
        }
        clear();
    }
    __device__ __forceinline__ void begin_half(int id) {
        if ((id & (kEpoch - 1)) == 0) commit(id - kEpoch);
        cur = id;
        tag_base = (id & (kEpoch - 1)) << 4;
        g1 = g2 = g3 = kNegInf;
    }
    __device__ __forceinline__ float tagged(float x, int reg) const {
        return __uint_as_float((__float_as_uint(x) & keep) | (unsigned)(tag_base + reg));
    }
    __device__ __forceinline__ void insert(float t) {
#pragma unroll
        for (int s = K; s > 0; --s) asm("v_med3_f32 %0, %1, %2, %3" : "=v"(v[s]) : "v"(t), "v"(v[s - 1]), "v"(v[s]));
        asm("v_max_f32 %0, %1, %2" : "=v"(v[0]) : "v"(t), "v"(v[0]));
    }
    __device__ __forceinline__ void push(float x, int reg) {
        const float t = tagged(x, reg);
        if (ISPA_TOPK_GROUP) {
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(g3) : "v"(t), "v"(g2), "v"(g3));
            asm("v_med3_f32 %0, %1, %2, %3" : "=v"(g2) : "v"(t), "v"(g1), "v"(g2));
            asm("v_max_f32 %0, %1, %2" : "=v"(g1) : "v"(t), "v"(g1));
        } else {
            insert(t);
        }
    }
    // `scores`: the 16 accumulator registers pushed since begin_half
    template <typename Acc>
    __device__ __forceinline__ void end_half(const Acc &scores) {
        if (!ISPA_TOPK_GROUP) return;
        const float have = fmaxf(v[K], run.v[K]);         // nothing at or below it can end up in the list
        if (__builtin_amdgcn_ballot_w64(g3 > have) == 0) {
            insert(g1);
            insert(g2);
        } else {
#pragma unroll
            for (int reg = 0; reg < 16; ++reg) insert(tagged(scores[reg], reg));
        }
    }
    __device__ __forceinline__ void flush() { commit(cur & ~(kEpoch - 1)); }
    __device__ __forceinline__ float value(int s) const { return run.value(s); }
    __device__ __forceinline__ int code(int s) const { return run.code(s); }
};

template <int K, bool TAGGED>
struct ListFor {
    typedef TopList<K> type;
};
template <int K>
struct ListFor<K, true> {           // top-k: the tagged value lists (defined above for the bfloat16 filter)
    typedef TagTopK<K> type;
};
template <>
struct ListFor<1, true> {
    typedef TagTop1 type;
};
// list of the bfloat16 filter: tagged for every k
template <int K, int KS>
struct BfListFor {
    typedef TagTopK<K> type;
};
template <int KS>
struct BfListFor<1, KS> {
    typedef TagTop1 type;
};

}  // namespace
