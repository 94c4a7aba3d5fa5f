// Shared host-side helpers for libispa_hip.so (error reporting, argument checks, launch sizing).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "ispa_abi.h"

namespace ispa {

char *last_error_buffer();  // thread-local, 512 bytes

inline int fail(int code, const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(last_error_buffer(), 512, fmt, ap);
    va_end(ap);
    return code;
}

inline int check_launch(const char *what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return fail(ISPA_EHIP, "%s: %s", what, hipGetErrorString(e));
    return ISPA_OK;
}

#define ISPA_REQUIRE(cond, ...)                                   \
    do {                                                          \
        if (!(cond)) return ::ispa::fail(ISPA_EINVAL, __VA_ARGS__); \
    } while (0)

#define ISPA_HIP(call)                                                                   \
    do {                                                                                 \
        hipError_t e__ = (call);                                                         \
        if (e__ != hipSuccess)                                                           \
            return ::ispa::fail(ISPA_EHIP, "%s failed: %s", #call, hipGetErrorString(e__)); \
    } while (0)

// Optional timing of the dominant kernel of a call with HIP events on the call's stream
// (ispa_timing_enable / ispa_timing_collect).  begin/end are no-ops while no slot is armed.
struct TimingSlot {
    bool armed;
};
TimingSlot timing_begin(hipStream_t stream);
void timing_end(const TimingSlot &slot, hipStream_t stream);

// Kernels that need more than 64 KiB of dynamic LDS must be told so once per device.
inline int allow_dynamic_lds(const void *kernel, int bytes, unsigned long long *done_mask) {
    int dev = 0;
    ISPA_HIP(hipGetDevice(&dev));
    const unsigned long long bit = 1ull << (dev & 63);
    // host threads may race here: setting the attribute twice is harmless, the mask update is atomic
    if (!(__atomic_load_n(done_mask, __ATOMIC_ACQUIRE) & bit)) {
        ISPA_HIP(hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
        __atomic_fetch_or(done_mask, bit, __ATOMIC_RELEASE);
    }
    return ISPA_OK;
}

// One thread per work item: the grid's x dimension holds up to 2^31 - 1 blocks, far beyond any
// size the entry points accept, so nothing is ever silently left unprocessed.
inline unsigned blocks_for(int64_t work, int threads, int64_t cap = 0x7fffffff) {
    int64_t b = (work + threads - 1) / threads;
    if (b < 1) b = 1;
    if (b > cap) b = cap;
    return (unsigned)b;
}

}  // namespace ispa
