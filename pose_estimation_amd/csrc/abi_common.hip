// ABI version and thread-local error text.
#include "common.h"

namespace ispa {
char *last_error_buffer() {
    static thread_local char buf[512] = "";
    return buf;
}
}  // namespace ispa

namespace ispa {
namespace {
struct Timing {
    int capacity = 0, used = 0;
    hipEvent_t *start = nullptr, *stop = nullptr;
} g_timing;
}  // namespace

TimingSlot timing_begin(hipStream_t stream) {
    TimingSlot s{g_timing.used < g_timing.capacity};
    if (s.armed) (void)hipEventRecord(g_timing.start[g_timing.used], stream);
    return s;
}

void timing_end(const TimingSlot &slot, hipStream_t stream) {
    if (slot.armed) (void)hipEventRecord(g_timing.stop[g_timing.used++], stream);
}
}  // namespace ispa

extern "C" int ispa_timing_enable(int max_records) {
    using ispa::g_timing;
    ISPA_REQUIRE(max_records >= 0 && max_records <= 65536, "max_records out of range");
    for (int i = 0; i < g_timing.capacity; ++i) {
        (void)hipEventDestroy(g_timing.start[i]);
        (void)hipEventDestroy(g_timing.stop[i]);
    }
    delete[] g_timing.start;
    delete[] g_timing.stop;
    g_timing = ispa::Timing();
    if (max_records == 0) return ISPA_OK;
    g_timing.start = new hipEvent_t[max_records];
    g_timing.stop = new hipEvent_t[max_records];
    for (int i = 0; i < max_records; ++i) {
        ISPA_HIP(hipEventCreate(&g_timing.start[i]));
        ISPA_HIP(hipEventCreate(&g_timing.stop[i]));
    }
    g_timing.capacity = max_records;
    return ISPA_OK;
}

extern "C" int ispa_timing_collect(float *host_ms, int capacity, int *host_count) {
    using ispa::g_timing;
    ISPA_REQUIRE(host_ms && host_count && capacity >= 0, "null pointer");
    int n = g_timing.used < capacity ? g_timing.used : capacity;
    for (int i = 0; i < n; ++i) {
        ISPA_HIP(hipEventSynchronize(g_timing.stop[i]));
        ISPA_HIP(hipEventElapsedTime(&host_ms[i], g_timing.start[i], g_timing.stop[i]));
    }
    *host_count = n;
    g_timing.used = 0;
    return ISPA_OK;
}

extern "C" int ispa_abi_version(void) { return ISPA_ABI_VERSION; }
extern "C" const char *ispa_last_error(void) { return ispa::last_error_buffer(); }
