// api_internal.hpp -- helpers shared by the C-ABI translation units (api.hip, train_api.hip).
#pragma once
#include "../../include/cellscreen.h"
#include "common.hpp"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace cs {

int fail(int code, const char* fmt, ...);      // records the thread-local message, returns code
const char* last_error_cstr();

#define HIPCHK(expr)                                                                         \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess)                                                               \
            return cs::fail(CS_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), \
                            __FILE__, __LINE__);                                             \
    } while (0)

// ---- the reference graph (CAE_improved_modeltrain.py:184-229) ---------------------------
static const int kRefChannels[7] = {32, 64, 32, 32, 64, 32, 1};
static const int kNConv = 7, kNEnc = 3, kH = 64, kW = 64;
static const int kConvGrid[7] = {64, 32, 16, 8, 16, 32, 64};      // conv grid (= pre-pool output) side
// stored per-cell size (floats) of each conv's output tensor (after pool / before upsample)
static const size_t kLayerFloats[7] = {32 * 32 * 32, 16 * 16 * 64, 8 * 8 * 32, 8 * 8 * 32,
                                       16 * 16 * 64, 32 * 32 * 32, 64 * 64};
// conv MACs per cell, SURVEY.md Appendix A.1
static const double kLayerMacs[7] = {1179648, 18874368, 4718592, 589824, 4718592, 18874368, 1179648};

struct DevBuf {
    void* p = nullptr;
    size_t bytes = 0;
    DevBuf() = default;
    DevBuf(const DevBuf&) = delete;
    DevBuf& operator=(const DevBuf&) = delete;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int ensure(size_t need)
    {
        if (need <= bytes) return CS_OK;
        if (p) { (void)hipFree(p); p = nullptr; bytes = 0; }
        hipError_t e = hipMalloc(&p, need);
        if (e != hipSuccess) { p = nullptr; return fail(CS_ERR_NOMEM, "hipMalloc(%zu) failed: %s", need, hipGetErrorString(e)); }
        bytes = need;
        return CS_OK;
    }
    template <class T> T* as() const { return (T*)p; }
};

// Orders `mine` (a handle's private stream) after everything enqueued so far on `other` (the caller's stream:
// torch's current stream, an RCCL stream, ...): the cs_*_wait_stream entry points.  other == nullptr is the
// legacy default stream, with which a hipStreamNonBlocking stream is NOT implicitly ordered either.
inline int wait_on_stream(hipStream_t mine, void* other)
{
    hipEvent_t ev = nullptr;
    HIPCHK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    hipError_t e = hipEventRecord(ev, (hipStream_t)other);
    if (e == hipSuccess) e = hipStreamWaitEvent(mine, ev, 0);
    (void)hipEventDestroy(ev);                 // released once the recorded work has completed
    if (e != hipSuccess) return fail(CS_ERR_HIP, "ordering the handle's stream after the caller's failed: %s", hipGetErrorString(e));
    return CS_OK;
}

int upload(DevBuf& d, const void* src, size_t bytes);
int check_arch(const cs_cae_weights* w, int expect_convs, const char* what);
int require_gfx950(int device_id);

}  // namespace cs
