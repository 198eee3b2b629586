// ABI bookkeeping entry points.
#include "common.h"

extern "C" int vhap_abi_version(void) {
    VHAP_ENTER(); return VHAP_ABI_VERSION; }

extern "C" const char* vhap_strerror(int code) {
    switch (code) {
        case VHAP_OK: return "ok";
        case VHAP_E_NULLPTR: return "required pointer is NULL";
        case VHAP_E_BADDIM: return "dimension out of range";
        case VHAP_E_WORKSPACE: return "workspace too small";
        case VHAP_E_HIP: return "HIP runtime / launch failure";
        case VHAP_E_UNSUPPORTED: return "unsupported configuration";
        default: return "unknown error";
    }
}

extern "C" int vhap_stream_create(vhap_stream_t* stream, int high_priority) {
    VHAP_ENTER();
    if (!stream) return VHAP_E_NULLPTR;
    int least = 0, greatest = 0;
    if (hipDeviceGetStreamPriorityRange(&least, &greatest) != hipSuccess) return VHAP_E_HIP;
    hipStream_t st = nullptr;
    if (hipStreamCreateWithPriority(&st, hipStreamNonBlocking, high_priority ? greatest : 0) != hipSuccess) return VHAP_E_HIP;   // 0 = normal (not `least`)
    *stream = static_cast<vhap_stream_t>(st);
    return VHAP_OK;
}

extern "C" int vhap_stream_destroy(vhap_stream_t stream) {
    VHAP_ENTER();
    if (!stream) return VHAP_E_NULLPTR;
    return hipStreamDestroy(static_cast<hipStream_t>(stream)) == hipSuccess ? VHAP_OK : VHAP_E_HIP;
}

// Profiling / A-B switches (not part of the stable ABI): 16 = strided row order in the rasteriser, 32 = per-pixel texture backward
// thread-local: the calling (profiling) thread only; not part of the stable ABI (not declared in vhap_hip.h)
thread_local int vhap_g_debug_flags = 0;
extern "C" void vhap_debug_set_flags(int flags) { vhap_g_debug_flags = flags; }

// Calibration helpers for PMC-based traffic measurements (tools/ri_fwd_pmc.py): known byte counts through the library's own
// streaming kernels.  Not part of the stable ABI (not declared in vhap_hip.h).
extern "C" int vhap_debug_fill(void* p, size_t bytes, vhap_stream_t stream) {
    VHAP_ENTER();
    vhap_zero_async(p, bytes, vhap_stream(stream));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
extern "C" int vhap_debug_copy(void* dst, const void* src, size_t bytes, vhap_stream_t stream) {
    VHAP_ENTER();
    vhap_copy_async(dst, src, bytes, vhap_stream(stream));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
