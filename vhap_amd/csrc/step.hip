// Glue kernels of the native fit step (vhap_amd/step.py): the handful of scalar operations that remain between the fused stages
// once torch autograd no longer chains them -- assembling the energy from the stage accumulators (tracker.py:692-750), the
// photometric normaliser and its upstream gradient (tracker.py:430-439), the sum over frames of the canonical-vertex gradient
// for the shared static offset, and the focal-length gradient (tracker.py:141-157).  Each is one tiny launch instead of
// 5-15 elementwise / reduction launches of the autograd formulation.
#include "common.h"
#include "energy_common.h"

namespace {

__global__ void energy_finalize_kernel(const float* __restrict__ frame_terms, const float* __restrict__ lmk, const float* __restrict__ tex_terms,
                                       const float* __restrict__ off_terms, const unsigned* __restrict__ shade_stats, float w_lmk,
                                       float w_reg_diffuse, float npix, float* __restrict__ log) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    vhap_energy::finalize(frame_terms, lmk, tex_terms, off_terms, shade_stats, w_lmk, w_reg_diffuse, npix, log);
}

__global__ void energy_total_kernel(float* __restrict__ log, const float* __restrict__ photo2, const float* __restrict__ n_global,
                                    float w_photo, float world, float* __restrict__ d_sum, const unsigned* __restrict__ shade_stats,
                                    float* __restrict__ gmax_bound) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    vhap_energy::total(log, photo2 != nullptr, photo2 ? photo2[0] : 0.f, photo2 ? n_global[0] : 1.f, w_photo, world, d_sum, shade_stats, gmax_bound);
}

__global__ __launch_bounds__(256) void sum_frames_kernel(const float* __restrict__ x, int B, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; b++) s += x[(size_t)b * n + i];
    out[i] += s;
}

}  // namespace

extern "C" int vhap_energy_finalize(const float* frame_terms, const float* lmk_energy, const float* tex_terms, const float* off_terms,
                                    const float* shade_stats, float w_landmark, float w_reg_diffuse, int B, int H, int W, float* log,
                                    vhap_stream_t stream) {
    VHAP_ENTER();
    if (!log) return VHAP_E_NULLPTR;
    energy_finalize_kernel<<<1, 64, 0, vhap_stream(stream)>>>(frame_terms, lmk_energy, tex_terms, off_terms,
                                                             reinterpret_cast<const unsigned*>(shade_stats), w_landmark, w_reg_diffuse,
                                                             (float)B * (float)H * (float)W, log);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_energy_total(float* log, const float* photo2, const float* n_global, float w_photo, int world_size, float* d_sum,
                                 vhap_stream_t stream) {
    VHAP_ENTER();
    if (!log || (photo2 && !n_global)) return VHAP_E_NULLPTR;
    energy_total_kernel<<<1, 64, 0, vhap_stream(stream)>>>(log, photo2, n_global, w_photo, (float)world_size, d_sum, nullptr, nullptr);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_energy_total_bound(float* log, const float* photo2, const float* n_global, float w_photo, int world_size, float* d_sum,
                                       const float* shade_stats, float* gmax_bound, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!log || (photo2 && !n_global) || !gmax_bound) return VHAP_E_NULLPTR;
    energy_total_kernel<<<1, 64, 0, vhap_stream(stream)>>>(log, photo2, n_global, w_photo, (float)world_size, d_sum,
                                                           reinterpret_cast<const unsigned*>(shade_stats), gmax_bound);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_sum_frames(const float* x, int B, int n, float* out_accum, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!x || !out_accum) return VHAP_E_NULLPTR;
    if (B <= 0 || n <= 0) return VHAP_E_BADDIM;
    sum_frames_kernel<<<vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(x, B, n, out_accum);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

