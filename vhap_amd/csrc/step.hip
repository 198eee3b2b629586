// Glue kernels of the native fit step (vhap_amd/step.py): the handful of scalar operations that remain between the fused stages
// once torch autograd no longer chains them -- assembling the energy from the stage accumulators (tracker.py:692-750), the
// photometric normaliser and its upstream gradient (tracker.py:430-439), the sum over frames of the canonical-vertex gradient
// for the shared static offset, and the focal-length gradient (tracker.py:141-157).  Each is one tiny launch instead of
// 5-15 elementwise / reduction launches of the autograd formulation.
#include "common.h"

namespace {

__device__ __forceinline__ float decode_ordered(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// log layout (VHAP_LOG_*): see vhap_hip.h
__global__ void energy_finalize_kernel(const float* __restrict__ frame_terms, const float* __restrict__ lmk, const float* __restrict__ tex_terms,
                                       const float* __restrict__ off_terms, const unsigned* __restrict__ shade_stats, float w_lmk,
                                       float w_reg_diffuse, float npix, float* __restrict__ log) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float v[VHAP_LOG_COUNT];
    for (int i = 0; i < VHAP_LOG_COUNT; i++) v[i] = 0.f;
    if (lmk) v[VHAP_LOG_LMK] = w_lmk * lmk[0];
    if (frame_terms)
        for (int i = 0; i < 6; i++) v[VHAP_LOG_SMOOTH_POSE + i] = frame_terms[i];
    if (tex_terms) { v[VHAP_LOG_TEX_TV] = tex_terms[0]; v[VHAP_LOG_TEX_RES] = tex_terms[1]; }
    if (shade_stats) {
        const float mx = decode_ordered(shade_stats[1]);
        v[VHAP_LOG_REG_DIFFUSE] = w_reg_diffuse * (fmaxf(mx - 1.0f, 0.0f) + __uint_as_float(shade_stats[2]) / npix);
    }
    if (off_terms)
        for (int i = 0; i < 3; i++) v[VHAP_LOG_OFF_LAP + i] = off_terms[i];
    float rest = 0.f;
    for (int i = 0; i < VHAP_LOG_REST; i++)
        if (i != VHAP_LOG_PHOTO) rest += v[i];
    v[VHAP_LOG_REST] = rest;
    for (int i = 0; i < VHAP_LOG_COUNT; i++) log[i] = v[i];
}

// photo2 = (sum |gt - pred|, #(alpha > 0)) ; n_global = the alpha count summed over ranks (== photo2[1] on one GPU)
__global__ void energy_total_kernel(float* __restrict__ log, const float* __restrict__ photo2, const float* __restrict__ n_global,
                                    float w_photo, float world, float* __restrict__ d_sum, const unsigned* __restrict__ shade_stats,
                                    float* __restrict__ gmax_bound) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float photo = 0.f, g = 0.f;
    if (photo2) {
        const float inv_n = world / (3.0f * n_global[0]);
        g = w_photo * inv_n;
        photo = g * photo2[0];
    }
    log[VHAP_LOG_PHOTO] = photo;
    log[VHAP_LOG_TOTAL] = log[VHAP_LOG_REST] + photo;
    if (d_sum) d_sum[0] = g;
    if (gmax_bound) {
        // |d albedo| = |d rgb| diffuse <= (|g| + antialias colour part <= |g|) max(diffuse); without the statistic: a generous constant
        const float dmax = shade_stats ? fmaxf(decode_ordered(shade_stats[1]), 1.0f) : 4.0f;
        gmax_bound[0] = 4.0f * fabsf(g) * dmax;      // (|d rgb| <= |g| + four antialias pairs x 0.5 |g|; 2^23 of fixed-point headroom on top)
    }
}

__global__ __launch_bounds__(256) void sum_frames_kernel(const float* __restrict__ x, int B, int n, float* __restrict__ out) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; b++) s += x[(size_t)b * n + i];
    out[i] += s;
}

__global__ void focal_bwd_kernel(const float* __restrict__ d_K, int B, float scale, float* __restrict__ d_focal) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    float s = 0.f;
    for (int b = 0; b < B; b++) s += d_K[4 * b] + d_K[4 * b + 1];
    d_focal[0] += s * scale;
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

extern "C" int vhap_focal_bwd(const float* d_K, int B, float scale, float* d_focal_accum, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!d_K || !d_focal_accum) return VHAP_E_NULLPTR;
    if (B <= 0) return VHAP_E_BADDIM;
    focal_bwd_kernel<<<1, 64, 0, vhap_stream(stream)>>>(d_K, B, scale, d_focal_accum);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
