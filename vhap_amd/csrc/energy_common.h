// Assembly of the step energy from the stage accumulators (tracker.py:692-750) and of the photometric normaliser with its upstream
// gradient (tracker.py:430-439): single-thread device code shared by the stand-alone glue kernels (step.hip) and by the epilogue of the
// photometric sum (pixel.hip: the last workgroup of photo_fwd does it, so that no launch sits between the forward and the backward).
#pragma once
#include "common.h"

namespace vhap_energy {

__device__ __forceinline__ float decode_ordered(unsigned u) { return __uint_as_float((u & 0x80000000u) ? (u & 0x7fffffffu) : ~u); }

// log layout (VHAP_LOG_*): see vhap_hip.h.  Any input may be NULL (= term absent).
__device__ inline void finalize(const float* frame_terms, const float* lmk, const float* tex_terms, const float* off_terms,
                                const unsigned* shade_stats, float w_lmk, float w_reg_diffuse, float npix, float* log) {
    float v[VHAP_LOG_COUNT];
    for (int i = 0; i < VHAP_LOG_COUNT; i++) v[i] = 0.f;
    if (lmk) v[VHAP_LOG_LMK] = w_lmk * lmk[0];
    if (frame_terms)
        for (int i = 0; i < 6; i++) v[VHAP_LOG_SMOOTH_POSE + i] = frame_terms[i];
    if (tex_terms) { v[VHAP_LOG_TEX_TV] = tex_terms[0]; v[VHAP_LOG_TEX_RES] = tex_terms[1]; v[VHAP_LOG_TEX_PCA] = tex_terms[2]; }   // tex_terms[3]
    if (shade_stats) {
        const float mx = decode_ordered(shade_stats[1]);
        v[VHAP_LOG_REG_DIFFUSE] = w_reg_diffuse * (fmaxf(mx - 1.0f, 0.0f) + __uint_as_float(shade_stats[2]) / npix);
    }
    if (off_terms) {          // [4]: Laplacian, L1, rigidity, temporal smoothness of the dynamic offset
        for (int i = 0; i < 3; i++) v[VHAP_LOG_OFF_LAP + i] = off_terms[i];
        v[VHAP_LOG_OFF_DYNAMIC] = off_terms[3];
    }
    float rest = 0.f;
    for (int i = 0; i < VHAP_LOG_REST; i++)
        if (i != VHAP_LOG_PHOTO) rest += v[i];
    rest += v[VHAP_LOG_OFF_DYNAMIC];
    rest += v[VHAP_LOG_TEX_PCA];
    v[VHAP_LOG_REST] = rest;
    for (int i = 0; i < VHAP_LOG_COUNT; i++) log[i] = v[i];
}

// (sum |gt - pred|, #(alpha > 0)) ; n_global = the alpha count summed over ranks (== the local count on one GPU)
__device__ inline void total(float* log, bool has_photo, float photo_sum, float n_global, float w_photo, float world, float* d_sum,
                             const unsigned* shade_stats, float* gmax_bound) {
    float photo = 0.f, g = 0.f;
    if (has_photo) {
        const float inv_n = world / (3.0f * n_global);
        g = w_photo * inv_n;
        photo = g * photo_sum;
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

// finalize() + total() by ONE WAVE (all 64 lanes must call): lane i fetches term i, so the ~20 dependent scalar loads of the single-thread
// form (1-2 us each under load) become one round of parallel loads.  Same arithmetic, same summation order.
__device__ inline void finalize_total_wave(const float* frame_terms, const float* lmk, const float* tex_terms, const float* off_terms,
                                           const unsigned* shade_stats, float w_lmk, float w_reg_diffuse, float npix, float photo_sum,
                                           float n_global, float w_photo, float world, float* log, float* d_sum, float* gmax_bound) {
    const int i = threadIdx.x & 63;
    // every lane's term through ONE load (address select), the two statistics words next to it: as an else-if chain with a load per arm
    // this was eight dependent round trips on the tail of the kernel the whole backward waits for
    const float* src = nullptr;
    if (i == VHAP_LOG_LMK) src = lmk;
    else if (i >= VHAP_LOG_SMOOTH_POSE && i < VHAP_LOG_SMOOTH_POSE + 6) src = frame_terms ? frame_terms + (i - VHAP_LOG_SMOOTH_POSE) : nullptr;
    else if (i == VHAP_LOG_TEX_TV) src = tex_terms;
    else if (i == VHAP_LOG_TEX_RES) src = tex_terms ? tex_terms + 1 : nullptr;
    else if (i == VHAP_LOG_TEX_PCA) src = tex_terms ? tex_terms + 2 : nullptr;
    else if (i >= VHAP_LOG_OFF_LAP && i < VHAP_LOG_OFF_LAP + 3) src = off_terms ? off_terms + (i - VHAP_LOG_OFF_LAP) : nullptr;
    else if (i == VHAP_LOG_OFF_DYNAMIC) src = off_terms ? off_terms + 3 : nullptr;
    const unsigned* st = shade_stats ? shade_stats : reinterpret_cast<const unsigned*>(log);      // (stand-in address, values unused)
    const unsigned s1 = st[1], s2 = st[2];
    float v = 0.f;
    if (src) v = *src;
    if (i == VHAP_LOG_LMK) v = w_lmk * v;
    float mx = 4.0f;
    if (shade_stats) {
        mx = decode_ordered(s1);
        if (i == VHAP_LOG_REG_DIFFUSE) v = w_reg_diffuse * (fmaxf(mx - 1.0f, 0.0f) + __uint_as_float(s2) / npix);
        mx = fmaxf(mx, 1.0f);
    }
    float rest = 0.f;                                  // serial order 0..13 like finalize() (bit-identical log)
    for (int k = 0; k < VHAP_LOG_REST; k++) {
        const float vk = __shfl(v, k, 64);
        if (k != VHAP_LOG_PHOTO) rest += vk;
    }
    rest += __shfl(v, VHAP_LOG_OFF_DYNAMIC, 64);
    rest += __shfl(v, VHAP_LOG_TEX_PCA, 64);
    const float g = w_photo * (world / (3.0f * n_global));
    const float photo = g * photo_sum;
    if (i == VHAP_LOG_PHOTO) v = photo;
    if (i == VHAP_LOG_REST) v = rest;
    if (i == VHAP_LOG_TOTAL) v = rest + photo;
    if (i < VHAP_LOG_COUNT) log[i] = v;
    if (i == 0) {
        if (d_sum) d_sum[0] = g;
        if (gmax_bound) gmax_bound[0] = 4.0f * fabsf(g) * mx;
    }
}

}  // namespace vhap_energy
