// Frame ingest: the frames of a sequence stay in HBM as the uint8 HWC arrays the decoder produced
// (1 B / channel instead of the 4 B / channel of the fp32 NCHW tensors the fit consumes) and one launch per batch gathers the
// requested frames, composites them over the background colour and converts them -- what the reference does per image on the
// DataLoader workers (video_dataset.py:253-259 apply_transforms, :302-323 apply_background_color, :261-268 apply_to_tensor).
//
// Bit-exact restatement of the host arithmetic:
//   w   = alpha / 255                       (numpy true divide: fp64)
//   img = uint8(w * fg + (1 - w) * bg)      (fp64 products and sum, truncating cast; bg = 255 'white' or 0 'black')
//   out = float32(img) / 255                (torchvision to_tensor: fp32 divide)
// The fp64 expression is evaluated in the same order with contraction off, so every one of the 2 x 256 x 256 (alpha, fg) cases
// rounds as on the host (tests/golden/ingest_golden.npz holds the exhaustive table made by the reference's own code).
// HBM-bound: 4 B read and 16 B written per pixel; four pixels per lane so the planar stores are 16 B wide.
#include "common.h"

namespace {

#pragma clang fp contract(off)
__device__ __forceinline__ unsigned composite_u8(unsigned fg, double w, double bg) {
    const double a = w * (double)fg;
    const double b = (1.0 - w) * bg;
    return (unsigned)(int)(a + b);
}

__device__ __forceinline__ float unit_f32(unsigned v) { return __fdiv_rn((float)v, 255.0f); }

// VEC = 4: H*W % 4 == 0 and all bases suitably aligned; VEC = 1: anything else.
template <int VEC>
__global__ __launch_bounds__(256) void frame_ingest_kernel(const unsigned char* __restrict__ rgb, const unsigned char* __restrict__ alpha,
                                                           const long long* __restrict__ index, int N, int HW, int bg_mode,
                                                           float* __restrict__ rgb_out, float* __restrict__ alpha_out, int* __restrict__ bad) {
    const int b = blockIdx.y;
    const long long p = ((long long)blockIdx.x * 256 + threadIdx.x) * VEC;
    if (p >= HW) return;
    long long f = index ? index[b] : b;
    if (f < 0) f += N;                                        // python-style negative timestep indices
    if (f < 0 || f >= N) {                                    // out of range: flag it, write zeros (never read out of bounds)
        if (bad && p == 0) atomicOr(bad, 1);
        f = -1;
    }
    unsigned px[VEC][3];
    unsigned al[VEC];
    if (f >= 0) {
        const unsigned char* src = rgb + ((size_t)f * HW + p) * 3;
        if (VEC == 4) {
            const uint3 q = *reinterpret_cast<const uint3*>(src);            // 12 bytes = 4 pixels
            const unsigned w[3] = {q.x, q.y, q.z};
#pragma unroll
            for (int k = 0; k < 12; k++) px[k / 3][k % 3] = (w[k / 4] >> (8 * (k % 4))) & 0xffu;
        } else {
            px[0][0] = src[0]; px[0][1] = src[1]; px[0][2] = src[2];
        }
        if (alpha) {
            const unsigned char* as = alpha + (size_t)f * HW + p;
            if (VEC == 4) {
                const unsigned q = *reinterpret_cast<const unsigned*>(as);
#pragma unroll
                for (int k = 0; k < 4; k++) al[k] = (q >> (8 * k)) & 0xffu;
            } else {
                al[0] = as[0];
            }
        }
    } else {
#pragma unroll
        for (int k = 0; k < VEC; k++) { px[k][0] = px[k][1] = px[k][2] = 0; al[k] = 0; }
    }
    if (bg_mode != 0) {
        const double bg = bg_mode == 1 ? 255.0 : 0.0;
#pragma unroll
        for (int k = 0; k < VEC; k++) {
            const double w = (double)al[k] / 255.0;
#pragma unroll
            for (int c = 0; c < 3; c++) px[k][c] = composite_u8(px[k][c], w, bg);
        }
    }
    float* dst = rgb_out + (size_t)b * 3 * HW + p;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        if (VEC == 4) {
            *reinterpret_cast<float4*>(dst + (size_t)c * HW) =
                make_float4(unit_f32(px[0][c]), unit_f32(px[1][c]), unit_f32(px[2][c]), unit_f32(px[3][c]));
        } else {
            dst[(size_t)c * HW] = unit_f32(px[0][c]);
        }
    }
    if (alpha_out && alpha) {
        float* ad = alpha_out + (size_t)b * HW + p;
        if (VEC == 4) *reinterpret_cast<float4*>(ad) = make_float4(unit_f32(al[0]), unit_f32(al[1]), unit_f32(al[2]), unit_f32(al[3]));
        else ad[0] = unit_f32(al[0]);
    }
}

}  // namespace

extern "C" int vhap_frame_ingest(const unsigned char* rgb_u8, const unsigned char* alpha_u8, const long long* index, int N, int B, int H,
                                 int W, int bg_mode, float* rgb_out, float* alpha_out, int* bad_index, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!rgb_u8 || !rgb_out) return VHAP_E_NULLPTR;
    if (bg_mode != VHAP_BG_NONE && !alpha_u8) return VHAP_E_NULLPTR;      // "'alpha_map' is required to apply background color"
    if (alpha_out && !alpha_u8) return VHAP_E_NULLPTR;
    if (N <= 0 || B <= 0 || H <= 0 || W <= 0 || (long long)H * W > 0x7fffffffLL) return VHAP_E_BADDIM;
    if (bg_mode != VHAP_BG_NONE && bg_mode != VHAP_BG_WHITE && bg_mode != VHAP_BG_BLACK) return VHAP_E_BADDIM;
    const int HW = H * W;
    const bool vec = HW % 4 == 0 && (reinterpret_cast<uintptr_t>(rgb_u8) & 3) == 0 && (reinterpret_cast<uintptr_t>(alpha_u8) & 3) == 0 &&
                     (reinterpret_cast<uintptr_t>(rgb_out) & 15) == 0 && (reinterpret_cast<uintptr_t>(alpha_out) & 15) == 0;
    if (vec) {
        const dim3 grid(vhap_cdiv(HW / 4, 256), B);
        frame_ingest_kernel<4><<<grid, 256, 0, vhap_stream(stream)>>>(rgb_u8, alpha_u8, index, N, HW, bg_mode, rgb_out, alpha_out, bad_index);
    } else {
        const dim3 grid(vhap_cdiv(HW, 256), B);
        frame_ingest_kernel<1><<<grid, 256, 0, vhap_stream(stream)>>>(rgb_u8, alpha_u8, index, N, HW, bg_mode, rgb_out, alpha_out, bad_index);
    }
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
