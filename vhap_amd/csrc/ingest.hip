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


// ---- batch feed: the per-step hand-over of a stage loop as a NODE of the captured step ----
// A stage replays one captured step hundreds of times, each time on another batch of the resident sequence (tracker.py:1376-1385).  The
// host used to enqueue, between two replays, the frame ingest and the copies of the batch's timesteps / landmarks / cameras into the
// step's static buffers: four small launches on the launch stream, in front of the step's first kernel.  Here the whole table of a pass
// -- frame index and timestep of every frame of every batch, in the order the batches will be taken -- is uploaded once, and the step
// itself begins with this gather: batch number cursor[0] of the table -> the static index / timestep buffers (+ up to three per-frame
// row arrays: landmarks, intrinsics, extrinsics), then the cursor is advanced.  The frame ingest (vhap_frame_ingest on the index buffer)
// is an ordinary node of the step after it.  One replay = one step on the next batch, nothing else to enqueue.
namespace {

struct FeedRows {
    const float* src[3];     // [N, width] per-frame rows (null: unused)
    float* dst[3];           // [n, width]
    int width[3];
};

__global__ __launch_bounds__(128) void batch_feed_kernel(const long long* __restrict__ frame_table, const long long* __restrict__ ts_table,
                                                         const int* __restrict__ cursor, int n, int capacity, long long* __restrict__ frame_out,
                                                         long long* __restrict__ ts_out, const FeedRows R, int N) {
    const int i = blockIdx.x;
    // cursor[0] = next batch, cursor[1] = batches in the table: past its end the LAST batch is taken again (a table of one batch = the same
    // batch for every replay: the sequential-tracking pattern)
    const int c = min(max(cursor[0], 0), min(max(cursor[1], 1), capacity) - 1);
    long long f = frame_table[(size_t)c * n + i];
    if (threadIdx.x == 0) {
        frame_out[i] = f;
        ts_out[i] = ts_table[(size_t)c * n + i];
    }
    if (f < 0) f += N;
    f = f < 0 ? 0 : (f >= N ? N - 1 : f);
#pragma unroll
    for (int k = 0; k < 3; k++) {
        if (!R.src[k]) continue;
        const float* s = R.src[k] + (size_t)f * R.width[k];
        float* d = R.dst[k] + (size_t)i * R.width[k];
        for (int j = threadIdx.x; j < R.width[k]; j += 128) d[j] = s[j];
    }
}
__global__ void batch_feed_advance_kernel(int* cursor) { cursor[0] += 1; }

}  // namespace

extern "C" int vhap_batch_feed(const long long* frame_table, const long long* ts_table, int* cursor, int n, int capacity, int N,
                               const float* rows0, float* out0, int width0, const float* rows1, float* out1, int width1,
                               const float* rows2, float* out2, int width2, long long* frame_out, long long* ts_out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!frame_table || !ts_table || !cursor || !frame_out || !ts_out) return VHAP_E_NULLPTR;
    if (n <= 0 || n > 65535 || capacity <= 0 || N <= 0 || width0 < 0 || width1 < 0 || width2 < 0) return VHAP_E_BADDIM;
    if ((rows0 && !out0) || (rows1 && !out1) || (rows2 && !out2)) return VHAP_E_NULLPTR;
    FeedRows R{{rows0, rows1, rows2}, {out0, out1, out2}, {width0, width1, width2}};
    hipStream_t st = vhap_stream(stream);
    batch_feed_kernel<<<n, 128, 0, st>>>(frame_table, ts_table, cursor, n, capacity, frame_out, ts_out, R, N);
    VHAP_LAUNCH_CHECK();
    batch_feed_advance_kernel<<<1, 1, 0, st>>>(cursor);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
