// fp32 convolution on the matrix cores for the landmark detector's network (vhap/model/tracker.py:1263-1277 ->
// vhap/util/landmark_detector_fa.py:41-46, which drives the third-party `face_alignment` package -- absent from the reference checkout; its
// network is the published FAN of Bulat & Tzimiropoulos, ICCV 2017: a stem and four stacked hourglasses of pre-activation residual blocks).
//
//   vhap_conv2d_nhwc : out[n, y, x, co] (+)= bias[co] + sum_{ky,kx,ci} act(in[n, y*s + ky - p, x*s + kx - p, ci]) * w[ky, kx, ci, co]
//                      act(v) = relu?(v * in_scale[ci] + in_shift[ci])   -- the pre-activation BatchNorm + ReLU of the residual blocks, applied while the
//                      input tile is staged (padding is zero AFTER the activation, as torch pads the activated tensor); an optional ReLU on the way out.
//                      Input / output are channel SLICES of NHWC buffers (pointer to the slice's first channel + the buffer's channel count), so that
//                      a block's three convolutions read and write the slices of its concatenated output in place.
//   Implicit GEMM, exact fp32 (v_mfma_f32_16x16x4_f32): a workgroup of four waves owns 64 output pixels x 64 output channels; per (ky, kx) and per
//   32 input channels the activated input tile [64 px][32 ci] and the weight tile [32 ci][64 co] go through double-buffered LDS (two float4 per thread
//   each, fetched one step ahead), each wave multiplies its 16 pixels into the four 16-channel column tiles: 32 MFMA per staged pair.
//   vhap_nhwc_avgpool2, vhap_nhwc_upsample2_add, vhap_nhwc_add : the hourglass's elementwise glue.
//   vhap_nhwc_maxpool2, vhap_nhwc_l2norm : the face detector's (S3FD: a VGG-16 trunk with six detection heads) two other layer kinds.
#include "common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int CV_BM = 64, CV_BN = 64, CV_BK = 32, CV_T = 256;
constexpr int CV_AS = CV_BK + 2;      // row stride of the input tile: the 16 rows x 2 k of a half-wave's operand read fall into 32 different banks
constexpr int CV_BS = CV_BN + 16;     // row stride of the weight tile: k and k + 1 sixteen banks apart

struct ConvArgs {
    const float* in;      // first channel of the input slice
    int in_cs;            // channels of the buffer the slice lives in (pixel stride)
    int N, H, W, Cin;     // input extents
    const float* w;       // [KH, KW, Cin, Cout]
    const float* bias;    // [Cout] or null
    const float* in_scale;   // [Cin] or null
    const float* in_shift;   // [Cin] or null
    int in_relu;
    float* out;           // first channel of the output slice
    int out_cs;
    int Ho, Wo, Cout;
    int KH, KW, stride, pad;
    int out_relu, accumulate;
    float* part;          // split K (gridDim.z slices of the K tiles): partial sums [gridDim.z][N*Ho*Wo][Cout], summed by conv_splitk_finish_kernel; or null
};

// One K tile = one filter tap x 32 input channels.  The tile pair of step t + 1 is fetched into registers BEFORE the 32 MFMA of step t and stored into the
// other LDS buffer after them (one barrier per step): a deep level of the hourglass is a handful of workgroups walking 72 K tiles one after the other, so what
// a step costs is the latency of its loads unless they fly under the previous step's arithmetic (the first build staged, waited, multiplied: 21.9 ms for the
// network at batch 2, 4.6 TFLOP/s; `profiles/r06_fan_bench.txt`).
// FAST: Cin a multiple of 32, 16-byte loads possible on both operands -- every load of a step is unconditional (addresses clamped, zeros selected
// afterwards): with branches around them the compiler put a wait for ALL earlier loads in front of the later ones.
// WM: 16-pixel row tiles per wave -- a workgroup owns 64 * WM output pixels x 64 output channels.  WM = 2 where the grid is large anyway (the face
// detector's VGG trunk, the landmark network from ~8 frames on): every weight fragment read from LDS feeds two MFMA, twice the arithmetic per barrier.
template <bool FAST, bool ACT, int WM>
__global__ __launch_bounds__(CV_T) void conv2d_nhwc_kernel(const ConvArgs a) {
    constexpr int BM = CV_BM * WM;
    __shared__ __attribute__((aligned(16))) float As[2][BM][CV_AS];
    __shared__ __attribute__((aligned(16))) float Bs[2][CV_BK][CV_BS];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const long long npix = (long long)a.N * a.Ho * a.Wo;
    const long long p0 = (long long)blockIdx.x * BM;
    const int n0 = blockIdx.y * CV_BN;
    // the pixels this thread stages (tid / 4, + 64 per further row block) and its 8 channels of the tile (tid % 4)
    const int sp = tid >> 2, sc = (tid & 3) * 8;
    bool sp_ok[WM];
    int sn[WM], sy[WM], sx[WM];
#pragma unroll
    for (int m = 0; m < WM; m++) {
        const long long spi = p0 + m * CV_BM + sp;
        sp_ok[m] = spi < npix;
        sn[m] = sy[m] = sx[m] = 0;
        if (sp_ok[m]) {
            sn[m] = (int)(spi / ((long long)a.Ho * a.Wo));
            const int rem = (int)(spi - (long long)sn[m] * a.Ho * a.Wo);
            sy[m] = rem / a.Wo;
            sx[m] = rem - sy[m] * a.Wo;
        }
    }
    // the weight row this thread stages (tid / 8) and its 8 columns (tid % 8)
    const int wr = tid >> 3, wc = (tid & 7) * 8;
    f32x4 acc[WM][4];
#pragma unroll
    for (int m = 0; m < WM; m++)
#pragma unroll
        for (int j = 0; j < 4; j++) acc[m][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // 16-byte staging loads where the layout allows them (uniform): the slice and its buffer in multiples of four channels, 16-byte aligned
    const bool vec_in = (a.in_cs & 3) == 0 && (reinterpret_cast<uintptr_t>(a.in) & 15) == 0;
    const bool vec_w = (a.Cout & 3) == 0 && (reinterpret_cast<uintptr_t>(a.w) & 15) == 0;
    constexpr bool act = ACT;                                // (a.in_scale != nullptr)
    const int ctiles = (a.Cin + CV_BK - 1) / CV_BK, nt_all = a.KH * a.KW * ctiles;
    // this workgroup's slice of the K tiles (gridDim.z = 1: all of them)
    const int t0 = (int)((long long)nt_all * blockIdx.z / gridDim.z), t1 = (int)((long long)nt_all * (blockIdx.z + 1) / gridDim.z);

    float av[WM][8], wv[8], scv[8], shv[8];
    bool cur_ok[WM], w_ok0 = false, w_ok1 = false;
    int cur_ci0 = 0;
    auto issue = [&](int t) {                               // the loads of K tile t (no use of their results here)
        const int tap = t / ctiles, c0 = (t - tap * ctiles) * CV_BK;
        const int ky = tap / a.KW, kx = tap - ky * a.KW;
        const int ci0 = c0 + sc;
        cur_ci0 = ci0;
        const int wci = c0 + wr;
#pragma unroll
        for (int m = 0; m < WM; m++) {
            const int iy = sy[m] * a.stride + ky - a.pad, ix = sx[m] * a.stride + kx - a.pad;
            const bool in_ok = sp_ok[m] && iy >= 0 && iy < a.H && ix >= 0 && ix < a.W;
            const float* ip = a.in + ((size_t)((size_t)sn[m] * a.H + (in_ok ? iy : 0)) * a.W + (in_ok ? ix : 0)) * a.in_cs;
            cur_ok[m] = in_ok;
            if constexpr (FAST) {
                const float4 a0 = *reinterpret_cast<const float4*>(ip + ci0), a1 = *reinterpret_cast<const float4*>(ip + ci0 + 4);
                av[m][0] = a0.x; av[m][1] = a0.y; av[m][2] = a0.z; av[m][3] = a0.w; av[m][4] = a1.x; av[m][5] = a1.y; av[m][6] = a1.z; av[m][7] = a1.w;
            } else {
#pragma unroll
                for (int h = 0; h < 2; h++) {
                    const int c = ci0 + h * 4;
                    if (in_ok && vec_in && c + 3 < a.Cin) {
                        const float4 v4 = *reinterpret_cast<const float4*>(ip + c);
                        av[m][h * 4 + 0] = v4.x; av[m][h * 4 + 1] = v4.y; av[m][h * 4 + 2] = v4.z; av[m][h * 4 + 3] = v4.w;
                    } else {
#pragma unroll
                        for (int u = 0; u < 4; u++) av[m][h * 4 + u] = (in_ok && c + u < a.Cin) ? ip[c + u] : 0.f;
                    }
                }
            }
        }
        if constexpr (FAST) {
            if constexpr (ACT) {
                const float4 s0 = *reinterpret_cast<const float4*>(a.in_scale + ci0), s1 = *reinterpret_cast<const float4*>(a.in_scale + ci0 + 4);
                const float4 h0 = *reinterpret_cast<const float4*>(a.in_shift + ci0), h1 = *reinterpret_cast<const float4*>(a.in_shift + ci0 + 4);
                scv[0] = s0.x; scv[1] = s0.y; scv[2] = s0.z; scv[3] = s0.w; scv[4] = s1.x; scv[5] = s1.y; scv[6] = s1.z; scv[7] = s1.w;
                shv[0] = h0.x; shv[1] = h0.y; shv[2] = h0.z; shv[3] = h0.w; shv[4] = h1.x; shv[5] = h1.y; shv[6] = h1.z; shv[7] = h1.w;
            }
            const float* wrow = a.w + ((size_t)tap * a.Cin + wci) * a.Cout;
            const int co = n0 + wc;
            w_ok0 = co + 3 < a.Cout;
            w_ok1 = co + 7 < a.Cout;
            const float4 w0 = *reinterpret_cast<const float4*>(wrow + (w_ok0 ? co : 0)), w1 = *reinterpret_cast<const float4*>(wrow + (w_ok1 ? co + 4 : 0));
            wv[0] = w0.x; wv[1] = w0.y; wv[2] = w0.z; wv[3] = w0.w; wv[4] = w1.x; wv[5] = w1.y; wv[6] = w1.z; wv[7] = w1.w;
        } else {
            if (act) {
#pragma unroll
                for (int u = 0; u < 8; u++) {
                    const bool ok = ci0 + u < a.Cin;
                    scv[u] = ok ? a.in_scale[ci0 + u] : 0.f;
                    shv[u] = ok ? a.in_shift[ci0 + u] : 0.f;
                }
            }
            const float* wrow = a.w + ((size_t)tap * a.Cin + (wci < a.Cin ? wci : 0)) * a.Cout;
            w_ok0 = w_ok1 = true;
#pragma unroll
            for (int h = 0; h < 2; h++) {
                const int co = n0 + wc + h * 4;
                if (wci < a.Cin && vec_w && co + 3 < a.Cout) {
                    const float4 w4 = *reinterpret_cast<const float4*>(wrow + co);
                    wv[h * 4 + 0] = w4.x; wv[h * 4 + 1] = w4.y; wv[h * 4 + 2] = w4.z; wv[h * 4 + 3] = w4.w;
                } else {
#pragma unroll
                    for (int u = 0; u < 4; u++) wv[h * 4 + u] = (wci < a.Cin && co + u < a.Cout) ? wrow[co + u] : 0.f;
                }
            }
        }
    };
    auto commit = [&](int buf) {                            // activation (zero past the image and past Cin: torch pads the ACTIVATED tensor), then LDS
#pragma unroll
        for (int m = 0; m < WM; m++) {
#pragma unroll
            for (int u = 0; u < 8; u++) {
                float v = av[m][u];
                if (cur_ok[m] && (FAST || cur_ci0 + u < a.Cin)) {
                    if (act) v = v * scv[u] + shv[u];
                    if (a.in_relu) v = fmaxf(v, 0.f);
                } else {
                    v = 0.f;
                }
                av[m][u] = v;
            }
#pragma unroll
            for (int u = 0; u < 8; u += 2) *reinterpret_cast<float2*>(&As[buf][m * CV_BM + sp][sc + u]) = make_float2(av[m][u], av[m][u + 1]);
        }
        *reinterpret_cast<float4*>(&Bs[buf][wr][wc]) = w_ok0 ? make_float4(wv[0], wv[1], wv[2], wv[3]) : make_float4(0.f, 0.f, 0.f, 0.f);
        *reinterpret_cast<float4*>(&Bs[buf][wr][wc + 4]) = w_ok1 ? make_float4(wv[4], wv[5], wv[6], wv[7]) : make_float4(0.f, 0.f, 0.f, 0.f);
    };
    auto multiply = [&](int buf) {
        // (reading every fragment of the K tile ahead of its MFMA -- 20 LDS reads in flight instead of read, wait, two MFMA -- was measured: no faster, 32 VGPRs more)
#pragma unroll
        for (int kk = 0; kk < CV_BK / 4; kk++) {
            float am[WM], bj[4];
#pragma unroll
            for (int m = 0; m < WM; m++) am[m] = As[buf][m * CV_BM + wave * 16 + li][kk * 4 + lk];
#pragma unroll
            for (int j = 0; j < 4; j++) bj[j] = Bs[buf][kk * 4 + lk][j * 16 + li];
#pragma unroll
            for (int m = 0; m < WM; m++)
#pragma unroll
                for (int j = 0; j < 4; j++) acc[m][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(am[m], bj[j], acc[m][j], 0, 0, 0);
        }
    };
    issue(t0);
    commit(0);
    __syncthreads();
    for (int t = t0; t + 1 < t1; t++) {                     // (no branch between the loads and their use: the last step is peeled)
        issue(t + 1);
        __builtin_amdgcn_sched_barrier(0);                  // (the scheduler otherwise pulls the activation -- and the wait for its loads -- up among the MFMA)
        multiply((t - t0) & 1);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int m = 0; m < WM; m++)
#pragma unroll
            for (int u = 0; u < 8; u++) asm volatile("" : "+v"(av[m][u]) : : "memory");   // (the fetched tile is not touched before this point)
        commit((t + 1 - t0) & 1);                           // (its last readers passed the barrier that ended step t - 1)
        __syncthreads();
    }
    multiply((t1 - 1 - t0) & 1);
    // lane (li, lk) holds rows lk * 4 + r (pixels) of column li (channel) of each of the four column tiles
#pragma unroll
    for (int m = 0; m < WM; m++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const long long pi = p0 + m * CV_BM + wave * 16 + lk * 4 + r;
            if (pi >= npix) continue;
            if (a.part) {                                   // split K: this slice's partial sums; bias / accumulate / ReLU in the finish pass
                float* pp = a.part + ((size_t)blockIdx.z * npix + pi) * a.Cout;
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int co = n0 + j * 16 + li;
                    if (co < a.Cout) pp[co] = acc[m][j][r];
                }
                continue;
            }
            float* op = a.out + (size_t)pi * a.out_cs;
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int co = n0 + j * 16 + li;
                if (co >= a.Cout) continue;
                float v = acc[m][j][r] + (a.bias ? a.bias[co] : 0.f);
                if (a.accumulate) v += op[co];
                if (a.out_relu) v = fmaxf(v, 0.f);
                op[co] = v;
            }
        }
    }
}

// the slices of a split-K convolution summed in slice order (deterministic), then the epilogue of the unsplit kernel
__global__ __launch_bounds__(256) void conv_splitk_finish_kernel(const float* __restrict__ part, int ks, long long npix, int Cout, const float* __restrict__ bias,
                                                                 float* out, int out_cs, int out_relu, int accumulate) {
    const long long n = npix * Cout, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const long long pi = i / Cout;
    const int co = (int)(i - pi * Cout);
    float v = part[i];
    for (int z = 1; z < ks; z++) v += part[(size_t)z * n + i];
    v += bias ? bias[co] : 0.f;
    float* op = out + (size_t)pi * out_cs + co;
    if (accumulate) v += *op;
    if (out_relu) v = fmaxf(v, 0.f);
    *op = v;
}

__global__ __launch_bounds__(256) void nhwc_avgpool2_kernel(const float* __restrict__ in, int N, int H, int W, int C, float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const long long n = (long long)N * Ho * Wo * C, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int x = (int)(p % Wo), y = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
    const float* q = in + (((size_t)b * H + 2 * y) * W + 2 * x) * C + c;
    out[i] = (q[0] + q[C] + q[(size_t)W * C] + q[(size_t)W * C + C]) * 0.25f;      // torch's avg_pool2d: the sum of the window, times 1 / 4
}

// torch's max_pool2d(x, 2, 2): floor mode, a trailing odd row / column is dropped
__global__ __launch_bounds__(256) void nhwc_maxpool2_kernel(const float* __restrict__ in, int N, int H, int W, int C, float* __restrict__ out) {
    const int Ho = H / 2, Wo = W / 2;
    const long long n = (long long)N * Ho * Wo * C, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int x = (int)(p % Wo), y = (int)((p / Wo) % Ho), b = (int)(p / ((long long)Wo * Ho));
    const float* q = in + (((size_t)b * H + 2 * y) * W + 2 * x) * C + c;
    out[i] = fmaxf(fmaxf(q[0], q[C]), fmaxf(q[(size_t)W * C], q[(size_t)W * C + C]));
}

// the detector's L2Norm layer: out[p, c] = in[p, c] / (sqrt(sum_c in[p, c]^2) + eps) * weight[c]; one wave per pixel
__global__ __launch_bounds__(256) void nhwc_l2norm_kernel(const float* __restrict__ in, long long npix, int C, const float* __restrict__ weight, float eps,
                                                          float* __restrict__ out) {
    const long long p = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int lane = threadIdx.x & 63;
    if (p >= npix) return;
    const float* q = in + (size_t)p * C;
    float ss = 0.f;
    for (int c = lane; c < C; c += 64) ss += q[c] * q[c];
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) ss += __shfl_xor(ss, o, 64);
    const float nrm = sqrtf(ss) + eps;
    for (int c = lane; c < C; c += 64) out[(size_t)p * C + c] = q[c] / nrm * weight[c];       // (x / norm * weight: torch's order)
}

// out[n, y, x, c] = a[n, y, x, c] + b[n, y / 2, x / 2, c]   (F.interpolate(scale_factor = 2, mode = 'nearest') + the skip branch)
__global__ __launch_bounds__(256) void nhwc_upsample2_add_kernel(const float* __restrict__ a, const float* __restrict__ b, int N, int H, int W, int C,
                                                                 float* __restrict__ out) {
    const long long n = (long long)N * H * W * C, i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const int c = (int)(i % C);
    const long long p = i / C;
    const int x = (int)(p % W), y = (int)((p / W) % H), bb = (int)(p / ((long long)W * H));
    out[i] = a[i] + b[(((size_t)bb * (H / 2) + y / 2) * (W / 2) + x / 2) * C + c];
}

__global__ __launch_bounds__(256) void nhwc_add_kernel(const float* a, const float* __restrict__ b, const float* __restrict__ c,
                                                       long long n, float* out) {          // (out may be a: the identity skip is added in place)
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    float v = a[i] + b[i];                                       // ((a + b) + c): torch's `previous + ll + tmp_out_`
    if (c) v += c[i];
    out[i] = v;
}

}  // namespace

extern "C" int vhap_conv2d_nhwc_ws(const float* in, int in_channel_stride, int N, int H, int W, int Cin, const float* weight, const float* bias,
                                   const float* in_scale, const float* in_shift, int KH, int KW, int stride, int pad, float* out,
                                   int out_channel_stride, int Cout, float* workspace, long long workspace_floats, int call_flags,
                                   vhap_stream_t stream) {
    VHAP_ENTER();
    if (!in || !weight || !out) return VHAP_E_NULLPTR;
    if ((in_scale == nullptr) != (in_shift == nullptr)) return VHAP_E_NULLPTR;
    if (N <= 0 || H <= 0 || W <= 0 || Cin <= 0 || Cout <= 0 || KH <= 0 || KW <= 0 || stride <= 0 || pad < 0 || in_channel_stride < Cin ||
        out_channel_stride < Cout || workspace_floats < 0)
        return VHAP_E_BADDIM;
    const int Ho = (H + 2 * pad - KH) / stride + 1, Wo = (W + 2 * pad - KW) / stride + 1;
    if (Ho <= 0 || Wo <= 0) return VHAP_E_BADDIM;
    const long long npix = (long long)N * Ho * Wo;
    // 128-pixel workgroups (every weight fragment read from LDS feeds two MFMA) were measured on both networks at every size: no gain over 64
    // (profiles/r06_sfd_bench.txt) -- kept behind debug flag 8388608 for grids that leave >= 4 of them per CU, the same bits either way
    const unsigned gy = (unsigned)vhap_cdiv(Cout, CV_BN);
    const int wm = (vhap_cdiv(npix, 2 * CV_BM) * gy >= 1024 && (vhap_g_debug_flags & 8388608)) ? 2 : 1;
    const unsigned gx = (unsigned)vhap_cdiv(npix, CV_BM * wm);
    // Split K when the pixel x channel tiles alone leave the chip empty (the deep levels of the hourglass are 1 .. 64 tiles walking 72 K tiles
    // one after the other): aim at ~768 workgroups, at least two K tiles per slice, what the workspace holds.  A function of the shapes and of
    // the workspace size only: the same call gives the same sums.
    int ks = 1;
    if (workspace) {
        const int nt = KH * KW * (int)vhap_cdiv(Cin, CV_BK);
        const long long wgs = (long long)gx * gy;
        long long want = (768 + wgs - 1) / wgs;
        if (want > 16) want = 16;
        if (want > nt / 2) want = nt / 2;
        while (want > 1 && want * npix * Cout > workspace_floats) want--;
        if (want > 1) ks = (int)want;
    }
    ConvArgs a{in, in_channel_stride, N, H, W, Cin, weight, bias, in_scale, in_shift, (call_flags & VHAP_CONV_IN_RELU) ? 1 : 0,
               out, out_channel_stride, Ho, Wo, Cout, KH, KW, stride, pad, (call_flags & VHAP_CONV_OUT_RELU) ? 1 : 0,
               (call_flags & VHAP_CONV_ACCUMULATE) ? 1 : 0, ks > 1 ? workspace : nullptr};
    const dim3 grid(gx, gy, (unsigned)ks);
    const bool al16 = ((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(weight) | reinterpret_cast<uintptr_t>(in_scale) |
                        reinterpret_cast<uintptr_t>(in_shift)) & 15) == 0;
    const bool fast = al16 && Cin % CV_BK == 0 && (in_channel_stride & 3) == 0 && (Cout & 3) == 0;
    void (*k)(const ConvArgs);
    if (wm == 2)
        k = fast ? (in_scale ? conv2d_nhwc_kernel<true, true, 2> : conv2d_nhwc_kernel<true, false, 2>)
                 : (in_scale ? conv2d_nhwc_kernel<false, true, 2> : conv2d_nhwc_kernel<false, false, 2>);
    else
        k = fast ? (in_scale ? conv2d_nhwc_kernel<true, true, 1> : conv2d_nhwc_kernel<true, false, 1>)
                 : (in_scale ? conv2d_nhwc_kernel<false, true, 1> : conv2d_nhwc_kernel<false, false, 1>);
    k<<<grid, CV_T, 0, vhap_stream(stream)>>>(a);
    VHAP_LAUNCH_CHECK();
    if (ks > 1) {
        conv_splitk_finish_kernel<<<(unsigned)vhap_cdiv(npix * Cout, 256), 256, 0, vhap_stream(stream)>>>(workspace, ks, npix, Cout, bias, out, out_channel_stride,
                                                                                                         a.out_relu, a.accumulate);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

extern "C" int vhap_conv2d_nhwc(const float* in, int in_channel_stride, int N, int H, int W, int Cin, const float* weight, const float* bias,
                                const float* in_scale, const float* in_shift, int KH, int KW, int stride, int pad, float* out,
                                int out_channel_stride, int Cout, int call_flags, vhap_stream_t stream) {
    return vhap_conv2d_nhwc_ws(in, in_channel_stride, N, H, W, Cin, weight, bias, in_scale, in_shift, KH, KW, stride, pad, out, out_channel_stride, Cout,
                               nullptr, 0, call_flags, stream);
}

extern "C" int vhap_nhwc_avgpool2(const float* in, int N, int H, int W, int C, float* out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!in || !out) return VHAP_E_NULLPTR;
    if (N <= 0 || H < 2 || W < 2 || C <= 0 || (H & 1) || (W & 1)) return VHAP_E_BADDIM;
    const long long n = (long long)N * (H / 2) * (W / 2) * C;
    nhwc_avgpool2_kernel<<<(unsigned)vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(in, N, H, W, C, out);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_nhwc_maxpool2(const float* in, int N, int H, int W, int C, float* out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!in || !out) return VHAP_E_NULLPTR;
    if (N <= 0 || H < 2 || W < 2 || C <= 0) return VHAP_E_BADDIM;
    const long long n = (long long)N * (H / 2) * (W / 2) * C;
    nhwc_maxpool2_kernel<<<(unsigned)vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(in, N, H, W, C, out);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_nhwc_l2norm(const float* in, long long npix, int C, const float* weight, float eps, float* out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!in || !weight || !out) return VHAP_E_NULLPTR;
    if (npix <= 0 || C <= 0) return VHAP_E_BADDIM;
    nhwc_l2norm_kernel<<<(unsigned)vhap_cdiv(npix, 4), 256, 0, vhap_stream(stream)>>>(in, npix, C, weight, eps, out);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_nhwc_upsample2_add(const float* skip, const float* low, int N, int H, int W, int C, float* out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!skip || !low || !out) return VHAP_E_NULLPTR;
    if (N <= 0 || H < 2 || W < 2 || C <= 0 || (H & 1) || (W & 1)) return VHAP_E_BADDIM;
    const long long n = (long long)N * H * W * C;
    nhwc_upsample2_add_kernel<<<(unsigned)vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(skip, low, N, H, W, C, out);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_nhwc_add(const float* a, const float* b, const float* c, long long n, float* out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!a || !b || !out) return VHAP_E_NULLPTR;
    if (n <= 0) return VHAP_E_BADDIM;
    nhwc_add_kernel<<<(unsigned)vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(a, b, c, n, out);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
