// FLAME PCA texture model (vhap/model/flame.py:665-688 FlameTexPCA; tracker.py:57-60, 241-244 get_base_texture, 519-521 reg_tex_pca):
//   texture = mean + basis . code            [S, S, 3] in 0..255, channels B, G, R       (S = 512)
//   base    = clamp(nearest_resize(texture, T)[R, G, B] / 255, 0, 1)                     [3, T, T]  -- what FlameTexPainted returns otherwise
//   reg_tex_pca = w * mean(code^2)                                                       (std_tex = 1)
// and the backward to `code`.  The `use_flame_tex` / tex_painted = False configuration: `base` takes the place of the painted texture in
// vhap_tex_prep_fwd.  HBM-bound: the basis is [S*S*3, n] floats = 315 MB at n = 100, read once forward and once backward:
//   forward : one WAVE per row of the basis (lanes over the n coefficients, 400 contiguous bytes per row, DPP sum), four rows in flight;
//             then one thread per texel of the T x T base (nearest source texel, channel swap, / 255, clamp)
//   backward: per source texel the sum of d(base) over the texels it was copied to, masked by the clamp; then d(code) = basis^T g with
//             one wave per run of rows, two coefficients per lane in registers, one atomic per coefficient and wave.
#include "common.h"

namespace {

constexpr int PCA_ROWS_PER_WAVE = 4;

// src[r] = mean[r] + sum_k basis[r, k] code[k];  n <= 256
__global__ __launch_bounds__(256) void tex_pca_rows_kernel(const float* __restrict__ mean, const float* __restrict__ basis, const float* __restrict__ code,
                                                           int n, int R, float* __restrict__ src) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    float c[4];
#pragma unroll
    for (int j = 0; j < 4; j++) c[j] = lane + 64 * j < n ? code[lane + 64 * j] : 0.f;
    const long long r0 = wave * PCA_ROWS_PER_WAVE;
    float acc[PCA_ROWS_PER_WAVE];
#pragma unroll
    for (int u = 0; u < PCA_ROWS_PER_WAVE; u++) {
        const long long r = r0 + u;
        float a = 0.f;
        if (r < R) {
            const float* b = basis + (size_t)r * n;
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (lane + 64 * j < n) a += b[lane + 64 * j] * c[j];
        }
        acc[u] = a;
    }
#pragma unroll
    for (int u = 0; u < PCA_ROWS_PER_WAVE; u++) {
        const float s = vhap_wave_sum_dpp(acc[u]);
        const long long r = r0 + u;
        if (lane == 0 && r < R) src[r] = mean[r] + s;
    }
}

// torch's 'nearest' source index: floor(dst * (in / out)) in float, clamped
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) { return min((int)floorf((float)dst * scale), in_size - 1); }

// base[c][y][x] = clamp(src[(sy S + sx) 3 + (2 - c)] / 255, 0, 1); block 0 also adds the regulariser s_reg * sum(code^2)
__global__ __launch_bounds__(256) void tex_pca_resize_kernel(const float* __restrict__ src, int S, int T, float scale, float* __restrict__ base,
                                                             const float* __restrict__ code, int n, float s_reg, float* __restrict__ term) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x < T) {
        const int sy = nearest_src(y, scale, S), sx = nearest_src(x, scale, S);
        const float* q = src + ((size_t)sy * S + sx) * 3;
        const size_t plane = (size_t)T * T, i = (size_t)y * T + x;
#pragma unroll
        for (int c = 0; c < 3; c++) base[c * plane + i] = fminf(fmaxf(q[2 - c] / 255.0f, 0.0f), 1.0f);
    }
    if (term && s_reg != 0.f && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        float a = 0.f;
        for (int k = threadIdx.x; k < n; k += 64) a += code[k] * code[k];
        a = vhap_wave_sum_dpp(a);
        if (threadIdx.x == 0) atomicAdd(term, s_reg * a);
    }
}

// g[r] (source texel, source channel) = 1/255 * [0 <= src/255 <= 1] * sum over the base texels copied from it of d_base
__global__ __launch_bounds__(256) void tex_pca_gather_kernel(const float* __restrict__ src, const float* __restrict__ d_base, int S, int T, float scale,
                                                             float* __restrict__ g) {
    const int sx = blockIdx.x * 256 + threadIdx.x, sy = blockIdx.y;
    if (sx >= S) return;
    // the base texels whose nearest source is (sy, sx): a contiguous run per axis -- found by stepping from the estimate
    auto run = [&](int s, int& lo, int& hi) {
        int d = (int)floorf((float)s / scale);
        d = min(max(d, 0), T - 1);
        while (d > 0 && nearest_src(d - 1, scale, S) >= s) d--;
        while (d < T && nearest_src(d, scale, S) < s) d++;
        lo = d;
        while (d < T && nearest_src(d, scale, S) == s) d++;
        hi = d;
    };
    int y0, y1, x0, x1;
    run(sy, y0, y1);
    run(sx, x0, x1);
    const size_t plane = (size_t)T * T;
    float a[3] = {0.f, 0.f, 0.f};
    for (int y = y0; y < y1; y++)
        for (int x = x0; x < x1; x++) {
            const size_t i = (size_t)y * T + x;
#pragma unroll
            for (int c = 0; c < 3; c++) a[c] += d_base[c * plane + i];
        }
    const size_t r = ((size_t)sy * S + sx) * 3;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float v = src[r + (2 - c)] / 255.0f;
        g[r + (2 - c)] = (v >= 0.0f && v <= 1.0f) ? a[c] / 255.0f : 0.f;
    }
}

// d_code[k] += sum_r basis[r, k] g[r] over this wave's run of rows (+ the regulariser's gradient, once)
__global__ __launch_bounds__(256) void tex_pca_code_bwd_kernel(const float* __restrict__ basis, const float* __restrict__ g, int n, int R, int rows_per_wave,
                                                               const float* __restrict__ code, float s_reg, const float* __restrict__ d_term,
                                                               float* __restrict__ d_code) {
    const int lane = threadIdx.x & 63;
    const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
    const long long r0 = wave * rows_per_wave, r1 = min((long long)R, r0 + rows_per_wave);
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (long long r = r0; r < r1; r++) {
        const float gr = g[r];
        const float* b = basis + (size_t)r * n;
#pragma unroll
        for (int j = 0; j < 4; j++)
            if (lane + 64 * j < n) acc[j] += b[lane + 64 * j] * gr;
    }
#pragma unroll
    for (int j = 0; j < 4; j++) {
        const int k = lane + 64 * j;
        if (k < n) {
            float v = acc[j];
            if (wave == 0 && s_reg != 0.f) v += 2.0f * s_reg * code[k] * (d_term ? d_term[0] : 1.0f);
            if (v != 0.f) atomicAdd(&d_code[k], v);
        }
    }
}

}  // namespace

extern "C" int vhap_tex_pca_fwd(const float* mean, const float* basis, const float* code, int n, int S, int T, float s_reg, float* src,
                                float* base, float* term_accum, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!mean || !basis || !code || !src || !base) return VHAP_E_NULLPTR;
    if (n <= 0 || n > 256 || S <= 0 || T <= 0 || S > 8192 || T > 16384) return VHAP_E_BADDIM;
    const int R = S * S * 3;
    hipStream_t st = vhap_stream(stream);
    tex_pca_rows_kernel<<<vhap_cdiv(R, 4 * PCA_ROWS_PER_WAVE), 256, 0, st>>>(mean, basis, code, n, R, src);
    VHAP_LAUNCH_CHECK();
    tex_pca_resize_kernel<<<dim3(vhap_cdiv(T, 256), T), 256, 0, st>>>(src, S, T, (float)S / (float)T, base, code, n, s_reg, term_accum);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_tex_pca_bwd(const float* basis, const float* src, const float* d_base, const float* code, int n, int S, int T, float s_reg,
                                const float* d_term, float* g_work, float* d_code, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!basis || !src || !d_base || !code || !g_work || !d_code) return VHAP_E_NULLPTR;
    if (n <= 0 || n > 256 || S <= 0 || T <= 0 || S > 8192 || T > 16384) return VHAP_E_BADDIM;
    const int R = S * S * 3;
    hipStream_t st = vhap_stream(stream);
    tex_pca_gather_kernel<<<dim3(vhap_cdiv(S, 256), S), 256, 0, st>>>(src, d_base, S, T, (float)S / (float)T, g_work);
    VHAP_LAUNCH_CHECK();
    const int rows_per_wave = 192;
    tex_pca_code_bwd_kernel<<<vhap_cdiv(R, 4 * rows_per_wave), 256, 0, st>>>(basis, g_work, n, R, rows_per_wave, code, s_reg, d_term, d_code);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
