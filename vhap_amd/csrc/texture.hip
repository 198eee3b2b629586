// Trilinear mip-mapped texture sampling (forward / backward) for gfx950.
//
// Replaces dr.texture(tex, uv, uv_da, filter_mode='linear-mipmap-linear') with boundary mode 'wrap'
// and max_mip_level=None (vhap/util/render_nvdiffrast.py:399) and filter_mode='linear'
// (vhap/util/render_uvmap.py:41).  The reference replicates ONE texture B times
// (tracker.py:234, render_nvdiffrast.py:398: 805 MB at B=16, T=2048) and nvdiffrast rebuilds the
// mip chain of every copy on every call; here the texture may be shared (TB=1) and the pyramid is
// built once per step into a caller-owned buffer.
//
// Semantics (restated in oracle/torch_ref.py texture()):
//   mip l+1 = ((a00 + a01) + (a10 + a11)) * 0.25 down to 1x1 (while both extents are even)
//   s = uv_da * (Wt, Ht); A = sx^2+tx^2, B = sy^2+ty^2, C = sx*sy+tx*ty
//   lambda = (A+B)/2 + sqrt((A-B)^2/4 + C^2); level = clamp(log2(lambda)/2, 0, L)
//   l0 = min(floor(level), L-1), f = level - l0; out = c(l0) + f*(c(l0+1) - c(l0))
//   c(l): bilinear, texel centres at half-integers, wrap addressing
//   d(level)/d(uv_da) is taken only where level is strictly inside (0, L).
#include <type_traits>

#include "common.h"
#include "tex_sample.h"

namespace {

// one level per launch: dst[y][x] = box(src[2y..2y+1][2x..2x+1])
__global__ __launch_bounds__(256) void mip_down_kernel(const float* __restrict__ src, float* __restrict__ dst, int TB, int h,
                                                       int w, int C, long long src_stride, long long dst_stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)TB * h * w * C;
    if (i >= n) return;
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int tb = (int)(r / h);
    const float* s = src + tb * src_stride;
    const int sw = 2 * w;
    const float a00 = s[((size_t)(2 * y) * sw + 2 * x) * C + c], a01 = s[((size_t)(2 * y) * sw + 2 * x + 1) * C + c];
    const float a10 = s[((size_t)(2 * y + 1) * sw + 2 * x) * C + c], a11 = s[((size_t)(2 * y + 1) * sw + 2 * x + 1) * C + c];
    dst[tb * dst_stride + ((size_t)y * w + x) * C + c] = ((a00 + a01) + (a10 + a11)) * 0.25f;
}

// backward of one down-sampling step: every fine texel receives 0.25 * its parent's gradient (added)
__global__ __launch_bounds__(256) void mip_fold_kernel(float* __restrict__ fine, const float* __restrict__ coarse, int TB, int h,
                                                       int w, int C, long long fine_stride, long long coarse_stride) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over FINE texels (h, w = fine extents)
    const long long n = (long long)TB * h * w * C;
    if (i >= n) return;
    const int c = (int)(i % C);
    long long r = i / C;
    const int x = (int)(r % w); r /= w;
    const int y = (int)(r % h);
    const int tb = (int)(r / h);
    const float g = coarse[tb * coarse_stride + ((size_t)(y >> 1) * (w >> 1) + (x >> 1)) * C + c];
    fine[tb * fine_stride + ((size_t)y * w + x) * C + c] += 0.25f * g;
}

// Four levels in one launch: a workgroup loads a 32x32 tile of level l (12 KB of LDS at C = 3) and writes the 16x16 / 8x8 / 4x4 / 2x2
// tiles of levels l+1 .. l+4, each computed from the STORED values of the level above with mip_down_kernel's expression (same bits).
// Extents of level l must be multiples of 32.
template <int C>
__global__ __launch_bounds__(256) void mip_down4_kernel(const float* __restrict__ src, float* __restrict__ mips, const TexDesc D, int l) {
    __shared__ float a[32 * 32 * C], b[16 * 16 * C];
    const int w = D.W >> l, tx = blockIdx.x, ty = blockIdx.y, tb = blockIdx.z;
    const float* S = src + (size_t)tb * (l == 0 ? (size_t)D.H * D.W * C : (size_t)D.per_tex);
    for (int i = threadIdx.x; i < 32 * 32 * C; i += 256) {
        const int c = i % C, x = (i / C) % 32, y = i / (C * 32);
        a[i] = S[((size_t)(ty * 32 + y) * w + tx * 32 + x) * C + c];
    }
    __syncthreads();
    float* cur = a;
    float* nxt = b;
    int n = 32;
#pragma unroll
    for (int s = 1; s <= 4; s++) {
        const int m = n >> 1;
        float* dst = mips + (size_t)tb * D.per_tex + D.off[l + s];
        const int wl = D.W >> (l + s);
        for (int i = threadIdx.x; i < m * m * C; i += 256) {
            const int c = i % C, x = (i / C) % m, y = i / (C * m);
            const float a00 = cur[((2 * y) * n + 2 * x) * C + c], a01 = cur[((2 * y) * n + 2 * x + 1) * C + c];
            const float a10 = cur[((2 * y + 1) * n + 2 * x) * C + c], a11 = cur[((2 * y + 1) * n + 2 * x + 1) * C + c];
            const float v = ((a00 + a01) + (a10 + a11)) * 0.25f;
            nxt[i] = v;
            dst[((size_t)(ty * m + y) * wl + tx * m + x) * C + c] = v;
        }
        __syncthreads();
        float* t = cur; cur = nxt; nxt = t;
        n = m;
    }
}

// The small levels in ONE single-workgroup launch each way (they are pure launch latency otherwise: 7 of the 11 levels of a
// 2048^2 texture hold <= 64x64 texels).  Levels are processed in sequence with a barrier in between; global memory written by
// the workgroup is visible to it after __syncthreads + __threadfence_block.
constexpr int TAIL_MAX = 64;      // levels whose SOURCE is at most TAIL_MAX x TAIL_MAX go to the tail kernel (one workgroup: keep it short)
__global__ __launch_bounds__(1024) void mip_down_tail_kernel(float* __restrict__ mips, const TexDesc D, int l_first) {
    for (int tb = 0; tb < D.TB; tb++) {
        for (int l = l_first; l <= D.L; l++) {
            const int h = D.H >> l, w = D.W >> l, C = D.C;
            const float* src = mips + (size_t)tb * D.per_tex + D.off[l - 1];
            float* dst = mips + (size_t)tb * D.per_tex + D.off[l];
            const int sw = 2 * w;
            for (int i = threadIdx.x; i < h * w * C; i += 1024) {
                const int c = i % C, x = (i / C) % w, y = i / (C * w);
                const float a00 = src[((size_t)(2 * y) * sw + 2 * x) * C + c], a01 = src[((size_t)(2 * y) * sw + 2 * x + 1) * C + c];
                const float a10 = src[((size_t)(2 * y + 1) * sw + 2 * x) * C + c], a11 = src[((size_t)(2 * y + 1) * sw + 2 * x + 1) * C + c];
                dst[i] = ((a00 + a01) + (a10 + a11)) * 0.25f;
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}
// fold levels L .. l_last+1 into level l_last (coarse -> fine), l_last >= 1
__global__ __launch_bounds__(1024) void mip_fold_tail_kernel(float* __restrict__ d_mips, const TexDesc D, int l_last) {
    for (int tb = 0; tb < D.TB; tb++) {
        for (int l = D.L; l > l_last; l--) {
            const int h = D.H >> (l - 1), w = D.W >> (l - 1), C = D.C;       // fine extents
            float* fine = d_mips + (size_t)tb * D.per_tex + D.off[l - 1];
            const float* coarse = d_mips + (size_t)tb * D.per_tex + D.off[l];
            for (int i = threadIdx.x; i < h * w * C; i += 1024) {
                const int c = i % C, x = (i / C) % w, y = i / (C * w);
                fine[i] += 0.25f * coarse[((size_t)(y >> 1) * (w >> 1) + (x >> 1)) * C + c];
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void texture_fwd_kernel(const float* __restrict__ tex, const float* __restrict__ mips,
                                                          const TexDesc D, const float2* __restrict__ uv,
                                                          const float4* __restrict__ uv_da, long long npix, int HW,
                                                          float* __restrict__ out) {
    const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pi >= npix) return;
    const int tb = D.TB == 1 ? 0 : (int)(pi / HW);
    const float2 c = uv[pi];
    float res[C];
    if (uv_da == nullptr) {
        const Taps t = make_taps(c.x, c.y, D.W, D.H, C);
        const float* T = level_ptr(tex, mips, D, tb, 0);
#pragma unroll
        for (int k = 0; k < C; k++) {
            const float top = T[t.i00 + k] + t.fx * (T[t.i10 + k] - T[t.i00 + k]);
            const float bot = T[t.i01 + k] + t.fx * (T[t.i11 + k] - T[t.i01 + k]);
            res[k] = top + t.fy * (bot - top);
        }
    } else {
        tex_sample<C>(tex, mips, D, tb, c, uv_da[pi], res);
    }
#pragma unroll
    for (int k = 0; k < C; k++) out[(size_t)pi * C + k] = res[k];
}

template <int C>
__global__ __launch_bounds__(256) void texture_bwd_kernel(const float* __restrict__ tex, const float* __restrict__ mips,
                                                          const TexDesc D, const float2* __restrict__ uv,
                                                          const float4* __restrict__ uv_da, const float* __restrict__ d_out,
                                                          long long npix, int HW, float* __restrict__ d_tex,
                                                          float* __restrict__ d_mips, float2* __restrict__ d_uv,
                                                          float4* __restrict__ d_uv_da) {
    const long long pi = (long long)blockIdx.x * 256 + threadIdx.x;
    if (pi >= npix) return;
    const int tb = D.TB == 1 ? 0 : (int)(pi / HW);
    const float2 c = uv[pi];
    float g[C];
    bool any = false;
#pragma unroll
    for (int k = 0; k < C; k++) {
        g[k] = d_out[(size_t)pi * C + k];
        any = any || g[k] != 0.f;
    }
    float2 guv = make_float2(0.f, 0.f);
    float4 gda = make_float4(0.f, 0.f, 0.f, 0.f);
    if (any) {
        if (uv_da == nullptr) {
            const Taps t = make_taps(c.x, c.y, D.W, D.H, C);
            float gfx, gfy, val[C];
            bilinear_bwd<C>(level_ptr(tex, mips, D, tb, 0), d_tex ? level_ptr_w(d_tex, d_mips, D, tb, 0) : nullptr, t, g, 1.0f, gfx,
                            gfy, val);
            guv.x = gfx * (float)D.W;
            guv.y = gfy * (float)D.H;
        } else {
            float val[C];
            tex_sample_bwd_uv<C>(tex, mips, D, tb, c, uv_da[pi], g, d_tex, d_mips, guv, gda, d_uv_da != nullptr, val);
        }
    }
    if (d_uv) d_uv[pi] = guv;
    if (d_uv_da) d_uv_da[pi] = gda;
}

// ------------------------------------------------------------------------------------------------------------
// Tiled backward: a workgroup owns a 16x16 pixel tile.  Neighbouring pixels hit the same texels (every texel of a level
// sampled near 1 texel/pixel receives ~4 bilinear taps), and device-scope float atomics to HBM-resident memory are the
// bottleneck of the per-pixel kernel (24 per covered pixel).  The tile therefore accumulates its taps in an LDS hash
// table keyed by (level, texel) -- LDS float atomics -- and flushes every touched texel ONCE: ~3x fewer global atomics.
// Probing is bounded; a tap that finds no slot goes straight to global memory, so the result never depends on the table.
// ------------------------------------------------------------------------------------------------------------
constexpr int TT = 16;            // tile edge (pixels)
constexpr int NSLOT = 1536;       // hash slots per tile (a 16x16 tile touches ~700-1400 texels over two levels)
constexpr unsigned EMPTY_KEY = 0xffffffffu;

// LDS accumulators are 64-bit FIXED POINT, not float: on gfx950 ds_add_f32 costs ~200 cycles per wave instruction whatever
// the conflict pattern, ds_add_u64 ~9 cycles (+2 per extra lane on the same address) -- tools/ubench/lds_atomics.hip.  The
// scale is a power of two chosen per tile from max|d_out| so that every product w*g keeps 40 fractional bits below the
// tile maximum (more than the 24 of fp32) and 2048 taps cannot overflow; integer sums are exact and order-independent.
template <int C>
struct TileAcc {
    unsigned* keys;
    unsigned long long* vals;
    float* d_tex;
    float* d_mips;
    const TexDesc* D;
    int tb;
    float scale;
    __device__ __forceinline__ void add(int level, int texelC, const float (&v)[C]) const {
        const unsigned key = ((unsigned)level << 27) | (unsigned)(texelC / C);
        unsigned slot = (unsigned)(((unsigned long long)(key * 2654435761u) * NSLOT) >> 32);
#pragma unroll 1
        for (int probe = 0; probe < 16; probe++) {
            const unsigned prev = atomicCAS(&keys[slot], EMPTY_KEY, key);
            if (prev == EMPTY_KEY || prev == key) {
#pragma unroll
                for (int k = 0; k < C; k++)
                    if (v[k] != 0.f) atomicAdd(&vals[slot * C + k], (unsigned long long)__float2ll_rn(v[k] * scale));
                return;
            }
            slot = slot + 1 == NSLOT ? 0 : slot + 1;
        }
        float* G = level_ptr_w(d_tex, d_mips, *D, tb, level);
#pragma unroll
        for (int k = 0; k < C; k++)
            if (v[k] != 0.f) atomicAdd(&G[texelC + k], v[k]);
    }
};

template <int C, bool WANT_UV>
__device__ __forceinline__ void bilinear_bwd_tile(const float* __restrict__ T, const TileAcc<C>& acc, bool use_acc, int level, const Taps& t,
                                                  const float (&g)[C], float wgt, float& gfx, float& gfy, float (&val)[C]) {
    gfx = 0.f;
    gfy = 0.f;
    const float w00 = (1.f - t.fx) * (1.f - t.fy), w10 = t.fx * (1.f - t.fy), w01 = (1.f - t.fx) * t.fy, w11 = t.fx * t.fy;
    float gk[C];
#pragma unroll
    for (int k = 0; k < C; k++) {
        gk[k] = g[k] * wgt;
        val[k] = 0.f;
        if constexpr (WANT_UV) {     // the texel values are only needed for the gradient w.r.t. uv / the level
            const float a00 = T[t.i00 + k], a10 = T[t.i10 + k], a01 = T[t.i01 + k], a11 = T[t.i11 + k];
            const float top = a00 + t.fx * (a10 - a00), bot = a01 + t.fx * (a11 - a01);
            val[k] = top + t.fy * (bot - top);
            gfx += gk[k] * ((1.f - t.fy) * (a10 - a00) + t.fy * (a11 - a01));
            gfy += gk[k] * (bot - top);
        }
    }
    if (use_acc) {
        float v[C];
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = w00 * gk[k];
        acc.add(level, t.i00, v);
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = w10 * gk[k];
        acc.add(level, t.i10, v);
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = w01 * gk[k];
        acc.add(level, t.i01, v);
#pragma unroll
        for (int k = 0; k < C; k++) v[k] = w11 * gk[k];
        acc.add(level, t.i11, v);
    }
}

template <int C, bool WANT_UV>
__global__ __launch_bounds__(TT * TT) void texture_bwd_tiled_kernel(const float* __restrict__ tex, const float* __restrict__ mips,
                                                                    const TexDesc D, const float2* __restrict__ uv,
                                                                    const float4* __restrict__ uv_da, const float* __restrict__ d_out,
                                                                    int H, int W, float* __restrict__ d_tex, float* __restrict__ d_mips,
                                                                    float2* __restrict__ d_uv, float4* __restrict__ d_uv_da, int dbg) {
    __shared__ unsigned keys[NSLOT];
    __shared__ unsigned long long vals[NSLOT * C];
    __shared__ unsigned smax;
    __shared__ int lmin_s, bb[4][4];     // per relative level: xmin, ymin, xmax, ymax of the texels touched by the tile
    __shared__ unsigned irregular;       // bit r: level lmin + r wraps around / too large -> flushed in slot order; bit 4: deeper levels
    const int tid = threadIdx.x;
    const int px = blockIdx.x * TT + (tid & (TT - 1)), py = blockIdx.y * TT + (tid >> 4), b = blockIdx.z;
    const bool inside = px < W && py < H;
    const size_t pi = ((size_t)b * H + (inside ? py : 0)) * W + (inside ? px : 0);
    const int tb = D.TB == 1 ? 0 : b;
    float g[C];
    bool any = false;
    float gmax = 0.f;
#pragma unroll
    for (int k = 0; k < C; k++) {
        g[k] = inside ? d_out[pi * C + k] : 0.f;
        any = any || g[k] != 0.f;
        gmax = fmaxf(gmax, fabsf(g[k]));
    }
    if (tid == 0) { smax = 0u; lmin_s = 99; irregular = 0u; }
    if (tid < 16) bb[tid >> 2][tid & 3] = (tid & 2) ? -1 : 0x7fffffff;
    const bool tile_any = __syncthreads_or(any ? 1 : 0) != 0;
    if (!tile_any) {          // background tile: nothing to accumulate
        if (inside) {
            if (d_uv) d_uv[pi] = make_float2(0.f, 0.f);
            if (d_uv_da) d_uv_da[pi] = make_float4(0.f, 0.f, 0.f, 0.f);
        }
        return;
    }
    const bool want_tex = d_tex != nullptr;
    LevelSel s;
    s.l0 = 0; s.f = 0.f; s.two = false; s.diff = false;
    if (any && uv_da != nullptr) s = select_level(uv_da[pi], D.W, D.H, D.L);
    if (want_tex) {
        int l = any ? s.l0 : 99;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) l = min(l, __shfl_xor(l, o, 64));
        if ((tid & 63) == 0) atomicMin(&lmin_s, l);
        for (int i = tid; i < NSLOT; i += TT * TT) keys[i] = EMPTY_KEY;
        for (int i = tid; i < NSLOT * C; i += TT * TT) vals[i] = 0ull;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) gmax = fmaxf(gmax, __shfl_xor(gmax, o, 64));
        if ((tid & 63) == 0) atomicMax(&smax, __float_as_uint(gmax));     // non-negative floats order like their bit patterns
    }
    __syncthreads();
    // power-of-two scale: max|g| * scale in [2^39, 2^40)
    int ex = 0;
    (void)frexpf(__uint_as_float(smax), &ex);
    const int sh = min(max(40 - ex, -100), 100);
    const float scale = ldexpf(1.0f, sh), inv_scale = ldexpf(1.0f, -sh);
    const TileAcc<C> acc{keys, vals, d_tex, d_mips, &D, tb, scale};
    const bool use_acc = want_tex && !(dbg & 256);
    const int lmin = lmin_s;
    float2 guv = make_float2(0.f, 0.f);
    float4 gda = make_float4(0.f, 0.f, 0.f, 0.f);
    // texel boxes of this lane's (up to two) levels, for the ordered flush
    int bl[2] = {-1, -1}, bx0[2] = {0, 0}, by0[2] = {0, 0}, bx1[2] = {0, 0}, by1[2] = {0, 0};
    if (any) {
        const float2 c = uv[pi];
        if (uv_da == nullptr) {
            const Taps t = make_taps(c.x, c.y, D.W, D.H, C);
            float gfx, gfy, val[C];
            bilinear_bwd_tile<C, WANT_UV>(level_ptr(tex, mips, D, tb, 0), acc, use_acc, 0, t, g, 1.0f, gfx, gfy, val);
            guv.x = gfx * (float)D.W;
            guv.y = gfy * (float)D.H;
            bl[0] = 0; bx0[0] = t.x0; by0[0] = t.y0; bx1[0] = t.x1; by1[0] = t.y1;
        } else {
            const int w0 = D.W >> s.l0, h0 = D.H >> s.l0;
            const Taps t0 = make_taps(c.x, c.y, w0, h0, C);
            const bool two = s.two && s.f > 0.0f;
            float gfx0, gfy0, c0[C];
            bilinear_bwd_tile<C, WANT_UV>(level_ptr(tex, mips, D, tb, s.l0), acc, use_acc, s.l0, t0, g, two ? 1.0f - s.f : 1.0f, gfx0, gfy0, c0);
            guv.x = gfx0 * (float)w0;
            guv.y = gfy0 * (float)h0;
            bl[0] = s.l0; bx0[0] = t0.x0; by0[0] = t0.y0; bx1[0] = t0.x1; by1[0] = t0.y1;
            if (two) {
                const int w1 = D.W >> (s.l0 + 1), h1 = D.H >> (s.l0 + 1);
                const Taps t1 = make_taps(c.x, c.y, w1, h1, C);
                float gfx1, gfy1, c1[C];
                bilinear_bwd_tile<C, WANT_UV>(level_ptr(tex, mips, D, tb, s.l0 + 1), acc, use_acc, s.l0 + 1, t1, g, s.f, gfx1, gfy1, c1);
                guv.x += gfx1 * (float)w1;
                guv.y += gfy1 * (float)h1;
                bl[1] = s.l0 + 1; bx0[1] = t1.x0; by0[1] = t1.y0; bx1[1] = t1.x1; by1[1] = t1.y1;
                if (s.diff && d_uv_da) {
                    float gf = 0.f;
#pragma unroll
                    for (int k = 0; k < C; k++) gf += g[k] * (c1[k] - c0[k]);
                    const float glam = gf * 0.5f / (s.lambda * 0.69314718056f);
                    const float q = s.l2n_sqrt > 0.f ? 0.5f / s.l2n_sqrt : 0.f;
                    const float gl2n = glam * q;
                    const float gA = 0.5f * glam + gl2n * 0.5f * (s.A - s.B);
                    const float gB = 0.5f * glam - gl2n * 0.5f * (s.A - s.B);
                    const float gC = gl2n * 2.0f * s.Cq;
                    const float gsx = 2.f * s.sx * gA + s.sy * gC, gsy = 2.f * s.sy * gB + s.sx * gC;
                    const float gtx = 2.f * s.tx * gA + s.ty * gC, gty = 2.f * s.ty * gB + s.tx * gC;
                    gda = make_float4(gsx * (float)D.W, gsy * (float)D.W, gtx * (float)D.H, gty * (float)D.H);
                }
            }
        }
    }
    if (inside) {
        if (d_uv) d_uv[pi] = guv;
        if (d_uv_da) d_uv_da[pi] = gda;
    }
    if (!want_tex || (dbg & 128)) return;
    // ---- flush ----
    // In slot (hash) order the 64 atomics of a wave land on 64 scattered cache lines and the memory system handles them one
    // transaction each (~70 G/s measured); walking each level's texel BOX in row-major order and looking the texels up in the
    // table makes consecutive lanes hit consecutive addresses.  Boxes that wrap around the texture edge (seams) or are too large
    // fall back to slot order.
    {
        unsigned irr = 0u;
#pragma unroll
        for (int r = 0; r < 4; r++) {
            int xmin = 0x7fffffff, ymin = 0x7fffffff, xmax = -1, ymax = -1;
#pragma unroll
            for (int q = 0; q < 2; q++) {
                if (bl[q] >= 0 && bl[q] - lmin == r) {
                    if (bx1[q] < bx0[q] || by1[q] < by0[q]) irr |= 1u << r;
                    else { xmin = bx0[q]; ymin = by0[q]; xmax = bx1[q]; ymax = by1[q]; }
                }
            }
            if (__ballot(xmax >= 0) == 0ull) continue;        // wave-uniform: nobody touches this level
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) {
                xmin = min(xmin, __shfl_xor(xmin, o, 64)); ymin = min(ymin, __shfl_xor(ymin, o, 64));
                xmax = max(xmax, __shfl_xor(xmax, o, 64)); ymax = max(ymax, __shfl_xor(ymax, o, 64));
            }
            if ((tid & 63) == 0) {
                atomicMin(&bb[r][0], xmin); atomicMin(&bb[r][1], ymin); atomicMax(&bb[r][2], xmax); atomicMax(&bb[r][3], ymax);
            }
        }
#pragma unroll
        for (int q = 0; q < 2; q++)
            if (bl[q] >= 0 && bl[q] - lmin > 3) irr |= 16u;
        if (__ballot(irr != 0u) != 0ull) {
#pragma unroll
            for (int o = 32; o > 0; o >>= 1) irr |= (unsigned)__shfl_xor((int)irr, o, 64);
            if ((tid & 63) == 0) atomicOr(&irregular, irr);
        }
    }
    __syncthreads();
    unsigned slot_order = irregular;      // levels (relative) that must be flushed in slot order
#pragma unroll 1
    for (int r = 0; r < 4; r++) {
        const int xmin = bb[r][0], ymin = bb[r][1], xmax = bb[r][2], ymax = bb[r][3];
        if (xmax < 0) continue;
        const int bw = xmax - xmin + 1, bh = ymax - ymin + 1;
        if (bw * bh > 6144 || (slot_order & (1u << r))) { slot_order |= 1u << r; continue; }
        const int level = lmin + r, wl = D.W >> level;
        float* G = level_ptr_w(d_tex, d_mips, D, tb, level);
        for (int i = tid; i < bw * bh; i += TT * TT) {
            const int yy = i / bw, xx = i - yy * bw;
            const unsigned texel = (unsigned)((ymin + yy) * wl + xmin + xx);
            const unsigned key = ((unsigned)level << 27) | texel;
            unsigned slot = (unsigned)(((unsigned long long)(key * 2654435761u) * NSLOT) >> 32);
            int found = -1;
#pragma unroll 1
            for (int probe = 0; probe < 16; probe++) {
                const unsigned k = keys[slot];
                if (k == key) { found = (int)slot; break; }
                if (k == EMPTY_KEY) break;
                slot = slot + 1 == NSLOT ? 0 : slot + 1;
            }
            if (found < 0) continue;
#pragma unroll
            for (int k = 0; k < C; k++) {
                const long long q = (long long)vals[found * C + k];
                if (q != 0) atomicAdd(&G[(size_t)texel * C + k], (float)q * inv_scale);
            }
        }
    }
    if (slot_order == 0u) return;
    for (int sidx = tid; sidx < NSLOT; sidx += TT * TT) {
        const unsigned key = keys[sidx];
        if (key == EMPTY_KEY) continue;
        const int level = (int)(key >> 27);
        const int r = level - lmin;
        if (!((r >= 0 && r < 4 && (slot_order & (1u << r))) || (r > 3 && (slot_order & 16u)))) continue;
        float* G = level_ptr_w(d_tex, d_mips, D, tb, level) + (size_t)(key & 0x7ffffffu) * C;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const long long q = (long long)vals[sidx * C + k];
            if (q != 0) atomicAdd(&G[k], (float)q * inv_scale);
        }
    }
}

// ------------------------------------------------------------------------------------------------------------
// Texture gradient through UV-SPACE binning (shared texture, TB == 1).
//
// The screen-tiled kernel above dedups taps only inside one 16x16 pixel tile of one frame; every frame of the batch and
// every neighbouring tile flushes the same texels again, so a 16-frame batch still issues ~20 M device-scope float atomics
// (the backward critical path of the fit step).  Here the covered pixels of ALL frames are first counting-sorted by the
// tile of the uv square they sample (NT x NT tiles); one workgroup per uv tile then accumulates all of its pixels' taps into
// a DENSE LDS image of the tile -- every mip level of the tile plus a one-texel halo, 64-bit fixed point as above -- and
// flushes each touched texel once.  Global atomics drop to (touched texels + halos) x C, their addresses are row-contiguous,
// and the result is independent of the pixel order (integer sums), i.e. deterministic up to the <= 4 halo contributions.
//   texbin_count   : per-tile pixel count and max|g|            (LDS histogram per 4096 pixels, then global atomics)
//   texbin_scan    : exclusive scan of the counts               (one workgroup)
//   texbin_scatter : pixel indices into their tile's list       (LDS ranks, one global cursor bump per workgroup and tile)
//   texgrad_tile   : accumulate + flush                         (one workgroup per non-empty tile)
// A pixel with uf = u - floor(u) belongs to tile tx = min(NT-1, int(uf * NT)); at level l (width w) its taps x0 = floor(uf*w - 0.5),
// x0 + 1 lie in [lo, hi + 1] with lo = floor(fl(tx/NT * w) - 0.5), hi = floor(fl((tx+1)/NT * w) - 0.5) -- fp32 rounding is
// monotone and tx/NT is exact, so the bounds hold for the rounded products too; a tap outside them (never observed) goes
// straight to global memory, so memory safety does not depend on the argument.
// ------------------------------------------------------------------------------------------------------------
constexpr int TG_PPT = 16;          // pixels per lane in the count / scatter passes (4096 pixels per workgroup)
struct TileGeo {
    int NT, cells;                                  // tiles per axis; LDS cells per tile over all levels
    int nx[MAX_LEVELS + 1], ny[MAX_LEVELS + 1], off[MAX_LEVELS + 1];
};

TileGeo make_tile_geo(const TexDesc& D) {
    TileGeo G;
    const int nt = texbin_nt(D.H, D.W);
    G.NT = nt;
    int o = 0;
    for (int l = 0; l <= D.L; l++) {
        G.nx[l] = (D.W >> l) / nt + 3;
        G.ny[l] = (D.H >> l) / nt + 3;
        G.off[l] = o;
        o += G.nx[l] * G.ny[l];
    }
    for (int l = D.L + 1; l <= MAX_LEVELS; l++) { G.nx[l] = G.ny[l] = 0; G.off[l] = o; }
    G.cells = o;
    return G;
}

__device__ __forceinline__ int tile_lo(int t, int NT, int w) { return (int)floorf(((float)t / (float)NT) * (float)w - 0.5f); }

// tile_ids (optional): the uv tile of every pixel as 16 bits (0xFFFF = no gradient), written by the producer of d_out
// (vhap_deferred_shade_bwd): the passes then read 2 B/px (+ d_out of the covered third in the count pass, for the per-tile scale) instead
// of uv + d_out = 20 B/px each.
template <int C, bool SCATTER>
__global__ __launch_bounds__(256) void texbin_pass_kernel(const float2* __restrict__ uv, const float* __restrict__ d_out, long long npix,
                                                          int NT, unsigned* __restrict__ counts, unsigned* __restrict__ tilemax,
                                                          const unsigned* __restrict__ offsets, unsigned* __restrict__ cursors,
                                                          unsigned* __restrict__ list, const unsigned short* __restrict__ tile_ids,
                                                          const float* __restrict__ keep) {
    extern __shared__ unsigned tg_sh[];          // [NT*NT] counts / ranks, [NT*NT] max|g| bits (count pass) or list bases (scatter pass)
    const int nt2 = NT * NT, tid = threadIdx.x;
    unsigned* shc = tg_sh;
    unsigned* shb = tg_sh + nt2;
    for (int i = tid; i < 2 * nt2; i += 256) tg_sh[i] = 0u;
    __syncthreads();
    int tile[TG_PPT];
    const long long p0 = (long long)blockIdx.x * (256 * TG_PPT) + tid;
    // (tile_ids path: the thread's TG_PPT ids in one batch of loads, then -- count pass -- the gradients of its covered pixels in a second one,
    // uncovered lanes re-reading pixel 0: one pixel per trip was id -> wait -> gradient -> wait, 2 x TG_PPT round trips in series)
    unsigned short ids[TG_PPT];
    float kp[TG_PPT];
#pragma unroll
    for (int k = 0; k < TG_PPT; k++) tile[k] = -1;
    if (tile_ids) {
#pragma unroll
        for (int k = 0; k < TG_PPT; k++) {
            const long long p = p0 + (long long)k * 256, pc = p < npix ? p : npix - 1;
            ids[k] = tile_ids[pc];
            kp[k] = *(keep ? keep + pc : reinterpret_cast<const float*>(tile_ids));       // (stand-in address, value unused)
        }
#pragma unroll
        for (int k = 0; k < TG_PPT; k++) {
            const long long p = p0 + (long long)k * 256;
            tile[k] = (p < npix && ids[k] != 0xFFFF && !(keep && kp[k] == 0.f)) ? (int)ids[k] : -1;   // (keep == 0: colour replaced, no gradient)
        }
        if (!SCATTER && d_out) {
            float gm[TG_PPT];
#pragma unroll
            for (int k = 0; k < TG_PPT; k++) {
                const long long q = tile[k] >= 0 ? p0 + (long long)k * 256 : 0;
                float m = 0.f;
#pragma unroll
                for (int c = 0; c < C; c++) m = fmaxf(m, fabsf(d_out[q * C + c]));
                gm[k] = m;
            }
#pragma unroll
            for (int k = 0; k < TG_PPT; k++)
                if (tile[k] >= 0) atomicMax(&shb[tile[k]], __float_as_uint(gm[k]));
        }
#pragma unroll
        for (int k = 0; k < TG_PPT; k++)
            if (tile[k] >= 0) atomicAdd(&shc[tile[k]], 1u);
    }
#pragma unroll
    for (int k = 0; k < TG_PPT; k++) {
        const long long p = p0 + (long long)k * 256;
        if (!tile_ids && p < npix) {
            float gmax = 0.f;
#pragma unroll
            for (int c = 0; c < C; c++) gmax = fmaxf(gmax, fabsf(d_out[p * C + c]));
            if (gmax != 0.f) {
                tile[k] = tile_of(uv[p], NT);
                atomicAdd(&shc[tile[k]], 1u);
                if (!SCATTER) atomicMax(&shb[tile[k]], __float_as_uint(gmax));      // non-negative floats order like their bit patterns
            }
        }
    }
    __syncthreads();
    for (int i = tid; i < nt2; i += 256) {
        const unsigned c = shc[i];
        if (c == 0u) continue;
        if (SCATTER) {
            shb[i] = offsets[i] + atomicAdd(&cursors[i], c);
            shc[i] = 0u;
        } else {
            atomicAdd(&counts[i], c);
            if (tilemax) atomicMax(&tilemax[i], shb[i]);
        }
    }
    if (!SCATTER) return;
    __syncthreads();
#pragma unroll
    for (int k = 0; k < TG_PPT; k++)
        if (tile[k] >= 0) list[shb[tile[k]] + atomicAdd(&shc[tile[k]], 1u)] = (unsigned)(p0 + (long long)k * 256);
}

// exclusive scan of counts[0..n) -> offsets[0..n], n <= 4096; clears the scatter cursors
// `order` (optional, [n]): the tiles sorted by the power of two of their pixel count, fullest first -- the accumulation kernel takes its
// tiles in this order, so that the long ones start first instead of setting the tail (empty tiles come last)
__global__ __launch_bounds__(1024) void texbin_scan_kernel(const unsigned* __restrict__ counts, int n, unsigned* __restrict__ offsets,
                                                           unsigned* __restrict__ cursors, unsigned* __restrict__ order) {
    __shared__ unsigned wtot[16];
    __shared__ unsigned bh[33], bo[33];
    if (order) {
        if (threadIdx.x < 33) bh[threadIdx.x] = 0u;
        __syncthreads();
        const int per_ = (n + 1023) / 1024, j0 = threadIdx.x * per_, j1 = min(j0 + per_, n);
        for (int i = j0; i < j1; i++) atomicAdd(&bh[counts[i] ? 32 - __clz((int)counts[i]) : 0], 1u);
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned pos = 0u;
            for (int b = 32; b >= 0; b--) { bo[b] = pos; pos += bh[b]; }
        }
        __syncthreads();
        for (int i = j0; i < j1; i++) order[atomicAdd(&bo[counts[i] ? 32 - __clz((int)counts[i]) : 0], 1u)] = (unsigned)i;
    }
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int per = (n + 1023) / 1024, i0 = tid * per, i1 = min(i0 + per, n);
    unsigned mine = 0u;
    for (int i = i0; i < i1; i++) mine += counts[i];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned pre = incl - mine;
    for (int w = 0; w < wave; w++) pre += wtot[w];
    for (int i = i0; i < i1; i++) {
        offsets[i] = pre;
        cursors[i] = 0u;
        pre += counts[i];
    }
    if (tid == 1023) offsets[n] = pre;
}

template <int C>
__global__ __launch_bounds__(256) void texgrad_tile_kernel(const TexDesc D, const TileGeo G, const float2* __restrict__ uv,
                                                           const float4* __restrict__ uv_da, const float* __restrict__ d_out,
                                                           const unsigned* __restrict__ offsets, const unsigned* __restrict__ list,
                                                           const unsigned* __restrict__ tilemax, float* __restrict__ d_tex,
                                                           float* __restrict__ d_mips, const float* __restrict__ gmax_bound,
                                                           const unsigned* __restrict__ order) {
    extern __shared__ unsigned long long tg_vals[];      // [G.cells * C]
    const int t = order ? (int)order[blockIdx.x] : (int)blockIdx.x, tid = threadIdx.x;
    const unsigned beg = offsets[t], end = offsets[t + 1];
    if (beg == end) return;
    for (int i = tid; i < G.cells * C; i += 256) tg_vals[i] = 0ull;
    // the per-level tile geometry in LDS: indexed with a per-lane level, the kernel-argument arrays are fetched with GLOBAL loads -- three
    // dependent round trips in front of every level's LDS atomics (like TexDesc::off, tex_sample.h: level_off)
    __shared__ int s_nx[MAX_LEVELS + 1], s_ny[MAX_LEVELS + 1], s_off[MAX_LEVELS + 1];
    if (tid <= MAX_LEVELS) { s_nx[tid] = G.nx[tid]; s_ny[tid] = G.ny[tid]; s_off[tid] = G.off[tid]; }
    __shared__ unsigned s_tmax;
    unsigned tmax_bits;
    if (tilemax) {
        tmax_bits = tilemax[t];
    } else if (gmax_bound) {         // an upper bound of |d_out| supplied by the caller: one scale for all tiles, no extra pass over the list
        tmax_bits = __float_as_uint(fabsf(gmax_bound[0]));
        if (tmax_bits == 0u) return;
    } else {            // the list was sorted before the gradient existed (tile ids from the forward): find the tile's max |g| here
        if (tid == 0) s_tmax = 0u;
        __syncthreads();
        float m = 0.f;
        for (unsigned i = beg + tid; i < end; i += 256) {
            const size_t p = list[i];
#pragma unroll
            for (int k = 0; k < C; k++) m = fmaxf(m, fabsf(d_out[p * C + k]));
        }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if ((tid & 63) == 0) atomicMax(&s_tmax, __float_as_uint(m));
        __syncthreads();
        tmax_bits = s_tmax;
        if (tmax_bits == 0u) return;         // nothing to add (uniform)
    }
    int ex = 0;
    (void)frexpf(__uint_as_float(tmax_bits), &ex);
    const int sh = min(max(40 - ex, -100), 100);
    const float scale = ldexpf(1.0f, sh), inv_scale = ldexpf(1.0f, -sh);
    const int tx = t % G.NT, ty = t / G.NT;
    __syncthreads();

    auto tap_level = [&](int l, float2 c, const float (&g)[C], float wgt) {
        const int w = D.W >> l, h = D.H >> l;
        const float u = c.x - floorf(c.x), v = c.y - floorf(c.y);
        const float x = u * (float)w - 0.5f, y = v * (float)h - 0.5f;
        const float x0f = floorf(x), y0f = floorf(y);
        const float fx = x - x0f, fy = y - y0f;
        const int x0 = (int)x0f, y0 = (int)y0f;
        const int lx = x0 - tile_lo(tx, G.NT, w), ly = y0 - tile_lo(ty, G.NT, h);
        const float w00 = (1.f - fx) * (1.f - fy), w10 = fx * (1.f - fy), w01 = (1.f - fx) * fy, w11 = fx * fy;
        const float wq[4] = {w00, w10, w01, w11};
        const int nxl = s_nx[l], nyl = s_ny[l], offl = s_off[l];
        const bool local = lx >= 0 && ly >= 0 && lx + 1 < nxl && ly + 1 < nyl;
#pragma unroll
        for (int q = 0; q < 4; q++) {
            const int dx = q & 1, dy = q >> 1;
            if (local) {
                unsigned long long* cell = tg_vals + (size_t)(offl + (ly + dy) * nxl + lx + dx) * C;
#pragma unroll
                for (int k = 0; k < C; k++) {
                    const float val = wq[q] * (g[k] * wgt);
                    if (val != 0.f) atomicAdd(&cell[k], (unsigned long long)__float2ll_rn(val * scale));
                }
            } else {
                int gx = x0 + dx, gy = y0 + dy;
                gx = gx < 0 ? gx + w : (gx >= w ? gx - w : gx);
                gy = gy < 0 ? gy + h : (gy >= h ? gy - h : gy);
                if (gx >= w) gx -= w;
                if (gy >= h) gy -= h;
                float* Gp = level_ptr_w(d_tex, d_mips, D, 0, l) + ((size_t)gy * w + gx) * C;
#pragma unroll
                for (int k = 0; k < C; k++) {
                    const float val = wq[q] * (g[k] * wgt);
                    if (val != 0.f) atomicAdd(&Gp[k], val);
                }
            }
        }
    };

    // TG_UNR pixels per lane and round: the list entries first, then all their attributes, then the accumulation -- a pixel is a chain of
    // two dependent global loads (~2 us each under load) and the fullest tiles hold 8 pixels per lane; un-batched, those tiles set the
    // kernel's duration
    constexpr int TG_UNR = 4;
    for (unsigned i0 = beg + tid; i0 < end; i0 += 256 * TG_UNR) {
        size_t p[TG_UNR];
        bool ok[TG_UNR];
#pragma unroll
        for (int u = 0; u < TG_UNR; u++) {
            ok[u] = i0 + 256u * u < end;
            p[u] = ok[u] ? (size_t)list[i0 + 256u * u] : (size_t)0;
        }
        float g[TG_UNR][C];
        float2 c[TG_UNR];
        float4 da[TG_UNR];
#pragma unroll
        for (int u = 0; u < TG_UNR; u++) {      // (unconditional: lanes past the end re-read pixel 0; a load per `if (ok)` is a round trip each)
#pragma unroll
            for (int k = 0; k < C; k++) g[u][k] = d_out[p[u] * C + k];
            c[u] = uv[p[u]];
            da[u] = *(uv_da ? uv_da + p[u] : reinterpret_cast<const float4*>(offsets));      // (stand-in: >= 65 words, value unused)
        }
#pragma unroll
        for (int u = 0; u < TG_UNR; u++) {
            if (!ok[u]) continue;
            if (uv_da == nullptr) {
                tap_level(0, c[u], g[u], 1.0f);
            } else {
                const LevelSel s = select_level(da[u], D.W, D.H, D.L);
                const bool two = s.two && s.f > 0.0f;
                tap_level(s.l0, c[u], g[u], two ? 1.0f - s.f : 1.0f);
                if (two) tap_level(s.l0 + 1, c[u], g[u], s.f);
            }
        }
    }
    __syncthreads();
    // flush: level by level, rows of the tile image are contiguous texel runs in memory
    for (int l = 0; l <= D.L; l++) {
        const int w = D.W >> l, h = D.H >> l, nx = G.nx[l], n = nx * G.ny[l];
        const int lox = tile_lo(tx, G.NT, w), loy = tile_lo(ty, G.NT, h);
        float* Gl = level_ptr_w(d_tex, d_mips, D, 0, l);
        for (int i = tid; i < n; i += 256) {
            const unsigned long long* cell = tg_vals + (size_t)(G.off[l] + i) * C;
            bool nz = false;
#pragma unroll
            for (int k = 0; k < C; k++) nz = nz || cell[k] != 0ull;
            if (!nz) continue;
            const int cy = i / nx, cx = i - cy * nx;
            int gx = (lox + cx) % w, gy = (loy + cy) % h;
            if (gx < 0) gx += w;
            if (gy < 0) gy += h;
            float* Gp = Gl + ((size_t)gy * w + gx) * C;
#pragma unroll
            for (int k = 0; k < C; k++) {
                const long long q = (long long)cell[k];
                if (q != 0) atomicAdd(&Gp[k], (float)q * inv_scale);
            }
        }
    }
}

template <typename F>
int dispatch_C(int C, F&& f) {
    switch (C) {
        case 1: return f(std::integral_constant<int, 1>());
        case 2: return f(std::integral_constant<int, 2>());
        case 3: return f(std::integral_constant<int, 3>());
        case 4: return f(std::integral_constant<int, 4>());
        default: return VHAP_E_BADDIM;
    }
}

int check_tex(int TB, int Ht, int Wt, int C) {
    if (TB <= 0 || Ht <= 0 || Wt <= 0 || C <= 0 || C > 4) return VHAP_E_BADDIM;
    if (Ht > 16384 || Wt > 16384) return VHAP_E_BADDIM;
    return VHAP_OK;
}

}  // namespace

extern "C" int vhap_texture_num_levels(int Ht, int Wt) {
    VHAP_ENTER(); return num_levels(Ht, Wt); }

extern "C" size_t vhap_texture_mip_floats(int TB, int Ht, int Wt, int C) {
    if (check_tex(TB, Ht, Wt, C) != VHAP_OK) return 0;
    return (size_t)make_desc(TB, Ht, Wt, C).per_tex * TB;
}

extern "C" int vhap_texture_mip_build(const float* tex, int TB, int Ht, int Wt, int C, float* mips, vhap_stream_t stream) {
    VHAP_ENTER();
    if (int e = check_tex(TB, Ht, Wt, C)) return e;
    const TexDesc D = make_desc(TB, Ht, Wt, C);
    if (D.L == 0) return VHAP_OK;
    if (!tex || !mips) return VHAP_E_NULLPTR;
    // first level whose source is small enough for the single-workgroup tail (never level 1: its source is `tex`)
    int l_tail = D.L + 1;
    for (int l = 2; l <= D.L; l++)
        if ((Ht >> (l - 1)) <= TAIL_MAX && (Wt >> (l - 1)) <= TAIL_MAX) { l_tail = l; break; }
    for (int l = 1; l <= D.L && l < l_tail; l++) {
        const int h = Ht >> l, w = Wt >> l;
        const float* src = l == 1 ? tex : mips + D.off[l - 1];
        const long long sstride = l == 1 ? (long long)Ht * Wt * C : D.per_tex;
        const long long n = (long long)TB * h * w * C;
        mip_down_kernel<<<vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(src, mips + D.off[l], TB, h, w, C, sstride, D.per_tex);
        VHAP_LAUNCH_CHECK();
    }
    if (l_tail <= D.L) {
        mip_down_tail_kernel<<<1, 1024, 0, vhap_stream(stream)>>>(mips, D, l_tail);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

// levels first_level .. L from level first_level - 1 (which the caller has already: vhap_tex_prep_mip1_fwd writes level 1 itself)
extern "C" int vhap_texture_mip_build_from(const float* tex, int TB, int Ht, int Wt, int C, float* mips, int first_level,
                                           vhap_stream_t stream) {
    VHAP_ENTER();
    if (int e = check_tex(TB, Ht, Wt, C)) return e;
    const TexDesc D = make_desc(TB, Ht, Wt, C);
    if (D.L == 0 || first_level > D.L) return VHAP_OK;
    if (!mips || (first_level <= 1 && !tex)) return VHAP_E_NULLPTR;
    if (first_level < 1) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    int l_tail = D.L + 1;
    for (int l = 2; l <= D.L; l++)
        if ((Ht >> (l - 1)) <= TAIL_MAX && (Wt >> (l - 1)) <= TAIL_MAX) { l_tail = l; break; }
    int l = first_level;
    while (l <= D.L && l < l_tail) {
        const int hs = Ht >> (l - 1), ws = Wt >> (l - 1);           // source extents (level l - 1)
        const float* src = l == 1 ? tex : mips + D.off[l - 1];
        if (C == 3 && l + 3 <= D.L && hs % 32 == 0 && ws % 32 == 0) {
            // four levels at once (the tail kernel takes over where the sources get small)
            mip_down4_kernel<3><<<dim3(ws / 32, hs / 32, TB), 256, 0, st>>>(src, mips, D, l - 1);
            VHAP_LAUNCH_CHECK();
            l += 4;
            continue;
        }
        const int h = Ht >> l, w = Wt >> l;
        const long long sstride = l == 1 ? (long long)Ht * Wt * C : D.per_tex;
        const long long n = (long long)TB * h * w * C;
        mip_down_kernel<<<vhap_cdiv(n, 256), 256, 0, st>>>(src, mips + D.off[l], TB, h, w, C, sstride, D.per_tex);
        VHAP_LAUNCH_CHECK();
        l++;
    }
    if (l <= D.L) {
        mip_down_tail_kernel<<<1, 1024, 0, st>>>(mips, D, l);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

extern "C" int vhap_texture_mip_fold(float* d_tex, float* d_mips, int TB, int Ht, int Wt, int C, int stop_level, vhap_stream_t stream) {
    VHAP_ENTER();
    if (int e = check_tex(TB, Ht, Wt, C)) return e;
    const TexDesc D = make_desc(TB, Ht, Wt, C);
    if (D.L == 0) return VHAP_OK;
    if (stop_level < 0 || stop_level > D.L) return VHAP_E_BADDIM;
    if ((!d_tex && stop_level == 0) || !d_mips) return VHAP_E_NULLPTR;
    // coarse -> fine; the levels whose FINE side is at most TAIL_MAX^2 in one single-workgroup launch
    int l_tail = D.L + 1;      // levels >= l_tail are folded by the tail kernel (into level l_tail - 1)
    for (int l = 2; l <= D.L; l++)
        if ((Ht >> (l - 1)) <= TAIL_MAX && (Wt >> (l - 1)) <= TAIL_MAX) { l_tail = l; break; }
    int l_top = D.L;          // coarsest level not yet folded
    if (l_tail <= D.L) {      // levels D.L .. l_last+1 in one launch, into level l_last
        const int l_last = max(l_tail - 1, stop_level);
        if (D.L > l_last) {
            if (l_last == 0) return VHAP_E_UNSUPPORTED;       // (cannot happen: l_tail >= 2)
            mip_fold_tail_kernel<<<1, 1024, 0, vhap_stream(stream)>>>(d_mips, D, l_last);
            VHAP_LAUNCH_CHECK();
        }
        l_top = l_last;
    }
    for (int l = l_top; l > stop_level; l--) {
        const int h = Ht >> (l - 1), w = Wt >> (l - 1);   // fine extents
        float* fine = l == 1 ? d_tex : d_mips + D.off[l - 1];
        const long long fstride = l == 1 ? (long long)Ht * Wt * C : D.per_tex;
        const long long n = (long long)TB * h * w * C;
        mip_fold_kernel<<<vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(fine, d_mips + D.off[l], TB, h, w, C, fstride, D.per_tex);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

// The whole pyramid folded into level 0 in ONE pass by GATHERING: texel (y, x) of level 0 += sum_l 4^-l level_l[y >> l][x >> l] -- exactly
// what the cascade of vhap_texture_mip_fold(stop_level = 0) leaves there, without its L dependent read-modify-write launches (74 us at
// T = 2048; this pass: one read + one write of level 0 and cache hits on the 17 MB above it).  The loads of a texel are one unconditional
// batch over a compile-time bound (levels beyond L re-read the last one with weight 0).  TB = 1.
template <int C>
__global__ __launch_bounds__(256) void mip_fold_gather_kernel(float* __restrict__ d_tex, const float* __restrict__ d_mips, const TexDesc D) {
    const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
    if (x >= D.W) return;
    float g[C];
    float* p = d_tex + ((size_t)y * D.W + x) * C;
#pragma unroll
    for (int c = 0; c < C; c++) g[c] = p[c];
    float m[MAX_LEVELS][C], sc[MAX_LEVELS];
    float w = 0.25f;
#pragma unroll
    for (int u = 0; u < MAX_LEVELS; u++) {
        const int l = min(u + 1, D.L);
        const float* q = d_mips + D.off[l] + ((size_t)(y >> l) * (D.W >> l) + (x >> l)) * C;
#pragma unroll
        for (int c = 0; c < C; c++) m[u][c] = q[c];
        sc[u] = u + 1 <= D.L ? w : 0.f;
        w *= 0.25f;
    }
#pragma unroll
    for (int u = 0; u < MAX_LEVELS; u++)
#pragma unroll
        for (int c = 0; c < C; c++) g[c] += sc[u] * m[u][c];
#pragma unroll
    for (int c = 0; c < C; c++) p[c] = g[c];
}

extern "C" int vhap_texture_mip_fold_gather(float* d_tex, const float* d_mips, int Ht, int Wt, int C, vhap_stream_t stream) {
    VHAP_ENTER();
    if (int e = check_tex(1, Ht, Wt, C)) return e;
    if (!d_tex || !d_mips) return VHAP_E_NULLPTR;
    const TexDesc D = make_desc(1, Ht, Wt, C);
    if (D.L == 0) return VHAP_OK;
    return dispatch_C(C, [&](auto c) {
        constexpr int CC = decltype(c)::value;
        mip_fold_gather_kernel<CC><<<dim3(vhap_cdiv(Wt, 256), Ht), 256, 0, vhap_stream(stream)>>>(d_tex, d_mips, D);
        VHAP_LAUNCH_CHECK();
        return VHAP_OK;
    });
}

extern "C" int vhap_texture_fwd(const float* tex, const float* mips, int TB, int Ht, int Wt, int C, const float* uv,
                                const float* uv_da, int B, int H, int W, float* out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!tex || !uv || !out) return VHAP_E_NULLPTR;
    if (int e = check_tex(TB, Ht, Wt, C)) return e;
    if (B <= 0 || H <= 0 || W <= 0 || (TB != 1 && TB != B)) return VHAP_E_BADDIM;
    const TexDesc D = make_desc(TB, Ht, Wt, C);
    if (uv_da && D.L > 0 && !mips) return VHAP_E_NULLPTR;
    const long long npix = (long long)B * H * W;
    return dispatch_C(C, [&](auto c) {
        texture_fwd_kernel<decltype(c)::value><<<vhap_cdiv(npix, 256), 256, 0, vhap_stream(stream)>>>(
            tex, mips, D, reinterpret_cast<const float2*>(uv), reinterpret_cast<const float4*>(uv_da), npix, H * W, out);
        VHAP_LAUNCH_CHECK();
        return VHAP_OK;
    });
}

extern "C" int vhap_texture_bwd(const float* tex, const float* mips, int TB, int Ht, int Wt, int C, const float* uv,
                                const float* uv_da, const float* d_out, int B, int H, int W, float* d_tex, float* d_mips,
                                float* d_uv, float* d_uv_da, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!tex || !uv || !d_out) return VHAP_E_NULLPTR;
    if (int e = check_tex(TB, Ht, Wt, C)) return e;
    if (B <= 0 || H <= 0 || W <= 0 || (TB != 1 && TB != B)) return VHAP_E_BADDIM;
    const TexDesc D = make_desc(TB, Ht, Wt, C);
    if (uv_da && D.L > 0 && (!mips || (d_tex && !d_mips))) return VHAP_E_NULLPTR;
    const long long npix = (long long)B * H * W;
    // the tile machinery pays off only when there is a texture gradient to accumulate: a uv-only call is 1.6x faster per pixel
    const bool tiled = d_tex != nullptr && !(vhap_g_debug_flags & 32) && (long long)Ht * Wt < (1ll << 27);      // (flag 32: A/B switch)
    if (tiled) {
        const bool want_uv = d_uv != nullptr || d_uv_da != nullptr;     // a texture-gradient-only call skips the texel gathers
        return dispatch_C(C, [&](auto c) {
            constexpr int CC = decltype(c)::value;
            const dim3 grid(vhap_cdiv(W, TT), vhap_cdiv(H, TT), B);
            if (want_uv)
                texture_bwd_tiled_kernel<CC, true><<<grid, TT * TT, 0, vhap_stream(stream)>>>(
                    tex, mips, D, reinterpret_cast<const float2*>(uv), reinterpret_cast<const float4*>(uv_da), d_out, H, W, d_tex, d_mips,
                    reinterpret_cast<float2*>(d_uv), reinterpret_cast<float4*>(d_uv_da), vhap_g_debug_flags);
            else
                texture_bwd_tiled_kernel<CC, false><<<grid, TT * TT, 0, vhap_stream(stream)>>>(
                    tex, mips, D, reinterpret_cast<const float2*>(uv), reinterpret_cast<const float4*>(uv_da), d_out, H, W, d_tex, d_mips,
                    nullptr, nullptr, vhap_g_debug_flags);
            VHAP_LAUNCH_CHECK();
            return VHAP_OK;
        });
    }
    return dispatch_C(C, [&](auto c) {
        texture_bwd_kernel<decltype(c)::value><<<vhap_cdiv(npix, 256), 256, 0, vhap_stream(stream)>>>(
            tex, mips, D, reinterpret_cast<const float2*>(uv), reinterpret_cast<const float4*>(uv_da), d_out, npix, H * W, d_tex,
            d_mips, reinterpret_cast<float2*>(d_uv), reinterpret_cast<float4*>(d_uv_da));
        VHAP_LAUNCH_CHECK();
        return VHAP_OK;
    });
}

extern "C" size_t vhap_texture_grad_binned_work_bytes(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return texbin_layout((long long)B * H * W).total;
}

static int texture_grad_binned_impl(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out, int B, int H, int W,
                                    float* d_tex, float* d_mips, void* work, size_t work_bytes, int call_flags, vhap_stream_t stream,
                                    const unsigned short* tile_ids = nullptr) {
    VHAP_ENTER();
    if (!uv || !d_out || !d_tex || !work) return VHAP_E_NULLPTR;
    if (int e = check_tex(1, Ht, Wt, C)) return e;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    if (npix >= (1ll << 32)) return VHAP_E_BADDIM;
    const TexDesc D = make_desc(1, Ht, Wt, C);
    if (uv_da && D.L > 0 && !d_mips) return VHAP_E_NULLPTR;
    const TileGeo G = make_tile_geo(D);
    const size_t lds = (size_t)G.cells * C * sizeof(unsigned long long);
    if ((Wt > Ht ? Wt : Ht) / G.NT > 32 || lds > 64 * 1024) return VHAP_E_UNSUPPORTED;       // larger textures: vhap_texture_bwd
    const TexBinWs l = texbin_layout(npix);
    if (work_bytes < l.total) return VHAP_E_WORKSPACE;
    char* w = static_cast<char*>(work);
    unsigned* counts = reinterpret_cast<unsigned*>(w + l.counts);
    unsigned* tilemax = reinterpret_cast<unsigned*>(w + l.tilemax);
    unsigned* cursors = reinterpret_cast<unsigned*>(w + l.cursors);
    unsigned* offsets = reinterpret_cast<unsigned*>(w + l.offsets);
    unsigned* list = reinterpret_cast<unsigned*>(w + l.list);
    const int nt2 = G.NT * G.NT;
    hipStream_t st = vhap_stream(stream);
    const bool counted = (call_flags & VHAP_CALL_TEXBIN_COUNTED) != 0;     // counts / tilemax already filled in by vhap_deferred_shade_bwd
    if (!counted) {
        vhap_zero_async(w + l.counts, (size_t)2 * TG_MAX_NT * TG_MAX_NT * 4, st);      // counts + tilemax
        VHAP_LAUNCH_CHECK();
    }
    const int nwg = vhap_cdiv(npix, 256 * TG_PPT);
    const size_t hist = (size_t)2 * nt2 * sizeof(unsigned);
    return dispatch_C(C, [&](auto c) {
        constexpr int CC = decltype(c)::value;
        const float2* uv2 = reinterpret_cast<const float2*>(uv);
        if (!counted) {
            texbin_pass_kernel<CC, false><<<nwg, 256, hist, st>>>(uv2, d_out, npix, G.NT, counts, tilemax, nullptr, nullptr, nullptr, tile_ids, nullptr);
            VHAP_LAUNCH_CHECK();
        }
        texbin_scan_kernel<<<1, 1024, 0, st>>>(counts, nt2, offsets, cursors, nullptr);
        VHAP_LAUNCH_CHECK();
        texbin_pass_kernel<CC, true><<<nwg, 256, hist, st>>>(uv2, d_out, npix, G.NT, nullptr, nullptr, offsets, cursors, list, tile_ids, nullptr);
        VHAP_LAUNCH_CHECK();
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(texgrad_tile_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return (int)VHAP_E_HIP;
        texgrad_tile_kernel<CC><<<nt2, 256, lds, st>>>(D, G, uv2, reinterpret_cast<const float4*>(uv_da), d_out, offsets, list, tilemax, d_tex, d_mips,
                                                       nullptr, nullptr);
        VHAP_LAUNCH_CHECK();
        return (int)VHAP_OK;
    });
}

extern "C" int vhap_texture_grad_binned(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out, int B, int H, int W,
                                        float* d_tex, float* d_mips, void* work, size_t work_bytes, vhap_stream_t stream) {
    return texture_grad_binned_impl(Ht, Wt, C, uv, uv_da, d_out, B, H, W, d_tex, d_mips, work, work_bytes, 0, stream);
}

extern "C" int vhap_texture_grad_binned_counted(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out, int B, int H,
                                                int W, float* d_tex, float* d_mips, void* work, size_t work_bytes, vhap_stream_t stream) {
    return texture_grad_binned_impl(Ht, Wt, C, uv, uv_da, d_out, B, H, W, d_tex, d_mips, work, work_bytes, VHAP_CALL_TEXBIN_COUNTED, stream);
}

// same, with the uv tile of every pixel supplied by the producer of d_out (tile_ids [B,H,W] uint16, 0xFFFF = no gradient;
// vhap_deferred_shade_bwd writes them): the two sorting passes read 2 B per pixel instead of 20
extern "C" int vhap_texture_grad_binned_ids(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out,
                                            const uint16_t* tile_ids, int B, int H, int W, float* d_tex, float* d_mips, void* work,
                                            size_t work_bytes, vhap_stream_t stream) {
    if (!tile_ids) return VHAP_E_NULLPTR;
    return texture_grad_binned_impl(Ht, Wt, C, uv, uv_da, d_out, B, H, W, d_tex, d_mips, work, work_bytes, 0, stream, tile_ids);
}

// The sort and the accumulation as two calls, for a caller that knows every pixel's uv tile BEFORE the gradient exists (the deferred-shading
// forward writes tile_ids): vhap_texbin_sort_ids counting-sorts the pixels by tile into `work` (three small launches, no dependence on
// d_out -- it can run next to the rest of the forward pass), vhap_texture_grad_binned_sorted then only accumulates (one launch on the
// backward's critical path instead of four); pixels of the lists whose gradient turns out to be zero add nothing.
extern "C" int vhap_texbin_sort_ids(const uint16_t* tile_ids, const float* keep, int Ht, int Wt, int B, int H, int W, void* work,
                                    size_t work_bytes, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!tile_ids || !work) return VHAP_E_NULLPTR;
    if (int e = check_tex(1, Ht, Wt, 3)) return e;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    if (npix >= (1ll << 32)) return VHAP_E_BADDIM;
    const TexDesc D = make_desc(1, Ht, Wt, 3);
    const TileGeo G = make_tile_geo(D);
    if ((Wt > Ht ? Wt : Ht) / G.NT > 32) return VHAP_E_UNSUPPORTED;
    const TexBinWs l = texbin_layout(npix);
    if (work_bytes < l.total) return VHAP_E_WORKSPACE;
    char* w = static_cast<char*>(work);
    unsigned* counts = reinterpret_cast<unsigned*>(w + l.counts);
    unsigned* cursors = reinterpret_cast<unsigned*>(w + l.cursors);
    unsigned* offsets = reinterpret_cast<unsigned*>(w + l.offsets);
    unsigned* list = reinterpret_cast<unsigned*>(w + l.list);
    const int nt2 = G.NT * G.NT;
    hipStream_t st = vhap_stream(stream);
    vhap_zero_async(w + l.counts, (size_t)TG_MAX_NT * TG_MAX_NT * 4, st);
    VHAP_LAUNCH_CHECK();
    const int nwg = vhap_cdiv(npix, 256 * TG_PPT);
    const size_t hist = (size_t)2 * nt2 * sizeof(unsigned);
    texbin_pass_kernel<3, false><<<nwg, 256, hist, st>>>(nullptr, nullptr, npix, G.NT, counts, nullptr, nullptr, nullptr, nullptr, tile_ids, keep);
    VHAP_LAUNCH_CHECK();
    texbin_scan_kernel<<<1, 1024, 0, st>>>(counts, nt2, offsets, cursors, reinterpret_cast<unsigned*>(w + l.tilemax));     // (tilemax slot: the tile order)
    VHAP_LAUNCH_CHECK();
    texbin_pass_kernel<3, true><<<nwg, 256, hist, st>>>(nullptr, nullptr, npix, G.NT, nullptr, nullptr, offsets, cursors, list, tile_ids, keep);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_texture_grad_binned_sorted(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out, int B, int H,
                                               int W, float* d_tex, float* d_mips, const void* work, size_t work_bytes,
                                               const float* gmax_bound, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!uv || !d_out || !d_tex || !work) return VHAP_E_NULLPTR;
    if (int e = check_tex(1, Ht, Wt, C)) return e;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    const TexDesc D = make_desc(1, Ht, Wt, C);
    if (uv_da && D.L > 0 && !d_mips) return VHAP_E_NULLPTR;
    const TileGeo G = make_tile_geo(D);
    const size_t lds = (size_t)G.cells * C * sizeof(unsigned long long);
    if ((Wt > Ht ? Wt : Ht) / G.NT > 32 || lds > 64 * 1024) return VHAP_E_UNSUPPORTED;
    const TexBinWs l = texbin_layout(npix);
    if (work_bytes < l.total) return VHAP_E_WORKSPACE;
    const char* w = static_cast<const char*>(work);
    const unsigned* offsets = reinterpret_cast<const unsigned*>(w + l.offsets);
    const unsigned* list = reinterpret_cast<const unsigned*>(w + l.list);
    hipStream_t st = vhap_stream(stream);
    return dispatch_C(C, [&](auto c) {
        constexpr int CC = decltype(c)::value;
        if (lds > 48 * 1024 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(texgrad_tile_kernel<CC>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return (int)VHAP_E_HIP;
        texgrad_tile_kernel<CC><<<G.NT * G.NT, 256, lds, st>>>(D, G, reinterpret_cast<const float2*>(uv), reinterpret_cast<const float4*>(uv_da), d_out,
                                                               offsets, list, nullptr, d_tex, d_mips, gmax_bound,
                                                               (vhap_g_debug_flags & 8192) ? nullptr : reinterpret_cast<const unsigned*>(w + l.tilemax));
        VHAP_LAUNCH_CHECK();
        return (int)VHAP_OK;
    });
}
