// Fused per-pixel stages of the photometric energy for gfx950:
//   shade_fwd / shade_bwd : safe-normalise the interpolated normal, 9-band SH diffuse shading,
//                           rgb = albedo * diffuse, alpha = coverage, composite over the background
//                           (constant colour or the target image, flipped to the renderer's y-up
//                           frame), plus the statistics of the diffuse regulariser.
//                           Replaces render_nvdiffrast.py:386 (safe_normalize), :402-421
//                           (shade x2, rgb, alpha, background where) and tracker.py:547-550
//                           (reg_diffuse: relu(max(diffuse) - 1) + mean(var_channels(diffuse))),
//                           ~25 eager kernels and 340 MB of [B,H,W,9,3] temporaries in the reference.
//   photo_fwd / photo_bwd : sum |gt - pred| and #(alpha > 0) (tracker.py:430-439) and its gradient.
// One thread per pixel, 16-byte accesses where the layout allows, block reduction + one atomic per
// block for the scalar / [9,3] outputs.
#include "common.h"
#include "aa_items.h"
#include "energy_common.h"

#ifndef VHAP_PHOTO_PU
#define VHAP_PHOTO_PU 4
#endif
namespace {

// Reducing kernels: PB-thread workgroups, at most MAX_BLOCKS of them.  Every workgroup ends with ONE atomic per output;
// same-address atomics serialise (~12 ns each, more under CAS retries), and all workgroups finish at about the same time,
// so the chain length is what is paid: 2048 x 256 threads cost ~150 us of tail, 512 x 1024 threads ~6 us.
constexpr int PB = 1024;
constexpr int NW = PB / 64;
constexpr int PBB = 512;      // shade_bwd: 27 accumulators + the SH algebra need ~150 VGPRs -> 2 waves per SIMD, no scratch
constexpr int NWB = PBB / 64;
constexpr int MAX_BLOCKS = 512;

struct SH9 {
    float v[9];
};

__device__ __forceinline__ void sh_basis(float x, float y, float z, const float* __restrict__ sc, SH9& b) {
    b.v[0] = sc[0];
    b.v[1] = x * sc[1];
    b.v[2] = y * sc[2];
    b.v[3] = z * sc[3];
    b.v[4] = x * y * sc[4];
    b.v[5] = x * z * sc[5];
    b.v[6] = y * z * sc[6];
    b.v[7] = (x * x - y * y) * sc[7];
    b.v[8] = (3.0f * z * z - 1.0f) * sc[8];
}

__device__ __forceinline__ unsigned f2ord(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}

struct ShadeParams {
    const float* normal_raw;  // [B,H,W,3]
    const float* albedo;      // [B,H,W,3]
    const float4* rast;       // [B,H,W,4]
    const float* bg_image;    // [B,3,H,W] image space (row 0 = top) or nullptr
    float bg_r, bg_g, bg_b;   // constant background when bg_image == nullptr
    const float* lights;      // [9,3]
    const float* sh_const;    // [9]
    int B, H, W;
    const int* fid2cid;       // optional: triangle id + 1 -> colour cluster (0 = background) ...
    int nfid;
    unsigned char* cid;       // ... written per pixel for the disturbance pass (it then never touches `rast`)
};

// stats (4 words): [0..1] = u64 (ordered-uint max of diffuse << 32 | number of entries equal to it), [2] = float sum over
// pixels of var_channels(diffuse).  The tie count makes the backward of max() distribute evenly like torch's.
__global__ __launch_bounds__(PB) __attribute__((amdgpu_waves_per_eu(4, 4))) void shade_fwd_kernel(const ShadeParams P, float4* __restrict__ rgba,
                                                        unsigned* __restrict__ stats) {
    __shared__ float s_l[27], s_c[9];
    __shared__ float red_var[NW];
    __shared__ unsigned long long red_max[NW];
    if (threadIdx.x < 27) s_l[threadIdx.x] = P.lights[threadIdx.x];
    if (threadIdx.x < 9) s_c[threadIdx.x] = P.sh_const[threadIdx.x];
    __syncthreads();
    float var = 0.f;
    float mxv = -INFINITY;          // running max of the diffuse colour channels and the number of entries equal to it
    unsigned mxn = 0u;
    auto merge = [](unsigned long long a, unsigned long long b) {
        const unsigned ha = (unsigned)(a >> 32), hb = (unsigned)(b >> 32);
        return ha > hb ? a : (hb > ha ? b : a + (b & 0xffffffffull));
    };
    const unsigned npix = (unsigned)P.B * P.H * P.W, HW = (unsigned)P.H * P.W;     // < 2^31 (check_img): 32-bit index math only
    // Software-pipelined: the normal and the rasteriser word of the NEXT pixel are requested before the current one is shaded, so the
    // coverage test that decides between the albedo and the background loads never waits for memory, and the SH algebra runs
    // under the latency of those loads (three dependent round trips per pixel otherwise: 100 us instead of the ~55 us the traffic costs).
    const unsigned stride = gridDim.x * PB;
    unsigned pi = blockIdx.x * PB + threadIdx.x;
    float nx = 0.f, ny = 0.f, nz = 0.f, rw = 0.f;
    if (pi < npix) {
        const float* nr = P.normal_raw + 3 * (size_t)pi;
        nx = nr[0]; ny = nr[1]; nz = nr[2];
        rw = P.rast[pi].w;
    }
    while (pi < npix) {
        const unsigned nxt = pi + stride;
        float nnx = 0.f, nny = 0.f, nnz = 0.f, nrw = 0.f;
        if (nxt < npix) {
            const float* nr = P.normal_raw + 3 * (size_t)nxt;
            nnx = nr[0]; nny = nr[1]; nnz = nr[2];
            nrw = P.rast[nxt].w;
        }
        const bool fg = rw > 0.0f;
        float c0, c1, c2;                      // albedo (foreground) or background colour: issued now, consumed after the shading
        if (fg) {
            const float* al = P.albedo + 3 * (size_t)pi;
            c0 = al[0]; c1 = al[1]; c2 = al[2];
        } else if (P.bg_image) {
            const unsigned bI = pi / HW, rem = pi - bI * HW;
            const unsigned y = rem / (unsigned)P.W, x = rem - y * (unsigned)P.W;
            const float* g = P.bg_image + (size_t)bI * 3 * HW + (size_t)(P.H - 1 - y) * P.W + x;
            c0 = g[0]; c1 = g[HW]; c2 = g[2 * HW];
        } else {
            c0 = P.bg_r; c1 = P.bg_g; c2 = P.bg_b;
        }
        if (P.cid) P.cid[pi] = (unsigned char)P.fid2cid[min(max((int)rw, 0), P.nfid - 1)];
        const float inv = 1.0f / sqrtf(fmaxf(nx * nx + ny * ny + nz * nz, 1e-20f));
        SH9 b;
        sh_basis(nx * inv, ny * inv, nz * inv, s_c, b);
        float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; k++) {
            d[0] += b.v[k] * s_l[3 * k]; d[1] += b.v[k] * s_l[3 * k + 1]; d[2] += b.v[k] * s_l[3 * k + 2];
        }
        const float mean = (d[0] + d[1] + d[2]) * (1.0f / 3.0f);
        var += 0.5f * ((d[0] - mean) * (d[0] - mean) + (d[1] - mean) * (d[1] - mean) + (d[2] - mean) * (d[2] - mean));
#pragma unroll
        for (int c = 0; c < 3; c++) {
            if (d[c] > mxv) { mxv = d[c]; mxn = 1u; }
            else if (d[c] == mxv) mxn++;
        }
        rgba[pi] = fg ? make_float4(c0 * d[0], c1 * d[1], c2 * d[2], 1.0f) : make_float4(c0, c1, c2, 0.0f);
        pi = nxt; nx = nnx; ny = nny; nz = nnz; rw = nrw;
    }
    unsigned long long mx = mxn ? (((unsigned long long)f2ord(mxv) << 32) | mxn) : 0ull;   // (ordered max << 32) | tie count
    if (stats) {
        var = vhap_wave_sum(var);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)mx, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(mx >> 32), o, 64);
            mx = merge(mx, ((unsigned long long)hi << 32) | lo);
        }
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (lane == 0) { red_var[wave] = var; red_max[wave] = mx; }
        __syncthreads();
        if (threadIdx.x == 0) {
            float vs = 0.f;
            unsigned long long m = 0ull;
            for (int w = 0; w < NW; w++) { vs += red_var[w]; m = merge(m, red_max[w]); }
            atomicAdd(reinterpret_cast<float*>(stats) + 2, vs);
            unsigned long long* g = reinterpret_cast<unsigned long long*>(stats);
            unsigned long long old = *reinterpret_cast<volatile unsigned long long*>(g);
            while (true) {   // (max, count) monoid: CAS loop, at most one per block
                const unsigned long long want = merge(old, m);
                const unsigned long long seen = atomicCAS(g, old, want);
                if (seen == old) break;
                old = seen;
            }
        }
    }
}

// d_rgba -> d_albedo, d_normal_raw, d_lights (accumulated).  reg: gradient of the diffuse regulariser w.r.t.
// lights only (the reference computes it on shade(normal.detach())):
//   g_var = d_reg / npix  (coefficient of d var / d diffuse_c = diffuse_c - mean)
//   g_max = d_reg if max(diffuse) > 1 else 0, applied where diffuse_c equals the max (stats[0]).
// (one 1024-thread workgroup per CU = 4 waves per SIMD: tell the compiler, or it budgets 64 VGPRs and spills the 27 light
// accumulators to scratch)
__global__ __launch_bounds__(PBB) __attribute__((amdgpu_waves_per_eu(2, 2))) void shade_bwd_kernel(const ShadeParams P, const float4* __restrict__ d_rgba,
                                                        const float* __restrict__ keep, const float* __restrict__ d_reg,
                                                        const unsigned* __restrict__ stats,
                                                        float* __restrict__ d_albedo, float* __restrict__ d_normal_raw,
                                                        float* __restrict__ d_lights) {
    __shared__ float s_l[27], s_c[9];
    __shared__ float red[NWB][27];
    if (threadIdx.x < 27) s_l[threadIdx.x] = P.lights[threadIdx.x];
    if (threadIdx.x < 9) s_c[threadIdx.x] = P.sh_const[threadIdx.x];
    __syncthreads();
    const long long npix = (long long)P.B * P.H * P.W;
    float gl[27];
#pragma unroll
    for (int i = 0; i < 27; i++) gl[i] = 0.f;
    float g_var = 0.f, g_max = 0.f;
    unsigned mx_ord = 0u;
    if (d_reg && stats) {
        const float dr = d_reg[0];
        g_var = dr / (float)npix;
        mx_ord = stats[1];                                  // high word of the packed (max, ties)
        const unsigned ties = stats[0];
        const unsigned u = (mx_ord & 0x80000000u) ? (mx_ord & 0x7fffffffu) : ~mx_ord;
        g_max = __uint_as_float(u) > 1.0f ? dr / (float)max(ties, 1u) : 0.f;   // evenly among ties, like torch.max()
    }
    // software-pipelined like the forward: normal + rasteriser word of the next pixel are in flight while this one is processed
    const size_t bstride = (size_t)gridDim.x * PBB;
    size_t pi = (size_t)blockIdx.x * PBB + threadIdx.x;
    float pnx = 0.f, pny = 0.f, pnz = 0.f, prw = 0.f;
    if (pi < (size_t)npix) {
        const float* nr0 = P.normal_raw + 3 * pi;
        pnx = nr0[0]; pny = nr0[1]; pnz = nr0[2];
        prw = P.rast[pi].w;
    }
    for (; pi < (size_t)npix; pi += bstride) {
        const float rx = pnx, ry = pny, rz = pnz, rwc = prw;
        if (pi + bstride < (size_t)npix) {
            const float* nr1 = P.normal_raw + 3 * (pi + bstride);
            pnx = nr1[0]; pny = nr1[1]; pnz = nr1[2];
            prw = P.rast[pi + bstride].w;
        }
        const float l2 = rx * rx + ry * ry + rz * rz;
        const bool clampd = !(l2 > 1e-20f);
        const float inv = 1.0f / sqrtf(fmaxf(l2, 1e-20f));
        const float x = rx * inv, y = ry * inv, z = rz * inv;
        SH9 b;
        sh_basis(x, y, z, s_c, b);
        float d[3] = {0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 9; k++) {
            d[0] += b.v[k] * s_l[3 * k]; d[1] += b.v[k] * s_l[3 * k + 1]; d[2] += b.v[k] * s_l[3 * k + 2];
        }
        const bool fg = rwc > 0.0f;
        float gd[3] = {0.f, 0.f, 0.f};   // photometric part of d(diffuse): flows to lights AND normal
        float ga[3] = {0.f, 0.f, 0.f};
        if (fg) {
            float4 g = d_rgba[pi];
            if (keep) { const float k = keep[pi]; g.x *= k; g.y *= k; g.z *= k; }     // backward of the colour disturbance, folded in
            const float* al = P.albedo + 3 * pi;
            ga[0] = g.x * d[0]; ga[1] = g.y * d[1]; ga[2] = g.z * d[2];
            gd[0] = g.x * al[0]; gd[1] = g.y * al[1]; gd[2] = g.z * al[2];
        }
        if (d_albedo) { d_albedo[3 * pi] = ga[0]; d_albedo[3 * pi + 1] = ga[1]; d_albedo[3 * pi + 2] = ga[2]; }
        // regulariser part: lights only
        float gr[3] = {0.f, 0.f, 0.f};
        if (g_var != 0.f || g_max != 0.f) {
            const float mean = (d[0] + d[1] + d[2]) * (1.0f / 3.0f);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                gr[c] = g_var * (d[c] - mean);
                if (g_max != 0.f && f2ord(d[c]) == mx_ord) gr[c] += g_max;
            }
        }
#pragma unroll
        for (int k = 0; k < 9; k++) {
            gl[3 * k] += b.v[k] * (gd[0] + gr[0]); gl[3 * k + 1] += b.v[k] * (gd[1] + gr[1]); gl[3 * k + 2] += b.v[k] * (gd[2] + gr[2]);
        }
        if (d_normal_raw) {
            float gnx = 0.f, gny = 0.f, gnz = 0.f;
            if (fg) {
                float gb[9];   // d L / d(basis_k / const_k)
#pragma unroll
                for (int k = 0; k < 9; k++) gb[k] = s_c[k] * (s_l[3 * k] * gd[0] + s_l[3 * k + 1] * gd[1] + s_l[3 * k + 2] * gd[2]);
                gnx = gb[1] + y * gb[4] + z * gb[5] + 2.f * x * gb[7];
                gny = gb[2] + x * gb[4] + z * gb[6] - 2.f * y * gb[7];
                gnz = gb[3] + x * gb[5] + y * gb[6] + 6.f * z * gb[8];
                // n = r / max(|r|, 1e-10):  d r = (d n - n (n . d n)) / |r|   (or d n / 1e-10 when clamped)
                if (!clampd) {
                    const float dot = x * gnx + y * gny + z * gnz;
                    gnx = (gnx - x * dot) * inv; gny = (gny - y * dot) * inv; gnz = (gnz - z * dot) * inv;
                } else {
                    gnx *= inv; gny *= inv; gnz *= inv;
                }
            }
            d_normal_raw[3 * pi] = gnx; d_normal_raw[3 * pi + 1] = gny; d_normal_raw[3 * pi + 2] = gnz;
        }
    }
    if (d_lights) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 27; i++) {
            const float s = vhap_wave_sum(gl[i]);
            if (lane == 0) red[wave][i] = s;
        }
        __syncthreads();
        if (threadIdx.x < 27) {
            float s = 0.f;
            for (int w = 0; w < NWB; w++) s += red[w][threadIdx.x];
            if (s != 0.f) atomicAdd(&d_lights[threadIdx.x], s);
        }
    }
}

// pred rgba [B,H,W,4] (renderer space, row 0 = bottom) vs gt [B,3,H,W] (image space).
// out[0] += sum |gt - pred_rgb|, out[1] += #(alpha > 0) (as float, exact below 2^24 per block partial)
// TOTAL: the workgroup that finishes last (ticket counter in out[2], left at zero again) assembles the step energy from the stage accumulators
// and derives the upstream gradient of the photometric term -- what vhap_energy_finalize + vhap_energy_total do as two launches.
struct PhotoTotal {
    const float *frame_terms, *lmk, *tex_terms, *off_terms;
    const unsigned* shade_stats;
    float w_lmk, w_reg_diffuse, w_photo;
    float *log, *d_sum, *gmax_bound;
    float* part;       // [2 x gridDim.x] per-workgroup partial sums
    // optional: the colour part of the antialias backward for this loss (aa_items.h), UNSCALED, done by this launch -- it needs the final
    // image and the pair list, like the sum itself, and nothing of the sum's result
    const int* aa_work;
    float* d_delta;
    int tex_consume;   // VHAP_CALL_TEX_TERMS_CONSUME: tex_terms[0..1] are cleared once read (the carried texture's finish pass accumulates the next step's)
};
template <bool TOTAL>
__global__ __launch_bounds__(PB) void photo_fwd_kernel(const float4* __restrict__ pred, const float* __restrict__ gt, int B, int H,
                                                        int W, float* __restrict__ out, const PhotoTotal E) {
    __shared__ float rs[NW], rn[NW];
    float s = 0.f, n = 0.f;
    const unsigned npix = (unsigned)B * H * W, HW = (unsigned)H * W;
    const unsigned nblk = gridDim.x, bid = blockIdx.x;
    if constexpr (TOTAL) {
        // the antialias job (aa_items.h): every thread takes its share of the pair list FIRST -- at most one item each for the BASELINE
        // configurations: three dependent loads, then fire-and-forget atomics -- and goes on to the sum.  (As dedicated workgroups of this
        // launch the job cost the sum 14 us: the chip holds exactly the sum's 512 workgroups of 1024 threads at once.)
        if (E.aa_work) {
            const int count = E.aa_work[0];
            for (int i = (int)(bid * PB + threadIdx.x); i < count; i += (int)(nblk * PB)) aa_colour_bwd_item(E.aa_work, i, pred, gt, H, W, E.d_delta);
        }
    }
    // PU pixels per trip (4 shipped; 8 measured the same, profiles/r04_call29_apply_photo_ab.txt), their 4 PU loads issued before the first is consumed (clamped addresses past the end): one pixel per trip
    // is load -> wait -> add with a run-time trip count, eight round trips in series per thread at 16 x 512^2.  The order of the
    // additions into s is the one-pixel-per-trip order.
    constexpr int PU = VHAP_PHOTO_PU;
    const unsigned stride = nblk * PB;
    for (unsigned long long q0 = bid * PB + threadIdx.x; q0 < npix; q0 += (unsigned long long)PU * stride) {
        const unsigned p0 = (unsigned)q0;
        float4 pv[PU];
        float g0[PU], g1[PU], g2[PU];
#pragma unroll
        for (int u = 0; u < PU; u++) {
            const unsigned long long pq = (unsigned long long)p0 + (unsigned long long)u * stride;
            const unsigned pi = pq < npix ? (unsigned)pq : npix - 1;
            const unsigned b = pi / HW, rem = pi - b * HW;
            const unsigned y = rem / (unsigned)W, x = rem - y * (unsigned)W;
            const float* g = gt + (size_t)b * 3 * HW + (size_t)(H - 1 - y) * W + x;
            pv[u] = pred[pi];
            g0[u] = g[0]; g1[u] = g[HW]; g2[u] = g[2 * HW];
        }
#pragma unroll
        for (int u = 0; u < PU; u++) {
            if ((unsigned long long)p0 + (unsigned long long)u * stride < npix) {
                s += fabsf(g0[u] - pv[u].x) + fabsf(g1[u] - pv[u].y) + fabsf(g2[u] - pv[u].z);
                n += pv[u].w > 0.0f ? 1.0f : 0.0f;
            }
        }
    }
    s = vhap_wave_sum(s);
    n = vhap_wave_sum(n);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) { rs[wave] = s; rn[wave] = n; }
    __syncthreads();
    if (wave == 0) {
        int last = 0;
        if (lane == 0) {
            float a = 0.f, c = 0.f;
            for (int w = 0; w < NW; w++) { a += rs[w]; c += rn[w]; }
            if constexpr (TOTAL) {
                // per-workgroup partials + a ticket instead of two contended float atomics per workgroup: the sums come out in a fixed
                // order (bit-reproducible energy) and the kernel's tail is one round of 64-lane loads instead of ~1500 serialised atomics
                E.part[2 * bid] = a;
                E.part[2 * bid + 1] = c;
                __threadfence();
                last = atomicAdd(reinterpret_cast<unsigned*>(out + 2), 1u) == nblk - 1;
            } else {
                atomicAdd(&out[0], a);
                atomicAdd(&out[1], c);
            }
        }
        if constexpr (TOTAL) {
            if (__shfl(last, 0, 64)) {                 // the workgroup that finished last: its first wave assembles the energy
                __threadfence();
                float sum = 0.f, cnt = 0.f;
                for (unsigned i = lane; i < nblk; i += 64) {
                    sum += __hip_atomic_load(&E.part[2 * i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    cnt += __hip_atomic_load(&E.part[2 * i + 1], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                sum = vhap_wave_sum(sum);
                cnt = vhap_wave_sum(cnt);
                if (lane == 0) { out[0] = sum; out[1] = cnt; reinterpret_cast<unsigned*>(out + 2)[0] = 0u; }
                vhap_energy::finalize_total_wave(E.frame_terms, E.lmk, E.tex_terms, E.off_terms, E.shade_stats, E.w_lmk, E.w_reg_diffuse,
                                                 (float)npix, sum, cnt, E.w_photo, 1.0f, E.log, E.d_sum, E.gmax_bound);
                if (E.tex_consume && E.tex_terms && lane < 2) const_cast<float*>(E.tex_terms)[lane] = 0.f;
            }
        }
    }
}

// d_pred.rgb = -sign(gt - pred) * d_sum[0]; d_pred.a = 0
__global__ __launch_bounds__(256) void photo_bwd_kernel(const float4* __restrict__ pred, const float* __restrict__ gt,
                                                        const float* __restrict__ d_sum, int B, int H, int W,
                                                        float4* __restrict__ d_pred, float4* __restrict__ d_pred2) {
    const unsigned npix = (unsigned)B * H * W;      // < 2^31 (check_img)
    const unsigned pi = blockIdx.x * 256u + threadIdx.x;
    if (pi >= npix) return;
    const float gs = d_sum[0];
    const unsigned HW = (unsigned)H * W;
    const unsigned b = pi / HW, rem = pi - b * HW;
    const unsigned y = rem / (unsigned)W, x = rem - y * (unsigned)W;
    const float* g = gt + (size_t)b * 3 * HW + (size_t)(H - 1 - y) * W + x;
    const float4 p = pred[pi];
    auto sg = [](float e) { return e > 0.f ? 1.0f : (e < 0.f ? -1.0f : 0.0f); };
    const float4 d = make_float4(-sg(g[0] - p.x) * gs, -sg(g[HW] - p.y) * gs, -sg(g[2 * HW] - p.z) * gs, 0.0f);
    d_pred[pi] = d;
    if (d_pred2) d_pred2[pi] = d;     // second copy: the pass-through part of the antialias backward, written here instead of copied there
}

int check_img(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0 || (long long)B * H * W >= (1ll << 31)) return VHAP_E_BADDIM;
    return VHAP_OK;
}

}  // namespace

extern "C" int vhap_shade_fwd(const float* normal_raw, const float* albedo, const float* rast, const float* bg_image,
                              const float* bg_color, const float* lights, const float* sh_const, const int32_t* fid2cid, int nfid,
                              int B, int H, int W, float* rgba, float* stats, uint8_t* cid, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!normal_raw || !albedo || !rast || !lights || !sh_const || !rgba) return VHAP_E_NULLPTR;
    if (!bg_image && !bg_color) return VHAP_E_NULLPTR;
    if (cid && (!fid2cid || nfid <= 0)) return VHAP_E_NULLPTR;
    if (int e = check_img(B, H, W)) return e;
    ShadeParams P{normal_raw, albedo, reinterpret_cast<const float4*>(rast), bg_image, 0.f, 0.f, 0.f, lights, sh_const, B, H, W, fid2cid, nfid, cid};
    if (!bg_image) { P.bg_r = bg_color[0]; P.bg_g = bg_color[1]; P.bg_b = bg_color[2]; }
    hipStream_t st = vhap_stream(stream);
    if (stats) VHAP_ZERO_ACC(stats, 16, st);
    const long long npix = (long long)B * H * W;
    shade_fwd_kernel<<<min(vhap_cdiv(npix, PB), MAX_BLOCKS), PB, 0, st>>>(P, reinterpret_cast<float4*>(rgba), reinterpret_cast<unsigned*>(stats));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_shade_bwd(const float* normal_raw, const float* albedo, const float* rast, const float* lights,
                              const float* sh_const, const float* d_rgba, const float* keep, const float* d_reg, const float* stats,
                              int B, int H, int W, float* d_albedo, float* d_normal_raw, float* d_lights, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!normal_raw || !albedo || !rast || !lights || !sh_const || !d_rgba) return VHAP_E_NULLPTR;
    if (int e = check_img(B, H, W)) return e;
    ShadeParams P{normal_raw, albedo, reinterpret_cast<const float4*>(rast), nullptr, 0.f, 0.f, 0.f, lights, sh_const, B, H, W, nullptr, 0, nullptr};
    const long long npix = (long long)B * H * W;
    shade_bwd_kernel<<<min(vhap_cdiv(npix, PBB), MAX_BLOCKS), PBB, 0, vhap_stream(stream)>>>(
        P, reinterpret_cast<const float4*>(d_rgba), keep, d_reg, reinterpret_cast<const unsigned*>(stats), d_albedo, d_normal_raw, d_lights);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_photo_fwd(const float* pred_rgba, const float* gt_nchw, int B, int H, int W, float* out2,
                              int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pred_rgba || !gt_nchw || !out2) return VHAP_E_NULLPTR;
    if (int e = check_img(B, H, W)) return e;
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(out2, 8, st);
    const long long npix = (long long)B * H * W;
    photo_fwd_kernel<false><<<min(vhap_cdiv(npix, PB), MAX_BLOCKS), PB, 0, st>>>(reinterpret_cast<const float4*>(pred_rgba), gt_nchw, B, H, W, out2,
                                                                                   PhotoTotal{});
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_photo_fwd_total(const float* pred_rgba, const float* gt_nchw, int B, int H, int W, float* out3, const float* frame_terms,
                                    const float* lmk_energy, const float* tex_terms, const float* off_terms, const float* shade_stats,
                                    float w_landmark, float w_reg_diffuse, float w_photo, float* log, float* d_sum, float* gmax_bound,
                                    float* work, const int32_t* aa_work, float* d_delta_unscaled, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pred_rgba || !gt_nchw || !out3 || !log || !work) return VHAP_E_NULLPTR;
    if (int e = check_img(B, H, W)) return e;
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(out3, 12, st);
    const long long npix = (long long)B * H * W;
    if ((aa_work == nullptr) != (d_delta_unscaled == nullptr)) return VHAP_E_NULLPTR;
    const PhotoTotal E{frame_terms, lmk_energy, tex_terms, off_terms, reinterpret_cast<const unsigned*>(shade_stats), w_landmark, w_reg_diffuse,
                       w_photo, log, d_sum, gmax_bound, work, aa_work, d_delta_unscaled, (call_flags & VHAP_CALL_TEX_TERMS_CONSUME) ? 1 : 0};
    photo_fwd_kernel<true><<<min(vhap_cdiv(npix, PB), MAX_BLOCKS), PB, 0, st>>>(reinterpret_cast<const float4*>(pred_rgba), gt_nchw, B, H, W,
                                                                                              out3, E);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_photo_bwd(const float* pred_rgba, const float* gt_nchw, const float* d_sum, int B, int H, int W,
                              float* d_pred, float* d_pred_copy, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pred_rgba || !gt_nchw || !d_sum || !d_pred) return VHAP_E_NULLPTR;
    if (int e = check_img(B, H, W)) return e;
    const long long npix = (long long)B * H * W;
    photo_bwd_kernel<<<vhap_cdiv(npix, 256), 256, 0, vhap_stream(stream)>>>(reinterpret_cast<const float4*>(pred_rgba), gt_nchw,
                                                                           d_sum, B, H, W, reinterpret_cast<float4*>(d_pred), reinterpret_cast<float4*>(d_pred_copy));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
