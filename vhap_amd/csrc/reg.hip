// Regularisers on the vertex offsets and on the texture, and the fused Adam update, for gfx950.
//
//   offset_reg : Laplacian smoothness, L1 magnitude and per-region rigidity of the static vertex offsets
//                (vhap/model/tracker.py:552-600, 682-690).  The reference forms L (v0 + o) - L v0 with two dense
//                [V,V] batched matmuls per step; L is linear, so this is L o, one CSR gather per vertex.
//   tex_prep   : albedo = painted base + residual in the channel-last layout the texture sampler wants, fused with the
//                total-variation and masked-residual energies (tracker.py:247-258, 518-541): one pass over the texture
//                instead of ~12 full-size elementwise launches.
//   adam       : one launch for every parameter tensor of the step (torch.optim.Adam semantics, tracker.py:159-211).
#include "common.h"
#include <string.h>
#include <algorithm>
#include <type_traits>

namespace {

constexpr int RB = 256;

__device__ __forceinline__ float block_sum256(float v, float* red) {
    v = vhap_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

struct OffCfg {
    int V, nreg, nbv;               // vertices, rigid regions, number of vertex blocks
    float s_lap, s_abs, s_rigid;    // weight / normaliser of each term
};

// blocks [0, nbv): Laplacian + magnitude over a slice of vertices; blocks [nbv, nbv + nreg): one rigid region each.
// d_terms == nullptr: forward (terms accumulated); else backward (d_off accumulated).
__global__ __launch_bounds__(RB) void offset_reg_kernel(OffCfg c, const float* __restrict__ off, const int* __restrict__ lap_ptr,
                                                        const int* __restrict__ lap_col, const float* __restrict__ lap_val,
                                                        const float* __restrict__ w_lap, const float* __restrict__ w_abs,
                                                        const int* __restrict__ reg_ptr, const int* __restrict__ reg_idx,
                                                        float* __restrict__ terms, const float* __restrict__ d_terms,
                                                        float* __restrict__ d_off, long long frame_stride) {
    __shared__ float red[4];
    const bool bwd = d_terms != nullptr;
    off += (size_t)blockIdx.y * frame_stride;           // (per-frame offsets, `use_dynamic_offset`: one grid row per frame, shared accumulators)
    if (d_off) d_off += (size_t)blockIdx.y * frame_stride;
    if ((int)blockIdx.x < c.nbv) {
        const int v = blockIdx.x * RB + threadIdx.x;
        float e_lap = 0.f, e_abs = 0.f;
        if (v < c.V) {
            const int k0 = lap_ptr[v], k1 = lap_ptr[v + 1];
            float lo[3] = {0.f, 0.f, 0.f};
            for (int k = k0; k < k1; k++) {
                const float a = lap_val[k];
                const float* o = off + 3 * lap_col[k];
                lo[0] += a * o[0]; lo[1] += a * o[1]; lo[2] += a * o[2];
            }
            const float wl = w_lap ? w_lap[v] : 1.0f, wa = w_abs ? w_abs[v] : 1.0f;
            const float o0 = off[3 * v], o1 = off[3 * v + 1], o2 = off[3 * v + 2];
            e_lap = (lo[0] * lo[0] + lo[1] * lo[1] + lo[2] * lo[2]) * wl;
            e_abs = (fabsf(o0) + fabsf(o1) + fabsf(o2)) * wa;
            if (bwd) {
                const float gl = 2.0f * c.s_lap * wl * d_terms[0];
                for (int k = k0; k < k1; k++) {
                    const float a = lap_val[k] * gl;
                    float* d = d_off + 3 * lap_col[k];
                    atomicAdd(&d[0], a * lo[0]); atomicAdd(&d[1], a * lo[1]); atomicAdd(&d[2], a * lo[2]);
                }
                const float ga = c.s_abs * wa * d_terms[1];
                const float sg[3] = {o0 > 0.f ? 1.f : (o0 < 0.f ? -1.f : 0.f), o1 > 0.f ? 1.f : (o1 < 0.f ? -1.f : 0.f),
                                     o2 > 0.f ? 1.f : (o2 < 0.f ? -1.f : 0.f)};
                atomicAdd(&d_off[3 * v], ga * sg[0]); atomicAdd(&d_off[3 * v + 1], ga * sg[1]); atomicAdd(&d_off[3 * v + 2], ga * sg[2]);
            }
        }
        if (!bwd) {
            e_lap = block_sum256(e_lap, red);
            e_abs = block_sum256(e_abs, red);
            if (threadIdx.x == 0) { atomicAdd(&terms[0], e_lap * c.s_lap); atomicAdd(&terms[1], e_abs * c.s_abs); }
        }
        return;
    }
    // rigid region: unbiased variance over the region's vertices, mean over x, y, z
    const int r = blockIdx.x - c.nbv;
    const int i0 = reg_ptr[r], i1 = reg_ptr[r + 1], n = i1 - i0;
    if (n < 2) return;
    float s[3] = {0.f, 0.f, 0.f};
    for (int i = i0 + threadIdx.x; i < i1; i += RB) {
        const float* o = off + 3 * reg_idx[i];
        s[0] += o[0]; s[1] += o[1]; s[2] += o[2];
    }
    float mean[3];
#pragma unroll
    for (int k = 0; k < 3; k++) mean[k] = block_sum256(s[k], red) / (float)n;
    const float inv = 1.0f / (float)(n - 1);
    if (bwd) {
        const float g = 2.0f * inv * c.s_rigid * d_terms[2];
        for (int i = i0 + threadIdx.x; i < i1; i += RB) {
            const int v = reg_idx[i];
#pragma unroll
            for (int k = 0; k < 3; k++) atomicAdd(&d_off[3 * v + k], g * (off[3 * v + k] - mean[k]));
        }
        return;
    }
    float q = 0.f;
    for (int i = i0 + threadIdx.x; i < i1; i += RB) {
        const float* o = off + 3 * reg_idx[i];
#pragma unroll
        for (int k = 0; k < 3; k++) { const float d = o[k] - mean[k]; q += d * d; }
    }
    q = block_sum256(q, red);
    if (threadIdx.x == 0) atomicAdd(&terms[2], q * inv * c.s_rigid);
}

// ---- texture ----
struct TexCfg {
    int T;
    float s_tv, s_res;
};

__device__ __forceinline__ void load_texel(const float* __restrict__ painted, const float* __restrict__ extra, size_t plane, size_t i,
                                           float* a) {
#pragma unroll
    for (int c = 0; c < 3; c++) a[c] = (painted ? painted[c * plane + i] : 0.f) + (extra ? extra[c * plane + i] : 0.f);
}

// albedo (channel-last) + TV / residual energies.  A workgroup owns a (4 x 63)-column x TEX_ROWS-row strip; every lane walks down its
// column keeping the previous row in registers (vertical differences cost one extra halo row per strip), the horizontal neighbour comes
// from the next lane.  Waves OVERLAP by one column: lane 63 of a wave holds the column lane 0 of the next wave owns and only serves as the
// right neighbour of lane 62 -- no halo loads, no divergent branch in the row loop (1.6 % of the columns are read twice).  The loads of row
// r + 1 are issued before row r is processed: round 3's first version waited three times per row (own texels, the lane-63 halo, the mask)
// and ran at 2.6 TB/s.
constexpr int TEX_ROWS = 8;
constexpr int TEX_WCOLS = 63;                    // columns a wave owns
constexpr int TEX_BCOLS = (RB / 64) * TEX_WCOLS; // columns a workgroup owns

struct TexRow {
    float p[3], e[3];
    unsigned m;
};
// ALL: painted, extra and res_mask are all there (the tracker's case) -- no pointer tests between the loads, so that they issue back to back
template <bool ALL>
__device__ __forceinline__ TexRow tex_row_load(const float* __restrict__ painted, const float* __restrict__ extra,
                                               const unsigned char* __restrict__ res_mask, size_t plane, size_t i, bool ok) {
    TexRow r;
    const size_t j = ok ? i : 0;                 // (clamped: the load is issued by every lane, its value dropped where !ok -- a select, no branch)
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float p = (ALL || painted) ? painted[k * plane + j] : 0.f;
        const float e = (ALL || extra) ? extra[k * plane + j] : 0.f;
        r.p[k] = ok ? p : 0.f;
        r.e[k] = ok ? e : 0.f;
    }
    const unsigned m = (ALL || (res_mask && extra)) ? res_mask[j] : 0u;
    r.m = ok ? m : 0u;
    return r;
}

template <bool ALL>
__global__ __launch_bounds__(RB) void tex_prep_fwd_kernel(TexCfg c, const float* __restrict__ painted, const float* __restrict__ extra,
                                                          const unsigned char* __restrict__ res_mask, float* __restrict__ albedo,
                                                          float* __restrict__ terms, float* __restrict__ mip1) {
    __shared__ float red[4];
    const int T = c.T;
    const size_t plane = (size_t)T * T;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int x = blockIdx.x * TEX_BCOLS + wave * TEX_WCOLS + lane, y0 = blockIdx.y * TEX_ROWS;
    const bool in_x = x < T;
    const bool own = in_x && lane < TEX_WCOLS;   // lane 63: the next wave's column, a neighbour only
    float e_tv = 0.f, e_res = 0.f;
    float up[3] = {0.f, 0.f, 0.f}, uprt[3] = {0.f, 0.f, 0.f};
    TexRow cur = tex_row_load<ALL>(painted, extra, res_mask, plane, (size_t)y0 * T + x, in_x && y0 < T);
#pragma unroll
    for (int r = 0; r <= TEX_ROWS; r++) {
        const int y = y0 + r;
        const bool row_ok = y < T;               // (uniform; false only for the halo row below the last strip)
        const size_t i = (size_t)y * T + x;
        TexRow nxt = cur;
        if (r < TEX_ROWS) nxt = tex_row_load<ALL>(painted, extra, res_mask, plane, i + T, in_x && y + 1 < T);
        float a[3];
#pragma unroll
        for (int k = 0; k < 3; k++) a[k] = cur.p[k] + cur.e[k];
        if (r > 0 && own && row_ok) {
#pragma unroll
            for (int k = 0; k < 3; k++) { const float d = up[k] - a[k]; e_tv += d * d; }
        }
        if (r < TEX_ROWS) {
            float rt[3];
#pragma unroll
            for (int k = 0; k < 3; k++) rt[k] = __shfl_down(a[k], 1, 64);
            if (own && row_ok) {
                albedo[3 * i] = a[0]; albedo[3 * i + 1] = a[1]; albedo[3 * i + 2] = a[2];
                if (x + 1 < T) {
#pragma unroll
                    for (int k = 0; k < 3; k++) { const float d = a[k] - rt[k]; e_tv += d * d; }
                }
                if (cur.m) e_res += cur.e[0] * cur.e[0] + cur.e[1] * cur.e[1] + cur.e[2] * cur.e[2];
                // first level of the pyramid, fused (T even; strips start on even rows): the 2x2 box ((a00 + a01) + (a10 + a11)) / 4 of
                // vhap_texture_mip_build, same op order -> same bits, without re-reading the 50 MB the kernel has just written
                if (mip1 && (r & 1) && !(x & 1) && x + 1 < T) {
                    float* m = mip1 + 3 * ((size_t)(y >> 1) * (T >> 1) + (x >> 1));
#pragma unroll
                    for (int k = 0; k < 3; k++) m[k] = ((up[k] + uprt[k]) + (a[k] + rt[k])) * 0.25f;
                }
            }
#pragma unroll
            for (int k = 0; k < 3; k++) uprt[k] = rt[k];
        }
#pragma unroll
        for (int k = 0; k < 3; k++) up[k] = a[k];
        cur = nxt;
    }
    e_tv = block_sum256(e_tv, red);
    e_res = block_sum256(e_res, red);
    if (threadIdx.x == 0) {
        if (e_tv != 0.f && c.s_tv != 0.f) atomicAdd(&terms[0], e_tv * c.s_tv);
        if (e_res != 0.f && c.s_res != 0.f) atomicAdd(&terms[1], e_res * c.s_res);
    }
}

// d_extra[c,y,x] = d_albedo[y,x,c] + TV stencil on the saved albedo + masked residual
// ADAM: the torch.optim.Adam update of `extra` (adam_kernel's arithmetic) applied to the freshly computed gradient in the same pass --
// the gradient is still written (it is the parameter's .grad), but never read back, and the separate 350 MB update pass disappears
struct TexAdam {
    float* p;
    float* m;
    float* v;
    const float* lr;
    const int* step;
    float beta1, beta2, eps;
};
// A workgroup owns a 256-column x TEXB_ROWS-row strip like the forward: every lane walks down its column with a three-row window of the
// saved albedo in registers (one load per texel + two halo rows per strip), the horizontal neighbours come from the adjacent lanes
// (one halo load per wave side).  The mip part of the texture gradient is GATHERED: level l contributes 4^-l of its texel
// d_mips[l][y >> l][x >> l] -- with n_gather = every level this replaces the whole fold cascade (vhap_texture_mip_fold).
// (A software-pipelined variant -- 16 rows unrolled, the record of row r + 1 requested before row r is computed, waves overlapping by two
// columns instead of halo loads -- runs this pass in 95 instead of 106 us ALONE and makes the step 28 us SLOWER: the pass is the open tail
// that overlaps the next step's latency-bound geometry head, and a deeper memory queue starves that head.  With the side streams at the
// lowest priority the two variants tie.  profiles/r03_call15_finish_kernel_x_priority.txt)
// (Round 2's version was one texel per thread on a 1-D grid: a 64-bit division per texel, five scattered taps of the albedo, 2.7 TB/s
// against the 5.9 TB/s of the Adam pass behind it -- 25 + 63 + 60 us on the tail of the step for fold + this + Adam.)
constexpr int TEXB_ROWS = 16;
constexpr int TEXB_MAXG = 12;         // mip levels gathered per texel (T <= 4096 whole; larger textures fold their coarsest levels first)

// CARRY (round 6): the texture is CARRIED from step to step.  The pass holds every texel it has just updated in registers, so it also
// writes what the NEXT step's rasteriser samples -- the assembled albedo (painted + updated residual, the arithmetic of tex_prep_fwd)
// IN PLACE over the one this step sampled, and level 1 of the pyramid -- and sums the TV / residual ENERGIES of that next texture
// (terms[0..1], accumulated: the consumer -- the photometric sum's energy assembly -- reads and clears them, VHAP_CALL_TEX_TERMS_CONSUME;
// the TV pairs that straddle two ownership tiles are added by tex_carry_border_kernel from the halo copies below).  tex_prep_fwd (49 us,
// 134 MB read + 64 MB written at T = 2048) then leaves the head of the step, where it ran beside the binning launch and the rasteriser.
// In place means a workgroup's neighbours may already have rewritten the texels its TV stencil reads across the strip's borders (row
// y0 - 1, row y0 + 16, the columns left / right of a wave).  Those come from HALO copies instead: the first / last row of every 16-row
// strip and the first / last column of every 64-column wave segment, kept twice -- a pass reads the copy of parity (Adam step & 1), which
// the previous pass wrote, and writes the new border values into the other one (12.6 + 3.1 MB at T = 2048 for both parities).
// vhap_tex_carry_prime assembles the albedo, level 1 and both parities of the halos from scratch (first step of a loop, or whenever
// anything but this pass has touched the texture).  Needs T % 64 == 0.
struct TexCarry {
    const float* painted;      // [3,T,T]
    float* mip1;               // levels 1 and 2 of the pyramid ([T/2,T/2,3] then [T/4,T/4,3]: the head of the vhap_texture_mip_build buffer)
    float* row_halo;           // [2][T/16][2][T][3]
    float* col_halo;           // [2][T][T/64][2][3]
    float* terms;              // [2]: TV / residual energies of the texture this pass hands on, accumulated
    int write_grad;            // 0: d_extra is not written (the update is applied here; nothing reads the gradient)
};
__host__ __device__ inline size_t tex_carry_row_halo_floats(int T) { return (size_t)(T / TEXB_ROWS) * 2 * T * 3; }   // per parity
__host__ __device__ inline size_t tex_carry_col_halo_floats(int T) { return (size_t)T * (T / 64) * 2 * 3; }          // per parity

// (at most 80 VGPRs -- 6 waves per SIMD: the pass is the step's open tail, all of its 1 024 workgroups are resident from start to end (4 waves
// per SIMD at T = 2048), and whatever the main chain launches meanwhile must fit beside them.  At 82 VGPRs the carried form left 160 of a
// SIMD's 512 registers free and the per-frame backward -- 170 VGPRs per wave -- waited for the pass to END: 20 -> 84 us on the main chain,
// profiles/r06_call3_trace_stats_*.txt)
template <bool ADAM, bool CARRY = false>
__global__ __launch_bounds__(RB, 6) void tex_prep_bwd_kernel(TexCfg c, std::conditional_t<CARRY, float*, const float*> __restrict__ albedo,
                                                          const float* __restrict__ extra,
                                                          const unsigned char* __restrict__ res_mask, const float* __restrict__ d_albedo,
                                                          const float* __restrict__ d_mips, int n_gather, const float* __restrict__ d_terms,
                                                          float* __restrict__ d_extra, const TexAdam A, int step_add, int y_base,
                                                          float* __restrict__ d_base, const TexCarry C) {
#pragma clang fp contract(off)
    // no fp contraction in this kernel: its instantiations (whole texture / row strip / carried) must produce the SAME bits from the same
    // inputs -- the carried form is tested bit for bit against the plain one -- and hipcc fuses a * b + c per instantiation, as the
    // surrounding code happens to suggest (the carried form's gradient differed from the plain one's by an ulp from its second step on)
    static_assert(!CARRY || ADAM, "the carried texture is written by the pass that updates it");
    __shared__ float red[4];
    const int T = c.T;
    const size_t plane = (size_t)T * T;
    const float gtv = 2.0f * c.s_tv * d_terms[0], gres = 2.0f * c.s_res * d_terms[1];
    float bc2s = 1.f, step_size = 0.f;
    int parity = 0;
    if constexpr (ADAM) {
        const int sti = A.step[0] + step_add;
        const float st = (float)sti;
        const float bc1 = 1.0f - powf(A.beta1, st);
        bc2s = sqrtf(1.0f - powf(A.beta2, st));
        step_size = A.lr[0] / bc1;
        parity = sti & 1;
    }
    const int x = blockIdx.x * RB + threadIdx.x, y0 = y_base + blockIdx.y * TEXB_ROWS;     // (y_base: the first row of a ROW STRIP of the texture)
    const int lane = threadIdx.x & 63;
    const bool in_x = x < T;
    const bool tv = gtv != 0.f;
    auto load_row = [&](int y, int xx, float* a) {
        if (xx >= 0 && xx < T && y >= 0 && y < T) {
            const float* q = albedo + 3 * ((size_t)y * T + xx);
            a[0] = q[0]; a[1] = q[1]; a[2] = q[2];
        } else {
            a[0] = a[1] = a[2] = 0.f;
        }
    };
    // CARRY: the borders of the OLD texture (this step's), from the halo copies of this step's parity; the new ones go to the other parity
    const int strip = y0 / TEXB_ROWS, nstrips = T / TEXB_ROWS, wseg = __builtin_amdgcn_readfirstlane(x >> 6), nwseg = T >> 6;   // (wseg: wave-uniform)
    const float* __restrict__ const rh_old = CARRY ? C.row_halo + (size_t)parity * tex_carry_row_halo_floats(T) : nullptr;
    const float* __restrict__ const ch_old = CARRY ? C.col_halo + (size_t)parity * tex_carry_col_halo_floats(T) : nullptr;
    float* __restrict__ const rh_new = CARRY ? C.row_halo + (size_t)(parity ^ 1) * tex_carry_row_halo_floats(T) : nullptr;
    float* __restrict__ const ch_new = CARRY ? C.col_halo + (size_t)(parity ^ 1) * tex_carry_col_halo_floats(T) : nullptr;
    const float* __restrict__ const painted = CARRY ? C.painted : nullptr;
    float* __restrict__ const mip1 = CARRY ? C.mip1 : nullptr;
    auto halo_row = [&](int s, int slot, float* a) {          // first (slot 0) / last (slot 1) row of strip s
        if (in_x && s >= 0 && s < nstrips) {
            const float* q = rh_old + 3 * (((size_t)s * 2 + slot) * T + x);
            a[0] = q[0]; a[1] = q[1]; a[2] = q[2];
        } else {
            a[0] = a[1] = a[2] = 0.f;
        }
    };
    auto halo_col = [&](int y, int w, int slot, float* a) {   // first (slot 0) / last (slot 1) column of wave segment w
        if (w >= 0 && w < nwseg) {
            const float* q = ch_old + 3 * (((size_t)y * nwseg + w) * 2 + slot);
            a[0] = q[0]; a[1] = q[1]; a[2] = q[2];
        } else {
            a[0] = a[1] = a[2] = 0.f;
        }
    };
    float up[3] = {0.f, 0.f, 0.f}, cur[3] = {0.f, 0.f, 0.f}, dn[3] = {0.f, 0.f, 0.f};
    if (tv) {
        if constexpr (CARRY) halo_row(strip - 1, 1, up);
        else load_row(y0 - 1, x, up);
        load_row(y0, x, cur);
    }
    float e_tv = 0.f, e_res = 0.f;                            // CARRY: TV / residual energies of the NEXT texture (tex_prep_fwd's sums)
    float nprev[3] = {0.f, 0.f, 0.f}, m1up[3] = {0.f, 0.f, 0.f};
    // (restrict-qualified copies: stores through members of a by-value struct are otherwise assumed to alias every later load, which
    // pins each row's loads behind the previous row's stores)
    float* __restrict__ const adam_p = A.p;
    float* __restrict__ const adam_m = A.m;
    float* __restrict__ const adam_v = A.v;
    // the mip levels to gather: (offset, shift, weight) per slot, uniform.  ALWAYS TEXB_MAXG slots -- the ones beyond n_gather re-read the
    // last level with weight 0 (g + 0 * m == g exactly) -- so that the loads of a texel are one unconditional, independent batch (a
    // run-time trip count made every level wait for its own load: eleven memory latencies per texel)
    size_t g_off[TEXB_MAXG];
    int g_sh[TEXB_MAXG];
    float g_sc[TEXB_MAXG];
    {
        size_t off = 0;
        float sc = 0.25f;
#pragma unroll
        for (int u = 0; u < TEXB_MAXG; u++) {
            const int l = u + 1;
            const bool on = l <= n_gather;
            g_off[u] = on ? off : (u > 0 ? g_off[u - 1] : 0);
            g_sh[u] = on ? l : (u > 0 ? g_sh[u - 1] : 0);
            g_sc[u] = on ? sc : 0.f;
            if (on) off += (size_t)(T >> l) * (T >> l) * 3;
            sc *= 0.25f;
        }
    }
    // levels 4 and up cover 16 x 16 texels and more per texel of theirs: over the 16 rows of this strip (y0 is a multiple of 16) a lane
    // reads the SAME texel of each of them -- fetched and summed once per strip, not once per row (8 of the 12 gathers of a texel)
    static_assert(TEXB_ROWS == 16 && TEXB_MAXG > 3, "hoisting of the coarse levels assumes 16-row strips");
    float coarse[3] = {0.f, 0.f, 0.f};
    if (d_mips && in_x) {
        float m[TEXB_MAXG - 3][3];
#pragma unroll
        for (int u = 3; u < TEXB_MAXG; u++) {
            const float* q = d_mips + g_off[u] + 3 * ((size_t)(y0 >> g_sh[u]) * (T >> g_sh[u]) + (x >> g_sh[u]));
            m[u - 3][0] = q[0]; m[u - 3][1] = q[1]; m[u - 3][2] = q[2];
        }
#pragma unroll
        for (int u = 3; u < TEXB_MAXG; u++) {          // level order
            coarse[0] += g_sc[u] * m[u - 3][0]; coarse[1] += g_sc[u] * m[u - 3][1]; coarse[2] += g_sc[u] * m[u - 3][2];
        }
    }
    const int nrows = min(TEXB_ROWS, T - y0);
    for (int r = 0; r < nrows; r++) {
        const int y = y0 + r;
        const size_t i = (size_t)y * T + x;
        if (tv) {
            if (CARRY && r + 1 == TEXB_ROWS) halo_row(strip + 1, 0, dn);
            else load_row(y + 1, x, dn);
        }
        float pt[3] = {0.f, 0.f, 0.f};
        if constexpr (CARRY) {
            if (in_x) { pt[0] = painted[i]; pt[1] = painted[plane + i]; pt[2] = painted[2 * plane + i]; }
        }
        float g[3] = {0.f, 0.f, 0.f};
        if (in_x) {
            if (d_albedo) { g[0] = d_albedo[3 * i]; g[1] = d_albedo[3 * i + 1]; g[2] = d_albedo[3 * i + 2]; }
            if (d_mips) {
                float m[3][3];
#pragma unroll
                for (int u = 0; u < 3; u++) {
                    const float* q = d_mips + g_off[u] + 3 * ((size_t)(y >> g_sh[u]) * (T >> g_sh[u]) + (x >> g_sh[u]));
                    m[u][0] = q[0]; m[u][1] = q[1]; m[u][2] = q[2];
                }
#pragma unroll
                for (int u = 0; u < 3; u++) { g[0] += g_sc[u] * m[u][0]; g[1] += g_sc[u] * m[u][1]; g[2] += g_sc[u] * m[u][2]; }    // levels 1-3
                g[0] += coarse[0]; g[1] += coarse[1]; g[2] += coarse[2];
            }
        }
        if (tv) {
            float lf[3], rt[3];
#pragma unroll
            for (int k = 0; k < 3; k++) { lf[k] = __shfl_up(cur[k], 1, 64); rt[k] = __shfl_down(cur[k], 1, 64); }
            if constexpr (CARRY) {
                if (lane == 0) halo_col(y, wseg - 1, 1, lf);
                if (lane == 63) halo_col(y, wseg + 1, 0, rt);
            } else {
                if (lane == 0) load_row(y, x - 1, lf);
                if (lane == 63) load_row(y, x + 1, rt);
            }
            if (in_x) {
                float acc[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    if (y + 1 < T) acc[k] += cur[k] - dn[k];
                    if (y > 0) acc[k] += cur[k] - up[k];
                    if (x + 1 < T) acc[k] += cur[k] - rt[k];
                    if (x > 0) acc[k] += cur[k] - lf[k];
                    g[k] += gtv * acc[k];
                }
            }
        }
        if (in_x) {
            // d(base texture) = d(albedo) without the residual's own regulariser: what the PCA texture model's backward takes (vhap_tex_pca_bwd)
            if (d_base) { d_base[i] = g[0]; d_base[plane + i] = g[1]; d_base[2 * plane + i] = g[2]; }
            float ex[3] = {0.f, 0.f, 0.f};
            const bool res = gres != 0.f && res_mask && res_mask[i];
            if (res || ADAM) {
#pragma unroll
                for (int k = 0; k < 3; k++) ex[k] = ADAM ? adam_p[k * plane + i] : extra[k * plane + i];    // (ADAM: `extra` IS the parameter being updated)
            }
            if (res) {
#pragma unroll
                for (int k = 0; k < 3; k++) g[k] += gres * ex[k];
            }
            if (!CARRY || C.write_grad) { d_extra[i] = g[0]; d_extra[plane + i] = g[1]; d_extra[2 * plane + i] = g[2]; }
            float pn[3] = {0.f, 0.f, 0.f};
            if constexpr (ADAM) {
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    const size_t j = k * plane + i;
                    const float gi = g[k];
                    const float m0 = adam_m[j];
                    const float mi = m0 + (gi - m0) * (1.0f - A.beta1);              // lerp, like torch
                    const float vi = A.beta2 * adam_v[j] + (1.0f - A.beta2) * gi * gi;
                    adam_m[j] = mi;
                    adam_v[j] = vi;
                    pn[k] = ex[k] - step_size * mi / (sqrtf(vi) / bc2s + A.eps);
                    adam_p[j] = pn[k];
                }
            }
            if constexpr (CARRY) {
                // the next step's texture: albedo = painted + residual (tex_prep_fwd's sum), in place; borders into the other parity's halos;
                // level 1 of the pyramid = ((a00 + a01) + (a10 + a11)) / 4 (vhap_texture_mip_build's order) on odd rows by even lanes
                float nv[3], nrt[3];
#pragma unroll
                for (int k = 0; k < 3; k++) { nv[k] = pt[k] + pn[k]; nrt[k] = __shfl_down(nv[k], 1, 64); }
                albedo[3 * i] = nv[0]; albedo[3 * i + 1] = nv[1]; albedo[3 * i + 2] = nv[2];
                // its energies, the pairs inside this wave's 64-column x 16-row tile (the others: tex_carry_border_kernel)
                if (tv) {
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        const float dh = nv[k] - nrt[k], dv = nprev[k] - nv[k];
                        if (lane < 63) e_tv += dh * dh;
                        if (r > 0) e_tv += dv * dv;
                    }
                }
                if (res) e_res += pn[0] * pn[0] + pn[1] * pn[1] + pn[2] * pn[2];
                if (r == 0 || r + 1 == TEXB_ROWS) {
                    float* q = rh_new + 3 * (((size_t)strip * 2 + (r == 0 ? 0 : 1)) * T + x);
                    q[0] = nv[0]; q[1] = nv[1]; q[2] = nv[2];
                }
                if (lane == 0 || lane == 63) {
                    float* q = ch_new + 3 * (((size_t)y * nwseg + wseg) * 2 + (lane == 0 ? 0 : 1));
                    q[0] = nv[0]; q[1] = nv[1]; q[2] = nv[2];
                }
                if (r & 1) {                                  // (uniform; the upper row's right neighbours are fetched again rather than kept: registers)
                    float prt[3], m1[3];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        prt[k] = __shfl_down(nprev[k], 1, 64);
                        m1[k] = ((nprev[k] + prt[k]) + (nv[k] + nrt[k])) * 0.25f;      // (meaningful on even lanes)
                    }
                    if (!(lane & 1)) {
                        float* q = mip1 + 3 * ((size_t)(y >> 1) * (T >> 1) + (x >> 1));
                        q[0] = m1[0]; q[1] = m1[1]; q[2] = m1[2];
                    }
                    // level 2 from the level-1 texels of two consecutive odd rows (lanes 4j hold the even level-1 column, lanes 4j + 2 the
                    // odd one: DPP quad_perm [2,3,0,1], issued by every lane), same ((a + b) + (c + d)) / 4 as vhap_texture_mip_build:
                    // the pyramid build of the next step then starts from 3 MB instead of 12.6
                    if ((r & 3) == 1) {
                        m1up[0] = m1[0]; m1up[1] = m1[1]; m1up[2] = m1[2];
                    } else {
                        float m2[3];
#pragma unroll
                        for (int k = 0; k < 3; k++) {
                            const float up_hi = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m1up[k]), 0x4E, 0xF, 0xF, true));
                            const float dn_hi = __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(m1[k]), 0x4E, 0xF, 0xF, true));
                            m2[k] = ((m1up[k] + up_hi) + (m1[k] + dn_hi)) * 0.25f;
                        }
                        if (!(lane & 3)) {
                            float* q = mip1 + 3 * ((size_t)(T >> 1) * (T >> 1)) + 3 * ((size_t)(y >> 2) * (T >> 2) + (x >> 2));
                            q[0] = m2[0]; q[1] = m2[1]; q[2] = m2[2];
                        }
                    }
                }
#pragma unroll
                for (int k = 0; k < 3; k++) nprev[k] = nv[k];
            }
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { up[k] = cur[k]; cur[k] = dn[k]; }
    }
    if constexpr (CARRY) {
        e_tv = block_sum256(e_tv, red);                       // weighted like tex_prep_fwd's terms
        e_res = block_sum256(e_res, red);
        if (threadIdx.x == 0) {
            if (e_tv != 0.f && c.s_tv != 0.f) atomicAdd(&C.terms[0], e_tv * c.s_tv);
            if (e_res != 0.f && c.s_res != 0.f) atomicAdd(&C.terms[1], e_res * c.s_res);
        }
    }
}

// TV pairs of the carried texture that straddle two ownership tiles of the finish pass, from the halo copies that pass has just written
// (the parity it wrote: the OTHER one of its step): last row of strip s | first row of strip s + 1, last column of wave segment w | first
// column of segment w + 1.
__global__ __launch_bounds__(RB) void tex_carry_border_kernel(int T, float s_tv, const int* __restrict__ step, int step_add,
                                                              const float* __restrict__ row_halo, const float* __restrict__ col_halo,
                                                              float* __restrict__ terms) {
    __shared__ float red[4];
    const int par_new = step ? (((step[0] + step_add) & 1) ^ 1) : 0;          // (step == NULL: after vhap_tex_carry_prime both copies are the same)
    const float* __restrict__ rh = row_halo + (size_t)par_new * tex_carry_row_halo_floats(T);
    const float* __restrict__ ch = col_halo + (size_t)par_new * tex_carry_col_halo_floats(T);
    const int nstrips = T / TEXB_ROWS, nw = T >> 6;
    const size_t nrow = (size_t)(nstrips - 1) * T, ncol = (size_t)T * (nw - 1);
    float e = 0.f;
    for (size_t i = (size_t)blockIdx.x * RB + threadIdx.x; i < nrow + ncol; i += (size_t)gridDim.x * RB) {      // (<= 128 workgroups: one atomic each)
        const float *a, *b;
        if (i < nrow) {
            const size_t s = i / T, x = i % T;
            a = rh + 3 * ((s * 2 + 1) * T + x);
            b = rh + 3 * (((s + 1) * 2) * T + x);
        } else {
            const size_t j = i - nrow, y = j / (nw - 1), w = j % (nw - 1);
            a = ch + 3 * ((y * nw + w) * 2 + 1);
            b = ch + 3 * ((y * nw + w + 1) * 2);
        }
#pragma unroll
        for (int k = 0; k < 3; k++) { const float d = a[k] - b[k]; e += d * d; }
    }
    e = block_sum256(e, red);
    if (threadIdx.x == 0 && e != 0.f) atomicAdd(&terms[0], e * s_tv);
}

// both parities of the carried texture's halo copies from the assembled albedo (vhap_tex_carry_prime)
__global__ __launch_bounds__(RB) void tex_carry_halo_init_kernel(int T, const float* __restrict__ albedo, float* __restrict__ row_halo,
                                                                 float* __restrict__ col_halo) {
    const size_t nrow = tex_carry_row_halo_floats(T) / 3, ncol = tex_carry_col_halo_floats(T) / 3;     // texels per parity
    const size_t i = (size_t)blockIdx.x * RB + threadIdx.x;
    if (i < nrow) {
        const int x = (int)(i % T), slot = (int)((i / T) & 1), s = (int)(i / T / 2);
        const float* q = albedo + 3 * ((size_t)(s * TEXB_ROWS + (slot ? TEXB_ROWS - 1 : 0)) * T + x);
#pragma unroll
        for (int k = 0; k < 3; k++) { row_halo[3 * i + k] = q[k]; row_halo[3 * (nrow + i) + k] = q[k]; }
    } else if (i < nrow + ncol) {
        const size_t j = i - nrow;
        const int nw = T >> 6, slot = (int)(j & 1), w = (int)((j >> 1) % nw), y = (int)((j >> 1) / nw);
        const float* q = albedo + 3 * ((size_t)y * T + w * 64 + (slot ? 63 : 0));
#pragma unroll
        for (int k = 0; k < 3; k++) { col_halo[3 * j + k] = q[k]; col_halo[3 * (ncol + j) + k] = q[k]; }
    }
}

// ---- per-frame vertex offsets (use_dynamic_offset) ----
// reg_offset_dynamic (tracker.py:594-600): temporal smoothness of the dynamic offset; the previous timestep's row is NOT detached
__global__ __launch_bounds__(RB) void offset_dynamic_reg_kernel(const float* __restrict__ dyn, const long long* __restrict__ ts, int B, int N, int V,
                                                                float scale, const float* __restrict__ d_term, float* __restrict__ energy,
                                                                float* __restrict__ d_dyn) {
    __shared__ float red[4];
    const int n = V * 3;
    const int i = blockIdx.x * RB + threadIdx.x, b = blockIdx.y;
    const long long t = ts[b], p = t > 0 ? t - 1 : 0;
    float e = 0.f;
    if (i < n && t >= 0 && t < N) {
        const float d = dyn[(size_t)t * n + i] - dyn[(size_t)p * n + i];
        e = d * d;
        if (d_dyn && t != p) {
            const float g = 2.0f * scale * (d_term ? d_term[0] : 1.0f) * d;
            atomicAdd(&d_dyn[(size_t)t * n + i], g);
            atomicAdd(&d_dyn[(size_t)p * n + i], -g);
        }
    }
    if (energy) {
        e = block_sum256(e, red);
        if (threadIdx.x == 0 && e != 0.f) atomicAdd(energy, e * scale);
    }
}

// d_static[v] += sum_b (g_a + g_b)[b][v] ; d_dyn[ts[b]][v] += (g_a + g_b)[b][v]   (frames of one timestep -- multi-view -- add up: atomics)
__global__ __launch_bounds__(RB) void offset_grad_finish_kernel(const float* __restrict__ g_a, const float* __restrict__ g_b,
                                                                const long long* __restrict__ ts, int B, int N, int V,
                                                                float* __restrict__ d_static, float* __restrict__ d_dyn) {
    const int n = V * 3;
    const int i = blockIdx.x * RB + threadIdx.x;
    if (i >= n) return;
    float s = 0.f;
    for (int b = 0; b < B; b++) {
        float g = g_a[(size_t)b * n + i];
        if (g_b) g += g_b[(size_t)b * n + i];
        s += g;
        const long long t = ts[b];
        if (d_dyn && t >= 0 && t < N && g != 0.f) atomicAdd(&d_dyn[(size_t)t * n + i], g);
    }
    if (d_static) d_static[i] += s;
}

// ---- Adam ----
struct AdamTable {
    float* p[VHAP_ADAM_MAX_TENSORS];
    const float* g[VHAP_ADAM_MAX_TENSORS];
    float* m[VHAP_ADAM_MAX_TENSORS];
    float* v[VHAP_ADAM_MAX_TENSORS];
    long long n[VHAP_ADAM_MAX_TENSORS];
    int lr_index[VHAP_ADAM_MAX_TENSORS];
    int blk_start[VHAP_ADAM_MAX_TENSORS + 1];   // 1-D grid: tensor k owns blocks [blk_start[k], blk_start[k + 1])
    int n_tensors;
};

__global__ __launch_bounds__(RB) void adam_kernel(AdamTable t, const float* __restrict__ lr, const int* __restrict__ step, int step_add,
                                                  float beta1, float beta2, float eps) {
    int k = 0;
    while (k + 1 < t.n_tensors && (int)blockIdx.x >= t.blk_start[k + 1]) k++;
    const long long n = t.n[k];
    const long long blk = (int)blockIdx.x - t.blk_start[k], nblk = t.blk_start[k + 1] - t.blk_start[k];
    const float st = (float)(step[0] + step_add);
    const float bc1 = 1.0f - powf(beta1, st), bc2s = sqrtf(1.0f - powf(beta2, st));
    const float step_size = lr[t.lr_index[k]] / bc1;
    float* __restrict__ p = t.p[k];
    const float* __restrict__ g = t.g[k];
    float* __restrict__ m = t.m[k];
    float* __restrict__ v = t.v[k];
    for (long long i = blk * RB + threadIdx.x; i < n; i += nblk * RB) {
        const float gi = g[i];
        const float mi = m[i] + (gi - m[i]) * (1.0f - beta1);          // lerp, like torch
        const float vi = beta2 * v[i] + (1.0f - beta2) * gi * gi;
        m[i] = mi;
        v[i] = vi;
        p[i] -= step_size * mi / (sqrtf(vi) / bc2s + eps);
    }
}

__global__ void adam_bump_kernel(int* step) { step[0] += 1; }

}  // namespace

extern "C" int vhap_offset_reg_fwd(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col, const float* lap_val,
                                   const float* w_lap, const float* w_abs, const int32_t* region_ptr, const int32_t* region_idx,
                                   int V, int n_regions, float s_lap, float s_abs, float s_rigid, float* terms, int call_flags,
                                   vhap_stream_t stream) {
    VHAP_ENTER();
    if (!offset || !lap_ptr || !lap_col || !lap_val || !terms) return VHAP_E_NULLPTR;
    if (n_regions > 0 && (!region_ptr || !region_idx)) return VHAP_E_NULLPTR;
    if (V <= 0 || n_regions < 0) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(terms, 3 * sizeof(float), st);
    OffCfg c{V, n_regions, vhap_cdiv(V, RB), s_lap, s_abs, s_rigid};
    offset_reg_kernel<<<c.nbv + n_regions, RB, 0, st>>>(c, offset, lap_ptr, lap_col, lap_val, w_lap, w_abs, region_ptr, region_idx, terms, nullptr,
                                                      nullptr, 0);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// The same terms for B per-frame offsets [B,V,3] in ONE launch (`use_dynamic_offset`: static_offset + dynamic_offset[timesteps], tracker.py:552-600;
// the scales carry the 1 / B of the means over the frames): `terms` accumulates over the frames, d_offset is [B,V,3].
extern "C" int vhap_offset_reg_fwd_batch(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col, const float* lap_val,
                                         const float* w_lap, const float* w_abs, const int32_t* region_ptr, const int32_t* region_idx, int B,
                                         int V, int n_regions, float s_lap, float s_abs, float s_rigid, float* terms, int call_flags,
                                         vhap_stream_t stream) {
    VHAP_ENTER();
    if (!offset || !lap_ptr || !lap_col || !lap_val || !terms) return VHAP_E_NULLPTR;
    if (n_regions > 0 && (!region_ptr || !region_idx)) return VHAP_E_NULLPTR;
    if (B <= 0 || B > 65535 || V <= 0 || n_regions < 0) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(terms, 3 * sizeof(float), st);
    OffCfg c{V, n_regions, vhap_cdiv(V, RB), s_lap, s_abs, s_rigid};
    offset_reg_kernel<<<dim3(c.nbv + n_regions, B), RB, 0, st>>>(c, offset, lap_ptr, lap_col, lap_val, w_lap, w_abs, region_ptr, region_idx, terms,
                                                                nullptr, nullptr, 3ll * V);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_offset_reg_bwd_batch(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col, const float* lap_val,
                                         const float* w_lap, const float* w_abs, const int32_t* region_ptr, const int32_t* region_idx, int B,
                                         int V, int n_regions, float s_lap, float s_abs, float s_rigid, const float* d_terms, float* d_offset,
                                         vhap_stream_t stream) {
    VHAP_ENTER();
    if (!offset || !lap_ptr || !lap_col || !lap_val || !d_terms || !d_offset) return VHAP_E_NULLPTR;
    if (n_regions > 0 && (!region_ptr || !region_idx)) return VHAP_E_NULLPTR;
    if (B <= 0 || B > 65535 || V <= 0 || n_regions < 0) return VHAP_E_BADDIM;
    OffCfg c{V, n_regions, vhap_cdiv(V, RB), s_lap, s_abs, s_rigid};
    offset_reg_kernel<<<dim3(c.nbv + n_regions, B), RB, 0, vhap_stream(stream)>>>(c, offset, lap_ptr, lap_col, lap_val, w_lap, w_abs, region_ptr,
                                                                                 region_idx, nullptr, d_terms, d_offset, 3ll * V);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// off_b [B,V,3] = (static_offset [V,3] or 0) + dynamic_offset[timesteps[b]] (tracker.py:555-559): the per-frame offset rows of a step in one
// launch of the library's own (a host framework's index_select + add would be two nodes the deferred-join decision cannot see into)
static __global__ __launch_bounds__(RB) void offset_combine_kernel(const float* __restrict__ stat, const float* __restrict__ dyn,
                                                                   const long long* __restrict__ ts, int N, int n3, float* __restrict__ out) {
    const int i = blockIdx.x * RB + threadIdx.x, b = blockIdx.y;
    if (i >= n3) return;
    long long t = ts[b];
    t = t < 0 ? 0 : (t >= N ? N - 1 : t);
    out[(size_t)b * n3 + i] = (stat ? stat[i] : 0.f) + dyn[(size_t)t * n3 + i];
}
extern "C" int vhap_offset_combine(const float* static_offset, const float* dynamic_offset, const int64_t* timesteps, int B, int N, int V,
                                   float* out, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!dynamic_offset || !timesteps || !out) return VHAP_E_NULLPTR;
    if (B <= 0 || B > 65535 || N <= 0 || V <= 0) return VHAP_E_BADDIM;
    offset_combine_kernel<<<dim3(vhap_cdiv(3ll * V, RB), B), RB, 0, vhap_stream(stream)>>>(static_offset, dynamic_offset,
                                                                                          reinterpret_cast<const long long*>(timesteps), N, 3 * V, out);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_offset_reg_bwd(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col, const float* lap_val,
                                   const float* w_lap, const float* w_abs, const int32_t* region_ptr, const int32_t* region_idx,
                                   int V, int n_regions, float s_lap, float s_abs, float s_rigid, const float* d_terms, float* d_offset,
                                   vhap_stream_t stream) {
    VHAP_ENTER();
    if (!offset || !lap_ptr || !lap_col || !lap_val || !d_terms || !d_offset) return VHAP_E_NULLPTR;
    if (n_regions > 0 && (!region_ptr || !region_idx)) return VHAP_E_NULLPTR;
    if (V <= 0 || n_regions < 0) return VHAP_E_BADDIM;
    OffCfg c{V, n_regions, vhap_cdiv(V, RB), s_lap, s_abs, s_rigid};
    offset_reg_kernel<<<c.nbv + n_regions, RB, 0, vhap_stream(stream)>>>(c, offset, lap_ptr, lap_col, lap_val, w_lap, w_abs, region_ptr, region_idx,
                                                                         nullptr, d_terms, d_offset, 0);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_tex_prep_fwd(const float* painted, const float* extra, const uint8_t* res_mask, int T, float s_tv, float s_res,
                                 float* albedo_hwc, float* terms, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if ((!painted && !extra) || !albedo_hwc || !terms) return VHAP_E_NULLPTR;
    if (T <= 0) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(terms, 2 * sizeof(float), st);
    TexCfg c{T, s_tv, s_res};
    const dim3 grid(vhap_cdiv(T, TEX_BCOLS), vhap_cdiv(T, TEX_ROWS));
    if (painted && extra && res_mask) tex_prep_fwd_kernel<true><<<grid, RB, 0, st>>>(c, painted, extra, res_mask, albedo_hwc, terms, nullptr);
    else tex_prep_fwd_kernel<false><<<grid, RB, 0, st>>>(c, painted, extra, res_mask, albedo_hwc, terms, nullptr);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_tex_prep_mip1_fwd(const float* painted, const float* extra, const uint8_t* res_mask, int T, float s_tv, float s_res,
                                      float* albedo_hwc, float* mips_hwc, float* terms, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if ((!painted && !extra) || !albedo_hwc || !terms || !mips_hwc) return VHAP_E_NULLPTR;
    if (T <= 0 || (T & 1)) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(terms, 2 * sizeof(float), st);
    TexCfg c{T, s_tv, s_res};
    const dim3 grid(vhap_cdiv(T, TEX_BCOLS), vhap_cdiv(T, TEX_ROWS));
    if (painted && extra && res_mask) tex_prep_fwd_kernel<true><<<grid, RB, 0, st>>>(c, painted, extra, res_mask, albedo_hwc, terms, mips_hwc);
    else tex_prep_fwd_kernel<false><<<grid, RB, 0, st>>>(c, painted, extra, res_mask, albedo_hwc, terms, mips_hwc);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_tex_prep_bwd(const float* albedo_hwc, const float* extra, const uint8_t* res_mask, const float* d_albedo_hwc,
                                 const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res,
                                 float* d_extra, vhap_stream_t stream) {
    return vhap_tex_prep_bwd_base(albedo_hwc, extra, res_mask, d_albedo_hwc, d_mips_hwc, n_gather, d_terms, T, s_tv, s_res, d_extra, nullptr, stream);
}

// ... with d_base [3,T,T] (may be NULL): d(base texture) = d(albedo) without the residual's own regulariser (PCA texture model)
extern "C" int vhap_tex_prep_bwd_base(const float* albedo_hwc, const float* extra, const uint8_t* res_mask, const float* d_albedo_hwc,
                                      const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res,
                                      float* d_extra, float* d_base, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!albedo_hwc || !extra || !d_terms || !d_extra) return VHAP_E_NULLPTR;
    if (T <= 0 || n_gather < 0 || n_gather > TEXB_MAXG || (d_mips_hwc && n_gather > 0 && (T & ((1 << n_gather) - 1)))) return VHAP_E_BADDIM;
    TexCfg c{T, s_tv, s_res};
    tex_prep_bwd_kernel<false><<<dim3(vhap_cdiv(T, RB), vhap_cdiv(T, TEXB_ROWS)), RB, 0, vhap_stream(stream)>>>(
        c, albedo_hwc, extra, res_mask, d_albedo_hwc, n_gather > 0 ? d_mips_hwc : nullptr, n_gather, d_terms, d_extra, TexAdam{}, 0, 0, d_base, TexCarry{});
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_tex_prep_bwd_adam(const float* albedo_hwc, float* extra, const uint8_t* res_mask, const float* d_albedo_hwc,
                                      const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res,
                                      float* d_extra, float* exp_avg, float* exp_avg_sq, const float* lr_device, const int32_t* step_device,
                                      float beta1, float beta2, float eps, int call_flags, vhap_stream_t stream) {
    return vhap_tex_prep_bwd_adam_base(albedo_hwc, extra, res_mask, d_albedo_hwc, d_mips_hwc, n_gather, d_terms, T, s_tv, s_res, d_extra, exp_avg,
                                       exp_avg_sq, lr_device, step_device, beta1, beta2, eps, nullptr, call_flags, stream);
}

extern "C" int vhap_tex_prep_bwd_adam_base(const float* albedo_hwc, float* extra, const uint8_t* res_mask, const float* d_albedo_hwc,
                                           const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res,
                                           float* d_extra, float* exp_avg, float* exp_avg_sq, const float* lr_device, const int32_t* step_device,
                                           float beta1, float beta2, float eps, float* d_base, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!albedo_hwc || !extra || !d_terms || !d_extra || !exp_avg || !exp_avg_sq || !lr_device || !step_device) return VHAP_E_NULLPTR;
    if (T <= 0 || n_gather < 0 || n_gather > TEXB_MAXG || (d_mips_hwc && n_gather > 0 && (T & ((1 << n_gather) - 1)))) return VHAP_E_BADDIM;
    TexCfg c{T, s_tv, s_res};
    tex_prep_bwd_kernel<true><<<dim3(vhap_cdiv(T, RB), vhap_cdiv(T, TEXB_ROWS)), RB, 0, vhap_stream(stream)>>>(
        c, albedo_hwc, extra, res_mask, d_albedo_hwc, n_gather > 0 ? d_mips_hwc : nullptr, n_gather, d_terms, d_extra,
        TexAdam{extra, exp_avg, exp_avg_sq, lr_device, step_device, beta1, beta2, eps}, (call_flags & VHAP_CALL_ADAM_STEP_ADVANCED) ? 0 : 1, 0, d_base, TexCarry{});
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// ---- the carried texture (see TexCarry) ----
extern "C" size_t vhap_tex_carry_halo_floats(int T) {
    if (T <= 0 || (T % 64)) return 0;
    return 2 * (tex_carry_row_halo_floats(T) + tex_carry_col_halo_floats(T));
}

// albedo = painted + extra, level 1 of the pyramid and both parities of the halo copies, from scratch
extern "C" int vhap_tex_carry_prime(const float* painted, const float* extra, const uint8_t* res_mask, int T, float s_tv, float s_res,
                                    float* albedo_hwc, float* mips_hwc, float* halo, float* terms, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!painted || !extra || !albedo_hwc || !mips_hwc || !halo || !terms) return VHAP_E_NULLPTR;
    if (T <= 0 || (T % 64)) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    vhap_zero_async(terms, 2 * sizeof(float), st);
    VHAP_LAUNCH_CHECK();
    TexCfg c{T, s_tv, s_res};
    const dim3 grid(vhap_cdiv(T, TEX_BCOLS), vhap_cdiv(T, TEX_ROWS));
    if (res_mask) tex_prep_fwd_kernel<true><<<grid, RB, 0, st>>>(c, painted, extra, res_mask, albedo_hwc, terms, mips_hwc);
    else tex_prep_fwd_kernel<false><<<grid, RB, 0, st>>>(c, painted, extra, res_mask, albedo_hwc, terms, mips_hwc);
    VHAP_LAUNCH_CHECK();
    const size_t n = (tex_carry_row_halo_floats(T) + tex_carry_col_halo_floats(T)) / 3;
    tex_carry_halo_init_kernel<<<(unsigned)vhap_cdiv((long long)n, RB), RB, 0, st>>>(T, albedo_hwc, halo, halo + 2 * tex_carry_row_halo_floats(T));
    VHAP_LAUNCH_CHECK();
    // terms = the texture's energies WITHOUT the TV pairs across the finish pass's ownership tiles -- what vhap_tex_finish_carry leaves
    // behind -- so that the vhap_tex_carry_border every step issues completes them after a prime as it does after a finish pass
    const size_t nb = (size_t)(T / TEXB_ROWS - 1) * T + (size_t)T * ((T >> 6) - 1);
    if (s_tv != 0.f && nb) {
        tex_carry_border_kernel<<<(unsigned)std::min(vhap_cdiv((long long)nb, RB), 128), RB, 0, st>>>(T, -s_tv, nullptr, 0, halo,
                                                                                                      halo + 2 * tex_carry_row_halo_floats(T), terms);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

// vhap_tex_prep_bwd_adam + the next step's texture: see TexCarry.  d_extra may be NULL (the gradient is consumed here and not written).
extern "C" int vhap_tex_finish_carry(float* albedo_hwc, float* extra, const uint8_t* res_mask, const float* painted, const float* d_albedo_hwc,
                                     const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res,
                                     float* d_extra, float* exp_avg, float* exp_avg_sq, const float* lr_device, const int32_t* step_device,
                                     float beta1, float beta2, float eps, float* mips_hwc, float* halo, float* terms, int call_flags,
                                     vhap_stream_t stream) {
    VHAP_ENTER();
    if (!albedo_hwc || !extra || !painted || !d_terms || !exp_avg || !exp_avg_sq || !lr_device || !step_device || !mips_hwc || !halo || !terms)
        return VHAP_E_NULLPTR;
    if (T <= 0 || (T % 64) || n_gather < 0 || n_gather > TEXB_MAXG || (d_mips_hwc && n_gather > 0 && (T & ((1 << n_gather) - 1)))) return VHAP_E_BADDIM;
    TexCfg c{T, s_tv, s_res};
    float* col_halo = halo + 2 * tex_carry_row_halo_floats(T);
    const TexCarry C{painted, mips_hwc, halo, col_halo, terms, d_extra ? 1 : 0};
    const int step_add = (call_flags & VHAP_CALL_ADAM_STEP_ADVANCED) ? 0 : 1;
    hipStream_t st = vhap_stream(stream);
    tex_prep_bwd_kernel<true, true><<<dim3(vhap_cdiv(T, RB), T / TEXB_ROWS), RB, 0, st>>>(
        c, albedo_hwc, extra, res_mask, d_albedo_hwc, n_gather > 0 ? d_mips_hwc : nullptr, n_gather, d_terms, d_extra ? d_extra : extra,
        TexAdam{extra, exp_avg, exp_avg_sq, lr_device, step_device, beta1, beta2, eps}, step_add, 0, nullptr, C);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// The TV pairs vhap_tex_finish_carry leaves out (those that straddle two of its ownership tiles), added into terms[0] from the halo copies
// it wrote.  A call of its own so that it need not sit between the finish pass and the pyramid the next step's rasteriser waits for:
// issue it anywhere behind the finish pass and ahead of (i) the energy assembly that consumes terms and (ii) the next advance of the
// step counter, whose parity tells which halo copy is the new one (call_flags as given to vhap_tex_finish_carry).
extern "C" int vhap_tex_carry_border(int T, float s_tv, const int32_t* step_device, const float* halo, float* terms, int call_flags,
                                     vhap_stream_t stream) {
    VHAP_ENTER();
    if (!step_device || !halo || !terms) return VHAP_E_NULLPTR;
    if (T <= 0 || (T % 64)) return VHAP_E_BADDIM;
    const size_t n = (size_t)(T / TEXB_ROWS - 1) * T + (size_t)T * ((T >> 6) - 1);
    if (s_tv == 0.f || n == 0) return VHAP_OK;
    tex_carry_border_kernel<<<(unsigned)std::min(vhap_cdiv((long long)n, RB), 128), RB, 0, vhap_stream(stream)>>>(
        T, s_tv, step_device, (call_flags & VHAP_CALL_ADAM_STEP_ADVANCED) ? 0 : 1, halo, halo + 2 * tex_carry_row_halo_floats(T), terms);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// The same pass on a ROW STRIP [row0, row0 + nrows) of the texture (frame sharding over N GPUs: rank r finishes and updates rows
// [r T / N, (r + 1) T / N) -- the level-0 gradient arrives reduce-scattered, the updated rows go out in an all-gather; the finish + Adam
// pass, 124 us and 497 MB for the whole texture, shrinks N-fold).  d_albedo_strip: the strip's [nrows, T, 3] slice of the level-0
// gradient (the whole pyramid folded into it: vhap_texture_mip_fold(stop_level = 0)); everything else as vhap_tex_prep_bwd_adam, full size.
extern "C" int vhap_tex_prep_bwd_adam_rows(const float* albedo_hwc, float* extra, const uint8_t* res_mask, const float* d_albedo_strip,
                                           const float* d_terms, int T, int row0, int nrows, float s_tv, float s_res, float* d_extra,
                                           float* exp_avg, float* exp_avg_sq, const float* lr_device, const int32_t* step_device, float beta1,
                                           float beta2, float eps, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!albedo_hwc || !extra || !d_albedo_strip || !d_terms || !d_extra || !exp_avg || !exp_avg_sq || !lr_device || !step_device) return VHAP_E_NULLPTR;
    if (T <= 0 || row0 < 0 || nrows <= 0 || row0 + nrows > T || (row0 % TEXB_ROWS) || (nrows % TEXB_ROWS)) return VHAP_E_BADDIM;
    TexCfg c{T, s_tv, s_res};
    // (the kernel indexes the gradient by absolute row: hand it the address row 0 WOULD have; only the strip's rows are dereferenced)
    const float* d_base = d_albedo_strip - (size_t)row0 * T * 3;
    tex_prep_bwd_kernel<true><<<dim3(vhap_cdiv(T, RB), nrows / TEXB_ROWS), RB, 0, vhap_stream(stream)>>>(
        c, albedo_hwc, extra, res_mask, d_base, nullptr, 0, d_terms, d_extra,
        TexAdam{extra, exp_avg, exp_avg_sq, lr_device, step_device, beta1, beta2, eps}, (call_flags & VHAP_CALL_ADAM_STEP_ADVANCED) ? 0 : 1, row0, nullptr, TexCarry{});
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_offset_dynamic_reg(const float* dyn, const int64_t* timesteps, int B, int N, int V, float scale, const float* d_term,
                                       float* energy_accum, float* d_dyn, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!dyn || !timesteps || (!energy_accum && !d_dyn)) return VHAP_E_NULLPTR;
    if (B <= 0 || B > 65535 || N <= 0 || V <= 0) return VHAP_E_BADDIM;
    offset_dynamic_reg_kernel<<<dim3(vhap_cdiv(3ll * V, RB), B), RB, 0, vhap_stream(stream)>>>(dyn, reinterpret_cast<const long long*>(timesteps), B, N, V,
                                                                                              scale, d_term, energy_accum, d_dyn);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_offset_grad_finish(const float* g_a, const float* g_b, const int64_t* timesteps, int B, int N, int V, float* d_static,
                                       float* d_dyn, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!g_a || !timesteps || (!d_static && !d_dyn)) return VHAP_E_NULLPTR;
    if (B <= 0 || N <= 0 || V <= 0) return VHAP_E_BADDIM;
    offset_grad_finish_kernel<<<vhap_cdiv(3ll * V, RB), RB, 0, vhap_stream(stream)>>>(g_a, g_b, reinterpret_cast<const long long*>(timesteps), B, N, V, d_static,
                                                                                     d_dyn);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                              float* const* exp_avg_sq, const int64_t* numel, const int32_t* lr_index, const float* lr_device,
                              int32_t* step_device, float beta1, float beta2, float eps, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!params || !grads || !exp_avg || !exp_avg_sq || !numel || !lr_index || !lr_device || !step_device) return VHAP_E_NULLPTR;
    if (n_tensors <= 0 || n_tensors > VHAP_ADAM_MAX_TENSORS) return VHAP_E_BADDIM;
    AdamTable t;
    int nblocks = 0;
    t.n_tensors = n_tensors;
    for (int k = 0; k < n_tensors; k++) {
        if (!params[k] || !grads[k] || !exp_avg[k] || !exp_avg_sq[k]) return VHAP_E_NULLPTR;
        if (numel[k] < 0 || lr_index[k] < 0) return VHAP_E_BADDIM;
        t.p[k] = params[k]; t.g[k] = grads[k]; t.m[k] = exp_avg[k]; t.v[k] = exp_avg_sq[k]; t.n[k] = numel[k]; t.lr_index[k] = lr_index[k];
        const long long want = (numel[k] + RB * 4 - 1) / (RB * 4);
        t.blk_start[k] = nblocks;
        nblocks += (int)(want < 1 ? 1 : (want > 4096 ? 4096 : want));
    }
    t.blk_start[n_tensors] = nblocks;
    hipStream_t st = vhap_stream(stream);
    const bool advanced = (call_flags & VHAP_CALL_ADAM_STEP_ADVANCED) != 0;      // the counter already names THIS step (vhap_adam_advance)
    adam_kernel<<<nblocks, RB, 0, st>>>(t, lr_device, step_device, advanced ? 0 : 1, beta1, beta2, eps);
    VHAP_LAUNCH_CHECK();
    if (!advanced && !(call_flags & VHAP_CALL_ADAM_KEEP_STEP)) {
        adam_bump_kernel<<<1, 1, 0, st>>>(step_device);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

// n <= 16 floats handed over BY VALUE (kernel arguments): a learning-rate table changes once per epoch, and neither a pageable
// hipMemcpy (blocks the host until the stream has drained: the whole queue of replays) nor a pinned staging buffer (lifetime) is
// wanted for 2 - 12 numbers
struct FloatVals {
    float v[16];
};
static __global__ void set_floats_kernel(float* __restrict__ dst, FloatVals vals, int n) {
    if ((int)threadIdx.x < n) dst[threadIdx.x] = vals.v[threadIdx.x];
}
extern "C" int vhap_set_floats(float* dst_device, const float* values_host, int n, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!dst_device || !values_host) return VHAP_E_NULLPTR;
    if (n <= 0 || n > 16) return VHAP_E_BADDIM;
    FloatVals v{};
    memcpy(v.v, values_host, sizeof(float) * n);      // bits, not values: callers also pass int32 patterns (subnormals / NaNs as floats)
    set_floats_kernel<<<1, 64, 0, vhap_stream(stream)>>>(dst_device, v, n);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_adam_advance(int32_t* step_device, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!step_device) return VHAP_E_NULLPTR;
    adam_bump_kernel<<<1, 1, 0, vhap_stream(stream)>>>(step_device);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
