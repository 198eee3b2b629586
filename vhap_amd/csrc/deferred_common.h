// Per-pixel backward of the deferred-shading forward (raster.hip mode 2), shared by the stand-alone pass (deferred.hip) and the pass
// fused with the G-buffer backward (interp.hip): re-compute the interpolants of a covered pixel from (triangle id, geometry) with the
// forward's own arithmetic, chain the colour gradient through rgb = albedo * diffuse, the SH shading and the texture fetch.
#pragma once
#include "common.h"
#include "frag_common.h"
#include "shade_common.h"
#include "tex_sample.h"

namespace {

constexpr int DB_SLOTS = 64;      // partial-sum rows of the lights gradient: workgroup w adds into row w % 64 (short atomic chains)
constexpr int DB_ROW = 28;        // 27 sums + the number of background pixels (their regulariser term is one constant, added once)

struct DeferredParams {
    const float4* pos;       // [B,V,4]
    const int* tri;          // [F,3]
    const float* vnormal;    // [B,V,3]
    const float2* uv;        // [VT,2]
    const int* tri_uv;       // [F,3]
    const float* tex;
    const float* mips;
    TexDesc D;
    const float* lights;
    const float* sh_const;
    const float4* rast;
    const float4* d_rgba;    // upstream gradient image, or null: photometric gradient on the fly from (pred, gt, d_sum)
    const float4* pred;      // [B,H,W,4] the antialiased prediction (renderer space)
    const float* gt;         // [B,3,H,W] target (image space)
    const float* d_sum;      // device scalar: d E / d sum|gt - pred|
    const float4* d_delta;   // optional with the on-the-fly gradient: sparse additional colour gradient (antialias backward), zero elsewhere
    int delta_unscaled;      // d_delta is per unit of the upstream gradient d_sum (VHAP_CALL_DELTA_UNSCALED): multiplied here
    const float* keep;
    const float* d_reg;
    const unsigned* stats;
    int B, V, F, H, W;
    float xs, xo, ys, yo;
    float2* texc;
    float4* texd;
    float* d_albedo;
    float* d_normal;         // (stand-alone pass only)
    float2* d_texc;
    float4* d_texd;
    float* part;             // [DB_SLOTS][DB_ROW] partial sums of d_lights (zero on entry)
    unsigned* tb_counts;     // optional: the uv-tile histogram of vhap_texture_grad_binned (counts / max|g| bits per tile, zero on entry)
    unsigned* tb_max;
    int NT;
    unsigned short* tile_ids; // optional [B,H,W]: uv tile of every pixel for vhap_texture_grad_binned_ids (0xFFFF = no gradient)
    int skip_bg;             // VHAP_CALL_SKIP_BG_GRAD: d_albedo of background pixels is not written (its reader walks a list of covered pixels)
    int tiled;               // deferred_shade_bwd: 0 = a workgroup is 256 consecutive pixels of a row; 1 = a 16 x 16 tile, a wave an 8 x 8 block of it
    int tiles_x, tiles_y;    // (tiled) 16 x 16 tiles per frame
    const unsigned* cov_list; // deferred_shade_bwd, optional: the covered pixels in pixel order (vhap_disturb_inplace_list) -- thread k takes pixel cov_list[k]
    const int* n_bg;          // ... and the number of background pixels (device): npix - *n_bg entries
};

// regulariser part of d(diffuse) (lights only, on shade(normal.detach()): tracker.py:547-550), see shade_bwd_kernel
struct DiffuseReg {
    float g_var, g_max;
    unsigned mx_ord;
    bool on;
};
__device__ __forceinline__ DiffuseReg diffuse_reg(const float* d_reg, const unsigned* stats, unsigned npix) {
    DiffuseReg R{0.f, 0.f, 0u, false};
    if (d_reg && stats) {
        const float dr = d_reg[0];
        R.g_var = dr / (float)npix;
        R.mx_ord = stats[1];
        const unsigned ties = stats[0];
        const unsigned u = (R.mx_ord & 0x80000000u) ? (R.mx_ord & 0x7fffffffu) : ~R.mx_ord;
        R.g_max = __uint_as_float(u) > 1.0f ? dr / (float)max(ties, 1u) : 0.f;
    }
    R.on = R.g_var != 0.f || R.g_max != 0.f;
    return R;
}
__device__ __forceinline__ void diffuse_reg_grad(const DiffuseReg& R, const float (&d)[3], float (&gr)[3]) {
    const float mean = (d[0] + d[1] + d[2]) * (1.0f / 3.0f);
#pragma unroll
    for (int c = 0; c < 3; c++) {
        gr[c] = R.g_var * (d[c] - mean);
        if (R.g_max != 0.f && sh_f2ord(d[c]) == R.mx_ord) gr[c] += R.g_max;
    }
}

struct DeferredGrad {
    float gn[3];     // d L / d (raw interpolated normal)
    float2 guv;      // d L / d uv
    float4 gda;      // d L / d (uv screen-space derivatives)
    int i0, i1, i2, j0, j1, j2;
    float4 p0, p1, p2;
};

// One covered pixel `pi` = (b, py, px) with triangle t.  Writes texc / texd / d_albedo of the pixel, returns the gradients that flow on
// into the G-buffer backward and this pixel's 27 contributions to d_lights in gl.  s_l / s_c: lights [27] and SH constants [9] (LDS).
__device__ __forceinline__ DeferredGrad deferred_pixel(const DeferredParams& P, const DiffuseReg& R, const float* s_l, const float* s_c,
                                                       unsigned pi, unsigned b, unsigned py, unsigned px, int t, float (&gl)[27],
                                                       int& tb_tile, float& tb_g) {
    DeferredGrad o;
    const unsigned HW = (unsigned)P.H * P.W;
    o.i0 = P.tri[3 * t]; o.i1 = P.tri[3 * t + 1]; o.i2 = P.tri[3 * t + 2];
    o.j0 = P.tri_uv[3 * t]; o.j1 = P.tri_uv[3 * t + 1]; o.j2 = P.tri_uv[3 * t + 2];
    const float4* PV = P.pos + (size_t)b * P.V;
    o.p0 = PV[o.i0]; o.p1 = PV[o.i1]; o.p2 = PV[o.i2];
    const float fx = __fmaf_rn(P.xs, (float)px, P.xo), fy = __fmaf_rn(P.ys, (float)py, P.yo);
    const Frag fr = shade_frag(o.p0, o.p1, o.p2, fx, fy);
    const float4 o_db = frag_db(o.p0, o.p1, o.p2, fr, P.xs, P.ys);
    const FragAttr at = frag_attr(P.vnormal + (size_t)b * P.V * 3, P.uv, o.i0, o.i1, o.i2, o.j0, o.j1, o.j2, fr, o_db);
    // The texture taps and every load that depends on the pixel index alone (the upstream gradient's inputs) are ISSUED together here and
    // consumed below: behind their own uniform branches the compiler waited for each in turn (pred / gt -> d_delta -> keep -> level
    // offset -> taps -> level offset -> taps: seven dependent round trips where one does, of the ~11 a covered wave walked through at
    // 75 % of its cycles in s_waitcnt -- profiles/r04_call22_step_sq_pmc.json).
    const TexFetch<3> tf = tex_fetch<3>(P.tex, P.mips, P.D, 0, make_float2(at.tu, at.tv), at.td);
    float4 g = make_float4(0.f, 0.f, 0.f, 0.f), pr = g, e = g;
    float gs = 0.f, gt0 = 0.f, gt1 = 0.f, gt2 = 0.f, kp = 1.0f;
    const bool fly = P.d_rgba == nullptr;
    if (fly) {
        const float* gp = P.gt + (size_t)b * 3 * HW + (size_t)(P.H - 1 - py) * P.W + px;
        gs = P.d_sum[0];
        pr = P.pred[pi];
        gt0 = gp[0]; gt1 = gp[HW]; gt2 = gp[2 * HW];
        if (P.d_delta) e = P.d_delta[pi];
    } else {
        g = P.d_rgba[pi];
    }
    if (P.keep) kp = P.keep[pi];
    P.texc[pi] = make_float2(at.tu, at.tv);      // (stores after the batch of loads: one in-order counter covers both on gfx9, and the
    P.texd[pi] = at.td;                          //  wait for the loads must not also wait for these)
    SH9 bsh;
    float x, y, z, inv, d[3];
    sh_diffuse(at.n0, at.n1, at.n2, s_c, s_l, bsh, x, y, z, inv, d);
    if (fly) {                        // d sum|gt - pred| / d pred = -sign(gt - pred) (tracker.py:430-439), scaled by d_sum
        auto sg = [](float v) { return v > 0.f ? 1.0f : (v < 0.f ? -1.0f : 0.0f); };
        const float k = P.delta_unscaled ? gs : 1.0f;   // (e = 0 without d_delta)
        g = make_float4(-sg(gt0 - pr.x) * gs + k * e.x, -sg(gt1 - pr.y) * gs + k * e.y, -sg(gt2 - pr.z) * gs + k * e.z, 0.0f);
    }
    g.x *= kp; g.y *= kp; g.z *= kp;                     // backward of the colour disturbance, folded in (kp = 1 without it)
    const float ga[3] = {g.x * d[0], g.y * d[1], g.z * d[2]};                     // d L / d albedo
    float* da = P.d_albedo + 3 * (size_t)pi;
    da[0] = ga[0]; da[1] = ga[1]; da[2] = ga[2];
    tb_g = fmaxf(fabsf(ga[0]), fmaxf(fabsf(ga[1]), fabsf(ga[2])));
    tb_tile = ((P.tb_counts || P.tile_ids) && tb_g != 0.f) ? tile_of(make_float2(at.tu, at.tv), P.NT) : -1;   // criterion / tile of texbin_pass_kernel
    if (P.tile_ids) P.tile_ids[pi] = (unsigned short)(tb_tile < 0 ? 0xFFFF : tb_tile);
    float alb[3];
    tex_bwd_uv<3>(tf, P.D, ga, o.guv, o.gda, alb);
    const float gd[3] = {g.x * alb[0], g.y * alb[1], g.z * alb[2]};               // photometric part of d L / d diffuse
    const float l2 = at.n0 * at.n0 + at.n1 * at.n1 + at.n2 * at.n2;
    sh_normal_bwd(x, y, z, inv, !(l2 > 1e-20f), s_c, s_l, gd, o.gn[0], o.gn[1], o.gn[2]);
    float gr[3] = {0.f, 0.f, 0.f};
    if (R.on) diffuse_reg_grad(R, d, gr);
#pragma unroll
    for (int k = 0; k < 9; k++) {
        gl[3 * k] = bsh.v[k] * (gd[0] + gr[0]); gl[3 * k + 1] = bsh.v[k] * (gd[1] + gr[1]); gl[3 * k + 2] = bsh.v[k] * (gd[2] + gr[2]);
    }
    return o;
}

// count pass of the uv-space binning of the texture gradient, fused: a wave covers consecutive pixels, which sample one or two uv tiles
__device__ __forceinline__ void deferred_tile_histogram(const DeferredParams& P, int tb_tile, float tb_g, int lane) {
    unsigned long long todo = __ballot(tb_tile >= 0);
    while (todo) {
        const int leader = __builtin_ctzll(todo);
        const int tl = __builtin_amdgcn_readlane(tb_tile, leader);
        const unsigned long long same = __ballot(tb_tile == tl);
        float m = tb_tile == tl ? tb_g : 0.f;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) m = fmaxf(m, __shfl_xor(m, o, 64));
        if (lane == leader) {
            atomicAdd(&P.tb_counts[tl], (unsigned)__popcll(same));
            atomicMax(&P.tb_max[tl], __float_as_uint(m));
        }
        todo &= ~same;
    }
}

// workgroup epilogue: gl = this lane's 27 contributions (zeros for lanes without a covered pixel) -> one row of the partial-sum table.
// Rows of 16 lanes are summed with DPP adds; the row leaders park their sums in LDS (red: [NW * 4][27] floats) and 27 threads finish --
// ~5 instructions per value instead of a full wave reduction each.  n_bg_wave = number of background pixels of the wave.  `any`:
// wave-uniform, false when the wave has nothing to add (its LDS rows are then zero-filled by the caller's init).  Ends with a barrier.
template <int NW>
__device__ __forceinline__ void deferred_lights_epilogue(const DeferredParams& P, float (&gl)[27], float (*red)[27], bool any, int n_bg_wave,
                                                         int* s_nbg) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (any) vhap_row_sums_dpp<27>(gl);
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < 27; i++) red[wave * 4 + (lane >> 4)][i] = any ? gl[i] : 0.f;
        if (lane == 0 && n_bg_wave) atomicAdd(s_nbg, n_bg_wave);
    }
    __syncthreads();
    float* row = P.part + (size_t)(((unsigned)blockIdx.x + (unsigned)blockIdx.y * 7u + (unsigned)blockIdx.z * 13u) % DB_SLOTS) * DB_ROW;
    if (threadIdx.x < 27) {
        float s = 0.f;
#pragma unroll
        for (int w = 0; w < NW * 4; w++) s += red[w][threadIdx.x];
        if (s != 0.f) atomicAdd(&row[threadIdx.x], s);
    } else if (threadIdx.x == 27) {
        if (*s_nbg) atomicAdd(&row[27], (float)*s_nbg);
    }
}

}  // namespace
