// Backward of the deferred-shading part of vhap_raster_shade_fwd (raster.hip, mode 2) for gfx950.
//
// The forward keeps the interpolated normal / uv / uv derivatives and the sampled albedo in registers and writes only rast + rgba.
// This kernel RE-COMPUTES them per covered pixel from (triangle id, clip positions, vertex normals, uv table) with the forward's own
// arithmetic (frag_common.h, tex_sample.h, shade_common.h: same bits), chains the upstream colour gradient through
//      rgb = albedo * diffuse,  diffuse = SH(normalize(n)) . lights,  albedo = texture(uv, uv_da)
// (render_nvdiffrast.py:386-421, 399) and emits what the two remaining consumers need: (uv, uv_da, d_albedo) for the texture-gradient
// accumulation and (d_normal, d_uv, d_uv_da) for the G-buffer backward; d_lights is reduced per workgroup.  One pass instead of
// vhap_shade_bwd + vhap_texture_bwd(uv part), and none of normal / texc / texd / albedo / rast_db is ever read back from HBM.
#include "common.h"
#include "frag_common.h"
#include "shade_common.h"
#include "tex_sample.h"

namespace {

constexpr int DB_T = 256;
constexpr int DB_NW = DB_T / 64;
constexpr int DB_SLOTS = 64;      // partial-sum rows of the lights gradient: workgroup w adds into row w % 64 (chains of ~256 atomics per address)

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
    const float* keep;
    const float* d_reg;
    const unsigned* stats;
    int B, V, F, H, W;
    float xs, xo, ys, yo;
    float2* texc;
    float4* texd;
    float* d_albedo;
    float* d_normal;
    float2* d_texc;
    float4* d_texd;
    float* part;             // [DB_SLOTS][27] partial sums of d_lights (zero on entry)
    unsigned* tb_counts;     // optional: the uv-tile histogram of vhap_texture_grad_binned (counts / max|g| bits per tile, zero on entry)
    unsigned* tb_max;
    int NT;
};

// One thread = one pixel, no loop: nothing is carried between pixels, so the register budget is set by the gather chain of ONE pixel
// (rast -> triangle -> vertices / normals / uvs -> texture taps) and several waves per SIMD overlap those round trips.  The [9,3] lights
// gradient is reduced per wave (DPP, on the VALU) and per workgroup, added into one of DB_SLOTS rows of `part` (short atomic chains), and a
// second tiny launch sums the rows into d_lights.
__global__ __launch_bounds__(DB_T) void deferred_shade_bwd_kernel(const DeferredParams P) {
    __shared__ float s_l[27], s_c[9];
    __shared__ float red[DB_NW][27];
    if (threadIdx.x < 27) s_l[threadIdx.x] = P.lights[threadIdx.x];
    if (threadIdx.x < 9) s_c[threadIdx.x] = P.sh_const[threadIdx.x];
    __syncthreads();
    const unsigned HW = (unsigned)P.H * P.W, npix = (unsigned)P.B * HW;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // regulariser part of d(diffuse) (lights only, on shade(normal.detach()): tracker.py:547-550), see shade_bwd_kernel
    float g_var = 0.f, g_max = 0.f;
    unsigned mx_ord = 0u;
    if (P.d_reg && P.stats) {
        const float dr = P.d_reg[0];
        g_var = dr / (float)npix;
        mx_ord = P.stats[1];
        const unsigned ties = P.stats[0];
        const unsigned u = (mx_ord & 0x80000000u) ? (mx_ord & 0x7fffffffu) : ~mx_ord;
        g_max = __uint_as_float(u) > 1.0f ? dr / (float)max(ties, 1u) : 0.f;
    }
    const bool reg_on = g_var != 0.f || g_max != 0.f;
    auto reg_grad = [&](const float (&d)[3], float (&gr)[3]) {
        const float mean = (d[0] + d[1] + d[2]) * (1.0f / 3.0f);
#pragma unroll
        for (int c = 0; c < 3; c++) {
            gr[c] = g_var * (d[c] - mean);
            if (g_max != 0.f && sh_f2ord(d[c]) == mx_ord) gr[c] += g_max;
        }
    };
    const unsigned pi = blockIdx.x * DB_T + threadIdx.x;
    const bool valid = pi < npix;
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) r = P.rast[pi];
    const int t = (int)r.w - 1;
    const bool cov = valid && t >= 0 && t < P.F;
    float gl[27];
#pragma unroll
    for (int i = 0; i < 27; i++) gl[i] = 0.f;
    if (valid && !cov) {                     // background: nothing flows (its colour is the detached target / a constant)
        float* da = P.d_albedo + 3 * (size_t)pi;
        da[0] = 0.f; da[1] = 0.f; da[2] = 0.f;
    }
    const unsigned long long covm = __ballot(cov);
    int tb_tile = -1;
    float tb_g = 0.f;
    if (covm) {
        if (cov) {
            const unsigned b = pi / HW, rem = pi - b * HW;
            const unsigned py = rem / (unsigned)P.W, px = rem - py * (unsigned)P.W;
            const int i0 = P.tri[3 * t], i1 = P.tri[3 * t + 1], i2 = P.tri[3 * t + 2];
            const int j0 = P.tri_uv[3 * t], j1 = P.tri_uv[3 * t + 1], j2 = P.tri_uv[3 * t + 2];
            float4 g;
            if (P.d_rgba) {
                g = P.d_rgba[pi];
            } else {                          // d sum|gt - pred| / d pred = -sign(gt - pred) (tracker.py:430-439), scaled by d_sum
                const float gs = P.d_sum[0];
                const float* gp = P.gt + (size_t)b * 3 * HW + (size_t)(P.H - 1 - py) * P.W + px;
                const float4 p = P.pred[pi];
                auto sg = [](float e) { return e > 0.f ? 1.0f : (e < 0.f ? -1.0f : 0.0f); };
                g = make_float4(-sg(gp[0] - p.x) * gs, -sg(gp[HW] - p.y) * gs, -sg(gp[2 * HW] - p.z) * gs, 0.0f);
            }
            if (P.keep) { const float k = P.keep[pi]; g.x *= k; g.y *= k; g.z *= k; }     // backward of the colour disturbance, folded in
            const float4* PV = P.pos + (size_t)b * P.V;
            const float4 p0 = PV[i0], p1 = PV[i1], p2 = PV[i2];
            const float fx = __fmaf_rn(P.xs, (float)px, P.xo), fy = __fmaf_rn(P.ys, (float)py, P.yo);
            const Frag fr = shade_frag(p0, p1, p2, fx, fy);
            const float4 o_db = frag_db(p0, p1, p2, fr, P.xs, P.ys);
            const FragAttr at = frag_attr(P.vnormal + (size_t)b * P.V * 3, P.uv, i0, i1, i2, j0, j1, j2, fr, o_db);
            P.texc[pi] = make_float2(at.tu, at.tv);
            P.texd[pi] = at.td;
            SH9 bsh;
            float x, y, z, inv, d[3];
            sh_diffuse(at.n0, at.n1, at.n2, s_c, s_l, bsh, x, y, z, inv, d);
            const float ga[3] = {g.x * d[0], g.y * d[1], g.z * d[2]};                     // d L / d albedo
            float* da = P.d_albedo + 3 * (size_t)pi;
            da[0] = ga[0]; da[1] = ga[1]; da[2] = ga[2];
            tb_g = fmaxf(fabsf(ga[0]), fmaxf(fabsf(ga[1]), fabsf(ga[2])));
            if (tb_g != 0.f) tb_tile = tile_of(make_float2(at.tu, at.tv), P.NT);      // same criterion and tile as texbin_pass_kernel
            float2 guv;
            float4 gda;
            float alb[3];
            tex_sample_bwd_uv<3>(P.tex, P.mips, P.D, 0, make_float2(at.tu, at.tv), at.td, ga, nullptr, nullptr, guv, gda, true, alb);
            P.d_texc[pi] = guv;
            P.d_texd[pi] = gda;
            const float gd[3] = {g.x * alb[0], g.y * alb[1], g.z * alb[2]};               // photometric part of d L / d diffuse
            float gnx, gny, gnz;
            const float l2 = at.n0 * at.n0 + at.n1 * at.n1 + at.n2 * at.n2;
            sh_normal_bwd(x, y, z, inv, !(l2 > 1e-20f), s_c, s_l, gd, gnx, gny, gnz);
            float* dn = P.d_normal + 3 * (size_t)pi;
            dn[0] = gnx; dn[1] = gny; dn[2] = gnz;
            float gr[3] = {0.f, 0.f, 0.f};
            if (reg_on) reg_grad(d, gr);
#pragma unroll
            for (int k = 0; k < 9; k++) {
                gl[3 * k] = bsh.v[k] * (gd[0] + gr[0]); gl[3 * k + 1] = bsh.v[k] * (gd[1] + gr[1]); gl[3 * k + 2] = bsh.v[k] * (gd[2] + gr[2]);
            }
        }
#pragma unroll
        for (int i = 0; i < 27; i++) gl[i] = vhap_wave_sum_dpp(gl[i]);
        if (P.tb_counts) {
            // count pass of the uv-space binning, fused: a wave covers 64 consecutive pixels of a row, which sample one or two uv tiles --
            // one pair of atomics per (wave, tile) instead of a separate pass over uv / d_albedo
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
    }
    if (!P.part) return;
    if (reg_on) {
        // background pixels: normal 0 -> the same basis / diffuse colour for all of them; counted per wave
        const int n_bg = __popcll(__ballot(valid && !cov));
        if (n_bg) {
            SH9 bsh;
            float x, y, z, inv, d[3], gr[3];
            sh_diffuse(0.f, 0.f, 0.f, s_c, s_l, bsh, x, y, z, inv, d);
            reg_grad(d, gr);
            const float nb = (float)n_bg;
#pragma unroll
            for (int k = 0; k < 9; k++) {
                gl[3 * k] += nb * (bsh.v[k] * gr[0]); gl[3 * k + 1] += nb * (bsh.v[k] * gr[1]); gl[3 * k + 2] += nb * (bsh.v[k] * gr[2]);
            }
        }
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 27; i++) red[wave][i] = gl[i];
    }
    __syncthreads();
    if (threadIdx.x < 27) {
        float s = 0.f;
        for (int w = 0; w < DB_NW; w++) s += red[w][threadIdx.x];
        if (s != 0.f) atomicAdd(&P.part[(size_t)(blockIdx.x % DB_SLOTS) * 27 + threadIdx.x], s);
    }
}

// d_lights[i] += sum over the DB_SLOTS rows of part: one wave per entry
__global__ __launch_bounds__(64) void deferred_lights_reduce_kernel(const float* __restrict__ part, float* __restrict__ d_lights) {
    const float s = vhap_wave_sum_dpp(part[(size_t)threadIdx.x * 27 + blockIdx.x]);
    if (threadIdx.x == 0) d_lights[blockIdx.x] += s;
}

}  // namespace

extern "C" size_t vhap_deferred_shade_bwd_work_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)DB_SLOTS * 27;
}

extern "C" int vhap_deferred_shade_bwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                       const float* tex, const float* mips, int Ht, int Wt, const float* lights, const float* sh_const,
                                       const float* rast, const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                                       const float* d_sum, const float* keep, const float* d_reg, const float* stats,
                                       int B, int V, int VT, int F, int H, int W, float* texc, float* texd, float* d_albedo,
                                       float* d_normal, float* d_texc, float* d_texd, float* d_lights, float* work, size_t work_floats,
                                       void* texbin_work, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !vnormal || !uv || !tri_uv || !tex || !lights || !sh_const || !rast || !texc || !texd || !d_albedo ||
        !d_normal || !d_texc || !d_texd)
        return VHAP_E_NULLPTR;
    if (!d_rgba && (!pred_rgba || !gt_nchw || !d_sum)) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || VT <= 0 || F <= 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || (long long)B * H * W >= (1ll << 31))
        return VHAP_E_BADDIM;
    if (d_lights && (!work || work_floats < vhap_deferred_shade_bwd_work_floats(B, H, W))) return VHAP_E_WORKSPACE;
    DeferredParams P{};
    P.pos = reinterpret_cast<const float4*>(pos); P.tri = tri; P.vnormal = vnormal; P.uv = reinterpret_cast<const float2*>(uv);
    P.tri_uv = tri_uv; P.tex = tex; P.mips = mips; P.D = make_desc(1, Ht, Wt, 3);
    if (P.D.L > 0 && !mips) return VHAP_E_NULLPTR;
    P.lights = lights; P.sh_const = sh_const; P.rast = reinterpret_cast<const float4*>(rast);
    P.d_rgba = reinterpret_cast<const float4*>(d_rgba); P.pred = reinterpret_cast<const float4*>(pred_rgba); P.gt = gt_nchw; P.d_sum = d_sum;
    P.keep = keep; P.d_reg = d_reg; P.stats = reinterpret_cast<const unsigned*>(stats);
    P.B = B; P.V = V; P.F = F; P.H = H; P.W = W;
    P.xs = 2.0f / (float)W; P.xo = 1.0f / (float)W - 1.0f; P.ys = 2.0f / (float)H; P.yo = 1.0f / (float)H - 1.0f;
    P.texc = reinterpret_cast<float2*>(texc); P.texd = reinterpret_cast<float4*>(texd); P.d_albedo = d_albedo; P.d_normal = d_normal;
    P.d_texc = reinterpret_cast<float2*>(d_texc); P.d_texd = reinterpret_cast<float4*>(d_texd);
    P.part = d_lights ? work : nullptr;
    if (texbin_work) {                          // layout of vhap_texture_grad_binned's workspace: counts, then max|g| bits
        const TexBinWs l = texbin_layout(1);
        P.tb_counts = reinterpret_cast<unsigned*>(static_cast<char*>(texbin_work) + l.counts);
        P.tb_max = reinterpret_cast<unsigned*>(static_cast<char*>(texbin_work) + l.tilemax);
        P.NT = texbin_nt(Ht, Wt);
    }
    const long long npix = (long long)B * H * W;
    const int blocks = (int)((npix + DB_T - 1) / DB_T);
    hipStream_t st = vhap_stream(stream);
    deferred_shade_bwd_kernel<<<blocks, DB_T, 0, st>>>(P);
    VHAP_LAUNCH_CHECK();
    if (d_lights) {
        deferred_lights_reduce_kernel<<<27, DB_SLOTS, 0, st>>>(work, d_lights);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}
