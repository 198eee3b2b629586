// Backward of the deferred-shading part of vhap_raster_shade_fwd (raster.hip, mode 2) for gfx950 -- the stand-alone pass.
//
// The forward keeps the interpolated normal / uv / uv derivatives and the sampled albedo in registers and writes only rast + rgba.
// This kernel RE-COMPUTES them per covered pixel from (triangle id, clip positions, vertex normals, uv table) with the forward's own
// arithmetic (frag_common.h, tex_sample.h, shade_common.h: same bits), chains the upstream colour gradient through
//      rgb = albedo * diffuse,  diffuse = SH(normalize(n)) . lights,  albedo = texture(uv, uv_da)
// (render_nvdiffrast.py:386-421, 399) and emits what the two remaining consumers need: (uv, uv_da, d_albedo) for the texture-gradient
// accumulation and (d_normal, d_uv, d_uv_da) for the G-buffer backward; d_lights is reduced per workgroup.  One pass instead of
// vhap_photo_bwd + vhap_shade_bwd + vhap_texture_bwd(uv part), and none of normal / texc / texd / albedo / rast_db is read back from HBM.
// (A variant fused with the G-buffer backward -- the gradients of the interpolants never leaving registers -- was measured in round 3:
// VALU-bound, no gain; removed in round 4, last in commit 2579046.)
#include "deferred_common.h"

namespace {

constexpr int DB_T = 256;
constexpr int DB_NW = DB_T / 64;

// One thread = one pixel, no loop: nothing is carried between pixels, so the register budget is set by the gather chain of ONE pixel
// (rast -> triangle -> vertices / normals / uvs -> texture taps) and several waves per SIMD overlap those round trips.  The [9,3] lights
// gradient is reduced per wave (DPP, on the VALU) and per workgroup, added into one of DB_SLOTS rows of `part` (short atomic chains), and a
// second tiny launch sums the rows into d_lights.
// (Round 3, 112 VGPRs = 4 waves per SIMD: forced to 5 waves -- 96 VGPRs, seven spilled dwords -- it took the same time, 168 vs 167 us in the step:
// profiles/r03_call10_plan_timeline_w5.txt.)
// (142 VGPRs = 3 waves per SIMD since round 4.  Held to 4 waves -- 128 VGPRs, 17 spilled dwords -- the list-driven step takes 0.786 instead of 0.757 ms, held
// to 5 -- 96 VGPRs, 79 spilled -- 0.88: profiles/r06_call44_shade_bwd_waves_ab.txt.)
__global__ __launch_bounds__(DB_T) void deferred_shade_bwd_kernel(const DeferredParams P) {
    __shared__ float s_l[27], s_c[9];
    __shared__ float red[DB_NW * 4][27];
    __shared__ int s_nbg;
    if (threadIdx.x < 27) s_l[threadIdx.x] = P.lights[threadIdx.x];
    if (threadIdx.x < 9) s_c[threadIdx.x] = P.sh_const[threadIdx.x];
    if (threadIdx.x == 0) s_nbg = 0;
    __syncthreads();
    const unsigned HW = (unsigned)P.H * P.W, npix = (unsigned)P.B * HW;
    const int lane = threadIdx.x & 63;
    const DiffuseReg R = diffuse_reg(P.d_reg, P.stats, npix);
    // Which pixel a thread takes: 256 consecutive pixels of a row per workgroup, or (P.tiled) a tile of the frame per workgroup and a block of it per
    // wave.  In row order a 16 x 512^2 head batch leaves 16 % of the workgroups without a covered pixel, 40 % of all WAVES are background waves inside
    // mixed workgroups -- resident until their workgroup's epilogue -- and the waves that work are 76 % full (profiles/r06_coverage_stats.txt); as tiles
    // the mixed workgroups are the silhouette only.  Which is faster depends on the frame size: see the launch.
    unsigned pi, wg_valid;
    bool valid;
    if (P.cov_list) {
        // thread k takes the k-th covered pixel (pixel order: a wave is 64 consecutive covered pixels of a row, or the end of one row's run and the
        // start of the next): no background lane, no background wave; workgroups past the list leave at once
        const unsigned n_cov = npix - (unsigned)*P.n_bg, k = blockIdx.x * DB_T + threadIdx.x;
        valid = k < n_cov;
        pi = valid ? P.cov_list[k] : 0u;
        wg_valid = 0u;
        if (blockIdx.x == 0 && threadIdx.x == 0 && P.part && R.on && n_cov < npix)      // the background's share of the lights regulariser: one count
            atomicAdd(&P.part[27], (float)(npix - n_cov));
    } else if (P.tiled) {
        // tile shapes (workgroup / wave): 1 = 16 x 16 / 8 x 8, 2 = 64 x 4 / 64 x 1, 3 = 32 x 8 / 32 x 2, 4 = 16 x 16 / 16 x 4
        const unsigned tw = P.tiled == 2 ? 64u : (P.tiled == 3 ? 32u : 16u), th = 256u / tw;
        const unsigned tpf = (unsigned)P.tiles_x * P.tiles_y, b = blockIdx.x / tpf, tr = blockIdx.x - b * tpf;
        const unsigned ty = tr / (unsigned)P.tiles_x, tx = tr - ty * (unsigned)P.tiles_x;
        const unsigned wv = threadIdx.x >> 6, ln = (unsigned)lane;
        unsigned lx, ly;
        if (P.tiled == 1) { lx = (wv & 1u) * 8u + (ln & 7u); ly = (wv >> 1) * 8u + (ln >> 3); }
        else if (P.tiled == 2) { lx = ln; ly = wv; }
        else if (P.tiled == 3) { lx = ln & 31u; ly = wv * 2u + (ln >> 5); }
        else { lx = ln & 15u; ly = wv * 4u + (ln >> 4); }
        const unsigned px = tx * tw + lx, py = ty * th + ly;
        valid = px < (unsigned)P.W && py < (unsigned)P.H;
        pi = valid ? b * HW + py * (unsigned)P.W + px : 0u;
        wg_valid = min(tw, (unsigned)P.W - tx * tw) * min(th, (unsigned)P.H - ty * th);
    } else {
        pi = blockIdx.x * DB_T + threadIdx.x;
        valid = pi < npix;
        wg_valid = min((unsigned)DB_T, npix - min(npix, blockIdx.x * (unsigned)DB_T));
    }
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (valid) r = P.rast[pi];
    const int t = (int)r.w - 1;
    const bool cov = valid && t >= 0 && t < P.F;
    float gl[27];
#pragma unroll
    for (int i = 0; i < 27; i++) gl[i] = 0.f;
    if (valid && !cov && !P.skip_bg) {       // background: nothing flows (its colour is the detached target / a constant)
        float* da = P.d_albedo + 3 * (size_t)pi;
        da[0] = 0.f; da[1] = 0.f; da[2] = 0.f;
        if (P.tile_ids) P.tile_ids[pi] = (unsigned short)0xFFFF;
    }
    // a workgroup without a covered pixel (two thirds of them on a head frame) has nothing to reduce: it leaves here -- its background pixels'
    // share of the lights regulariser is one count -- instead of walking the epilogue's LDS rows and barriers with 27 zeros per lane
    if (__syncthreads_or(cov ? 1 : 0) == 0) {
        if (P.part && R.on && threadIdx.x == 0) {
            const unsigned n_bg = wg_valid;
            float* row = P.part + (size_t)(((unsigned)blockIdx.x + (unsigned)blockIdx.y * 7u + (unsigned)blockIdx.z * 13u) % DB_SLOTS) * DB_ROW;
            if (n_bg) atomicAdd(&row[27], (float)n_bg);
        }
        return;
    }
    const bool any = __ballot(cov) != 0ull;
    if (any) {
        int tb_tile = -1;
        float tb_g = 0.f;
        if (cov) {
            const unsigned b = pi / HW, rem = pi - b * HW;
            const unsigned py = rem / (unsigned)P.W, px = rem - py * (unsigned)P.W;
            const DeferredGrad o = deferred_pixel(P, R, s_l, s_c, pi, b, py, px, t, gl, tb_tile, tb_g);
            float* dn = P.d_normal + 3 * (size_t)pi;
            dn[0] = o.gn[0]; dn[1] = o.gn[1]; dn[2] = o.gn[2];
            P.d_texc[pi] = o.guv;
            P.d_texd[pi] = o.gda;
        }
        if (P.tb_counts) deferred_tile_histogram(P, tb_tile, tb_g, lane);
    }
    if (!P.part) return;
    deferred_lights_epilogue<DB_NW>(P, gl, red, any, R.on ? __popcll(__ballot(valid && !cov)) : 0, &s_nbg);
}

}  // namespace

// d_lights[i] += sum over the DB_SLOTS rows of part (+ the background pixels' regulariser term, one constant x their number): one wave
// per entry
static __global__ __launch_bounds__(64) void vhap_deferred_lights_reduce_kernel(const float* __restrict__ part, const float* __restrict__ lights,
                                                                         const float* __restrict__ sh_const, const float* __restrict__ d_reg,
                                                                         const unsigned* __restrict__ stats, unsigned npix,
                                                                         float* __restrict__ d_lights) {
    const int i = blockIdx.x;
    float s = vhap_wave_sum_dpp(part[(size_t)threadIdx.x * DB_ROW + i]);
    const float n_bg = vhap_wave_sum_dpp(part[(size_t)threadIdx.x * DB_ROW + 27]);
    if (threadIdx.x != 0) return;
    const DiffuseReg R = diffuse_reg(d_reg, stats, npix);
    if (R.on && n_bg != 0.f) {               // background pixels: normal 0 -> the same basis / diffuse colour for all of them
        SH9 bsh;
        float x, y, z, inv, d[3], gr[3];
        sh_diffuse(0.f, 0.f, 0.f, sh_const, lights, bsh, x, y, z, inv, d);
        diffuse_reg_grad(R, d, gr);
        s += n_bg * (bsh.v[i / 3] * gr[i % 3]);
    }
    d_lights[i] += s;
}

static int vhap_fill_deferred_params(DeferredParams& P, const float* pos, const int32_t* tri, const float* vnormal, const float* uv,
                              const int32_t* tri_uv, const float* tex, const float* mips, int Ht, int Wt, const float* lights,
                              const float* sh_const, const float* rast, const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                              const float* d_sum, const float* d_delta, const float* keep, const float* d_reg, const float* stats, int B, int V, int VT, int F, int H,
                              int W, float* texc, float* texd, float* d_albedo, float* d_lights, float* work, size_t work_floats,
                              void* texbin_work, uint16_t* tile_ids) {
    if (!pos || !tri || !vnormal || !uv || !tri_uv || !tex || !lights || !sh_const || !rast || !texc || !texd || !d_albedo) return VHAP_E_NULLPTR;
    if (!d_rgba && (!pred_rgba || !gt_nchw || !d_sum)) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || VT <= 0 || F <= 0 || H <= 0 || W <= 0 || Ht <= 0 || Wt <= 0 || (long long)B * H * W >= (1ll << 31))
        return VHAP_E_BADDIM;
    if ((d_lights || work) && (!work || work_floats < (size_t)DB_SLOTS * DB_ROW)) return VHAP_E_WORKSPACE;
    P.pos = reinterpret_cast<const float4*>(pos); P.tri = tri; P.vnormal = vnormal; P.uv = reinterpret_cast<const float2*>(uv);
    P.tri_uv = tri_uv; P.tex = tex; P.mips = mips; P.D = make_desc(1, Ht, Wt, 3);
    if (P.D.L > 0 && !mips) return VHAP_E_NULLPTR;
    P.lights = lights; P.sh_const = sh_const; P.rast = reinterpret_cast<const float4*>(rast);
    P.d_rgba = reinterpret_cast<const float4*>(d_rgba); P.pred = reinterpret_cast<const float4*>(pred_rgba); P.gt = gt_nchw; P.d_sum = d_sum;
    P.d_delta = reinterpret_cast<const float4*>(d_delta);
    P.keep = keep; P.d_reg = d_reg; P.stats = reinterpret_cast<const unsigned*>(stats);
    P.B = B; P.V = V; P.F = F; P.H = H; P.W = W;
    P.xs = 2.0f / (float)W; P.xo = 1.0f / (float)W - 1.0f; P.ys = 2.0f / (float)H; P.yo = 1.0f / (float)H - 1.0f;
    P.texc = reinterpret_cast<float2*>(texc); P.texd = reinterpret_cast<float4*>(texd); P.d_albedo = d_albedo;
    P.part = work;                              // (d_lights == NULL with a work table: partial sums only, vhap_deferred_lights_reduce later)
    P.tile_ids = tile_ids;
    P.NT = texbin_nt(Ht, Wt);
    if (texbin_work) {                          // layout of vhap_texture_grad_binned's workspace: counts, then max|g| bits
        const TexBinWs l = texbin_layout(1);
        P.tb_counts = reinterpret_cast<unsigned*>(static_cast<char*>(texbin_work) + l.counts);
        P.tb_max = reinterpret_cast<unsigned*>(static_cast<char*>(texbin_work) + l.tilemax);
        P.NT = texbin_nt(Ht, Wt);
    }
    return VHAP_OK;
}

extern "C" size_t vhap_deferred_shade_bwd_work_floats(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    return (size_t)DB_SLOTS * DB_ROW;
}

static int deferred_shade_bwd_run(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                       const float* tex, const float* mips, int Ht, int Wt, const float* lights, const float* sh_const,
                                       const float* rast, const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                                       const float* d_sum, const float* d_delta, const float* keep, const float* d_reg, const float* stats,
                                       int B, int V, int VT, int F, int H, int W, float* texc, float* texd, float* d_albedo,
                                       float* d_normal, float* d_texc, float* d_texd, float* d_lights, float* work, size_t work_floats,
                                       void* texbin_work, uint16_t* tile_ids, const uint32_t* covered_list, const int32_t* n_background,
                                       int call_flags, vhap_stream_t stream);

extern "C" int vhap_deferred_shade_bwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                       const float* tex, const float* mips, int Ht, int Wt, const float* lights, const float* sh_const,
                                       const float* rast, const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                                       const float* d_sum, const float* d_delta, const float* keep, const float* d_reg, const float* stats,
                                       int B, int V, int VT, int F, int H, int W, float* texc, float* texd, float* d_albedo,
                                       float* d_normal, float* d_texc, float* d_texd, float* d_lights, float* work, size_t work_floats,
                                       void* texbin_work, uint16_t* tile_ids, int call_flags, vhap_stream_t stream) {
    return deferred_shade_bwd_run(pos, tri, vnormal, uv, tri_uv, tex, mips, Ht, Wt, lights, sh_const, rast, d_rgba, pred_rgba, gt_nchw, d_sum, d_delta,
                                  keep, d_reg, stats, B, V, VT, F, H, W, texc, texd, d_albedo, d_normal, d_texc, d_texd, d_lights, work, work_floats,
                                  texbin_work, tile_ids, nullptr, nullptr, call_flags, stream);
}

extern "C" int vhap_deferred_shade_bwd_list(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                            const float* tex, const float* mips, int Ht, int Wt, const float* lights, const float* sh_const,
                                            const float* rast, const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                                            const float* d_sum, const float* d_delta, const float* keep, const float* d_reg, const float* stats,
                                            int B, int V, int VT, int F, int H, int W, float* texc, float* texd, float* d_albedo,
                                            float* d_normal, float* d_texc, float* d_texd, float* d_lights, float* work, size_t work_floats,
                                            void* texbin_work, uint16_t* tile_ids, const uint32_t* covered_list, const int32_t* n_background,
                                            int call_flags, vhap_stream_t stream) {
    if (!covered_list || !n_background) return VHAP_E_NULLPTR;
    if (!(call_flags & VHAP_CALL_SKIP_BG_GRAD) || tile_ids) return VHAP_E_BADDIM;       // (a pass over the covered pixels writes nothing for the background)
    return deferred_shade_bwd_run(pos, tri, vnormal, uv, tri_uv, tex, mips, Ht, Wt, lights, sh_const, rast, d_rgba, pred_rgba, gt_nchw, d_sum, d_delta,
                                  keep, d_reg, stats, B, V, VT, F, H, W, texc, texd, d_albedo, d_normal, d_texc, d_texd, d_lights, work, work_floats,
                                  texbin_work, tile_ids, covered_list, n_background, call_flags, stream);
}

static int deferred_shade_bwd_run(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                       const float* tex, const float* mips, int Ht, int Wt, const float* lights, const float* sh_const,
                                       const float* rast, const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                                       const float* d_sum, const float* d_delta, const float* keep, const float* d_reg, const float* stats,
                                       int B, int V, int VT, int F, int H, int W, float* texc, float* texd, float* d_albedo,
                                       float* d_normal, float* d_texc, float* d_texd, float* d_lights, float* work, size_t work_floats,
                                       void* texbin_work, uint16_t* tile_ids, const uint32_t* covered_list, const int32_t* n_background,
                                       int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!d_normal || !d_texc || !d_texd) return VHAP_E_NULLPTR;
    DeferredParams P{};
    if (int e = vhap_fill_deferred_params(P, pos, tri, vnormal, uv, tri_uv, tex, mips, Ht, Wt, lights, sh_const, rast, d_rgba, pred_rgba, gt_nchw,
                                          d_sum, d_delta, keep, d_reg, stats, B, V, VT, F, H, W, texc, texd, d_albedo, d_lights, work, work_floats,
                                          texbin_work, tile_ids))
        return e;
    P.d_normal = d_normal; P.d_texc = reinterpret_cast<float2*>(d_texc); P.d_texd = reinterpret_cast<float4*>(d_texd);
    P.delta_unscaled = (call_flags & VHAP_CALL_DELTA_UNSCALED) ? 1 : 0;
    P.skip_bg = ((call_flags & VHAP_CALL_SKIP_BG_GRAD) && !tile_ids) ? 1 : 0;
    P.cov_list = covered_list;
    P.n_bg = n_background;
    const long long npix = (long long)B * H * W;
    // Row order or tiles, by measurement (the pass alone, us; profiles/r06_call31_shade_bwd_shapes.txt, r06_call32_shade_bwd_probe_sizes.txt):
    //   16 x 512^2: row 149, tiles 168 | 16 x 504^2: 139 / 170 | 16 x 768x512 (W = 512): 249 / 263 | 16 x 520^2: 149 / 147 | 32 x 384^2: 171 / 146 |
    //   16 x 640^2: 216 / 187 | 16 x 802x550: 210 / 191 | 16 x 768^2: 295 / 265 | 8 x 1000^2: 245 / 233 | 8 x 1024^2: 331 / 251
    // (64x4/64x1, 32x8/32x2 and 16x16/16x4 never beat 16x16/8x8).  Tiles win everywhere but in a band of widths at and just below 512, where the
    // 256-pixel row segments happen to cut a centred head into two workgroups with one idle wave each (and where every tiled shape is ~14 % slower
    // than 8 pixels further on: a power-of-two row pitch) -- frames 449 .. 512 pixels wide stay in row order, all others run as tiles.
    // debug flags: 16777216 row order, 268435456 tiles 16x16/8x8, 33554432 / 67108864 / 134217728 shapes 2 / 3 / 4, whatever the width
    P.tiled = (vhap_g_debug_flags & 16777216) ? 0 : (vhap_g_debug_flags & 268435456) ? 1 : (vhap_g_debug_flags & 33554432) ? 2 :
              (vhap_g_debug_flags & 67108864) ? 3 : (vhap_g_debug_flags & 134217728) ? 4 : ((W > 448 && W <= 512) ? 0 : 1);
    const int tw = P.tiled == 2 ? 64 : (P.tiled == 3 ? 32 : 16), th = 256 / tw;
    P.tiles_x = (W + tw - 1) / tw;
    P.tiles_y = (H + th - 1) / th;
    const int blocks = (P.tiled && !P.cov_list) ? B * P.tiles_x * P.tiles_y : (int)((npix + DB_T - 1) / DB_T);   // (a list: as many workgroups as a full list needs)
    hipStream_t st = vhap_stream(stream);
    deferred_shade_bwd_kernel<<<blocks, DB_T, 0, st>>>(P);
    VHAP_LAUNCH_CHECK();
    if (d_lights) {
        vhap_deferred_lights_reduce_kernel<<<27, DB_SLOTS, 0, st>>>(work, lights, sh_const, d_reg, reinterpret_cast<const unsigned*>(stats),
                                                                    (unsigned)npix, d_lights);
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

extern "C" int vhap_deferred_lights_reduce(const float* work, const float* lights, const float* sh_const, const float* d_reg, const float* stats,
                                           int B, int H, int W, float* d_lights, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!work || !lights || !sh_const || !d_lights) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    vhap_deferred_lights_reduce_kernel<<<27, DB_SLOTS, 0, vhap_stream(stream)>>>(work, lights, sh_const, d_reg, reinterpret_cast<const unsigned*>(stats),
                                                                                 (unsigned)((long long)B * H * W), d_lights);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
