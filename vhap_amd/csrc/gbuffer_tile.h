// Tile back end of the G-buffer backward, shared by vhap_gbuffer_bwd (interp.hip) and the pass fused with the deferred-shading backward
// (deferred.hip).  A workgroup owns a 16x16 pixel tile of one frame; every covered pixel chains the gradient of its barycentrics /
// barycentric derivatives to the three clip-space vertices of ITS triangle (gb_chain) and adds the 18 vertex contributions into an LDS
// hash table keyed by vertex id (64-bit fixed point: integer LDS atomics); the table is flushed with one global atomic per touched
// vertex component (gb_tile_commit), ~20x fewer than a plain per-pixel kernel.
#pragma once
#include "common.h"

namespace {

constexpr int GT = 16;          // tile edge
constexpr int GSLOT = 256;      // hash slots (vertices) per tile: a tile touches a few dozen; 512 slots (28 KB of LDS) cost 20 % in occupancy
constexpr unsigned GEMPTY = 0xffffffffu;

struct GbTile {
    unsigned keys[GSLOT];
    unsigned long long vals[GSLOT * 6];     // [0..2] = d_pos x, y, w ; [3..5] = d_vnormal: 64-bit fixed point (see texture.hip)
    unsigned smax[2];
};

__device__ __forceinline__ void gb_tile_init(GbTile& S) {
    const int tid = threadIdx.x;
    for (int i = tid; i < GSLOT; i += GT * GT) S.keys[i] = GEMPTY;
    for (int i = tid; i < GSLOT * 6; i += GT * GT) S.vals[i] = 0ull;
    if (tid < 2) S.smax[tid] = 0u;
}

// (g0, g1): gradient w.r.t. the barycentrics (u, v); gd: w.r.t. their screen-space derivatives (du/dX, du/dY, dv/dX, dv/dY);
// adds the clip-space position gradients of the three vertices into acc[0..8] (x, y, w per vertex)
__device__ __forceinline__ void gb_chain(const float4 p0, const float4 p1, const float4 p2, float b0, float b1, int px, int py, int H, int W,
                                         float g0, float g1, const float4 gd, float (&acc)[18]) {
    const float xs = 2.0f / (float)W, xo = 1.0f / (float)W - 1.0f;
    const float ys = 2.0f / (float)H, yo = 1.0f / (float)H - 1.0f;
    const float X0 = p2.y * p1.w - p1.y * p2.w, Y0 = p1.x * p2.w - p2.x * p1.w;
    const float X1 = p0.y * p2.w - p2.y * p0.w, Y1 = p2.x * p0.w - p0.x * p2.w;
    const float X2 = p1.y * p0.w - p0.y * p1.w, Y2 = p0.x * p1.w - p1.x * p0.w;
    const float Tx = X0 + X1 + X2, Ty = Y0 + Y1 + Y2;
    const float fx = xs * (float)px + xo, fy = ys * (float)py + yo;
    const float p0x = p0.x - fx * p0.w, p0y = p0.y - fy * p0.w;
    const float p1x = p1.x - fx * p1.w, p1y = p1.y - fy * p1.w;
    const float p2x = p2.x - fx * p2.w, p2y = p2.y - fy * p2.w;
    const float a0 = p1x * p2y - p1y * p2x, a1 = p2x * p0y - p2y * p0x, a2 = p0x * p1y - p0y * p1x;
    const float at = a0 + a1 + a2;
    if (fabsf(at) > 0.0f) {
        const float iw = 1.0f / at;
        const float r0 = a0 * iw, r1 = a1 * iw;
        float G0 = g0 + xs * iw * Tx * gd.x + ys * iw * Ty * gd.y;
        float G1 = g1 + xs * iw * Tx * gd.z + ys * iw * Ty * gd.w;
        if (!(r0 >= 0.0f && r0 <= 1.0f)) G0 = 0.0f;
        if (!(r1 >= 0.0f && r1 <= 1.0f)) G1 = 0.0f;
        const float giw = xs * (b0 * Tx - X0) * gd.x + ys * (b0 * Ty - Y0) * gd.y + xs * (b1 * Tx - X1) * gd.z +
                          ys * (b1 * Ty - Y1) * gd.w;
        const float sg = G0 * r0 + G1 * r1;
        const float gat = -iw * iw * giw;
        const float ga0 = (G0 - sg) * iw + gat, ga1 = (G1 - sg) * iw + gat, ga2 = (-sg) * iw + gat;
        const float gp0x = -p2y * ga1 + p1y * ga2, gp0y = p2x * ga1 - p1x * ga2;
        const float gp1x = p2y * ga0 - p0y * ga2, gp1y = -p2x * ga0 + p0x * ga2;
        const float gp2x = -p1y * ga0 + p0y * ga1, gp2y = p1x * ga0 - p0x * ga1;
        acc[0] += gp0x; acc[1] += gp0y; acc[2] += -fx * gp0x - fy * gp0y;
        acc[3] += gp1x; acc[4] += gp1y; acc[5] += -fx * gp1x - fy * gp1y;
        acc[6] += gp2x; acc[7] += gp2y; acc[8] += -fx * gp2x - fy * gp2y;
        const float cx = xs * iw, cy = ys * iw;
        const float sxg = b0 * gd.x + b1 * gd.z, syg = b0 * gd.y + b1 * gd.w;
        const float gX0 = cx * (sxg - gd.x), gX1 = cx * (sxg - gd.z), gX2 = cx * sxg;
        const float gY0 = cy * (syg - gd.y), gY1 = cy * (syg - gd.w), gY2 = cy * syg;
        acc[7] += p1.w * gX0; acc[5] += p2.y * gX0; acc[4] -= p2.w * gX0; acc[8] -= p1.y * gX0;
        acc[1] += p2.w * gX1; acc[8] += p0.y * gX1; acc[7] -= p0.w * gX1; acc[2] -= p2.y * gX1;
        acc[4] += p0.w * gX2; acc[2] += p1.y * gX2; acc[1] -= p1.w * gX2; acc[5] -= p0.y * gX2;
        acc[3] += p2.w * gY0; acc[8] += p1.x * gY0; acc[6] -= p1.w * gY0; acc[5] -= p2.x * gY0;
        acc[6] += p0.w * gY1; acc[2] += p2.x * gY1; acc[0] -= p2.w * gY1; acc[8] -= p0.x * gY1;
        acc[0] += p1.w * gY2; acc[5] += p0.x * gY2; acc[3] -= p0.w * gY2; acc[2] -= p1.x * gY2;
    }
}

// workgroup-wide: scales, LDS hash insert of this thread's 3 vertices, flush.  Contains barriers: call from ALL threads of the tile.
__device__ __forceinline__ void gb_tile_commit(GbTile& S, const float (&acc)[18], bool have, int i0, int i1, int i2, int b, int V,
                                               float* __restrict__ d_pos, float* __restrict__ d_vnormal, int dbg) {
    const int tid = threadIdx.x;
    unsigned* keys = S.keys;
    unsigned long long* vals = S.vals;
    unsigned* smax = S.smax;
    // per-tile power-of-two scales (positions / normals) from the largest contribution
    {
        float mp = 0.f, mn = 0.f;
#pragma unroll
        for (int k = 0; k < 9; k++) { mp = fmaxf(mp, fabsf(acc[k])); mn = fmaxf(mn, fabsf(acc[9 + k])); }
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) { mp = fmaxf(mp, __shfl_xor(mp, o, 64)); mn = fmaxf(mn, __shfl_xor(mn, o, 64)); }
        if ((tid & 63) == 0) { atomicMax(&smax[0], __float_as_uint(mp)); atomicMax(&smax[1], __float_as_uint(mn)); }
    }
    __syncthreads();
    int exp_p = 0, exp_n = 0;
    (void)frexpf(__uint_as_float(smax[0]), &exp_p);
    (void)frexpf(__uint_as_float(smax[1]), &exp_n);
    const int shp = min(max(40 - exp_p, -100), 100), shn = min(max(40 - exp_n, -100), 100);
    const float sc_p = ldexpf(1.0f, shp), sc_n = ldexpf(1.0f, shn), isc_p = ldexpf(1.0f, -shp), isc_n = ldexpf(1.0f, -shn);
    // The 18 contributions of a pixel are MERGED across neighbouring lanes that hold the same triangle before anything touches the table.
    // Why: the table's LDS atomics are the kernel's bottleneck, not its gathers -- a wave's 64 lanes hit ~10 distinct vertex slots,
    // ds_add_u64 costs ~2 cycles per lane that shares an address (tools/ubench/lds_atomics.hip), there are 18 per covered pixel, and the CU
    // has ONE LDS pipe for its four SIMDs: 25 M lane-atomics per 16 x 512^2 step = ~80 us of the kernel's 117.  A triangle of a head frame
    // covers ~17 pixels, runs of ~4 along a row: four butterfly levels over the 16 lanes of a tile row (lane ^ 1, ^ 2, then + 4, + 8 -- DPP
    // moves on the VALU, which is per SIMD) leave one lane per run to do the atomics.  The partial sums are fp32 in a FIXED tree order
    // (deterministic; the fixed-point form of the 18 values would cost 36 more registers and three of the kernel's seven waves per SIMD).
    float a[18];
#pragma unroll
    for (int k = 0; k < 18; k++) a[k] = have ? acc[k] : 0.f;
    bool alive = have;
    if (!(dbg & 1024)) {                 // (debug flag 1024: A/B without the merge)
        const int lane = tid & 63;
        // One level: `receiver` lanes absorb their partner (the lane `up` reads from), `donor` lanes (the others; their partner is the lane
        // `down` reads from) retire if absorbed.  Every DPP move is executed by ALL lanes -- a DPP read of a lane that EXEC has switched
        // off returns nothing useful on gfx9, so no move may sit inside a divergent branch -- and each side selects what it needs.
        auto level = [&](auto up, auto down, bool receiver, bool symmetric) {
            const int a_u = up(alive ? 1 : 0), k0_u = up(i0), k1_u = up(i1), k2_u = up(i2);
            int a_p = a_u, k0_p = k0_u, k1_p = k1_u, k2_p = k2_u;
            if (!symmetric) {
                const int a_d = down(alive ? 1 : 0), k0_d = down(i0), k1_d = down(i1), k2_d = down(i2);
                if (!receiver) { a_p = a_d; k0_p = k0_d; k1_p = k1_d; k2_p = k2_d; }
            }
            const bool same = alive && a_p != 0 && k0_p == i0 && k1_p == i1 && k2_p == i2;      // (evaluated identically on both sides of a pair)
#pragma unroll
            for (int k = 0; k < 18; k++) {
                const float pv = __int_as_float(up(__float_as_int(a[k])));
                if (same && receiver) a[k] += pv;
            }
            if (same && !receiver) alive = false;
        };
        auto x1 = [](int x) { return __builtin_amdgcn_mov_dpp(x, 0xB1, 0xF, 0xF, true); };      // quad_perm [1,0,3,2]: lane ^ 1
        auto x2 = [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x4E, 0xF, 0xF, true); };      // quad_perm [2,3,0,1]: lane ^ 2
        auto l4 = [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x104, 0xF, 0xF, true); };     // row_shl:4: lane + 4 of the row of 16 (0 past its end)
        auto r4 = [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x114, 0xF, 0xF, true); };     // row_shr:4: lane - 4
        level(x1, x1, !(lane & 1), true);
        level(x2, x2, !(lane & 2), true);
        level(l4, r4, !(lane & 4), false);
        auto l8 = [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x108, 0xF, 0xF, true); };     // row_shl:8 / row_shr:8: the two halves of a tile row
        auto r8 = [](int x) { return __builtin_amdgcn_mov_dpp(x, 0x118, 0xF, 0xF, true); };     // (runs longer than 8 pixels: the 1024^2 frames)
        level(l8, r8, !(lane & 8), false);
    }
    if (alive && !(dbg & 256)) {
        {
            // three vertices -> LDS table (bounded probing; overflow goes straight to global memory)
#pragma unroll
            for (int vtx = 0; vtx < 3; vtx++) {
                const int vi = vtx == 0 ? i0 : (vtx == 1 ? i1 : i2);
                unsigned slot = ((unsigned)vi * 2654435761u) >> 24;     // 8 bits
                bool done = false;
#pragma unroll 1
                for (int probe = 0; probe < 8 && !done; probe++) {
                    const unsigned prev = atomicCAS(&keys[slot], GEMPTY, (unsigned)vi);
                    if (prev == GEMPTY || prev == (unsigned)vi) {
#pragma unroll
                        for (int c = 0; c < 3; c++) {
                            const float vp = a[3 * vtx + c], vn = a[9 + 3 * vtx + c];
                            if (vp != 0.f) atomicAdd(&vals[slot * 6 + c], (unsigned long long)__float2ll_rn(vp * sc_p));
                            if (vn != 0.f) atomicAdd(&vals[slot * 6 + 3 + c], (unsigned long long)__float2ll_rn(vn * sc_n));
                        }
                        done = true;
                    } else {
                        slot = (slot + 1) & (GSLOT - 1);
                    }
                }
                if (!done) {
#pragma unroll
                    for (int c = 0; c < 3; c++) {
                        const float vp = a[3 * vtx + c], vn = a[9 + 3 * vtx + c];
                        if (d_pos && vp != 0.f) atomicAdd(&d_pos[((size_t)b * V + vi) * 4 + (c == 2 ? 3 : c)], vp);
                        if (d_vnormal && vn != 0.f) atomicAdd(&d_vnormal[((size_t)b * V + vi) * 3 + c], vn);
                    }
                }
            }
        }
    }
    __syncthreads();
    if (dbg & 128) return;
    for (int sidx = tid; sidx < GSLOT; sidx += GT * GT) {
        const unsigned vi = keys[sidx];
        if (vi == GEMPTY) continue;
#pragma unroll
        for (int c = 0; c < 3; c++) {
            const long long qp = (long long)vals[sidx * 6 + c], qn = (long long)vals[sidx * 6 + 3 + c];
            if (d_pos && qp != 0) atomicAdd(&d_pos[((size_t)b * V + vi) * 4 + (c == 2 ? 3 : c)], (float)qp * isc_p);
            if (d_vnormal && qn != 0) atomicAdd(&d_vnormal[((size_t)b * V + vi) * 3 + c], (float)qn * isc_n);
        }
    }
}

}  // namespace
