// Analytic silhouette antialiasing (forward / backward) for gfx950.
//
// Replaces dr.antialias(color, rast, pos, tri) (vhap/util/render_nvdiffrast.py:465), the op that
// carries silhouette gradients from the image to the geometry.  nvdiffrast rebuilds an edge hash of
// the mesh on every call because VHAP passes no topology hash; the FLAME topology is fixed, so the
// edge -> opposite-vertex table `opp` is built once on the host (vhap_amd/topology.py).
//
// Per pixel pair (p, p+x) and (p, p+y) with different triangle ids (restated in
// oracle/torch_ref.py antialias()):
//   - take the triangle of the nearer pixel (smaller z/w; a background pixel never wins), express
//     its vertices in pixel units relative to that pixel's centre
//   - an edge is a silhouette candidate if its opposite vertex lies on the same side as the
//     triangle's own third vertex (or the edge is a mesh boundary)
//   - among the edges straddling the centre-to-centre segment pick the one that crosses it farthest
//     towards the neighbour; it must be a silhouette edge and steeper than 45 degrees
//   - crossing position dc in (-1/16, 1+1/16), clamped to [0,1]; alpha = ds*(0.5 - dc);
//     out[alpha > 0 ? p0 : p1] += alpha * (color[p1] - color[p0])
// The forward appends one work item per blended pair; the backward replays them.
#include <type_traits>

#include "common.h"
#include "aa_items.h"

namespace {

struct Geo {
    bool ok;
    int di;        // selected edge
    float ds;      // +1: triangle of pixel 0, -1: triangle of pixel 1
    float dc_raw;  // unclamped crossing position
    float xa, ya, xb, yb;  // selected edge end points in the (possibly flipped) pixel frame
    int va, vb;            // their vertex indices
};

// d = 0: horizontal pair (px,py)-(px+1,py); d = 1: vertical pair (px,py)-(px,py+1)
__device__ __forceinline__ Geo analyse(const float4* __restrict__ P, const int* __restrict__ tri, const int* __restrict__ opp,
                                       int t0, int t1, float z0, float z1, int px, int py, int d, int H, int W) {
    Geo g;
    g.ok = false;
    int t = t0 >= 0 ? t0 : t1;
    if (t0 >= 0 && t1 >= 0) t = z0 < z1 ? t0 : t1;
    const bool use1 = t == t1;
    const int cpx = use1 ? px + (d == 0 ? 1 : 0) : px;
    const int cpy = use1 ? py + (d == 1 ? 1 : 0) : py;
    const float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
    const float fx = (float)cpx + 0.5f - xh, fy = (float)cpy + 0.5f - yh;
    int vi[3], oi[3];
    float x[3], y[3], ox[3], oy[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        vi[k] = tri[3 * t + k];
        oi[k] = opp[3 * t + k];
        if (oi[k] < 0) oi[k] = vi[k];  // boundary edge: the vertex itself (always a silhouette)
    }
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const float4 p = P[vi[k]], o = P[oi[k]];
        x[k] = p.x / p.w * xh - fx;
        y[k] = p.y / p.w * yh - fy;
        ox[k] = o.x / o.w * xh - fx;
        oy[k] = o.y / o.w * yh - fy;
    }
    const float bb = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
    const float a0 = (x[1] - ox[0]) * (y[2] - oy[0]) - (x[2] - ox[0]) * (y[1] - oy[0]);
    const float a1 = (x[2] - ox[1]) * (y[0] - oy[1]) - (x[0] - ox[1]) * (y[2] - oy[1]);
    const float a2 = (x[0] - ox[2]) * (y[1] - oy[2]) - (x[1] - ox[2]) * (y[0] - oy[2]);
    const bool nb = bb < 0.f;
    const bool sil[3] = {(a0 < 0.f) == nb, (a1 < 0.f) == nb, (a2 < 0.f) == nb};
    if (d == 1) {
#pragma unroll
        for (int k = 0; k < 3; k++) { const float tmp = x[k]; x[k] = y[k]; y[k] = tmp; }
    }
    const float ds = use1 ? -1.0f : 1.0f;
    float best = -INFINITY;
    int di = -1;
#pragma unroll
    for (int k = 0; k < 3; k++) {   // edge k joins vertex (k+1)%3 -> (k+2)%3
        const int a = (k + 1) % 3, b = (k + 2) % 3;
        const bool straddle = (y[a] < 0.f) != (y[b] < 0.f);
        if (straddle) {
            const float dx = x[b] - x[a], dy = y[b] - y[a];
            const float dc = ds * (x[a] * dy - y[a] * dx) / dy;
            if (dc > best) { best = dc; di = k; }   // first maximum wins (lowest edge index on ties)
        }
    }
    if (di < 0) return g;
    const int a = (di + 1) % 3, b = (di + 2) % 3;
    // (runtime-indexed small arrays: select explicitly to stay in registers)
    const float xa = a == 0 ? x[0] : (a == 1 ? x[1] : x[2]), ya = a == 0 ? y[0] : (a == 1 ? y[1] : y[2]);
    const float xb = b == 0 ? x[0] : (b == 1 ? x[1] : x[2]), yb = b == 0 ? y[0] : (b == 1 ? y[1] : y[2]);
    const bool s = di == 0 ? sil[0] : (di == 1 ? sil[1] : sil[2]);
    if (!s) return g;
    if (!(fabsf(yb - ya) >= fabsf(xb - xa))) return g;
    const float eps = 0.0625f;
    if (!(best > -eps && best < 1.0f + eps)) return g;
    g.ok = true;
    g.di = di;
    g.ds = ds;
    g.dc_raw = best;
    g.xa = xa; g.ya = ya; g.xb = xb; g.yb = yb;
    g.va = a == 0 ? vi[0] : (a == 1 ? vi[1] : vi[2]);
    g.vb = b == 0 ? vi[0] : (b == 1 ? vi[1] : vi[2]);
    return g;
}

// Pre-pass: which triangles of which frame have at least one silhouette edge.  The silhouette test of analyse() is a property
// of (frame, triangle, edge) -- cross products are translation invariant -- and fewer than 10 % of the triangles of a closed
// surface have one, while about half of all neighbouring pixel pairs straddle two different triangles (the mesh is ~9
// pixels per triangle at 512^2): one byte per pair decides whether the full edge analysis (8 vertex gathers) is needed at all.
__global__ __launch_bounds__(256) void aa_silhouette_kernel(const float4* __restrict__ pos, const int* __restrict__ tri,
                                                            const int* __restrict__ opp, int B, int V, int F, int H, int W,
                                                            unsigned char* __restrict__ sil, int* __restrict__ work_header) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < 4) work_header[i] = 0;          // item / candidate counters of this call (the detect pass comes next in stream order)
    if (i >= B * F) return;
    const int b = i / F, t = i - b * F;
    const float4* P = pos + (size_t)b * V;
    const float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
    float x[3], y[3], ox[3], oy[3];
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const int vi = tri[3 * t + k];
        int oi = opp[3 * t + k];
        if (oi < 0) oi = vi;
        const float4 p = P[vi], o = P[oi];
        x[k] = p.x / p.w * xh; y[k] = p.y / p.w * yh;
        ox[k] = o.x / o.w * xh; oy[k] = o.y / o.w * yh;
    }
    const float bb = (x[1] - x[0]) * (y[2] - y[0]) - (x[2] - x[0]) * (y[1] - y[0]);
    const float a0 = (x[1] - ox[0]) * (y[2] - oy[0]) - (x[2] - ox[0]) * (y[1] - oy[0]);
    const float a1 = (x[2] - ox[1]) * (y[0] - oy[1]) - (x[0] - ox[1]) * (y[2] - oy[1]);
    const float a2 = (x[0] - ox[2]) * (y[1] - oy[2]) - (x[1] - ox[2]) * (y[0] - oy[2]);
    // conservative: an area within rounding noise of zero counts as "maybe" (the per-pair analysis then decides exactly)
    const float tol = 1e-4f * (fabsf(bb) + 1e-12f);
    const bool nb = bb < 0.f;
    auto maybe = [&](float a) { return fabsf(a) <= tol + 1e-6f * fabsf(a) || ((a < 0.f) == nb); };
    sil[i] = (unsigned char)((maybe(a0) ? 1 : 0) | (maybe(a1) ? 2 : 0) | (maybe(a2) ? 4 : 0) | (fabsf(bb) <= 1e-12f ? 7 : 0));
}

// work[0] = item count, work[1] = candidate count; items (4 ints each) start at work[4]:
//   {pixel index of p0, d | use1 << 1 | edge << 2, alpha bits, frame}
// Forward in two passes so that the expensive edge analysis runs on DENSE waves: a candidate pair is rare (a few per cent of
// the pixels), but a wave pays for the analysis if ANY of its 64 lanes needs it.
//   detect: stream over the pixels -- out = color, and every neighbouring pair whose front triangle has a silhouette edge is
//           appended to a compact candidate list (one atomic per wave);
//   blend:  one lane per candidate: analyse(), blend, append the work item for the backward.
// 256-thread workgroups (round 6; they were 1 024 threads with 32 KB of LDS): a 1 024-thread workgroup needs sixteen free wave slots on
// ONE CU at the same moment, and beside the disturbance's colour-pool kernels -- which is where this pass runs -- it waited for them to
// drain: 98 us in the step for 33 us of work, and the blend waits for it (profiles/r06_call8_step_timeline.txt).  A workgroup still owns
// DET_CHUNKS x 1 024 consecutive pixels and keeps its candidates in LDS until the list could overflow, so the returning atomic on the
// ONE global counter is as rare as before.
constexpr int DET_T = 256, DET_PPT = 4, DET_CHUNKS = 4;      // detect: 256 lanes x 4 pixels per chunk, 4 chunks per workgroup
constexpr int DET_LCAP = 2 * DET_T * DET_PPT * 2;            // LDS candidate slots (16 KB): room for two full chunks
template <int C>
__global__ __launch_bounds__(DET_T) void aa_detect_kernel(const float* __restrict__ color, const float4* __restrict__ rast,
                                                          const unsigned char* __restrict__ sil, int B, int H, int W, int F,
                                                          float* __restrict__ out, int* __restrict__ work, unsigned* __restrict__ cand,
                                                          int dbg) {
    __shared__ unsigned lcand[DET_LCAP];
    __shared__ int lcount, gbase;
    if (threadIdx.x == 0) lcount = 0;
    __syncthreads();
    const unsigned npix = (unsigned)B * H * W;             // < 2^30 (checked by the entry point): 32-bit index math
    const unsigned HW = (unsigned)H * W;
    const int lane = threadIdx.x & 63;
    auto flush = [&]() {                                   // (all threads; lcount settled by a barrier before the call, nobody adds to it meanwhile)
        const int n = lcount;
        __syncthreads();
        if (n != 0) {                                      // (uniform)
            if (threadIdx.x == 0) gbase = atomicAdd(&work[1], n);
            __syncthreads();
            for (int i = threadIdx.x; i < n; i += DET_T) cand[gbase + i] = lcand[i];
            __syncthreads();
            if (threadIdx.x == 0) lcount = 0;
            __syncthreads();
        }
    };
    for (int ch = 0; ch < DET_CHUNKS; ch++) {
        const unsigned base = ((unsigned)blockIdx.x * DET_CHUNKS + (unsigned)ch) * (DET_PPT * DET_T);
        if (base >= npix) break;                           // (uniform)
        // Three batches of loads for the thread's DET_PPT pixels -- the pixel's own rast word, its two neighbours' and its colour; then the
        // silhouette bytes of the pairs that need one -- instead of eight dependent round trips per pixel (rast -> neighbour -> flag, twice,
        // and the colour copy between them): this pass is 4 MB of flags and 130 MB of streaming, and it was 32 round trips long.
        float zw0[DET_PPT][2], zw1[DET_PPT][2][2];
        float colc[DET_PPT][C];
#pragma unroll
        for (int it = 0; it < DET_PPT; it++) {
            const unsigned pi = base + it * DET_T + threadIdx.x;
            const unsigned pc = pi < npix ? pi : npix - 1u;
            const unsigned px1 = min(pc + 1u, npix - 1u), py1 = min(pc + (unsigned)W, npix - 1u);
            zw0[it][0] = rast[pc].z; zw0[it][1] = rast[pc].w;
            zw1[it][0][0] = rast[px1].z; zw1[it][0][1] = rast[px1].w;
            zw1[it][1][0] = rast[py1].z; zw1[it][1][1] = rast[py1].w;
            if (out) {
                if constexpr (C == 4) {
                    const float4 v4 = reinterpret_cast<const float4*>(color)[pc];
                    colc[it][0] = v4.x; colc[it][1] = v4.y; colc[it][2] = v4.z; colc[it][3] = v4.w;
                } else {
#pragma unroll
                    for (int k = 0; k < C; k++) colc[it][k] = color[(size_t)pc * C + k];
                }
            }
        }
        int cnd[DET_PPT][2];
        unsigned sv[DET_PPT][2];
        const unsigned char* sil_q = sil ? sil : reinterpret_cast<const unsigned char*>(rast);      // (stand-in address: values unused)
#pragma unroll
        for (int it = 0; it < DET_PPT; it++) {
            const unsigned pi = base + it * DET_T + threadIdx.x;
            const bool live = pi < npix;
            const unsigned b = live ? pi / HW : 0u;
            const unsigned rem = pi - b * HW;
            const int py = (int)(rem / (unsigned)W), px = (int)(rem - (unsigned)py * W);
            const int t0 = (int)zw0[it][1] - 1;
#pragma unroll
            for (int d = 0; d < 2; d++) {
                bool c = false;
                int tf = 0;
                if (live && !(d == 0 ? px + 1 >= W : py + 1 >= H) && !(dbg & 1024)) {
                    const int t1 = (int)zw1[it][d][1] - 1;
                    if (t0 != t1 && t0 < F && t1 < F) {
                        // the triangle analyse() will pick: the nearer one; a background pixel never wins
                        tf = (t0 >= 0 && t1 >= 0) ? (zw0[it][0] < zw1[it][d][0] ? t0 : t1) : (t0 >= 0 ? t0 : t1);
                        c = true;
                    }
                }
                cnd[it][d] = c ? 1 : 0;
                sv[it][d] = (unsigned)sil_q[c && sil ? (size_t)b * F + tf : (size_t)0];
            }
        }
#pragma unroll
        for (int it = 0; it < DET_PPT; it++) {
            const unsigned pi = base + it * DET_T + threadIdx.x;
            const bool live = pi < npix;
            if (live && out) {
                if constexpr (C == 4) {
                    reinterpret_cast<float4*>(out)[pi] = make_float4(colc[it][0], colc[it][1], colc[it][2], colc[it][3]);
                } else {
#pragma unroll
                    for (int k = 0; k < C; k++) out[(size_t)pi * C + k] = colc[it][k];
                }
            }
#pragma unroll
            for (int d = 0; d < 2; d++) {
                const bool c = cnd[it][d] != 0 && (sil == nullptr || sv[it][d] != 0u);
                const unsigned long long m = __ballot(c);
                if (m == 0ull) continue;
                const int leader = __ffsll((long long)m) - 1;
                int lbase = 0;
                if (lane == leader) lbase = atomicAdd(&lcount, __popcll(m));
                lbase = __shfl(lbase, leader, 64);
                if (c) lcand[lbase + __popcll(m & ((1ull << lane) - 1ull))] = (pi << 1) | (unsigned)d;
            }
        }
        __syncthreads();
        // a chunk adds at most 2 x DET_T x DET_PPT candidates: hand the list over when the next chunk might not fit
        const int lc = lcount;
        __syncthreads();                                   // (every thread has read the SAME count before the next chunk's waves add to it)
        if (lc > DET_LCAP - 2 * DET_T * DET_PPT) flush();
    }
    flush();
}

template <int C>
__global__ __launch_bounds__(256) void aa_blend_kernel(const float* __restrict__ color, const float4* __restrict__ rast,
                                                       const float4* __restrict__ pos, const int* __restrict__ tri,
                                                       const int* __restrict__ opp, const unsigned* __restrict__ cand, int H, int W,
                                                       int V, int F, float* __restrict__ out, int* __restrict__ work, int dbg) {
    const int count = work[1];
    const unsigned HW = (unsigned)H * W;
    const int lane = threadIdx.x & 63;
    const int nwave_iter = (count + (int)(gridDim.x * 256) - 1) / (int)(gridDim.x * 256);
    for (int it = 0; it < nwave_iter; it++) {          // uniform trip count: every lane reaches the ballot
        const int i = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
        bool need = false;
        Geo g;
        float alpha = 0.f;
        unsigned pi = 0, pj = 0, b = 0;
        int d = 0;
        if (i < count) {
            const unsigned e = cand[i];
            pi = e >> 1; d = (int)(e & 1u);
            pj = pi + (d == 0 ? 1u : (unsigned)W);
            b = pi / HW;
            const unsigned rem = pi - b * HW;
            const int py = (int)(rem / (unsigned)W), px = (int)(rem - (unsigned)py * W);
            const float4 r0 = rast[pi], r1 = rast[pj];
            g = analyse(pos + (size_t)b * V, tri, opp, (int)r0.w - 1, (int)r1.w - 1, r0.z, r1.z, px, py, d, H, W);
            if (g.ok) {
                const float dc = fminf(fmaxf(g.dc_raw, 0.0f), 1.0f);
                alpha = g.ds * (0.5f - dc);
                need = true;
            }
        }
        const unsigned long long m = __ballot(need);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&work[0], __popcll(m));
        base = __shfl(base, leader, 64);
        if (need) {
            const float* c0 = color + (size_t)pi * C;
            const float* c1 = color + (size_t)pj * C;
            float* o = out + (size_t)(alpha > 0.0f ? pi : pj) * C;
#pragma unroll
            for (int k = 0; k < C; k++) if (!(dbg & 512)) atomicAdd(&o[k], alpha * (c1[k] - c0[k]));
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            const int4 item = make_int4((int)pi, d | (g.ds < 0.f ? 2 : 0) | (g.di << 2), __float_as_int(alpha), (int)b);
            reinterpret_cast<int4*>(work + 4)[slot] = item;
        }
    }
}

template <int C>
__global__ __launch_bounds__(256) void aa_bwd_kernel(const float* __restrict__ color, const float4* __restrict__ rast,
                                                     const float4* __restrict__ pos, const int* __restrict__ tri,
                                                     const int* __restrict__ opp, const float* __restrict__ d_out,
                                                     const int* __restrict__ work, const unsigned char* __restrict__ pos_nograd, int H,
                                                     int W, int V, int F, float* __restrict__ d_color, float* __restrict__ d_pos) {
    const int count = work[0];
    const int HW = H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        const int4 item = reinterpret_cast<const int4*>(work + 4)[i];
        const long long pi = (unsigned)item.x;
        const int d = item.y & 1;
        const float alpha = __int_as_float(item.z);
        const int b = item.w;
        const long long pj = pi + (d == 0 ? 1 : W);
        const float* go = d_out + (size_t)(alpha > 0.0f ? pi : pj) * C;
        const float* c0 = color + (size_t)pi * C;
        const float* c1 = color + (size_t)pj * C;
        float dd = 0.f;
#pragma unroll
        for (int k = 0; k < C; k++) {
            const float gk = go[k];
            dd += gk * (c1[k] - c0[k]);
            if (d_color) {
                atomicAdd(&d_color[(size_t)pj * C + k], alpha * gk);
                atomicAdd(&d_color[(size_t)pi * C + k], -alpha * gk);
            }
        }
        if (!d_pos || dd == 0.f) continue;
        const int rem = (int)(pi - (long long)b * HW);
        const int py = rem / W, px = rem - py * W;
        const float4 r0 = rast[pi], r1 = rast[pj];
        const float4* P = pos + (size_t)b * V;
        const Geo g = analyse(P, tri, opp, (int)r0.w - 1, (int)r1.w - 1, r0.z, r1.z, px, py, d, H, W);
        if (!g.ok) continue;                                   // cannot happen: same inputs as the forward
        if (!(g.dc_raw >= 0.0f && g.dc_raw <= 1.0f)) continue;  // clamp() passes no gradient outside [0,1]
        // q = dc / ds = xa - ya * dx / dy ; dL/dq = -dd
        const float dx = g.xb - g.xa, dy = g.yb - g.ya;
        const float idy = 1.0f / dy;
        const float gq = -dd;
        float gxa = gq * (1.0f + g.ya * idy);
        float gxb = gq * (-g.ya * idy);
        float gya = gq * (-dx * g.yb * idy * idy);
        float gyb = gq * (g.ya * dx * idy * idy);
        if (d == 1) {   // undo the XY flip
            float tmp = gxa; gxa = gya; gya = tmp;
            tmp = gxb; gxb = gyb; gyb = tmp;
        }
        const float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
        const float4 pa = P[g.va], pb = P[g.vb];
        const float iwa = 1.0f / pa.w, iwb = 1.0f / pb.w;
        float* D = d_pos + (size_t)b * V * 4;
        if (!(pos_nograd && pos_nograd[g.va])) {     // (detached vertices: render_nvdiffrast.py:462-464)
            atomicAdd(&D[4 * g.va + 0], gxa * xh * iwa);
            atomicAdd(&D[4 * g.va + 1], gya * yh * iwa);
            atomicAdd(&D[4 * g.va + 3], -(gxa * pa.x * xh + gya * pa.y * yh) * iwa * iwa);
        }
        if (!(pos_nograd && pos_nograd[g.vb])) {
            atomicAdd(&D[4 * g.vb + 0], gxb * xh * iwb);
            atomicAdd(&D[4 * g.vb + 1], gyb * yh * iwb);
            atomicAdd(&D[4 * g.vb + 3], -(gxb * pb.x * xh + gyb * pb.y * yh) * iwb * iwb);
        }
    }
}

// ---- in-place variant for the photometric step (C = 4): the image is never copied.  Blending `out[q] += alpha (c1 - c0)` reads the
// ORIGINAL colours of both pixels of every pair, so it runs in two tiny passes over the pair list: `blend2` analyses the candidates and
// records (pair, alpha, c0, c1); `apply` adds the deltas into the image itself.  The backward takes the original colours from the record.
// item (12 ints): {pixel index of p0, d | use1 << 1 | edge << 2, alpha bits, frame, c0[4], c1[4]}
constexpr int ITEM2 = AA_ITEM2;     // (aa_items.h: the photometric sum's launch reads the list too)
__global__ __launch_bounds__(256) void aa_blend2_kernel(const float4* __restrict__ color, const float4* __restrict__ rast,
                                                        const float4* __restrict__ pos, const int* __restrict__ tri,
                                                        const int* __restrict__ opp, const unsigned* __restrict__ cand, int H, int W,
                                                        int V, int F, int* __restrict__ work) {
    const int count = work[1];
    const unsigned HW = (unsigned)H * W;
    const int lane = threadIdx.x & 63;
    const int nwave_iter = (count + (int)(gridDim.x * 256) - 1) / (int)(gridDim.x * 256);
    for (int it = 0; it < nwave_iter; it++) {          // uniform trip count: every lane reaches the ballot
        const int i = (it * (int)gridDim.x + (int)blockIdx.x) * 256 + (int)threadIdx.x;
        bool need = false;
        Geo g;
        float alpha = 0.f;
        unsigned pi = 0, pj = 0, b = 0;
        int d = 0;
        if (i < count) {
            const unsigned e = cand[i];
            pi = e >> 1; d = (int)(e & 1u);
            pj = pi + (d == 0 ? 1u : (unsigned)W);
            b = pi / HW;
            const unsigned rem = pi - b * HW;
            const int py = (int)(rem / (unsigned)W), px = (int)(rem - (unsigned)py * W);
            const float4 r0 = rast[pi], r1 = rast[pj];
            g = analyse(pos + (size_t)b * V, tri, opp, (int)r0.w - 1, (int)r1.w - 1, r0.z, r1.z, px, py, d, H, W);
            if (g.ok) {
                const float dc = fminf(fmaxf(g.dc_raw, 0.0f), 1.0f);
                alpha = g.ds * (0.5f - dc);
                need = true;
            }
        }
        const unsigned long long m = __ballot(need);
        if (m == 0ull) continue;
        const int leader = __ffsll((long long)m) - 1;
        int base = 0;
        if (lane == leader) base = atomicAdd(&work[0], __popcll(m));
        base = __shfl(base, leader, 64);
        if (need) {
            const int slot = base + __popcll(m & ((1ull << lane) - 1ull));
            int4* item = reinterpret_cast<int4*>(work + 4) + (size_t)slot * (ITEM2 / 4);
            item[0] = make_int4((int)pi, d | (g.ds < 0.f ? 2 : 0) | (g.di << 2), __float_as_int(alpha), (int)b);
            reinterpret_cast<float4*>(item)[1] = color[pi];
            reinterpret_cast<float4*>(item)[2] = color[pj];
        }
    }
}

__global__ __launch_bounds__(256) void aa_apply_kernel(const int* __restrict__ work, int W, float* __restrict__ color) {
    const int count = work[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        const int4* item = reinterpret_cast<const int4*>(work + 4) + (size_t)i * (ITEM2 / 4);
        const int4 h = item[0];
        const float4 c0 = reinterpret_cast<const float4*>(item)[1], c1 = reinterpret_cast<const float4*>(item)[2];
        const float alpha = __int_as_float(h.z);
        const size_t pi = (unsigned)h.x, pj = pi + ((h.y & 1) == 0 ? 1 : W);
        float* o = color + 4 * (alpha > 0.0f ? pi : pj);
        atomicAdd(&o[0], alpha * (c1.x - c0.x));
        atomicAdd(&o[1], alpha * (c1.y - c0.y));
        atomicAdd(&o[2], alpha * (c1.z - c0.z));
        atomicAdd(&o[3], alpha * (c1.w - c0.w));
    }
}

// Backward for the photometric loss: d L / d out[q] = -sign(gt - out[q]) * d_sum on rgb, 0 on alpha (tracker.py:430-439), computed on
// the fly for the (few) pixels of the pair list -- no dense gradient image exists.  The colour part of the antialias backward (what flows
// to c0 / c1 BESIDES the pass-through) is ADDED into the dense image d_delta, which is zero everywhere else (aa_clear_delta restores that).
__global__ __launch_bounds__(256) void aa_photo_bwd_kernel(const float4* __restrict__ pred, const float* __restrict__ gt,
                                                           const float* __restrict__ d_sum, const float4* __restrict__ rast,
                                                           const float4* __restrict__ pos, const int* __restrict__ tri,
                                                           const int* __restrict__ opp, const int* __restrict__ work,
                                                           const unsigned char* __restrict__ pos_nograd, int H, int W, int V, int F,
                                                           float* __restrict__ d_delta, float* __restrict__ d_pos) {
    const int count = work[0];
    const int HW = H * W;
    const float gs = d_sum[0];
    auto sg = [](float e) { return e > 0.f ? 1.0f : (e < 0.f ? -1.0f : 0.0f); };
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        const int4* item = reinterpret_cast<const int4*>(work + 4) + (size_t)i * (ITEM2 / 4);
        const int4 h = item[0];
        const float4 c0 = reinterpret_cast<const float4*>(item)[1], c1 = reinterpret_cast<const float4*>(item)[2];
        const long long pi = (unsigned)h.x;
        const int d = h.y & 1;
        const float alpha = __int_as_float(h.z);
        const int b = h.w;
        const long long pj = pi + (d == 0 ? 1 : W);
        const long long q = alpha > 0.0f ? pi : pj;
        const int remq = (int)(q - (long long)b * HW);
        const int qy = remq / W, qx = remq - qy * W;
        const float* g = gt + (size_t)b * 3 * HW + (size_t)(H - 1 - qy) * W + qx;
        const float4 p = pred[q];
        const float go[3] = {-sg(g[0] - p.x) * gs, -sg(g[HW] - p.y) * gs, -sg(g[2 * HW] - p.z) * gs};
        const float dd = go[0] * (c1.x - c0.x) + go[1] * (c1.y - c0.y) + go[2] * (c1.z - c0.z);
        if (d_delta) {
#pragma unroll
            for (int k = 0; k < 3; k++) {
                atomicAdd(&d_delta[(size_t)pj * 4 + k], alpha * go[k]);
                atomicAdd(&d_delta[(size_t)pi * 4 + k], -alpha * go[k]);
            }
        }
        if (!d_pos || dd == 0.f) continue;
        const int rem = (int)(pi - (long long)b * HW);
        const int py = rem / W, px = rem - py * W;
        const float4 r0 = rast[pi], r1 = rast[pj];
        const float4* P = pos + (size_t)b * V;
        const Geo ge = analyse(P, tri, opp, (int)r0.w - 1, (int)r1.w - 1, r0.z, r1.z, px, py, d, H, W);
        if (!ge.ok) continue;                                    // cannot happen: same inputs as the forward
        if (!(ge.dc_raw >= 0.0f && ge.dc_raw <= 1.0f)) continue;  // clamp() passes no gradient outside [0,1]
        const float dx = ge.xb - ge.xa, dy = ge.yb - ge.ya;
        const float idy = 1.0f / dy;
        const float gq = -dd;
        float gxa = gq * (1.0f + ge.ya * idy);
        float gxb = gq * (-ge.ya * idy);
        float gya = gq * (-dx * ge.yb * idy * idy);
        float gyb = gq * (ge.ya * dx * idy * idy);
        if (d == 1) {   // undo the XY flip
            float tmp = gxa; gxa = gya; gya = tmp;
            tmp = gxb; gxb = gyb; gyb = tmp;
        }
        const float xh = 0.5f * (float)W, yh = 0.5f * (float)H;
        const float4 pa = P[ge.va], pb = P[ge.vb];
        const float iwa = 1.0f / pa.w, iwb = 1.0f / pb.w;
        float* D = d_pos + (size_t)b * V * 4;
        if (!(pos_nograd && pos_nograd[ge.va])) {     // (detached vertices: render_nvdiffrast.py:462-464)
            atomicAdd(&D[4 * ge.va + 0], gxa * xh * iwa);
            atomicAdd(&D[4 * ge.va + 1], gya * yh * iwa);
            atomicAdd(&D[4 * ge.va + 3], -(gxa * pa.x * xh + gya * pa.y * yh) * iwa * iwa);
        }
        if (!(pos_nograd && pos_nograd[ge.vb])) {
            atomicAdd(&D[4 * ge.vb + 0], gxb * xh * iwb);
            atomicAdd(&D[4 * ge.vb + 1], gyb * yh * iwb);
            atomicAdd(&D[4 * ge.vb + 3], -(gxb * pb.x * xh + gyb * pb.y * yh) * iwb * iwb);
        }
    }
}

__global__ __launch_bounds__(256) void aa_clear_delta_kernel(const int* __restrict__ work, int W, float4* __restrict__ d_delta) {
    const int count = work[0];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < count; i += gridDim.x * 256) {
        const int4 h = reinterpret_cast<const int4*>(work + 4)[(size_t)i * (ITEM2 / 4)];
        const size_t pi = (unsigned)h.x, pj = pi + ((h.y & 1) == 0 ? 1 : W);
        d_delta[pi] = make_float4(0.f, 0.f, 0.f, 0.f);
        d_delta[pj] = make_float4(0.f, 0.f, 0.f, 0.f);
    }
}

template <typename Fn>
int dispatch_C(int C, Fn&& f) {
    switch (C) {
        case 1: return f(std::integral_constant<int, 1>());
        case 3: return f(std::integral_constant<int, 3>());
        case 4: return f(std::integral_constant<int, 4>());
        default: return VHAP_E_BADDIM;
    }
}

}  // namespace

extern "C" size_t vhap_antialias_work_ints(int B, int H, int W, int F) {
    if (B <= 0 || H <= 0 || W <= 0 || F <= 0) return 0;
    // header + item list (worst case: every pixel blends with both neighbours) + one silhouette byte per (frame, triangle)
    // + candidate list (two pairs per pixel)
    return 4 + 4 * 2 * (size_t)B * H * W + ((size_t)B * F + 3) / 4 + 2 * (size_t)B * H * W;
}

extern "C" int vhap_antialias_fwd(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                  const int32_t* opp, int B, int H, int W, int C, int V, int F, float* out, int32_t* work,
                                  vhap_stream_t stream) {
    VHAP_ENTER();
    if (!color || !rast || !pos || !tri || !opp || !out || !work) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0 || (long long)B * H * W >= (1ll << 30)) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    hipStream_t st = vhap_stream(stream);
    unsigned char* sil = reinterpret_cast<unsigned char*>(work + 4 + 4 * 2 * (size_t)npix);
    unsigned* cand = reinterpret_cast<unsigned*>(work + 4 + 4 * 2 * (size_t)npix + ((size_t)B * F + 3) / 4);
    aa_silhouette_kernel<<<vhap_cdiv((long long)B * F, 256), 256, 0, st>>>(reinterpret_cast<const float4*>(pos), tri, opp, B, V, F, H, W, sil, work);
    VHAP_LAUNCH_CHECK();
    const int dbg = vhap_g_debug_flags;
    return dispatch_C(C, [&](auto c) {
        constexpr int CC = decltype(c)::value;
        aa_detect_kernel<CC><<<vhap_cdiv(npix, DET_T * DET_PPT * DET_CHUNKS), DET_T, 0, st>>>(color, reinterpret_cast<const float4*>(rast), (dbg & 2048) ? nullptr : sil, B, H, W,
                                                                   F, out, work, cand, dbg);
        VHAP_LAUNCH_CHECK();
        aa_blend_kernel<CC><<<1024, 256, 0, st>>>(color, reinterpret_cast<const float4*>(rast), reinterpret_cast<const float4*>(pos), tri, opp,
                                                  cand, H, W, V, F, out, work, dbg);
        VHAP_LAUNCH_CHECK();
        return VHAP_OK;
    });
}

extern "C" int vhap_antialias_bwd(const float* color, const float* rast, const float* pos, const int32_t* tri,
                                   const int32_t* opp, const float* d_out, const int32_t* work, const uint8_t* pos_nograd_verts, int B,
                                   int H, int W, int C, int V, int F, float* d_color, float* d_pos, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!color || !rast || !pos || !tri || !opp || !d_out || !work) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    const long long n = (long long)B * H * W * C;
    if (d_color && !(call_flags & VHAP_CALL_AA_PASSTHROUGH_DONE)) {   // pass-through part of the gradient
        vhap_copy_async(d_color, d_out, sizeof(float) * n, st);
        VHAP_LAUNCH_CHECK();
    }
    return dispatch_C(C, [&](auto c) {
        aa_bwd_kernel<decltype(c)::value><<<1024, 256, 0, st>>>(color, reinterpret_cast<const float4*>(rast),
                                                              reinterpret_cast<const float4*>(pos), tri, opp, d_out, work,
                                                              pos_nograd_verts, H, W, V, F, d_color, d_pos);
        VHAP_LAUNCH_CHECK();
        return VHAP_OK;
    });
}

// ---- in-place variant (photometric step) ----
extern "C" size_t vhap_antialias_inplace_work_ints(int B, int H, int W, int F) {
    if (B <= 0 || H <= 0 || W <= 0 || F <= 0) return 0;
    // header + item list (worst case: every pixel blends with both neighbours; 12 ints per item) + one silhouette byte per
    // (frame, triangle) + candidate list (two pairs per pixel).  Only the part that is used is ever touched.
    return 4 + (size_t)ITEM2 * 2 * (size_t)B * H * W + ((size_t)B * F + 3) / 4 + 2 * (size_t)B * H * W;
}

extern "C" int vhap_antialias_inplace_fwd(float* color, const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B,
                                          int H, int W, int V, int F, int32_t* work, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!color || !rast || !pos || !tri || !opp || !work) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0 || (long long)B * H * W >= (1ll << 30)) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    hipStream_t st = vhap_stream(stream);
    unsigned char* sil = reinterpret_cast<unsigned char*>(work + 4 + (size_t)ITEM2 * 2 * (size_t)npix);
    unsigned* cand = reinterpret_cast<unsigned*>(work + 4 + (size_t)ITEM2 * 2 * (size_t)npix + ((size_t)B * F + 3) / 4);
    aa_silhouette_kernel<<<vhap_cdiv((long long)B * F, 256), 256, 0, st>>>(reinterpret_cast<const float4*>(pos), tri, opp, B, V, F, H, W, sil, work);
    VHAP_LAUNCH_CHECK();
    aa_detect_kernel<4><<<vhap_cdiv(npix, DET_T * DET_PPT * DET_CHUNKS), DET_T, 0, st>>>(nullptr, reinterpret_cast<const float4*>(rast), sil, B, H, W, F, nullptr,
                                                                          work, cand, 0);
    VHAP_LAUNCH_CHECK();
    aa_blend2_kernel<<<1024, 256, 0, st>>>(reinterpret_cast<const float4*>(color), reinterpret_cast<const float4*>(rast),
                                          reinterpret_cast<const float4*>(pos), tri, opp, cand, H, W, V, F, work);
    VHAP_LAUNCH_CHECK();
    aa_apply_kernel<<<256, 256, 0, st>>>(work, W, color);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}


// The forward in pieces (silhouette + pairs + blend == vhap_antialias_inplace_fwd), for a step executor that wants the silhouette flags
// (geometry only: they need neither the rasteriser's output nor the colours) out of the way BEFORE the rasteriser has finished and the
// pixel-pair discovery (rast only) beside whatever still produces the colours.
extern "C" int vhap_antialias_inplace_silhouette(const float* pos, const int32_t* tri, const int32_t* opp, int B, int H, int W, int V, int F,
                                                 int32_t* work, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !opp || !work) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0 || (long long)B * H * W >= (1ll << 30)) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    unsigned char* sil = reinterpret_cast<unsigned char*>(work + 4 + (size_t)ITEM2 * 2 * (size_t)npix);
    aa_silhouette_kernel<<<vhap_cdiv((long long)B * F, 256), 256, 0, vhap_stream(stream)>>>(reinterpret_cast<const float4*>(pos), tri, opp, B, V, F, H, W,
                                                                                           sil, work);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_antialias_inplace_pairs(const float* rast, int B, int H, int W, int F, int32_t* work, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!rast || !work) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || F <= 0 || (long long)B * H * W >= (1ll << 30)) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    unsigned char* sil = reinterpret_cast<unsigned char*>(work + 4 + (size_t)ITEM2 * 2 * (size_t)npix);
    unsigned* cand = reinterpret_cast<unsigned*>(work + 4 + (size_t)ITEM2 * 2 * (size_t)npix + ((size_t)B * F + 3) / 4);
    aa_detect_kernel<4><<<vhap_cdiv(npix, DET_T * DET_PPT * DET_CHUNKS), DET_T, 0, vhap_stream(stream)>>>(nullptr, reinterpret_cast<const float4*>(rast), sil, B, H, W, F,
                                                                                            nullptr, work, cand, 0);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_antialias_inplace_blend(float* color, const float* rast, const float* pos, const int32_t* tri, const int32_t* opp, int B,
                                            int H, int W, int V, int F, int32_t* work, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!color || !rast || !pos || !tri || !opp || !work) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0 || (long long)B * H * W >= (1ll << 30)) return VHAP_E_BADDIM;
    const long long npix = (long long)B * H * W;
    hipStream_t st = vhap_stream(stream);
    unsigned* cand = reinterpret_cast<unsigned*>(work + 4 + (size_t)ITEM2 * 2 * (size_t)npix + ((size_t)B * F + 3) / 4);
    aa_blend2_kernel<<<1024, 256, 0, st>>>(reinterpret_cast<const float4*>(color), reinterpret_cast<const float4*>(rast),
                                          reinterpret_cast<const float4*>(pos), tri, opp, cand, H, W, V, F, work);
    VHAP_LAUNCH_CHECK();
    aa_apply_kernel<<<256, 256, 0, st>>>(work, W, color);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_antialias_photo_bwd(const float* pred_rgba, const float* gt_nchw, const float* d_sum, const float* rast, const float* pos,
                                        const int32_t* tri, const int32_t* opp, const int32_t* work, const uint8_t* pos_nograd_verts, int B,
                                        int H, int W, int V, int F, float* d_delta, float* d_pos, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pred_rgba || !gt_nchw || !d_sum || !rast || !pos || !tri || !opp || !work) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || V <= 0 || F <= 0) return VHAP_E_BADDIM;
    aa_photo_bwd_kernel<<<256, 256, 0, vhap_stream(stream)>>>(reinterpret_cast<const float4*>(pred_rgba), gt_nchw, d_sum,
                                                               reinterpret_cast<const float4*>(rast), reinterpret_cast<const float4*>(pos), tri,
                                                               opp, work, pos_nograd_verts, H, W, V, F, d_delta, d_pos);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_antialias_clear_delta(const int32_t* work, int B, int H, int W, float* d_delta, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!work || !d_delta) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    aa_clear_delta_kernel<<<256, 256, 0, vhap_stream(stream)>>>(work, W, reinterpret_cast<float4*>(d_delta));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
