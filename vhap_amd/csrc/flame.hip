// FLAME geometry kernels for gfx950: blendshapes + pose correctives + linear blend skinning
// (vhap/model/flame.py:595-634, vhap/model/lbs.py:218-239 blend_shapes, :164-166 pose offsets,
// :182-193 skinning), camera transform (vhap/util/render_nvdiffrast.py:162-206) and area-weighted
// vertex normals (:297-316), each with its backward.
//
// The two dense contractions -- [B,436] x [436,3V] forward and its transpose [B,3V] x [3V,436]
// backward -- run on the matrix cores with the exact-fp32 MFMA (v_mfma_f32_16x16x4_f32): 16 frames are
// the M dimension, so one wave produces a 16-frame x 16-vertex tile per component and keeps x, y, z of a
// vertex in the SAME lane, which lets the skinning epilogue run in registers (the reference materialises
// [B,V,4,4] transforms).  The basis is stored once per component, K-major for the forward
// ([3][K][Vp]) and vertex-major for the backward ([3][Vp][Kp]), so every MFMA operand load is a 64-byte
// run and consecutive tiles are contiguous.  Everything else is one thread per (frame, vertex).
// The tiny per-frame algebra (Rodrigues, joint regression, kinematic chain) stays on the host side.
#include "common.h"
#include "vnormal_common.h"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

constexpr int NJ = 5;  // FLAME joints

// grid (Vp/16, ceil(B/16)), 256 threads = 4 waves.  coef [Bp,Kp] (rows padded to 16, zero-filled), basis [3][K][Vp],
// A [B,5,12] (row-major 3x4 per joint), w [V,5], templ [V,3], offset [V,3] or null, transl [B,3].
// A workgroup owns a 16-frame x 16-vertex tile; the K axis (shape+expression rows, then the pose-corrective rows) is split over
// its four waves and summed through LDS: 324 x 4 waves stream the 27 MB basis instead of 81 single waves walking 109
// dependent steps each (87 -> ~20 us at V = 5143).
__global__ __launch_bounds__(256) void flame_skin_fwd_kernel(const float* __restrict__ coef, const float* __restrict__ basis,
                                                             const float* __restrict__ A, const float* __restrict__ w,
                                                             const float* __restrict__ templ, const float* __restrict__ offset,
                                                             const float* __restrict__ transl, int B, int V, int Vp, int K,
                                                             int Kb, int Kp, float* __restrict__ verts,
                                                             float* __restrict__ v_shaped, float* __restrict__ v_posed,
                                                             const float* __restrict__ mvp, float4* __restrict__ clip,
                                                             long long offset_stride) {
    __shared__ float sA[16 * NJ * 12];
    __shared__ float sT[16 * 3];
    __shared__ float sM[16 * 16];              // per-frame world -> clip matrices (fused vhap_transform_fwd), when clip != null
    __shared__ float red[2][4][3][64][4];      // [phase][wave][component][lane][r]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int v0 = blockIdx.x * 16, b0 = blockIdx.y * 16;
    {   // one batch of loads (frames past the end clamped and zeroed afterwards), then the LDS writes
        constexpr int NA = (16 * NJ * 12 + 255) / 256;
        float ra[NA];
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int i = min(tid + 256 * u, 16 * NJ * 12 - 1), f = i / (NJ * 12);
            ra[u] = A[(size_t)min(b0 + f, B - 1) * NJ * 12 + (i - f * NJ * 12)];
        }
        const int tt = min(tid, 47);
        const float rt = transl[(size_t)min(b0 + tt / 3, B - 1) * 3 + tt % 3];
        const float rm = *(clip ? mvp + (size_t)min(b0 + tid / 16, B - 1) * 16 + tid % 16 : transl);
#pragma unroll
        for (int u = 0; u < NA; u++) {
            const int i = tid + 256 * u;
            if (i < 16 * NJ * 12) sA[i] = (b0 + i / (NJ * 12) < B) ? ra[u] : 0.f;
        }
        if (tid < 48) sT[tid] = (b0 + tid / 3 < B) ? rt : 0.f;
        if (clip) sM[tid] = (b0 + tid / 16 < B) ? rm : 0.f;
    }
    const int li = lane & 15, lk = lane >> 4;
    // the epilogue's per-vertex inputs, requested before the contraction (clamped vertex / frame, stand-in address for an absent offset):
    // after it they were three more dependent round trips on the wave that finishes the tile
    const int vq = min(v0 + li, V - 1);
    const float* ofs = offset ? offset + 3 * vq : templ + 3 * vq;
    const float e_t0 = templ[3 * vq], e_t1 = templ[3 * vq + 1], e_t2 = templ[3 * vq + 2];
    float e_w[NJ], e_o[4][3];
#pragma unroll
    for (int j = 0; j < NJ; j++) e_w[j] = w[(size_t)vq * NJ + j];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const float* q = ofs + (size_t)min(b0 + lk * 4 + r, B - 1) * offset_stride;      // (stride 0: the one shared row, four times)
        e_o[r][0] = q[0]; e_o[r][1] = q[1]; e_o[r][2] = q[2];
    }
    const float* cf = coef + (size_t)(b0 + li) * Kp + lk;
    const size_t cs = (size_t)K * Vp;  // component stride
    auto run = [&](int k_begin, int k_end, int phase) {
        // this wave's share of [k_begin, k_end), in whole 4-row MFMA steps
        const int steps = (k_end - k_begin) / 4;
        const int per = (steps + 3) / 4;
        const int s0 = min(wave * per, steps), s1 = min(s0 + per, steps);
        f32x4 acc[3] = {f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}, f32x4{0.f, 0.f, 0.f, 0.f}};
        // eight K steps per trip, their 32 operand loads issued before the first MFMA (`#pragma unroll 4` on the one-step loop did not
        // unroll it -- run-time bounds inside a lambda -- and a wave walked its 25 steps as 25 dependent round trips: the whole kernel)
        constexpr int KU = 8;
        for (int st = s0; st < s1; st += KU) {
            float a[KU], bv[KU][3];
#pragma unroll
            for (int u = 0; u < KU; u++) {
                const int k = k_begin + 4 * min(st + u, s1 - 1);
                const float* bp = basis + (size_t)(k + lk) * Vp + v0 + li;
                a[u] = cf[k];
#pragma unroll
                for (int c = 0; c < 3; c++) bv[u][c] = bp[c * cs];
            }
#pragma unroll
            for (int u = 0; u < KU; u++) {
                if (st + u < s1) {
#pragma unroll
                    for (int c = 0; c < 3; c++) acc[c] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[u], bv[u][c], acc[c], 0, 0, 0);
                }
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++)
#pragma unroll
            for (int r = 0; r < 4; r++) red[phase][wave][c][lane][r] = acc[c][r];
    };
    run(0, Kb, 0);   // shape + expression (Kb is a multiple of 4)
    run(Kb, K, 1);   // pose correctives
    __syncthreads();
    if (wave != 0) return;
    // epilogue on wave 0: lane (li = vertex, lk) holds frames lk*4 + r
    const int v = v0 + li;
    if (v >= V) return;
    // offset_stride == 0: ONE offset [V,3] for the batch (static_offset); > 0: an offset row per frame (static + dynamic_offset[timesteps],
    // tracker.py:213-235) -- added per frame below
    float tx = e_t0, ty = e_t1, tz = e_t2;
    if (offset && offset_stride == 0) { tx += e_o[0][0]; ty += e_o[0][1]; tz += e_o[0][2]; }
    float wj[NJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) wj[j] = e_w[j];
#pragma unroll
    for (int r = 0; r < 4; r++) {
        const int fl = lk * 4 + r, f = b0 + fl;
        if (f >= B) continue;
        float sh[3], po[3];
#pragma unroll
        for (int c = 0; c < 3; c++) {
            sh[c] = (red[0][0][c][lane][r] + red[0][1][c][lane][r]) + (red[0][2][c][lane][r] + red[0][3][c][lane][r]);
            po[c] = (red[1][0][c][lane][r] + red[1][1][c][lane][r]) + (red[1][2][c][lane][r] + red[1][3][c][lane][r]);
        }
        float sx = tx + sh[0], sy = ty + sh[1], sz = tz + sh[2];
        const size_t o = ((size_t)f * V + v) * 3;
        if (offset && offset_stride != 0) { sx += e_o[r][0]; sy += e_o[r][1]; sz += e_o[r][2]; }
        v_shaped[o] = sx; v_shaped[o + 1] = sy; v_shaped[o + 2] = sz;
        const float px = sx + po[0], py = sy + po[1], pz = sz + po[2];
        float T[12];
#pragma unroll
        for (int q = 0; q < 12; q++) {
            float s_ = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) s_ += wj[j] * sA[(fl * NJ + j) * 12 + q];
            T[q] = s_;
        }
        v_posed[o] = px; v_posed[o + 1] = py; v_posed[o + 2] = pz;
        const float wx = T[0] * px + T[1] * py + T[2] * pz + T[3] + sT[fl * 3];
        const float wy = T[4] * px + T[5] * py + T[6] * pz + T[7] + sT[fl * 3 + 1];
        const float wz = T[8] * px + T[9] * py + T[10] * pz + T[11] + sT[fl * 3 + 2];
        verts[o] = wx; verts[o + 1] = wy; verts[o + 2] = wz;
        if (clip) {                            // same expression as transform_fwd_kernel: identical bits
            const float* M = sM + fl * 16;
            clip[(size_t)f * V + v] = make_float4(M[0] * wx + M[1] * wy + M[2] * wz + M[3], M[4] * wx + M[5] * wy + M[6] * wz + M[7],
                                                  M[8] * wx + M[9] * wy + M[10] * wz + M[11], M[12] * wx + M[13] * wy + M[14] * wz + M[15]);
        }
    }
}

// grid (ceil(V/256), B).  d_verts -> G_posed (= T_R^T d_vert), d_A [B,5,12] (atomics), d_transl [B,3] (atomics),
// G_shaped = G_posed + d_vshaped (if given)
__global__ __launch_bounds__(256) void flame_skin_bwd_kernel(const float* __restrict__ d_verts, const float* __restrict__ d_vshaped,
                                                             const float* __restrict__ v_posed, const float* __restrict__ A,
                                                             const float* __restrict__ w, int B, int V,
                                                             float* __restrict__ g_posed, float* __restrict__ g_shaped,
                                                             float* __restrict__ d_A, float* __restrict__ d_transl) {
    __shared__ float sA[NJ * 12];
    __shared__ float red[4][NJ * 12 + 3];
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < NJ * 12) sA[threadIdx.x] = A[(size_t)b * NJ * 12 + threadIdx.x];
    __syncthreads();
    float gA[NJ * 12 + 3];
#pragma unroll
    for (int i = 0; i < NJ * 12 + 3; i++) gA[i] = 0.f;
    if (v < V) {
        const size_t o = ((size_t)b * V + v) * 3;
        const float gx = d_verts[o], gy = d_verts[o + 1], gz = d_verts[o + 2];
        const float px = v_posed[o], py = v_posed[o + 1], pz = v_posed[o + 2];
        float wj[NJ], T[12];
#pragma unroll
        for (int j = 0; j < NJ; j++) wj[j] = w[(size_t)v * NJ + j];
#pragma unroll
        for (int q = 0; q < 12; q++) {
            float s = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) s += wj[j] * sA[j * 12 + q];
            T[q] = s;
        }
        const float hx = T[0] * gx + T[4] * gy + T[8] * gz;
        const float hy = T[1] * gx + T[5] * gy + T[9] * gz;
        const float hz = T[2] * gx + T[6] * gy + T[10] * gz;
        g_posed[o] = hx; g_posed[o + 1] = hy; g_posed[o + 2] = hz;
        if (g_shaped) {
            g_shaped[o] = hx + (d_vshaped ? d_vshaped[o] : 0.f);
            g_shaped[o + 1] = hy + (d_vshaped ? d_vshaped[o + 1] : 0.f);
            g_shaped[o + 2] = hz + (d_vshaped ? d_vshaped[o + 2] : 0.f);
        }
        const float dT[12] = {gx * px, gx * py, gx * pz, gx, gy * px, gy * py, gy * pz, gy, gz * px, gz * py, gz * pz, gz};
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 12; q++) gA[j * 12 + q] = wj[j] * dT[q];
        gA[NJ * 12] = gx; gA[NJ * 12 + 1] = gy; gA[NJ * 12 + 2] = gz;
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int i = 0; i < NJ * 12 + 3; i++) {
        const float s = vhap_wave_sum(gA[i]);
        if (lane == 0) red[wave][i] = s;
    }
    __syncthreads();
    if (threadIdx.x < NJ * 12 + 3) {
        const float s = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
        if (threadIdx.x < NJ * 12) atomicAdd(&d_A[(size_t)b * NJ * 12 + threadIdx.x], s);
        else atomicAdd(&d_transl[(size_t)b * 3 + threadIdx.x - NJ * 12], s);
    }
}

// d_coef [Bp,Kp] += G^T-contraction over vertices: d_coef[b][k] = sum_{v,c} g[b][v][c] * basisT[c][v][k] with g = g_shaped for
// the shape/expression columns (k < Kb) and g_posed for the pose-corrective columns.  A skinny GEMM (M = 16 frames, N = Kp,
// K = 3V ~ 15k): the work is the 27 MB read of the basis, so the grid is (N tiles) x (S splits of the vertex axis) ~ 700
// workgroups -- enough loads in flight to stream it -- each wave reducing its slice with 16x16x4 MFMAs, the four waves of a
// workgroup summed through LDS, and one atomic per output element and workgroup (S-deep chains only).
__global__ __launch_bounds__(256) void flame_coef_bwd_kernel(const float* __restrict__ g_shaped, const float* __restrict__ g_posed,
                                                             const float* __restrict__ basisT, int B, int V, int Vp, int Kb,
                                                             int Kp, int v_per_wave, float* __restrict__ d_coef) {
    __shared__ float red[4][64][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int li = lane & 15, lk = lane >> 4;
    const int nt = blockIdx.x, b0 = blockIdx.z * 16;
    const float* __restrict__ G = (nt * 16 < Kb) ? g_shaped : g_posed;     // tiles never straddle Kb (a multiple of 16)
    const int vbeg = (blockIdx.y * 4 + wave) * v_per_wave;
    const int vend = min(vbeg + v_per_wave, V);
    const bool bvalid = b0 + li < B;
    const size_t cs = (size_t)Vp * Kp;
    const float* gp = G + (size_t)min(b0 + li, B - 1) * V * 3;      // (rows past the batch: clamped, zeroed at the MFMA)
    const float* bp = basisT + (size_t)nt * 16 + li;
    f32x4 acc = f32x4{0.f, 0.f, 0.f, 0.f};
    // eight 4-vertex steps per trip, their 48 operand loads issued before the first MFMA (clamped vertices, zeroed afterwards): the
    // one-step loop did not unroll (`#pragma unroll 4`, run-time bounds) and every step was a dependent round trip
    constexpr int VU = 8;
    const int vlast = max(vend - 1, vbeg);
    for (int v0 = vbeg; v0 < vend; v0 += 4 * VU) {
        float a[VU][3], bb[VU][3];
#pragma unroll
        for (int u = 0; u < VU; u++) {
            const int v = min(v0 + 4 * u + lk, vlast);
#pragma unroll
            for (int c = 0; c < 3; c++) {
                a[u][c] = gp[(size_t)v * 3 + c];
                bb[u][c] = bp[c * cs + (size_t)v * Kp];
            }
        }
#pragma unroll
        for (int u = 0; u < VU; u++) {
            if (v0 + 4 * u < vend) {
                const bool ok = v0 + 4 * u + lk < vend;
#pragma unroll
                for (int c = 0; c < 3; c++)
                    acc = __builtin_amdgcn_mfma_f32_16x16x4f32((ok && bvalid) ? a[u][c] : 0.f, ok ? bb[u][c] : 0.f, acc, 0, 0, 0);
            }
        }
    }
#pragma unroll
    for (int r = 0; r < 4; r++) red[wave][lane][r] = acc[r];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            const float s = (red[0][lane][r] + red[1][lane][r]) + (red[2][lane][r] + red[3][lane][r]);
            if (s != 0.f) atomicAdd(&d_coef[(size_t)(b0 + lk * 4 + r) * Kp + nt * 16 + li], s);
        }
    }
}

// clip = [v;1] @ M^T with M [B,4,4] row-major (clip_r = M[r][0..2].v + M[r][3])
__global__ __launch_bounds__(256) void transform_fwd_kernel(const float* __restrict__ verts, const float* __restrict__ M, int V,
                                                            float4* __restrict__ clip) {
    __shared__ float sM[16];
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 16) sM[threadIdx.x] = M[(size_t)b * 16 + threadIdx.x];
    __syncthreads();
    if (v >= V) return;
    const float* p = verts + ((size_t)b * V + v) * 3;
    const float x = p[0], y = p[1], z = p[2];
    clip[(size_t)b * V + v] = make_float4(sM[0] * x + sM[1] * y + sM[2] * z + sM[3], sM[4] * x + sM[5] * y + sM[6] * z + sM[7],
                                           sM[8] * x + sM[9] * y + sM[10] * z + sM[11], sM[12] * x + sM[13] * y + sM[14] * z + sM[15]);
}

// d_clip -> d_verts (+= or =) and d_M [B,4,4] (atomics)
__global__ __launch_bounds__(256) void transform_bwd_kernel(const float* __restrict__ verts, const float* __restrict__ M,
                                                            const float4* __restrict__ d_clip, int V, int accumulate,
                                                            float* __restrict__ d_verts, float* __restrict__ d_M) {
    __shared__ float sM[16];
    __shared__ float red[4][16];
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < 16) sM[threadIdx.x] = M[(size_t)b * 16 + threadIdx.x];
    __syncthreads();
    float gm[16];
#pragma unroll
    for (int i = 0; i < 16; i++) gm[i] = 0.f;
    if (v < V) {
        const size_t o = ((size_t)b * V + v) * 3;
        const float4 g = d_clip[(size_t)b * V + v];
        const float gx = sM[0] * g.x + sM[4] * g.y + sM[8] * g.z + sM[12] * g.w;
        const float gy = sM[1] * g.x + sM[5] * g.y + sM[9] * g.z + sM[13] * g.w;
        const float gz = sM[2] * g.x + sM[6] * g.y + sM[10] * g.z + sM[14] * g.w;
        if (accumulate) { d_verts[o] += gx; d_verts[o + 1] += gy; d_verts[o + 2] += gz; }
        else { d_verts[o] = gx; d_verts[o + 1] = gy; d_verts[o + 2] = gz; }
        if (d_M) {
            const float x = verts[o], y = verts[o + 1], z = verts[o + 2];
            const float gg[4] = {g.x, g.y, g.z, g.w};
#pragma unroll
            for (int r = 0; r < 4; r++) { gm[4 * r] = gg[r] * x; gm[4 * r + 1] = gg[r] * y; gm[4 * r + 2] = gg[r] * z; gm[4 * r + 3] = gg[r]; }
        }
    }
    if (d_M) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < 16; i++) {
            const float s = vhap_wave_sum(gm[i]);
            if (lane == 0) red[wave][i] = s;
        }
        __syncthreads();
        if (threadIdx.x < 16) atomicAdd(&d_M[(size_t)b * 16 + threadIdx.x], red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x]);
    }
}

// Area-weighted vertex normals by GATHER over the static vertex->corner CSR (deterministic, no atomics):
// n_raw[v] = sum over incident faces of (v1-v0) x (v2-v0); fallback (0,0,1) if |n|^2 <= 1e-20; normalise.
__global__ __launch_bounds__(256) void vnormal_fwd_kernel(const float* __restrict__ verts, const int* __restrict__ tri,
                                                          const int* __restrict__ vc_ptr, const int* __restrict__ vc_idx, int V,
                                                          float* __restrict__ vn, float* __restrict__ inv_len) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    vhap_vnormal_vertex(verts + (size_t)b * V * 3, tri, vc_ptr, vc_idx, v, vn + ((size_t)b * V + v) * 3, inv_len ? inv_len + (size_t)b * V + v : nullptr);
}

// pass 1 of the backward from what the forward saved (unit normal + 1 / |raw normal|): no second gather over the incident faces
__global__ __launch_bounds__(256) void vnormal_bwd1_saved_kernel(const float* __restrict__ vn, const float* __restrict__ inv_len,
                                                                 const float* __restrict__ d_vn, int n, float* __restrict__ d_nraw) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i >= n) return;
    const float inv = inv_len[i];
    const float ux = vn[3 * i], uy = vn[3 * i + 1], uz = vn[3 * i + 2];
    const float dx = d_vn[3 * i], dy = d_vn[3 * i + 1], dz = d_vn[3 * i + 2];
    const float dot = ux * dx + uy * dy + uz * dz;
    d_nraw[3 * i] = (dx - ux * dot) * inv; d_nraw[3 * i + 1] = (dy - uy * dot) * inv; d_nraw[3 * i + 2] = (dz - uz * dot) * inv;
}

// pass 1: d_vn -> d_nraw (through the normalisation; zero where the fallback was taken)
__global__ __launch_bounds__(256) void vnormal_bwd1_kernel(const float* __restrict__ verts, const int* __restrict__ tri,
                                                           const int* __restrict__ vc_ptr, const int* __restrict__ vc_idx,
                                                           const float* __restrict__ d_vn, int V, float* __restrict__ d_nraw) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* P = verts + (size_t)b * V * 3;
    float nx = 0.f, ny = 0.f, nz = 0.f;
    for (int k = vc_ptr[v]; k < vc_ptr[v + 1]; k++) {
        const int t = vc_idx[k] / 3;
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        const float ax = P[3 * i1] - P[3 * i0], ay = P[3 * i1 + 1] - P[3 * i0 + 1], az = P[3 * i1 + 2] - P[3 * i0 + 2];
        const float bx = P[3 * i2] - P[3 * i0], by = P[3 * i2 + 1] - P[3 * i0 + 1], bz = P[3 * i2 + 2] - P[3 * i0 + 2];
        nx += ay * bz - az * by; ny += az * bx - ax * bz; nz += ax * by - ay * bx;
    }
    const size_t o = ((size_t)b * V + v) * 3;
    const float l2 = nx * nx + ny * ny + nz * nz;
    float gx = 0.f, gy = 0.f, gz = 0.f;
    if (l2 > 1e-20f) {
        const float inv = 1.0f / sqrtf(l2);
        const float ux = nx * inv, uy = ny * inv, uz = nz * inv;
        const float dx = d_vn[o], dy = d_vn[o + 1], dz = d_vn[o + 2];
        const float dot = ux * dx + uy * dy + uz * dz;
        gx = (dx - ux * dot) * inv; gy = (dy - uy * dot) * inv; gz = (dz - uz * dot) * inv;
    }
    d_nraw[o] = gx; d_nraw[o + 1] = gy; d_nraw[o + 2] = gz;
}

// pass 2 (gather again): for every corner (t,i) incident to v, g = d_nraw[i0]+d_nraw[i1]+d_nraw[i2];
// fn = e1 x e2 (e1 = v1-v0, e2 = v2-v0): d e1 = e2 x g, d e2 = g x e1; corner 0 gets -(d e1 + d e2), 1 gets d e1, 2 gets d e2
__global__ __launch_bounds__(256) void vnormal_bwd2_kernel(const float* __restrict__ verts, const int* __restrict__ tri,
                                                           const int* __restrict__ vc_ptr, const int* __restrict__ vc_idx,
                                                           const float* __restrict__ d_nraw, int V, int accumulate,
                                                           float* __restrict__ d_verts) {
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (v >= V) return;
    const float* P = verts + (size_t)b * V * 3;
    const float* G = d_nraw + (size_t)b * V * 3;
    float sx = 0.f, sy = 0.f, sz = 0.f;
    for (int k = vc_ptr[v]; k < vc_ptr[v + 1]; k++) {
        const int c = vc_idx[k], t = c / 3, i = c - 3 * t;
        const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
        const float gx = G[3 * i0] + G[3 * i1] + G[3 * i2], gy = G[3 * i0 + 1] + G[3 * i1 + 1] + G[3 * i2 + 1],
                    gz = G[3 * i0 + 2] + G[3 * i1 + 2] + G[3 * i2 + 2];
        const float ax = P[3 * i1] - P[3 * i0], ay = P[3 * i1 + 1] - P[3 * i0 + 1], az = P[3 * i1 + 2] - P[3 * i0 + 2];
        const float bx = P[3 * i2] - P[3 * i0], by = P[3 * i2 + 1] - P[3 * i0 + 1], bz = P[3 * i2 + 2] - P[3 * i0 + 2];
        const float d1x = by * gz - bz * gy, d1y = bz * gx - bx * gz, d1z = bx * gy - by * gx;   // e2 x g
        const float d2x = gy * az - gz * ay, d2y = gz * ax - gx * az, d2z = gx * ay - gy * ax;   // g x e1
        if (i == 0) { sx -= d1x + d2x; sy -= d1y + d2y; sz -= d1z + d2z; }
        else if (i == 1) { sx += d1x; sy += d1y; sz += d1z; }
        else { sx += d2x; sy += d2y; sz += d2z; }
    }
    const size_t o = ((size_t)b * V + v) * 3;
    if (accumulate) { d_verts[o] += sx; d_verts[o + 1] += sy; d_verts[o + 2] += sz; }
    else { d_verts[o] = sx; d_verts[o + 1] = sy; d_verts[o + 2] = sz; }
}

// ---- the vertex stage of the backward in ONE launch: gradient w.r.t. the world-space vertices assembled in registers from its three
// sources -- what already sits in d_verts (landmarks), the second pass of the vertex-normal backward (CSR gather over incident faces)
// and the clip transform's backward M^T d_clip -- and chained at once through the skinning backward (g_posed, g_shaped, d_A, d_transl)
// and the per-vertex sum over frames for the shared static offset.  Replaces vnormal_bwd2 + transform_bwd + flame_skin_bwd + sum_frames
// (four dependent latency-bound launches over 82 k vertices) and never writes d_verts back.  The 16 + 63 per-frame sums (d_M, d_A,
// d_transl) are reduced per row of 16 lanes with DPP adds and finished through LDS.  grid (ceil(V/256), B).
constexpr int VB_NRED = 16 + NJ * 12 + 3;
__global__ __launch_bounds__(256) void verts_bwd_fused_kernel(const float* __restrict__ verts, const int* __restrict__ tri,
                                                              const int* __restrict__ vc_ptr, const int* __restrict__ vc_idx,
                                                              const float* __restrict__ vn, const float* __restrict__ inv_len,
                                                              const float* __restrict__ d_vn, const float* __restrict__ M,
                                                              const float4* __restrict__ d_clip, const float* __restrict__ d_verts_in,
                                                              const float* __restrict__ v_posed, const float* __restrict__ A,
                                                              const float* __restrict__ w, int V, float* __restrict__ g_posed,
                                                              float* __restrict__ g_shaped, float* __restrict__ d_A,
                                                              float* __restrict__ d_transl, float* __restrict__ d_M,
                                                              float* __restrict__ d_offset) {
    __shared__ float sA[NJ * 12], sM[16];
    __shared__ float red[16][VB_NRED];
    const int b = blockIdx.y, v = blockIdx.x * 256 + threadIdx.x;
    if (threadIdx.x < NJ * 12) sA[threadIdx.x] = A[(size_t)b * NJ * 12 + threadIdx.x];
    if (threadIdx.x >= 64 && threadIdx.x < 80) sM[threadIdx.x - 64] = M[(size_t)b * 16 + threadIdx.x - 64];
    __syncthreads();
    float acc[VB_NRED];
#pragma unroll
    for (int i = 0; i < VB_NRED; i++) acc[i] = 0.f;
    if (v < V) {
        const float* P = verts + (size_t)b * V * 3;
        // pass 1 of the vertex-normal backward (d_vn -> gradient of the raw normal, through the normalisation: vnormal_bwd1_saved_kernel) is
        // re-done per gathered vertex from what the forward saved -- 7 floats instead of 3 per vertex of an incident face, and one launch
        // (6 us + a hand-over on the step's tail) less
        const float* UN = vn + (size_t)b * V * 3;
        const float* IL = inv_len + (size_t)b * V;
        const float* DN = d_vn + (size_t)b * V * 3;
        const size_t o = ((size_t)b * V + v) * 3;
        // every load that depends on the vertex alone is requested now and consumed after the walk over the incident faces
        const int k0 = vc_ptr[v], k1 = vc_ptr[v + 1];
        const float* dvi = d_verts_in ? d_verts_in + o : P + 3 * v;            // (stand-in address for the optional input)
        const float r_dv0 = dvi[0], r_dv1 = dvi[1], r_dv2 = dvi[2];
        const float4 gc = d_clip[(size_t)b * V + v];
        const float vx = P[3 * v], vy = P[3 * v + 1], vz = P[3 * v + 2];
        const float px = v_posed[o], py = v_posed[o + 1], pz = v_posed[o + 2];
        float wj[NJ];
#pragma unroll
        for (int j = 0; j < NJ; j++) wj[j] = w[(size_t)v * NJ + j];
        float gx = d_verts_in ? r_dv0 : 0.f, gy = d_verts_in ? r_dv1 : 0.f, gz = d_verts_in ? r_dv2 : 0.f;
        // vertex-normal backward, pass 2 (see vnormal_bwd2_kernel).  The incident faces FOUR at a time: corner ids -> vertex ids ->
        // normals' gradients and positions are three dependent round trips per face, and one face per trip (valence ~6) made this kernel
        // ~20 of them in series; the contributions are added in the order of the corner list, as before.
        constexpr int FB = 4;
        for (int k = k0; k < k1; k += FB) {
            int cc[FB], ii[FB][3];
            float g_[FB][3][3], p_[FB][3][3], u_[FB][3][3], il_[FB][3];
#pragma unroll
            for (int u = 0; u < FB; u++) cc[u] = vc_idx[k + u < k1 ? k + u : k1 - 1];
#pragma unroll
            for (int u = 0; u < FB; u++) {
                const int t = cc[u] / 3;
                ii[u][0] = tri[3 * t]; ii[u][1] = tri[3 * t + 1]; ii[u][2] = tri[3 * t + 2];
            }
#pragma unroll
            for (int u = 0; u < FB; u++)
#pragma unroll
                for (int q = 0; q < 3; q++)
#pragma unroll
                    for (int c = 0; c < 3; c++) { g_[u][q][c] = DN[3 * ii[u][q] + c]; u_[u][q][c] = UN[3 * ii[u][q] + c]; p_[u][q][c] = P[3 * ii[u][q] + c]; }
#pragma unroll
            for (int u = 0; u < FB; u++)
#pragma unroll
                for (int q = 0; q < 3; q++) il_[u][q] = IL[ii[u][q]];
#pragma unroll
            for (int u = 0; u < FB; u++)
#pragma unroll
                for (int q = 0; q < 3; q++) {          // d_vn -> d_nraw, the expressions of vnormal_bwd1_saved_kernel
                    const float dx = g_[u][q][0], dy = g_[u][q][1], dz = g_[u][q][2];
                    const float ux = u_[u][q][0], uy = u_[u][q][1], uz = u_[u][q][2];
                    const float dot = ux * dx + uy * dy + uz * dz;
                    g_[u][q][0] = (dx - ux * dot) * il_[u][q]; g_[u][q][1] = (dy - uy * dot) * il_[u][q]; g_[u][q][2] = (dz - uz * dot) * il_[u][q];
                }
#pragma unroll
            for (int u = 0; u < FB; u++) {
                if (k + u >= k1) break;
                const int i = cc[u] - 3 * (cc[u] / 3);
                const float hx = g_[u][0][0] + g_[u][1][0] + g_[u][2][0], hy = g_[u][0][1] + g_[u][1][1] + g_[u][2][1],
                            hz = g_[u][0][2] + g_[u][1][2] + g_[u][2][2];
                const float ax = p_[u][1][0] - p_[u][0][0], ay = p_[u][1][1] - p_[u][0][1], az = p_[u][1][2] - p_[u][0][2];
                const float bx = p_[u][2][0] - p_[u][0][0], by = p_[u][2][1] - p_[u][0][1], bz = p_[u][2][2] - p_[u][0][2];
                const float d1x = by * hz - bz * hy, d1y = bz * hx - bx * hz, d1z = bx * hy - by * hx;   // e2 x g
                const float d2x = hy * az - hz * ay, d2y = hz * ax - hx * az, d2z = hx * ay - hy * ax;   // g x e1
                if (i == 0) { gx -= d1x + d2x; gy -= d1y + d2y; gz -= d1z + d2z; }
                else if (i == 1) { gx += d1x; gy += d1y; gz += d1z; }
                else { gx += d2x; gy += d2y; gz += d2z; }
            }
        }
        // clip transform backward (see transform_bwd_kernel)
        gx += sM[0] * gc.x + sM[4] * gc.y + sM[8] * gc.z + sM[12] * gc.w;
        gy += sM[1] * gc.x + sM[5] * gc.y + sM[9] * gc.z + sM[13] * gc.w;
        gz += sM[2] * gc.x + sM[6] * gc.y + sM[10] * gc.z + sM[14] * gc.w;
        {
            const float x = vx, y = vy, z = vz;
            const float gg[4] = {gc.x, gc.y, gc.z, gc.w};
#pragma unroll
            for (int r = 0; r < 4; r++) { acc[4 * r] = gg[r] * x; acc[4 * r + 1] = gg[r] * y; acc[4 * r + 2] = gg[r] * z; acc[4 * r + 3] = gg[r]; }
        }
        // skinning backward (see flame_skin_bwd_kernel)
        float T[12];
#pragma unroll
        for (int q = 0; q < 12; q++) {
            float s_ = 0.f;
#pragma unroll
            for (int j = 0; j < NJ; j++) s_ += wj[j] * sA[j * 12 + q];
            T[q] = s_;
        }
        const float hx = T[0] * gx + T[4] * gy + T[8] * gz;
        const float hy = T[1] * gx + T[5] * gy + T[9] * gz;
        const float hz = T[2] * gx + T[6] * gy + T[10] * gz;
        g_posed[o] = hx; g_posed[o + 1] = hy; g_posed[o + 2] = hz;
        g_shaped[o] = hx; g_shaped[o + 1] = hy; g_shaped[o + 2] = hz;
        if (d_offset) {                          // shared static offset: sum over the frames (16-way atomics per component)
            atomicAdd(&d_offset[3 * v], hx); atomicAdd(&d_offset[3 * v + 1], hy); atomicAdd(&d_offset[3 * v + 2], hz);
        }
        const float dT[12] = {gx * px, gx * py, gx * pz, gx, gy * px, gy * py, gy * pz, gy, gz * px, gz * py, gz * pz, gz};
#pragma unroll
        for (int j = 0; j < NJ; j++)
#pragma unroll
            for (int q = 0; q < 12; q++) acc[16 + j * 12 + q] = wj[j] * dT[q];
        acc[16 + NJ * 12] = gx; acc[16 + NJ * 12 + 1] = gy; acc[16 + NJ * 12 + 2] = gz;
    }
    vhap_row_sums_dpp<VB_NRED>(acc);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if ((lane & 15) == 0) {
#pragma unroll
        for (int i = 0; i < VB_NRED; i++) red[wave * 4 + (lane >> 4)][i] = acc[i];
    }
    __syncthreads();
    if (threadIdx.x < VB_NRED) {
        float s_ = 0.f;
#pragma unroll
        for (int r = 0; r < 16; r++) s_ += red[r][threadIdx.x];
        const int i = threadIdx.x;
        if (i < 16) { if (d_M) atomicAdd(&d_M[(size_t)b * 16 + i], s_); }
        else if (i < 16 + NJ * 12) atomicAdd(&d_A[(size_t)b * NJ * 12 + i - 16], s_);
        else atomicAdd(&d_transl[(size_t)b * 3 + i - 16 - NJ * 12], s_);
    }
}

}  // namespace

extern "C" int vhap_flame_skin_fwd(const float* coef, const float* basis, const float* A, const float* lbs_weights,
                                   const float* v_template, const float* offset, const float* transl, int B, int V, int Vp,
                                   int K, int Kb, int Kp, float* verts, float* v_shaped, float* v_posed, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!coef || !basis || !A || !lbs_weights || !v_template || !transl || !verts || !v_shaped || !v_posed) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || Vp < V || Vp % 64 || K <= 0 || K % 4 || Kb % 4 || Kb > K || Kp < K) return VHAP_E_BADDIM;
    flame_skin_fwd_kernel<<<dim3(Vp / 16, (B + 15) / 16), 256, 0, vhap_stream(stream)>>>(coef, basis, A, lbs_weights, v_template, offset,
                                                                                        transl, B, V, Vp, K, Kb, Kp, verts, v_shaped,
                                                                                        v_posed, nullptr, nullptr,
                                                                                        (call_flags & VHAP_CALL_OFFSET_PER_FRAME) ? 3ll * V : 0ll);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_flame_skin_clip_fwd(const float* coef, const float* basis, const float* A, const float* lbs_weights,
                                        const float* v_template, const float* offset, const float* transl, const float* mvp, int B,
                                        int V, int Vp, int K, int Kb, int Kp, float* verts, float* v_shaped, float* v_posed,
                                        float* clip, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!coef || !basis || !A || !lbs_weights || !v_template || !transl || !verts || !v_shaped || !v_posed || !mvp || !clip)
        return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || Vp < V || Vp % 64 || K <= 0 || K % 4 || Kb % 4 || Kb > K || Kp < K) return VHAP_E_BADDIM;
    flame_skin_fwd_kernel<<<dim3(Vp / 16, (B + 15) / 16), 256, 0, vhap_stream(stream)>>>(coef, basis, A, lbs_weights, v_template, offset,
                                                                                        transl, B, V, Vp, K, Kb, Kp, verts, v_shaped,
                                                                                        v_posed, mvp, reinterpret_cast<float4*>(clip),
                                                                                        (call_flags & VHAP_CALL_OFFSET_PER_FRAME) ? 3ll * V : 0ll);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" size_t vhap_flame_bwd_partial_floats(int B, int Vp, int Kp) {
    (void)B; (void)Vp; (void)Kp;
    return 0;   // no scratch needed any more (split-K sums use atomics); kept for ABI stability
}

extern "C" int vhap_flame_skin_bwd(const float* d_verts, const float* d_vshaped, const float* v_posed, const float* A,
                                   const float* lbs_weights, const float* basisT, int B, int V, int Vp, int Kb, int Kp,
                                   float* g_posed, float* g_shaped, float* partials, float* d_coef, float* d_A, float* d_transl,
                                   int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!d_verts || !v_posed || !A || !lbs_weights || !basisT || !g_posed || !g_shaped || !d_coef || !d_A || !d_transl)
        return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || Vp < V || Vp % 64 || Kb % 16 || Kp % 16 || Kp / 16 > 32) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    flame_skin_bwd_kernel<<<dim3(vhap_cdiv(V, 256), B), 256, 0, st>>>(d_verts, d_vshaped, v_posed, A, lbs_weights, B, V, g_posed, g_shaped,
                                                                      d_A, d_transl);
    VHAP_LAUNCH_CHECK();
    (void)partials;                                  // (kept in the signature; the split-K sums go through atomics now)
    const int ntiles = Kp / 16, btiles = (B + 15) / 16;
    VHAP_ZERO_ACC(d_coef, sizeof(float) * (size_t)btiles * 16 * Kp, st);
    int S = 768 / (ntiles * btiles);                 // ~768 workgroups in flight
    S = S < 1 ? 1 : S;
    int v_per_wave = (V + S * 4 - 1) / (S * 4);
    v_per_wave = (v_per_wave + 3) / 4 * 4;           // whole 4-vertex MFMA steps
    S = (V + v_per_wave * 4 - 1) / (v_per_wave * 4);
    flame_coef_bwd_kernel<<<dim3(ntiles, S, btiles), 256, 0, st>>>(g_shaped, g_posed, basisT, B, V, Vp, Kb, Kp, v_per_wave, d_coef);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_verts_bwd_fused(const float* verts, const int32_t* tri, const int32_t* vc_ptr, const int32_t* vc_idx, const float* vn,
                                    const float* inv_len, const float* d_vn, const float* mvp, const float* d_clip, const float* d_verts_in,
                                    const float* v_posed, const float* A, const float* lbs_weights, const float* basisT, int B, int V, int Vp,
                                    int Kb, int Kp, float* scratch, float* g_posed, float* g_shaped, float* d_coef, float* d_A,
                                    float* d_transl, float* d_mvp, float* d_offset, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !tri || !vc_ptr || !vc_idx || !vn || !inv_len || !d_vn || !mvp || !d_clip || !v_posed || !A || !lbs_weights || !basisT ||
        !scratch || !g_posed || !g_shaped || !d_coef || !d_A || !d_transl)
        return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || B > 65535 || Vp < V || Vp % 64 || Kb % 16 || Kp % 16 || Kp / 16 > 32) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    // (`scratch` -- [B,V,3], the gradient of the raw normals -- is no longer written: the kernel re-derives it per gathered vertex)
    verts_bwd_fused_kernel<<<dim3(vhap_cdiv(V, 256), B), 256, 0, st>>>(verts, tri, vc_ptr, vc_idx, vn, inv_len, d_vn, mvp, reinterpret_cast<const float4*>(d_clip),
                                                                       d_verts_in, v_posed, A, lbs_weights, V, g_posed, g_shaped, d_A, d_transl,
                                                                       d_mvp, d_offset);
    VHAP_LAUNCH_CHECK();
    const int ntiles = Kp / 16, btiles = (B + 15) / 16;
    VHAP_ZERO_ACC(d_coef, sizeof(float) * (size_t)btiles * 16 * Kp, st);
    int S = 768 / (ntiles * btiles);                 // ~768 workgroups in flight
    S = S < 1 ? 1 : S;
    int v_per_wave = (V + S * 4 - 1) / (S * 4);
    v_per_wave = (v_per_wave + 3) / 4 * 4;           // whole 4-vertex MFMA steps
    S = (V + v_per_wave * 4 - 1) / (v_per_wave * 4);
    flame_coef_bwd_kernel<<<dim3(ntiles, S, btiles), 256, 0, st>>>(g_shaped, g_posed, basisT, B, V, Vp, Kb, Kp, v_per_wave, d_coef);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_transform_fwd(const float* verts, const float* M, int B, int V, float* clip, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !M || !clip) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || B > 65535) return VHAP_E_BADDIM;
    transform_fwd_kernel<<<dim3(vhap_cdiv(V, 256), B), 256, 0, vhap_stream(stream)>>>(verts, M, V, reinterpret_cast<float4*>(clip));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_transform_bwd(const float* verts, const float* M, const float* d_clip, int B, int V, int accumulate,
                                  float* d_verts, float* d_M, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !M || !d_clip || !d_verts) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || B > 65535) return VHAP_E_BADDIM;
    transform_bwd_kernel<<<dim3(vhap_cdiv(V, 256), B), 256, 0, vhap_stream(stream)>>>(verts, M, reinterpret_cast<const float4*>(d_clip), V,
                                                                                      accumulate, d_verts, d_M);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_vnormal_fwd(const float* verts, const int32_t* tri, const int32_t* vc_ptr, const int32_t* vc_idx, int B, int V,
                                float* vn, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !tri || !vc_ptr || !vc_idx || !vn) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || B > 65535) return VHAP_E_BADDIM;
    vnormal_fwd_kernel<<<dim3(vhap_cdiv(V, 256), B), 256, 0, vhap_stream(stream)>>>(verts, tri, vc_ptr, vc_idx, V, vn, nullptr);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_vnormal_fwd_saved(const float* verts, const int32_t* tri, const int32_t* vc_ptr, const int32_t* vc_idx, int B, int V,
                                      float* vn, float* inv_len, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !tri || !vc_ptr || !vc_idx || !vn || !inv_len) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || B > 65535) return VHAP_E_BADDIM;
    vnormal_fwd_kernel<<<dim3(vhap_cdiv(V, 256), B), 256, 0, vhap_stream(stream)>>>(verts, tri, vc_ptr, vc_idx, V, vn, inv_len);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_vnormal_bwd_saved(const float* verts, const int32_t* tri, const int32_t* vc_ptr, const int32_t* vc_idx, const float* vn,
                                      const float* inv_len, const float* d_vn, int B, int V, int accumulate, float* scratch,
                                      float* d_verts, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !tri || !vc_ptr || !vc_idx || !vn || !inv_len || !d_vn || !scratch || !d_verts) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || B > 65535) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    vnormal_bwd1_saved_kernel<<<vhap_cdiv((long long)B * V, 256), 256, 0, st>>>(vn, inv_len, d_vn, B * V, scratch);
    VHAP_LAUNCH_CHECK();
    vnormal_bwd2_kernel<<<dim3(vhap_cdiv(V, 256), B), 256, 0, st>>>(verts, tri, vc_ptr, vc_idx, scratch, V, accumulate, d_verts);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_vnormal_bwd(const float* verts, const int32_t* tri, const int32_t* vc_ptr, const int32_t* vc_idx,
                                const float* d_vn, int B, int V, int accumulate, float* scratch, float* d_verts,
                                vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !tri || !vc_ptr || !vc_idx || !d_vn || !scratch || !d_verts) return VHAP_E_NULLPTR;
    if (B <= 0 || V <= 0 || B > 65535) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    const dim3 grid(vhap_cdiv(V, 256), B);
    vnormal_bwd1_kernel<<<grid, 256, 0, st>>>(verts, tri, vc_ptr, vc_idx, d_vn, V, scratch);
    VHAP_LAUNCH_CHECK();
    vnormal_bwd2_kernel<<<grid, 256, 0, st>>>(verts, tri, vc_ptr, vc_idx, scratch, V, accumulate, d_verts);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
