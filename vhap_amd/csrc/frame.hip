// Per-frame parameter stage of the fit step for gfx950: everything between the optimised parameter arrays and the
// FLAME skinning kernel, plus the parameter regularisers -- the ~400 tiny torch launches per step that the reference
// spends in FlameHead.forward (vhap/model/flame.py:571-634), lbs.batch_rodrigues / batch_rigid_transform
// (vhap/model/lbs.py:25-57, 254-301) and the L2 / temporal-smoothness energies (vhap/model/tracker.py:616-680).
//
//   frame_prep_fwd : gather the batch rows (timestep index), betas = [shape | expr], rest joints
//                    J = JT + JS betas (+ J_regressor static_offset), Rodrigues x J, kinematic chain, relative 3x4
//                    transforms A, pose feature (R_j - I), padded coefficient rows for the MFMA blend kernel, and the
//                    six weighted parameter energies.
//   frame_prep_bwd : the hand-derived reverse of all of the above, scattered straight into full-size gradient arrays.
//
// One workgroup per frame; the serial 5-joint algebra runs on lane 0 (a few hundred flops), the 400-wide gathers and
// the 15 x 400 joint regression on all 256 lanes.  HBM traffic is a few hundred KB: launch-latency bound by design,
// the point is 2 launches instead of ~400.
#include "common.h"

namespace {

constexpr int MAXJ = 8;
constexpr int FP_THREADS = 256;

struct FrameCfg {
    int B, Bp, N;            // frames in the batch, padded rows of coef, timesteps in the parameter arrays
    int NS, NE, J, P, Kp;    // shape / expr counts, joints, pose-feature length 9 (J - 1), coef row stride
    int V;                   // vertices (static offset / J_regressor)
    int parents[MAXJ];
    float w[12];             // see vhap_hip.h: VHAP_FW_*
};

struct Mat3 { float m[9]; };

__device__ __forceinline__ Mat3 mul(const Mat3& a, const Mat3& b) {
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[3 * i] * b.m[j] + a.m[3 * i + 1] * b.m[3 + j] + a.m[3 * i + 2] * b.m[6 + j];
    return c;
}
__device__ __forceinline__ Mat3 mul_nt(const Mat3& a, const Mat3& b) {   // a b^T
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[3 * i] * b.m[3 * j] + a.m[3 * i + 1] * b.m[3 * j + 1] + a.m[3 * i + 2] * b.m[3 * j + 2];
    return c;
}
__device__ __forceinline__ Mat3 mul_tn(const Mat3& a, const Mat3& b) {   // a^T b
    Mat3 c;
#pragma unroll
    for (int i = 0; i < 3; i++)
#pragma unroll
        for (int j = 0; j < 3; j++) c.m[3 * i + j] = a.m[i] * b.m[j] + a.m[3 + i] * b.m[3 + j] + a.m[6 + i] * b.m[6 + j];
    return c;
}
__device__ __forceinline__ void matvec(const Mat3& a, const float* v, float* o) {
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a.m[3 * i] * v[0] + a.m[3 * i + 1] * v[1] + a.m[3 * i + 2] * v[2];
}
__device__ __forceinline__ void matvec_t(const Mat3& a, const float* v, float* o) {   // a^T v
#pragma unroll
    for (int i = 0; i < 3; i++) o[i] = a.m[i] * v[0] + a.m[3 + i] * v[1] + a.m[6 + i] * v[2];
}

// lbs.batch_rodrigues: angle = ||r + 1e-8||, n = r / angle, R = I + sin K + (1 - cos) K K
__device__ __forceinline__ Mat3 rodrigues(const float* r) {
    const float ax = r[0] + 1e-8f, ay = r[1] + 1e-8f, az = r[2] + 1e-8f;
    const float th = sqrtf(ax * ax + ay * ay + az * az);
    const float x = r[0] / th, y = r[1] / th, z = r[2] / th;
    float s, c;
    sincosf(th, &s, &c);
    const float oc = 1.0f - c;
    Mat3 K = {{0.f, -z, y, z, 0.f, -x, -y, x, 0.f}};
    const Mat3 KK = mul(K, K);
    Mat3 R;
#pragma unroll
    for (int i = 0; i < 9; i++) R.m[i] = ((i % 4 == 0) ? 1.0f : 0.0f) + s * K.m[i] + oc * KK.m[i];
    return R;
}

// reverse of rodrigues(): dR [3x3] -> dr [3] (accumulated)
__device__ __forceinline__ void rodrigues_bwd(const float* r, const Mat3& dR, float* dr) {
    const float ax = r[0] + 1e-8f, ay = r[1] + 1e-8f, az = r[2] + 1e-8f;
    const float th = sqrtf(ax * ax + ay * ay + az * az);
    const float ith = 1.0f / th;
    const float x = r[0] * ith, y = r[1] * ith, z = r[2] * ith;
    float s, c;
    sincosf(th, &s, &c);
    const float oc = 1.0f - c;
    const Mat3 K = {{0.f, -z, y, z, 0.f, -x, -y, x, 0.f}};
    const Mat3 KK = mul(K, K);
    float ds = 0.f, doc = 0.f;
#pragma unroll
    for (int i = 0; i < 9; i++) { ds += dR.m[i] * K.m[i]; doc += dR.m[i] * KK.m[i]; }
    const Mat3 a = mul_nt(dR, K), b = mul_tn(K, dR);     // d<dR, K K>/dK = dR K^T + K^T dR
    Mat3 dK;
#pragma unroll
    for (int i = 0; i < 9; i++) dK.m[i] = s * dR.m[i] + oc * (a.m[i] + b.m[i]);
    float dth = c * ds + s * doc;
    const float dn[3] = {dK.m[7] - dK.m[5], dK.m[2] - dK.m[6], dK.m[3] - dK.m[1]};
    dth -= (dn[0] * r[0] + dn[1] * r[1] + dn[2] * r[2]) * ith * ith;
    dr[0] += dn[0] * ith + dth * ax * ith;
    dr[1] += dn[1] * ith + dth * ay * ith;
    dr[2] += dn[2] * ith + dth * az * ith;
}

// FLAME's kinematic tree (root, neck, jaw, left eye, right eye).  With the tree known at compile time the joint loops unroll
// completely and every per-joint matrix stays in registers; a run-time `parents` table forces them into scratch memory
// (dynamic indexing), which costs ~50 us of serial latency per launch.
__device__ __forceinline__ constexpr int flame_parent(int j) { return j == 0 ? -1 : (j == 1 ? 0 : 1); }

// The uncalibrated camera (tracker.py:148-157) rides in the per-frame launches as ONE more workgroup (vhap_frame_prep_fwd_camera /
// vhap_frame_prep_bwd_camera): its forward is a 7 us single-wave kernel at the head of the step's critical path, its backward a
// side-stream launch whose event record and wait sat in front of the last two kernels of the step.
struct CamFwdJob {
    const float *focal, *RT;          // focal == null: no job
    float fscale, cx, cy, h, w, near, far;
    int B, rtstride;
    float* mvp;
};
struct CamBwdJob {
    const float *RT, *d_mvp;          // RT == null: no job
    float h, w, scale;
    int B, rtstride;
    float* d_focal;
};
// mvp = P(K) [RT; 0 0 0 1], one lane per (frame, column): the body of camera_fwd_kernel for K = (f, f, cx, cy), f = focal[0] * fscale
__device__ __forceinline__ void camera_focal_fwd_item(const CamFwdJob& c, int i) {
    const int b = i >> 2, col = i & 3;
    const float f = c.focal[0] * c.fscale;
    const float* rt = c.RT + (size_t)b * c.rtstride;
    const float mv0 = rt[col], mv1 = rt[4 + col], mv2 = rt[8 + col], mv3 = col == 3 ? 1.0f : 0.0f;
    float* m = c.mvp + (size_t)b * 16;
    const float w = c.w, h = c.h, near = c.near, far = c.far;
    m[col] = f * 2.0f / w * mv0 + (w - 2.0f * c.cx) / w * mv2;
    m[4 + col] = f * 2.0f / h * mv1 + (h - 2.0f * c.cy) / h * mv2;
    m[8 + col] = -(far + near) / (far - near) * mv2 + (-2.0f * far * near / (far - near)) * mv3;
    m[12 + col] = -mv2;
}
// d(focal_length) += scale * sum_b (dK[b].fx + dK[b].fy): the body of camera_focal_bwd_kernel, run by the first 64 threads of a workgroup
// (every thread of the workgroup must call: barriers inside).  part: 64 floats of LDS.
__device__ __forceinline__ void camera_focal_bwd_block(const CamBwdJob& c, float* part) {
    float s = 0.f;
    for (int b0 = 0; b0 < c.B; b0 += 64) {
        const int b = b0 + (int)threadIdx.x;
        float v = 0.f;
        if (threadIdx.x < 64 && b < c.B) {
            const float* rt = c.RT + (size_t)b * c.rtstride;
            const float* d = c.d_mvp + (size_t)b * 16;
            float g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                g0 += d[k] * rt[k];
                g1 += d[4 + k] * rt[4 + k];
            }
            v = g0 * 2.0f / c.w + g1 * 2.0f / c.h;
        }
        if (threadIdx.x < 64) part[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < min(64, c.B - b0); k++) s += part[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) c.d_focal[0] += s * c.scale;
}

struct FrameIn {
    const long long* ts;
    const float *shape, *expr, *rotation, *translation, *neck, *jaw, *eyes;
    const float *JT, *JS, *Jw, *offset;     // [J,3], [3J, NS+NE], [M,J] weights of the M vertices with a non-zero J_regressor column, [V,3] or null
    const int* Jv;                          // [M] their vertex ids
    int M;
    long long offset_stride;                // 0: one offset [V,3] for the batch; 3 V: an offset row per frame of the batch ([B,V,3])
};

__device__ __forceinline__ void gather_pose(const FrameIn& in, long long t, float* pose /*15*/) {
#pragma unroll
    for (int c = 0; c < 3; c++) {
        pose[c] = in.rotation[3 * t + c];
        pose[3 + c] = in.neck[3 * t + c];
        pose[6 + c] = in.jaw[3 * t + c];
        pose[9 + c] = in.eyes[6 * t + c];
        pose[12 + c] = in.eyes[6 * t + 3 + c];
    }
}

__device__ __forceinline__ float block_sum(float v, float* red /*[FP_THREADS/64]*/) {
    v = vhap_wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.f;
#pragma unroll
    for (int w = 0; w < FP_THREADS / 64; w++) t += red[w];
    return t;
}

// terms: [0] smooth_pose [1] reg_joint [2] smooth_joint [3] reg_expr [4] smooth_expr [5] reg_shape   (weighted)
template <int JT>   // JT = 5: FLAME tree, fully unrolled; JT = 0: generic tree from cfg.parents
__global__ __launch_bounds__(FP_THREADS) void frame_prep_fwd_kernel(FrameCfg cfg, FrameIn in, float* __restrict__ coef,
                                                                    float* __restrict__ A, float* __restrict__ transl,
                                                                    float* __restrict__ Jrest, float* __restrict__ terms, const CamFwdJob cam) {
    __shared__ float Jl[3 * MAXJ];
    __shared__ float red[FP_THREADS / 64];
    const int b = blockIdx.x, tid = threadIdx.x;
    if (b == cfg.Bp) {          // the workgroup behind the frames': the camera (launched only with a job)
        for (int i = tid; i < cam.B * 4; i += FP_THREADS) camera_focal_fwd_item(cam, i);
        return;
    }
    const int NB = cfg.NS + cfg.NE;
    float* row = coef + (size_t)b * cfg.Kp;
    if (b >= cfg.B) {   // padding rows of the MFMA tile
        for (int k = tid; k < cfg.Kp; k += FP_THREADS) row[k] = 0.f;
        return;
    }
    // ---- all global loads of this kernel in two levels: what needs nothing (the offset list's first slice, JT) and the timestep index;
    // then what needs those (pose rows, coefficient slices with their JS columns, the listed vertices' offsets).  One workgroup per frame
    // has nothing to hide a round trip behind, and next to the bandwidth-bound texture assembly a round trip is 2-4 us: as separate
    // loops (coefficients -> three block sums -> JS slices -> vertex list -> weights / offsets) the kernel was ~8 of them long. ----
    const int NO = JT ? 3 * JT : 3 * MAXJ, NJ = JT ? JT : MAXJ, last = 3 * cfg.J - 1;
    const bool has_off = in.offset != nullptr && in.M > 0;
    const int m0 = has_off ? min(tid, in.M - 1) : 0;
    const int jv0 = *(has_off ? in.Jv + m0 : reinterpret_cast<const int*>(in.JT));           // (stand-in address: value unused)
    float wv0[MAXJ];
#pragma unroll
    for (int j = 0; j < NJ; j++) wv0[j] = *(has_off ? in.Jw + (size_t)m0 * cfg.J + (j < cfg.J ? j : cfg.J - 1) : in.JT);
    const float jt_v = in.JT[tid < 3 * cfg.J ? tid : 3 * cfg.J - 1];
    const long long t = in.ts[b], p = t > 0 ? t - 1 : 0;
    // (clamped lanes instead of branches: a branch around a load ends in a join, where the compiler waits for it)
    const int spc = tid < 2 * 18 ? tid : 2 * 18 - 1, sp_which = spc / 18, sp_i = spc - 18 * sp_which;
    const long long sp_ts = sp_which ? p : t;
    const float* sp_src = sp_i < 3 ? in.rotation + 3 * sp_ts + sp_i
                        : (sp_i < 6 ? in.neck + 3 * sp_ts + (sp_i - 3)
                        : (sp_i < 9 ? in.jaw + 3 * sp_ts + (sp_i - 6)
                        : (sp_i < 15 ? in.eyes + 6 * sp_ts + (sp_i - 9) : in.translation + 3 * sp_ts + (sp_i - 15))));
    const float sp_v = *sp_src;
    const float* of0 = has_off ? in.offset + (size_t)blockIdx.x * in.offset_stride + 3 * jv0 : in.JT;
    const float of0x = of0[0], of0y = of0[1], of0z = of0[2];
    float e_reg = 0.f, e_smooth = 0.f, e_shape = 0.f;
    float part[3 * MAXJ];
#pragma unroll
    for (int o = 0; o < 3 * MAXJ; o++) part[o] = 0.f;
    // coefficient row + rest joints J = JT + JS betas: a lane's coefficient k and column k of JS in the same trip (rows of JS past 3J
    // re-read the last one into sums nobody reads; coefficients past NB re-read the last one, unused)
    for (int k = tid; k < cfg.Kp; k += FP_THREADS) {
        const int kc = k < NB ? k : NB - 1;
        const bool is_shape = kc < cfg.NS;
        const float a = *(is_shape ? in.shape + kc : in.expr + t * cfg.NE + (kc - cfg.NS));
        const float pr = *(is_shape ? in.shape + kc : in.expr + p * cfg.NE + (kc - cfg.NS));
        float js[3 * MAXJ];
#pragma unroll
        for (int o = 0; o < NO; o++) js[o] = in.JS[(size_t)(o < last ? o : last) * NB + kc];
        float v = 0.f;
        if (k < NB) {
            v = a;
            if (is_shape) {
                e_shape += v * v;
            } else {
                const float d = v - pr;
                e_reg += v * v;
                e_smooth += d * d;
            }
#pragma unroll
            for (int o = 0; o < NO; o++) part[o] += js[o] * v;
        }
        if (k < NB || k >= NB + cfg.P) row[k] = v;
    }
    e_reg = block_sum(e_reg, red);
    e_smooth = block_sum(e_smooth, red);
    e_shape = block_sum(e_shape, red);
    {
        if (has_off) {          // (+ J_regressor offset.)  J_regressor is sparse (a few hundred non-zero columns): compact list instead of a walk over all V
            if (tid < in.M) {
#pragma unroll
                for (int j = 0; j < NJ; j++) { part[3 * j] += wv0[j] * of0x; part[3 * j + 1] += wv0[j] * of0y; part[3 * j + 2] += wv0[j] * of0z; }
            }
            for (int m = tid + FP_THREADS; m < in.M; m += FP_THREADS) {        // (lists longer than the workgroup: the general loop)
                const int v = in.Jv[m];
                float wv[MAXJ];
#pragma unroll
                for (int j = 0; j < NJ; j++) wv[j] = in.Jw[(size_t)m * cfg.J + (j < cfg.J ? j : cfg.J - 1)];
                const float* of = in.offset + (size_t)blockIdx.x * in.offset_stride + 3 * v;
                const float o0 = of[0], o1 = of[1], o2 = of[2];
#pragma unroll
                for (int j = 0; j < NJ; j++) { part[3 * j] += wv[j] * o0; part[3 * j + 1] += wv[j] * o1; part[3 * j + 2] += wv[j] * o2; }
            }
        }
        __shared__ float wred[FP_THREADS / 64][3 * MAXJ];
        const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
        for (int o = 0; o < 3 * MAXJ; o++) {
            const float v = vhap_wave_sum(part[o]);
            if (lane == 0) wred[wave][o] = v;
        }
        __syncthreads();
        if (tid < 3 * cfg.J) {
            float a = jt_v;
#pragma unroll
            for (int w = 0; w < FP_THREADS / 64; w++) a += wred[w][tid];
            Jl[tid] = a;
        }
    }
    __shared__ float spose[2][3 * MAXJ + 3];   // [current | previous]: 15 pose values + translation
    if (tid < 2 * 18) spose[sp_which][sp_i] = sp_v;
    __syncthreads();
    if (tid < 3 * cfg.J) Jrest[(size_t)b * 3 * cfg.J + tid] = Jl[tid];
    if (tid < 3) transl[3 * b + tid] = spose[0][15 + tid];
    // ---- lanes 0..J-1: one joint each (Rodrigues, pose feature, ||R - I||^2) -- the same code once instead of J unrolled
    // copies on one lane: these kernels run ONE wave per frame through tens of KB of straight-line code, instruction fetch
    // of cold code was most of their time ----
    __shared__ float sR[MAXJ][9], sreg[MAXJ];
    if (tid < cfg.J) {
        const int j = tid;
        const Mat3 R = rodrigues(&spose[0][3 * j]);
        float fro = 0.f;
#pragma unroll
        for (int i = 0; i < 9; i++) {
            const float d = R.m[i] - ((i % 4 == 0) ? 1.0f : 0.0f);
            if (j > 0) row[NB + 9 * (j - 1) + i] = d;
            fro += d * d;
            sR[j][i] = R.m[i];
        }
        sreg[j] = fro;
    }
    __syncthreads();
    if (tid != 0) return;
    // ---- lane 0: kinematic chain, parameter energies ----
    float pose[3 * MAXJ], prev[3 * MAXJ];
#pragma unroll
    for (int i = 0; i < 15; i++) { pose[i] = spose[0][i]; prev[i] = spose[1][i]; }
    Mat3 GR[MAXJ];
    float Gt[MAXJ][3];
    float reg_R[MAXJ];
#pragma unroll
    for (int j = 0; j < (JT ? JT : MAXJ); j++) {
        if (!JT && j >= cfg.J) break;
        Mat3 R;
#pragma unroll
        for (int i = 0; i < 9; i++) R.m[i] = sR[j][i];
        reg_R[j] = sreg[j];
        const int par = JT ? flame_parent(j) : cfg.parents[j];
        if (j == 0) {
            GR[0] = R;
#pragma unroll
            for (int c = 0; c < 3; c++) Gt[0][c] = Jl[c];
        } else {
            float rel[3], rt[3];
#pragma unroll
            for (int c = 0; c < 3; c++) rel[c] = Jl[3 * j + c] - Jl[3 * par + c];
            GR[j] = mul(GR[par], R);
            matvec(GR[par], rel, rt);
#pragma unroll
            for (int c = 0; c < 3; c++) Gt[j][c] = rt[c] + Gt[par][c];
        }
        float gj[3];
        matvec(GR[j], Jl + 3 * j, gj);
        float* a = A + ((size_t)b * cfg.J + j) * 12;
#pragma unroll
        for (int i = 0; i < 3; i++) {
            a[4 * i] = GR[j].m[3 * i]; a[4 * i + 1] = GR[j].m[3 * i + 1]; a[4 * i + 2] = GR[j].m[3 * i + 2];
            a[4 * i + 3] = Gt[j][i] - gj[i];
        }
    }
    const float* w = cfg.w;
    const float iB = 1.0f / (float)cfg.B;
    float sp_t = 0.f, sp_r = 0.f, sj_n = 0.f, sj_j = 0.f, sj_e = 0.f;
#pragma unroll
    for (int c = 0; c < 3; c++) {
        const float dt = spose[0][15 + c] - spose[1][15 + c];
        const float dr = pose[c] - prev[c], dn = pose[3 + c] - prev[3 + c], dj = pose[6 + c] - prev[6 + c];
        const float d0 = pose[9 + c] - prev[9 + c], d1 = pose[12 + c] - prev[12 + c];
        sp_t += dt * dt; sp_r += dr * dr; sj_n += dn * dn; sj_j += dj * dj; sj_e += d0 * d0 + d1 * d1;
    }
    atomicAdd(&terms[0], (sp_t * w[VHAP_FW_SMOOTH_TRANS] + sp_r * w[VHAP_FW_SMOOTH_ROT]) * iB * (1.0f / 3.0f));
    atomicAdd(&terms[2], (sj_n * w[VHAP_FW_SMOOTH_NECK] + sj_j * w[VHAP_FW_SMOOTH_JAW]) * iB * (1.0f / 3.0f) +
                             sj_e * w[VHAP_FW_SMOOTH_EYES] * iB * (1.0f / 6.0f));
    // tracker.py:650-680 (J == 5: neck, jaw, eye_l, eye_r are joints 1..4); the (2B - 1) is the reference's own quirk
    if (cfg.J == 5) {
        const float dn = 1.0f / (9.0f * (float)(2 * cfg.B - 1));
        const float* jaw = pose + 6;
        float ed = 0.f;
#pragma unroll
        for (int c = 0; c < 3; c++) { const float d = pose[9 + c] - pose[12 + c]; ed += d * d; }
        ed *= iB * (1.0f / 3.0f);
        const float e = reg_R[1] * dn * w[VHAP_FW_REG_NECK] +
                        (reg_R[2] * dn + fmaxf(-jaw[0], 0.f) * iB * 10.0f + (jaw[1] * jaw[1] + jaw[2] * jaw[2]) * iB * 0.5f * 3.0f) * w[VHAP_FW_REG_JAW] +
                        ((reg_R[3] + reg_R[4]) * dn + 2.0f * ed) * w[VHAP_FW_REG_EYES];
        atomicAdd(&terms[1], e);
    }
    atomicAdd(&terms[3], e_reg * w[VHAP_FW_REG_EXPR] * iB / (float)cfg.NE);
    atomicAdd(&terms[4], e_smooth * w[VHAP_FW_SMOOTH_EXPR] * iB / (float)cfg.NE);
    if (b == 0) atomicAdd(&terms[5], e_shape * w[VHAP_FW_REG_SHAPE] / (float)cfg.NS);
}

struct FrameGrad {
    float *shape, *expr, *rotation, *translation, *neck, *jaw, *eyes, *offset;   // full-size, ACCUMULATED (any may be null)
};

template <int JT>
__global__ __launch_bounds__(FP_THREADS) void frame_prep_bwd_kernel(FrameCfg cfg, FrameIn in, const float* __restrict__ Jrest,
                                                                    const float* __restrict__ d_coef, const float* __restrict__ d_A,
                                                                    const float* __restrict__ d_transl,
                                                                    const float* __restrict__ d_terms, FrameGrad g, const CamBwdJob cam) {
    __shared__ float dJl[3 * MAXJ];
    if ((int)blockIdx.x == cfg.B) {      // the workgroup behind the frames': the camera backward (launched only with a job)
        __shared__ float cam_part[64];
        camera_focal_bwd_block(cam, cam_part);
        return;
    }
    const int b = blockIdx.x, tid = threadIdx.x;
    const int NB = cfg.NS + cfg.NE;
    const long long t = in.ts[b], p = t > 0 ? t - 1 : 0;
    const float* w = cfg.w;
    const float iB = 1.0f / (float)cfg.B;
    float dt_[6];
#pragma unroll
    for (int i = 0; i < 6; i++) dt_[i] = d_terms ? d_terms[i] : 0.f;
    const float* drow = d_coef ? d_coef + (size_t)b * cfg.Kp : nullptr;
    // stage every small per-frame input in LDS with parallel loads; lane 0 then runs the serial algebra out of LDS
    __shared__ float spose[2][18], sJ[3 * MAXJ], sdA[12 * MAXJ], sdpf[9 * MAXJ], sdt[3];
    // (ONE batch of loads -- clamped lanes and stand-in addresses instead of branches, see frame_prep_fwd_kernel -- then the LDS writes:
    // as five branchy staging loops this was nine dependent round trips before the first flop)
    {
        const int spc = tid < 36 ? tid : 35, which = spc / 18, i = spc - 18 * which;
        const long long tt = which ? p : t;
        const float* sp_src = i < 3 ? in.rotation + 3 * tt + i
                            : (i < 6 ? in.neck + 3 * tt + (i - 3)
                            : (i < 9 ? in.jaw + 3 * tt + (i - 6)
                            : (i < 15 ? in.eyes + 6 * tt + (i - 9) : in.translation + 3 * tt + (i - 15))));
        const float* jr = Jrest + (size_t)b * 3 * cfg.J;                   // (also the stand-in address of an absent input)
        const int nA = 12 * cfg.J;                                         // <= 96 < FP_THREADS, like 3 J and P = 9 (J - 1)
        const float r_sp = *sp_src;
        const float r_J = jr[tid < 3 * cfg.J ? tid : 3 * cfg.J - 1];
        const float r_dA = *(d_A ? d_A + (size_t)b * nA + (tid < nA ? tid : nA - 1) : jr);
        const float r_pf = *(drow && cfg.P > 0 ? drow + NB + (tid < cfg.P ? tid : cfg.P - 1) : jr);
        const float r_dt = *(d_transl ? d_transl + 3 * b + (tid < 3 ? tid : 2) : jr);
        if (tid < 36) spose[which][i] = r_sp;
        if (tid < 3 * cfg.J) sJ[tid] = r_J;
        if (tid < nA) sdA[tid] = d_A ? r_dA : 0.f;
        if (tid < cfg.P) sdpf[tid] = drow ? r_pf : 0.f;
        if (tid < 3) sdt[tid] = d_transl ? r_dt : 0.f;
    }
    __syncthreads();
    // lanes 0..J-1: R_j (one Rodrigues per lane, SIMD) -> LDS;  lane 0: chain forward + reverse -> dR_j in LDS;
    // lanes 0..J-1: reverse of Rodrigues + the direct pose terms + the scatter (see frame_prep_fwd_kernel for why)
    __shared__ float sR[MAXJ][9], sdR[MAXJ][9];
    if (tid < cfg.J) {
        const Mat3 R = rodrigues(&spose[0][3 * tid]);
#pragma unroll
        for (int i = 0; i < 9; i++) sR[tid][i] = R.m[i];
    }
    __syncthreads();
    if (tid == 0) {
        const float* Jl = sJ;
        Mat3 R[MAXJ], GR[MAXJ], dGR[MAXJ];
        float dGt[MAXJ][3], dJ[3 * MAXJ];
#pragma unroll
        for (int j = 0; j < (JT ? JT : MAXJ); j++) {
            if (!JT && j >= cfg.J) break;
#pragma unroll
            for (int i = 0; i < 9; i++) R[j].m[i] = sR[j][i];
            GR[j] = j == 0 ? R[0] : mul(GR[JT ? (j ? flame_parent(j) : 0) : cfg.parents[j]], R[j]);
#pragma unroll
            for (int c = 0; c < 3; c++) dJ[3 * j + c] = 0.f;
        }
        // A_j = [GR_j | Gt_j - GR_j J_j]
#pragma unroll
        for (int j = 0; j < (JT ? JT : MAXJ); j++) {
            if (!JT && j >= cfg.J) break;
            const float* da = sdA + 12 * j;
            float dta[3] = {0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 3; i++) {
                dta[i] = da[4 * i + 3];
#pragma unroll
                for (int c = 0; c < 3; c++) dGR[j].m[3 * i + c] = da[4 * i + c] - dta[i] * Jl[3 * j + c];
                dGt[j][i] = dta[i];
            }
            float v[3];
            matvec_t(GR[j], dta, v);
#pragma unroll
            for (int c = 0; c < 3; c++) dJ[3 * j + c] -= v[c];
        }
#pragma unroll
        for (int jj = 0; jj < (JT ? JT : MAXJ); jj++) {
            const int j = (JT ? JT : cfg.J) - 1 - jj;
            if (!JT && j < 0) break;
            Mat3 dR;
            if (j > 0) {
                const int par = JT ? flame_parent(j > 0 ? j : 1) : cfg.parents[j];
                float rel[3], drel[3];
#pragma unroll
                for (int c = 0; c < 3; c++) rel[c] = Jl[3 * j + c] - Jl[3 * par + c];
                const Mat3 t1 = mul_nt(dGR[j], R[j]);
#pragma unroll
                for (int i = 0; i < 3; i++)
#pragma unroll
                    for (int c = 0; c < 3; c++) dGR[par].m[3 * i + c] += t1.m[3 * i + c] + dGt[j][i] * rel[c];
                dR = mul_tn(GR[par], dGR[j]);
                matvec_t(GR[par], dGt[j], drel);
#pragma unroll
                for (int c = 0; c < 3; c++) {
                    dGt[par][c] += dGt[j][c];
                    dJ[3 * j + c] += drel[c];
                    dJ[3 * par + c] -= drel[c];
                }
            } else {
                dR = dGR[0];
#pragma unroll
                for (int c = 0; c < 3; c++) dJ[c] += dGt[0][c];
            }
#pragma unroll
            for (int i = 0; i < 9; i++) sdR[j][i] = dR.m[i];
        }
        for (int o = 0; o < 3 * cfg.J; o++) dJl[o] = dJ[o];
    }
    __syncthreads();
    if (tid < cfg.J) {
        const int j = tid;
        const float* pj = &spose[0][3 * j];
        const float* qj = &spose[1][3 * j];
        Mat3 dR;
#pragma unroll
        for (int i = 0; i < 9; i++) dR.m[i] = sdR[j][i];
        if (j > 0) {
            // pose feature (R_j - I) feeds the corrective blendshapes; ||I - R_j||^2 is the joint regulariser
            const float dn9 = 1.0f / (9.0f * (float)(2 * cfg.B - 1));
            float wj = 0.f;
            if (cfg.J == 5) wj = (j == 1 ? w[VHAP_FW_REG_NECK] : (j == 2 ? w[VHAP_FW_REG_JAW] : w[VHAP_FW_REG_EYES])) * dn9 * dt_[1];
#pragma unroll
            for (int i = 0; i < 9; i++) {
                dR.m[i] += sdpf[9 * (j - 1) + i];
                dR.m[i] += 2.0f * wj * (sR[j][i] - ((i % 4 == 0) ? 1.0f : 0.0f));
            }
        }
        float dp[3] = {0.f, 0.f, 0.f};
        rodrigues_bwd(pj, dR, dp);
        // direct terms on the pose vectors (joint order: root rotation, neck, jaw, left eye, right eye)
        const float k3 = 2.0f * iB * (1.0f / 3.0f);
        float ws = 0.f;     // smoothness weight x upstream gradient of this joint's pose vector
        if (j == 0) ws = k3 * w[VHAP_FW_SMOOTH_ROT] * dt_[0];
        else if (j == 1) ws = k3 * w[VHAP_FW_SMOOTH_NECK] * dt_[2];
        else if (j == 2) ws = k3 * w[VHAP_FW_SMOOTH_JAW] * dt_[2];
        else ws = 0.5f * k3 * w[VHAP_FW_SMOOTH_EYES] * dt_[2];
#pragma unroll
        for (int c = 0; c < 3; c++) dp[c] += ws * (pj[c] - qj[c]);
        if (cfg.J == 5) {
            const float wjaw = w[VHAP_FW_REG_JAW] * dt_[1], weye = w[VHAP_FW_REG_EYES] * dt_[1];
            if (j == 2) {
                if (pj[0] < 0.f) dp[0] -= 10.0f * iB * wjaw;
                dp[1] += 3.0f * iB * wjaw * pj[1];      // d/dx of 3 * mean over (B, 2) of x^2
                dp[2] += 3.0f * iB * wjaw * pj[2];
            } else if (j >= 3) {
                const float* other = &spose[0][j == 3 ? 12 : 9];
#pragma unroll
                for (int c = 0; c < 3; c++) dp[c] += 2.0f * weye * 2.0f * iB * (1.0f / 3.0f) * (pj[c] - other[c]);
            }
        }
        float* dst = j == 0 ? g.rotation : (j == 1 ? g.neck : (j == 2 ? g.jaw : g.eyes));
        if (dst && j < 5) {
            const size_t o = j < 3 ? (size_t)3 * t : (size_t)6 * t + (j == 4 ? 3 : 0);
#pragma unroll
            for (int c = 0; c < 3; c++) atomicAdd(&dst[o + c], dp[c]);
        }
        if (j == 0 && g.translation) {
#pragma unroll
            for (int c = 0; c < 3; c++) {
                const float dtr = sdt[c] + k3 * w[VHAP_FW_SMOOTH_TRANS] * dt_[0] * (spose[0][15 + c] - spose[1][15 + c]);
                atomicAdd(&g.translation[3 * t + c], dtr);
            }
        }
    }
    __syncthreads();
    // betas: blend-kernel gradient + joint regression + L2 / smoothness
    const float ke = 2.0f * iB / (float)cfg.NE;
    {
        const int NO = JT ? 3 * JT : 3 * MAXJ, last = 3 * cfg.J - 1;
        float dj[3 * MAXJ];
#pragma unroll
        for (int o = 0; o < NO; o++) dj[o] = o <= last ? dJl[o] : 0.f;
        for (int k = tid; k < NB; k += FP_THREADS) {
            // all of this slice's loads first (compile-time trip count, unconditional: a run-time `o < 3J` loop is load -> wait -> fma 15 times)
            const bool is_shape = k < cfg.NS;
            const int e = is_shape ? 0 : k - cfg.NS;
            float js[3 * MAXJ];
#pragma unroll
            for (int o = 0; o < NO; o++) js[o] = in.JS[(size_t)(o < last ? o : last) * NB + k];
            const float r_d = *(drow ? drow + k : in.JS + k);                                       // (stand-in: JS has NB columns)
            const float r_a = *(is_shape ? in.shape + k : in.expr + t * cfg.NE + e);
            const float r_p = *(is_shape ? in.shape + k : in.expr + p * cfg.NE + e);
            float d = drow ? r_d : 0.f;
#pragma unroll
            for (int o = 0; o < NO; o++) d += js[o] * dj[o];
            if (is_shape) {
                if (b == 0) d += 2.0f * w[VHAP_FW_REG_SHAPE] * dt_[5] * r_a / (float)cfg.NS;
                if (g.shape) atomicAdd(&g.shape[k], d);
            } else if (g.expr) {
                const float x = r_a;
                d += ke * (w[VHAP_FW_REG_EXPR] * dt_[3] * x + w[VHAP_FW_SMOOTH_EXPR] * dt_[4] * (x - r_p));
                atomicAdd(&g.expr[t * cfg.NE + e], d);
            }
        }
        if (g.offset && in.offset) {
            const int NJ = JT ? JT : MAXJ;
            for (int m = tid; m < in.M; m += FP_THREADS) {
                const int v = in.Jv[m];
                float wv[MAXJ];
#pragma unroll
                for (int j = 0; j < NJ; j++) wv[j] = in.Jw[(size_t)m * cfg.J + (j < cfg.J ? j : cfg.J - 1)];
                float a[3] = {0.f, 0.f, 0.f};
#pragma unroll
                for (int j = 0; j < NJ; j++) { a[0] += wv[j] * dj[3 * j]; a[1] += wv[j] * dj[3 * j + 1]; a[2] += wv[j] * dj[3 * j + 2]; }
                float* go = g.offset + (size_t)blockIdx.x * in.offset_stride + 3 * v;      // (per-frame offsets: this frame's row of the gradient)
                atomicAdd(&go[0], a[0]); atomicAdd(&go[1], a[1]); atomicAdd(&go[2], a[2]);
            }
        }
    }
}

bool make_cfg(FrameCfg& c, int B, int Bp, int N, int NS, int NE, int J, int Kp, int V, const int32_t* parents, const float* weights) {
    if (B <= 0 || Bp < B || N <= 0 || NS < 0 || NE < 0 || NS + NE < 1 || J <= 0 || J > MAXJ || NS + NE > 1024 || NS + NE + 9 * (J - 1) > Kp || !parents) return false;
    c.B = B; c.Bp = Bp; c.N = N; c.NS = NS; c.NE = NE; c.J = J; c.P = 9 * (J - 1); c.Kp = Kp; c.V = V;
    for (int j = 0; j < MAXJ; j++) c.parents[j] = j < J ? parents[j] : -1;
    for (int j = 1; j < J; j++)
        if (c.parents[j] < 0 || c.parents[j] >= j) return false;
    for (int i = 0; i < 12; i++) c.w[i] = weights ? weights[i] : 0.f;
    return true;
}

}  // namespace

static int frame_prep_fwd_run(const int64_t* timesteps, const float* shape, const float* expr, const float* rotation,
                              const float* translation, const float* neck, const float* jaw, const float* eyes,
                              const float* JT, const float* JS, const int32_t* jreg_idx, const float* jreg_w, int jreg_n,
                              const float* static_offset, const int32_t* parents, const float* weights, int B, int Bp, int N,
                              int NS, int NE, int J, int Kp, int V, float* coef, float* A, float* transl, float* Jrest, float* terms,
                              int call_flags, vhap_stream_t stream, const CamFwdJob& cam) {
    if (!timesteps || !shape || !expr || !rotation || !translation || !neck || !jaw || !eyes || !JT || !JS || !coef || !A ||
        !transl || !Jrest || !terms)
        return VHAP_E_NULLPTR;
    if (static_offset && jreg_n > 0 && (!jreg_idx || !jreg_w)) return VHAP_E_NULLPTR;
    FrameCfg cfg;
    if (!make_cfg(cfg, B, Bp, N, NS, NE, J, Kp, V, parents, weights)) return VHAP_E_BADDIM;
    FrameIn in{reinterpret_cast<const long long*>(timesteps), shape, expr, rotation, translation, neck, jaw, eyes, JT, JS, jreg_w, static_offset, jreg_idx, jreg_n,
               (call_flags & VHAP_CALL_OFFSET_PER_FRAME) ? 3ll * V : 0ll};
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(terms, 6 * sizeof(float), st);
    const bool flame_tree = J == 5 && parents[1] == 0 && parents[2] == 1 && parents[3] == 1 && parents[4] == 1;
    const int nwg = Bp + (cam.focal ? 1 : 0);
    if (flame_tree) frame_prep_fwd_kernel<5><<<nwg, FP_THREADS, 0, st>>>(cfg, in, coef, A, transl, Jrest, terms, cam);
    else frame_prep_fwd_kernel<0><<<nwg, FP_THREADS, 0, st>>>(cfg, in, coef, A, transl, Jrest, terms, cam);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_frame_prep_fwd(const int64_t* timesteps, const float* shape, const float* expr, const float* rotation,
                                   const float* translation, const float* neck, const float* jaw, const float* eyes,
                                   const float* JT, const float* JS, const int32_t* jreg_idx, const float* jreg_w, int jreg_n,
                                   const float* static_offset, const int32_t* parents, const float* weights, int B, int Bp, int N,
                                   int NS, int NE, int J, int Kp, int V, float* coef, float* A, float* transl, float* Jrest, float* terms,
                                   int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    return frame_prep_fwd_run(timesteps, shape, expr, rotation, translation, neck, jaw, eyes, JT, JS, jreg_idx, jreg_w, jreg_n, static_offset, parents,
                              weights, B, Bp, N, NS, NE, J, Kp, V, coef, A, transl, Jrest, terms, call_flags, stream, CamFwdJob{});
}

extern "C" int vhap_frame_prep_fwd_camera(const int64_t* timesteps, const float* shape, const float* expr, const float* rotation,
                                          const float* translation, const float* neck, const float* jaw, const float* eyes,
                                          const float* JT, const float* JS, const int32_t* jreg_idx, const float* jreg_w, int jreg_n,
                                          const float* static_offset, const int32_t* parents, const float* weights, int B, int Bp, int N,
                                          int NS, int NE, int J, int Kp, int V, float* coef, float* A, float* transl, float* Jrest,
                                          float* terms, int call_flags, const float* focal_length, float focal_scale, float cx, float cy,
                                          const float* RT, int RT_batched, int H, int W, float near_plane, float far_plane, float* mvp,
                                          vhap_stream_t stream) {
    VHAP_ENTER();
    if (!focal_length || !RT || !mvp) return VHAP_E_NULLPTR;
    if (H <= 0 || W <= 0 || !(far_plane > near_plane)) return VHAP_E_BADDIM;
    const CamFwdJob cam{focal_length, RT, focal_scale, cx, cy, (float)H, (float)W, near_plane, far_plane, B, RT_batched ? 12 : 0, mvp};
    return frame_prep_fwd_run(timesteps, shape, expr, rotation, translation, neck, jaw, eyes, JT, JS, jreg_idx, jreg_w, jreg_n, static_offset, parents,
                              weights, B, Bp, N, NS, NE, J, Kp, V, coef, A, transl, Jrest, terms, call_flags, stream, cam);
}

static int frame_prep_bwd_run(const int64_t* timesteps, const float* shape, const float* expr, const float* rotation,
                              const float* translation, const float* neck, const float* jaw, const float* eyes,
                              const float* JS, const int32_t* jreg_idx, const float* jreg_w, int jreg_n, const float* static_offset,
                              const int32_t* parents, const float* weights, const float* Jrest, const float* d_coef, const float* d_A,
                              const float* d_transl, const float* d_terms, int B, int Bp, int N, int NS, int NE, int J, int Kp,
                              int V, float* g_shape, float* g_expr, float* g_rotation, float* g_translation, float* g_neck,
                              float* g_jaw, float* g_eyes, float* g_offset, int call_flags, vhap_stream_t stream, const CamBwdJob& cam) {
    if (!timesteps || !shape || !expr || !rotation || !translation || !neck || !jaw || !eyes || !JS || !Jrest) return VHAP_E_NULLPTR;
    if (g_offset && (!static_offset || (jreg_n > 0 && (!jreg_idx || !jreg_w)))) return VHAP_E_NULLPTR;
    FrameCfg cfg;
    if (!make_cfg(cfg, B, Bp, N, NS, NE, J, Kp, V, parents, weights)) return VHAP_E_BADDIM;
    FrameIn in{reinterpret_cast<const long long*>(timesteps), shape, expr, rotation, translation, neck, jaw, eyes, nullptr, JS, jreg_w, static_offset, jreg_idx, jreg_n,
               (call_flags & VHAP_CALL_OFFSET_PER_FRAME) ? 3ll * V : 0ll};
    FrameGrad g{g_shape, g_expr, g_rotation, g_translation, g_neck, g_jaw, g_eyes, g_offset};
    const bool flame_tree = J == 5 && parents[1] == 0 && parents[2] == 1 && parents[3] == 1 && parents[4] == 1;
    const int nwg = B + (cam.RT ? 1 : 0);
    if (flame_tree) frame_prep_bwd_kernel<5><<<nwg, FP_THREADS, 0, vhap_stream(stream)>>>(cfg, in, Jrest, d_coef, d_A, d_transl, d_terms, g, cam);
    else frame_prep_bwd_kernel<0><<<nwg, FP_THREADS, 0, vhap_stream(stream)>>>(cfg, in, Jrest, d_coef, d_A, d_transl, d_terms, g, cam);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_frame_prep_bwd(const int64_t* timesteps, const float* shape, const float* expr, const float* rotation,
                                   const float* translation, const float* neck, const float* jaw, const float* eyes,
                                   const float* JS, const int32_t* jreg_idx, const float* jreg_w, int jreg_n, const float* static_offset,
                                   const int32_t* parents, const float* weights, const float* Jrest, const float* d_coef, const float* d_A,
                                   const float* d_transl, const float* d_terms, int B, int Bp, int N, int NS, int NE, int J, int Kp,
                                   int V, float* g_shape, float* g_expr, float* g_rotation, float* g_translation, float* g_neck,
                                   float* g_jaw, float* g_eyes, float* g_offset, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    return frame_prep_bwd_run(timesteps, shape, expr, rotation, translation, neck, jaw, eyes, JS, jreg_idx, jreg_w, jreg_n, static_offset, parents,
                              weights, Jrest, d_coef, d_A, d_transl, d_terms, B, Bp, N, NS, NE, J, Kp, V, g_shape, g_expr, g_rotation,
                              g_translation, g_neck, g_jaw, g_eyes, g_offset, call_flags, stream, CamBwdJob{});
}

extern "C" int vhap_frame_prep_bwd_camera(const int64_t* timesteps, const float* shape, const float* expr, const float* rotation,
                                          const float* translation, const float* neck, const float* jaw, const float* eyes,
                                          const float* JS, const int32_t* jreg_idx, const float* jreg_w, int jreg_n,
                                          const float* static_offset, const int32_t* parents, const float* weights, const float* Jrest,
                                          const float* d_coef, const float* d_A, const float* d_transl, const float* d_terms, int B, int Bp,
                                          int N, int NS, int NE, int J, int Kp, int V, float* g_shape, float* g_expr, float* g_rotation,
                                          float* g_translation, float* g_neck, float* g_jaw, float* g_eyes, float* g_offset, int call_flags,
                                          const float* RT, const float* d_mvp, int RT_batched, int H, int W, float focal_scale,
                                          float* d_focal_accum, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!RT || !d_mvp || !d_focal_accum) return VHAP_E_NULLPTR;
    if (H <= 0 || W <= 0) return VHAP_E_BADDIM;
    const CamBwdJob cam{RT, d_mvp, (float)H, (float)W, focal_scale, B, RT_batched ? 12 : 0, d_focal_accum};
    return frame_prep_bwd_run(timesteps, shape, expr, rotation, translation, neck, jaw, eyes, JS, jreg_idx, jreg_w, jreg_n, static_offset, parents,
                              weights, Jrest, d_coef, d_A, d_transl, d_terms, B, Bp, N, NS, NE, J, Kp, V, g_shape, g_expr, g_rotation,
                              g_translation, g_neck, g_jaw, g_eyes, g_offset, call_flags, stream, cam);
}

// ------------------------------------------------------------------------------------------------------------
// Camera and landmark energy.
//   camera_fwd : mvp = P(K) [RT; 0 0 0 1]  (render_nvdiffrast.py:102-160, OpenGL-style projection, near/far planes)
//   landmark   : barycentric landmarks on the posed mesh (lbs.vertices2landmarks, vhap/model/lbs.py:60-98), projection to
//                NDC with the y flip, confidence-weighted L1 against the detected 2D landmarks (tracker.py:347-389).
// ------------------------------------------------------------------------------------------------------------
namespace {

// focal != null: the uncalibrated camera of tracker.py:148-157 -- K = (f, f, cx, cy) with f = focal[0] * fscale, built here instead of
// by separate elementwise launches
__global__ __launch_bounds__(64) void camera_fwd_kernel(const float* __restrict__ K, const float* __restrict__ RT, int B, int kstride,
                                                        int rtstride, float h, float w, float near, float far,
                                                        float* __restrict__ mvp, const float* __restrict__ focal, float fscale, float cx,
                                                        float cy) {
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= B * 4) return;
    const int b = i >> 2, c = i & 3;
    float kf[4];
    const float* k = K + (size_t)b * kstride;
    if (focal) {
        kf[0] = kf[1] = focal[0] * fscale; kf[2] = cx; kf[3] = cy;
        k = kf;
    }
    const float* rt = RT + (size_t)b * rtstride;
    const float mv0 = rt[c], mv1 = rt[4 + c], mv2 = rt[8 + c], mv3 = c == 3 ? 1.0f : 0.0f;
    float* m = mvp + (size_t)b * 16;
    m[c] = k[0] * 2.0f / w * mv0 + (w - 2.0f * k[2]) / w * mv2;
    m[4 + c] = k[1] * 2.0f / h * mv1 + (h - 2.0f * k[3]) / h * mv2;
    m[8 + c] = -(far + near) / (far - near) * mv2 + (-2.0f * far * near / (far - near)) * mv3;
    m[12 + c] = -mv2;
}

__global__ __launch_bounds__(64) void camera_bwd_kernel(const float* __restrict__ RT, const float* __restrict__ d_mvp, int B, int rtstride,
                                                        float h, float w, float* __restrict__ d_K) {
    const int b = blockIdx.x * 64 + threadIdx.x;
    if (b >= B) return;
    const float* rt = RT + (size_t)b * rtstride;
    const float* d = d_mvp + (size_t)b * 16;
    float g[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < 4; c++) {
        g[0] += d[c] * rt[c];
        g[2] += d[c] * rt[8 + c];
        g[1] += d[4 + c] * rt[4 + c];
        g[3] += d[4 + c] * rt[8 + c];
    }
    float* o = d_K + (size_t)b * 4;
    o[0] = g[0] * 2.0f / w; o[1] = g[1] * 2.0f / h; o[2] = -g[2] * 2.0f / w; o[3] = -g[3] * 2.0f / h;
}

struct LmkCfg {
    int B, V, L, L2;          // L landmarks on the mesh (>= l1), L2 = row stride (landmarks) of lmk2d
    int l0, l1;               // energy over landmarks [l0, l1)
    int boost0, boost1;       // confidence x boost in [boost0, boost1)
    float boost, H, W;
};

// one workgroup (128 lanes) per frame
__global__ __launch_bounds__(128) void landmark_kernel(LmkCfg c, const float* __restrict__ verts, const int* __restrict__ lmk_vidx,
                                                       const float* __restrict__ lmk_bary, const float* __restrict__ mvp,
                                                       const float* __restrict__ lmk2d, float* __restrict__ lmk3d,
                                                       float* __restrict__ energy, const float* __restrict__ d_energy,
                                                       float* __restrict__ d_verts, float* __restrict__ d_mvp) {
    __shared__ float red[2];
    __shared__ float dm[16];
    const int b = blockIdx.x, l = threadIdx.x;
    const bool bwd = d_energy != nullptr;
    if (bwd && l < 16) dm[l] = 0.f;
    if (bwd) __syncthreads();
    float e = 0.f;
    if (l < c.L) {
        const float* Vb = verts + (size_t)b * c.V * 3;
        const int i0 = lmk_vidx[3 * l], i1 = lmk_vidx[3 * l + 1], i2 = lmk_vidx[3 * l + 2];
        const float w0 = lmk_bary[3 * l], w1 = lmk_bary[3 * l + 1], w2 = lmk_bary[3 * l + 2];
        float p[3];
#pragma unroll
        for (int k = 0; k < 3; k++) p[k] = (Vb[3 * i0 + k] * w0 + Vb[3 * i1 + k] * w1) + Vb[3 * i2 + k] * w2;
        if (lmk3d && !bwd) {
#pragma unroll
            for (int k = 0; k < 3; k++) lmk3d[((size_t)b * c.L + l) * 3 + k] = p[k];
        }
        if (l >= c.l0 && l < c.l1) {
            const float* M = mvp + (size_t)b * 16;
            const float cx = M[0] * p[0] + M[1] * p[1] + M[2] * p[2] + M[3];
            const float cy = M[4] * p[0] + M[5] * p[1] + M[6] * p[2] + M[7];
            const float cw = M[12] * p[0] + M[13] * p[1] + M[14] * p[2] + M[15];
            const float iw = 1.0f / cw;
            const float px = cx * iw, py = -cy * iw;
            const float* g = lmk2d + ((size_t)b * c.L2 + l) * 3;
            const float gu = 2.0f * (g[0] - c.W * 0.5f) / c.W, gv = 2.0f * (g[1] - c.H * 0.5f) / c.H;
            const float conf = g[2] * ((l >= c.boost0 && l < c.boost1) ? c.boost : 1.0f);
            const float du = gu - px, dv = gv - py;
            e = (fabsf(du) + fabsf(dv)) * conf;
            if (bwd) {
                const float s = conf * d_energy[0] / (float)(c.B * (c.l1 - c.l0));
                // dE/dpx = -sign(du) * s ; torch's abs has sign(0) = 0
                const float gpx = -(du > 0.f ? 1.f : (du < 0.f ? -1.f : 0.f)) * s;
                const float gpy = -(dv > 0.f ? 1.f : (dv < 0.f ? -1.f : 0.f)) * s;
                const float gcx = gpx * iw, gcy = -gpy * iw;
                const float gcw = -(gpx * px + gpy * py) * iw;
                float gp[3];
#pragma unroll
                for (int k = 0; k < 3; k++) gp[k] = M[k] * gcx + M[4 + k] * gcy + M[12 + k] * gcw;
                if (d_mvp) {
#pragma unroll
                    for (int k = 0; k < 4; k++) {
                        const float pk = k < 3 ? p[k] : 1.0f;
                        atomicAdd(&dm[k], gcx * pk); atomicAdd(&dm[4 + k], gcy * pk); atomicAdd(&dm[12 + k], gcw * pk);
                    }
                }
                if (d_verts) {
                    float* D = d_verts + (size_t)b * c.V * 3;
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        atomicAdd(&D[3 * i0 + k], gp[k] * w0); atomicAdd(&D[3 * i1 + k], gp[k] * w1); atomicAdd(&D[3 * i2 + k], gp[k] * w2);
                    }
                }
            }
        }
    }
    if (!bwd) {
        e = vhap_wave_sum(e);
        if ((l & 63) == 0) red[l >> 6] = e;
        __syncthreads();
        if (l == 0) atomicAdd(energy, (red[0] + red[1]) / (float)(c.B * (c.l1 - c.l0)));
    } else if (d_mvp) {
        __syncthreads();
        if (l < 16) d_mvp[(size_t)b * 16 + l] = dm[l];
    }
}

}  // namespace

extern "C" int vhap_camera_fwd(const float* K, const float* RT, int B, int K_batched, int RT_batched, int H, int W, float near_plane,
                               float far_plane, float* mvp, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!K || !RT || !mvp) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || !(far_plane > near_plane)) return VHAP_E_BADDIM;
    camera_fwd_kernel<<<vhap_cdiv(B * 4, 64), 64, 0, vhap_stream(stream)>>>(K, RT, B, K_batched ? 4 : 0, RT_batched ? 12 : 0, (float)H, (float)W,
                                                                            near_plane, far_plane, mvp, nullptr, 0.f, 0.f, 0.f);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_camera_focal_fwd(const float* focal_length, float focal_scale, float cx, float cy, const float* RT, int B, int RT_batched,
                                     int H, int W, float near_plane, float far_plane, float* mvp, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!focal_length || !RT || !mvp) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || !(far_plane > near_plane)) return VHAP_E_BADDIM;
    camera_fwd_kernel<<<vhap_cdiv(B * 4, 64), 64, 0, vhap_stream(stream)>>>(nullptr, RT, B, 0, RT_batched ? 12 : 0, (float)H, (float)W, near_plane,
                                                                            far_plane, mvp, focal_length, focal_scale, cx, cy);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// camera backward of the monocular case in ONE launch: d(focal_length) += scale * sum_b (dK[b].fx + dK[b].fy)  (K = (f, f, cx, cy), f =
// focal_length * scale: until round 3 vhap_camera_bwd + a summing kernel, two launches on the tail of the step's geometry chain).  The per-frame values
// meet in LDS and lane 0 adds them in frame order -- the same sum, in the same order, as the two-launch form.
__global__ __launch_bounds__(64) void camera_focal_bwd_kernel(const float* __restrict__ RT, const float* __restrict__ d_mvp, int B, int rtstride,
                                                              float h, float w, float scale, float* __restrict__ d_focal) {
    __shared__ float part[64];
    float s = 0.f;
    for (int b0 = 0; b0 < B; b0 += 64) {
        const int b = b0 + (int)threadIdx.x;
        float v = 0.f;
        if (b < B) {
            const float* rt = RT + (size_t)b * rtstride;
            const float* d = d_mvp + (size_t)b * 16;
            float g0 = 0.f, g1 = 0.f;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                g0 += d[c] * rt[c];
                g1 += d[4 + c] * rt[4 + c];
            }
            v = g0 * 2.0f / w + g1 * 2.0f / h;
        }
        part[threadIdx.x] = v;
        __syncthreads();
        if (threadIdx.x == 0)
            for (int k = 0; k < min(64, B - b0); k++) s += part[k];
        __syncthreads();
    }
    if (threadIdx.x == 0) d_focal[0] += s * scale;
}

extern "C" int vhap_camera_focal_bwd(const float* RT, const float* d_mvp, int B, int RT_batched, int H, int W, float scale,
                                     float* d_focal_accum, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!RT || !d_mvp || !d_focal_accum) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    camera_focal_bwd_kernel<<<1, 64, 0, vhap_stream(stream)>>>(RT, d_mvp, B, RT_batched ? 12 : 0, (float)H, (float)W, scale, d_focal_accum);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_camera_bwd(const float* RT, const float* d_mvp, int B, int RT_batched, int H, int W, float* d_K, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!RT || !d_mvp || !d_K) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    camera_bwd_kernel<<<vhap_cdiv(B, 64), 64, 0, vhap_stream(stream)>>>(RT, d_mvp, B, RT_batched ? 12 : 0, (float)H, (float)W, d_K);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

static bool make_lmk_cfg(LmkCfg& c, int B, int V, int L, int L2, int l0, int l1, int boost0, int boost1, float boost, int H, int W) {
    if (B <= 0 || V <= 0 || L <= 0 || L > 128 || L2 < l1 || l0 < 0 || l1 > L || l0 >= l1 || H <= 0 || W <= 0) return false;
    c = LmkCfg{B, V, L, L2, l0, l1, boost0, boost1, boost, (float)H, (float)W};
    return true;
}

extern "C" int vhap_landmark_fwd(const float* verts, const int32_t* lmk_vidx, const float* lmk_bary, const float* mvp,
                                 const float* lmk2d, int B, int V, int L, int L2, int l0, int l1, int boost0, int boost1, float boost,
                                 int H, int W, float* lmk3d, float* energy, int call_flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !lmk_vidx || !lmk_bary || !mvp || !lmk2d || !energy) return VHAP_E_NULLPTR;
    LmkCfg c;
    if (!make_lmk_cfg(c, B, V, L, L2, l0, l1, boost0, boost1, boost, H, W)) return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    VHAP_ZERO_ACC(energy, sizeof(float), st);
    landmark_kernel<<<B, 128, 0, st>>>(c, verts, lmk_vidx, lmk_bary, mvp, lmk2d, lmk3d, energy, nullptr, nullptr, nullptr);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_landmark_bwd(const float* verts, const int32_t* lmk_vidx, const float* lmk_bary, const float* mvp,
                                 const float* lmk2d, const float* d_energy, int B, int V, int L, int L2, int l0, int l1, int boost0,
                                 int boost1, float boost, int H, int W, float* d_verts, float* d_mvp, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!verts || !lmk_vidx || !lmk_bary || !mvp || !lmk2d || !d_energy) return VHAP_E_NULLPTR;
    LmkCfg c;
    if (!make_lmk_cfg(c, B, V, L, L2, l0, l1, boost0, boost1, boost, H, W)) return VHAP_E_BADDIM;
    landmark_kernel<<<B, 128, 0, vhap_stream(stream)>>>(c, verts, lmk_vidx, lmk_bary, mvp, lmk2d, nullptr, nullptr, d_energy, d_verts, d_mvp);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
