// Cluster-wise colour disturbance for gfx950 (vhap/util/render_nvdiffrast.py:424-460).
//
// Reference semantics: every pixel belongs to a colour cluster cid = fid2cid[triangle id + 1]
// (0 = background, 1 = face in no cluster).  For each cluster i != 1, with probability `rate` a pixel is
// replaced by a pixel drawn uniformly from the pool of ALL pixels of that cluster in the whole batch
// (background pixels draw from the background image); the drawn colour is detached.  The reference does
// this with a 9-iteration Python loop of boolean-mask indexing (one host sync each) and randint.
//
// Here: a deterministic counting sort of the pixel ids by cluster (row-major order inside a cluster, i.e.
// exactly the order of the reference's boolean-mask gather) in three passes -- per-block cluster
// histograms (wave ballots), a one-workgroup scan over blocks, and a scatter using wave prefix ranks --
// followed by one gather pass.  No host sync, no atomics, graph-capturable.  Randomness is supplied by the
// caller (Bernoulli masks and one 31-bit integer per pixel), so runs can be made reproducible.
#include "common.h"

namespace {

constexpr int MAXC = 16;  // max clusters (the reference configures 7 + background + "none" = 9)
constexpr int DB = 1024;   // pixels per counting-sort block (16 waves): keeps the single-workgroup scan short

__device__ __forceinline__ int pixel_cluster(const float4* __restrict__ rast, const int* __restrict__ fid2cid, int nfid, long long p) {
    const int fid = (int)rast[p].w;
    return fid2cid[min(max(fid, 0), nfid - 1)];
}

// pass 1: block_counts[block][c]
__global__ __launch_bounds__(DB) void disturb_count_kernel(const float4* __restrict__ rast, const int* __restrict__ fid2cid, int nfid,
                                                            int ncl, long long n, int* __restrict__ block_counts) {
    __shared__ int cnt[DB / 64][MAXC];
    const long long p = (long long)blockIdx.x * DB + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = p < n ? pixel_cluster(rast, fid2cid, nfid, p) : -1;
    for (int k = 0; k < ncl; k++) {
        const int m = __popcll(__ballot(c == k));
        if (lane == 0) cnt[wave][k] = m;
    }
    __syncthreads();
    if (threadIdx.x < ncl) {
        int tot = 0;
        for (int w = 0; w < DB / 64; w++) tot += cnt[w][threadIdx.x];
        block_counts[(size_t)blockIdx.x * MAXC + threadIdx.x] = tot;
    }
}

// pass 2 (one workgroup of 1024 = 16 waves): exclusive scan over blocks for all clusters at once; wave-level shuffle
// scans + one LDS exchange (two barriers in total).  totals[c], totals[MAXC + c] = first slot of cluster c.
__global__ __launch_bounds__(1024) void disturb_scan_kernel(int* __restrict__ block_counts, int nblocks, int ncl, int* __restrict__ totals) {
    __shared__ int wtot[16][MAXC];
    const int t = threadIdx.x, lane = t & 63, wave = t >> 6;
    const int per = (nblocks + 1023) / 1024;
    const int b0 = t * per, b1 = min(b0 + per, nblocks);
    int s[MAXC], incl[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; c++) s[c] = 0;
    for (int b = b0; b < b1; b++) {
#pragma unroll
        for (int c = 0; c < MAXC; c++)
            if (c < ncl) s[c] += block_counts[(size_t)b * MAXC + c];
    }
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        int v = s[c];
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(v, o, 64);
            if (lane >= o) v += u;
        }
        incl[c] = v;
        if (lane == 63) wtot[wave][c] = v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < MAXC; c++) {
        if (c >= ncl) continue;
        int run = incl[c] - s[c];
        for (int w = 0; w < wave; w++) run += wtot[w][c];
        for (int b = b0; b < b1; b++) {
            const int v = block_counts[(size_t)b * MAXC + c];
            block_counts[(size_t)b * MAXC + c] = run;
            run += v;
        }
    }
    if (t == 0) {
        int start = 0;
        for (int c = 0; c < ncl; c++) {
            int tot = 0;
            for (int w = 0; w < 16; w++) tot += wtot[w][c];
            totals[c] = tot;
            totals[MAXC + c] = start;
            start += tot;
        }
    }
}

// pass 3: perm[starts[c] + block_offset[c] + rank in block] = pixel id
__global__ __launch_bounds__(DB) void disturb_scatter_kernel(const float4* __restrict__ rast, const int* __restrict__ fid2cid, int nfid,
                                                              int ncl, long long n, const int* __restrict__ block_offsets,
                                                              const int* __restrict__ totals, int* __restrict__ perm) {
    __shared__ int wcnt[DB / 64][MAXC];
    const long long p = (long long)blockIdx.x * DB + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = p < n ? pixel_cluster(rast, fid2cid, nfid, p) : -1;
    int rank = 0;
    for (int k = 0; k < ncl; k++) {
        const unsigned long long m = __ballot(c == k);
        if (c == k) rank = __popcll(m & ((1ull << lane) - 1ull));
        if (lane == 0) wcnt[wave][k] = __popcll(m);
    }
    __syncthreads();
    if (c >= 0) {
        int off = totals[MAXC + c] + block_offsets[(size_t)blockIdx.x * MAXC + c] + rank;
        for (int w = 0; w < wave; w++) off += wcnt[w][c];
        perm[off] = (int)p;
    }
}

// pass 4: out = w ? src[perm[start_c + idx % n_c]] : cur ; keep = 1 - w_eff (gradient mask for the backward)
__global__ __launch_bounds__(256) void disturb_apply_kernel(const float4* __restrict__ rgba, const float4* __restrict__ rgba_bg_or_null,
                                                            const float* __restrict__ bg_image, int B, int H, int W,
                                                            const float4* __restrict__ rast, const int* __restrict__ fid2cid, int nfid,
                                                            const int* __restrict__ w_fg, const int* __restrict__ w_bg,
                                                            const long long* __restrict__ idx, const int* __restrict__ totals,
                                                            const int* __restrict__ perm, float4* __restrict__ out,
                                                            float* __restrict__ keep) {
    const long long n = (long long)B * H * W;
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const int c = pixel_cluster(rast, fid2cid, nfid, p);
    const int w = c == 0 ? w_bg[p] : (c == 1 ? 0 : w_fg[p]);
    const int nc = totals[c];
    float4 v = rgba[p];   // after compositing, background pixels of `rgba` already hold the background colour
    float k = 1.0f;
    if (w != 0 && nc > 0) {
        const int q = perm[totals[MAXC + c] + (int)(idx[p] % (long long)nc)];
        v = rgba[q];
        k = 0.0f;
    }
    out[p] = v;
    keep[p] = k;
}

__global__ __launch_bounds__(256) void disturb_bwd_kernel(const float4* __restrict__ d_out, const float* __restrict__ keep, long long n,
                                                          float4* __restrict__ d_rgba) {
    const long long p = (long long)blockIdx.x * 256 + threadIdx.x;
    if (p >= n) return;
    const float k = keep[p];
    const float4 g = d_out[p];
    d_rgba[p] = make_float4(g.x * k, g.y * k, g.z * k, g.w * k);
}

}  // namespace

extern "C" size_t vhap_disturb_workspace_ints(int B, int H, int W) {
    if (B <= 0 || H <= 0 || W <= 0) return 0;
    const long long n = (long long)B * H * W;
    const long long nblocks = (n + DB - 1) / DB;
    return (size_t)(2 * MAXC + nblocks * MAXC + n);
}

extern "C" int vhap_disturb_fwd(const float* rgba, const float* rast, const int32_t* fid2cid, int nfid, int ncl, const int32_t* w_fg,
                                const int32_t* w_bg, const int64_t* idx, int B, int H, int W, int32_t* workspace, float* out,
                                float* keep, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!rgba || !rast || !fid2cid || !w_fg || !w_bg || !idx || !workspace || !out || !keep) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || ncl <= 0 || ncl > MAXC || nfid <= 0 || (long long)B * H * W >= (1ll << 31)) return VHAP_E_BADDIM;
    const long long n = (long long)B * H * W;
    const int nblocks = vhap_cdiv(n, DB);
    int* totals = workspace;
    int* block_counts = workspace + 2 * MAXC;
    int* perm = block_counts + (size_t)nblocks * MAXC;
    hipStream_t st = vhap_stream(stream);
    const float4* r4 = reinterpret_cast<const float4*>(rast);
    disturb_count_kernel<<<nblocks, DB, 0, st>>>(r4, fid2cid, nfid, ncl, n, block_counts);
    VHAP_LAUNCH_CHECK();
    disturb_scan_kernel<<<1, 1024, 0, st>>>(block_counts, nblocks, ncl, totals);
    VHAP_LAUNCH_CHECK();
    disturb_scatter_kernel<<<nblocks, DB, 0, st>>>(r4, fid2cid, nfid, ncl, n, block_counts, totals, perm);
    VHAP_LAUNCH_CHECK();
    disturb_apply_kernel<<<vhap_cdiv(n, 256), 256, 0, st>>>(reinterpret_cast<const float4*>(rgba), nullptr, nullptr, B, H, W, r4, fid2cid, nfid, w_fg,
                                                  w_bg, reinterpret_cast<const long long*>(idx), totals, perm,
                                                  reinterpret_cast<float4*>(out), keep);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_disturb_bwd(const float* d_out, const float* keep, int B, int H, int W, float* d_rgba, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!d_out || !keep || !d_rgba) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    const long long n = (long long)B * H * W;
    disturb_bwd_kernel<<<vhap_cdiv(n, 256), 256, 0, vhap_stream(stream)>>>(reinterpret_cast<const float4*>(d_out), keep, n,
                                                                          reinterpret_cast<float4*>(d_rgba));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}
