// Cluster-wise colour disturbance for gfx950 (vhap/util/render_nvdiffrast.py:424-460).
//
// Reference semantics: every pixel belongs to a colour cluster cid = fid2cid[triangle id + 1]
// (0 = background, 1 = face in no cluster).  For each cluster i != 1, with probability `rate` a pixel is
// replaced by a pixel drawn uniformly from the pool of ALL pixels of that cluster in the whole batch
// (background pixels draw from the background image); the drawn colour is detached.  The reference does
// this with a 9-iteration Python loop of boolean-mask indexing (one host sync each) and randint.
//
// Here: a deterministic counting sort of the pixel COLOURS by cluster (row-major order inside a cluster, i.e.
// exactly the order of the reference's boolean-mask gather) in two passes -- per-block cluster histograms
// (wave ballots over the cluster values present in a wave) and a scatter in which every workgroup derives its own prefix from the
// histograms and an LDS prefix table over its (iteration, wave) counters and copies its pixels' colours into the cluster's POOL
// (16 B per pixel, contiguous runs per workgroup and cluster) -- followed by one gather pass: a disturbed pixel reads ONE random
// 16-byte colour from its cluster's dense pool.  (Round 2 sorted pixel ids and gathered twice -- id, then colour at that id: two
// dependent random HBM sectors per disturbed pixel, 558 MB read per 16 x 512^2 step, the largest reader of the step.)  Because the
// pools are copies, the gather pass may run IN PLACE (vhap_disturb_inplace): undisturbed pixels are neither read nor written.
// No host sync, no atomics, graph-capturable.  Randomness: drawn in-kernel from a counter-based generator, or supplied by the
// caller (Bernoulli masks and one 31-bit integer per pixel) so that runs can be replayed by the oracle.
#include "common.h"

namespace {

constexpr int MAXC = 16;  // max clusters (the reference configures 7 + background + "none" = 9)
constexpr int DB = 256;   // threads per counting-sort workgroup
constexpr int PPT = 8;    // pixels per thread: a workgroup owns DB * PPT consecutive pixels (2048: 2048 workgroups at 16x512^2, all resident)
constexpr int DPIX = DB * PPT;
constexpr int PB = 1024;  // threads of the single prefix workgroup
#ifndef VHAP_DISTURB_APT
#define VHAP_DISTURB_APT 1
#endif
constexpr int APT = VHAP_DISTURB_APT;   // pixels per thread of the apply pass

// cluster of pixel p: from the compact one-byte cluster image when the shading kernel wrote one (4 MB instead of a 67 MB pass over
// the rasteriser output at 16x512^2), else through the triangle id of rast and the fid -> cluster table
struct ClusterSrc {
    const float4* rast;
    const int* fid2cid;
    int nfid;
    const unsigned char* cid;
};
__device__ __forceinline__ int pixel_cluster(const ClusterSrc& s, long long p) {
    if (s.cid) return (int)s.cid[p];
    const int fid = (int)s.rast[p].w;
    return s.fid2cid[min(max(fid, 0), s.nfid - 1)];
}
// the clusters of a thread's PPT pixels (p0 + it * stride; -1 past the end), ALL loads issued before the first is consumed: called per
// iteration, each value is load -> wait -> ballot, PPT round trips in series in kernels that are a few microseconds of work otherwise
template <int N>
__device__ __forceinline__ void pixel_clusters(const ClusterSrc& s, long long p0, int stride, long long n, int (&c)[N]) {
    if (s.cid) {
        unsigned char v[N];
#pragma unroll
        for (int it = 0; it < N; it++) { const long long p = p0 + (long long)it * stride; v[it] = s.cid[p < n ? p : n - 1]; }
#pragma unroll
        for (int it = 0; it < N; it++) c[it] = p0 + (long long)it * stride < n ? (int)v[it] : -1;
    } else {
        float w[N];
#pragma unroll
        for (int it = 0; it < N; it++) { const long long p = p0 + (long long)it * stride; w[it] = s.rast[p < n ? p : n - 1].w; }
#pragma unroll
        for (int it = 0; it < N; it++) c[it] = s.fid2cid[min(max((int)w[it], 0), s.nfid - 1)];
#pragma unroll
        for (int it = 0; it < N; it++) if (!(p0 + (long long)it * stride < n)) c[it] = -1;
    }
}
// start of every cluster's pool = the totals of the clusters before it: all MAXC totals in one batch (a loop to the lane's own cluster is
// up to 15 dependent round trips)
__device__ __forceinline__ int cluster_start(const int* __restrict__ totals, int c) {
    int tv[16];
#pragma unroll
    for (int k = 0; k < 16; k++) tv[k] = totals[k];
    int start = 0;
#pragma unroll
    for (int k = 0; k < 16; k++) start += k < c ? tv[k] : 0;
    return start;
}

// counter-based random bits (two rounds of the murmur3 finaliser over a Weyl-mixed key): statistically ample for a
// colour-dither; one stream per (call, pixel, draw)
__device__ __forceinline__ unsigned mix32(unsigned h) {
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
__device__ __forceinline__ unsigned rnd32(unsigned seed, unsigned p, unsigned draw) {
    return mix32(mix32(p * 0x9e3779b9u + seed) ^ (draw * 0x7f4a7c15u + 0x2545f491u));
}

// pass 1: block_counts[block][c]; also advances the random-stream counter of the call (nobody reads it during this pass)
__global__ __launch_bounds__(DB) void disturb_count_kernel(const ClusterSrc src,
                                                            int ncl, long long n, int* __restrict__ block_counts,
                                                            unsigned* __restrict__ rng_state) {
    __shared__ int cnt[DB / 64][MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (rng_state && blockIdx.x == 0 && threadIdx.x == 0) rng_state[0] += 1u;
    if (lane < MAXC) cnt[wave][lane] = 0;
    // a wave holds 64 consecutive pixels (one or two clusters, rarely more): loop over the cluster values present, not over all clusters
    int cs[PPT];
    pixel_clusters<PPT>(src, (long long)blockIdx.x * DPIX + threadIdx.x, DB, n, cs);
#pragma unroll
    for (int it = 0; it < PPT; it++) {
        int c = cs[it];
        if (c >= ncl) c = -1;                       // (ids outside the configured clusters are left alone)
        unsigned long long todo = __ballot(c >= 0);
        while (todo) {
            const int k = __builtin_amdgcn_readlane(c, __builtin_ctzll(todo));
            const unsigned long long m = __ballot(c == k);
            if (lane == 0) cnt[wave][k] += __popcll(m);
            todo &= ~m;
        }
    }
    __syncthreads();
    if (threadIdx.x < MAXC) {
        int tot = 0;
        for (int w = 0; w < DB / 64; w++) tot += cnt[w][threadIdx.x];
        block_counts[(size_t)blockIdx.x * MAXC + threadIdx.x] = threadIdx.x < ncl ? tot : 0;
    }
}

// pass 2 (one workgroup PER CLUSTER): block_counts[block][c] -> exclusive prefix over the blocks, in place; totals[c].  1024 rows per
// round (one per thread: wave scan + the 16 wave totals through LDS), a carry between rounds.  The consumers derive the start of a
// cluster's pool (the sum of the totals of the clusters before it) themselves, so the clusters' workgroups are independent.
// (Round 2 let every scatter workgroup of 1024 threads derive its own prefix from all rows: no extra launch, but with 106 VGPRs only
// one such workgroup fits a CU and its phases -- prefix, ranks, copy -- run one after the other with nothing to overlap; fine for
// 4-byte ids, 128 us once the pass copies 16-byte colours.  A single-workgroup scan over all 16 clusters took 19 us.)
__global__ __launch_bounds__(PB) void disturb_prefix_kernel(int* __restrict__ block_counts, int nblocks, int* __restrict__ totals) {
    __shared__ int wtot[PB / 64];
    __shared__ int carry_s;
    const int c = blockIdx.x, lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x == 0) carry_s = 0;
    __syncthreads();
    for (int j0 = 0; j0 < nblocks; j0 += PB) {
        const int j = j0 + threadIdx.x;
        const int v = j < nblocks ? block_counts[(size_t)j * MAXC + c] : 0;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const int u = __shfl_up(incl, o, 64);
            if (lane >= o) incl += u;
        }
        if (lane == 63) wtot[wave] = incl;
        __syncthreads();
        int pre = carry_s;
        for (int w = 0; w < wave; w++) pre += wtot[w];
        if (j < nblocks) block_counts[(size_t)j * MAXC + c] = pre + incl - v;
        __syncthreads();
        if (threadIdx.x == PB - 1) carry_s = pre + incl;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[c] = carry_s;
}

// pass 3: pool[start_c + prefix_c(block) + rank in block] = colour of the pixel.  Ranks inside the block in pixel order: iteration-major,
// then wave, then lane.  A wave holds 64 consecutive pixels -- one or two clusters, rarely more -- so it loops over the cluster values
// PRESENT (readlane + ballot) instead of over all clusters.  256-thread workgroups, all resident at once: the latency-bound rank
// phase of one overlaps the copy phase of another.
__global__ __launch_bounds__(DB) void disturb_scatter_kernel(const ClusterSrc src, const float4* __restrict__ rgba,
                                                              int ncl, long long n, const int* __restrict__ block_prefix,
                                                              const int* __restrict__ totals, float4* __restrict__ pool,
                                                              unsigned* __restrict__ cov_list, int* __restrict__ n_bg_out) {
    constexpr int NW = DB / 64;                     // waves
    constexpr int NE = PPT * NW;                    // (iteration, wave) counters per cluster: 32
    static_assert(NE == 32 && MAXC == 16, "the in-block scan below assumes 32 counters per cluster, two clusters per wave pass");
    __shared__ int base[MAXC];                      // start_c + prefix of this block
    __shared__ int wcnt[PPT][NW][MAXC];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (threadIdx.x < MAXC) {
        const int pre = block_prefix[(size_t)blockIdx.x * MAXC + threadIdx.x];
        base[threadIdx.x] = cluster_start(totals, (int)threadIdx.x) + pre;
    }
    for (int i = threadIdx.x; i < PPT * NW * MAXC; i += DB) (&wcnt[0][0][0])[i] = 0;
    __syncthreads();
    int key[PPT];                                   // cluster << 8 | rank in the wave's 64 pixels (-1: not sorted)
    int cs[PPT];
    int bgb[PPT];                                   // background pixels of the wave in front of this lane (for the list of covered pixels)
    pixel_clusters<PPT>(src, (long long)blockIdx.x * DPIX + threadIdx.x, DB, n, cs);
#pragma unroll
    for (int it = 0; it < PPT; it++) {
        int c = cs[it];
        if (c >= ncl) c = -1;                       // (ids outside the configured clusters are left alone)
        int rank = 0;
        bgb[it] = __popcll(__ballot(c == 0) & ((1ull << lane) - 1ull));
        unsigned long long todo = __ballot(c >= 0);
        while (todo) {
            const int k = __builtin_amdgcn_readlane(c, __builtin_ctzll(todo));
            const unsigned long long m = __ballot(c == k);
            if (c == k) rank = __popcll(m & ((1ull << lane) - 1ull));
            if (lane == 0) wcnt[it][wave][k] = __popcll(m);
            todo &= ~m;
        }
        key[it] = c < 0 ? -1 : ((c << 8) | rank);
    }
    __syncthreads();
    // exclusive prefix of wcnt over (iteration, wave) per cluster, in place: a wave pass scans two clusters (32 counters each), so that
    // a pixel needs ONE LDS read for its offset
#pragma unroll
    for (int r = 0; r < MAXC / (2 * NW); r++) {
        const int k = (wave * (MAXC / (2 * NW)) + r) * 2 + (lane >> 5), e = lane & 31;
        int* cell = &wcnt[0][0][0] + e * MAXC + k;
        const int v = *cell;
        int incl = v;
#pragma unroll
        for (int o = 1; o < 32; o <<= 1) {
            const int u = __shfl_up(incl, o, 32);
            if (e >= o) incl += u;
        }
        *cell = incl - v;
    }
    __syncthreads();
    // the copy: all of the thread's colours requested, then stored (clamped addresses: pixels past the end / outside the clusters are not stored)
    float4 col[PPT];
#pragma unroll
    for (int it = 0; it < PPT; it++) {
        const long long p = (long long)blockIdx.x * DPIX + it * DB + threadIdx.x;
        col[it] = rgba[p < n ? p : n - 1];
    }
#pragma unroll
    for (int it = 0; it < PPT; it++) asm volatile("" ::"v"(col[it].x), "v"(col[it].y), "v"(col[it].z), "v"(col[it].w));   // (or the compiler sinks each load into its `if` below)
#pragma unroll
    for (int it = 0; it < PPT; it++) {
        if (key[it] < 0) continue;
        const int c = key[it] >> 8, rank = key[it] & 255;
        pool[base[c] + wcnt[it][wave][c] + rank] = col[it];
    }
    // The list of covered pixels (cluster != 0) in pixel order, for passes that have nothing to do on the background (vhap_deferred_shade_bwd_list):
    // entry number (pixels in front of p) - (background pixels in front of p) -- the blocks before this one (the prefix of cluster 0), the
    // (iteration, wave) units of this block before p's (the in-block scan of cluster 0), the lanes of p's wave before p.
    if (cov_list) {
#pragma unroll
        for (int it = 0; it < PPT; it++) {
            const long long p = (long long)blockIdx.x * DPIX + it * DB + threadIdx.x;
            if (p < n && cs[it] != 0) cov_list[p - (base[0] + wcnt[it][wave][0] + bgb[it])] = (unsigned)p;
        }
        if (blockIdx.x == 0 && threadIdx.x == 0) *n_bg_out = totals[0];
    }
}

// pass 4: out = w ? pool[start_c + idx % n_c] : cur ; keep = 1 - w_eff (gradient mask for the backward).
// INPLACE: `out` IS the image the pools were copied from: only disturbed pixels are touched (no read of the pixel's own colour).
template <bool INPLACE>
__global__ __launch_bounds__(256) void disturb_apply_kernel(const float4* __restrict__ rgba, int B, int H, int W,
                                                            const ClusterSrc src,
                                                            const int* __restrict__ w_fg, const int* __restrict__ w_bg,
                                                            const long long* __restrict__ idx, const unsigned* __restrict__ rng_state,
                                                            float rate_fg, float rate_bg, const int* __restrict__ totals,
                                                            const float4* __restrict__ pool, float4* __restrict__ out,
                                                            float* __restrict__ keep) {
    __shared__ int s_tot[MAXC], s_start[MAXC];
    if (threadIdx.x < MAXC) {
        const int own = totals[threadIdx.x];
        s_start[threadIdx.x] = cluster_start(totals, (int)threadIdx.x);
        s_tot[threadIdx.x] = own;
    }
    __syncthreads();
    // APT pixels per thread (p0 + it * 256), three batches of loads: clusters, (injected draws,) the drawn colours.  The kernel is a random
    // 16-byte gather over a 67 MB pool and sits at s_waitcnt for 84 % of its wave-cycles (profiles/r04_call22_step_sq_pmc.json) -- but
    // four gathers in flight per lane instead of one change nothing (59.3 vs 58.2 us, profiles/r04_call29_apply_photo_ab.txt): it is bound
    // by the 64-byte sectors the gather drags in (253 MB read for 67 MB used), not by the round trip.  APT = 1 is shipped.
    const long long n = (long long)B * H * W;
    const long long p0 = (long long)blockIdx.x * (256 * APT) + threadIdx.x;
    int cs[APT];
    pixel_clusters<APT>(src, p0, 256, n, cs);
    int wd[APT];
    unsigned long long pick[APT];
    if (rng_state) {     // in-kernel random numbers: Bernoulli(rate) and a 32-bit index draw per pixel
        const unsigned seed = rng_state[0];
#pragma unroll
        for (int it = 0; it < APT; it++) {
            const unsigned p = (unsigned)(p0 + (long long)it * 256);
            const int c = cs[it] >= MAXC ? 1 : cs[it];
            const float u = (float)(rnd32(seed, p, 0u) >> 8) * (1.0f / 16777216.0f);
            wd[it] = c == 0 ? (u < rate_bg) : (c == 1 ? 0 : (u < rate_fg));
            pick[it] = rnd32(seed, p, 1u);
        }
    } else {
        int r_fg[APT], r_bg[APT];
        long long r_ix[APT];
#pragma unroll
        for (int it = 0; it < APT; it++) {
            const long long p = p0 + (long long)it * 256, pc = p < n ? p : n - 1;
            r_fg[it] = w_fg[pc]; r_bg[it] = w_bg[pc]; r_ix[it] = idx[pc];
        }
#pragma unroll
        for (int it = 0; it < APT; it++) {
            const int c = cs[it] >= MAXC ? 1 : cs[it];
            wd[it] = c == 0 ? r_bg[it] : (c == 1 ? 0 : r_fg[it]);
            pick[it] = (unsigned long long)r_ix[it];
        }
    }
    bool dis[APT];
    float4 col[APT];
#pragma unroll
    for (int it = 0; it < APT; it++) {
        const long long p = p0 + (long long)it * 256;
        int c = cs[it];                                 // (-1 past the end)
        if (c >= MAXC) c = 1;                           // (treated like 'face in no cluster': never disturbed)
        const int nc = c >= 0 ? s_tot[c] : 0;
        dis[it] = p < n && wd[it] != 0 && nc > 0;
        // injected indices: idx % n like the reference's randint-then-index (a 64-bit division per disturbed pixel -- parity path only);
        // in-kernel draws: floor(r * n / 2^32), the same distribution without a division
        const int j = !dis[it] ? 0 : (rng_state ? (int)((pick[it] * (unsigned long long)nc) >> 32) : (int)(pick[it] % (unsigned long long)nc));
        const float4* q = dis[it] ? pool + (s_start[c] + j) : (INPLACE ? pool : rgba + (p < n ? p : n - 1));
        col[it] = *q;                                   // (in place, an undisturbed pixel re-reads pool[0]: unused)
    }
#pragma unroll
    for (int it = 0; it < APT; it++) asm volatile("" ::"v"(col[it].x), "v"(col[it].y), "v"(col[it].z), "v"(col[it].w));
#pragma unroll
    for (int it = 0; it < APT; it++) {
        const long long p = p0 + (long long)it * 256;
        if (p >= n) continue;
        if (dis[it] || !INPLACE) out[p] = col[it];      // (not in place: after compositing, background pixels of `rgba` already hold the background colour)
        keep[p] = dis[it] ? 0.0f : 1.0f;
    }
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
    const long long nblocks = (n + DPIX - 1) / DPIX;
    return (size_t)(2 * MAXC + nblocks * MAXC + 4 * n);          // totals + starts, per-block histograms, the colour pools (16 B per pixel)
}

static int disturb_run(const float* rgba, const float* rast, const uint8_t* cid, const int32_t* fid2cid, int nfid, int ncl, const int32_t* w_fg,
                       const int32_t* w_bg, const int64_t* idx, uint32_t* rng_state, float rate_fg, float rate_bg, int B, int H, int W,
                       int32_t* workspace, float* out, float* keep, vhap_stream_t stream, uint32_t* cov_list = nullptr, int32_t* n_bg_out = nullptr) {
    if (!rgba || !workspace || !out || !keep) return VHAP_E_NULLPTR;
    if ((cov_list == nullptr) != (n_bg_out == nullptr)) return VHAP_E_NULLPTR;
    if (!cid && (!rast || !fid2cid)) return VHAP_E_NULLPTR;
    if (!rng_state && (!w_fg || !w_bg || !idx)) return VHAP_E_NULLPTR;
    if (B <= 0 || H <= 0 || W <= 0 || ncl <= 0 || ncl > MAXC || (!cid && nfid <= 0) || (long long)B * H * W >= (1ll << 31)) return VHAP_E_BADDIM;
    const long long n = (long long)B * H * W;
    const int nblocks = vhap_cdiv(n, DPIX);
    int* totals = workspace;
    int* block_counts = workspace + 2 * MAXC;
    float4* pool = reinterpret_cast<float4*>(block_counts + (size_t)nblocks * MAXC);   // (32 + 16 nblocks ints: 64-byte multiple)
    if ((reinterpret_cast<uintptr_t>(workspace) & 15u) || (reinterpret_cast<uintptr_t>(rgba) & 15u) || (reinterpret_cast<uintptr_t>(out) & 15u))
        return VHAP_E_BADDIM;
    hipStream_t st = vhap_stream(stream);
    const ClusterSrc src{reinterpret_cast<const float4*>(rast), fid2cid, nfid, cid};
    const float4* in = reinterpret_cast<const float4*>(rgba);
    disturb_count_kernel<<<nblocks, DB, 0, st>>>(src, ncl, n, block_counts, rng_state);
    VHAP_LAUNCH_CHECK();
    disturb_prefix_kernel<<<MAXC, PB, 0, st>>>(block_counts, nblocks, totals);
    VHAP_LAUNCH_CHECK();
    disturb_scatter_kernel<<<nblocks, DB, 0, st>>>(src, in, ncl, n, block_counts, totals, pool, cov_list, n_bg_out);
    VHAP_LAUNCH_CHECK();
    if (out == rgba)
        disturb_apply_kernel<true><<<vhap_cdiv(n, 256 * APT), 256, 0, st>>>(in, B, H, W, src, w_fg, w_bg, reinterpret_cast<const long long*>(idx),
                                                                     rng_state, rate_fg, rate_bg, totals, pool, reinterpret_cast<float4*>(out), keep);
    else
        disturb_apply_kernel<false><<<vhap_cdiv(n, 256 * APT), 256, 0, st>>>(in, B, H, W, src, w_fg, w_bg, reinterpret_cast<const long long*>(idx),
                                                                      rng_state, rate_fg, rate_bg, totals, pool, reinterpret_cast<float4*>(out), keep);
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

extern "C" int vhap_disturb_fwd(const float* rgba, const float* rast, const int32_t* fid2cid, int nfid, int ncl, const int32_t* w_fg,
                                const int32_t* w_bg, const int64_t* idx, int B, int H, int W, int32_t* workspace, float* out,
                                float* keep, vhap_stream_t stream) {
    VHAP_ENTER();
    return disturb_run(rgba, rast, nullptr, fid2cid, nfid, ncl, w_fg, w_bg, idx, nullptr, 0.f, 0.f, B, H, W, workspace, out, keep, stream);
}

extern "C" int vhap_disturb_fwd_rng(const float* rgba, const float* rast, const int32_t* fid2cid, int nfid, int ncl, float rate_fg,
                                    float rate_bg, uint32_t* rng_state, int B, int H, int W, int32_t* workspace, float* out,
                                    float* keep, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!rng_state) return VHAP_E_NULLPTR;
    return disturb_run(rgba, rast, nullptr, fid2cid, nfid, ncl, nullptr, nullptr, nullptr, rng_state, rate_fg, rate_bg, B, H, W, workspace, out, keep,
                       stream);
}


extern "C" int vhap_disturb_inplace(float* rgba, const uint8_t* cid, int ncl, const int32_t* w_fg, const int32_t* w_bg, const int64_t* idx,
                                    float rate_fg, float rate_bg, uint32_t* rng_state, int B, int H, int W, int32_t* workspace, float* keep,
                                    vhap_stream_t stream) {
    VHAP_ENTER();
    if (!cid) return VHAP_E_NULLPTR;
    return disturb_run(rgba, nullptr, cid, nullptr, 0, ncl, w_fg, w_bg, idx, rng_state, rate_fg, rate_bg, B, H, W, workspace, rgba, keep, stream);
}

extern "C" int vhap_disturb_inplace_list(float* rgba, const uint8_t* cid, int ncl, const int32_t* w_fg, const int32_t* w_bg, const int64_t* idx,
                                         float rate_fg, float rate_bg, uint32_t* rng_state, int B, int H, int W, int32_t* workspace, float* keep,
                                         uint32_t* covered_list, int32_t* n_background, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!cid || !covered_list || !n_background) return VHAP_E_NULLPTR;
    return disturb_run(rgba, nullptr, cid, nullptr, 0, ncl, w_fg, w_bg, idx, rng_state, rate_fg, rate_bg, B, H, W, workspace, rgba, keep, stream,
                       covered_list, n_background);
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
