// Block-binned triangle rasteriser + fused barycentric/attribute interpolation for gfx950.
//
// Replaces dr.rasterize / dr.interpolate of the reference (vhap/util/render_nvdiffrast.py:254,
// 384, 389).  Conventions are specified in DESIGN.md section 3 and restated independently in
// oracle/raster_oracle.c, against which this file is checked bit-for-bit (triangle ids, z/w, u, v).
//
// Structure (one frame batch = 2 launches):
//   bin_build  : 1 thread / (frame, triangle): near-plane clipping if the triangle crosses z = -w (one or two PIECES, raster_common.h;
//                record slots t and F + t), snap to 1/16 px, cull, pixel bbox -> range of 8x8 pixel BLOCKS, 80-byte setup
//                record; every 1024-triangle workgroup lays its (piece, block) pairs out in its own region of the pair list
//                (LDS histogram -> workgroup scan -> scatter) and publishes fragment descriptors -- no global counters.
//                (bin_count / bin_scan / bin_fill: the three-launch variant with contiguous lists, for meshes > 32768 triangles.)
//   raster     : 1 workgroup = 4 waves = 32x8 pixels, each WAVE owns one 8x8 block and walks its fragments.  Per 64-triangle
//                chunk every lane sets up ONE triangle (exact integer edge functions and the z/w test plane, relative to the
//                block origin) and publishes it in LDS; every lane (= pixel) then reads triangle j with broadcast ds_read_b128,
//                evaluates coverage (3 v_dot2) and the depth plane (2 FMAs) and keeps the smallest (depth, id) key in
//                registers.  No barriers, no atomics; the winner is independent of list order.  The lane then shades its pixel
//                once.  Three modes of the same kernel: 0 = rasterize only (rast, rast_db); 1 = + both interpolations (normal, uv,
//                uv derivatives: the G-buffer of the nvdiffrast-shaped ops); 2 = DEFERRED SHADING: the interpolated attributes stay
//                in registers, the texture is sampled (trilinear mip-mapped), the SH diffuse shading and the background composite are
//                applied and only rast (16 B) + rgba (16 B) + the colour-cluster byte leave the kernel -- 33 B/px instead of the
//                68 + 12 + 17 the three separate passes write (and 88 they re-read).
#include "raster_common.h"
#include "frag_common.h"
#include "shade_common.h"
#include "tex_sample.h"
#include "vnormal_common.h"
#include <algorithm>

#pragma clang fp contract(off)  // bit-exact op order vs the oracle: only explicit fma() fuses

namespace {

constexpr int BLK = 8;              // bin = 8x8 pixels = one wave
constexpr int WG_BLOCKS = 4;        // 4 horizontally adjacent bins per workgroup (32x8 pixels)
constexpr unsigned TRANGE_NONE = 0xffffffffu;
constexpr int LDS_BIN_LIMIT = 16384;  // dense per-workgroup bin histogram (<= 2 x 64 KiB of LDS)
constexpr int BIN_THREADS = 1024;     // fat workgroups: fewer LDS-histogram flushes per frame

// Per-(frame, piece) setup record written once by the binning kernel (80 B, five 16-byte quads): everything
// the raster kernel needs that does not depend on the 8x8 block -- snapped vertices, pixel bbox, the
// per-vertex z/w, 1/area (double) and the vertex / uv-vertex indices.  All the divisions of the setup
// happen once per triangle here instead of once per (triangle, block, wave) in the raster kernel.
struct TriRecord {
    int4 q0;    // sx0, sy0, sx1, sy1                       (1/16-pixel units)
    int4 q1;    // sx2, sy2, px0 | px1 << 16, py0 | py1 << 16 (inclusive pixel bbox)
    float4 q2;  // zw0, zw1, zw2, unused
    int4 q3;    // 1/area as double (lo, hi), i0, i1
    int4 q4;    // i2, j0, j1, j2
};
static_assert(sizeof(TriRecord) == 80, "record layout");

struct BinHeader {
    unsigned total;  // number of (triangle, block) pairs of this batch
    unsigned pad[15];
};

// Block range of one piece (a whole triangle, or a piece of a triangle cut by the near plane): bx0 | bx1<<9 | by0<<18 | span<<27 with
// span = by1-by0, 31 = "up to the last block row" (conservative; the raster kernel re-tests the bbox).  H,W <= 4096 -> < 512.
// Also fills the piece's setup record (geometry of the piece, vertex / uv indices of the triangle it belongs to).
__device__ __forceinline__ unsigned piece_range(const float4 p0, const float4 p1, const float4 p2, int H, int W, int i0, int i1, int i2,
                                                int j0, int j1, int j2, TriRecord& rec) {
    rec.q3.z = i0; rec.q3.w = i1;
    rec.q4 = make_int4(i2, j0, j1, j2);
    int sx[3], sy[3], px0, px1, py0, py1;
    long long area;
    if (!tri_bbox(p0, p1, p2, H, W, sx, sy, area, px0, px1, py0, py1)) {
        rec.q0 = rec.q1 = make_int4(0, 0, 0, 0);
        rec.q2 = make_float4(0.f, 0.f, 0.f, 0.f);
        rec.q3.x = rec.q3.y = 0;
        return TRANGE_NONE;
    }
    const double inv = 1.0 / (double)area;
    rec.q0 = make_int4(sx[0], sy[0], sx[1], sy[1]);
    rec.q1 = make_int4(sx[2], sy[2], px0 | (px1 << 16), py0 | (py1 << 16));
    rec.q2 = make_float4(__fdiv_rn(p0.z, p0.w), __fdiv_rn(p1.z, p1.w), __fdiv_rn(p2.z, p2.w), 0.0f);
    rec.q3.x = __double2loint(inv); rec.q3.y = __double2hiint(inv);
    const int bx0 = px0 / BLK, bx1 = px1 / BLK, by0 = py0 / BLK, by1 = py1 / BLK;
    const int span = by1 - by0;
    return (unsigned)bx0 | ((unsigned)bx1 << 9) | ((unsigned)by0 << 18) | ((unsigned)(span > 30 ? 31 : span) << 27);
}

// Setup of one (frame, triangle): RECORD SLOTS t and F + t of the frame hold its (up to) two pieces -- slot t the triangle itself or
// the first piece of a triangle cut by the near plane (raster_common.h), slot F + t the second piece (one vertex behind the plane: the
// visible part is a quad).  Writes both range words and the records of the drawable pieces; returns the two range words.  The second
// slot is TRANGE_NONE for every triangle that does not cross the plane (the whole head, always): one extra 4-byte store per triangle.
struct TriRanges {
    unsigned r0, r1;
};
__device__ __forceinline__ TriRanges triangle_setup(const float* __restrict__ pos, const int* __restrict__ tri,
                                                    const int* __restrict__ tri_uv, int b, int V, int F, int t, int H, int W,
                                                    unsigned* __restrict__ trange_b, TriRecord* __restrict__ records_b) {
    TriRanges R{TRANGE_NONE, TRANGE_NONE};
    const int i0 = tri[3 * t], i1 = tri[3 * t + 1], i2 = tri[3 * t + 2];
    if ((unsigned)i0 < (unsigned)V && (unsigned)i1 < (unsigned)V && (unsigned)i2 < (unsigned)V) {
        const float4* P = reinterpret_cast<const float4*>(pos) + (size_t)b * V;
        const float4 p0 = P[i0], p1 = P[i1], p2 = P[i2];
        int j0 = 0, j1 = 0, j2 = 0;
        if (tri_uv) { j0 = tri_uv[3 * t]; j1 = tri_uv[3 * t + 1]; j2 = tri_uv[3 * t + 2]; }
        const int behind = vhap_behind_mask(p0, p1, p2);
        TriRecord rec;
        if (behind == 0) {
            R.r0 = piece_range(p0, p1, p2, H, W, i0, i1, i2, j0, j1, j2, rec);
            if (R.r0 != TRANGE_NONE) records_b[t] = rec;
        } else if (behind != 7) {      // crosses the near plane (never the case for a tracked head: cold path)
            float4 a0, a1, a2, b0, b1, b2;
            const int np = vhap_clip_near(p0, p1, p2, behind, a0, a1, a2, b0, b1, b2);
            R.r0 = piece_range(a0, a1, a2, H, W, i0, i1, i2, j0, j1, j2, rec);
            records_b[t] = rec;        // always: the winner's indices are read from slot t whichever piece won
            if (np == 2) {
                R.r1 = piece_range(b0, b1, b2, H, W, i0, i1, i2, j0, j1, j2, rec);
                if (R.r1 != TRANGE_NONE) records_b[F + t] = rec;
            }
        }
    }
    trange_b[t] = R.r0;
    trange_b[F + t] = R.r1;
    return R;
}

struct BlockRange {
    int bx0, bx1, by0, by1;
};
__device__ __forceinline__ BlockRange decode_range(unsigned tr, int nby) {
    BlockRange r;
    r.bx0 = tr & 511; r.bx1 = (tr >> 9) & 511; r.by0 = (tr >> 18) & 511;
    const int span = tr >> 27;
    r.by1 = span == 31 ? nby - 1 : r.by0 + span;
    return r;
}

// The mesh is spatially coherent in triangle order, so the 256 triangles of a workgroup hit a handful
// of bins.  Global atomics on a few hot counters serialise in the L2 (measured: 30 us for 150 k adds);
// instead every workgroup histograms into LDS and flushes ONE global atomic per touched bin.
// grid = (ceil(F/256), B); dynamic LDS = nbin * 4 bytes when nbin <= LDS_BIN_LIMIT, else direct atomics.
__global__ __launch_bounds__(BIN_THREADS) void bin_count_kernel(const float* __restrict__ pos, const int* __restrict__ tri,
                                                                const int* __restrict__ tri_uv, int V, int F, int H, int W,
                                                                int nbx, int nby, unsigned* __restrict__ counts,
                                                                unsigned* __restrict__ trange, TriRecord* __restrict__ records,
                                                                BinHeader* __restrict__ hdr_to_zero) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_cnt[];
    const int nbin = nbx * nby;
    const bool use_lds = nbin <= LDS_BIN_LIMIT;
    const int b = blockIdx.y, t = blockIdx.x * BIN_THREADS + threadIdx.x;
    if (hdr_to_zero && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x == 0) hdr_to_zero->total = 0u;   // bin_scan (next launch) accumulates
    if (use_lds) {
        for (int i = threadIdx.x; i < nbin; i += BIN_THREADS) lds_cnt[i] = 0u;
        __syncthreads();
    }
    unsigned* c = counts + (size_t)b * nbin;
    TriRanges R{TRANGE_NONE, TRANGE_NONE};
    if (t < F) R = triangle_setup(pos, tri, tri_uv, b, V, F, t, H, W, trange + (size_t)b * 2 * F, records + (size_t)b * 2 * F);
#pragma unroll
    for (int piece = 0; piece < 2; piece++) {
        const unsigned tr = piece ? R.r1 : R.r0;
        if (tr == TRANGE_NONE) continue;
        const BlockRange r = decode_range(tr, nby);
        for (int y = r.by0; y <= r.by1; y++)
            for (int x = r.bx0; x <= r.bx1; x++) {
                if (use_lds) atomicAdd(&lds_cnt[y * nbx + x], 1u);
                else atomicAdd(&c[y * nbx + x], 1u);
            }
    }
    if (use_lds) {
        __syncthreads();
        for (int i = threadIdx.x; i < nbin; i += BIN_THREADS) {
            const unsigned n = lds_cnt[i];
            if (n) atomicAdd(&c[i], n);
        }
    }
}

__global__ __launch_bounds__(256) void bin_scan_kernel(const unsigned* __restrict__ counts, unsigned* __restrict__ offsets,
                                                       int n, BinHeader* __restrict__ hdr) {
    __shared__ unsigned wsum[4];
    __shared__ unsigned base;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned c = i < n ? counts[i] : 0u;
    unsigned v = c;  // inclusive wave scan
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(v, o, 64);
        if (lane >= o) v += u;
    }
    if (lane == 63) wsum[wave] = v;
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned s = wsum[0] + wsum[1] + wsum[2] + wsum[3];
        base = s ? atomicAdd(&hdr->total, s) : 0u;
    }
    __syncthreads();
    unsigned pre = base;
    for (int k = 0; k < wave; k++) pre += wsum[k];
    if (i < n) offsets[i] = pre + v - c;
}

__global__ __launch_bounds__(BIN_THREADS) void bin_fill_kernel(const unsigned* __restrict__ trange, int F, int nbx, int nby,
                                                               const unsigned* __restrict__ offsets,
                                                               unsigned* __restrict__ cursors, unsigned* __restrict__ list,
                                                               const BinHeader* __restrict__ hdr, unsigned capacity) {
    extern __shared__ __attribute__((aligned(16))) unsigned lds_fill[];
    if (hdr->total > capacity) return;  // the raster kernel brute-forces instead (uniform exit)
    const int nbin = nbx * nby;
    const bool use_lds = nbin <= LDS_BIN_LIMIT;
    unsigned* lcnt = lds_fill;
    unsigned* lbase = lds_fill + nbin;
    const int b = blockIdx.y, t = blockIdx.x * BIN_THREADS + threadIdx.x;
    const size_t tb = (size_t)b * nbin;
    const unsigned tr = t < F ? trange[(size_t)b * 2 * F + t] : TRANGE_NONE;
    const unsigned tr1 = t < F ? trange[(size_t)b * 2 * F + F + t] : TRANGE_NONE;     // second piece of a near-clipped triangle: slot F + t
    BlockRange r{0, -1, 0, -1}, r1{0, -1, 0, -1};
    if (tr != TRANGE_NONE) r = decode_range(tr, nby);
    if (tr1 != TRANGE_NONE) r1 = decode_range(tr1, nby);
    if (!use_lds) {
        for (int y = r.by0; y <= r.by1; y++)
            for (int x = r.bx0; x <= r.bx1; x++) {
                const size_t bi = tb + y * nbx + x;
                list[offsets[bi] + atomicAdd(&cursors[bi], 1u)] = (unsigned)t;
            }
        for (int y = r1.by0; y <= r1.by1; y++)
            for (int x = r1.bx0; x <= r1.bx1; x++) {
                const size_t bi = tb + y * nbx + x;
                list[offsets[bi] + atomicAdd(&cursors[bi], 1u)] = (unsigned)(F + t);
            }
        return;
    }
    for (int i = threadIdx.x; i < nbin; i += BIN_THREADS) lcnt[i] = 0u;
    __syncthreads();
    for (int y = r.by0; y <= r.by1; y++)
        for (int x = r.bx0; x <= r.bx1; x++) atomicAdd(&lcnt[y * nbx + x], 1u);
    for (int y = r1.by0; y <= r1.by1; y++)
        for (int x = r1.bx0; x <= r1.bx1; x++) atomicAdd(&lcnt[y * nbx + x], 1u);
    __syncthreads();
    for (int i = threadIdx.x; i < nbin; i += BIN_THREADS) {
        const unsigned n = lcnt[i];
        if (n) {
            lbase[i] = offsets[tb + i] + atomicAdd(&cursors[tb + i], n);
            lcnt[i] = 0u;
        }
    }
    __syncthreads();
    for (int y = r.by0; y <= r.by1; y++)
        for (int x = r.bx0; x <= r.bx1; x++) {
            const int bi = y * nbx + x;
            list[lbase[bi] + atomicAdd(&lcnt[bi], 1u)] = (unsigned)t;
        }
    for (int y = r1.by0; y <= r1.by1; y++)
        for (int x = r1.bx0; x <= r1.bx1; x++) {
            const int bi = y * nbx + x;
            list[lbase[bi] + atomicAdd(&lcnt[bi], 1u)] = (unsigned)(F + t);
        }
}

// ------------------------------------------------------------------------------------------------------------
// One-launch binning ("fragmented lists").  count + scan + fill above are three launches because a bin's list must be
// contiguous and its length is only known after all triangles were counted.  Here every workgroup (1024 consecutive triangles
// of one frame) owns a fixed REGION of the pair list and lays out its own contribution to every bin inside it: LDS histogram ->
// workgroup-wide exclusive scan in LDS -> second pass writes the triangle ids.  A bin's list is then the concatenation of at
// most nfrag = ceil(F/1024) fragments, described by frag[frame][workgroup][bin] = (offset, count); the raster kernel walks the
// fragments.  No global counters, nothing to clear between calls, no cross-workgroup dependency.  A workgroup whose pairs do not
// fit its region publishes OVERFLOW descriptors instead: the raster blocks then scan that workgroup's triangle range directly
// (bbox test against the setup records) -- a local, bounded fallback.
// grid = (nfrag, B); dynamic LDS = nbin * 4 bytes.
// ------------------------------------------------------------------------------------------------------------
// In-kernel wall-clock stamps (VHAP_RASTER_PROFILE): first start / last end of a kernel's waves, for timing the pass INSIDE a captured
// graph, where neither HIP events nor a profiler can look.  wall_clock64() = the constant 100 MHz counter.  Hashed over PROF_SLOTS (min, max)
// pairs so that the atomics do not queue up on one address.
constexpr int PROF_SLOTS = 256;
__device__ __forceinline__ void prof_begin(unsigned long long* prof) {
    if (prof && threadIdx.x == 0) atomicMin(&prof[2 * (blockIdx.x % PROF_SLOTS)], (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void prof_end(unsigned long long* prof) {
    if (prof && (threadIdx.x & 63) == 0) atomicMax(&prof[2 * (blockIdx.x % PROF_SLOTS) + 1], (unsigned long long)wall_clock64());
}
__global__ __launch_bounds__(256) void prof_init_kernel(unsigned long long* prof, int n) {
    const int i = blockIdx.x * 256 + threadIdx.x;
    if (i < n) prof[i] = (i & 1) ? 0ull : ~0ull;
}

constexpr unsigned FRAG_OVERFLOW = 0x80000000u;
constexpr int MAX_FRAG = 32;   // fragments per bin the raster kernel stages per wave (meshes up to 32768 triangles; beyond: count/scan/fill)

struct VnJob {                 // vertex normals computed by extra workgroups of the binning launch (verts == nullptr: none)
    const float* verts;        // [B,V,3] world-space vertices
    const int *vc_ptr, *vc_idx;
    float *vn, *inv_len;       // [B,V,3], [B,V] (may be null)
};

// EARLY STORES of the blocks no triangle can touch (round 5; OPT-IN, off in the shipped step: it loses).  Two thirds of a head frame's 8x8
// blocks are background, and more than half lie outside the screen-space bounding box of the frame's vertices -- a property of the clip
// positions alone.  Their output (zeros; in mode 2 the background composite) needs nothing from the binning, yet it is stored by raster
// waves that first wait for it.  With VHAP_RASTER_PREFILL a 3 us launch reduces that box per frame, extra workgroups of the BINNING launch --
// latency-bound, the memory system idle -- store every block outside it, and the raster kernel's waves of those blocks leave at once.
// A frame with a vertex behind the near plane or at w <= 0 gets the whole frame as its box (clipped pieces project anywhere).
// MEASURED (16 x 512^2, 52 % of the blocks outside the boxes = 148 MB of the pass's 285 MB of stores):
//   v1 (every prefill workgroup reduces the box itself, block-shaped stores, 512 workgroups): binning launch 18 -> 41 us, raster<1> 84 -> 68 us
//   v2 (box from its own launch, row-run stores, one round of 256 workgroups):               binning launch 18 -> 47 us, raster<1> 79 -> 66 us
// i.e. the pass gets 7 - 16 us LONGER (frac 0.36 -> 0.33), and raster<2> 107 -> 98 us for +31 us of binning launch: the step does not
// move (0.888 vs 0.884 ms).  Why: sixteen-wave workgroups beside the binning ones store at 3.4 TB/s, not at the 5.4 TB/s the raster
// kernel's 8192 resident waves reach, and the raster kernel is bound by the dependent round trips of its COVERED waves (descriptor ->
// list -> record -> winner's vertices, each inflated by the store traffic), not by the background waves' stores: taking half of the
// blocks away shortens it by a sixth.  Bit-exact either way (tests/test_raster_gpu.py::test_early_stores_outside_the_geometry_box).
struct PrefillJob {
    int mode;                  // -1: no prefill; 0 / 1 / 2: the raster kernel's mode (which outputs exist)
    int npre;                  // prefill workgroups per frame
    int first;                 // blockIdx.x of the first one
    int nwx;                   // 32x8 tiles per block row
    const int4* bbox;          // [B] (bx0, bx1, by0, by1) in 8x8 blocks, inclusive; bx0 > bx1: nothing is drawn in this frame (frame_bbox_kernel)
    float *rast, *rast_db, *normal, *texc, *texd;
    float* rgba;               // mode 2
    const float* bg_image;
    float bg_r, bg_g, bg_b;
    unsigned char* cid;
    const int* fid2cid;
    unsigned short* tile_ids;
};

// The frame's geometry box: one workgroup per frame, ahead of the binning launch (3 us; a step executor that already transforms the
// vertices can produce the same box in that kernel's epilogue instead).
__global__ __launch_bounds__(BIN_THREADS) void frame_bbox_kernel(const float* __restrict__ pos, int V, int H, int W, int nbx, int nby,
                                                                 int4* __restrict__ bbox) {
    __shared__ int red[BIN_THREADS / 64][5];
    const int b = blockIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const float hw = 8.0f * (float)W, hh = 8.0f * (float)H;
    int mnx = 0x7fffffff, mxx = -0x7fffffff - 1, mny = 0x7fffffff, mxy = -0x7fffffff - 1, full = 0;
    const float4* PV = reinterpret_cast<const float4*>(pos) + (size_t)b * V;
    for (int v = threadIdx.x; v < V; v += BIN_THREADS) {
        const float4 q = PV[v];
        int sx, sy;
        if (!(q.w > 0.0f) || !(__fadd_rn(q.z, q.w) >= 0.0f)) full = 1;          // near-plane pieces / dropped triangles: no statement about this frame
        else if (snap_vertex(q, hw, hh, sx, sy)) {                                 // (a vertex beyond the guard band: its triangles are dropped, tri_bbox)
            mnx = min(mnx, sx); mxx = max(mxx, sx);
            mny = min(mny, sy); mxy = max(mxy, sy);
        }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        mnx = min(mnx, __shfl_xor(mnx, o, 64)); mxx = max(mxx, __shfl_xor(mxx, o, 64));
        mny = min(mny, __shfl_xor(mny, o, 64)); mxy = max(mxy, __shfl_xor(mxy, o, 64));
        full |= __shfl_xor(full, o, 64);
    }
    if (lane == 0) { red[wave][0] = mnx; red[wave][1] = mxx; red[wave][2] = mny; red[wave][3] = mxy; red[wave][4] = full; }
    __syncthreads();
    if (threadIdx.x != 0) return;
#pragma unroll
    for (int w = 0; w < BIN_THREADS / 64; w++) {
        mnx = min(mnx, red[w][0]); mxx = max(mxx, red[w][1]);
        mny = min(mny, red[w][2]); mxy = max(mxy, red[w][3]);
        full |= red[w][4];
    }
    // the union of the triangles' pixel boxes (tri_bbox: the same monotone formulas on the extreme snapped coordinates)
    int4 bb = make_int4(1, 0, 1, 0);
    if (full) bb = make_int4(0, nbx - 1, 0, nby - 1);
    else if (mnx <= mxx) {
        const int px0 = max((mnx - 8 + 15) >> 4, 0), px1 = min((mxx - 8) >> 4, W - 1);
        const int py0 = max((mny - 8 + 15) >> 4, 0), py1 = min((mxy - 8) >> 4, H - 1);
        if (px0 <= px1 && py0 <= py1) bb = make_int4(px0 / BLK, px1 / BLK, py0 / BLK, py1 / BLK);
    }
    bbox[b] = bb;
}

// Early stores of one frame by workgroup p of pj.npre: a wave takes 64 consecutive pixels of ONE row at a time -- every output is then
// one contiguous run per store instruction (1 KB of rast, 768 B of normals, ...), the friendliest shape there is for the write path --
// and a lane stores its pixel iff the pixel's 8x8 block lies outside the box.
__device__ __forceinline__ void prefill_frame(int H, int W, const PrefillJob pj, int b, int p) {
    const int4 bb = pj.bbox[b];
    if (bb.x == 0 && bb.z == 0 && bb.y >= (W + BLK - 1) / BLK - 1 && bb.w >= (H + BLK - 1) / BLK - 1) return;      // the box is the frame
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int nseg = (W + 63) >> 6;
    const int nitem = H * nseg;
    int cid0 = 0;
    if (pj.mode == 2 && pj.cid) cid0 = pj.fid2cid[0];
    const float4 z4 = make_float4(0.f, 0.f, 0.f, 0.f);
    const size_t HW = (size_t)H * W;
    for (int it = p * (BIN_THREADS / 64) + wave; it < nitem; it += pj.npre * (BIN_THREADS / 64)) {
        const int py = it / nseg, px = (it - py * nseg) * 64 + lane;
        const int by = py >> 3, bx = px >> 3;
        if (px >= W) continue;
        if (!(bx < bb.x || bx > bb.y || by < bb.z || by > bb.w)) continue;       // inside the box: the raster kernel's block
        const size_t pidx = ((size_t)b * H + py) * W + px;
        reinterpret_cast<float4*>(pj.rast)[pidx] = z4;
        if (pj.mode == 2) {
            float4 o;
            if (pj.bg_image) {
                const float* g = pj.bg_image + (size_t)b * 3 * HW + (size_t)(H - 1 - py) * W + px;
                o = make_float4(g[0], g[HW], g[2 * HW], 0.0f);
            } else {
                o = make_float4(pj.bg_r, pj.bg_g, pj.bg_b, 0.0f);
            }
            reinterpret_cast<float4*>(pj.rgba)[pidx] = o;
            if (pj.cid) pj.cid[pidx] = (unsigned char)cid0;
            if (pj.tile_ids) pj.tile_ids[pidx] = (unsigned short)0xFFFF;
        } else {
            if (pj.rast_db) reinterpret_cast<float4*>(pj.rast_db)[pidx] = z4;
            if (pj.mode == 1) {
                float* no = pj.normal + 3 * pidx;
                no[0] = 0.f; no[1] = 0.f; no[2] = 0.f;
                reinterpret_cast<float2*>(pj.texc)[pidx] = make_float2(0.f, 0.f);
                reinterpret_cast<float4*>(pj.texd)[pidx] = z4;
            }
        }
    }
}

__global__ __launch_bounds__(BIN_THREADS) void bin_build_kernel(const float* __restrict__ pos, const int* __restrict__ tri,
                                                                const int* __restrict__ tri_uv, int V, int F, int H, int W,
                                                                int nbx, int nby, unsigned* __restrict__ trange,
                                                                TriRecord* __restrict__ records, uint2* __restrict__ frag,
                                                                unsigned* __restrict__ list, unsigned region, unsigned long long* prof,
                                                                int nfrag, const VnJob vj, const PrefillJob pj) {
    extern __shared__ __attribute__((aligned(16))) unsigned lb[];     // [nbin]: counts, then write cursors
    prof_begin(prof);
    if (pj.mode >= 0 && (int)blockIdx.x >= pj.first) {
        prefill_frame(H, W, pj, blockIdx.y, (int)blockIdx.x - pj.first);
        prof_end(prof);
        return;
    }
    if ((int)blockIdx.x >= nfrag) {
        // vhap_raster_bin_vnormal: the workgroups behind the binning ones compute the frame's vertex normals (independent work the raster
        // kernel needs too -- one launch instead of two on two queues with a hand-over each)
        const int v = ((int)blockIdx.x - nfrag) * BIN_THREADS + (int)threadIdx.x, b = blockIdx.y;
        if (v < V)
            vhap_vnormal_vertex(vj.verts + (size_t)b * V * 3, tri, vj.vc_ptr, vj.vc_idx, v, vj.vn + ((size_t)b * V + v) * 3,
                                vj.inv_len ? vj.inv_len + (size_t)b * V + v : nullptr);
        prof_end(prof);
        return;
    }
    __shared__ unsigned wtot[BIN_THREADS / 64];
    const int nbin = nbx * nby;
    const int b = blockIdx.y, wg = blockIdx.x;
    const int t = wg * BIN_THREADS + threadIdx.x;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < nbin; i += BIN_THREADS) lb[i] = 0u;
    __syncthreads();
    TriRanges R{TRANGE_NONE, TRANGE_NONE};
    if (t < F) R = triangle_setup(pos, tri, tri_uv, b, V, F, t, H, W, trange + (size_t)b * 2 * F, records + (size_t)b * 2 * F);
    BlockRange r{0, -1, 0, -1}, r1{0, -1, 0, -1};
    if (R.r0 != TRANGE_NONE) r = decode_range(R.r0, nby);
    if (R.r1 != TRANGE_NONE) r1 = decode_range(R.r1, nby);     // second piece of a triangle cut by the near plane (cold)
    for (int y = r.by0; y <= r.by1; y++)
        for (int x = r.bx0; x <= r.bx1; x++) atomicAdd(&lb[y * nbx + x], 1u);
    for (int y = r1.by0; y <= r1.by1; y++)
        for (int x = r1.bx0; x <= r1.bx1; x++) atomicAdd(&lb[y * nbx + x], 1u);
    __syncthreads();
    // exclusive scan of lb[] over the workgroup: every lane owns `per` consecutive bins
    const int per = (nbin + BIN_THREADS - 1) / BIN_THREADS;
    const int i0 = threadIdx.x * per, i1 = min(i0 + per, nbin);
    unsigned mine = 0u;
    for (int i = i0; i < i1; i++) mine += lb[i];
    unsigned incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const unsigned u = __shfl_up(incl, o, 64);
        if (lane >= o) incl += u;
    }
    if (lane == 63) wtot[wave] = incl;
    __syncthreads();
    unsigned pre = incl - mine, total = 0u;
    for (int w = 0; w < BIN_THREADS / 64; w++) {
        if (w < wave) pre += wtot[w];
        total += wtot[w];
    }
    const bool overflow = total > region;
    const unsigned base = (unsigned)((size_t)(b * nfrag + wg) * region);
    const int ntri = min(BIN_THREADS, F - wg * BIN_THREADS);
    for (int i = i0; i < i1; i++) {
        const unsigned n = lb[i];
        uint2 d = make_uint2(0u, 0u);
        // overflow: the raster blocks scan this workgroup's 2 * ntri record slots (ntri first pieces at wg * 1024 + j, then the ntri second
        // pieces at F + wg * 1024 + j) directly
        if (n) d = overflow ? make_uint2((unsigned)(wg * BIN_THREADS), FRAG_OVERFLOW | (unsigned)(2 * ntri)) : make_uint2(base + pre, n);
        frag[((size_t)b * nfrag + wg) * nbin + i] = d;     // workgroup-major: coalesced here, nfrag scattered 8-byte reads per raster wave
        lb[i] = pre;          // write cursor of this bin inside the region
        pre += n;
    }
    if (overflow) { prof_end(prof); return; }     // uniform
    __syncthreads();
    for (int y = r.by0; y <= r.by1; y++)
        for (int x = r.bx0; x <= r.bx1; x++) list[base + atomicAdd(&lb[y * nbx + x], 1u)] = (unsigned)t;
    for (int y = r1.by0; y <= r1.by1; y++)
        for (int x = r1.bx0; x <= r1.bx1; x++) list[base + atomicAdd(&lb[y * nbx + x], 1u)] = (unsigned)(F + t);
    prof_end(prof);
}

// ---- winner arithmetic: frag_common.h (same op order as shade_frag() in the oracle) ----
__device__ __forceinline__ unsigned f2ord(float f) {   // order-preserving float -> unsigned: negative ? ~u : u | sign  (branch-free: 3 VALU)
    const unsigned u = __float_as_uint(f);
    return u ^ ((unsigned)((int)u >> 31) | 0x80000000u);
}

__device__ __forceinline__ int sat30(long long e) {
    const long long lim = 1ll << 30;
    return (int)(e > lim ? lim : (e < -lim ? -lim : e));
}

__device__ __forceinline__ float rl(float v, int lane) {
    return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, v), lane));
}
__device__ __forceinline__ int rli(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }

struct RasterParams {
    const float* pos;
    const int* tri;
    const float* vnormal;  // [B,V,3]   (INTERP only)
    const float* uv;       // [VT,2]
    const int* tri_uv;     // [F,3]
    int B, V, VT, F, H, W, nbx, nby, nwx;  // nwx = workgroups per block row
    int row_mul;
    const unsigned* counts;
    unsigned* counts_w;
    unsigned* cursors_w;
    const unsigned* offsets;
    const unsigned* list;
    const unsigned* trange;
    const TriRecord* records;
    const BinHeader* hdr;
    unsigned capacity;
    const uint2* frag;     // fragmented lists (bin_build_kernel) or null: contiguous lists (bin_count/scan/fill)
    int nfrag;
    int debug;  // ablation switches for profiling only (vhap_debug_set_flags)
    float xs, xo, ys, yo;  // pixel centre -> NDC: fx = xs * px + xo (2/W, 1/W - 1; correctly rounded divisions done once on the host)
    float* rast;
    float* rast_db;
    float* normal;
    float* texc;
    float* texd;
    // deferred shading (mode 2)
    const float* tex;         // [Ht,Wt,3] ONE texture shared by the batch
    const float* mips;        // its pyramid (vhap_texture_mip_build)
    TexDesc D;
    const float* lights;      // [9,3]
    const float* sh_const;    // [9]
    const float* bg_image;    // [B,3,H,W] image space (row 0 = top) or null -> (bg_r, bg_g, bg_b)
    float bg_r, bg_g, bg_b;
    const int* fid2cid;       // [nfid] triangle id + 1 -> colour cluster, or null
    int nfid;
    float* rgba;              // [B,H,W,4] shaded + composited colour
    unsigned char* cid;       // [B,H,W] or null
    uint4* stats_part;        // per-wave partials of the diffuse-regulariser statistics ((max << 32 | ties) lo, hi, var sum, -) or null
    unsigned short* tile_ids; // [B,H,W] or null: uv tile of the texture-gradient binning (texbin) each pixel samples, 0xFFFF = background
    int NT;
    unsigned long long* prof; // VHAP_RASTER_PROFILE: (first start, last end) stamps of this kernel, PROF_SLOTS pairs; or null
    const int4* bbox;         // [B] block box of each frame's geometry when the binning launch prefilled everything outside it (PrefillJob), or null
};

// Residency (MI355X_MICROARCH.md, "Residency"): 256-thread workgroups per CU = min(8, 800 / (ceil(sgpr / 16) * 16 + 16), VGPR limit).  Left to
// itself the compiler takes 94-100 SGPRs for modes 0 / 1 (6 workgroups per CU instead of 8); capped at 80 it parks 19 of them in VGPR lanes, and
// 8 waves per SIMD cost mode 1 one VGPR (63 -> 64).  Measured (profiles/r04_call2_raster_residency_ab_*.txt): mode 1 81.0 -> 79.1 us alone,
// 91 -> 84 us in the step.  Mode 2 is held at 6 waves per SIMD (74 VGPRs; with the mip offsets computed in closed form -- tex_sample.h, level_off -- it
// takes 81 left to itself: 5 waves, 111 us against 104.7): forced to 8 it spills 8 VGPRs to scratch and runs 104 -> 123 us.
template <int MODE>
__global__ __launch_bounds__(256, MODE == 2 ? 6 : 8) __attribute__((amdgpu_num_sgpr(80))) void raster_kernel(const RasterParams P) {
    constexpr bool INTERP = MODE >= 1;
    prof_begin(P.prof);
    const unsigned L = vhap_xcd_remap(blockIdx.x, gridDim.x);
    const int nwg = P.nwx * P.nby;  // workgroups per frame
    const int b = L / nwg, wgi = L - b * nwg;
    const int wy_lin = wgi / P.nwx, wx = wgi - wy_lin * P.nwx;
    // visit block rows in a strided order (row_mul is coprime to nby): rows through the head are compute-heavy, background
    // rows are pure stores -- interleaving them in dispatch order lets the stores overlap the triangle loops
    const int wy = (int)(((long long)wy_lin * P.row_mul) % P.nby);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int bx = wx * WG_BLOCKS + wave, by = wy;  // this wave's 8x8 block
    if (bx >= P.nbx) {                              // whole wave outside the image
        if constexpr (MODE == 2) {
            if (P.stats_part && lane == 0) P.stats_part[(size_t)blockIdx.x * 4 + wave] = make_uint4(0u, 0u, 0u, 0u);
        }
        prof_end(P.prof);
        return;
    }
    const int bx0 = bx * BLK, by0 = by * BLK;
    const int dxp = lane & 7, dyp = lane >> 3;
    const int px = bx0 + dxp, py = by0 + dyp;
    const bool in_img = px < P.W && py < P.H;
    const int H = P.H, W = P.W;

    // a block outside the frame's geometry box was stored by the binning launch (prefill_frame): nothing to do here -- in mode 2 only the
    // wave's share of the shading statistics (the constant background value), computed below without touching pixel memory
    bool prefilled = false;
    if (P.bbox) {
        const int4 bb = P.bbox[b];
        prefilled = __builtin_amdgcn_readfirstlane((bx < bb.x || bx > bb.y || by < bb.z || by > bb.w) ? 1 : 0) != 0;
    }
    if constexpr (MODE != 2) {
        if (prefilled) { prof_end(P.prof); return; }
    }

    const bool fragmented = P.frag != nullptr;
    const bool use_list = fragmented || P.hdr->total <= P.capacity;
    const size_t bin = (size_t)b * P.nbx * P.nby + (size_t)by * P.nbx + bx;
    __shared__ int4 sfr[4][MAX_FRAG];   // per wave: fragment f = (first work item, count | overflow flag, list offset / first triangle, -)
    unsigned n, off = 0u;
    int nfr = 0;                  // non-empty fragments of this bin (the mesh is coherent in triangle order: usually 1-3)
    // (round 5, measured and dropped: the three background-image loads of mode 2 issued HERE, ahead of the fragment descriptors -- one round
    // trip less for every background wave -- change nothing: raster<2> 106.2 vs 104.1 us, step 0.8876 vs 0.8858 ms, 77 instead of 74
    // VGPRs; profiles/r05_call4_log.txt "debug 0 / debug 16384")
    if (MODE == 2 && prefilled) {
        n = 0u;
    } else if (fragmented) {
        uint2 d = make_uint2(0u, 0u);
        if (lane < P.nfrag) d = P.frag[((size_t)b * P.nfrag + lane) * ((size_t)P.nbx * P.nby) + (size_t)by * P.nbx + bx];
        const unsigned c = d.y & ~FRAG_OVERFLOW;
        // walk the non-empty fragments (usually 1-3; none for a background block) with scalar code: a wave-wide prefix scan here
        // would put six dependent cross-lane steps on the critical path of EVERY wave, background ones included
        unsigned long long nz = __ballot(c != 0u);
        unsigned run = 0u;
        while (nz) {
            const int l = __builtin_ctzll(nz);
            nz &= nz - 1;
            const unsigned dy = (unsigned)rli((int)d.y, l);
            if (lane == 0) sfr[wave][nfr] = make_int4((int)run, (int)dy, rli((int)d.x, l), 0);
            run += dy & ~FRAG_OVERFLOW;
            nfr++;
        }
        n = (P.debug & 1) ? 0u : run;
    } else {
        n = (P.debug & 1) ? 0u : (use_list ? P.counts[bin] : 2u * (unsigned)P.F);   // brute force: every record slot (two per triangle)
        off = use_list ? P.offsets[bin] : 0u;
        if (lane == 0) {   // leave the workspace clean for the next call (see VHAP_RASTER_WS_CLEAN)
            P.counts_w[bin] = 0u;
            P.cursors_w[bin] = 0u;
        }
    }

    // (experiment, profiles/r04_call4_split_launch_probe.txt -- debug flags 32 / 64: the pass as two launches, 32: blocks without candidates
    // leave at once, 64: only those store)
    if ((P.debug & 32) && n == 0u) return;
    if ((P.debug & 64) && n != 0u) return;
    unsigned long long best = ~0ull;  // (ordered z/w test value << 32) | triangle id

    const int bx1 = bx0 + BLK - 1, by1 = by0 + BLK - 1;
    const int cx = 16 * bx0 + 8, cy = 16 * by0 + 8;  // sub-pixel position of the block-origin pixel centre
    const int dx16 = dxp * 16, dy16 = dyp * 16;
    const float fdx = (float)dxp, fdy = (float)dyp;
    const TriRecord* REC = P.records + (size_t)b * 2 * P.F;     // record slots t and F + t: the (up to) two pieces of triangle t
    const unsigned* TR = P.trange + (size_t)b * 2 * P.F;

    __shared__ int4 sd[4][4][64];  // per-wave broadcast staging of the current chunk (4 KiB per wave)
    typedef short short2_t __attribute__((ext_vector_type(2)));
    const short2_t dxy = __builtin_bit_cast(short2_t, dx16 | (dy16 << 16));

    for (unsigned base = 0; base < n; base += 64) {
        const unsigned k = base + lane;
        bool hit = false, small = true;
        int t = 0;
        int AB0 = 0, AB1 = 0, AB2 = 0, C0 = 0, C1 = 0, C2 = 0;  // AB = A (low 16) | B (high 16)
        float zwc = 0.f, gx = 0.f, gy = 0.f;
        int4 vidx = make_int4(0, 0, 0, 0);                     // (i2, uv0, uv1, uv2) of the winner's lookup (i0, i1 ride in sd[2])
        int vi0 = 0, vi1 = 0;
        if (k < n && !(P.debug & 8)) {
            bool direct = !use_list;     // triangle id taken as is (no list): culled triangles have no record, test the range word
            if (fragmented) {
                int4 sel = make_int4(0, 0, 0, 0);
                for (int f = 0; f < nfr; f++) {              // broadcast LDS reads over the non-empty fragments
                    const int4 d = sfr[wave][f];
                    if ((int)k >= d.x) sel = d;
                }
                direct = ((unsigned)sel.y & FRAG_OVERFLOW) != 0u;
                const unsigned loc = k - (unsigned)sel.x;
                if (direct) {                 // overflowed workgroup: its ntri first-piece slots, then its ntri second-piece slots
                    const unsigned ntri = ((unsigned)sel.y & ~FRAG_OVERFLOW) >> 1;
                    t = (int)(loc < ntri ? (unsigned)sel.z + loc : (unsigned)P.F + (unsigned)sel.z + (loc - ntri));
                } else {
                    t = (int)P.list[(unsigned)sel.z + loc];
                }
            } else {
                t = use_list ? (int)P.list[off + k] : (int)k;
            }
            if (!direct || TR[t] != TRANGE_NONE) {  // direct mode (t < 2F by construction): skip culled triangles / absent pieces (no record)
                // the whole 80-byte record in one batch of loads: a wave's life is a chain of dependent global loads (descriptor ->
                // list -> record -> winner's vertices), each a full memory latency while the chip is saturated with stores, so the
                // bbox test must not gate a second round trip; the vertex / uv indices ride along for the same reason (below)
                const TriRecord* rp = REC + t;
                const int4 q0 = rp->q0, q1 = rp->q1;
                const float4 q2 = rp->q2;
                const int4 q3 = rp->q3, q4 = rp->q4;
                vi0 = q3.z; vi1 = q3.w;
                vidx = make_int4(q4.x, q4.y, q4.z, q4.w);
                asm volatile("" ::"v"(q0.x), "v"(q1.x), "v"(q2.x), "v"(q3.x));   // keeps the compiler from sinking these loads below the bbox test
                const int qx0 = q1.z & 0xffff, qx1 = q1.z >> 16, qy0 = q1.w & 0xffff, qy1 = q1.w >> 16;
                hit = !(qx1 < bx0 || qx0 > bx1 || qy1 < by0 || qy0 > by1);
                if (hit) {
                    const int sx0 = q0.x, sy0 = q0.y, sx1 = q0.z, sy1 = q0.w, sx2 = q1.x, sy2 = q1.y;
                    // edge i is opposite vertex i: a = v[(i+1)%3], b = v[(i+2)%3];  E = A*x + B*y + C,
                    // inside iff E > 0 or (E == 0 and top-left).  Fold the tie rule and a -1 into the
                    // constant so that inside <=> (E0|E1|E2) >= 0; saturate at +-2^30 (the value varies
                    // by < 2^30 over an 8x8 block, so the sign of every pixel survives).
                    const int A0 = sy1 - sy2, B0 = sx2 - sx1;
                    const int A1 = sy2 - sy0, B1 = sx0 - sx2;
                    const int A2 = sy0 - sy1, B2 = sx1 - sx0;
                    const int tl0 = (A0 > 0 || (A0 == 0 && B0 > 0)) ? 1 : 0;
                    const int tl1 = (A1 > 0 || (A1 == 0 && B1 > 0)) ? 1 : 0;
                    const int tl2 = (A2 > 0 || (A2 == 0 && B2 > 0)) ? 1 : 0;
                    // edges shorter than 2048 px: A, B fit int16 -> one v_dot2c_i32_i16 per edge and pixel
                    const int amax = max(max(abs(A0), abs(B0)), max(max(abs(A1), abs(B1)), max(abs(A2), abs(B2))));
                    small = amax < 32768;
                    AB0 = (A0 & 0xffff) | (B0 << 16);
                    AB1 = (A1 & 0xffff) | (B1 << 16);
                    AB2 = (A2 & 0xffff) | (B2 << 16);
                    // z/w test plane anchored at the block origin (depth_plane() in the oracle)
                    const double d1 = (double)q2.y - (double)q2.x, d2 = (double)q2.z - (double)q2.x;
                    const double inv = __hiloint2double(q3.y, q3.x);
                    double e1d, e2d;
                    if (amax < 16384) {
                        // edges shorter than 1024 px (practically all): the block overlaps the triangle's bounding box, so the block
                        // origin is within amax + 136 sub-pixels of every vertex and each edge function fits 31 bits -- 24-bit
                        // multiplies (full rate; 64-bit mads are quarter rate), no saturation, one int->double conversion.  Same
                        // integers as the 64-bit path below, hence the same bits.
                        const int e0 = __mul24(A0, cx - sx1) + __mul24(B0, cy - sy1);
                        const int e1 = __mul24(A1, cx - sx2) + __mul24(B1, cy - sy2);
                        const int e2 = __mul24(A2, cx - sx0) + __mul24(B2, cy - sy0);
                        C0 = e0 + tl0 - 1;
                        C1 = e1 + tl1 - 1;
                        C2 = e2 + tl2 - 1;
                        e1d = (double)e1;
                        e2d = (double)e2;
                    } else {
                        const long long E0 = (long long)A0 * (cx - sx1) + (long long)B0 * (cy - sy1);
                        const long long E1 = (long long)A1 * (cx - sx2) + (long long)B1 * (cy - sy2);
                        const long long E2 = (long long)A2 * (cx - sx0) + (long long)B2 * (cy - sy0);
                        C0 = sat30(E0 + tl0 - 1);
                        C1 = sat30(E1 + tl1 - 1);
                        C2 = sat30(E2 + tl2 - 1);
                        e1d = (double)E1;
                        e2d = (double)E2;
                    }
                    zwc = (float)((double)q2.x + (e1d * d1 + e2d * d2) * inv);
                    gx = (float)((((double)A1 * d1 + (double)A2 * d2) * 16.0) * inv);
                    gy = (float)((((double)B1 * d1 + (double)B2 * d2) * 16.0) * inv);
                }
            }
        }
        if (P.debug & 4) hit = false;
        // Publish this chunk's per-triangle constants in LDS; every pixel lane then reads triangle j with
        // three broadcast ds_read_b128 (measured: 10 v_readlane cost ~57 cycles per triangle on gfx950,
        // the LDS broadcast hides completely behind the VALU work).  Same-wave DS ops execute in order,
        // so no barrier is needed.
        sd[wave][0][lane] = make_int4(AB0, AB1, AB2, C0);
        sd[wave][1][lane] = make_int4(C1, C2, __float_as_int(zwc), __float_as_int(gx));
        const int tid = t >= P.F ? t - P.F : t;          // a record slot's triangle id (slot F + t: second piece of a near-clipped triangle t)
        sd[wave][2][lane] = make_int4(__float_as_int(gy), (tid << 6) | lane, vi0, vi1);   // key word: triangle id, then its slot in this chunk
        sd[wave][3][lane] = vidx;
        // Coverage loop, software-pipelined: the broadcast reads of the NEXT triangle are issued before the arithmetic of the current
        // one, so the DS latency (~100 cycles, paid per triangle otherwise) overlaps the ~17 VALU instructions of the test.
        // (software-pipelining this loop -- next triangle's DS reads before the current arithmetic -- was measured 3 % SLOWER: the
        // kernel is sensitive to VALU issue slots, the DS latency is already hidden by the other waves)
        unsigned long long mask = __ballot(hit && small);
        while (mask) {
            const int j = __builtin_ctzll(mask);
            mask &= mask - 1;
            const int4 x = sd[wave][0][j], y = sd[wave][1][j];
            const int2 z = *reinterpret_cast<const int2*>(&sd[wave][2][j]);
            const int e0 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x.x), dxy, x.w, false);
            const int e1 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x.y), dxy, y.x, false);
            const int e2 = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, x.z), dxy, y.y, false);
            const float zt = __fmaf_rn(__int_as_float(y.w), fdx, __fmaf_rn(__int_as_float(z.x), fdy, __int_as_float(y.z)));
            const bool inside = ((e0 | e1 | e2) >= 0) & (fabsf(zt) <= 1.0f);
            const unsigned long long key = ((unsigned long long)f2ord(zt) << 32) | (unsigned)z.y;
            const bool take = inside & (key < best);
            best = take ? key : best;
        }
        // rare: triangles with an edge longer than 2048 px -- exact 64-bit edge functions per pixel
        unsigned long long mbig = __ballot(hit && !small);
        while (mbig) {
            const int j = __builtin_ctzll(mbig);
            mbig &= mbig - 1;
            const int tj = rli(t, j);
            const int4 q0 = REC[tj].q0, q1 = REC[tj].q1;
            const int sx0 = q0.x, sy0 = q0.y, sx1 = q0.z, sy1 = q0.w, sx2 = q1.x, sy2 = q1.y;
            const int A0 = sy1 - sy2, B0 = sx2 - sx1, A1 = sy2 - sy0, B1 = sx0 - sx2, A2 = sy0 - sy1, B2 = sx1 - sx0;
            const int pcx = 16 * px + 8, pcy = 16 * py + 8;
            const long long E0 = (long long)A0 * (pcx - sx1) + (long long)B0 * (pcy - sy1) + ((A0 > 0 || (A0 == 0 && B0 > 0)) ? 0 : -1);
            const long long E1 = (long long)A1 * (pcx - sx2) + (long long)B1 * (pcy - sy2) + ((A1 > 0 || (A1 == 0 && B1 > 0)) ? 0 : -1);
            const long long E2 = (long long)A2 * (pcx - sx0) + (long long)B2 * (pcy - sy0) + ((A2 > 0 || (A2 == 0 && B2 > 0)) ? 0 : -1);
            const float zt = __fmaf_rn(rl(gx, j), fdx, __fmaf_rn(rl(gy, j), fdy, rl(zwc, j)));
            const bool inside = ((E0 | E1 | E2) >= 0) && (zt >= -1.0f && zt <= 1.0f);
            const unsigned long long key = ((unsigned long long)f2ord(zt) << 32) | (unsigned)(((tj >= P.F ? tj - P.F : tj) << 6) | j);
            if (inside && key < best) best = key;
        }
    }

    if ((P.debug & 2) && best != 12345ull) return;  // ablation: no stores
    if constexpr (MODE != 2) {
        if (__ballot(in_img) == 0ull) prof_end(P.prof);
        if (!in_img) return;
    }
    const size_t pidx = in_img ? ((size_t)b * H + py) * W + px : 0;
    float4 o_rast = make_float4(0.f, 0.f, 0.f, 0.f), o_db = o_rast;
    FragAttr at;
    at.n0 = at.n1 = at.n2 = at.tu = at.tv = 0.f;
    at.td = o_rast;
    const bool cov = in_img && best != ~0ull;
    if (cov) {
        const int t = (int)((unsigned)best >> 6), slot = (int)((unsigned)best & 63u);
        int i0, i1, i2;
        int4 q4;
        if (n <= 64u) {          // single chunk (the common case): the winner's indices are still staged in LDS -- one round trip less
            const int4 a = sd[wave][2][slot];
            q4 = sd[wave][3][slot];
            i0 = a.z; i1 = a.w; i2 = q4.x;
        } else {
            const TriRecord* rp = REC + t;
            const int4 q3 = rp->q3;
            q4 = rp->q4;
            i0 = q3.z; i1 = q3.w; i2 = q4.x;
        }
        const float4* PV = reinterpret_cast<const float4*>(P.pos) + (size_t)b * P.V;
        const float4 p0 = PV[i0], p1 = PV[i1], p2 = PV[i2];
        const float fx = __fmaf_rn(P.xs, (float)px, P.xo), fy = __fmaf_rn(P.ys, (float)py, P.yo);
        const Frag fr = shade_frag(p0, p1, p2, fx, fy);
        o_rast = make_float4(fr.b0, fr.b1, fr.zw, (float)(t + 1));
        o_db = frag_db(p0, p1, p2, fr, P.xs, P.ys);
        if constexpr (INTERP)
            at = frag_attr(P.vnormal + (size_t)b * P.V * 3, reinterpret_cast<const float2*>(P.uv), i0, i1, i2, q4.y, q4.z, q4.w, fr, o_db);
    }
    if constexpr (MODE == 2) {
        // ---- deferred shading: texture sample, SH diffuse, composite over the background (render_nvdiffrast.py:386-421) ----
        float var = 0.f;
        unsigned long long mx = 0ull;
        if (in_img) {
            SH9 bsh;
            float nx, ny, nz, inv, d[3];
            sh_diffuse(at.n0, at.n1, at.n2, P.sh_const, P.lights, bsh, nx, ny, nz, inv, d);   // background: normal 0 -> the constant bands
            float4 o_rgba;
            if (cov) {
                float alb[3];
                // (the two levels' taps one after the other: fetched in one batch -- tex_fetch, as the shading backward does -- this kernel needs
                // 81 VGPRs and, held to 6 waves per SIMD, runs 107 us against 104.7: profiles/r04_call26_load_batching_ab.txt)
                tex_sample<3>(P.tex, P.mips, P.D, 0, make_float2(at.tu, at.tv), at.td, alb);
                o_rgba = make_float4(alb[0] * d[0], alb[1] * d[1], alb[2] * d[2], 1.0f);
            } else if (prefilled) {
                o_rgba = make_float4(0.f, 0.f, 0.f, 0.f);         // (stored by the binning launch)
            } else if (P.bg_image) {
                const size_t HW = (size_t)H * W;
                const float* g = P.bg_image + (size_t)b * 3 * HW + (size_t)(H - 1 - py) * W + px;
                o_rgba = make_float4(g[0], g[HW], g[2 * HW], 0.0f);
            } else {
                o_rgba = make_float4(P.bg_r, P.bg_g, P.bg_b, 0.0f);
            }
            if (!prefilled) {
                reinterpret_cast<float4*>(P.rast)[pidx] = o_rast;
                reinterpret_cast<float4*>(P.rgba)[pidx] = o_rgba;
                if (P.cid) P.cid[pidx] = (unsigned char)P.fid2cid[min(max((int)o_rast.w, 0), P.nfid - 1)];
                if (P.tile_ids) P.tile_ids[pidx] = (unsigned short)(cov ? tile_of(make_float2(at.tu, at.tv), P.NT) : 0xFFFF);
            }
            if (P.stats_part) {
                const float mean = (d[0] + d[1] + d[2]) * (1.0f / 3.0f);
                var = 0.5f * ((d[0] - mean) * (d[0] - mean) + (d[1] - mean) * (d[1] - mean) + (d[2] - mean) * (d[2] - mean));
                float mxv = d[0];
                unsigned mxn = 1u;
#pragma unroll
                for (int c = 1; c < 3; c++) {
                    if (d[c] > mxv) { mxv = d[c]; mxn = 1u; }
                    else if (d[c] == mxv) mxn++;
                }
                mx = ((unsigned long long)sh_f2ord(mxv) << 32) | mxn;
            }
        }
        if (P.stats_part) {         // statistics of the diffuse regulariser (tracker.py:547-550): one partial per wave, no atomics
            if (__ballot(cov) == 0ull && __ballot(in_img) == ~0ull) {
                // a full background block (70 % of the waves): every lane shaded the same constant normal -- 64 identical values, whose
                // tree sum is exactly 64 v and whose maximum is the value itself with 64 times its ties: no cross-lane traffic at all
                var *= 64.0f;
                mx = (mx & 0xffffffff00000000ull) | ((mx & 0xffffffffull) * 64ull);
            } else {
                // DPP reductions on the VALU (18 ds_bpermute per wave before): sum of the variances; the maximum of the ordered high
                // words, then the tie counts of the lanes that hold it
                var = vhap_wave_sum_dpp(var);
                const unsigned hi = (unsigned)(mx >> 32);
                const unsigned wmax = vhap_wave_max_u32_dpp(hi);
                const unsigned ties = vhap_wave_sum_u32_dpp(hi == wmax ? (unsigned)mx : 0u);
                mx = ((unsigned long long)wmax << 32) | ties;
            }
            if (lane == 0) P.stats_part[(size_t)blockIdx.x * 4 + wave] = make_uint4((unsigned)mx, (unsigned)(mx >> 32), __float_as_uint(var), 0u);
        }
        prof_end(P.prof);
        return;
    }
    reinterpret_cast<float4*>(P.rast)[pidx] = o_rast;
    if (P.rast_db) reinterpret_cast<float4*>(P.rast_db)[pidx] = o_db;
    if constexpr (MODE == 1) {
        float* no = P.normal + 3 * pidx;
        no[0] = at.n0; no[1] = at.n1; no[2] = at.n2;
        reinterpret_cast<float2*>(P.texc)[pidx] = make_float2(at.tu, at.tv);
        reinterpret_cast<float4*>(P.texd)[pidx] = at.td;
    }
    if (P.prof && lane == __builtin_ctzll(__ballot(true))) atomicMax(&P.prof[2 * (blockIdx.x % PROF_SLOTS) + 1], (unsigned long long)wall_clock64());
}

// final reduction of the per-wave shading statistics -> stats[4] in the layout of vhap_shade_fwd: (ties, ordered max, var sum, -).
// STATS_BLOCKS workgroups reduce a slice each into `part2`; the one that finishes last (device-scope counter, reset for the next call)
// folds the slices: a single launch of ~4 us instead of one workgroup walking 65536 partials (23 us on the forward critical path).
constexpr int STATS_BLOCKS = 64;
__global__ __launch_bounds__(1024) void shade_stats_reduce_kernel(const uint4* __restrict__ part, int n, uint4* __restrict__ part2,
                                                                  unsigned* __restrict__ counter, unsigned* __restrict__ stats) {
    __shared__ float rv[16];
    __shared__ unsigned long long rm[16];
    __shared__ bool last;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    auto block_reduce = [&](float var, unsigned long long mx, float& v_out, unsigned long long& m_out) {
        var = vhap_wave_sum(var);
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            const unsigned lo = (unsigned)__shfl_xor((int)(unsigned)mx, o, 64), hi = (unsigned)__shfl_xor((int)(unsigned)(mx >> 32), o, 64);
            mx = sh_merge_max(mx, ((unsigned long long)hi << 32) | lo);
        }
        if (lane == 0) { rv[wave] = var; rm[wave] = mx; }
        __syncthreads();
        v_out = 0.f;
        m_out = 0ull;
        if (threadIdx.x == 0)
            for (int w = 0; w < 16; w++) { v_out += rv[w]; m_out = sh_merge_max(m_out, rm[w]); }
        __syncthreads();
    };
    const int per = (n + gridDim.x - 1) / gridDim.x;
    const int i0 = blockIdx.x * per, i1 = min(i0 + per, n);
    float var = 0.f;
    unsigned long long mx = 0ull;
    for (int i = i0 + threadIdx.x; i < i1; i += 1024) {
        const uint4 p = part[i];
        var += __uint_as_float(p.z);
        mx = sh_merge_max(mx, ((unsigned long long)p.y << 32) | p.x);
    }
    float v;
    unsigned long long m;
    block_reduce(var, mx, v, m);
    if (threadIdx.x == 0) {
        part2[blockIdx.x] = make_uint4((unsigned)m, (unsigned)(m >> 32), __float_as_uint(v), 0u);
        __threadfence();
        last = atomicAdd(counter, 1u) == gridDim.x - 1;
    }
    __syncthreads();
    if (!last) return;
    __threadfence();
    var = 0.f;
    mx = 0ull;
    if (threadIdx.x < gridDim.x) {
        const volatile unsigned* q = reinterpret_cast<const volatile unsigned*>(part2 + threadIdx.x);   // written by other workgroups
        const unsigned p0 = q[0], p1 = q[1], p2 = q[2];
        var = __uint_as_float(p2);
        mx = ((unsigned long long)p1 << 32) | p0;
    }
    block_reduce(var, mx, v, m);
    if (threadIdx.x == 0) {
        stats[0] = (unsigned)m;
        stats[1] = (unsigned)(m >> 32);
        stats[2] = __float_as_uint(v);
        stats[3] = 0u;
        *counter = 0u;                       // clean for the next call
    }
}


struct WsLayout {
    size_t hdr, counts, cursors, offsets, trange, records, list, frag, stats, stats2, prof, bbox, total;
};

WsLayout ws_layout(int B, int F, int nbin, size_t cap, size_t npart) {
    WsLayout l;
    auto al = [](size_t x) { return (x + 255) & ~(size_t)255; };
    size_t o = 0;
    l.hdr = o; o = al(o + sizeof(BinHeader));
    l.counts = o; o = al(o + sizeof(unsigned) * (size_t)B * nbin);
    l.cursors = o; o = al(o + sizeof(unsigned) * (size_t)B * nbin);
    l.offsets = o; o = al(o + sizeof(unsigned) * (size_t)B * nbin);
    l.trange = o; o = al(o + sizeof(unsigned) * (size_t)B * F * 2);      // two slots per triangle (near-plane clipping)
    l.records = o; o = al(o + sizeof(TriRecord) * (size_t)B * F * 2);
    l.list = o; o = al(o + sizeof(unsigned) * (cap ? cap : 1));
    l.frag = o; o = al(o + sizeof(uint2) * (size_t)B * nbin * ((F + BIN_THREADS - 1) / BIN_THREADS));
    l.stats = o; o = al(o + sizeof(uint4) * npart);                 // per-wave shading-statistics partials (mode 2): 4 per raster workgroup
    l.stats2 = o; o = al(o + sizeof(uint4) * (STATS_BLOCKS + 1));    // second-level partials + the completion counter
    l.prof = o; o = al(o + sizeof(unsigned long long) * 2 * PROF_SLOTS * 2);      // VHAP_RASTER_PROFILE stamps: binning, then raster kernel
    l.bbox = o; o = al(o + sizeof(int4) * (size_t)B);               // per-frame block box of the geometry (PrefillJob)
    l.total = o;
    return l;
}

int check_dims(int B, int V, int F, int H, int W) {
    if (B <= 0 || V <= 0 || F <= 0 || H <= 0 || W <= 0) return VHAP_E_BADDIM;
    if (H > 4096 || W > 4096 || F >= (1 << 24)) return VHAP_E_BADDIM;
    if ((long long)B * F >= (1ll << 30) || B > 65535) return VHAP_E_BADDIM;
    return VHAP_OK;
}

template <int MODE>
int launch_raster(RasterParams P, void* ws, size_t ws_bytes, size_t cap, int flags, hipStream_t st, float* stats_out = nullptr,
                  const VnJob vj = VnJob{}) {
    const int B = P.B, F = P.F;
    P.nbx = (P.W + BLK - 1) / BLK;
    P.nby = (P.H + BLK - 1) / BLK;
    P.nwx = (P.nbx + WG_BLOCKS - 1) / WG_BLOCKS;
    P.xs = 2.0f / (float)P.W;
    P.xo = 1.0f / (float)P.W - 1.0f;
    P.ys = 2.0f / (float)P.H;
    P.yo = 1.0f / (float)P.H - 1.0f;
    const int nbin = P.nbx * P.nby;
    if ((long long)B * nbin >= (1ll << 31) || cap > 0xfffffff0u) return VHAP_E_BADDIM;
    const WsLayout l = ws_layout(B, F, nbin, cap, (size_t)B * P.nwx * P.nby * 4);
    if (!ws) return VHAP_E_NULLPTR;
    if (ws_bytes < l.total) return VHAP_E_WORKSPACE;
    char* w = static_cast<char*>(ws);
    BinHeader* hdr = reinterpret_cast<BinHeader*>(w + l.hdr);
    unsigned* counts = reinterpret_cast<unsigned*>(w + l.counts);
    unsigned* cursors = reinterpret_cast<unsigned*>(w + l.cursors);
    unsigned* offsets = reinterpret_cast<unsigned*>(w + l.offsets);
    unsigned* trange = reinterpret_cast<unsigned*>(w + l.trange);
    unsigned* list = reinterpret_cast<unsigned*>(w + l.list);
    TriRecord* records = reinterpret_cast<TriRecord*>(w + l.records);
    const dim3 gbin(vhap_cdiv(F, BIN_THREADS), B);
    const bool use_lds = nbin <= LDS_BIN_LIMIT;
    const int nfrag = (int)gbin.x;
    const size_t region = cap / ((size_t)B * nfrag);       // pair-list entries owned by one (frame, 1024-triangle workgroup)
    const bool fragmented = use_lds && nfrag <= MAX_FRAG && region >= 1 && !(vhap_g_debug_flags & 4096);   // (flag 4096: A/B switch)
    P.frag = nullptr;
    P.nfrag = nfrag;
    const bool prebinned = (flags & VHAP_RASTER_PREBINNED) != 0;
    if (((flags & (VHAP_RASTER_BIN_ONLY | VHAP_RASTER_PREBINNED)) || vj.verts) && !fragmented) return VHAP_E_UNSUPPORTED;   // split calls: one-launch binning only
    if (fragmented && prebinned) {
        P.frag = reinterpret_cast<uint2*>(w + l.frag);
        if ((flags & VHAP_RASTER_PREFILL) && !(vhap_g_debug_flags & 8192)) P.bbox = reinterpret_cast<const int4*>(w + l.bbox);    // (the BIN_ONLY call prefilled with the same flag)
    } else if (fragmented) {
        // binning in ONE launch: per-workgroup regions of the pair list, fragment descriptors instead of global counters
        const size_t lds = sizeof(unsigned) * nbin;
        if (lds > 65536 &&
            hipFuncSetAttribute(reinterpret_cast<const void*>(bin_build_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
            return VHAP_E_HIP;
        uint2* frag = reinterpret_cast<uint2*>(w + l.frag);
        unsigned long long* prof_bin = nullptr;
        if (flags & VHAP_RASTER_PROFILE) {
            prof_bin = reinterpret_cast<unsigned long long*>(w + l.prof);
            prof_init_kernel<<<vhap_cdiv(2 * PROF_SLOTS, 256), 256, 0, st>>>(prof_bin, 2 * PROF_SLOTS);
            VHAP_LAUNCH_CHECK();
        }
        const int nvn = vj.verts ? vhap_cdiv(P.V, BIN_THREADS) : 0;
        PrefillJob pj{};
        pj.mode = -1;
        // only on request (VHAP_RASTER_PREFILL; split calls: BOTH carry it, the BIN_ONLY call then needs the outputs): measured, it LOSES on
        // MI355X -- profiles/r05_call2_early_stores_v1_ab.txt, r05_call3_early_stores_v2_ab.txt
        if (P.rast && (flags & VHAP_RASTER_PREFILL) && !(vhap_g_debug_flags & 8192)) {   // (8192: A/B switch)
            pj.mode = MODE;
            // ONE round of workgroups over the chip together with the binning (and vertex-normal) ones: they all start at once
            const int room = 2 * 256 - (int)gbin.x * B - nvn * B;
            pj.npre = std::max(1, std::min(room / B, 64));
            pj.first = (int)gbin.x + nvn;
            pj.nwx = P.nwx;
            int4* bbox = reinterpret_cast<int4*>(w + l.bbox);
            frame_bbox_kernel<<<B, BIN_THREADS, 0, st>>>(P.pos, P.V, P.H, P.W, P.nbx, P.nby, bbox);
            VHAP_LAUNCH_CHECK();
            pj.bbox = bbox;
            pj.rast = P.rast; pj.rast_db = P.rast_db; pj.normal = P.normal; pj.texc = P.texc; pj.texd = P.texd;
            pj.rgba = P.rgba; pj.bg_image = P.bg_image; pj.bg_r = P.bg_r; pj.bg_g = P.bg_g; pj.bg_b = P.bg_b;
            pj.cid = P.cid; pj.fid2cid = P.fid2cid; pj.tile_ids = P.tile_ids;
        }
        const dim3 gbv(gbin.x + nvn + (pj.mode >= 0 ? pj.npre : 0), B);
        bin_build_kernel<<<gbv, BIN_THREADS, lds, st>>>(P.pos, P.tri, P.tri_uv, P.V, F, P.H, P.W, P.nbx, P.nby, trange, records, frag, list,
                                                       (unsigned)region, prof_bin, nfrag, vj, pj);
        VHAP_LAUNCH_CHECK();
        P.frag = frag;
        if (pj.mode >= 0) P.bbox = pj.bbox;
    } else {
        // header, counts and cursors are contiguous: one zero-fill launch -- skipped when the caller vouches that the workspace was
        // zero-initialised once and only ever used by completed calls of this function (every call leaves it clean again)
        if (!(flags & VHAP_RASTER_WS_CLEAN)) {
            vhap_zero_async(w + l.hdr, l.offsets - l.hdr, st);
            VHAP_LAUNCH_CHECK();
        }
        const size_t fill_lds = use_lds ? 2 * sizeof(unsigned) * nbin : 0;
        if (fill_lds > 65536) {  // raise the dynamic-LDS cap (160 KiB per CU on gfx950)
            if (hipFuncSetAttribute(reinterpret_cast<const void*>(bin_fill_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                    (int)fill_lds) != hipSuccess)
                return VHAP_E_HIP;
        }
        bin_count_kernel<<<gbin, BIN_THREADS, use_lds ? sizeof(unsigned) * nbin : 0, st>>>(P.pos, P.tri, P.tri_uv, P.V, F, P.H, P.W, P.nbx,
                                                                                        P.nby, counts, trange, records, hdr);
        VHAP_LAUNCH_CHECK();
        bin_scan_kernel<<<vhap_cdiv((long long)B * nbin, 256), 256, 0, st>>>(counts, offsets, B * nbin, hdr);
        VHAP_LAUNCH_CHECK();
        bin_fill_kernel<<<gbin, BIN_THREADS, fill_lds, st>>>(trange, F, P.nbx, P.nby, offsets, cursors, list, hdr, (unsigned)cap);
        VHAP_LAUNCH_CHECK();
    }
    if (flags & VHAP_RASTER_BIN_ONLY) return VHAP_OK;
    P.counts = counts;
    P.counts_w = counts;
    P.cursors_w = cursors;
    {   // odd multiplier near 0.38 * nby, made coprime to nby
        int m = (int)(0.381966f * (float)P.nby) | 1;
        auto gcd = [](int a, int b) { while (b) { const int t = a % b; a = b; b = t; } return a; };
        while (m > 1 && gcd(m, P.nby) != 1) m -= 2;
        P.row_mul = m < 1 ? 1 : m;
        if (!(vhap_g_debug_flags & 16)) P.row_mul = 1;   // A/B switch (profiling): strided row order off by default
    }
    P.offsets = offsets;
    P.list = list;
    P.trange = trange;
    P.records = records;
    P.hdr = hdr;
    P.capacity = (unsigned)cap;
    P.debug = vhap_g_debug_flags;
    const int nwg = B * P.nwx * P.nby;
    if (MODE == 2 && stats_out) P.stats_part = reinterpret_cast<uint4*>(w + l.stats);
    if (flags & VHAP_RASTER_PROFILE) {
        P.prof = reinterpret_cast<unsigned long long*>(w + l.prof) + 2 * PROF_SLOTS;
        prof_init_kernel<<<vhap_cdiv(2 * PROF_SLOTS, 256), 256, 0, st>>>(P.prof, 2 * PROF_SLOTS);
        VHAP_LAUNCH_CHECK();
    }
    raster_kernel<MODE><<<nwg, 256, 0, st>>>(P);
    VHAP_LAUNCH_CHECK();
    if (MODE == 2 && stats_out && !(flags & VHAP_RASTER_STATS_LATER)) {
        uint4* part2 = reinterpret_cast<uint4*>(w + l.stats2);
        unsigned* counter = reinterpret_cast<unsigned*>(part2 + STATS_BLOCKS);
        if (!(flags & VHAP_RASTER_WS_CLEAN)) {      // (a zero-initialised workspace keeps the counter at 0 between calls)
            vhap_zero_async(counter, sizeof(uint4), st);
            VHAP_LAUNCH_CHECK();
        }
        shade_stats_reduce_kernel<<<STATS_BLOCKS, 1024, 0, st>>>(P.stats_part, nwg * 4, part2, counter, reinterpret_cast<unsigned*>(stats_out));
        VHAP_LAUNCH_CHECK();
    }
    return VHAP_OK;
}

}  // namespace


extern "C" size_t vhap_raster_workspace_bytes(int B, int F, int H, int W, size_t pair_capacity) {
    if (check_dims(B, 1, F, H, W) != VHAP_OK) return 0;
    const int nbx = (W + BLK - 1) / BLK, nby = (H + BLK - 1) / BLK;
    return ws_layout(B, F, nbx * nby, pair_capacity, (size_t)B * ((nbx + WG_BLOCKS - 1) / WG_BLOCKS) * nby * 4).total;
}

extern "C" int vhap_raster_fwd(const float* pos, const int32_t* tri, int B, int V, int F, int H, int W, float* rast,
                               float* rast_db, void* workspace, size_t workspace_bytes, size_t pair_capacity,
                               int flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !rast) return VHAP_E_NULLPTR;
    if (int e = check_dims(B, V, F, H, W)) return e;
    RasterParams P{};
    P.pos = pos; P.tri = tri; P.B = B; P.V = V; P.F = F; P.H = H; P.W = W;
    P.rast = rast; P.rast_db = rast_db;
    return launch_raster<0>(P, workspace, workspace_bytes, pair_capacity, flags, vhap_stream(stream));
}

extern "C" int vhap_raster_interp_fwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv,
                                      const int32_t* tri_uv, int B, int V, int VT, int F, int H, int W, float* rast,
                                      float* rast_db, float* normal, float* texc, float* texd, void* workspace,
                                      size_t workspace_bytes, size_t pair_capacity, int flags, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !vnormal || !uv || !tri_uv || !rast || !rast_db || !normal || !texc || !texd) return VHAP_E_NULLPTR;
    if (int e = check_dims(B, V, F, H, W)) return e;
    if (VT <= 0) return VHAP_E_BADDIM;
    RasterParams P{};
    P.pos = pos; P.tri = tri; P.vnormal = vnormal; P.uv = uv; P.tri_uv = tri_uv;
    P.B = B; P.V = V; P.VT = VT; P.F = F; P.H = H; P.W = W;
    P.rast = rast; P.rast_db = rast_db; P.normal = normal; P.texc = texc; P.texd = texd;
    return launch_raster<1>(P, workspace, workspace_bytes, pair_capacity, flags, vhap_stream(stream));
}

extern "C" int vhap_raster_shade_fwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                     const float* tex, const float* mips, int Ht, int Wt, const float* lights, const float* sh_const,
                                     const float* bg_image, const float* bg_color, const int32_t* fid2cid, int nfid, int B, int V,
                                     int VT, int F, int H, int W, float* rast, float* rgba, uint8_t* cid, float* stats, uint16_t* tile_ids,
                                     void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags, vhap_stream_t stream) {
    VHAP_ENTER();
    const bool bin_only = (flags & VHAP_RASTER_BIN_ONLY) != 0;
    if (!pos || !tri || !uv || !tri_uv) return VHAP_E_NULLPTR;
    if (!bin_only && (!vnormal || !tex || !lights || !sh_const || !rast || !rgba || (!bg_image && !bg_color))) return VHAP_E_NULLPTR;
    if (cid && (!fid2cid || nfid <= 0)) return VHAP_E_NULLPTR;
    if (int e = check_dims(B, V, F, H, W)) return e;
    if (VT <= 0 || Ht <= 0 || Wt <= 0) return VHAP_E_BADDIM;
    RasterParams P{};
    P.pos = pos; P.tri = tri; P.vnormal = vnormal; P.uv = uv; P.tri_uv = tri_uv;
    P.B = B; P.V = V; P.VT = VT; P.F = F; P.H = H; P.W = W;
    P.rast = rast; P.rgba = rgba; P.cid = cid; P.fid2cid = fid2cid; P.nfid = nfid;
    P.tile_ids = tile_ids; P.NT = texbin_nt(Ht, Wt);
    P.tex = tex; P.mips = mips; P.D = make_desc(1, Ht, Wt, 3);
    if (P.D.L > 0 && !mips && !bin_only) return VHAP_E_NULLPTR;
    P.lights = lights; P.sh_const = sh_const; P.bg_image = bg_image;
    if (!bg_image && bg_color) { P.bg_r = bg_color[0]; P.bg_g = bg_color[1]; P.bg_b = bg_color[2]; }
    return launch_raster<2>(P, workspace, workspace_bytes, pair_capacity, flags, vhap_stream(stream), stats);
}

// The BIN_ONLY call of vhap_raster_shade_fwd with the frames' vertex normals (vhap_vnormal_fwd_saved) computed by extra workgroups of the
// same launch: the raster pass needs both, neither needs the other.
extern "C" int vhap_raster_bin_vnormal(const float* pos, const int32_t* tri, const int32_t* tri_uv, int B, int V, int F, int H, int W,
                                       void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags, const float* verts,
                                       const int32_t* vc_ptr, const int32_t* vc_idx, float* vn, float* inv_len, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !tri_uv || !verts || !vc_ptr || !vc_idx || !vn) return VHAP_E_NULLPTR;
    if (int e = check_dims(B, V, F, H, W)) return e;
    RasterParams P{};
    P.pos = pos; P.tri = tri; P.tri_uv = tri_uv;
    P.B = B; P.V = V; P.F = F; P.H = H; P.W = W;
    return launch_raster<2>(P, workspace, workspace_bytes, pair_capacity, (flags & ~VHAP_RASTER_PREBINNED) | VHAP_RASTER_BIN_ONLY,
                            vhap_stream(stream), nullptr, VnJob{verts, vc_ptr, vc_idx, vn, inv_len});
}

// vhap_raster_bin_vnormal + the EARLY STORES of the deferred-shading pass (PrefillJob): further workgroups of the same launch store rast /
// rgba (background composite) / cid / tile_ids of every 8x8 block outside the frame's geometry box; the vhap_raster_shade_fwd call that
// follows must carry VHAP_RASTER_PREBINNED | VHAP_RASTER_PREFILL and the same output pointers.
extern "C" int vhap_raster_bin_vnormal_prefill(const float* pos, const int32_t* tri, const int32_t* tri_uv, int B, int V, int F, int H, int W,
                                               void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags, const float* verts,
                                               const int32_t* vc_ptr, const int32_t* vc_idx, float* vn, float* inv_len,
                                               const float* bg_image, const float* bg_color, const int32_t* fid2cid, int nfid,
                                               float* rast, float* rgba, uint8_t* cid, uint16_t* tile_ids, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!pos || !tri || !tri_uv || !verts || !vc_ptr || !vc_idx || !vn || !rast || !rgba || (!bg_image && !bg_color)) return VHAP_E_NULLPTR;
    if (cid && (!fid2cid || nfid <= 0)) return VHAP_E_NULLPTR;
    if (int e = check_dims(B, V, F, H, W)) return e;
    RasterParams P{};
    P.pos = pos; P.tri = tri; P.tri_uv = tri_uv;
    P.B = B; P.V = V; P.F = F; P.H = H; P.W = W;
    P.rast = rast; P.rgba = rgba; P.cid = cid; P.fid2cid = fid2cid; P.nfid = nfid; P.tile_ids = tile_ids;
    P.bg_image = bg_image;
    if (!bg_image && bg_color) { P.bg_r = bg_color[0]; P.bg_g = bg_color[1]; P.bg_b = bg_color[2]; }
    return launch_raster<2>(P, workspace, workspace_bytes, pair_capacity, (flags & ~VHAP_RASTER_PREBINNED) | VHAP_RASTER_BIN_ONLY | VHAP_RASTER_PREFILL,
                            vhap_stream(stream), nullptr, VnJob{verts, vc_ptr, vc_idx, vn, inv_len});
}

// The statistics reduction a vhap_raster_shade_fwd(..., VHAP_RASTER_STATS_LATER) call left out (same workspace, same sizes): nothing on the
// pixel chain reads `stats` before the energy assembly, so a step executor issues it beside the chain instead of inside it.
extern "C" int vhap_raster_shade_stats(int B, int F, int H, int W, void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags,
                                       float* stats, vhap_stream_t stream) {
    VHAP_ENTER();
    if (!workspace || !stats) return VHAP_E_NULLPTR;
    if (int e = check_dims(B, 1, F, H, W)) return e;
    const int nbx = (W + BLK - 1) / BLK, nby = (H + BLK - 1) / BLK, nwx = (nbx + WG_BLOCKS - 1) / WG_BLOCKS;
    const int nwg = B * nwx * nby;
    const WsLayout l = ws_layout(B, F, nbx * nby, pair_capacity, (size_t)nwg * 4);
    if (workspace_bytes < l.total) return VHAP_E_WORKSPACE;
    char* w = static_cast<char*>(workspace);
    hipStream_t st = vhap_stream(stream);
    uint4* part2 = reinterpret_cast<uint4*>(w + l.stats2);
    unsigned* counter = reinterpret_cast<unsigned*>(part2 + STATS_BLOCKS);
    if (!(flags & VHAP_RASTER_WS_CLEAN)) {
        vhap_zero_async(counter, sizeof(uint4), st);
        VHAP_LAUNCH_CHECK();
    }
    shade_stats_reduce_kernel<<<STATS_BLOCKS, 1024, 0, st>>>(reinterpret_cast<const uint4*>(w + l.stats), nwg * 4, part2, counter,
                                                            reinterpret_cast<unsigned*>(stats));
    VHAP_LAUNCH_CHECK();
    return VHAP_OK;
}

// VHAP_RASTER_PROFILE: byte offset inside `workspace` of the stamps -- 2 x 256 pairs of uint64 (first start, last end; 100 MHz ticks):
// pairs [0,256) belong to the binning kernel, [256,512) to the raster kernel; reduce with min over the starts / max over the ends.
extern "C" size_t vhap_raster_profile_offset(int B, int F, int H, int W, size_t pair_capacity) {
    if (check_dims(B, 1, F, H, W) != VHAP_OK) return 0;
    const int nbx = (W + BLK - 1) / BLK, nby = (H + BLK - 1) / BLK;
    return ws_layout(B, F, nbx * nby, pair_capacity, (size_t)B * ((nbx + WG_BLOCKS - 1) / WG_BLOCKS) * nby * 4).prof;
}
