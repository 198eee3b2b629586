/*
 * vhap_hip.h -- C ABI of libvhap_hip.so, the MI355X (gfx950) implementation of the
 * photometric FLAME-fitting hot path of ShenhanQian/VHAP.
 *
 * Every entry point is stateless and re-entrant: the caller owns all buffers (device pointers,
 * row-major, contiguous, fp32 / int32), passes an explicit workspace where one is needed, and
 * the HIP stream to enqueue on (as void*, i.e. a hipStream_t; NULL = default stream).  Calls
 * return immediately after enqueueing.  Return value: VHAP_OK (0) or a negative VHAP_E_* code;
 * nothing throws across this boundary.  vhap_strerror() names a code.
 *
 * The functions replace, one for one, what the reference binds from nvdiffrast
 * (`import nvdiffrast.torch as dr`, vhap/util/render_nvdiffrast.py:12) plus the eager-torch
 * arithmetic around it; the reference call site each one replaces is cited per function.
 * Conventions (pixel centres, fill rule, depth rule, culling): DESIGN.md section 3, identical to
 * oracle/raster_oracle.c.
 */
#ifndef VHAP_HIP_H
#define VHAP_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define VHAP_ABI_VERSION 1

#define VHAP_OK 0
#define VHAP_E_NULLPTR (-1)   /* a required pointer is NULL */
#define VHAP_E_BADDIM (-2)    /* a dimension is out of range (B,V,F,H,W <= 0, H/W > 4096, F >= 2^24, ...) */
#define VHAP_E_WORKSPACE (-3) /* workspace too small (see *_workspace_bytes) */
#define VHAP_E_HIP (-4)       /* a HIP runtime call / kernel launch failed */
#define VHAP_E_UNSUPPORTED (-5)

typedef void* vhap_stream_t;

int vhap_abi_version(void);
const char* vhap_strerror(int code);
/* Profiling-only ablation switches (bit 0: rasterizer skips triangle work, bit 1: skips stores).
 * Never set in production; 0 restores normal behaviour. */
void vhap_debug_set_flags(int flags);

/* ---------------------------------------------------------------------------------------------
 * Rasterize: replaces dr.rasterize(glctx, pos, tri, resolution)  (render_nvdiffrast.py:254,257)
 *   pos  [B,V,4] clip-space positions, tri [F,3] int32
 *   rast [B,H,W,4] = (u, v, z/w, float(tri_id+1)); rast_db [B,H,W,4] = (du/dX,du/dY,dv/dX,dv/dY)
 *   (rast_db may be NULL).  Row 0 = bottom (y-up), back faces culled.
 * Workspace: tile bins (counts, offsets, triangle lists).  `pair_capacity` = number of
 * (triangle,tile) pairs the lists can hold; if a batch needs more the kernel falls back to a
 * brute-force per-tile scan (slower, same result), so any capacity >= 0 is correct.
 * ------------------------------------------------------------------------------------------- */
size_t vhap_raster_workspace_bytes(int B, int F, int H, int W, size_t pair_capacity);
int vhap_raster_fwd(const float* pos, const int32_t* tri, int B, int V, int F, int H, int W,
                    float* rast, float* rast_db, void* workspace, size_t workspace_bytes,
                    size_t pair_capacity, vhap_stream_t stream);

/* Fused rasterize + interpolate ("RI-fwd", the G-buffer pass): replaces dr.rasterize followed by
 * dr.interpolate(v_normal, rast, tri) (render_nvdiffrast.py:384) and
 * dr.interpolate(uv[None], rast, tri_uv, rast_db, diff_attrs='all') (:389) in ONE launch.
 *   vnormal [B,V,3], uv [VT,2] (shared by all frames), tri_uv [F,3]
 *   normal [B,H,W,3], texc [B,H,W,2], texd [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY) of uv        */
int vhap_raster_interp_fwd(const float* pos, const int32_t* tri, const float* vnormal,
                           const float* uv, const int32_t* tri_uv, int B, int V, int VT, int F,
                           int H, int W, float* rast, float* rast_db, float* normal, float* texc,
                           float* texd, void* workspace, size_t workspace_bytes,
                           size_t pair_capacity, vhap_stream_t stream);

/* Rasterize backward: replaces nvdiffrast's RasterizeGradKernel(Db).
 *   d_rast [B,H,W,4] (only .xy is used, like nvdiffrast), d_rast_db [B,H,W,4] or NULL
 *   d_pos  [B,V,4] is ACCUMULATED into (caller zero-fills).                                        */
int vhap_raster_bwd(const float* pos, const int32_t* tri, const float* rast, const float* d_rast,
                    const float* d_rast_db, int B, int V, int F, int H, int W, float* d_pos,
                    vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Interpolate: replaces dr.interpolate(attr, rast, tri, rast_db, diff_attrs)  (:384, :389)
 *   attr [AB,V,A] with AB in {1,B}; out [B,H,W,A]; out_da [B,H,W,2A] (NULL when rast_db is NULL),
 *   laid out (da_k/dX, da_k/dY) per attribute k.
 * Backward accumulates into d_attr [AB,V,A], d_rast [B,H,W,4] (.xy), d_rast_db [B,H,W,4] (may be
 * NULL); the caller zero-fills d_attr; d_rast / d_rast_db are overwritten.
 * ------------------------------------------------------------------------------------------- */
int vhap_interp_fwd(const float* attr, int AB, const float* rast, const int32_t* tri,
                    const float* rast_db, int B, int H, int W, int V, int F, int A, float* out,
                    float* out_da, vhap_stream_t stream);
int vhap_interp_bwd(const float* attr, int AB, const float* rast, const int32_t* tri,
                    const float* rast_db, const float* d_out, const float* d_out_da, int B, int H,
                    int W, int V, int F, int A, float* d_attr, float* d_rast, float* d_rast_db,
                    vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Texture: replaces dr.texture(tex, uv, uv_da, filter_mode='linear-mipmap-linear',
 * boundary 'wrap', max_mip_level=None)  (:399) and filter_mode='linear' (render_uvmap.py:41).
 *   tex [TB,Ht,Wt,C] with TB in {1,B} (the reference replicates one texture B times,
 *   tracker.py:234; pass TB=1 to share it), C <= 4.
 *   The mip pyramid lives in a caller-owned buffer of vhap_texture_mip_floats() floats that
 *   holds levels 1..L (level 0 is `tex` itself); build it once per optimiser step.
 *   uv [B,H,W,2]; uv_da [B,H,W,4] (NULL -> plain bilinear on level 0); out [B,H,W,C].
 * Backward: d_tex [TB,Ht,Wt,C] and d_mips are ACCUMULATED with atomics (caller zero-fills
 * both), then vhap_texture_mip_fold() folds d_mips down into d_tex.  d_uv [B,H,W,2] and
 * d_uv_da [B,H,W,4] are overwritten (either may be NULL).
 * ------------------------------------------------------------------------------------------- */
int vhap_texture_num_levels(int Ht, int Wt);
size_t vhap_texture_mip_floats(int TB, int Ht, int Wt, int C);
int vhap_texture_mip_build(const float* tex, int TB, int Ht, int Wt, int C, float* mips,
                           vhap_stream_t stream);
int vhap_texture_fwd(const float* tex, const float* mips, int TB, int Ht, int Wt, int C,
                     const float* uv, const float* uv_da, int B, int H, int W, float* out,
                     vhap_stream_t stream);
int vhap_texture_bwd(const float* tex, const float* mips, int TB, int Ht, int Wt, int C,
                     const float* uv, const float* uv_da, const float* d_out, int B, int H, int W,
                     float* d_tex, float* d_mips, float* d_uv, float* d_uv_da,
                     vhap_stream_t stream);
int vhap_texture_mip_fold(float* d_tex, float* d_mips, int TB, int Ht, int Wt, int C,
                          vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Antialias: replaces dr.antialias(color, rast, pos, tri)  (:465).
 *   opp [F,3] int32 = static edge->opposite-vertex table (-1 = boundary edge) built once on the
 *   host from the fixed topology (vhap_amd.topology.build_opposite_table) -- it replaces the
 *   edge hash nvdiffrast rebuilds on every call.
 *   work: caller-owned int32 buffer of vhap_antialias_work_ints(B,H,W) ints; the forward records
 *   the pixel pairs it blended there and the backward replays them.
 * Backward: d_color [B,H,W,C] overwritten; d_pos [B,V,4] ACCUMULATED (caller zero-fills).
 * ------------------------------------------------------------------------------------------- */
size_t vhap_antialias_work_ints(int B, int H, int W);
int vhap_antialias_fwd(const float* color, const float* rast, const float* pos,
                       const int32_t* tri, const int32_t* opp, int B, int H, int W, int C, int V,
                       int F, float* out, int32_t* work, vhap_stream_t stream);
int vhap_antialias_bwd(const float* color, const float* rast, const float* pos,
                       const int32_t* tri, const int32_t* opp, const float* d_out,
                       const int32_t* work, int B, int H, int W, int C, int V, int F,
                       float* d_color, float* d_pos, vhap_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VHAP_HIP_H */
