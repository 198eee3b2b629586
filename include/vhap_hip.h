/*
 * vhap_hip.h -- C ABI of libvhap_hip.so, the MI355X (gfx950) implementation of the
 * photometric FLAME-fitting hot path of ShenhanQian/VHAP.
 *
 * Every entry point is stateless and re-entrant -- the library has NO mutable global state (since ABI 2 the per-call behaviour
 * switches that ABI 1 kept in a process-wide variable are the `call_flags` argument of the entry points that honour them):
 * the caller owns all buffers (device pointers,
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

#define VHAP_ABI_VERSION 10

#define VHAP_OK 0
#define VHAP_E_NULLPTR (-1)   /* a required pointer is NULL */
#define VHAP_E_BADDIM (-2)    /* a dimension is out of range (B,V,F,H,W <= 0, H/W > 4096, F >= 2^24, ...) */
#define VHAP_E_WORKSPACE (-3) /* workspace too small (see *_workspace_bytes) */
#define VHAP_E_HIP (-4)       /* a HIP runtime call / kernel launch failed */
#define VHAP_E_UNSUPPORTED (-5)

typedef void* vhap_stream_t;

int vhap_abi_version(void);
const char* vhap_strerror(int code);
/* Streams owned by the caller of this library (hipStreamCreateWithPriority, non-blocking; priority 0 = normal, or, high_priority != 0,
 * the greatest priority the device offers).  The entry points below take any stream; these two exist so that a host which gets its streams from a POOL -- torch
 * hands out 32 streams per device round-robin, so the 33rd torch.cuda.Stream() IS the first one again -- can hold streams that alias
 * nothing else in the process.  The step executor forks work onto side streams inside a stream capture and launches the resulting graph
 * on a stream of its own: an alias between those roles serialises the branches at best and crashed hipGraphLaunch (ROCm 7) at worst. */
int vhap_stream_create(vhap_stream_t* stream, int high_priority);
int vhap_stream_destroy(vhap_stream_t stream);
/* call_flags (an argument of the entry points that honour them; 0 = the plain behaviour):
 *   VHAP_CALL_ACC_PREZEROED       the small accumulators this call adds into (terms / energy / stats / out2 of frame_prep, landmark,
 *                                 tex_prep, offset_reg, shade, photo; d_coef of flame_skin_bwd) were zero-filled by the caller --
 *                                 a step executor keeps all of them in one arena cleared by ONE launch; the call skips its own clear
 *   VHAP_CALL_AA_PASSTHROUGH_DONE vhap_antialias_bwd: d_color already holds a copy of d_out (skip the pass-through copy)
 *   VHAP_CALL_ADAM_KEEP_STEP      vhap_adam_step: do not advance the step counter (a further call of the same step follows)
 *   VHAP_CALL_ADAM_STEP_ADVANCED  vhap_adam_step: the counter was advanced for this step already (vhap_adam_advance at its head): use it as
 *                                 it is and leave it alone -- the pieces of one step may then run in any order on any streams */
#define VHAP_CALL_ACC_PREZEROED 1
#define VHAP_CALL_AA_PASSTHROUGH_DONE 2
#define VHAP_CALL_ADAM_KEEP_STEP 4
#define VHAP_CALL_ADAM_STEP_ADVANCED 16
#define VHAP_CALL_TEXBIN_COUNTED 8      /* (internal to vhap_texture_grad_binned_counted) */
/*   VHAP_CALL_OFFSET_PER_FRAME    vhap_frame_prep_fwd / _bwd, vhap_flame_skin_fwd / _clip_fwd: the vertex offset argument is [B,V,3] -- one row
 *                                 per frame of the batch (static_offset + dynamic_offset[timesteps], `use_dynamic_offset`, tracker.py:213-235)
 *                                 -- instead of one [V,3] offset shared by the batch; g_offset of vhap_frame_prep_bwd is then [B,V,3] too */
#define VHAP_CALL_OFFSET_PER_FRAME 32

/* ---------------------------------------------------------------------------------------------
 * Rasterize: replaces dr.rasterize(glctx, pos, tri, resolution)  (render_nvdiffrast.py:254,257)
 *   pos  [B,V,4] clip-space positions, tri [F,3] int32
 *   rast [B,H,W,4] = (u, v, z/w, float(tri_id+1)); rast_db [B,H,W,4] = (du/dX,du/dY,dv/dX,dv/dY)
 *   (rast_db may be NULL).  Row 0 = bottom (y-up), back faces culled.
 * Workspace: tile bins (counts, offsets, triangle lists).  `pair_capacity` = number of
 * (triangle,tile) pairs the lists can hold; if a batch needs more the kernel falls back to a
 * brute-force per-tile scan (slower, same result), so any capacity >= 0 is correct.
 * ------------------------------------------------------------------------------------------- */
/* flags: VHAP_RASTER_WS_CLEAN -- the caller guarantees that `workspace` was zero-filled once and has since been used only
 * by COMPLETED calls of vhap_raster_fwd / vhap_raster_interp_fwd with the same (B, F, H, W, pair_capacity): every call
 * leaves the bin counters zeroed again, so the per-call memset is skipped.  Without the flag any memory may be passed. */
#define VHAP_RASTER_WS_CLEAN 1
/* Split calls (so that work the raster kernel depends on but the binning does not -- vertex normals, the texture pyramid -- can run next
 * to the binning on another stream): VHAP_RASTER_BIN_ONLY runs only the triangle set-up + binning into `workspace`;
 * VHAP_RASTER_PREBINNED skips it and rasterises from the bins a BIN_ONLY call with the same (pos, tri, B, F, H, W, pair_capacity) left
 * there.  Only with the one-launch binning (F <= 32768, H*W/64 <= 16384 bins): otherwise VHAP_E_UNSUPPORTED. */
#define VHAP_RASTER_BIN_ONLY 2
#define VHAP_RASTER_PREBINNED 4
/* VHAP_RASTER_PROFILE (measurement only): the binning and raster kernels stamp the wall clock of their first wave's start and last wave's
 * end into `workspace` at vhap_raster_profile_offset(): 2 x 256 pairs of uint64 (100 MHz ticks; pairs [0,256) = binning kernel, [256,512)
 * = raster kernel; duration = max(ends) - min(starts)) -- the only way to time a kernel INSIDE a captured graph replay (HIP refuses to read
 * event-record nodes, a profiler is not always there).  Costs two tiny launches and one atomic per wave. */
#define VHAP_RASTER_PROFILE 8
/* VHAP_RASTER_STATS_LATER (vhap_raster_shade_fwd): leave the reduction of the per-wave shading statistics to a later
 * vhap_raster_shade_stats call on the same workspace (nothing on the pixel chain reads them before the energy is assembled). */
#define VHAP_RASTER_STATS_LATER 16
/* VHAP_RASTER_PREFILL (ABI 9): EARLY STORES.  The binning launch also reduces, per frame, the 8x8-block bounding box of the clip positions
 * and stores the output of every block outside it (zeros; deferred shading: the background composite) with workgroups of its own; the
 * raster kernel's waves of those blocks leave at once.  Opt-in: measured on MI355X it makes the pass LONGER (csrc/raster.hip,
 * profiles/r05_call3_early_stores_v2_ab.txt); kept as a switch with its parity test.  Split calls do it when BOTH carry the flag: the
 * VHAP_RASTER_BIN_ONLY call then needs the output pointers (vhap_raster_bin_vnormal_prefill), the VHAP_RASTER_PREBINNED call reads
 * the boxes from `workspace`.  Results are bit-identical with and without. */
#define VHAP_RASTER_PREFILL 32
size_t vhap_raster_profile_offset(int B, int F, int H, int W, size_t pair_capacity);
size_t vhap_raster_workspace_bytes(int B, int F, int H, int W, size_t pair_capacity);
int vhap_raster_fwd(const float* pos, const int32_t* tri, int B, int V, int F, int H, int W,
                    float* rast, float* rast_db, void* workspace, size_t workspace_bytes,
                    size_t pair_capacity, int flags, vhap_stream_t stream);

/* Fused rasterize + interpolate ("RI-fwd", the G-buffer pass): replaces dr.rasterize followed by
 * dr.interpolate(v_normal, rast, tri) (render_nvdiffrast.py:384) and
 * dr.interpolate(uv[None], rast, tri_uv, rast_db, diff_attrs='all') (:389) in ONE launch.
 *   vnormal [B,V,3], uv [VT,2] (shared by all frames), tri_uv [F,3]
 *   normal [B,H,W,3], texc [B,H,W,2], texd [B,H,W,4] = (du/dX, du/dY, dv/dX, dv/dY) of uv        */
int vhap_raster_interp_fwd(const float* pos, const int32_t* tri, const float* vnormal,
                           const float* uv, const int32_t* tri_uv, int B, int V, int VT, int F,
                           int H, int W, float* rast, float* rast_db, float* normal, float* texc,
                           float* texd, void* workspace, size_t workspace_bytes,
                           size_t pair_capacity, int flags, vhap_stream_t stream);

/* Rasterize backward: replaces nvdiffrast's RasterizeGradKernel(Db).
 *   d_rast [B,H,W,4] (only .xy is used, like nvdiffrast), d_rast_db [B,H,W,4] or NULL
 *   d_pos  [B,V,4] is ACCUMULATED into (caller zero-fills).                                        */
int vhap_raster_bwd(const float* pos, const int32_t* tri, const float* rast, const float* d_rast,
                    const float* d_rast_db, int B, int V, int F, int H, int W, float* d_pos,
                    vhap_stream_t stream);

/* DEFERRED SHADING: the G-buffer pass fused with everything up to the composited colour -- replaces dr.rasterize (:254), both
 * dr.interpolate calls (:384, :389), dr.texture (:399), safe_normalize (:386), the SH shading, rgb = albedo * diffuse, alpha = coverage
 * and the background composite with its y-flip (:402-421) in ONE launch.  The interpolated normal / uv / uv derivatives and the sampled
 * albedo never leave registers: per pixel the kernel writes rast (16 B), rgba (16 B) and optionally the colour-cluster byte (for the
 * disturbance pass) instead of the 97 B the separate passes write and the 88 B they read back; the backward re-computes them
 * (vhap_deferred_shade_bwd).
 *   tex [Ht,Wt,3] ONE texture shared by all frames (tracker.py:234 replicates it B x) + mips (vhap_texture_mip_build)
 *   lights [9,3], sh_const [9]; bg_image [B,3,H,W] image space (row 0 = top) or NULL -> bg_color (HOST pointer to 3 floats)
 *   fid2cid [nfid] (index = triangle id + 1) / cid [B,H,W] uint8: optional
 *   rast [B,H,W,4] as vhap_raster_fwd; rgba [B,H,W,4] renderer space (row 0 = bottom): shaded colour, alpha = coverage
 *   stats (4 words, may be NULL): as vhap_shade_fwd, OVERWRITTEN (a tiny second launch reduces per-wave partials kept in `workspace`)
 *   tile_ids [B,H,W] uint16 (may be NULL): the uv tile of vhap_texture_grad_binned each covered pixel samples, 0xFFFF on the background --
 *   lets the texture-gradient sort run during the forward pass (vhap_texbin_sort_ids), off the backward's critical path
 * With VHAP_RASTER_BIN_ONLY only pos / tri / uv / tri_uv / the dimensions / workspace are used. */
int vhap_raster_shade_fwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv,
                          const int32_t* tri_uv, const float* tex, const float* mips, int Ht, int Wt,
                          const float* lights, const float* sh_const, const float* bg_image,
                          const float* bg_color, const int32_t* fid2cid, int nfid, int B, int V, int VT,
                          int F, int H, int W, float* rast, float* rgba, uint8_t* cid, float* stats,
                          uint16_t* tile_ids, void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags,
                          vhap_stream_t stream);
/* The VHAP_RASTER_BIN_ONLY call of vhap_raster_shade_fwd with the vertex normals of vhap_vnormal_fwd_saved (compute_v_normals,
 * render_nvdiffrast.py:216-232; verts [B,V,3] world space, vertex->corner CSR, vn [B,V,3], inv_len [B,V] or NULL) computed by extra
 * workgroups of the same launch -- the raster pass needs both, neither needs the other.  VHAP_E_UNSUPPORTED as BIN_ONLY. */
int vhap_raster_bin_vnormal(const float* pos, const int32_t* tri, const int32_t* tri_uv, int B, int V, int F, int H, int W,
                            void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags, const float* verts,
                            const int32_t* vc_ptr, const int32_t* vc_idx, float* vn, float* inv_len, vhap_stream_t stream);
/* vhap_raster_bin_vnormal + the early stores of the deferred-shading pass (VHAP_RASTER_PREFILL, above): rast / rgba (background composite:
 * bg_image [B,3,H,W] image space or bg_color[3]) / cid (fid2cid[0]) / tile_ids (0xFFFF) of every block outside the frame's geometry box.
 * Follow with vhap_raster_shade_fwd(..., VHAP_RASTER_PREBINNED | VHAP_RASTER_PREFILL) on the same outputs and workspace.
 * Replaces (reference): the background part of dr.rasterize + the composite, render_nvdiffrast.py:254,405-421. */
int vhap_raster_bin_vnormal_prefill(const float* pos, const int32_t* tri, const int32_t* tri_uv, int B, int V, int F, int H, int W,
                                    void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags, const float* verts,
                                    const int32_t* vc_ptr, const int32_t* vc_idx, float* vn, float* inv_len,
                                    const float* bg_image, const float* bg_color, const int32_t* fid2cid, int nfid,
                                    float* rast, float* rgba, uint8_t* cid, uint16_t* tile_ids, vhap_stream_t stream);

/* The statistics reduction a vhap_raster_shade_fwd(..., VHAP_RASTER_STATS_LATER) call left out: same sizes, workspace and flags. */
int vhap_raster_shade_stats(int B, int F, int H, int W, void* workspace, size_t workspace_bytes, size_t pair_capacity, int flags,
                            float* stats, vhap_stream_t stream);
/* Backward of the shading part of vhap_raster_shade_fwd (everything between the interpolated attributes and rgba): per covered pixel the
 * normal / uv / uv derivatives are re-computed from (rast, geometry) with the forward's arithmetic, the texture is re-sampled, and the
 * upstream gradient is chained through rgb = albedo * diffuse and the SH shading.  The upstream gradient is either the image
 * d_rgba [B,H,W,4], or (d_rgba == NULL) the photometric gradient computed on the fly, -sign(gt - pred) * d_sum[0] (tracker.py:430-439;
 * pred_rgba [B,H,W,4] renderer space, gt_nchw [B,3,H,W] image space, d_sum device scalar: what vhap_photo_bwd would have written) plus
 * d_delta [B,H,W,4] if given (the sparse colour part of vhap_antialias_photo_bwd);
 * in both cases x keep [B,H,W] if given (= the colour-disturbance backward).
 * Outputs, all [B,H,W,*] and OVERWRITTEN (d_albedo: zeros on background pixels; the others are written on covered pixels only):
 *   texc [..,2], texd [..,4], d_albedo [..,3]  -> the texture-gradient accumulation (vhap_texture_grad_binned / vhap_texture_bwd)
 *   d_normal [..,3], d_texc [..,2], d_texd [..,4] -> vhap_gbuffer_bwd
 *   d_lights [9,3] ACCUMULATED (photometric part + the diffuse regulariser: d_reg device scalar and stats as in vhap_shade_bwd, may be NULL);
 *   work: vhap_deferred_shade_bwd_work_floats(B,H,W) floats, ZERO on entry (partial sums of d_lights are accumulated there, the
 *         call leaves them in it; required when d_lights != NULL)
 *   texbin_work (may be NULL): the workspace of vhap_texture_grad_binned with its first 2 x 64 x 64 x 4 bytes ZERO on entry -- the uv-tile
 *         histogram (the count pass of the binning) is filled in here, from registers; follow with vhap_texture_grad_binned_counted()
 *   tile_ids [B,H,W] uint16 (may be NULL): the uv tile (of vhap_texture_grad_binned) each pixel's texture gradient falls into, 0xFFFF = none --
 *         follow with vhap_texture_grad_binned_ids(), whose sorting passes then read 2 B per pixel instead of uv + d_albedo
 *   call_flags: VHAP_CALL_DELTA_UNSCALED -- d_delta holds the antialias backward's colour part per unit of d_sum (written by
 *         vhap_photo_fwd_total's antialias job, which runs beside the photometric sum and therefore cannot know d_sum): multiplied here
 * Replaces vhap_photo_bwd (optionally) + vhap_shade_bwd + the d_uv / d_uv_da part of vhap_texture_bwd (and the re-reading of five
 * G-buffer images). */
#define VHAP_CALL_DELTA_UNSCALED 128
/*   VHAP_CALL_TEX_TERMS_CONSUME (ABI 10)  vhap_photo_fwd_total: tex_terms[0..1] are set to zero once the energy assembly has read them -- they
 *                                 were accumulated by the previous step's vhap_tex_finish_carry, which this step's will do again */
#define VHAP_CALL_TEX_TERMS_CONSUME 256
/*   VHAP_CALL_SKIP_BG_GRAD (ABI 10)  vhap_deferred_shade_bwd: d_albedo of BACKGROUND pixels is left unwritten (12 B x two thirds of a head
 *                                 frame) -- for a caller whose texture-gradient pass reads d_albedo through a list of covered pixels only
 *                                 (vhap_texbin_sort_ids + vhap_texture_grad_binned_sorted); ignored when tile_ids is handed in */
#define VHAP_CALL_SKIP_BG_GRAD 512
size_t vhap_deferred_shade_bwd_work_floats(int B, int H, int W);
/* d_lights == NULL with a work table: only the partial sums are accumulated into `work`; finish later (off the critical path) with this */
int vhap_deferred_lights_reduce(const float* work, const float* lights, const float* sh_const, const float* d_reg,
                                const float* stats, int B, int H, int W, float* d_lights, vhap_stream_t stream);
int vhap_deferred_shade_bwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv,
                            const int32_t* tri_uv, const float* tex, const float* mips, int Ht, int Wt,
                            const float* lights, const float* sh_const, const float* rast,
                            const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                            const float* d_sum, const float* d_delta, const float* keep, const float* d_reg,
                            const float* stats, int B, int V, int VT, int F, int H, int W, float* texc, float* texd,
                            float* d_albedo, float* d_normal, float* d_texc, float* d_texd,
                            float* d_lights, float* work, size_t work_floats, void* texbin_work,
                            uint16_t* tile_ids, int call_flags, vhap_stream_t stream);
/* vhap_deferred_shade_bwd over the COVERED pixels only: thread k takes pixel covered_list[k], k < B*H*W - *n_background (both written on the device by
 * vhap_disturb_inplace_list; the list in pixel order, so a wave is 64 consecutive covered pixels of a row).  On a head frame two thirds of the pixels are
 * background, which the pass has nothing to do for: in row order 40 % of its waves were background waves held by mixed workgroups and the working
 * waves 76 % full.  Requires VHAP_CALL_SKIP_BG_GRAD and tile_ids == NULL (nothing is written for background pixels); same results. */
int vhap_deferred_shade_bwd_list(const float* pos, const int32_t* tri, const float* vnormal, const float* uv, const int32_t* tri_uv,
                                 const float* tex, const float* mips, int Ht, int Wt, const float* lights, const float* sh_const,
                                 const float* rast, const float* d_rgba, const float* pred_rgba, const float* gt_nchw,
                                 const float* d_sum, const float* d_delta, const float* keep, const float* d_reg, const float* stats,
                                 int B, int V, int VT, int F, int H, int W, float* texc, float* texd, float* d_albedo,
                                 float* d_normal, float* d_texc, float* d_texd, float* d_lights, float* work, size_t work_floats,
                                 void* texbin_work, uint16_t* tile_ids, const uint32_t* covered_list, const int32_t* n_background,
                                 int call_flags, vhap_stream_t stream);

/* Triangle-parallel backward of the fused G-buffer pass (vhap_raster_interp_fwd): chains the gradients of
 * normal [B,H,W,3], texc [B,H,W,2], texd [B,H,W,4] (and, optionally, direct gradients of rast / rast_db) into
 * d_pos [B,V,4] and d_vnormal [B,V,3] (both ACCUMULATED, caller zero-fills) with one set of atomics per
 * triangle vertex instead of per pixel.  Any gradient pointer may be NULL (= zero).
 * uv_nograd_faces [F] uint8 or NULL: triangles whose texture coordinates are treated as constants (d_texc ignored;
 * d_texd still flows) -- the reference's `texc = torch.where(rast_mask, texc.detach(), texc)` (render_nvdiffrast.py:391-396). */
int vhap_gbuffer_bwd(const float* pos, const int32_t* tri, const float* vnormal, const float* uv,
                     const int32_t* tri_uv, const float* rast, const float* d_normal,
                     const float* d_texc, const float* d_texd, const float* d_rast,
                     const float* d_rast_db, const uint8_t* uv_nograd_faces, int B, int V, int F, int H,
                     int W, float* d_pos, float* d_vnormal, vhap_stream_t stream);

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
 * both), then vhap_texture_mip_fold() folds d_mips down into d_tex (stop_level = 0) or only down to level `stop_level`
 * (the caller gathers levels 1..stop_level itself, see vhap_tex_prep_bwd).  d_uv [B,H,W,2] and
 * d_uv_da [B,H,W,4] are overwritten (either may be NULL).
 * ------------------------------------------------------------------------------------------- */
int vhap_texture_num_levels(int Ht, int Wt);
size_t vhap_texture_mip_floats(int TB, int Ht, int Wt, int C);
int vhap_texture_mip_build(const float* tex, int TB, int Ht, int Wt, int C, float* mips,
                           vhap_stream_t stream);
/* levels first_level .. L of the pyramid from level first_level - 1, which is already in place (first_level = 1: the same as
 * vhap_texture_mip_build; 2: after vhap_tex_prep_mip1_fwd, which writes level 1 while it assembles the texture).  Four levels per
 * launch where the extents allow.  Bit-identical to vhap_texture_mip_build. */
int vhap_texture_mip_build_from(const float* tex, int TB, int Ht, int Wt, int C, float* mips, int first_level,
                                vhap_stream_t stream);
int vhap_texture_fwd(const float* tex, const float* mips, int TB, int Ht, int Wt, int C,
                     const float* uv, const float* uv_da, int B, int H, int W, float* out,
                     vhap_stream_t stream);
int vhap_texture_bwd(const float* tex, const float* mips, int TB, int Ht, int Wt, int C,
                     const float* uv, const float* uv_da, const float* d_out, int B, int H, int W,
                     float* d_tex, float* d_mips, float* d_uv, float* d_uv_da,
                     vhap_stream_t stream);
int vhap_texture_mip_fold(float* d_tex, float* d_mips, int TB, int Ht, int Wt, int C, int stop_level,
                          vhap_stream_t stream);
/* the same result as vhap_texture_mip_fold(stop_level = 0) for ONE texture (TB = 1) in one gathering pass: level 0 += sum_l 4^-l level_l
 * (the pyramid above level 0 is left as it is) */
int vhap_texture_mip_fold_gather(float* d_tex, const float* d_mips, int Ht, int Wt, int C, vhap_stream_t stream);
/* The d_tex / d_mips part of vhap_texture_bwd for ONE SHARED texture (TB == 1; the reference's expanded batch of copies,
 * tracker.py:234) through uv-space binning: the covered pixels (d_out != 0) of all frames are counting-sorted by the uv tile
 * they sample, one workgroup per tile accumulates them in LDS and every touched texel is flushed once -- an order of
 * magnitude fewer global atomics than the screen-tiled kernel when B frames sample the same texture.  Same result up to fp32
 * summation order (ACCUMULATES into d_tex / d_mips like vhap_texture_bwd).  Textures larger than 2048 texels per side return
 * VHAP_E_UNSUPPORTED (use vhap_texture_bwd).  `work`: vhap_texture_grad_binned_work_bytes(B,H,W) bytes, no state across calls. */
size_t vhap_texture_grad_binned_work_bytes(int B, int H, int W);
int vhap_texture_grad_binned(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out,
                             int B, int H, int W, float* d_tex, float* d_mips, void* work, size_t work_bytes,
                             vhap_stream_t stream);
/* same, with the uv tile of every pixel supplied by the producer of d_out (vhap_deferred_shade_bwd: tile_ids [B,H,W] uint16, 0xFFFF = none) */
int vhap_texture_grad_binned_ids(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out,
                                 const uint16_t* tile_ids, int B, int H, int W, float* d_tex, float* d_mips,
                                 void* work, size_t work_bytes, vhap_stream_t stream);
/* The sort and the accumulation as two calls: vhap_texbin_sort_ids counting-sorts the pixels by their uv tile (tile_ids from
 * vhap_raster_shade_fwd) into `work` without looking at any gradient -- it can run during the forward pass -- and
 * vhap_texture_grad_binned_sorted then only accumulates d_out through the sorted lists (one launch).
 *   keep [B,H,W] (may be NULL): pixels with keep == 0 (their colour was replaced by the disturbance) are left out of the lists
 *   gmax_bound (device scalar, may be NULL): an upper bound of |d_out| -- the fixed-point scale of the accumulation is then taken from it
 *   (one scale for all tiles); without it every tile first scans its list for its own maximum. */
int vhap_texbin_sort_ids(const uint16_t* tile_ids, const float* keep, int Ht, int Wt, int B, int H, int W, void* work,
                         size_t work_bytes, vhap_stream_t stream);
int vhap_texture_grad_binned_sorted(int Ht, int Wt, int C, const float* uv, const float* uv_da, const float* d_out,
                                    int B, int H, int W, float* d_tex, float* d_mips, const void* work,
                                    size_t work_bytes, const float* gmax_bound, vhap_stream_t stream);
/* same, for a `work` whose tile histogram was already filled in by vhap_deferred_shade_bwd(texbin_work = work): skips the count pass */
int vhap_texture_grad_binned_counted(int Ht, int Wt, int C, const float* uv, const float* uv_da,
                                     const float* d_out, int B, int H, int W, float* d_tex, float* d_mips,
                                     void* work, size_t work_bytes, vhap_stream_t stream);
/* ---------------------------------------------------------------------------------------------
 * Antialias: replaces dr.antialias(color, rast, pos, tri)  (:465).
 *   opp [F,3] int32 = static edge->opposite-vertex table (-1 = boundary edge) built once on the
 *   host from the fixed topology (vhap_amd.topology.build_opposite_table) -- it replaces the
 *   edge hash nvdiffrast rebuilds on every call.
 *   work: caller-owned int32 buffer of vhap_antialias_work_ints(B,H,W,F) ints (contents need not be
 *   initialised): the forward keeps its silhouette table and candidate list there and records the pixel
 *   pairs it blended; the backward replays them.
 * Backward: d_color [B,H,W,C] overwritten; d_pos [B,V,4] ACCUMULATED (caller zero-fills);
 *   pos_nograd_verts [V] uint8 or NULL: vertices whose position receives no silhouette gradient (the
 *   reference detaches them beforehand, render_nvdiffrast.py:462-464).
 * ------------------------------------------------------------------------------------------- */
size_t vhap_antialias_work_ints(int B, int H, int W, int F);
int vhap_antialias_fwd(const float* color, const float* rast, const float* pos,
                       const int32_t* tri, const int32_t* opp, int B, int H, int W, int C, int V,
                       int F, float* out, int32_t* work, vhap_stream_t stream);
int vhap_antialias_bwd(const float* color, const float* rast, const float* pos,
                       const int32_t* tri, const int32_t* opp, const float* d_out,
                       const int32_t* work, const uint8_t* pos_nograd_verts, int B, int H, int W,
                       int C, int V, int F, float* d_color, float* d_pos, int call_flags, vhap_stream_t stream);

/* In-place variant for the photometric step (C = 4): `color` [B,H,W,4] is antialiased IN PLACE (no copy of the image: the deltas
 * alpha (c1 - c0) are computed for all pairs first, from the original colours, then added); the pair records in `work`
 * (vhap_antialias_inplace_work_ints ints, sized for the worst case, only the used part is touched) keep the original colours.
 * vhap_antialias_photo_bwd: the backward for the photometric loss sum|gt - out| with upstream d_sum (tracker.py:430-439) -- the loss
 * gradient -sign(gt - out) d_sum is evaluated on the fly at the pixels of the pair list (pred_rgba = the antialiased image, gt_nchw
 * [B,3,H,W] image space), position gradients go to d_pos [B,V,4] (ACCUMULATED, pos_nograd_verts as above), and the colour part of the
 * antialias backward BESIDES the pass-through is ADDED into d_delta [B,H,W,4], an image the caller keeps at zero: hand it to
 * vhap_deferred_shade_bwd (d_delta) and restore the zeros afterwards with vhap_antialias_clear_delta (it visits the pair list only).
 * Replaces vhap_photo_bwd + vhap_antialias_bwd and their three dense gradient images. */
size_t vhap_antialias_inplace_work_ints(int B, int H, int W, int F);
int vhap_antialias_inplace_fwd(float* color, const float* rast, const float* pos, const int32_t* tri,
                               const int32_t* opp, int B, int H, int W, int V, int F, int32_t* work,
                               vhap_stream_t stream);
/* The same pass in pieces, silhouette + pairs + blend == vhap_antialias_inplace_fwd.  The silhouette flags are a property of the geometry
 * alone and the pixel-pair discovery reads only rast, NOT the colours: a step executor computes the flags beside the rasteriser and the
 * pairs right behind it, beside whatever still produces the colours (render_nvdiffrast.py:424-460 runs between the two); `blend` (edge
 * analysis, in-place update) needs the final colours. */
int vhap_antialias_inplace_silhouette(const float* pos, const int32_t* tri, const int32_t* opp, int B, int H, int W, int V, int F,
                                      int32_t* work, vhap_stream_t stream);
int vhap_antialias_inplace_pairs(const float* rast, int B, int H, int W, int F, int32_t* work, vhap_stream_t stream);
int vhap_antialias_inplace_blend(float* color, const float* rast, const float* pos, const int32_t* tri,
                                 const int32_t* opp, int B, int H, int W, int V, int F, int32_t* work,
                                 vhap_stream_t stream);
int vhap_antialias_photo_bwd(const float* pred_rgba, const float* gt_nchw, const float* d_sum,
                             const float* rast, const float* pos, const int32_t* tri, const int32_t* opp,
                             const int32_t* work, const uint8_t* pos_nograd_verts, int B, int H, int W, int V,
                             int F, float* d_delta, float* d_pos, vhap_stream_t stream);
int vhap_antialias_clear_delta(const int32_t* work, int B, int H, int W, float* d_delta, vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-pixel shading / compositing and the photometric sum (vhap_amd/csrc/pixel.hip).
 * Replace render_nvdiffrast.py:386 (safe_normalize), :402-421 (SH shade, rgb = albedo*diffuse, alpha,
 * background composite with the y-flip of :419) and tracker.py:430-439 / :547-550.
 *   normal_raw, albedo [B,H,W,3]; rast [B,H,W,4]; bg_image [B,3,H,W] in IMAGE space (row 0 = top) or
 *   NULL, in which case bg_color (HOST pointer to 3 floats) is used; lights [9,3]; sh_const [9].
 *   rgba [B,H,W,4] in renderer space.  stats (4 words, may be NULL): [0] = number of entries equal to the
 *   maximum, [1] = max(diffuse) as an order-preserving uint, [2] = sum over pixels of the unbiased
 *   variance of diffuse across RGB.
 * Backward: d_albedo / d_normal_raw [B,H,W,3] overwritten (either may be NULL); d_lights [9,3]
 * ACCUMULATED (may be NULL).  d_reg (device scalar, may be NULL) is the upstream gradient of
 * reg = relu(max(diffuse) - 1) + mean(var); it reaches d_lights only, like the reference's
 * shade(normal.detach()).  keep [B,H,W] or NULL: the `keep` mask of vhap_disturb_fwd -- d_rgba is multiplied by it on the
 * fly (= vhap_disturb_bwd folded into this pass).
 * ------------------------------------------------------------------------------------------- */
int vhap_shade_fwd(const float* normal_raw, const float* albedo, const float* rast,
                   const float* bg_image, const float* bg_color, const float* lights,
                   const float* sh_const, const int32_t* fid2cid, int nfid, int B, int H, int W,
                   float* rgba, float* stats, uint8_t* cid, int call_flags, vhap_stream_t stream);
int vhap_shade_bwd(const float* normal_raw, const float* albedo, const float* rast,
                   const float* lights, const float* sh_const, const float* d_rgba, const float* keep,
                   const float* d_reg, const float* stats, int B, int H, int W, float* d_albedo,
                   float* d_normal_raw, float* d_lights, vhap_stream_t stream);
/* out2[0] = sum |gt - pred_rgb|, out2[1] = #(pred_alpha > 0); pred [B,H,W,4] renderer space,
 * gt [B,3,H,W] image space.  Backward: d_pred.rgb = -sign(gt - pred) * d_sum[0] (device scalar); d_pred_copy (may be NULL)
 * receives the same values -- hand it to vhap_antialias_bwd as d_color with VHAP_CALL_AA_PASSTHROUGH_DONE set. */
int vhap_photo_fwd(const float* pred_rgba, const float* gt_nchw, int B, int H, int W, float* out2,
                   int call_flags, vhap_stream_t stream);
int vhap_photo_bwd(const float* pred_rgba, const float* gt_nchw, const float* d_sum, int B, int H,
                   int W, float* d_pred, float* d_pred_copy, vhap_stream_t stream);
/* vhap_photo_fwd + vhap_energy_finalize + vhap_energy_total_bound (world size 1) in ONE launch: the workgroup that finishes last assembles
 * log[VHAP_LOG_COUNT], d_sum and gmax_bound (either may be NULL) from the stage accumulators (any of frame_terms .. shade_stats may be
 * NULL = term absent), so that no single-thread launch sits between the forward and the backward pass.  out3 = (sum, count, ticket): three
 * words, zero on entry (cleared here unless VHAP_CALL_ACC_PREZEROED), the ticket word is left at zero; work: VHAP_PHOTO_WORK_FLOATS floats
 * of scratch (per-workgroup partial sums: the totals are summed in a fixed order, i.e. bit-reproducible).
 * aa_work + d_delta_unscaled (both or neither): the colour part of the in-place antialiasing's backward for this loss (what
 * vhap_antialias_photo_bwd adds into d_delta), per unit of d_sum, computed by the SAME launch ahead of the sum -- it needs the final image
 * and the pair list like the sum, and nothing of the sum's result; hand d_delta to vhap_deferred_shade_bwd with VHAP_CALL_DELTA_UNSCALED and
 * call vhap_antialias_photo_bwd with d_delta = NULL for the position part (off the critical path). */
#define VHAP_PHOTO_WORK_FLOATS 1024
int vhap_photo_fwd_total(const float* pred_rgba, const float* gt_nchw, int B, int H, int W, float* out3,
                         const float* frame_terms, const float* lmk_energy, const float* tex_terms,
                         const float* off_terms, const float* shade_stats, float w_landmark, float w_reg_diffuse,
                         float w_photo, float* log, float* d_sum, float* gmax_bound, float* work,
                         const int32_t* aa_work, float* d_delta_unscaled, int call_flags, vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * FLAME geometry (vhap_amd/csrc/flame.hip): replaces lbs.blend_shapes (vhap/model/lbs.py:218-239), the
 * pose-corrective matmul (:164-166) and the skinning (:182-193) of FlameHead.forward
 * (vhap/model/flame.py:595-634); world->clip (render_nvdiffrast.py:162-206); compute_v_normals (:297-316).
 *   coef  [Bp,Kp]  per-frame coefficients (shape|expr|pose_feature), rows padded to 16, zero-filled
 *   basis [3,K,Vp] K-major per component; basisT [3,Vp,Kp] vertex-major (backward); Vp % 64 == 0
 *   A [B,5,12] relative joint transforms (3x4 row-major); Kb = number of shape+expr rows
 * skin_bwd: d_coef [Bp,Kp] overwritten, d_A [B,5,12] / d_transl [B,3] ACCUMULATED (caller zero-fills),
 *   g_posed / g_shaped [B,V,3] scratch (g_shaped = gradient w.r.t. v_shaped incl. d_vshaped),
 *   partials: vhap_flame_bwd_partial_floats() floats (0 since ABI 1: may be NULL).
 * ------------------------------------------------------------------------------------------- */
int vhap_flame_skin_fwd(const float* coef, const float* basis, const float* A,
                        const float* lbs_weights, const float* v_template, const float* offset,
                        const float* transl, int B, int V, int Vp, int K, int Kb, int Kp,
                        float* verts, float* v_shaped, float* v_posed, int call_flags, vhap_stream_t stream);
/* same, fused with vhap_transform_fwd: also writes clip [B,V,4] = [verts;1] @ mvp^T (mvp [B,4,4]) -- bit-identical to the two calls */
int vhap_flame_skin_clip_fwd(const float* coef, const float* basis, const float* A,
                             const float* lbs_weights, const float* v_template, const float* offset,
                             const float* transl, const float* mvp, int B, int V, int Vp, int K, int Kb,
                             int Kp, float* verts, float* v_shaped, float* v_posed, float* clip,
                             int call_flags, vhap_stream_t stream);
size_t vhap_flame_bwd_partial_floats(int B, int Vp, int Kp);
int vhap_flame_skin_bwd(const float* d_verts, const float* d_vshaped, const float* v_posed,
                        const float* A, const float* lbs_weights, const float* basisT, int B, int V,
                        int Vp, int Kb, int Kp, float* g_posed, float* g_shaped, float* partials,
                        float* d_coef, float* d_A, float* d_transl, int call_flags,
                        vhap_stream_t stream);
/* The vertex stage of the backward in one call (three launches instead of seven): vhap_vnormal_bwd_saved + vhap_transform_bwd +
 * vhap_flame_skin_bwd + vhap_sum_frames.  The gradient w.r.t. the world-space vertices is assembled in registers from d_verts_in [B,V,3]
 * (may be NULL: e.g. the landmark part), the vertex-normal backward of d_vn and M^T d_clip, and chained through the skinning backward
 * without being written back.  d_A / d_transl / d_mvp [B,16] (may be NULL) / d_offset [V,3] (may be NULL: sum over frames of g_shaped)
 * are ACCUMULATED; g_posed / g_shaped / d_coef as vhap_flame_skin_bwd (VHAP_CALL_ACC_PREZEROED applies to d_coef); scratch [B,V,3] (kept in
 * the signature, no longer written: since ABI 8 the normalisation's backward is re-derived per gathered vertex inside the vertex kernel, two
 * launches instead of three). */
int vhap_verts_bwd_fused(const float* verts, const int32_t* tri, const int32_t* vc_ptr, const int32_t* vc_idx,
                         const float* vn, const float* inv_len, const float* d_vn, const float* mvp,
                         const float* d_clip, const float* d_verts_in, const float* v_posed, const float* A,
                         const float* lbs_weights, const float* basisT, int B, int V, int Vp, int Kb, int Kp,
                         float* scratch, float* g_posed, float* g_shaped, float* d_coef, float* d_A,
                         float* d_transl, float* d_mvp, float* d_offset, int call_flags, vhap_stream_t stream);
/* clip [B,V,4] = [verts;1] @ M^T, M [B,4,4]; backward: d_verts (= or += when accumulate), d_M ACCUMULATED */
int vhap_transform_fwd(const float* verts, const float* M, int B, int V, float* clip,
                       vhap_stream_t stream);
int vhap_transform_bwd(const float* verts, const float* M, const float* d_clip, int B, int V,
                       int accumulate, float* d_verts, float* d_M, vhap_stream_t stream);
/* vertex normals over the static vertex->corner CSR (vc_ptr [V+1], vc_idx [3F]); scratch [B,V,3] */
int vhap_vnormal_fwd(const float* verts, const int32_t* tri, const int32_t* vc_ptr,
                     const int32_t* vc_idx, int B, int V, float* vn, vhap_stream_t stream);
int vhap_vnormal_bwd(const float* verts, const int32_t* tri, const int32_t* vc_ptr,
                     const int32_t* vc_idx, const float* d_vn, int B, int V, int accumulate,
                     float* scratch, float* d_verts, vhap_stream_t stream);
/* same pair with the normalisation saved by the forward (inv_len [B,V] = 1 / |raw normal|, 0 where the constant fallback normal was taken):
 * the backward's first pass then needs no second gather over the incident faces */
int vhap_vnormal_fwd_saved(const float* verts, const int32_t* tri, const int32_t* vc_ptr,
                           const int32_t* vc_idx, int B, int V, float* vn, float* inv_len, vhap_stream_t stream);
int vhap_vnormal_bwd_saved(const float* verts, const int32_t* tri, const int32_t* vc_ptr,
                           const int32_t* vc_idx, const float* vn, const float* inv_len, const float* d_vn,
                           int B, int V, int accumulate, float* scratch, float* d_verts, vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Colour disturbance (vhap_amd/csrc/disturb.hip): replaces render_nvdiffrast.py:424-460.
 *   rgba [B,H,W,4] composited image (background pixels hold the background colour), rast [B,H,W,4],
 *   fid2cid [nfid] int32 (index = triangle id + 1, 0 = background), ncl clusters (<= 16),
 *   w_fg / w_bg [B,H,W] int32 Bernoulli masks, idx [B*H*W] int64 non-negative random integers.
 *   out [B,H,W,4]; keep [B,H,W] = 1 where the pixel kept its own colour (backward mask).
 *   workspace: vhap_disturb_workspace_ints() int32, 16-byte aligned (per-block histograms + the per-cluster colour POOLS: the
 *   colours of every cluster's pixels, batch-wide, in row-major pixel order, 16 B per pixel -- a disturbed pixel gathers one colour
 *   from its cluster's pool).  out == rgba is allowed (the pools are copies): undisturbed pixels are then not touched at all.
 * ------------------------------------------------------------------------------------------- */
size_t vhap_disturb_workspace_ints(int B, int H, int W);
int vhap_disturb_fwd(const float* rgba, const float* rast, const int32_t* fid2cid, int nfid, int ncl,
                     const int32_t* w_fg, const int32_t* w_bg, const int64_t* idx, int B, int H,
                     int W, int32_t* workspace, float* out, float* keep, vhap_stream_t stream);
/* same, with the random numbers drawn inside the kernel (counter-based hash): Bernoulli(rate_fg / rate_bg) masks and one
 * 32-bit index draw per pixel.  rng_state [1] uint32 on the device is the stream counter; every call advances it, so replays of a
 * captured graph draw fresh numbers. */
int vhap_disturb_fwd_rng(const float* rgba, const float* rast, const int32_t* fid2cid, int nfid, int ncl,
                         float rate_fg, float rate_bg, uint32_t* rng_state, int B, int H, int W,
                         int32_t* workspace, float* out, float* keep, vhap_stream_t stream);
/* the step executor's form: IN PLACE on rgba, clusters from the one-byte image `cid`; random numbers drawn in-kernel (rng_state != NULL,
 * w_fg / w_bg / idx ignored) or INJECTED (rng_state == NULL: w_fg / w_bg / idx as in vhap_disturb_fwd -- the same kernels, so that a
 * step replayed with the oracle's random numbers exercises the code that ships) */
int vhap_disturb_inplace(float* rgba, const uint8_t* cid, int ncl, const int32_t* w_fg, const int32_t* w_bg, const int64_t* idx,
                         float rate_fg, float rate_bg, uint32_t* rng_state, int B, int H, int W, int32_t* workspace, float* keep,
                         vhap_stream_t stream);
/* vhap_disturb_inplace that also writes the list of COVERED pixels (cluster != 0) in pixel order -- covered_list [B*H*W] uint32, of which the first
 * B*H*W - *n_background are written -- and the number of background pixels: a by-product of the counting sort (one 4-byte store per covered pixel),
 * for passes that have nothing to do on the background (vhap_deferred_shade_bwd_list). */
int vhap_disturb_inplace_list(float* rgba, const uint8_t* cid, int ncl, const int32_t* w_fg, const int32_t* w_bg, const int64_t* idx,
                              float rate_fg, float rate_bg, uint32_t* rng_state, int B, int H, int W, int32_t* workspace, float* keep,
                              uint32_t* covered_list, int32_t* n_background, vhap_stream_t stream);
int vhap_disturb_bwd(const float* d_out, const float* keep, int B, int H, int W, float* d_rgba,
                     vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Per-frame parameter stage (vhap_amd/csrc/frame.hip): replaces the gathers by timestep, lbs.batch_rodrigues
 * (vhap/model/lbs.py:25-57), batch_rigid_transform (:254-301), the joint regression and pose feature of
 * lbs.lbs (:148-176) as called from FlameHead.forward (vhap/model/flame.py:571-634), and the parameter
 * energies compute_pose_smooth_energy / compute_joint_smooth_energy / compute_expr_smooth_energy /
 * compute_joint_L2_energy / reg_expr / reg_shape (vhap/model/tracker.py:486-500, 616-680).
 *   timesteps [B] int64 rows of the [N,*] parameter arrays; previous frame = max(t - 1, 0), detached
 *   JT [J,3] = J_regressor v_template; JS [3J, NS+NE] = J_regressor shapedirs; jreg_idx [M] / jreg_w [M,J]: the M vertices
 *   with a non-zero J_regressor column and their weights (only used with static_offset)
 *   weights[12]: VHAP_FW_* (0 disables a term); parents[J]
 *   coef [Bp,Kp] (rows >= B zero-filled), A [B,J,12], transl [B,3], Jrest [B,J,3] (saved for the backward),
 *   terms[6] = smooth_pose, reg_joint, smooth_joint, reg_expr, smooth_expr, reg_shape (weighted)
 * bwd: gradients are ACCUMULATED into the full-size arrays g_* (caller zero-fills; any may be NULL);
 *   d_coef / d_A / d_transl / d_terms may be NULL (= zero).
 * ------------------------------------------------------------------------------------------- */
enum {
    VHAP_FW_SMOOTH_TRANS = 0, VHAP_FW_SMOOTH_ROT = 1, VHAP_FW_SMOOTH_NECK = 2, VHAP_FW_SMOOTH_JAW = 3,
    VHAP_FW_SMOOTH_EYES = 4, VHAP_FW_SMOOTH_EXPR = 5, VHAP_FW_REG_NECK = 6, VHAP_FW_REG_JAW = 7,
    VHAP_FW_REG_EYES = 8, VHAP_FW_REG_EXPR = 9, VHAP_FW_REG_SHAPE = 10
};
int vhap_frame_prep_fwd(const int64_t* timesteps, const float* shape, const float* expr,
                        const float* rotation, const float* translation, const float* neck,
                        const float* jaw, const float* eyes, const float* JT, const float* JS,
                        const int32_t* jreg_idx, const float* jreg_w, int jreg_n,
                        const float* static_offset, const int32_t* parents,
                        const float* weights, int B, int Bp, int N, int NS, int NE, int J, int Kp, int V,
                        float* coef, float* A, float* transl, float* Jrest, float* terms,
                        int call_flags, vhap_stream_t stream);
int vhap_frame_prep_bwd(const int64_t* timesteps, const float* shape, const float* expr,
                        const float* rotation, const float* translation, const float* neck,
                        const float* jaw, const float* eyes, const float* JS, const int32_t* jreg_idx,
                        const float* jreg_w, int jreg_n, const float* static_offset,
                        const int32_t* parents, const float* weights,
                        const float* Jrest, const float* d_coef, const float* d_A, const float* d_transl,
                        const float* d_terms, int B, int Bp, int N, int NS, int NE, int J, int Kp, int V,
                        float* g_shape, float* g_expr, float* g_rotation, float* g_translation,
                        float* g_neck, float* g_jaw, float* g_eyes, float* g_offset,
                        int call_flags, vhap_stream_t stream);
/* mvp [B,4,4] = P(K) [RT; 0 0 0 1] (render_nvdiffrast.py:102-160); K [B,4] = fx, fy, cx, cy (or one row shared
 * when K_batched == 0), RT [B,3,4] (or one shared); d_K [B,4] overwritten */
int vhap_camera_fwd(const float* K, const float* RT, int B, int K_batched, int RT_batched, int H, int W,
                    float near_plane, float far_plane, float* mvp, vhap_stream_t stream);
int vhap_camera_bwd(const float* RT, const float* d_mvp, int B, int RT_batched, int H, int W, float* d_K,
                    vhap_stream_t stream);
/* monocular case, vhap_camera_bwd and the sum over the batch in one launch: d_focal_accum[0] += scale * sum_b (d_K[b].fx + d_K[b].fy), summed in
 * frame order (K = (f, f, cx, cy), f = focal_length * scale; tracker.py:148-157) */
int vhap_camera_focal_bwd(const float* RT, const float* d_mvp, int B, int RT_batched, int H, int W, float scale,
                          float* d_focal_accum, vhap_stream_t stream);
/* the uncalibrated camera (tracker.py:148-157): K = (f, f, cx, cy) with f = focal_length[0] * focal_scale (device scalar x max(H, W)) */
int vhap_camera_focal_fwd(const float* focal_length, float focal_scale, float cx, float cy, const float* RT, int B,
                          int RT_batched, int H, int W, float near_plane, float far_plane, float* mvp,
                          vhap_stream_t stream);
/* vhap_frame_prep_fwd + vhap_camera_focal_fwd / vhap_frame_prep_bwd + vhap_camera_focal_bwd in ONE launch each (the camera is one more
 * workgroup of the per-frame kernel: FlameHead.forward and the camera set-up of the reference, tracker.py:141-157, are independent of each
 * other, and as launches of their own the two camera kernels sat on the step's critical path -- 7 us at its head, an event record and a
 * wait at its tail).  Same arguments, same results (bit for bit) as the two calls. */
int vhap_frame_prep_fwd_camera(const int64_t* timesteps, const float* shape, const float* expr,
                               const float* rotation, const float* translation, const float* neck,
                               const float* jaw, const float* eyes, const float* JT, const float* JS,
                               const int32_t* jreg_idx, const float* jreg_w, int jreg_n,
                               const float* static_offset, const int32_t* parents,
                               const float* weights, int B, int Bp, int N, int NS, int NE, int J, int Kp, int V,
                               float* coef, float* A, float* transl, float* Jrest, float* terms, int call_flags,
                               const float* focal_length, float focal_scale, float cx, float cy, const float* RT,
                               int RT_batched, int H, int W, float near_plane, float far_plane, float* mvp,
                               vhap_stream_t stream);
int vhap_frame_prep_bwd_camera(const int64_t* timesteps, const float* shape, const float* expr,
                               const float* rotation, const float* translation, const float* neck,
                               const float* jaw, const float* eyes, const float* JS, const int32_t* jreg_idx,
                               const float* jreg_w, int jreg_n, const float* static_offset,
                               const int32_t* parents, const float* weights,
                               const float* Jrest, const float* d_coef, const float* d_A, const float* d_transl,
                               const float* d_terms, int B, int Bp, int N, int NS, int NE, int J, int Kp, int V,
                               float* g_shape, float* g_expr, float* g_rotation, float* g_translation,
                               float* g_neck, float* g_jaw, float* g_eyes, float* g_offset, int call_flags,
                               const float* RT, const float* d_mvp, int RT_batched, int H, int W, float focal_scale,
                               float* d_focal_accum, vhap_stream_t stream);
/* Landmark energy (lbs.vertices2landmarks, vhap/model/lbs.py:60-98 + compute_lmk_energy, tracker.py:347-389):
 * mean over B x [l0,l1) of (|du| + |dv|) conf, with conf x boost for landmarks [boost0,boost1).
 *   lmk_vidx [L,3] int32 vertex ids of each landmark's triangle, lmk_bary [L,3], lmk2d [B,L2,3] = u, v, confidence (pixels)
 *   lmk3d [B,L,3] optional output; energy: device scalar (overwritten)
 * bwd: d_verts [B,V,3] ACCUMULATED, d_mvp [B,16] overwritten (may be NULL). */
int vhap_landmark_fwd(const float* verts, const int32_t* lmk_vidx, const float* lmk_bary, const float* mvp,
                      const float* lmk2d, int B, int V, int L, int L2, int l0, int l1, int boost0, int boost1,
                      float boost, int H, int W, float* lmk3d, float* energy, int call_flags,
                      vhap_stream_t stream);
int vhap_landmark_bwd(const float* verts, const int32_t* lmk_vidx, const float* lmk_bary, const float* mvp,
                      const float* lmk2d, const float* d_energy, int B, int V, int L, int L2, int l0, int l1,
                      int boost0, int boost1, float boost, int H, int W, float* d_verts, float* d_mvp,
                      vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Offset / texture regularisers and Adam (vhap_amd/csrc/reg.hip).
 * offset_reg: terms[3] = s_lap sum_v w_lap[v] |(L o)_v|^2, s_abs sum_v w_abs[v] |o_v|_1, s_rigid sum_regions sum_xyz var
 *   (tracker.py:552-600, 682-690; L in CSR: lap_ptr [V+1], lap_col, lap_val; regions in CSR: region_ptr, region_idx;
 *   w_lap / w_abs [V] or NULL).  bwd: d_offset [V,3] ACCUMULATED.
 * tex_prep: albedo_hwc [T,T,3] = painted [3,T,T] + extra [3,T,T] (either may be NULL); terms[2] = s_tv * TV(albedo),
 *   s_res * sum(extra^2 * res_mask) (tracker.py:247-258, 518-541; res_mask [T,T] uint8 or NULL).  bwd: d_extra [3,T,T]
 *   overwritten (d_albedo_hwc may be NULL); d_mips_hwc: the gradient pyramid (layout of vhap_texture_mip_build) folded down to
 *   level n_gather by vhap_texture_mip_fold(stop_level = n_gather); levels 1..n_gather are gathered here as
 *   4^-l * d_level_l[y >> l][x >> l] (the last steps of the fold fused into this pass; NULL / 0 = none).
 * adam_step: torch.optim.Adam update (tracker.py:159-211) of up to VHAP_ADAM_MAX_TENSORS tensors in one launch; the
 *   pointer tables are HOST arrays of device pointers; lr_device[lr_index[k]] and step_device[0] live on the device
 *   (graph replays see their current values); step_device is incremented.
 * ------------------------------------------------------------------------------------------- */
#define VHAP_ADAM_MAX_TENSORS 16
int vhap_offset_reg_fwd(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col,
                        const float* lap_val, const float* w_lap, const float* w_abs,
                        const int32_t* region_ptr, const int32_t* region_idx, int V, int n_regions,
                        float s_lap, float s_abs, float s_rigid, float* terms, int call_flags,
                        vhap_stream_t stream);
int vhap_offset_reg_bwd(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col,
                        const float* lap_val, const float* w_lap, const float* w_abs,
                        const int32_t* region_ptr, const int32_t* region_idx, int V, int n_regions,
                        float s_lap, float s_abs, float s_rigid, const float* d_terms, float* d_offset,
                        vhap_stream_t stream);
/* `use_dynamic_offset` (base.py:69; tracker.py:213-235, 552-600): vhap_offset_combine: off_b [B,V,3] = (static_offset [V,3] | NULL) +
 * dynamic_offset [N,V,3][timesteps[b]]; vhap_offset_reg_fwd_batch / _bwd_batch: the three terms of vhap_offset_reg_fwd / _bwd for all B
 * per-frame offsets in one launch (terms accumulated over the frames: the caller's scales carry the 1 / B; d_offset [B,V,3] ACCUMULATED). */
int vhap_offset_combine(const float* static_offset, const float* dynamic_offset, const int64_t* timesteps, int B, int N, int V, float* out,
                        vhap_stream_t stream);
int vhap_offset_reg_fwd_batch(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col, const float* lap_val, const float* w_lap,
                              const float* w_abs, const int32_t* region_ptr, const int32_t* region_idx, int B, int V, int n_regions, float s_lap,
                              float s_abs, float s_rigid, float* terms, int call_flags, vhap_stream_t stream);
int vhap_offset_reg_bwd_batch(const float* offset, const int32_t* lap_ptr, const int32_t* lap_col, const float* lap_val, const float* w_lap,
                              const float* w_abs, const int32_t* region_ptr, const int32_t* region_idx, int B, int V, int n_regions, float s_lap,
                              float s_abs, float s_rigid, const float* d_terms, float* d_offset, vhap_stream_t stream);
int vhap_tex_prep_fwd(const float* painted, const float* extra, const uint8_t* res_mask, int T, float s_tv,
                      float s_res, float* albedo_hwc, float* terms, int call_flags, vhap_stream_t stream);
/* vhap_tex_prep_fwd that also writes level 1 of the pyramid (mips_hwc: the buffer of vhap_texture_mip_build, whose first (T/2)^2 x 3
 * floats are level 1); T must be even.  Follow with vhap_texture_mip_build_from(first_level = 2). */
int vhap_tex_prep_mip1_fwd(const float* painted, const float* extra, const uint8_t* res_mask, int T, float s_tv,
                           float s_res, float* albedo_hwc, float* mips_hwc, float* terms, int call_flags,
                           vhap_stream_t stream);
int vhap_tex_prep_bwd(const float* albedo_hwc, const float* extra, const uint8_t* res_mask,
                      const float* d_albedo_hwc, const float* d_mips_hwc, int n_gather, const float* d_terms,
                      int T, float s_tv, float s_res, float* d_extra, vhap_stream_t stream);
/* vhap_tex_prep_bwd fused with the torch.optim.Adam update of `extra` (one tensor of a vhap_adam_step call issued in pieces: it reads the
 * step counter but does not advance it -- VHAP_CALL_ADAM_KEEP_STEP semantics, or, with VHAP_CALL_ADAM_STEP_ADVANCED in call_flags, the
 * counter was advanced for this step already; lr_device points at THIS tensor's learning rate): the gradient d_extra is still written,
 * but never read back, and the separate update pass over the texture disappears.  With n_gather = every level of the pyramid
 * (vhap_texture_num_levels) the fold cascade (vhap_texture_mip_fold) disappears as well: ONE pass over the texture turns the gradient
 * pyramid the sampling backward accumulated into the updated parameter. */
int vhap_tex_prep_bwd_adam(const float* albedo_hwc, float* extra, const uint8_t* res_mask,
                           const float* d_albedo_hwc, const float* d_mips_hwc, int n_gather, const float* d_terms,
                           int T, float s_tv, float s_res, float* d_extra, float* exp_avg, float* exp_avg_sq,
                           const float* lr_device, const int32_t* step_device, float beta1, float beta2, float eps,
                           int call_flags, vhap_stream_t stream);
/* vhap_tex_prep_bwd_adam on the ROW STRIP [row0, row0 + nrows) of the texture (multiples of 16): frame sharding over N GPUs lets rank r
 * finish and update rows [r T / N, (r + 1) T / N) only -- d_albedo_strip = the strip's [nrows, T, 3] slice of the level-0 gradient with the
 * whole pyramid folded into it (vhap_texture_mip_fold(stop_level = 0)), e.g. the output of a reduce-scatter; albedo_hwc, extra, res_mask,
 * d_extra, exp_avg, exp_avg_sq are the FULL arrays (rows outside the strip are not touched).  New in this build. */
/* vhap_tex_prep_bwd / vhap_tex_prep_bwd_adam with one more output, d_base [3,T,T] (may be NULL): d(base texture) = d(albedo) WITHOUT the
 * residual's own regulariser (reg_tex_res_clusters acts on tex_extra only, tracker.py:538-541) -- the input of vhap_tex_pca_bwd. */
int vhap_tex_prep_bwd_base(const float* albedo_hwc, const float* extra, const uint8_t* res_mask, const float* d_albedo_hwc,
                           const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res, float* d_extra,
                           float* d_base, vhap_stream_t stream);
int vhap_tex_prep_bwd_adam_base(const float* albedo_hwc, float* extra, const uint8_t* res_mask, const float* d_albedo_hwc,
                                const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res, float* d_extra,
                                float* exp_avg, float* exp_avg_sq, const float* lr_device, const int32_t* step_device, float beta1, float beta2,
                                float eps, float* d_base, int call_flags, vhap_stream_t stream);
int vhap_tex_prep_bwd_adam_rows(const float* albedo_hwc, float* extra, const uint8_t* res_mask, const float* d_albedo_strip,
                                const float* d_terms, int T, int row0, int nrows, float s_tv, float s_res, float* d_extra, float* exp_avg,
                                float* exp_avg_sq, const float* lr_device, const int32_t* step_device, float beta1, float beta2, float eps,
                                int call_flags, vhap_stream_t stream);
/* THE CARRIED TEXTURE (ABI 10).  tracker.py:237-258 re-assembles albedo = painted + tex_extra at the head of every step and
 * render_nvdiffrast.py:398-399 rebuilds the mip pyramid per call; inside a loop of steps the only writer of tex_extra is the step's own
 * Adam update, so the pass that applies it (vhap_tex_prep_bwd_adam) can hand the NEXT step its texture:
 *   vhap_tex_finish_carry = vhap_tex_prep_bwd_adam (whole pyramid gathered) that also (i) rewrites albedo_hwc IN PLACE with
 *     painted + updated tex_extra (the sum vhap_tex_prep_fwd forms -> the same bits), (ii) writes levels 1 AND 2 of the pyramid into mips_hwc
 *     (follow with vhap_texture_mip_build_from(first_level = 3) before the texture is sampled; after vhap_tex_carry_prime, which writes
 *     level 1 only: first_level = 2), (iii) ACCUMULATES the weighted TV /
 *     residual energies (reg_tex_tv, reg_tex_res_clusters: tracker.py:518-541) of that NEXT texture into terms[0..1] -- all but the TV
 *     pairs that straddle two ownership tiles, which vhap_tex_carry_border adds from the halo copies (behind this pass, ahead of the
 *     next vhap_adam_advance of step_device; same call_flags) -- hand them to the next step's energy assembly as its tex_terms with
 *     VHAP_CALL_TEX_TERMS_CONSUME (vhap_photo_fwd_total reads and clears them).
 *     d_extra may be NULL: the gradient is consumed by the update and not written.  Needs T % 64 == 0.
 *   halo: vhap_tex_carry_halo_floats(T) floats owned by the caller -- border rows / columns of the pass's 16-row x 64-column ownership
 *     tiles, twice (the TV stencil of a texel on a tile border reads neighbours another workgroup may already have rewritten: it reads
 *     the copy of parity (Adam step & 1) and writes the other one).
 *   vhap_tex_carry_prime: albedo_hwc, level 1, both halo parities and terms[0..1] (overwritten; like vhap_tex_finish_carry without the
 *     tile-straddling TV pairs: follow with vhap_tex_carry_border) from painted + extra, from scratch -- before the first step of a loop
 *     and whenever anything else has written tex_extra / painted since the last vhap_tex_finish_carry. */
size_t vhap_tex_carry_halo_floats(int T);
int vhap_tex_carry_prime(const float* painted, const float* extra, const uint8_t* res_mask, int T, float s_tv, float s_res, float* albedo_hwc,
                         float* mips_hwc, float* halo, float* terms, vhap_stream_t stream);
int vhap_tex_carry_border(int T, float s_tv, const int32_t* step_device, const float* halo, float* terms, int call_flags, vhap_stream_t stream);
int vhap_tex_finish_carry(float* albedo_hwc, float* extra, const uint8_t* res_mask, const float* painted, const float* d_albedo_hwc,
                          const float* d_mips_hwc, int n_gather, const float* d_terms, int T, float s_tv, float s_res, float* d_extra,
                          float* exp_avg, float* exp_avg_sq, const float* lr_device, const int32_t* step_device, float beta1, float beta2,
                          float eps, float* mips_hwc, float* halo, float* terms, int call_flags, vhap_stream_t stream);
/* FLAME PCA texture model (flame.py:665-688 FlameTexPCA, tracker.py:241-244, 519-521; the tex_painted = False configuration):
 *   fwd: src [S,S,3] = mean + basis [S*S*3, n] . code [n]  (0..255, B G R);  base [3,T,T] = clamp(nearest_resize(src)[R,G,B] / 255, 0, 1) -- takes
 *        the place of the painted texture in vhap_tex_prep_fwd; term_accum (may be NULL) += s_reg * sum(code^2)   (s_reg = w.reg_tex_pca / n)
 *   bwd: d_code [n] ACCUMULATED from d_base [3,T,T] (= d albedo without the residual regulariser: the optional d_base output of
 *        vhap_tex_prep_bwd) through the clamp and the resize, + the regulariser's gradient (d_term[0], NULL = 1); g_work: [S*S*3] floats. */
int vhap_tex_pca_fwd(const float* mean, const float* basis, const float* code, int n, int S, int T, float s_reg, float* src, float* base,
                     float* term_accum, vhap_stream_t stream);
int vhap_tex_pca_bwd(const float* basis, const float* src, const float* d_base, const float* code, int n, int S, int T, float s_reg,
                     const float* d_term, float* g_work, float* d_code, vhap_stream_t stream);
int vhap_adam_step(int n_tensors, float* const* params, const float* const* grads, float* const* exp_avg,
                   float* const* exp_avg_sq, const int64_t* numel, const int32_t* lr_index,
                   const float* lr_device, int32_t* step_device, float beta1, float beta2, float eps,
                   int call_flags, vhap_stream_t stream);
/* step_device[0] += 1 (one tiny launch): advance the counter at the HEAD of a step, then issue the step's vhap_adam_step calls with
 * VHAP_CALL_ADAM_STEP_ADVANCED. */
int vhap_adam_advance(int32_t* step_device, vhap_stream_t stream);
/* dst_device[0..n) = values_host[0..n), n <= 16, as ONE tiny launch whose kernel arguments carry the values: the learning-rate table of
 * vhap_adam_step after a scheduler step (torch.optim.lr_scheduler.ExponentialLR, tracker.py:1399-1413) without a host-blocking copy. */
int vhap_set_floats(float* dst_device, const float* values_host, int n, vhap_stream_t stream);

/* ---------------------------------------------------------------------------------------------
 * Step glue (vhap_amd/csrc/step.hip, misc.hip) for an executor that chains the stages itself instead of torch autograd
 * (vhap_amd/step.py): the assembly of the total energy (tracker.py:692-750), the batch-global photometric normaliser
 * (tracker.py:430-439), the static-offset gradient summed over frames and the focal-length gradient (tracker.py:141-157).
 *
 * vhap_energy_finalize: log[VHAP_LOG_COUNT] <- the weighted terms (any input may be NULL = term absent); log[VHAP_LOG_REST] = sum
 *   of everything but the photometric term.  frame_terms [6], tex_terms [3] (TV, residual, PCA code regulariser), off_terms [4] = (Laplacian, L1, rigidity as
 *   vhap_offset_reg_fwd writes them, then reg_offset_dynamic as vhap_offset_dynamic_reg adds it; 0 without a dynamic offset).
 * vhap_energy_total: inv_n = world_size / (3 n_global); log[PHOTO] = w_photo * photo2[0] * inv_n; log[TOTAL]; d_sum[0] = w_photo * inv_n
 *   (the upstream gradient handed to vhap_photo_bwd).
 * ------------------------------------------------------------------------------------------- */
enum {
    VHAP_LOG_LMK = 0, VHAP_LOG_PHOTO = 1, VHAP_LOG_SMOOTH_POSE = 2, VHAP_LOG_REG_JOINT = 3, VHAP_LOG_SMOOTH_JOINT = 4,
    VHAP_LOG_REG_EXPR = 5, VHAP_LOG_SMOOTH_EXPR = 6, VHAP_LOG_REG_SHAPE = 7, VHAP_LOG_TEX_TV = 8, VHAP_LOG_TEX_RES = 9,
    VHAP_LOG_REG_DIFFUSE = 10, VHAP_LOG_OFF_LAP = 11, VHAP_LOG_OFF_ABS = 12, VHAP_LOG_OFF_RIGID = 13, VHAP_LOG_REST = 14,
    VHAP_LOG_TOTAL = 15, VHAP_LOG_OFF_DYNAMIC = 16 /* reg_offset_dynamic: part of REST and TOTAL */,
    VHAP_LOG_TEX_PCA = 17 /* reg_tex_pca (tex_terms[2]): part of REST and TOTAL */, VHAP_LOG_COUNT = 18
};
int vhap_energy_finalize(const float* frame_terms, const float* lmk_energy, const float* tex_terms,
                         const float* off_terms, const float* shade_stats, float w_landmark,
                         float w_reg_diffuse, int B, int H, int W, float* log, vhap_stream_t stream);
int vhap_energy_total(float* log, const float* photo2, const float* n_global, float w_photo, int world_size,
                      float* d_sum, vhap_stream_t stream);
/* same; additionally gmax_bound[0] = 2 |d_sum| max(1, max(diffuse)) (shade_stats of vhap_shade_fwd / vhap_raster_shade_fwd, may be NULL
 * -> 8 |d_sum|): an upper bound of the per-pixel albedo gradient |d_rgb * diffuse| including the antialiasing's colour part, for
 * vhap_texture_grad_binned_sorted */
int vhap_energy_total_bound(float* log, const float* photo2, const float* n_global, float w_photo, int world_size,
                            float* d_sum, const float* shade_stats, float* gmax_bound, vhap_stream_t stream);
/* out_accum[i] += sum_b x[b][i]  (x [B,n]) */
int vhap_sum_frames(const float* x, int B, int n, float* out_accum, vhap_stream_t stream);
/* ---- per-frame vertex offsets (`use_dynamic_offset`, vhap/config/base.py:69; tracker.py:213-235, 552-600) ----------------------------
 * vhap_offset_dynamic_reg: reg_offset_dynamic = scale * sum_{b,v,c} (dyn[ts[b]] - dyn[prev(ts[b])])^2, prev(t) = max(t - 1, 0) (the caller
 *   folds weight / (B V 3) into `scale`); energy_accum[0] += the term (or NULL); d_dyn [N,V,3] += its gradient to BOTH rows (or NULL),
 *   scaled by d_term[0] (or 1 if NULL).
 * vhap_offset_grad_finish: the gradient w.r.t. the per-frame combined offset, g_a + g_b (each [B,V,3], g_b may be NULL), goes to
 *   d_static [V,3] += its sum over the frames (or NULL) and d_dyn[ts[b]] [N,V,3] += its row b (or NULL). */
int vhap_offset_dynamic_reg(const float* dyn, const int64_t* timesteps, int B, int N, int V, float scale, const float* d_term,
                            float* energy_accum, float* d_dyn, vhap_stream_t stream);
int vhap_offset_grad_finish(const float* g_a, const float* g_b, const int64_t* timesteps, int B, int N, int V, float* d_static,
                            float* d_dyn, vhap_stream_t stream);
/* ---- frame ingest --------------------------------------------------------------------------------------
 * Replaces, per batch and on the device, the reference's per-image host transforms video_dataset.py:253-259 (apply_transforms):
 * :302-323 apply_background_color (fp64 compositing over 'white' / 'black', truncating uint8 cast) and :261-268 apply_to_tensor
 * (HWC uint8 -> CHW fp32 / 255), bit for bit.  The sequence stays resident as uint8:
 *   rgb_u8   [N,H,W,3] uint8      alpha_u8 [N,H,W] uint8 or NULL (required when bg_mode != VHAP_BG_NONE or alpha_out != NULL)
 *   index    [B] int64 frame numbers (negative counts from the end) or NULL (= 0..B-1)
 *   rgb_out  [B,3,H,W] fp32       alpha_out [B,1,H,W] fp32 or NULL
 *   bad_index: optional device int, OR-ed with 1 when an index is out of range (that frame is written as zeros).
 * Other background colours are VHAP_E_BADDIM (the reference raises NotImplementedError). */
enum { VHAP_BG_NONE = 0, VHAP_BG_WHITE = 1, VHAP_BG_BLACK = 2 };
int vhap_frame_ingest(const unsigned char* rgb_u8, const unsigned char* alpha_u8, const long long* index, int N, int B, int H, int W,
                      int bg_mode, float* rgb_out, float* alpha_out, int* bad_index, vhap_stream_t stream);

/* Frame preparation, applied ONCE when decoded frames enter the resident store (ABI 9).
 * vhap_frame_color_correct -- NeRSembleDataset.apply_color_correction (vhap/data/nersemble_dataset.py:160-171): per frame f the affine colour
 *   transform of camera cam_of_frame[f] (null: camera 0): out = uint8(clip(rgb / 255 @ A[:3,:3] + A[:3,3], 0, 1) * 255), fp64, the matmul as
 *   numpy's dgemm evaluates it (an FMA chain over the input channels), truncating cast.  ccm [n_cam, 12] doubles = rows 0..2 of A, four
 *   columns each.  rgb_u8 / rgb_out [N,H,W,3] uint8 (in place allowed).
 * vhap_frame_resize_u8 -- VideoDataset.apply_scale_factor (vhap/data/video_dataset.py:266-300): PIL Image.resize((w, h), BILINEAR) of N
 *   8-bit images [N,H,W,C] -> [N,h,w,C] (C = 3 rgb, 1 alpha map).  The fixed-point coefficient tables are Pillow's (Resample.c
 *   precompute_coeffs + normalize_coeffs_8bpc), computed by the caller: bounds_* [2 * out] = (first tap, tap count), coef_* [out * ksize];
 *   ksize 0 = that axis keeps its size (Pillow skips the pass). */
int vhap_frame_color_correct(const unsigned char* rgb_u8, const int32_t* cam_of_frame, const double* ccm, int n_cam, int N, int H, int W,
                             unsigned char* rgb_out, vhap_stream_t stream);
int vhap_frame_resize_u8(const unsigned char* src_u8, int N, int H, int W, int C, unsigned char* dst_u8, int h, int w,
                         const int32_t* bounds_x, const int32_t* coef_x, int ksize_x, const int32_t* bounds_y, const int32_t* coef_y,
                         int ksize_y, vhap_stream_t stream);
/* Batch feed (the per-step hand-over of a stage loop, tracker.py:1376-1385, as a node of the captured step): batch number cursor[0] of a
 * table uploaded once per pass -- frame_table / ts_table [capacity, n] int64: frame index and timestep of every frame of every batch, in the
 * order the batches are taken -- is gathered into the step's static buffers frame_out / ts_out [n] and, for up to three per-frame row arrays
 * rowsK [N, widthK] (landmarks, intrinsics, extrinsics; NULL = unused), outK [n, widthK]; then cursor[0] += 1 (a second tiny launch).
 * cursor [2] int32: next batch, number of batches in the table (past the end the last batch is taken again).
 * vhap_frame_ingest(index = frame_out) follows as an ordinary node.  The host resets the cursor when it uploads a new table. */
int vhap_batch_feed(const long long* frame_table, const long long* ts_table, int* cursor, int n, int capacity, int N,
                    const float* rows0, float* out0, int width0, const float* rows1, float* out1, int width1,
                    const float* rows2, float* out2, int width2, long long* frame_out, long long* ts_out, vhap_stream_t stream);

/* ---- Step plans: the library's own executor for a captured step (vhap/model/tracker.py:1391-1416 runs the same
 * optimize_iter 50-500 times per stage) -------------------------------------------------------------------------------------------
 * The host records one step under HIP stream capture (hipStreamBeginCapture .. EndCapture -> a hipGraph_t, NOT instantiated) and hands
 * the graph over: vhap_plan_from_graph walks its kernel / memset / empty nodes and dependency edges, lays them out over at most
 * `max_streams` streams (1..8; stream 0 is the stream a replay is launched on, the others belong to the plan; a node's first-captured
 * successor stays on the node's stream, further successors fork onto side streams) and turns cross-stream edges into event record /
 * wait pairs.  vhap_plan_launch replays the plan with plain kernel launches on those streams -- hipGraphInstantiate / hipGraphLaunch are
 * never called (their branch layout is not under the caller's control and ROCm 7's faults when the launch stream shares a hardware
 * queue with two of its internal branch streams).  Any other node type -> VHAP_E_UNSUPPORTED (the caller keeps the graph).
 * The plan BORROWS the graph's kernel-argument storage: the hipGraph_t must stay alive until vhap_plan_destroy.  A plan is an object of
 * its creator: one replay at a time per plan (replays on one launch stream are ordered; different plans are independent). */
typedef struct vhap_plan* vhap_plan_t;
int vhap_plan_from_graph(void* hip_graph, int max_streams, vhap_plan_t* plan);
int vhap_plan_destroy(vhap_plan_t plan);
int vhap_plan_info(vhap_plan_t plan, int* n_nodes, int* n_streams, int* n_events);
/* text dump, one line per node in launch order ("index stream kernel <- predecessors | waits | records"); returns the bytes needed */
size_t vhap_plan_describe(vhap_plan_t plan, char* buf, size_t cap);
int vhap_plan_node_name(vhap_plan_t plan, int node, char* buf, size_t cap);
/* call_flags: 0, or VHAP_CALL_PLAN_DEFER_JOIN -- the replay does NOT make `stream` wait for the plan's side streams at its end: what
 * vhap_plan_open_tails lists keeps running while the caller enqueues more work on `stream`.  The NEXT replay of the same plan is ordered
 * behind that work wherever it matters through the side streams themselves (they are in order); the caller asserts that nothing it
 * enqueues on `stream` before the next vhap_plan_join (or the next joined replay) touches what those open tails read or write.  The step
 * host uses it to start step k+1's geometry chain under step k's texture update (vhap_amd/tracker.py::GraphedStep). */
#define VHAP_CALL_PLAN_DEFER_JOIN 64
/* The side streams of every plan of one host thread come from one pool (side stream k of a plan = pool stream base + k, base 0 unless
 * vhap_plan_set_side_base was called on this thread before the plan was created; ABI 9).  vhap_plan_touch_side_streams issues a 4-byte fill
 * on pool streams 0 .. n - 1 in order: HIP binds a stream to a hardware queue at its first command. */
int vhap_plan_set_side_base(int base);
int vhap_plan_touch_side_streams(int n, void* scratch_4_bytes);
/* pool stream k waits for what is enqueued on `other` now (an external dependency of ONE side chain of the next replay) */
int vhap_plan_side_stream_wait(int k, vhap_stream_t other);
/* The calling thread's side-stream pools, destroyed (every device; each stream synchronised first): for short-lived worker threads that
 * created plans.  Plans of this thread must not be replayed afterwards.  All plans created on one thread SHARE its pool streams (side stream k of
 * every plan is the same HIP stream): replays of unrelated plans serialise behind each other's open tails on those streams.  (ABI 10) */
int vhap_plan_pool_release(void);
int vhap_plan_launch(vhap_plan_t plan, vhap_stream_t stream, int call_flags);
int vhap_plan_join(vhap_plan_t plan, vhap_stream_t stream);
/* the nodes a DEFER_JOIN replay leaves un-joined (indices into launch order, at most `cap` written); returns their number */
int vhap_plan_open_tails(vhap_plan_t plan, int* nodes, int cap);
/* the nodes of the NEXT replay that are not ordered behind those open tails (not on a stream carrying one, not downstream of a node that
 * is): together with whatever the caller enqueues between the replays, these are what must not conflict with the open tails */
int vhap_plan_free_heads(vhap_plan_t plan, int* nodes, int cap);
/* vhap_plan_node_handle: the captured graph's node behind plan node `node` (identity only).  vhap_capture_nodes: the nodes of the graph
 * `stream` is capturing into right now (at most `cap` handles written; returns their number, 0: not capturing) -- read after every call of
 * a capture, it tells the host which call created which plan node, hence which buffers the node touches (the deferred join is decided on
 * those: vhap_amd/tracker.py, _lib.AccessLog). */
void* vhap_plan_node_handle(vhap_plan_t plan, int node);
int vhap_capture_nodes(vhap_stream_t stream, void** nodes, int cap);
/* one replay with every node bracketed by timing events; blocks until done.  start_us[k] (relative to the head of the replay) and
 * dur_us[k] of node k, n >= number of nodes.  For per-kernel numbers inside the step (bench.py's roofline line), not for the step time. */
int vhap_plan_launch_timed(vhap_plan_t plan, vhap_stream_t stream, float* start_us, float* dur_us, int n);

/* -------------------------------------------------------------------------------------------
 * Landmark detection on ROCm (ABI 10): the building blocks of the 2-D landmark network.
 * Reference: vhap/model/tracker.py:1263-1277 -> vhap/util/landmark_detector_fa.py:41-46 `face_alignment.FaceAlignment(TWO_HALF_D, 'sfd',
 * flip_input=True)` -- the third-party `face_alignment` package (absent from the reference checkout), whose network is the published FAN
 * (Bulat & Tzimiropoulos, ICCV 2017).  Host side: vhap_amd/landmarks.py.
 *   vhap_conv2d_nhwc: fp32 convolution on the matrix cores (exact-fp32 MFMA), NHWC.  `in` / `out` point at the FIRST CHANNEL of a channel slice of
 *     a [N,H,W,in_channel_stride] / [N,Ho,Wo,out_channel_stride] buffer (Ho = (H + 2 pad - KH) / stride + 1); weight [KH,KW,Cin,Cout]; bias [Cout] or
 *     NULL; in_scale / in_shift [Cin] or both NULL: the input is read as v * in_scale[ci] + in_shift[ci] (an inference BatchNorm), then through a ReLU
 *     with VHAP_CONV_IN_RELU -- zero padding applies to the activated input; VHAP_CONV_OUT_RELU: ReLU on the way out; VHAP_CONV_ACCUMULATE: out += .
 *   vhap_nhwc_avgpool2: 2x2 / stride-2 average pool.  vhap_nhwc_upsample2_add: out = skip + nearest-neighbour x2 of low ([N,H/2,W/2,C]).
 *   vhap_nhwc_add: out = (a + b) + c over n floats (c may be NULL).
 * ------------------------------------------------------------------------------------------- */
#define VHAP_CONV_IN_RELU 1
#define VHAP_CONV_OUT_RELU 2
#define VHAP_CONV_ACCUMULATE 4
int vhap_conv2d_nhwc(const float* in, int in_channel_stride, int N, int H, int W, int Cin, const float* weight, const float* bias,
                     const float* in_scale, const float* in_shift, int KH, int KW, int stride, int pad, float* out, int out_channel_stride,
                     int Cout, int call_flags, vhap_stream_t stream);
/* vhap_conv2d_nhwc with a workspace (device floats, may be NULL = vhap_conv2d_nhwc): when the pixel x channel tiles alone would leave the chip empty
 * (the deep levels of the hourglass: a handful of workgroups walking 72 K tiles each) the K tiles are split over up to 16 slices whose partial sums
 * [slices][N*Ho*Wo][Cout] go through the workspace and are summed in slice order by a second launch (bias / accumulate / ReLU there).  The split
 * is a function of the shapes and of workspace_floats only: the same call gives the same bits. */
int vhap_conv2d_nhwc_ws(const float* in, int in_channel_stride, int N, int H, int W, int Cin, const float* weight, const float* bias,
                        const float* in_scale, const float* in_shift, int KH, int KW, int stride, int pad, float* out, int out_channel_stride,
                        int Cout, float* workspace, long long workspace_floats, int call_flags, vhap_stream_t stream);
int vhap_nhwc_avgpool2(const float* in, int N, int H, int W, int C, float* out, vhap_stream_t stream);
int vhap_nhwc_upsample2_add(const float* skip, const float* low, int N, int H, int W, int C, float* out, vhap_stream_t stream);
/* The face detector of the same package (`face_detector='sfd'`, landmark_detector_fa.py:32,45,51: S3FD, Zhang et al., ICCV 2017 -- a VGG-16 trunk with six
 * detection heads; host side: vhap_amd/face_detector.py) needs two more layer kinds: vhap_nhwc_maxpool2 = torch's max_pool2d(x, 2, 2) (floor mode:
 * out [N, H/2, W/2, C]); vhap_nhwc_l2norm = its L2Norm layer, out[p, c] = in[p, c] / (sqrt(sum_c in[p, c]^2) + eps) * weight[c] over npix pixels. */
int vhap_nhwc_maxpool2(const float* in, int N, int H, int W, int C, float* out, vhap_stream_t stream);
int vhap_nhwc_l2norm(const float* in, long long npix, int C, const float* weight, float eps, float* out, vhap_stream_t stream);
int vhap_nhwc_add(const float* a, const float* b, const float* c, long long n, float* out, vhap_stream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* VHAP_HIP_H */
