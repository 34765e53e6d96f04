/*
 * drt_hip.h -- C ABI of libdrt_hip.so, the MI355X (gfx950) implementation of DRT's
 * differentiable refraction-tracing hot path.
 *
 * Plain pointers and sizes only: every `d_*` argument is a DEVICE pointer on the GPU
 * the scene was created for, `stream` is a hipStream_t passed as void* (NULL = the
 * null stream).  No call synchronises the host; all work is enqueued on `stream`.
 * Every function returns 0 on success or a negative DRT_E_* code; drt_last_error()
 * gives the message for the calling thread.
 *
 * Two nested boundaries of the reference are covered:
 *   B1  the native tracer class `optix_mesh` (reference optix_extend.cpp:6-83), which
 *       the reference builds on NVIDIA OptiX Prime 6.5;
 *   B2  the per-view math of `Scene` in reference DiffRender.py that consumes the face
 *       ids (render_transparent / silhouette / dihedral) and the ray loss of
 *       reference optim.py:91-108, forward and backward.
 */
#ifndef DRT_HIP_H
#define DRT_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DRT_OK            0
#define DRT_E_INVALID    -1   /* bad argument (null pointer, negative size, no mesh yet) */
#define DRT_E_HIP        -2   /* a HIP runtime call failed */
#define DRT_E_NOMEM      -3

typedef struct drt_scene drt_scene_t;

const char* drt_last_error(void);
int drt_version(void);       /* 2: drt_deterministic / drt_fx_finalize */

/* ---- deterministic accumulation ("race detection / sanitizers"; reference optim.py:155-171 clamps the SUM) --------
 * Every vertex gradient and loss of this library is a sum of contributions scattered with float64 atomics: the same inputs give results
 * that differ from run to run at 1e-16 relative, in the order the hardware served the atomics.  drt_deterministic(1) -- or DRT_DETERMINISTIC=1
 * in the environment, read at the first call that needs to know -- switches the whole PROCESS to order-independent accumulation: every
 * `double* d_grad_verts` / `double* d_loss` / `d_stash` ACCUMULATION TARGET of the entry points below must then point to an array of
 * accumulator cells instead of doubles, DRT_FX_BYTES_PER_VALUE bytes per float64 element (same element numbers), zero-filled by the
 * caller; the kernels add every contribution as a 128-bit fixed-point integer (unit 2^-80, drt_amd/csrc/drt_fixed.h: integer addition
 * is associative, so the sum is the same bit pattern whatever the order; contributions of magnitude >= 3.7e-9 enter exactly, smaller ones
 * are truncated at 8e-25 absolute; |x| >= 7e13, +-inf and NaN set sticky flags and give +-inf / NaN).  drt_fx_finalize converts n cells to
 * float64 with ONE rounding (nearest-even) of the exact sum: d_out[i] = value, or d_out[i] += value with `accumulate`.  Inputs that are
 * READ as gradients (d_grad_out_dir, d_grad_cos, ...) and per-ray outputs stay plain float64.  Same result for eager launches and graph
 * replays, any stream interleaving, any sub-batch split.  The price: no LDS pre-aggregation in the path kernels and two 64-bit integer
 * atomics per component (DESIGN.md section 7 has the measured cost).  drt_deterministic(-1) only queries; returns the previous mode. */
#define DRT_FX_BYTES_PER_VALUE 24
int drt_deterministic(int on);
int drt_fx_finalize(const void* d_cells, int64_t n, double* d_out, int accumulate, void* stream);
/* Sums of sums, still exact: drt_fx_add adds n cells of d_src_cells into d_dst_cells (several calls of one step).  For the step's ONE
 * all-reduce over ranks drt_fx_to_limbs rewrites n cells as int64 [n,4] -- three 43-bit limbs of the 128-bit sum,
 * the top one signed, and the flags as counters -- whose plain per-word SUM over up to 2^20 ranks cannot overflow; drt_fx_from_limbs
 * puts the reduced words back together.  All ranks then convert the SAME integer: N GPUs give the bits one GPU gives. */
int drt_fx_add(void* d_dst_cells, const void* d_src_cells, int64_t n, void* stream);
int drt_fx_to_limbs(const void* d_cells, int64_t n, int64_t* d_limbs, void* stream);
int drt_fx_from_limbs(const int64_t* d_limbs, int64_t n, void* d_cells, void* stream);

/* ---- lifetime: replaces optix_mesh::optix_mesh(cuda_device), optix_extend.cpp:8-12 ---- */
int drt_create(int device, drt_scene_t** out);
void drt_destroy(drt_scene_t* s);

/* ---- B1: acceleration structure -------------------------------------------------------
 * drt_update_mesh  <- optix_mesh::update_mesh(F int32 [F,3], V float32 [V,3]), optix_extend.cpp:14-21
 * drt_update_vert  <- optix_mesh::update_vert(V float32 [V,3]),               optix_extend.cpp:23-27
 * Both copy their inputs (the reference keeps tensor references alive instead) and rebuild
 * the LBVH on `stream`: scene bounds -> 30-bit Morton codes -> radix sort -> Karras
 * hierarchy -> bottom-up box refit -> collapse to a 4-wide tree with <= 4-triangle leaves.
 * drt_update_vert_f64 fuses the reference's `vertices.detach().to(float32)`
 * (DiffRender.py:379) into the rebuild. */
int drt_update_mesh(drt_scene_t* s, const int32_t* d_faces, int64_t n_faces,
                    const float* d_verts, int64_t n_verts, void* stream);
int drt_update_vert(drt_scene_t* s, const float* d_verts, int64_t n_verts, void* stream);
int drt_update_vert_f64(drt_scene_t* s, const double* d_verts, int64_t n_verts, void* stream);

/* drt_intersect <- optix_mesh::intersect(Ray float32 [N,6]) -> T float32 [N], ID int32 [N],
 * optix_extend.cpp:29-57.  Closest hit with t > 0 (equal t: the lowest face id); miss: T = -1, ID = -1.  What a hit is:
 * the float32 Moller-Trumbore test of drt_amd/csrc/drt_tri.h -- the transcription of the reference's own JIT_Dintersect --
 * plus "the hit point lies in the triangle's bounding box grown by 2^-14 of the scene extent", which is what makes the
 * closest hit over ALL triangles computable by a tree for rays that run inside a triangle's plane.
 * Outputs are caller-owned contiguous arrays (the reference returns strided aliases of one buffer).
 * drt_intersect_any writes only a hit flag (uint8) -- what Scene.optix_intersect's callers
 * at DiffRender.py:426 and :224 use. */
int drt_intersect(drt_scene_t* s, const float* d_rays, int64_t n_rays,
                  float* d_T, int32_t* d_ID, void* stream);
int drt_intersect_any(drt_scene_t* s, const float* d_rays, int64_t n_rays,
                      uint8_t* d_hit, void* stream);
/* Same results as drt_intersect by testing every triangle (no BVH); diagnostic used to
 * check the traversal at full workload sizes. */
int drt_intersect_bruteforce(drt_scene_t* s, const float* d_rays, int64_t n_rays,
                             float* d_T, int32_t* d_ID, void* stream);
/* Diagnostic: number of BVH child boxes that fail to enclose their subtree (0 when the
 * build is sound, also counted when the 4-wide tree is too deep for the traversal stack), the binary
 * tree height and (nullable) the depth of the 4-wide tree; host-synchronising. */
int drt_bvh_check(drt_scene_t* s, void* stream, int64_t* n_violations, int32_t* height, int32_t* wide_depth);
/* Diagnostic: copy the Morton-sorted face order (int32 [F]) to d_order. */
int drt_bvh_sorted_faces(drt_scene_t* s, int32_t* d_order, void* stream);
/* How updates treat the tree (also DRT_TREE / DRT_REBUILD_EVERY at drt_create).  mode 0 (default): every update is a full LBVH build --
 * Morton keys, radix sort, Karras hierarchy, refit, 4-wide collapse -- as the reference's OptiX model is rebuilt on every update_vert
 * (optix_extend.cpp:23-27, 61-67).  mode 1: the LBVH's topology is kept for `rebuild_every` updates; the updates in between refit the
 * boxes and re-quantise the wide nodes only (3 launches instead of 7).  mode 2: a binned-SAH topology is built on the HOST at every
 * drt_update_mesh (one device -> host copy of the mesh, milliseconds) and every drt_update_vert* refits it.  Results do not depend on the
 * mode: boxes only decide which triangles are tested (drt_bvh_check validates any of the trees). */
int drt_tree_mode(drt_scene_t* s, int mode, int rebuild_every);
/* Diagnostic: the scene box the last build derived from the vertices -- out7 = lo[3], 1/extent[3], leaf padding (float32; the margin of
 * the hit-point test is half the padding); host-synchronising.  What a replayed capture of an update must reproduce. */
int drt_build_params(drt_scene_t* s, float* out7, void* stream);

/* ---- B2: Scene.render_transparent, DiffRender.py:420-432 (trace2 :537-546, Dintersect
 * :492-501, refract_ray :503-535) -------------------------------------------------------
 * In : d_verts float64 [V,3] (the differentiable vertices; the BVH must have been built
 *      from their float32 cast), origin/dir float64 [N,3].
 * Out: out_ori, out_dir float64 [N,3] (zeros where the path is invalid), mask uint8 [N,3]
 *      (the reference's bool [N,3]), face1/face2 int32 [N]: faces hit at bounce 1 / 2,
 *      face2 >= 0 exactly for the rays with mask = 1 (saved for backward), face1 = -1 on
 *      a primary miss.  Optional (both or neither): d_valid_idx int32 [N] receives the indices of
 *      the rays with mask = 1 (unordered), *d_n_valid (int64, device) their number -- handing
 *      them to drt_render_backward spares it a pass over the dense arrays.
 *      tile_w, tile_h: 0, or the width / height in pixels of the image(s) whose rows the rays are (width a multiple
 *      of 64, height a multiple of 4, whole images concatenated): a HINT that lets the pipeline group rays by
 *      16x4-pixel screen tiles and, when every image verifies as a pinhole ray grid (generate_ray, reference
 *      captured_data.py:23-40), decide the primary hits by projecting the triangles instead of traversing the
 *      tree (csrc/drt_raster.h).  Every ray is checked against the fitted grid on the device and takes the
 *      tree otherwise: results do not depend on the hint.
 *      grid_mode, d_grid_cache: DRT_GRID_NONE / NULL, or a caller-owned device buffer of DRT_GRID_CACHE_BYTES per image that
 *      belongs to THIS (origin, dir) pair with THIS tile_w x tile_h.  DRT_GRID_ESTABLISH: the call fits and verifies as
 *      above and records, per image, the model and whether every ray verified.  DRT_GRID_TRUST: the caller guarantees the
 *      ray arrays have not changed since the establishing call; images recorded as all-verified are then not re-fitted,
 *      only the 8x8 lattice of 64 sample rays per image is re-verified against the recorded model (an image that fails is
 *      re-fitted, verified ray by ray like in a call without a cache, and marked untrusted IN THE CACHE for good), and rays of
 *      pixels that no projected triangle touches are not even loaded.  The lattice check catches ray arrays that were replaced
 *      wholesale; a change confined to rays off the lattice is the caller's responsibility.  (The Python layer ties the buffer
 *      to the tensor objects and their version counters.) */
#define DRT_GRID_NONE 0
#define DRT_GRID_ESTABLISH 1
#define DRT_GRID_TRUST 2
#define DRT_GRID_CACHE_BYTES 104
/* OR-ed into grid_mode: d_face1 / d_face2 are only guaranteed for the rays with mask = 1 (what drt_render_backward* read);
 * in DRT_GRID_TRUST mode -- and in any mode when the three dense outputs were offered through drt_outputs_clean -- the -1 entries of
 * all other rays are then not written (8 bytes per ray less to fill). */
#define DRT_GRID_SPARSE_FACES 16
/* OR-ed into DRT_GRID_TRUST: the caller has read the cache back after the establishing call and EVERY image in it is recorded
 * as a pinhole grid in all of its rays (int32 `ok` and `all` at byte offsets 96 and 100 of each DRT_GRID_CACHE_BYTES record
 * both non-zero).  Normally no ray then needs the tree for its primary hit: the launches that serve such rays in front of the
 * first shading stage are not issued and the call waits for the (asynchronous) tree build only in front of the second traversal.
 * A performance hint, not a promise the results depend on: an image can still lose its status inside the call (a triangle reaches
 * its camera plane, its large-triangle list overflows, its rays fail the lattice re-check below); its rays are then traced through
 * the tree by a small fallback launch behind the build (csrc/drt_pipeline.hip, k_gen_late). */
#define DRT_GRID_ALL_VERIFIED 32
int drt_render_forward(drt_scene_t* s, const double* d_verts, const double* d_origin,
                       const double* d_dir, int64_t n_rays, double ior_int, double ior_ext,
                       double* d_out_ori, double* d_out_dir, uint8_t* d_mask,
                       int32_t* d_face1, int32_t* d_face2,
                       int32_t* d_valid_idx, int64_t* d_n_valid, int tile_w, int tile_h,
                       int grid_mode, void* d_grid_cache, void* stream);
/* Adjoint of drt_render_forward w.r.t. the vertices: d_grad_verts float64 [V,3] += ...
 * (atomic accumulation; zero it first).  Either incoming gradient may be NULL (= zeros). */
int drt_render_backward(drt_scene_t* s, const double* d_verts, const double* d_origin,
                        const double* d_dir, int64_t n_rays, double ior_int, double ior_ext,
                        const int32_t* d_face1, const int32_t* d_face2,
                        const int32_t* d_valid_idx, const int64_t* d_n_valid,
                        const double* d_grad_out_ori, const double* d_grad_out_dir,
                        double* d_grad_verts, void* stream);

/* ---- Loss_calculator.ray_loss, optim.py:91-108 ------------------------------------------
 * loss = sum over rays with valid & mask of |out_dir - normalize(screen_pixel - out_ori)|^2.
 * Forward accumulates into *d_loss (float64 scalar, zero it first) and, when
 * d_grad_out_dir is not NULL, writes d loss / d out_dir float64 [N,3] (zeros elsewhere).  Optional
 * (both or neither): d_list int32 [N] receives the indices of the contributing rays and *d_n_list
 * (uint32, device, zero it first) their number. */
int drt_ray_loss(const double* d_out_ori, const double* d_out_dir, const uint8_t* d_mask,
                 const double* d_screen_pixel, const uint8_t* d_valid, int64_t n_rays,
                 double* d_loss, double* d_grad_out_dir,
                 int32_t* d_list, uint32_t* d_n_list, void* stream);
/* The same loss over a list of the rays with mask = 1 (drt_render_forward's d_valid_idx / d_n_valid) instead of a pass over
 * all N rays: *d_loss += ..., and (optional, both or neither) d_rows / *d_n_rows (uint32, zero it first) = the listed rays that
 * also have a target -- what drt_render_backward_ray_loss takes. */
int drt_ray_loss_listed(const double* d_out_ori, const double* d_out_dir, const double* d_screen_pixel,
                        const uint8_t* d_valid, const int32_t* d_paths, const int64_t* d_n_paths, int64_t n_rays,
                        double* d_loss, int32_t* d_rows, uint32_t* d_n_rows, void* stream);
/* Backward helper of the loss: x[list[k], 0..2] *= *d_scale for the *d_n_list rows listed (the rows
 * drt_ray_loss reported as contributing) -- rescales d loss / d out_dir by the incoming scalar
 * gradient without another pass over the dense [N,3] tensor. */
int drt_scale_rows3(double* d_x, const int32_t* d_list, const uint32_t* d_n_list,
                    const double* d_scale, void* stream);

/* Zero `bytes` bytes at d_buf on the library's idle stream (the build stream: free between a forward's join and the next build), after
 * everything already enqueued on `stream`.  A later drt_render_forward on this scene whose d_out_ori, d_out_dir or d_mask IS d_buf (same
 * pointer, the matching size) does not fill that output again -- the fill of the NEXT step's outputs then runs beside the caller's loss /
 * backward / optimiser kernels instead of beside the next forward.  Up to three buffers may be pending; entries are forgotten at the next
 * drt_render_forward.  The reference has no counterpart: its outputs are torch.zeros of every call (DiffRender.py:421-423).
 * drt_prefill_wait: `stream` waits for every zeroing enqueued so far and the pending entries are forgotten -- what a caller does before it
 * releases such a buffer WITHOUT rendering into it (its allocator may hand the memory to work on `stream`); drt_destroy waits by itself. */
int drt_prefill_zero(drt_scene_t* s, void* d_buf, int64_t bytes, void* stream);
int drt_prefill_wait(drt_scene_t* s, void* stream);
/* The outputs of an EARLIER drt_render_forward (same scene or not) offered again as the outputs of the next one of `n_rays` rays: they
 * are zeros in every row but the rows that call listed in d_valid_idx / d_n_valid (a call's dense outputs are zero wherever its mask is
 * -- the reference's torch.zeros + index_put, DiffRender.py:421-431 -- and its list of completed paths is exactly the set rows), so
 * zeroing THOSE rows (51 B per listed row) leaves the three buffers as freshly zeroed ones, and the next drt_render_forward on this scene
 * with exactly these pointers does not fill them again.  The zeroing kernel is enqueued BY that drt_render_forward (same `stream`: behind
 * the fork of its internal pipelines when the call is a trusted-grid one, in front of it otherwise; by the next drt_outputs_clean if no
 * render call comes in between) (the fills are 51 B per RAY: 3.85 GB of the
 * 72 x 1024^2 step, its one HBM-bound stage).  The caller vouches that nothing else wrote the buffers since that earlier call and that
 * nobody else still reads them (drt_amd/diffrender.py: storage use counts and version counters); entries are forgotten at the next
 * drt_render_forward, like drt_prefill_zero's.  Not while a graph is being captured. */
int drt_outputs_clean(drt_scene_t* s, double* d_out_ori, double* d_out_dir, uint8_t* d_mask, int64_t n_rays,
                      const int32_t* d_valid_idx, const int64_t* d_n_valid, void* stream);
/* Withdraws a drt_outputs_clean request that no drt_render_forward has taken yet (the caller could not make the render call after all:
 * an allocation failed in between): the library forgets the five pointers without touching the memory.  No-op without a pending request.
 * A drt_render_forward that returns an error withdraws the request (and every other pointer registered for it) by itself. */
int drt_outputs_cancel(drt_scene_t* s);
/* Temporal hit seeds for the NEXT drt_render_forward / drt_render_ray_loss_fused on this scene of exactly `n_rays` rays (one shot; a call
 * of another size ignores and forgets them).  d_seed_face2: int32 [n_rays], caller-owned, one entry per camera ray: the face id the
 * REFRACTED ray of that pixel hit (traversal #2 of reference DiffRender.py:542) in an earlier call on the same rays, -1 for none -- any
 * other value outside [0, F) is ignored, so a buffer that outlived a topology change is harmless.  The call tests that triangle first and
 * starts the traversal with its distance as the bound, then writes the face it found back into the buffer (entries of rays without a
 * second hit keep their value).  Results are identical bit for bit with any buffer content (drt_amd/csrc/drt_trace_kernel.h TraceSeed):
 * the optimisation loop moves vertices by at most lr x clamp per step (reference optim.py:155-171), so last step's face is nearly always
 * this step's -- what the seed buys is node visits.  The reference has no counterpart (OptiX Prime keeps no state between queries).
 * The buffer must stay valid until the call it is consumed by has finished on its stream.  DRT_HIT_SEED=0 disables.  The ORDER of the
 * entries is the library's business (ray number, or 4 x 4-pixel tiles of the image for whole-image calls: fewer cache lines per wave); a
 * caller only ever initialises the buffer (-1) and hands it back. */
int drt_render_seed(drt_scene_t* s, int32_t* d_seed_face2, int64_t n_rays);
/* ray_loss (reference optim.py:91-108) AND its vertex gradient in one pass over the forward's list of completed paths
 * (drt_render_forward's d_valid_idx / d_n_valid, face ids from the same call): *d_loss += the loss (float64 scalar, zero it first)
 * and d_grad_verts float64 [V,3] += d loss / d vertices with a UNIT seed (the caller scales it by the incoming gradient of the loss:
 * the loss is a scalar, its backward through render_transparent is linear in that seed).  Replaces drt_ray_loss_listed followed,
 * in the backward pass, by drt_render_backward_ray_loss when out_dir is the tensor drt_render_forward produced. */
int drt_ray_loss_listed_grad(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                             double ior_int, double ior_ext, const int32_t* d_face1, const int32_t* d_face2,
                             const double* d_screen_pixel, const uint8_t* d_valid, const int32_t* d_paths, const int64_t* d_n_paths,
                             double* d_loss, double* d_grad_verts, void* stream);
/* The same, started EARLY on the part of the list that is ready early: a drt_render_forward cut into sub-batches keeps the completed paths
 * of its first sub-batch as the head of d_valid_idx (never moved by the join); when d_paths is that list, the head is processed on the
 * internal stream that produced it -- beside the tail of the other pipeline -- and only the rest behind the join on `stream`, which then
 * waits for both.  PRECONDITIONS (the caller's): d_loss and d_grad_verts were zeroed on `stream` BEFORE that drt_render_forward was
 * enqueued, and d_screen_pixel / d_valid were complete by then.  Any other list, or a capture: exactly drt_ray_loss_listed_grad. */
int drt_ray_loss_listed_grad_split(drt_scene_t* s, const double* d_verts, const double* d_origin, const double* d_dir, int64_t n_rays,
                                   double ior_int, double ior_ext, const int32_t* d_face1, const int32_t* d_face2,
                                   const double* d_screen_pixel, const uint8_t* d_valid, const int32_t* d_paths, const int64_t* d_n_paths,
                                   double* d_loss, double* d_grad_verts, void* stream);
/* Adjoint of drt_render_forward followed by drt_ray_loss, for the rays drt_ray_loss listed (d_rows / *d_n_rows):
 * grad_verts [V,3] += *d_scale * d ray_loss / d vertices.  Equivalent to drt_scale_rows3 + drt_render_backward with the
 * dense d loss / d out_dir, without that [N,3] tensor (zero in all but the listed rows) ever being written or read:
 * the loss gradient of a listed ray is recomputed from its path. */
int drt_render_backward_ray_loss(drt_scene_t* s, const double* d_verts, const double* d_origin,
                                 const double* d_dir, int64_t n_rays, double ior_int, double ior_ext,
                                 const int32_t* d_face1, const int32_t* d_face2,
                                 const int32_t* d_rows, const uint32_t* d_n_rows,
                                 const double* d_screen_pixel, const double* d_scale,
                                 double* d_grad_verts, void* stream);

/* One pass: render_transparent + ray_loss + d ray_loss / d vertices, nothing dense written.
 * *d_loss += loss, d_grad_verts [V,3] += gradient (both float64, zero them first);
 * d_n_valid (int64, may be NULL) += number of contributing rays. */
int drt_render_ray_loss_fused(drt_scene_t* s, const double* d_verts, const double* d_origin,
                              const double* d_dir, const double* d_screen_pixel,
                              const uint8_t* d_valid, int64_t n_rays, double ior_int,
                              double ior_ext, double* d_loss, double* d_grad_verts,
                              int64_t* d_n_valid, int tile_w, int tile_h,
                              int grid_mode, void* d_grid_cache, void* stream);

/* ---- smoothness branch: Scene.dihedral_angle (DiffRender.py:440-443, edge_face_norm :149-163)
 * and Loss_calculator.sm_loss (optim.py:82-89) ---------------------------------------------------
 * d_e2f int64 [E,2,3]: the three vertex ids of the two faces of every unique edge (Scene.E2F).
 * forward: cos of the dihedral angle per edge, float64 [E].  backward: grad_verts [V,3] += adjoint
 * for a given d loss / d cos [E].  drt_sm_loss_fused: *d_loss += sum -log(1 + cos) and
 * grad_verts += its gradient in one pass.  (Atomic accumulation: zero the outputs first.) */
int drt_dihedral_forward(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, double* d_cos, void* stream);
int drt_dihedral_backward(const double* d_verts, const int64_t* d_e2f, int64_t n_edges,
                          const double* d_grad_cos, double* d_grad_verts, void* stream);
int drt_sm_loss_fused(const double* d_verts, const int64_t* d_e2f, int64_t n_edges, double* d_loss,
                      double* d_grad_verts, void* stream);

/* ---- silhouette branch ---------------------------------------------------------------------------
 * drt_silhouette_flags <- Scene.silhouette_edge (DiffRender.py:445-457): flags uint8 [E], 1 where the
 * two faces of the edge face opposite ways seen from d_origin3 (float64 [3], device).
 * drt_edge_sample_forward <- Scene.primary_visibility + primary_edge_sample.forward
 * (DiffRender.py:459-479, 189-258) for d_edges int64 [Es,2] (vertex ids): projects both endpoints with
 * d_camera = float64 [50] = R(4x4) | K(3x3) | R^-1(4x4) | K^-1(3x3) row-major on the device, probes one
 * pixel either side of the projected midpoint with any-hit rays; writes index int64 [Es,2] =
 * trunc(midpoint x, y) and f float32 [Es] = hit(+) - hit(-) in {-1, 0, 1}.  The caller keeps edges with
 * |f| > 1e-5 (DiffRender.py:244) and in-view indices (:478): d_keep (uint8 [Es], may be NULL) = exactly that test for a
 * resx x resy image, so that the caller needs one boolean index instead of ten elementwise launches.
 * d_flags (uint8 [Es], may be NULL): d_edges is then the list of ALL unique edges and only the flagged ones (drt_silhouette_flags)
 * are silhouette edges -- the others are skipped (f = 0, keep = 0 are written for them, their index rows are left untouched) -- which
 * spares the caller the device->host round trip of compacting Edges[flags] between the two calls.
 * drt_edge_sample_backward <- primary_edge_sample.backward (DiffRender.py:263-267) chained through the
 * projection (depth row detached when detach_depth != 0, DiffRender.py:470-471):
 * grad_verts [V,3] += sum_e coef[e] * f[e] * d(-N_e . E_pos)/dV, coef float64 [Es] = incoming
 * d loss / d output per edge (0 for dropped edges).
 * drt_edge_sample_backward_rows: the same sum over the kept rows only -- d_rows int64 [n_rows] = edge row of the k-th kept sample
 * (the caller's boolean index), d_g float32 [n_rows] = d loss / d output of that sample, as autograd hands it over. */
int drt_silhouette_flags(const double* d_verts, const int64_t* d_e2f, int64_t n_edges,
                         const double* d_origin3, uint8_t* d_flags, void* stream);
int drt_edge_sample_forward(drt_scene_t* s, const double* d_verts, const int64_t* d_edges, int64_t n_edges,
                            const double* d_camera, const double* d_origin3, int64_t* d_index,
                            float* d_f, uint8_t* d_keep, int resx, int resy, const uint8_t* d_flags, void* stream);
int drt_edge_sample_backward(const double* d_verts, const int64_t* d_edges, int64_t n_edges,
                             const double* d_camera, const float* d_f, const double* d_coef,
                             int detach_depth, double* d_grad_verts, void* stream);
int drt_edge_sample_backward_rows(const double* d_verts, const int64_t* d_edges, int64_t n_edges,
                                  const double* d_camera, const float* d_f, const int64_t* d_rows, int64_t n_rows,
                                  const float* d_g, int detach_depth, double* d_grad_verts, void* stream);

/* drt_vh_term <- the loss expression of ONE view, optim.py:78 `(mask.view(resy, resx)[index[:,1], index[:,0]] - output).abs().sum()`, over
 * the samples a drt_edge_sample_forward call left in place: d_index int64 [E,2] and d_keep uint8 [E] as that call wrote them (rows with
 * keep = 0 are not read), d_soft_mask float64 [resy*resx], `output` = 0.5.  *d_loss += the sum (zero it first); d_dterm float64 [E] =
 * d term / d output per row (-sign(mask - 0.5); 0 for dropped rows) -- times the incoming gradient it is drt_edge_sample_backward's
 * `coef`.  Lets the drop-in pair evaluate the reference's own expression without compacting the samples (no device->host round trip). */
int drt_vh_term(const int64_t* d_index, const uint8_t* d_keep, int64_t n_edges, const double* d_soft_mask, int resx, int resy,
                double* d_loss, double* d_dterm, void* stream);
/* drt_edge_sample_backward with coef[e] = d_dterm[e] * float32(*d_g): drt_vh_term's per-row derivative times the incoming scalar gradient of
 * the view's term (a float64 DEVICE scalar), rounded to float32 as autograd rounds it on its way into the reference's float32 `output`
 * (DiffRender.py:251, 263-267) -- the backward of the lazily evaluated expression in one launch. */
int drt_edge_sample_backward_term(const double* d_verts, const int64_t* d_edges, int64_t n_edges, const double* d_camera, const float* d_f,
                                  const double* d_dterm, const double* d_g, int detach_depth, double* d_grad_verts, void* stream);

/* drt_vh_loss_fused <- Loss_calculator.vh_loss (optim.py:73-78) for n_views views: silhouette_edge +
 * primary_visibility + sum |soft_mask[y, x] - output| and its vertex gradient, entirely on the device
 * (the drop-in methods above return dynamically sized tensors and cost two host syncs per view; here the
 * views are traced together, 16 per launch).  d_edges int64 [E,2] = Scene.Edges, d_e2f int64 [E,2,3] =
 * Scene.E2F (same edge order).  d_cameras / d_origins / d_soft_masks are HOST arrays of n_views DEVICE
 * pointers: camera float64 [50], origin float64 [3], soft mask float64 [resy*resx].
 * *d_loss += loss, d_grad_verts [V,3] += gradient (zero them first). */
int drt_vh_loss_fused(drt_scene_t* s, const double* d_verts, const int64_t* d_edges, const int64_t* d_e2f,
                      int64_t n_edges, int n_views, const double* const* d_cameras,
                      const double* const* d_origins, const double* const* d_soft_masks, int resx, int resy,
                      int detach_depth, double* d_loss, double* d_grad_verts, void* stream);

/* drt_closest_point <- the reference's acceptance metric, "average per-vertex distance (Hausdorff
 * Distance)" against the scanned mesh, which it delegates to meshlabserver (README.md:11; no code in the
 * repository): for each query point the distance to the closest point of the scene's surface, on the same
 * LBVH.  d_points float64 [n,3] -> d_dist float64 [n]; optional (nullable) d_face int32 [n] (a face
 * attaining the minimum) and d_closest float64 [n,3].  float64 arithmetic on the tracer's float32 vertices. */
int drt_closest_point(drt_scene_t* s, const double* d_points, int64_t n, double* d_dist, int32_t* d_face,
                      double* d_closest, void* stream);

/* ---- remeshing between passes ON THE DEVICE: the geometric kernels of drt_amd/remesh_gpu.py (csrc/drt_remesh_gpu.hip) ----------
 * The data-parallel form of drt_remesh_isotropic below (same algorithm and acceptance rules; every candidate operation evaluated at
 * once, the ones whose neighbourhoods do not overlap applied together, in rounds).  d_faces int64 [F,3], d_verts float64 [V,3];
 * d_vf_start int64 [V+1] / d_vf_face int64 [3F]: vertex -> incident faces (CSR, ascending face ids; drt_rm_vertex_faces); d_vn float64
 * [V,3]: area-weighted vertex normals (drt_rm_vertex_normals).  All pointers are DEVICE pointers; everything is enqueued on `stream`.
 *   drt_rm_vertex_faces     the CSR pair of the mesh as it stands (faces whose indices are -1 -- drt_rm_kill_faces -- are in nobody's
 *                           list): count, scan, fill, each run sorted ascending; d_count int32 [V]: workspace.  With d_vn (and d_verts)
 *                           non-null also the vertex normals, summed in list order (what drt_rm_vertex_normals gives).
 *   drt_rm_split_mark       d_flag uint8 [3F]: 1 on the lo -> hi directed-edge slot c = 3 f + k of every edge longer than max_len.
 *   drt_rm_split_plan       d_rank int64 [3F] = inclusive prefix sum of d_flag (the caller's): d_mid_of_slot int64 [3F] = for every slot, on
 *                           both sides of its edge, the new vertex n_verts + rank - 1 of that edge or -1; d_count int64 [F]: faces each face
 *                           becomes (1..4).
 *   drt_rm_split_faces      d_verts float64 [V + n_split, 3] holding the old vertices: writes the midpoints behind them and, with d_offset =
 *                           exclusive prefix sum of the counts, the new faces (1 -> 2, 2 -> 3 by the shorter diagonal, 3 -> 4).
 *   drt_rm_collapse_eval_all  every directed-edge slot c = 3 f + k of d_faces a candidate (below): ok uint8 [3F] = every rule but the surface
 *                           distance; d_query float64 [3F, max_q, 3] / d_n_query int32 [3F]: the points whose distance to the input surface
 *                           is still to be checked (midpoint, centroids of the faces that survive).
 *   drt_rm_collapse_apply   claim (64-bit atomicMin of (claim / apply pair number counting down, length class, hash(edge, seed), edge index)
 *                           on d_lock uint64 [V] -- workspace, preset here when round == 0, the first round of a step; d_length float64
 *                           [3F]: edge lengths at the start of the round) what each candidate with ok = 1 writes, and apply those that
 *                           nobody with a higher priority contests: faces rewritten in place, d_f_alive / d_v_alive cleared for what
 *                           dies, *d_n_done += number applied (cumulative: the caller zeroes it).  `sub_rounds` claim / apply pairs on the
 *                           same evaluation and tables: a collapse that went ahead stamps what it read or wrote in d_dirty uint8 [V]
 *                           (workspace, cleared when round == 0) with round + 1, and later sub-rounds admit only candidates whose
 *                           vertices and rings are clean (their verdict stands).  round in 0 .. 254, consecutive within a step.
 *   drt_rm_flip_eval/apply  the same for edge flips, again one candidate per directed-edge slot (the face across the edge and the "new edge
 *                           exists already" test come from the vertex -> face lists: no edge table); d_quad int64 [3F,6] = a b c d f1 f2 of a
 *                           flip that passes; d_query float64 [3F,3] the midpoint of the new edge; n_items = 3 F; round, d_lock uint64 [V],
 *                           d_dirty uint8 [V] and sub_rounds as above.
 *   drt_rm_smooth_target    tangential relaxation targets float64 [V,3].
 *   drt_rm_face_agreement   cosine between each face normal and the consensus of its corners, float64 [F] (before a move).
 *   drt_rm_move_check       after vertices moved: every face that degenerated or folded (against d_a0) takes its three vertices back from
 *                           d_old; *d_n_bad = their number (the caller repeats, four rounds at most). */
int drt_rm_split_mark(const int64_t* d_faces, int64_t n_faces, const double* d_verts, double max_len, uint8_t* d_flag, void* stream);
int drt_rm_split_plan(const int64_t* d_faces, int64_t n_faces, const int64_t* d_vf_start, const int64_t* d_vf_face, const uint8_t* d_flag,
                      const int64_t* d_rank, int64_t n_verts, int64_t* d_mid_of_slot, int64_t* d_count, void* stream);
int drt_rm_split_faces(const int64_t* d_faces, int64_t n_faces, const int64_t* d_mid_of_slot, double* d_verts, const int64_t* d_offset,
                       int64_t* d_faces_out, void* stream);
int drt_rm_vertex_faces(const int64_t* d_faces, int64_t n_faces, int64_t n_verts, int32_t* d_count, int64_t* d_vf_start, int64_t* d_vf_face,
                        const double* d_verts, double* d_vn, const int32_t* d_live, void* stream);
int drt_rm_vertex_normals(const int64_t* d_faces, const double* d_verts, const int64_t* d_vf_start, const int64_t* d_vf_face, int64_t n_verts,
                          double* d_vn, void* stream);
int drt_rm_collapse_apply(const int64_t* d_cand, int64_t n_cand, const uint8_t* d_ok, const int64_t* d_edges, int64_t* d_faces, double* d_verts,
                          const int64_t* d_vf_start, const int64_t* d_vf_face, int64_t n_verts, double min_len, uint32_t seed, int round, const double* d_length,
                          uint64_t* d_lock, uint8_t* d_f_alive, uint8_t* d_v_alive, uint8_t* d_dirty, int sub_rounds, int32_t* d_n_done, const int32_t* d_live, void* stream);
/* Round 6: the collapse evaluation runs over EVERY directed-edge slot c = 3 f + k of d_faces (no candidate list: no stream compaction, no host round
 * trip per round).  Slots that are not the lo -> hi representative of their edge, not short, or belong to a face an earlier round killed
 * (indices -1, drt_rm_kill_faces) report ok = 0.  Outputs sized [3 n_faces]: d_edge_snap int64 [.,2] (the slot's edge at the start of the
 * round: what drt_rm_collapse_apply takes as d_edges with d_cand = NULL), d_length, d_ok, d_n_query, d_query [., max_q, 3].
 * drt_rm_surface_filter: CheckSurfDist on the device -- item i keeps d_ok[i] only if all of its d_n_query[i] (NULL: one) points
 * d_query[i][k] lie within max_dist of the surface held by scene `s` (the verdict of drt_closest_point's distance, by a search that
 * starts bounded by max_dist and ends at the first triangle inside).
 * drt_rm_kill_faces: faces with d_f_alive[f] == 0 get the indices -1 in place, and d_f_alive[f] = 1 again (all ones for the next round). */
int drt_rm_collapse_eval_all(const int64_t* d_faces, int64_t n_faces, const double* d_verts, const double* d_vn, const int64_t* d_vf_start,
                             const int64_t* d_vf_face, double min_len, double max_len, int max_q, int64_t* d_edge_snap, double* d_length,
                             uint8_t* d_ok, int32_t* d_n_query, double* d_query, int32_t* d_list_item, double* d_list_point, uint32_t* d_list_count,
                             int64_t list_cap, const int32_t* d_live, void* stream);
/* (d_list_item / d_list_point / d_list_count, optional: the query points of the candidates that passed, appended to ONE compact list of at
 * most list_cap entries -- item = slot, point [3] -- whose length stays on the device; a candidate that does not fit is left to the next
 * round.  drt_rm_surface_filter_list is drt_rm_surface_filter over that list.) */
int drt_rm_surface_filter_list(drt_scene_t* s, uint8_t* d_ok, const int32_t* d_list_item, const double* d_list_point, const uint32_t* d_list_count,
                               int64_t list_cap, double max_dist, const int32_t* d_live, void* stream);
int drt_rm_surface_filter(drt_scene_t* s, uint8_t* d_ok, const int32_t* d_n_query, const double* d_query, int64_t n_items, int max_q,
                          double max_dist, const int32_t* d_live, void* stream);
/* The projection step's query: d_closest float64 [n,3] = the closest point of scene `s` to each of d_points float64 [n,3] -- the point
 * drt_closest_point reports, by a search that starts bounded by hint_radius (> 0; the vertices of a remesh sit within a fraction of an
 * edge length of the surface) and falls back to the unbounded one for a point farther out. */
int drt_rm_closest_near(drt_scene_t* s, const double* d_points, int64_t n, double hint_radius, double* d_closest, void* stream);
int drt_rm_kill_faces(int64_t* d_faces, uint8_t* d_f_alive, int64_t n_faces, const int32_t* d_live, void* stream);
/* Round termination on the device.  d_ctl int32 [8], preset {1, 0, 0, 0, 0, ...} at the start of a step: [0] live, [1] operations applied so
 * far (pass d_ctl + 1 as d_n_done of the apply calls), [2] / [3] the previous / the first round's figures, [4] rounds that ran.
 * drt_rm_round_end, enqueued after each round, clears [0] when the round applied nothing or fewer than first / tail_cut; every
 * call above that takes d_live (= d_ctl, or NULL: always live) turns into a no-op from then on -- so the driver enqueues several rounds
 * ahead and reads d_ctl back once per batch instead of once per round. */
int drt_rm_round_end(int32_t* d_ctl, int tail_cut, void* stream);
int drt_rm_flip_eval(const int64_t* d_faces, int64_t n_faces, const double* d_verts, const double* d_vn, const int64_t* d_vf_start,
                     const int64_t* d_vf_face, double max_len, uint8_t* d_ok, int64_t* d_quad, double* d_query, const int32_t* d_live, void* stream);
int drt_rm_flip_apply(int64_t n_items, const uint8_t* d_ok, const int64_t* d_quad, int64_t* d_faces, int64_t n_verts, int round, uint64_t* d_lock,
                      uint8_t* d_dirty, int sub_rounds, int32_t* d_n_done, const int32_t* d_live, void* stream);
int drt_rm_smooth_target(const int64_t* d_faces, const double* d_verts, const int64_t* d_vf_start, const int64_t* d_vf_face, int64_t n_verts,
                         double* d_target, void* stream);
int drt_rm_face_agreement(const int64_t* d_faces, const double* d_verts, const double* d_vn, int64_t n_faces, double* d_a0, void* stream);
int drt_rm_move_check(const int64_t* d_faces, double* d_verts, const double* d_old, const double* d_vn, const double* d_a0, int64_t n_faces,
                      int64_t n_verts, uint8_t* d_revert, int32_t* d_n_bad, void* stream);

/* ---- remeshing between passes (host code; all pointers here are HOST pointers) -------------------------
 * drt_remesh_isotropic <- Meshlabserver.remesh (optim.py:12-52): MeshLab's "Remeshing: Isotropic Explicit
 * Remeshing" with Iterations=iterations (3), TargetLen=target_len, CheckSurfDist / MaxSurfDist=max_surf_dist
 * (1), and the refine / collapse / edge-swap / smooth / reproject steps selected by `flags` (all on in the
 * reference).  Input: a closed manifold, verts float64 [V,3], faces int32 [F,3].  The result is returned
 * in an opaque buffer: query its size, copy it out, free it. */
#define DRT_REMESH_SPLIT 1u
#define DRT_REMESH_COLLAPSE 2u
#define DRT_REMESH_FLIP 4u
#define DRT_REMESH_SMOOTH 8u
#define DRT_REMESH_REPROJECT 16u
#define DRT_REMESH_CHECK_DIST 32u
#define DRT_REMESH_ALL 63u
typedef struct drt_mesh_buf drt_mesh_buf_t;
int drt_remesh_isotropic(const double* verts, int64_t n_verts, const int32_t* faces, int64_t n_faces,
                         double target_len, int iterations, double max_surf_dist, unsigned flags,
                         drt_mesh_buf_t** out);
/* stats4 (nullable): edges split, edges collapsed, edges flipped, iterations run */
int drt_mesh_buf_size(const drt_mesh_buf_t* b, int64_t* n_verts, int64_t* n_faces, int64_t* stats4);
int drt_mesh_buf_copy(const drt_mesh_buf_t* b, double* verts, int32_t* faces);
void drt_mesh_buf_free(drt_mesh_buf_t* b);

/* ---- limit_hook + SGD step (optim.py:155-171, 215) in one pass.  d_grad is sanitised in place when max_abs > 0 (NaN -> 0, clamp to
 * +-max_abs); d_buf is torch.optim.SGD's momentum buffer (first != 0: initialised from the gradient); n = number of float64
 * elements of the parameter. */
int drt_limit_sgd_step(double* d_param, double* d_grad, double* d_buf, int64_t n, double lr, double momentum,
                       int nesterov, int first, double max_abs, void* stream);
/* The same with the weighted sum of the three loss terms in front (all_loss, optim.py:127-129): d_terms float64 [3, n] = d ray / d vertices,
 * d vh / d vertices, d sm / d vertices; d_w3 float64 [3] (device) the weights; grad[i] = (w0 t0[i] + w1 t1[i]) + w2 t2[i] before the limit;
 * optional (both or neither) d_loss_parts float64 [3] -> *d_loss_total = (w0 l0 + w1 l1) + w2 l2. */
int drt_limit_sgd_step3(double* d_param, double* d_grad, double* d_buf, int64_t n, double lr, double momentum, int nesterov, int first,
                        double max_abs, const double* d_terms, const double* d_w3, const double* d_loss_parts, double* d_loss_total, void* stream);
/* The library's own HIP streams (which = 0: the build stream, idle once the tree is built; 1 .. DRT_STREAMS: the pipeline streams, of which
 * a call below 2^25 rays uses only the first).  A process gets four hardware queues and further streams are multiplexed onto them: a
 * caller that wants side work to run BESIDE a render call -- not behind the barrier with which the caller's own stream waits for it in
 * the same hardware queue -- enqueues it on one of these (optim.FusedIteration: the silhouette and smoothness terms on pipeline stream 2). */
int drt_internal_stream(drt_scene_t* s, int which, void** out);

/* ---- topology: Scene.init_edge (DiffRender.py:338-355; trimesh group_rows / edges_face on the host in the reference) and
 * the 1 -> 4 midpoint refinement of a level-of-detail step, on the device ---------------------------------------
 * drt_edge_tables: d_faces int64 [F,3], d_verts float64 [V,3] -> d_edges int64 [3F/2,2] (ascending by (min, max) vertex),
 * d_e2f int64 [3F/2,2,3] (vertex ids of the two faces of every edge; first the face with the lower directed-edge row),
 * d_row2edge int32 [3F] (nullable: unique-edge id of directed edge 3f+j = (F[f][j], F[f][(j+1)%3])), *d_mean_len float64
 * (mean length of the 3F directed edges), *d_status int32: 0; bit 0 set when some edge is not shared by exactly two faces (the
 * mesh is not watertight, DiffRender.py:305), bit 1 when a face indexes a vertex outside [0, V) (never dereferenced); the tables
 * are then unspecified.  d_workspace: drt_edge_tables_workspace(F)
 * bytes of device memory.  Everything is enqueued on `stream`; nothing synchronises for an even 3F. */
int64_t drt_edge_tables_workspace(int64_t n_faces);
int drt_edge_tables(const int64_t* d_faces, int64_t n_faces, const double* d_verts, int64_t n_verts,
                    void* d_workspace, int64_t* d_edges, int64_t* d_e2f, int32_t* d_row2edge,
                    double* d_mean_len, int32_t* d_status, void* stream);
/* d_verts_out float64 [V+E,3] = the vertices followed by the midpoint of every unique edge (rounded through float32 when
 * round_f32, like a PLY round trip); d_faces_out int64 [4F,3]: face f -> (v0,m01,m20), (m01,v1,m12), (m20,m12,v2),
 * (m01,m12,m20) at rows 4f..4f+3.  d_edges / d_row2edge: from drt_edge_tables of the same mesh. */
int drt_subdivide_midpoint(const int64_t* d_faces, int64_t n_faces, const double* d_verts, int64_t n_verts,
                           const int64_t* d_edges, int64_t n_edges, const int32_t* d_row2edge, int round_f32,
                           int64_t* d_faces_out, double* d_verts_out, void* stream);

/* ---- measurement (bench.py's live per-kernel timing) --------------------------------------------
 * When enabled (on = 1; on = 2 additionally collects the traversal statistics below, which perturbs
 * timing; on = 3 runs the sub-batches of a call one after the other on ONE internal stream instead of two, so that
 * every kernel is timed alone -- the default overlaps the HBM-bound and the latency-bound kernels of two sub-batches,
 * which stretches both) every kernel of the build / forward / backward / fused pipelines is bracketed by a
 * hipEvent pair on the stream it is launched on.  drt_profile_read synchronises that stream and
 * returns, per stage, the summed kernel time in ms, the number of launches and the number of work
 * items (rays in the stage's input queue) since the previous read.  Arrays have DRT_PROFILE_STAGES
 * entries: 0 build, 1 cull, 2 trace1, 3 shade1, 4 trace2, 5 shade2, 6 trace3 (occlusion),
 * 7 finish, 8 collect (backward compaction when no list was saved), 9 backward,
 * 10 fused loss+backward, 11 projected primary visibility (fit + raster kernels), 12 pre-fill of the dense outputs (memsets, DRT_GRID_TRUST),
 * 13 the one-kernel path of small sub-batches (k_path: stages 3-7 in one launch; their rows then only carry item counts).  The event pool grows with the number of launches between two reads; if it could not
 * (allocation failure), drt_profile_read FAILS (DRT_E_INVALID, message with the number of lost timings) instead
 * of returning under-reported stage times. */
#define DRT_PROFILE_STAGES 14
int drt_profile_enable(drt_scene_t* s, int on);
/* Which stages are timed while the profile is on: bit k = stage k of the list above (default: all).  Every timed launch costs
 * two event records on its stream (~1.5 us each inside a step of ~100 launches): a measurement that only needs the traversal
 * kernels' launch times -- bench.py's timed region -- selects those and leaves the rest of the step undisturbed. */
int drt_profile_select(drt_scene_t* s, uint32_t stage_mask);
int drt_profile_read(drt_scene_t* s, double* ms_out, int64_t* launches_out, int64_t* items_out);
/* Traversal diagnostics of the last drt_profile_read interval, 4 values for each of the three
 * k_trace stages: node visits summed over wavefronts ("wave-steps"), over lanes ("lane-steps";
 * lane-steps / (64 * wave-steps) = SIMD lane utilisation), the wave-steps that were leaf (triangle) visits
 * (the others are inner-node visits), longest wavefront. */
int drt_profile_trace_stats(drt_scene_t* s, int64_t* out12);

/* Debug builds (hipcc -DDRT_CHECK=1): the persistent traversal kernels assert the invariants of their bound-check-free LDS stack
 * (csrc/drt_traverse.h FastStack) and count violations on the device.  out4 = {stores above a lane's rows, pops of an empty stack,
 * overwritten guard rows, visits that started with an illegal stack} since the library was loaded (host-synchronising); every entry
 * is -1 in a build without the checks. */
int drt_check_violations(int64_t* out4);

#ifdef __cplusplus
}
#endif
#endif /* DRT_HIP_H */
