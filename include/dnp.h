/*
 * dnp.h - C ABI of libdnp.so: MI355X (gfx950) dipole field / potential evaluator.
 *
 * This is the drop-in boundary for the hot path of crazyMessi/dipole-normal-prop.  The
 * reference has no FFI layer of its own (pure PyTorch), so every entry point below cites
 * the reference *function* (file:line under /root/reference) whose arithmetic it replaces;
 * the Python mirror in dipole_normal_prop_amd/field_utils.py binds these through ctypes
 * (cffi ABI-mode works identically: plain C, no C++ types, no exceptions cross this line).
 *
 * Conventions
 *   - every pointer is a DEVICE pointer on the current HIP device unless stated otherwise;
 *     all buffers are caller-allocated and caller-freed (torch tensors' data_ptr()).
 *   - point clouds are row-major float rows: sources [S, >=6] = (x,y,z,nx,ny,nz) with row
 *     stride ld_src floats, targets [T, >=3] with row stride ld_tgt (only xyz is read).
 *   - *_idx are optional int64 row gathers (NULL = identity): row j of the operand is
 *     base + idx[j]*ld.  This is how the drivers' pts[mask] / E[mask] += ... are expressed
 *     without host-side gathers (field_utils.py:311,330-331).
 *   - work is enqueued on `stream` (a hipStream_t passed as void*; NULL = the null stream);
 *     no entry point synchronises the device or allocates device memory.  Scratch comes
 *     from the caller: ask dnp_*_workspace_bytes() and pass a buffer of at least that size.
 *   - return value: 0 = ok, negative = DNP_E* below; a message for the calling thread is
 *     available from dnp_last_error().  The library never aborts, lets no C++ exception cross this boundary (host-side
 *     allocation failures come back as DNP_ENOMEM; the *_workspace_bytes queries return 0 then) and is re-entrant
 *     (the reference calls field_grad concurrently from Python threads, util.py:187-196).
 */
#ifndef DNP_H
#define DNP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNP_VERSION 503 /* 0.5.3: + dnp_patch_fields_ordered_*; 0.5.2: + dnp_xie_order_blocked_*; 0.5.1: + dnp_xie_knn_*, dnp_xie_pairs_knn_* (0.5.0: the fp64 patch-driver entry points, dnp_xie_order_f64, dnp_xie_rowdots_*) */

enum {
    DNP_OK = 0,
    DNP_EINVAL = -1,    /* bad argument (NULL pointer, negative size, ld too small ...) */
    DNP_ENODEV = -2,    /* no HIP device / not a gfx950 code object for this device */
    DNP_EWORKSPACE = -3,/* workspace missing or too small */
    DNP_EHIP = -4,      /* a HIP runtime call failed; see dnp_last_error() */
    DNP_ENOMEM = -5,    /* a host-side allocation failed (launch planner, cell merge) */
    DNP_EINTERNAL = -6  /* an unexpected C++ exception was caught at the boundary; see dnp_last_error() */
};

/* ---- housekeeping ------------------------------------------------------------------- */
int dnp_version(void);
int dnp_device_count(void);            /* 0 when no HIP device is visible; never fails */
const char* dnp_last_error(void);      /* thread-local, never NULL */

/* ---- K1: dipole field  (replaces field_utils.field_grad, field_utils.py:61-116) -------
 *
 *   E[t] = - sum_s [ 3 (p_s . r^) r^ - p_s ] / (|r|^3 + eps),   r = x_s - x_t,
 *   pairs with |r| == 0 contribute 0/(0+eps) (field_utils.py:99-108).
 *
 * max_pts > 0 reproduces the reference's recursion semantics (field_utils.py:73-94): the
 * source range is halved at int(n/2) until every leaf has <= max_pts rows; each leaf sum has
 * its Inf/NaN components zeroed (field_utils.py:110-115) before the leaves are added.
 * max_pts <= 0 means one leaf (recursive=False).  The target split of the reference only
 * concatenates rows and needs no counterpart.
 *
 * out row for target j is  out + (out_scatter ? tgt_idx[j] : j) * ld_out  (3 floats);
 * accumulate != 0 adds to what is there (E[mask] = E[mask] + dE, field_utils.py:331).
 * nonfinite (device int32[3], may be NULL): [0] += number of Inf, [1] += number of NaN leaf components that
 * were zeroed - what the reference prints as "warning: %d inf/nan in field_grad" (field_utils.py:110-113); [2] is
 * the caller's and is not touched.  nonfinite_host (PINNED host int32[3], may be NULL): the three ints are copied
 * there with an asynchronous copy behind the kernels, so a caller that keeps a non-zero stamp in [2] can tell from
 * the host copy alone when the counters of this call have landed - no event, no synchronisation.
 */
size_t dnp_field_grad_workspace_bytes(int64_t S, int64_t T, int64_t max_pts);

int dnp_field_grad_f32(const float* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                       const float* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                       float eps, int64_t max_pts,
                       float* out, int64_t ld_out, int out_scatter, int accumulate, int32_t* nonfinite,
                       int32_t* nonfinite_host, void* workspace, size_t workspace_bytes, void* stream);

/* fp64 variant: the socket path of the reference feeds float64 clouds (util.py:71-77). */
int dnp_field_grad_f64(const double* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                       const double* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                       double eps, int64_t max_pts,
                       double* out, int64_t ld_out, int out_scatter, int accumulate, int32_t* nonfinite,
                       int32_t* nonfinite_host, void* workspace, size_t workspace_bytes, void* stream);

/* ---- K2: dipole potential  (replaces field_utils.potential, field_utils.py:12-55) ------
 *
 *   phi[t] = sum_s (p_s . r) / |r|^3      (no eps, no zero mask: a coincident pair makes
 *   the leaf sum NaN, which is zeroed after the sum, field_utils.py:53-54)
 * out is [T] with element stride ld_out.
 */
size_t dnp_potential_workspace_bytes(int64_t S, int64_t T, int64_t max_pts);

int dnp_potential_f32(const float* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                      const float* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                      int64_t max_pts, float* out, int64_t ld_out,
                      void* workspace, size_t workspace_bytes, void* stream);

int dnp_potential_f64(const double* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                      const double* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                      int64_t max_pts, double* out, int64_t ld_out,
                      void* workspace, size_t workspace_bytes, void* stream);

/* ---- reference_field  (field_utils.reference_field, field_utils.py:188-201) as ONE call: the field of the oriented
 *      cloud src[S, >=6] at the rows of tgt, and the function's tail fused into the reduction pass:
 *   form 1 (3-column targets, :191-194): out[T, >=6] = (x, y, z, E / |E|), rows with |E| == 0 keep E (zeros);
 *   form 2 (6-column targets, :195-199): tgt[:, 3:6] *= (E . n >= 0 ? +1 : -1) IN PLACE (note `>=`; out is not used).
 * The per-point products are rounded separately and added left to right (no fma contraction), as torch's
 * (E * n).sum(dim=-1).  Workspace as dnp_field_grad (dnp_field_grad_workspace_bytes(S, T, max_pts)).  Returns DNP_EINVAL
 * when the source set needs more than one round of chunks (beyond ~260 000 recursion-leaf sources): the caller then
 * uses dnp_field_grad and finishes itself.  nonfinite as in dnp_field_grad (device int32[3], may be NULL).
 */
int dnp_reference_field_f32(const float* src, int64_t S, int64_t ld_src, float* tgt, int64_t T, int64_t ld_tgt,
                            int form, float eps, int64_t max_pts, float* out, int64_t ld_out, int32_t* nonfinite,
                            void* workspace, size_t workspace_bytes, void* stream);
int dnp_reference_field_f64(const double* src, int64_t S, int64_t ld_src, double* tgt, int64_t T, int64_t ld_tgt,
                            int form, double eps, int64_t max_pts, double* out, int64_t ld_out, int32_t* nonfinite,
                            void* workspace, size_t workspace_bytes, void* stream);

/* ---- batched per-patch fields: the multi-GPU shard unit --------------------------------
 *
 * For patches k in [p_begin, p_end) of a partition of pts[N, >=6] given in CSR form
 * (patch_off[P+1], patch_idx[patch_off[P]], both device int64; patch_idx == NULL means the cloud is
 * already sorted by patch, i.e. patch k is the row range [patch_off[k], patch_off[k+1]) - the layout
 * the drivers use, because it makes every slab / interaction access coalesced):
 *
 *   dE[k - p_begin][t] = field of patch k on point t, for every t NOT in patch k (rows of
 *   patch k itself are written as 0) - i.e. exactly the dE of one greedy step
 *   `field_grad(pts[patch], pts[~patch_mask])` scattered to full length
 *   (field_utils.py:328-331), evaluated with the normals as they are in pts now.
 *
 * Because field_grad is linear in the dipoles and a flip negates a whole patch, the dE of
 * a flipped patch is exactly -dE (IEEE negation commutes with every op of
 * field_utils.py:105-109), so all P slabs can be computed up front, in one launch, in any
 * order and on any GPU.  point_patch[N] (device int64) maps a point to its patch (-1 = in
 * no patch: such points are targets only).  dE is [p_end-p_begin, N, 3] floats.
 */
int dnp_patch_fields_f32(const float* pts, int64_t N, int64_t ld_pts,
                         const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                         const int64_t* point_patch,
                         int64_t p_begin, int64_t p_end, float eps,
                         float* dE, void* stream);

/* The same with the patches' bounding boxes supplied: patch_box[P][6] floats (min x, y, z, max x, y, z) as
 * written by dnp_patch_boxes_f32, or NULL (then exactly dnp_patch_fields_f32).  The kernel decides per
 * (wavefront of 128 targets, patch) whether the whole patch is far enough for the one-transcendental chain;
 * with the boxes given a workgroup no longer scans its patch to find the box itself (1.1 % of a launch at
 * 100 000 points / 256 patches).  Results are bit-identical with and without the table.  The drivers compute
 * the boxes once per cloud.
 */
int dnp_patch_boxes_f32(const float* pts, int64_t N, int64_t ld_pts,
                        const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                        float* boxes, void* stream);
int dnp_patch_fields_boxed_f32(const float* pts, int64_t N, int64_t ld_pts,
                               const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                               const int64_t* point_patch, const float* patch_box,
                               int64_t p_begin, int64_t p_end, float eps,
                               float* dE, void* stream);

/* The drivers' form (round 3): the patch-sorted layout only (patch_idx must be NULL for the tables to be used), with
 *   tile_box[ceil(N / R)][6]  boxes of the target tiles, tile i = rows [i R, (i+1) R), R = dnp_patch_tile_rows() = the 128
 *                             rows one wavefront of the kernel owns (dnp_tile_boxes_f32; NULL = the wavefront finds the
 *                             box of its targets itself, 36 cross-lane steps per (wavefront, patch)), and
 *   w_part[p_end-p_begin][ceil(N / R)][w_slots]  (optional, NULL = not wanted; w_slots = 2 or 3) the interaction sums of
 *     every tile:
 *       w_part[k][i][0] = sum_{t in tile i, patch(t) == g} dE[k][t] . n_t,  g = the patch of the tile's first row
 *       w_slots = 2:  w_part[k][i][1] = the same over the tile's other rows
 *       w_slots = 3:  w_part[k][i][1] = over the rows of patch g + 1,  w_part[k][i][2] = over the remaining rows
 *     (fp32 dot per point as the reference's (E[patch] * pts[patch, 3:]).sum(dim=-1), field_utils.py:316, fp64 sums in a
 *     fixed order).  With every tile inside at most w_slots groups (2: patches of >= R points; 3, round 5: patches of >= R / 2
 *     points - the reference's own grid partitions with their 100-point minimum; rows in no patch last; no empty patch
 *     between two patches of a tile) W follows from dnp_interactions_from_tiles without a second pass over the 12 N P bytes
 *     of slabs; the caller checks that condition (the drivers do, on the host, from the patch sizes;
 *     dnp_check_tile_groups does it on the device) and uses dnp_interactions_f32 otherwise.
 * source_split (1 or -k; -k needs both tables and the exchange buffer): -k (k >= 1) is ONE launch in which the LAST k patches
 * of the range (k >= the range: all of them) are evaluated as SPLIT ITEMS - four wavefronts per target tile, wavefront i
 * on the patch's i-th run of 128 sources; every run term is written to `exchange` with write-through stores, an arrival
 * counter per (patch, tile) tells the wavefront that arrives last, and that one adds the terms in run order and finishes the
 * tile.  No LDS and no barrier: the other patches of the launch run exactly as in the plain form, the launch's last
 * resident set consists of items a third as long and the chip drains in a third of the time.  The drivers use k = 3 for
 * launches below 8 10^9 pairs (profiles/r04_xch_ab.txt: -11 % at 16 of the bench's patches, -3.7 % at 32, -0.5 % at 128).
 * dE and w_part do not depend on source_split (the same fp32 runs, the same fp64 additions in run order); a split patch of
 * more than 512 points is evaluated by one wavefront per tile whatever source_split says, patches of <= 128 points are a
 * single run - the drivers therefore size the tail by its sources, not by a patch count, and only use it when even the shortest
 * patches of the launch are long (patch_drivers._launch_plan / _pick_source_split: longest patch first - see
 * dnp_patch_fields_ordered_f32 -, then the fewest last rows holding 1000 points, never across a patch of more than 512 points);
 * without both tables a launch with source_split < 0 is the plain one.  (An eight-wavefront item for patches of 513..1024 points was built and measured in round 5 and not kept:
 * profiles/r05_xch_eight_wavefronts.patch.)
 * exchange: device scratch of at least dnp_patch_exchange_bytes(N, k) bytes = k * ceil(N / 128) * 12 416 (a 128-byte counter
 * line + 4 run slots x 6 doubles x 64 lanes per split (patch, tile) item; NULL / 0 with source_split = 1).  CONTRACT: the
 * buffer is zero before its first use (dnp_exchange_init, or any memset); every launch leaves its arrival counters zero
 * again, and a record's place in the buffer depends on its (patch, tile) index only, so one buffer serves launches of any
 * size one after the other - on ONE stream at a time.  A launch with source_split < 0 and no (or too small a) buffer
 * returns DNP_EWORKSPACE.
 * dE is bit-identical with dnp_patch_fields_boxed_f32's; w_part requires eps >= 1e-30 and both tables.
 */
int64_t dnp_patch_tile_rows(void);
int dnp_tile_boxes_f32(const float* pts, int64_t N, int64_t ld_pts, int64_t rows_per_tile, float* boxes, void* stream);
int dnp_patch_fields_tiled_f32(const float* pts, int64_t N, int64_t ld_pts,
                               const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                               const int64_t* point_patch, const float* patch_box, const float* tile_box,
                               int64_t p_begin, int64_t p_end, float eps,
                               float* dE, double* w_part, int w_slots, int source_split,
                               void* exchange, size_t exchange_bytes, void* stream);
/* The same slabs for a FLOAT64 cloud (round 5).  The reference computes in the dtype it is handed (field_utils.py:96-109; its
 * socket path hands it float64, util.py:71-77), so on a float64 cloud the greedy drivers' per-patch fields (field_utils.py:
 * 328-331, :258-265), interaction sums (:316, :244) and diffuse field are double precision here too.  dE is [p_end - p_begin, N, 3]
 * doubles; w_part as above (the per-point dot taken in fp64), same precondition (dnp_check_tile_groups), tiles of
 * dnp_patch_tile_rows() rows.  The scalar-unit kernel on the patch-sorted layout (patch_idx == NULL, eps > 0), the LDS kernel
 * otherwise (w_part must be NULL then).  patch_box / tile_box (both or neither; [P][6] / [ceil(N / R)][6] doubles from
 * dnp_patch_boxes_f64 / dnp_tile_boxes_f64; NULL = every pair through the exact chain): a (wavefront, patch) whose boxes are farther
 * apart than (eps / 6e-4)^(1/3) runs the fp64 far chain - 1 / (|r|^3 + eps) = u^3 (1 - e + e^2 - e^3 + e^4), e = eps u^3 < 6e-4,
 * truncation < 7.8e-17: one transcendental instead of two; results agree with the exact chain to fp64 rounding.  No split tail
 * in this precision.  Within ~1e-15 of the reference's float64 arithmetic per pair (refined v_rsq_f64 / v_rcp_f64). */
int dnp_patch_boxes_f64(const double* pts, int64_t N, int64_t ld_pts,
                        const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                        double* boxes, void* stream);
int dnp_tile_boxes_f64(const double* pts, int64_t N, int64_t ld_pts, int64_t rows_per_tile, double* boxes, void* stream);
int dnp_patch_fields_tiled_f64(const double* pts, int64_t N, int64_t ld_pts,
                               const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                               const int64_t* point_patch, const double* patch_box, const double* tile_box,
                               int64_t p_begin, int64_t p_end, double eps,
                               double* dE, double* w_part, int w_slots, void* stream);
/* The same launches on the patch-sorted layout (patch_idx == NULL, eps > 0) with a LAUNCH ORDER (round 5): patch_order[i] (device
 * int32, a permutation of 0 .. p_end - p_begin - 1; NULL = identity; at most 65535 patches) names the patch, relative to p_begin,
 * that launch row i evaluates.  Workgroups are dispatched in row order, so the last rows decide how the launch drains: the drivers
 * put the LONGEST patches first (a stable sort by descending size - longest-processing-time-first).  On partitions with uneven
 * patches that is worth more than the split tail: one rank's share of eight on the reference's grid partition of the bench sphere
 * (patches of 100..677 points) 0.914-0.951 of ideal over the eight ranks against 0.835-0.923, on boxunion's 369 representatives'
 * patches 0.936-0.956 against 0.91-0.94 (profiles/r05_tail_sweep.txt).  dE, w_part and the exchange contract are exactly those of
 * dnp_patch_fields_tiled_*: slab k is patch p_begin + k whatever the order; with source_split = -k the split items are the last k
 * ROWS of the launch, i.e. with the drivers' order its k smallest patches. */
int dnp_patch_fields_ordered_f32(const float* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off, int64_t P,
                                 const int64_t* point_patch, const float* patch_box, const float* tile_box,
                                 int64_t p_begin, int64_t p_end, const int32_t* patch_order, float eps,
                                 float* dE, double* w_part, int w_slots, int source_split,
                                 void* exchange, size_t exchange_bytes, void* stream);
int dnp_patch_fields_ordered_f64(const double* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off, int64_t P,
                                 const int64_t* point_patch, const double* patch_box, const double* tile_box,
                                 int64_t p_begin, int64_t p_end, const int32_t* patch_order, double eps,
                                 double* dE, double* w_part, int w_slots, void* stream);
size_t dnp_patch_exchange_bytes(int64_t N, int64_t split_patches);
int dnp_exchange_init(void* exchange, size_t bytes, void* stream);
/* The precondition of w_part, checked on the device for callers that cannot check it from patch sizes on the host (the
 * Python drivers do that: patch_drivers._tile_group_slots): violations[0] (a device int32, NOT cleared here) += the number of
 * target tiles of dnp_patch_tile_rows() rows that break it for this w_slots - more than two values of point_patch with
 * w_slots = 2; with w_slots = 3 more than three, or a third value while the second is not the first + 1.  Nonzero means w_part
 * would be wrong for those tiles - use dnp_interactions_f32 / _f64 on the slabs instead.  No synchronisation: read the counter
 * behind the stream.  (A -DDNP_BOUNDS build of the library checks the same inside the pair kernel.) */
int dnp_check_tile_groups(const int64_t* point_patch, int64_t N, int w_slots, int32_t* violations, void* stream);
/* W[k][j] = sum of w_part[k][i][slot] over the tiles i that overlap patch j (slot 0 when j is the patch of the tile's
 * first row; otherwise slot 1 with w_slots = 2, and with w_slots = 3 slot 1 when j is that patch + 1, else slot 2), in tile
 * order; W is [K, P] doubles.  Same quantity as dnp_interactions_f32 / _f64 up to fp64 reassociation. */
int dnp_interactions_from_tiles(const double* w_part, int w_slots, int64_t K, int64_t N, const int64_t* point_patch,
                                const int64_t* patch_off, int64_t P, double* W, void* stream);

/* ---- K3: patch interaction matrix -----------------------------------------------------
 *
 *   W[k][j] = sum_{t in patch j} dE[k][t] . n_t          (double accumulation)
 * the per-step interaction list of the greedy drivers (field_utils.py:316, :244) becomes
 * I_j = sum_{k visited} sigma_k W[k][j].  K = number of slabs in dE; W is [K, P] doubles.
 */
int dnp_interactions_f32(const float* dE, int64_t K, int64_t N,
                         const float* pts, int64_t ld_pts,
                         const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                         double* W, void* stream);
/* float64 slabs and cloud: the per-point dot in fp64 (the reference's (E[patch] * pts[patch, 3:]).sum(dim=-1) on a float64 cloud) */
int dnp_interactions_f64(const double* dE, int64_t K, int64_t N,
                         const double* pts, int64_t ld_pts,
                         const int64_t* patch_off, const int64_t* patch_idx, int64_t P,
                         double* W, void* stream);

/* ---- ordered combination of slabs ------------------------------------------------------
 *
 *   E[t] = sum_{i=0..n-1} coef[i] * dE[slab[i]][t]      (fp32, in the order given - the
 *   same order in which the reference accumulates E = E + dE, field_utils.py:331)
 * coef/slab are device arrays of length n (float / int64).  accumulate != 0 adds to E.
 */
int dnp_combine_fields_f32(const float* dE, int64_t K, int64_t N,
                           const float* coef, const int64_t* slab, int64_t n,
                           float* E, int accumulate, void* stream);

/* ---- K4: per-point greedy propagation  (field_utils.strongest_field_propagation_points,
 *      field_utils.py:353-388) as ONE persistent launch.
 *
 * pts[N, >=6] normals are flipped in place (the kernels write the oriented normals to scratch and a
 * stream-ordered second kernel copies them into pts, so no workgroup ever reads a row another one is
 * rewriting); order_out[N] (device int64, may be NULL) receives the visit order (order_out[0] = start).
 * E_out [N,3] (may be NULL) receives the accumulated field.  diffuse != 0 applies the final per-point sign
 * pass (:382-385).  _f64 is the same propagation in double precision: the reference's socket path hands
 * float64 clouds to this driver (util.py:71-77, socket_server.py:18-27).
 *
 * form: 0 = choose by N, 1 = single workgroup (N <= 512*20 fp32 / 512*8 fp64), 2 = one workgroup per CU
 * (256 threads per workgroup, <= 20 points per thread in fp32, <= 8 in fp64: N <= CUs x 5120 / CUs x 2048;
 * co-residency checked against the occupancy query; N < 2^20), 3 = form 2 with the time-out raised before the
 * launch (test hook for the abort path).  max_groups > 0 caps the workgroup count of form 2.  The first int of the
 * workspace is a status word: non-zero after the launch means a workgroup of form 2 gave up waiting for its
 * peers (GPU shared with another process); the copy kernel then stores nothing, so pts is bit for bit the caller's
 * input (order_out / E_out hold a partial run) and the caller should fall back to step-wise launches of
 * dnp_field_grad.
 */
size_t dnp_point_greedy_workspace_bytes(int64_t N, int elem_size /* 4 or 8 */);
int dnp_point_greedy_max_points(void);   /* capacity of the persistent forms: N < this */

int dnp_point_greedy_f32(float* pts, int64_t N, int64_t ld_pts, int64_t start, float eps,
                         int diffuse, int64_t* order_out, float* E_out, int form, int max_groups,
                         void* workspace, size_t workspace_bytes, void* stream);
int dnp_point_greedy_f64(double* pts, int64_t N, int64_t ld_pts, int64_t start, double eps,
                         int diffuse, int64_t* order_out, double* E_out, int form, int max_groups,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- per-patch PCA  (util.pca_eigen_values util.py:495-500; the start-patch rule
 *      field_utils.py:230-233 / :303-306; inference_utils.fix_n_filter :52-71; util.orient_center :39-44)
 *
 * For every patch of the CSR partition (patch_idx == NULL: contiguous row ranges): mean[P,3], the
 * eigenvalues of cov = (x - mean)^T (x - mean) / n in ascending order evals[P,3] and the eigenvectors
 * evecs[P,3,3] (evecs[p][c][k] = component c of eigenvector k, i.e. torch.linalg.eigh's layout), all in
 * fp64 from the fp32/fp64 coordinates, deterministic (fixed reduction tree, no atomics).  An eigenvector's
 * sign is arbitrary in the reference (LAPACK's choice); here its largest-magnitude component is positive.
 * One workgroup per patch.
 */
int dnp_patch_pca_f32(const float* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                      int64_t P, double* mean, double* evals, double* evecs, void* stream);
int dnp_patch_pca_f64(const double* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                      int64_t P, double* mean, double* evals, double* evecs, void* stream);

/* ---- the greedy loop on the interaction matrix  (field_utils.py:314-324, :242-254) -----------------
 *
 * With W from dnp_interactions_f32 (all P rows):  I_j = sum_{k visited} sigma_k W[k][j]; every step takes the
 * first maximum of |I_j| over the unvisited patches in patch order (torch.argmax over the reference's
 * `remaining` list), sets sigma_j = -1 when I_j < 0 and adds sigma_j W[j] to I.  start is a DEVICE int64
 * (so that a start patch chosen on the device needs no host round trip).  Outputs (device): order[P],
 * sigma[P] (+-1.0), chosen[P-1] (the signed interaction of each chosen patch).  One wavefront up to 2048
 * patches, one workgroup above; P <= dnp_patch_greedy_max_patches() (16384).  A start outside [0, P) is CLAMPED to
 * patch 0 (the value lives on the device and is not read back; the Python drivers range-check a host integer start
 * themselves and raise IndexError).
 */
int dnp_patch_greedy_max_patches(void);
int dnp_patch_greedy(const double* W, int64_t P, const int64_t* start, int64_t* order, double* sigma,
                     double* chosen, void* stream);

/* E[t] (+)= sum_{k=0..K-1} sigma[p_lo + k] * dE[k][t]  for the K slabs held here (patches p_lo .. p_lo+K-1),
 * accumulated in DOUBLE in slab order (E is [N,3] doubles): the diffuse field of the batched drivers
 * (field_utils.py:330-331) from the device outputs of dnp_patch_greedy.  sigma is +-1, so the products are
 * exact and the fp64 sum does not depend on the visit order or on how the patches are split over GPUs (the
 * reference's own fp32 chain E = E + dE in visit order is dnp_combine_fields_f32). */
int dnp_combine_signed_f32(const float* dE, int64_t K, int64_t N, const double* sigma, int64_t P, int64_t p_lo,
                           double* E, int accumulate, void* stream);
/* the same for float64 slabs (dnp_patch_fields_tiled_f64): the reference's E = E + dE chain in the visit order differs from this
 * slab-order sum by fp64 reassociation only (~1e-16 relative) */
int dnp_combine_signed_f64(const double* dE, int64_t K, int64_t N, const double* sigma, int64_t P, int64_t p_lo,
                           double* E, int accumulate, void* stream);

/* ---- patch-sorted working layout  (the set-up of the batched drivers) -----------------------------------------
 *
 * swork[i] = pts[patch_idx[i]] (6 floats per row) and sorted_patch[i] = p for the rows i in
 * [patch_off[p], patch_off[p+1]) of every patch p: the cloud sorted by patch, in which a patch is a contiguous row
 * range (what dnp_patch_fields_f32 with patch_idx == NULL expects).  swork is [patch_off[P], 6] floats.
 */
int dnp_patch_layout_f32(const float* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                         int64_t P, float* swork, int64_t* sorted_patch, void* stream);
int dnp_patch_layout_f64(const double* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                         int64_t P, double* swork, int64_t* sorted_patch, void* stream);

/* ---- tail of the batched patch drivers in one launch (field_utils.py:322-323, :337-342, :344-346) ----------
 *
 * On the patch-sorted working cloud work[N, >=6] (fp32, normals possibly weight-scaled), for sorted row t:
 *   n = work[t].n * sigma[point_patch[t]]                       (point_patch[t] < 0: in no patch, not flipped)
 *   E64 != NULL and (listed == NULL or listed[patch]):  n *= ((float)E64[t] . n > 0) ? +1 : -1   (the diffuse pass)
 *   weights != NULL:  n /= weights[t]                                                  (sorted order, clamped)
 *   out[perm ? perm[t] : t][3..5] = n     out is the caller's cloud (float or double rows of stride ld_out)
 * E64 is [N,3] doubles in sorted order (dnp_combine_signed_f32), listed [P] bytes, perm [N] the sorted -> caller
 * row map.  Everything on the device.
 */
int dnp_patch_finish_f32(const float* work, int64_t ld_work, int64_t N, const int64_t* point_patch,
                         const double* sigma, const double* E64, const unsigned char* listed, const float* weights,
                         const int64_t* perm, void* out, int64_t ld_out, int out_is_f64, void* stream);
/* float64 working cloud and weights: the field is used as held (doubles), dot, sign and un-scaling in fp64 */
int dnp_patch_finish_f64(const double* work, int64_t ld_work, int64_t N, const int64_t* point_patch,
                         const double* sigma, const double* E64, const unsigned char* listed, const double* weights,
                         const int64_t* perm, void* out, int64_t ld_out, int out_is_f64, void* stream);

/* ---- tail of the representatives driver for the non-representative points, in one launch
 *      (field_utils.strongest_field_propagation_reps: :251-252 a flipped patch flips its rest points, :273-276 every
 *      non-representative point takes the sign of the field of all representatives) ----------------------------------
 *
 * rest_off[P+1] / rest_idx: CSR of the patches' rest points (rows of work), every point listed once; sigma[P] = +-1 from
 * dnp_patch_greedy; E[M,3] = the field of the representatives at the rest points, row i for rest_idx[i] (dnp_field_grad_*
 * with tgt_idx = rest_idx).  In place on work[., 3..5]:  n = n * sigma[p];  n *= (E[i] . n > 0) ? +1 : -1  (the dot's products
 * rounded separately and added in order, as torch's (E * n).sum(dim=-1)).
 */
int dnp_rest_finish_f32(float* work, int64_t ld_work, const int64_t* rest_off, const int64_t* rest_idx, int64_t P,
                        const double* sigma, const float* E, void* stream);
int dnp_rest_finish_f64(double* work, int64_t ld_work, const int64_t* rest_off, const int64_t* rest_idx, int64_t P,
                        const double* sigma, const double* E, void* stream);

/* ---- merge of small voxel cells  (util.merge_nodes, util.py:448-492) - HOST function, host pointers ----
 *
 * cell_ijk[C,3] (int32 voxel coordinates in [0, 2^20)) and cell_size[C] describe the non-empty cells of the
 * voxel partition in the reference's (i, j, k)-lexicographic order.  Up to 10 sweeps over the cells in
 * order; a cell with fewer than min_patch points is appended to the LAST (highest index) other live cell
 * that has a voxel in the 26-neighbourhood of one of its voxels (the reference's find_dij keeps overwriting
 * its result, so the last match wins); cells still below min_patch at the end are dropped.  Output:
 * seq[C] = original cell ids grouped by surviving patch in concatenation order, seq_off[n_out + 1] the
 * group boundaries (room for C + 1 entries), *sweeps_out the number of sweeps made (10 = the reference
 * prints "recursive merge failed to merge some patches").  Sequential by definition; no device work.
 */
int dnp_merge_cells(const int32_t* cell_ijk, const int64_t* cell_size, int64_t C, int64_t min_patch,
                    int64_t* seq, int64_t* seq_off, int64_t* n_out, int32_t* sweeps_out);

/* ---- the fork's "xie" pair functions -----------------------------------------
 *
 * dnp_xie_pairs: per-pair reflected normal  ref[t][s] = (n_s - C (n_s . r^) r^) / |r|^3,  r = x_s - x_t,
 * left undivided when |r| == 0  (field_utils.xie_field, field_utils.py:431-469; its eps argument is unused
 * there).  vector_out != 0: out is [T, S, 3] = ref;  vector_out == 0: out is [T, S] = ref . n_t with NaN/Inf
 * zeroed (field_utils.xie_intersaction, field_utils.py:509-519).  Sources and targets are [*, >=6] rows.
 *
 * dnp_xie_knn + dnp_xie_pairs_knn: the knn_mask > 0 forms (field_utils.py:451-460, 467-468; the reference builds a scipy KDTree
 * on the targets, queries it with the sources and multiplies ref by the resulting 0/1 mask).  dnp_xie_knn gives, per source, the
 * squared distance kth_d2[s] (fp64 on the exact coordinates, as the tree computes) and the index kth_idx[s] of its k-th nearest
 * target, 1 <= k <= T, ties at equal distance to the lower index; dnp_xie_pairs_knn is dnp_xie_pairs with every entry whose
 * target is not among those k - (d2, t) > (kth_d2[s], kth_idx[s]) - multiplied by 0.  No T x S mask is ever stored.
 *
 * dnp_xie_order: the ordered propagation loop of field_utils.xie_propagation_points_in_order
 * (field_utils.py:590-595) for R visiting orders over an N x N interaction matrix M (row = receiving point):
 *   for i in 0..N-1:  idx = order[r][i];  inter[r][idx] = sum_j M[idx][j] * w[r][j];
 *                     w[r][idx] = inter[r][idx] < 0 ? -1 : +1            (w starts at 0)
 * weights / inter are [R, N] outputs in M's precision (the buffers need no initialisation: entries of points an order row
 * never visits - a row that repeats an index - come back 0, as the reference's torch.zeros).  _f64: float64 matrices (a
 * float64 cloud, field_utils.py:581-595 computes in pts.dtype).  Products are rounded in M's precision, row sums run in fp64.
 *
 * dnp_xie_order_blocked (round 5): the same loop evaluated in blocks of 256 steps - per block one HBM-bound launch sums each step's
 * matrix row against the weights decided before the block (and gathers the block's own 256 x 256 corner), then one wavefront per
 * visiting order runs the 256 dependent steps on registers.  Same products, fp64 sums in another order: `inter` agrees with
 * dnp_xie_order to fp64 rounding.  Order rows that are not permutations of 0..N-1 are detected on the device and evaluated by
 * dnp_xie_order's kernels (identical results for them); below 512 points the call IS dnp_xie_order.  workspace:
 * dnp_xie_order_workspace_bytes(N, R, sizeof element) bytes of device memory, contents irrelevant.
 *
 * dnp_xie_rowdots: the diffuse pass behind that loop (field_utils.py:597-603): out[r][i] = sum_j M[i][j] * w[r][j] for the R
 * weight vectors in one pass over M ([R, N] in, [R, N] out; one wavefront per matrix row, HBM-bound).
 */
int dnp_xie_pairs_f32(const float* src, int64_t S, int64_t ld_src, const float* tgt, int64_t T, int64_t ld_tgt,
                      float C, int vector_out, float* out, void* stream);
int dnp_xie_pairs_f64(const double* src, int64_t S, int64_t ld_src, const double* tgt, int64_t T, int64_t ld_tgt,
                      double C, int vector_out, double* out, void* stream);
int dnp_xie_knn_f32(const float* src, int64_t S, int64_t ld_src, const float* tgt, int64_t T, int64_t ld_tgt, int64_t k,
                    double* kth_d2, int64_t* kth_idx, void* stream);
int dnp_xie_knn_f64(const double* src, int64_t S, int64_t ld_src, const double* tgt, int64_t T, int64_t ld_tgt, int64_t k,
                    double* kth_d2, int64_t* kth_idx, void* stream);
int dnp_xie_pairs_knn_f32(const float* src, int64_t S, int64_t ld_src, const float* tgt, int64_t T, int64_t ld_tgt,
                          float C, int vector_out, const double* kth_d2, const int64_t* kth_idx, float* out, void* stream);
int dnp_xie_pairs_knn_f64(const double* src, int64_t S, int64_t ld_src, const double* tgt, int64_t T, int64_t ld_tgt,
                          double C, int vector_out, const double* kth_d2, const int64_t* kth_idx, double* out, void* stream);
int dnp_xie_order_f32(const float* M, int64_t N, const int64_t* order, int64_t R, float* weights, float* inter,
                      void* stream);
int dnp_xie_order_f64(const double* M, int64_t N, const int64_t* order, int64_t R, double* weights, double* inter,
                      void* stream);
size_t dnp_xie_order_workspace_bytes(int64_t N, int64_t R, int elem_bytes);
int dnp_xie_order_blocked_f32(const float* M, int64_t N, const int64_t* order, int64_t R, float* weights, float* inter,
                              void* workspace, size_t workspace_bytes, void* stream);
int dnp_xie_order_blocked_f64(const double* M, int64_t N, const int64_t* order, int64_t R, double* weights, double* inter,
                              void* workspace, size_t workspace_bytes, void* stream);
int dnp_xie_rowdots_f32(const float* M, int64_t N, const float* weights, int64_t R, float* out, void* stream);
int dnp_xie_rowdots_f64(const double* M, int64_t N, const double* weights, int64_t R, double* out, void* stream);

/* ---- '.xyz' text  (util.export_pc util.py:46-51, util.xyz2tensor util.py:53-69) - HOST functions, host pointers ---
 *
 * dnp_xyz_format_f32: rows[n_rows, n_cols] (contiguous floats) -> n_rows lines of n_cols numbers, each written as
 * Python's str(float(v)) (shortest round-trip digits of the float32 taken as a double; fixed notation for
 * 1e-4 <= |v| < 1e16, "1e-05" style otherwise), joined by ' ' and '\n', no trailing newline - byte for byte what
 * the reference writes.  out must hold dnp_xyz_format_bound(n_rows, n_cols) bytes; returns the byte count.
 *
 * dnp_xyz_parse_f32: text whose non-blank lines all hold the same number (3 or 6) of single-space separated
 * numbers -> out[rows, *ncol] (each token parsed as a double and rounded to float, as float(c) + torch.tensor(...,
 * float32) do).  Returns the row count, or -2 when the text is not of that regular form (a "nan" token, double
 * spaces, ragged lines, ...: the caller then takes the line-by-line path that defines the semantics);
 * DNP_EWORKSPACE when the text holds more than max_rows rows.
 *
 * Both work on up to 8 host threads for long inputs (blocks of rows / of whole lines, results laid end to end: the
 * bytes, the rows and the verdict of a single pass); they are reentrant and touch no device state.
 */
int64_t dnp_xyz_format_bound(int64_t n_rows, int64_t n_cols);
int64_t dnp_xyz_format_f32(const float* rows, int64_t n_rows, int64_t n_cols, char* out, int64_t cap);
int64_t dnp_xyz_parse_f32(const char* txt, int64_t len, float* out, int64_t max_rows, int32_t* ncol);

#ifdef __cplusplus
}
#endif
#endif /* DNP_H */
