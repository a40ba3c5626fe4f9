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
 *     available from dnp_last_error().  The library never aborts and is re-entrant
 *     (the reference calls field_grad concurrently from Python threads, util.py:187-196).
 */
#ifndef DNP_H
#define DNP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DNP_VERSION 100 /* 0.1.0 */

enum {
    DNP_OK = 0,
    DNP_EINVAL = -1,    /* bad argument (NULL pointer, negative size, ld too small ...) */
    DNP_ENODEV = -2,    /* no HIP device / not a gfx950 code object for this device */
    DNP_EWORKSPACE = -3,/* workspace missing or too small */
    DNP_EHIP = -4       /* a HIP runtime call failed; see dnp_last_error() */
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
 */
size_t dnp_field_grad_workspace_bytes(int64_t S, int64_t T, int64_t max_pts);

int dnp_field_grad_f32(const float* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                       const float* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                       float eps, int64_t max_pts,
                       float* out, int64_t ld_out, int out_scatter, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);

/* fp64 variant: the socket path of the reference feeds float64 clouds (util.py:71-77). */
int dnp_field_grad_f64(const double* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                       const double* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                       double eps, int64_t max_pts,
                       double* out, int64_t ld_out, int out_scatter, int accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);

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
 * pts[N, >=6] normals are flipped in place; order_out[N] (device int64, may be NULL)
 * receives the visit order (order_out[0] = start).  E_out [N,3] (may be NULL) receives the
 * accumulated field.  diffuse != 0 applies the final per-point sign pass (:382-385).
 */
size_t dnp_point_greedy_workspace_bytes(int64_t N);
int dnp_point_greedy_max_points(void);   /* capacity of the single-workgroup persistent form */

int dnp_point_greedy_f32(float* pts, int64_t N, int64_t ld_pts, int64_t start, float eps,
                         int diffuse, int64_t* order_out, float* E_out,
                         void* workspace, size_t workspace_bytes, void* stream);

/* ---- the fork's "xie" pair functions -----------------------------------------
 *
 * dnp_xie_pairs: per-pair reflected normal  ref[t][s] = (n_s - C (n_s . r^) r^) / |r|^3,  r = x_s - x_t,
 * left undivided when |r| == 0  (field_utils.xie_field, field_utils.py:431-469; its eps argument is unused
 * there).  vector_out != 0: out is [T, S, 3] = ref;  vector_out == 0: out is [T, S] = ref . n_t with NaN/Inf
 * zeroed (field_utils.xie_intersaction, field_utils.py:509-519).  Sources and targets are [*, >=6] rows.
 *
 * dnp_xie_order: the ordered propagation loop of field_utils.xie_propagation_points_in_order
 * (field_utils.py:590-595) for R visiting orders over an N x N interaction matrix M (row = receiving point):
 *   for i in 0..N-1:  idx = order[r][i];  inter[r][idx] = sum_j M[idx][j] * w[r][j];
 *                     w[r][idx] = inter[r][idx] < 0 ? -1 : +1            (w starts at 0)
 * weights / inter are [R, N] float outputs.
 */
int dnp_xie_pairs_f32(const float* src, int64_t S, int64_t ld_src, const float* tgt, int64_t T, int64_t ld_tgt,
                      float C, int vector_out, float* out, void* stream);
int dnp_xie_pairs_f64(const double* src, int64_t S, int64_t ld_src, const double* tgt, int64_t T, int64_t ld_tgt,
                      double C, int vector_out, double* out, void* stream);
int dnp_xie_order_f32(const float* M, int64_t N, const int64_t* order, int64_t R, float* weights, float* inter,
                      void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DNP_H */
