// dnp_patch.hip - batched per-patch fields (the multi-GPU shard unit), the patch interaction
// matrix (K3) and the ordered slab combination used by the greedy patch drivers
// (field_utils.strongest_field_propagation{,_reps}, field_utils.py:207-348).
#include <math.h>

#include "dnp_common.h"
#include "pair_kernel.h"

namespace dnp {

#ifndef DNP_KT
#define DNP_KT 4
#endif
constexpr int kPatchKT = DNP_KT;          // LDS kernel (gathered patches)
constexpr int kPatchScalarKT = 2;         // scalar kernel (patch-sorted cloud): 57 VGPRs, 8 waves per SIMD
// far-field (one transcendental) chain for (wave, patch) combinations whose boxes are far apart; the drivers sort
// the cloud by patch, so a wave's 128 consecutive targets and a patch's sources are both spatially compact
#ifndef DNP_FAR
#define DNP_FAR 1
#endif
constexpr bool kPatchFar = DNP_FAR != 0;
#ifndef DNP_TABLED_WAVES     // wavefronts per workgroup of the tabled scalar kernel (A/B builds: 1, 2, 4)
#define DNP_TABLED_WAVES 2
#endif
constexpr int kTabledWaves = DNP_TABLED_WAVES;
#ifndef DNP_XCH_WAVES        // wavefronts per workgroup of the launches with a split tail (A/B builds: 1, 2, 4)
#define DNP_XCH_WAVES 4
#endif
constexpr int kXchWaves = DNP_XCH_WAVES;
#ifndef DNP_FORCE_LDS   // A/B builds only (tools/gpu_ab_far.py): 1 sends the sorted layout through the LDS kernel too
#define DNP_FORCE_LDS 0
#endif

// W[k][j] = sum_{t in patch j} dE[k][t] . n_t    - one workgroup per (j, k), fp64 tree reduce.  F = the precision of the
// slabs and the cloud: the per-point dot is taken in it (float: the reference on a float32 cloud; double: on a float64 one).
template <typename F>
__global__ __launch_bounds__(256) void interactions_kernel(const F* __restrict__ dE, int64_t N,
                                                           const F* __restrict__ pts, int64_t ld_pts,
                                                           const int64_t* __restrict__ patch_off,
                                                           const int64_t* __restrict__ patch_idx, int64_t P,
                                                           double* __restrict__ W) {
    // newest slab first: the pair kernel wrote the slabs in patch order and the last ~256 MB of them are still in the
    // memory-side cache - walking them oldest-first would evict exactly the lines about to be read
    const int64_t j = blockIdx.x, k = (int64_t)gridDim.y - 1 - blockIdx.y;
    const int64_t lo = patch_off[j], hi = patch_off[j + 1];
    const F* slab = dE + k * N * 3;
    double s = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += blockDim.x) {
        const int64_t t = patch_idx ? patch_idx[i] : i;   // nullptr: patches are contiguous row ranges
        const F* e = slab + t * 3;
        const F* n = pts + t * ld_pts + 3;
        // per-point dot in the cloud's precision like (E[patch] * pts[patch, 3:]).sum(dim=-1) - products rounded separately,
        // added left to right, no fma contraction (the same statement as the pair kernel's epilogue) -, patch sum in fp64
        F d;
        {
#pragma clang fp contract(off)
            d = (e[0] * n[0] + e[1] * n[1]) + e[2] * n[2];
        }
        s += (double)d;
    }
    // wave reduce (64 lanes), then across the 4 waves through LDS
    for (int off = 32; off > 0; off >>= 1) s += __shfl_down(s, off, 64);
    __shared__ double part[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (lane == 0) part[wave] = s;
    __syncthreads();
    if (threadIdx.x == 0) W[k * P + j] = (part[0] + part[1]) + (part[2] + part[3]);
}

// E[t][c] (+)= sum_i coef[i] * dE[slab[i]][t][c], sequentially in fp32 (visit order).
__global__ __launch_bounds__(256) void combine_kernel(const float* __restrict__ dE, int64_t N3,
                                                      const float* __restrict__ coef,
                                                      const int64_t* __restrict__ slab, int64_t n,
                                                      float* __restrict__ E, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N3) return;
    float e = accumulate ? E[i] : 0.f;
    for (int64_t q = 0; q < n; ++q) {
        const float c = coef[q];
        const float v = dE[slab[q] * N3 + i];
        e = e + c * v;   // coef is +-1 (or 0): c*v is exact, so this is the reference's E + dE
    }
    E[i] = e;
}

}  // namespace dnp

using namespace dnp;

namespace dnp {
// bounding boxes of the patches' points: boxes[p] = (min x, y, z, max x, y, z) - one workgroup per patch
template <typename F>
__global__ __launch_bounds__(256) void patch_box_kernel(const F* __restrict__ pts, int64_t ld,
                                                        const int64_t* __restrict__ off,
                                                        const int64_t* __restrict__ idx, F* __restrict__ boxes) {
    __shared__ F red[4][6];
    const int64_t p = blockIdx.x;
    F lo[3] = {F(3.0e38), F(3.0e38), F(3.0e38)}, hi[3] = {F(-3.0e38), F(-3.0e38), F(-3.0e38)};
    for (int64_t q = off[p] + threadIdx.x; q < off[p + 1]; q += 256) {
        const F* r = pts + (idx ? idx[q] : q) * ld;
        for (int c = 0; c < 3; ++c) { lo[c] = r[c] < lo[c] ? r[c] : lo[c]; hi[c] = r[c] > hi[c] ? r[c] : hi[c]; }
    }
    for (int c = 0; c < 3; ++c) { lo[c] = wave_min<F>(lo[c]); hi[c] = wave_max<F>(hi[c]); }
    if ((threadIdx.x & 63) == 0)
        for (int c = 0; c < 3; ++c) { red[threadIdx.x >> 6][c] = lo[c]; red[threadIdx.x >> 6][3 + c] = hi[c]; }
    __syncthreads();
    if (threadIdx.x < 3) {
        const int c = threadIdx.x;
        F l = red[0][c], h = red[0][3 + c];
        for (int w = 1; w < 4; ++w) { l = red[w][c] < l ? red[w][c] : l; h = red[w][3 + c] > h ? red[w][3 + c] : h; }
        boxes[p * 6 + c] = l;
        boxes[p * 6 + 3 + c] = h;
    }
}

// boxes of the target tiles: tile i = rows [i * rows_per_tile, (i + 1) * rows_per_tile) - what one wavefront of the
// scalar-unit pair kernel owns (64 KT = 128 rows); one wavefront per tile
template <typename F>
__global__ __launch_bounds__(256) void tile_box_kernel(const F* __restrict__ pts, int64_t ld, int64_t N,
                                                       int rows_per_tile, int64_t n_tiles, F* __restrict__ boxes) {
    const int64_t tile = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= n_tiles) return;
    const int lane = threadIdx.x & 63;
    F lo[3] = {F(3.0e38), F(3.0e38), F(3.0e38)}, hi[3] = {F(-3.0e38), F(-3.0e38), F(-3.0e38)};
    const int64_t r0 = tile * rows_per_tile, r1 = (r0 + rows_per_tile < N) ? r0 + rows_per_tile : N;
    for (int64_t q = r0 + lane; q < r1; q += 64) {
        const F* r = pts + q * ld;
        for (int c = 0; c < 3; ++c) { lo[c] = r[c] < lo[c] ? r[c] : lo[c]; hi[c] = r[c] > hi[c] ? r[c] : hi[c]; }
    }
    for (int c = 0; c < 3; ++c) { lo[c] = wave_min<F>(lo[c]); hi[c] = wave_max<F>(hi[c]); }
    if (lane == 0)
        for (int c = 0; c < 3; ++c) { boxes[tile * 6 + c] = lo[c]; boxes[tile * 6 + 3 + c] = hi[c]; }
}

// W[k][j] = sum over the target tiles that overlap patch j of the tile's partial for j (slot 0 when j is the group of
// the tile's first row; with 2 slots everything else is slot 1, with 3 slots the next group is slot 1 and the rest slot 2),
// in tile order - the second half of the fused interaction matrix
__global__ __launch_bounds__(256) void tile_interactions_kernel(const double* __restrict__ w_part, int64_t n_tiles,
                                                                int rows_per_tile, const int64_t* __restrict__ point_patch,
                                                                const int64_t* __restrict__ patch_off, int64_t P,
                                                                int64_t K, double* __restrict__ W, int slots) {
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x, k = blockIdx.y;
    if (j >= P || k >= K) return;
    const int64_t lo = patch_off[j], hi = patch_off[j + 1];
    double s = 0.0;
    if (hi > lo) {
        const int64_t t0 = lo / rows_per_tile, t1 = (hi - 1) / rows_per_tile;
        const double* part = w_part + k * n_tiles * slots;
        for (int64_t t = t0; t <= t1; ++t) {
            const int64_t g0 = point_patch[t * rows_per_tile];
            const int slot = (g0 == j) ? 0 : ((slots == 3 && g0 + 1 == j) ? 1 : slots - 1);
            s += part[t * slots + slot];
        }
    }
    W[k * P + j] = s;
}

// the precondition of the pair kernel's interaction partials (w_part): the rows of a target tile take at most `slots` (2 or
// 3) group values, and with 3 slots the second one is the first + 1 (the epilogue files a row under "the group of the tile's
// first row", with 3 slots "that group + 1", or "the other one").  One thread per tile; violations[0] += tiles that break it.
__global__ __launch_bounds__(256) void tile_groups_kernel(const int64_t* __restrict__ point_patch, int64_t N, int rows_per_tile,
                                                          int64_t n_tiles, int32_t* __restrict__ violations, int slots) {
    const int64_t tile = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (tile >= n_tiles) return;
    const int64_t r0 = tile * rows_per_tile, r1 = (r0 + rows_per_tile < N) ? r0 + rows_per_tile : N;
    const int64_t first = point_patch[r0];
    int64_t other = first;                              // the one group value that goes into the last slot
    bool bad = false;
    for (int64_t r = r0 + 1; r < r1; ++r) {
        const int64_t v = point_patch[r];
        if (v == first || (slots == 3 && v == first + 1) || v == other) continue;
        if (other == first) other = v; else bad = true;
    }
    if (bad) atomicAdd(violations, 1);
}

}  // namespace dnp

extern "C" {

int dnp_check_tile_groups(const int64_t* point_patch, int64_t N, int w_slots, int32_t* violations, void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0, "negative size");
    DNP_REQUIRE(w_slots == 2 || w_slots == 3, "w_slots=%d (2 or 3)", w_slots);
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(point_patch && violations, "NULL pointer");
    const int rows = 64 * kPatchScalarKT;
    const int64_t n_tiles = ceil_div(N, (int64_t)rows);
    hipLaunchKernelGGL(tile_groups_kernel, dim3((unsigned)ceil_div(n_tiles, 256)), dim3(256), 0, (hipStream_t)stream, point_patch,
                       N, rows, n_tiles, violations, w_slots);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_tile_boxes_f32(const float* pts, int64_t N, int64_t ld_pts, int64_t rows_per_tile, float* boxes, void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0, "negative size");
    DNP_REQUIRE(rows_per_tile > 0 && rows_per_tile <= 1 << 20, "rows_per_tile=%lld", (long long)rows_per_tile);
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(pts && boxes, "NULL pointer");
    DNP_REQUIRE(ld_pts >= 3, "ld_pts=%lld < 3", (long long)ld_pts);
    const int64_t n_tiles = ceil_div(N, rows_per_tile);
    hipLaunchKernelGGL(tile_box_kernel<float>, dim3((unsigned)ceil_div(n_tiles, 4)), dim3(256), 0, (hipStream_t)stream, pts, ld_pts,
                       N, (int)rows_per_tile, n_tiles, boxes);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_tile_boxes_f64(const double* pts, int64_t N, int64_t ld_pts, int64_t rows_per_tile, double* boxes, void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0, "negative size");
    DNP_REQUIRE(rows_per_tile > 0 && rows_per_tile <= 1 << 20, "rows_per_tile=%lld", (long long)rows_per_tile);
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(pts && boxes, "NULL pointer");
    DNP_REQUIRE(ld_pts >= 3, "ld_pts=%lld < 3", (long long)ld_pts);
    const int64_t n_tiles = ceil_div(N, rows_per_tile);
    hipLaunchKernelGGL(tile_box_kernel<double>, dim3((unsigned)ceil_div(n_tiles, 4)), dim3(256), 0, (hipStream_t)stream, pts, ld_pts,
                       N, (int)rows_per_tile, n_tiles, boxes);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int64_t dnp_patch_tile_rows(void) { return 64 * kPatchScalarKT; }

int dnp_interactions_from_tiles(const double* w_part, int w_slots, int64_t K, int64_t N, const int64_t* point_patch,
                                const int64_t* patch_off, int64_t P, double* W, void* stream) {
    clear_error();
    DNP_REQUIRE(K >= 0 && N >= 0 && P >= 0, "negative size");
    DNP_REQUIRE(w_slots == 2 || w_slots == 3, "w_slots=%d (2 or 3)", w_slots);
    if (K == 0 || P == 0) return DNP_OK;
    DNP_REQUIRE(w_part && point_patch && patch_off && W, "NULL pointer");
    DNP_REQUIRE(K <= 65535, "K=%lld slabs exceed one launch (65535)", (long long)K);
    const int rows = 64 * kPatchScalarKT;
    hipLaunchKernelGGL(tile_interactions_kernel, dim3((unsigned)ceil_div(P, 256), (unsigned)K), dim3(256), 0,
                       (hipStream_t)stream, w_part, ceil_div(N, (int64_t)rows), rows, point_patch, patch_off, P, K, W, w_slots);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_boxes_f32(const float* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                        const int64_t* patch_idx, int64_t P, float* boxes, void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0 && P >= 0, "negative size");
    if (P == 0) return DNP_OK;
    DNP_REQUIRE(pts && patch_off && boxes, "NULL pointer");               // patch_idx may be NULL (contiguous)
    DNP_REQUIRE(ld_pts >= 3, "ld_pts=%lld < 3", (long long)ld_pts);
    hipLaunchKernelGGL(patch_box_kernel<float>, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pts, ld_pts, patch_off,
                       patch_idx, boxes);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_boxes_f64(const double* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                        const int64_t* patch_idx, int64_t P, double* boxes, void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0 && P >= 0, "negative size");
    if (P == 0) return DNP_OK;
    DNP_REQUIRE(pts && patch_off && boxes, "NULL pointer");               // patch_idx may be NULL (contiguous)
    DNP_REQUIRE(ld_pts >= 3, "ld_pts=%lld < 3", (long long)ld_pts);
    hipLaunchKernelGGL(patch_box_kernel<double>, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pts, ld_pts, patch_off,
                       patch_idx, boxes);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_fields_f32(const float* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                         const int64_t* patch_idx, int64_t P, const int64_t* point_patch,
                         int64_t p_begin, int64_t p_end, float eps, float* dE, void* stream) {
    return dnp_patch_fields_boxed_f32(pts, N, ld_pts, patch_off, patch_idx, P, point_patch, nullptr, p_begin, p_end, eps,
                                      dE, stream);
}

int dnp_patch_fields_boxed_f32(const float* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                               const int64_t* patch_idx, int64_t P, const int64_t* point_patch,
                               const float* patch_box, int64_t p_begin, int64_t p_end, float eps, float* dE,
                               void* stream) {
    return dnp_patch_fields_tiled_f32(pts, N, ld_pts, patch_off, patch_idx, P, point_patch, patch_box, nullptr, p_begin,
                                      p_end, eps, dE, nullptr, 2, 1, nullptr, 0, stream);
}

// exchange buffer of the split forms: one record per (split patch, target tile) - the arrival counter in a 128-byte line of
// its own, then 4 runs x 6 doubles x 64 lanes of run terms (pair_kernel.h, xch_item_bytes): 12 416 bytes per item,
// split_patches * ceil(N / 128) items (29 MB for 3 split patches at N = 100 000)
size_t dnp_patch_exchange_bytes(int64_t N, int64_t split_patches) {
    if (N <= 0 || split_patches <= 0) return 0;
    return (size_t)split_patches * (size_t)ceil_div(N, (int64_t)64 * kPatchScalarKT) * (size_t)xch_item_bytes(4, kPatchScalarKT, 3);
}

int dnp_exchange_init(void* exchange, size_t bytes, void* stream) {
    clear_error();
    if (bytes == 0) return DNP_OK;
    DNP_REQUIRE(exchange, "NULL exchange buffer");
    DNP_CHECK_HIP(hipMemsetAsync(exchange, 0, bytes, (hipStream_t)stream));
    return DNP_OK;
}

static int patch_fields_f32(const float* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                            const int64_t* patch_idx, int64_t P, const int64_t* point_patch,
                            const float* patch_box, const float* tile_box, int64_t p_begin, int64_t p_end, float eps,
                            float* dE, double* w_part, int w_slots, int source_split, void* exchange, size_t exchange_bytes,
                            const int32_t* patch_order, void* stream) {
    clear_error();
    DNP_REQUIRE(!w_part || w_slots == 2 || w_slots == 3, "w_slots=%d (2 or 3 group slots per tile)", w_slots);
    DNP_REQUIRE(source_split == 1 || (source_split < 0 && source_split >= -65535),
                "source_split=%d (1, or -k: the last k patches of the launch as split items)", source_split);
    DNP_REQUIRE(N >= 0 && P >= 0, "negative size");
    DNP_REQUIRE(0 <= p_begin && p_begin <= p_end && p_end <= P, "bad patch range [%lld,%lld) of %lld",
                (long long)p_begin, (long long)p_end, (long long)P);
    if (N == 0 || p_begin == p_end) return DNP_OK;
    DNP_REQUIRE(pts && patch_off && point_patch && dE, "NULL pointer");   // patch_idx may be NULL (contiguous)
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    // the tile tables belong to the scalar-unit kernel on the patch-sorted layout with the far chain on
    const bool scalar_path = !patch_idx && eps > 0.f && !DNP_FORCE_LDS;
    DNP_REQUIRE(!w_part || (scalar_path && patch_box && tile_box && kPatchFar && far_threshold_d2((double)eps) > 0.0),
                "w_part needs the patch-sorted layout (patch_idx == NULL), eps >= 1e-30 and both box tables");
    const int64_t t_tiles = ceil_div(N, (int64_t)kBlock * kPatchKT);
    const int64_t K = p_end - p_begin;
    DNP_REQUIRE(!patch_order || (scalar_path && K <= 65535), "patch_order needs the patch-sorted layout, eps > 0 and at most 65535 patches");
    // grid.y is limited to 65535 workgroups: walk the patch range in slices
    for (int64_t k0 = 0; k0 < K; k0 += 65535) {
        const int64_t kn = (K - k0 < 65535) ? (K - k0) : 65535;
        PairArgs<float, float> pa{};
        pa.src = pts; pa.ld_src = ld_pts; pa.src_idx = patch_idx;
        pa.tgt = pts; pa.ld_tgt = ld_pts; pa.tgt_idx = nullptr; pa.T = N;
        pa.chunk_off_dev = patch_off; pa.chunk_base = p_begin + k0; pa.tgt_group = point_patch;
        pa.eps = eps; pa.partial = dE + k0 * N * 3;
        pa.chunk_perm = patch_order;
        pa.far_d2 = (float)far_threshold_d2((double)eps);
#ifdef DNP_FAR_D2   // timing experiments only: force the far test (1e30f = never far, -1.f = always far)
        pa.far_d2 = DNP_FAR_D2;
#endif
        const hipStream_t st = (hipStream_t)stream;
#ifdef DNP_BOUNDS
        {
            const int64_t n_tiles_b = ceil_div(N, (int64_t)64 * kPatchScalarKT);
            pa.bnd.n_chunk_off = P + 1; pa.bnd.n_chunk_box = patch_box ? P : 0; pa.bnd.n_tile_box = tile_box ? n_tiles_b : 0;
            pa.bnd.n_tgt_group = N; pa.bnd.n_w_part = w_part ? kn * n_tiles_b * w_slots : 0; pa.bnd.n_partial = kn * N * 3;
            pa.bnd.n_xch_items = exchange ? (int64_t)(exchange_bytes / (size_t)xch_item_bytes(4, kPatchScalarKT, 3)) : 0;
            pa.bnd.n_src_rows = N; pa.bnd.n_chunk_perm = patch_order ? kn : 0; pa.bnd.err = bounds_err_buffer();
        }
#endif
        if (scalar_path) {
            // patch-sorted cloud (what the drivers pass): contiguous sources -> the scalar-unit kernel
            const bool tabled = patch_box && tile_box && kPatchFar && pa.far_d2 > 0.f;
            // -k: ONE launch whose last k patches are split items with their run terms in the exchange buffer
            // (pair_kernel.h XCH); it exists for the fully tabled form, needs the whole range in this slice, and k >= the
            // range means every patch
            const int tail = (tabled && source_split < 0 && k0 == 0 && kn == K) ? (int)(-source_split < K ? -source_split : K) : 0;
            pa.chunk_box = patch_box;
            pa.tile_box = tile_box;
            pa.w_part = w_part ? w_part + k0 * ceil_div(N, (int64_t)64 * kPatchScalarKT) * w_slots : nullptr;
            if (tail) {
                const int64_t n_tiles = ceil_div(N, (int64_t)64 * kPatchScalarKT);
                const size_t need = dnp_patch_exchange_bytes(N, tail);
                if (!exchange || exchange_bytes < need) {
                    set_error("exchange buffer of %zu bytes required for %d split patches, %zu given", need, tail,
                              exchange ? exchange_bytes : (size_t)0);
                    return DNP_EWORKSPACE;
                }
                constexpr int kW = kXchWaves;
                // split part: per patch the tiles padded to a multiple of 8 (XCD-first numbering), 4 / kW workgroups per tile
                const int64_t blocks = (K - tail) * ceil_div(n_tiles, (int64_t)kW) + (int64_t)tail * ceil_div(n_tiles, (int64_t)8) * 8 * (4 / kW);
                DNP_REQUIRE(blocks < ((int64_t)1 << 31), "the tail form's grid of %lld workgroups", (long long)blocks);
                pa.split_from = (int)(K - tail);
                pa.n_chunks = (int)K;
                pa.xch_ticket = (unsigned int*)exchange;
                pa.xch_terms = (double*)((char*)exchange + 128);
                const dim3 tgrid((unsigned)blocks);
                if (w_part && w_slots == 3)
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar, true, true, 3, 4, kW, true>),
                                       tgrid, dim3(kW * 64), 0, st, pa);
                else if (w_part)
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar, true, true, 2, 4, kW, true>),
                                       tgrid, dim3(kW * 64), 0, st, pa);
                else
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar, true, true, 0, 4, kW, true>),
                                       tgrid, dim3(kW * 64), 0, st, pa);
            } else if (tabled) {
                // the tabled, unsplit form runs in workgroups of kTabledWaves wavefronts (pair_kernel.h, WAVES)
                const dim3 sgrid((unsigned)ceil_div(N, (int64_t)kTabledWaves * 64 * kPatchScalarKT), (unsigned)kn);
                if (w_part && w_slots == 3)
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar, true, true, 3, 1, kTabledWaves>),
                                       sgrid, dim3(kTabledWaves * 64), 0, st, pa);
                else if (w_part)
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar, true, true, 2, 1, kTabledWaves>),
                                       sgrid, dim3(kTabledWaves * 64), 0, st, pa);
                else
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar, true, true, 0, 1, kTabledWaves>),
                                       sgrid, dim3(kTabledWaves * 64), 0, st, pa);
            } else {
                const dim3 sgrid((unsigned)ceil_div(N, (int64_t)(kBlock / 64) * 64 * kPatchScalarKT), (unsigned)kn);
                if (patch_box && kPatchFar)
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar, true>), sgrid,
                                       dim3(kBlock), 0, st, pa);
                else
                    hipLaunchKernelGGL((pair_kernel_scalar<float, float, kField, kPatchScalarKT, kFast, kPatchFar>), sgrid,
                                       dim3(kBlock), 0, st, pa);
            }
        } else {
            const dim3 grid((unsigned)t_tiles, (unsigned)kn);
            if (eps > 0.f)
                hipLaunchKernelGGL((pair_kernel<float, float, kField, kPatchKT, kFast>), grid, dim3(kBlock), 0, st, pa);
            else if (eps == 0.f)
                hipLaunchKernelGGL((pair_kernel<float, float, kField, kPatchKT, kNanCoinc>), grid, dim3(kBlock), 0, st, pa);
            else
                hipLaunchKernelGGL((pair_kernel<float, float, kField, kPatchKT, kRobust>), grid, dim3(kBlock), 0, st, pa);
        }
        DNP_CHECK_HIP(hipGetLastError());
    }
    return DNP_OK;
}

int dnp_patch_fields_tiled_f32(const float* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                               const int64_t* patch_idx, int64_t P, const int64_t* point_patch,
                               const float* patch_box, const float* tile_box, int64_t p_begin, int64_t p_end, float eps,
                               float* dE, double* w_part, int w_slots, int source_split, void* exchange, size_t exchange_bytes,
                               void* stream) {
    return patch_fields_f32(pts, N, ld_pts, patch_off, patch_idx, P, point_patch, patch_box, tile_box, p_begin, p_end, eps, dE, w_part,
                            w_slots, source_split, exchange, exchange_bytes, nullptr, stream);
}

int dnp_patch_fields_ordered_f32(const float* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off, int64_t P,
                                 const int64_t* point_patch, const float* patch_box, const float* tile_box, int64_t p_begin,
                                 int64_t p_end, const int32_t* patch_order, float eps, float* dE, double* w_part, int w_slots,
                                 int source_split, void* exchange, size_t exchange_bytes, void* stream) {
    return patch_fields_f32(pts, N, ld_pts, patch_off, nullptr, P, point_patch, patch_box, tile_box, p_begin, p_end, eps, dE, w_part,
                            w_slots, source_split, exchange, exchange_bytes, patch_order, stream);
}

// ---- the same slabs for a FLOAT64 cloud (round 5): the reference computes in the dtype it is handed (field_utils.py:96-109) and
// its socket path hands it float64 (util.py:71-77), so a float64 cloud's patch fields, interaction sums and diffuse field
// are evaluated in double: the scalar-unit kernel at KT = 2 on the patch-sorted layout (two-wavefront workgroups, XCD-aware
// tile mapping, interaction partials out of the epilogue), the LDS kernel for gathered patches and for eps <= 0.  With both box
// tables (dnp_patch_boxes_f64, dnp_tile_boxes_f64) a (wavefront, patch) whose boxes are far apart runs the fp64 far chain (one
// transcendental, the series to e^4: pair_kernel.h, kFarRatio64); no split tail in this precision.
#ifndef DNP_KT64
#define DNP_KT64 2
#endif
constexpr int kPatchScalarKT64 = DNP_KT64;
static_assert(kPatchScalarKT64 == kPatchScalarKT, "w_part tiles of both precisions are dnp_patch_tile_rows() rows");

static int patch_fields_f64(const double* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                            const int64_t* patch_idx, int64_t P, const int64_t* point_patch,
                            const double* patch_box, const double* tile_box,
                            int64_t p_begin, int64_t p_end, double eps, double* dE, double* w_part, int w_slots,
                            const int32_t* patch_order, void* stream) {
    clear_error();
    DNP_REQUIRE(!w_part || w_slots == 2 || w_slots == 3, "w_slots=%d (2 or 3 group slots per tile)", w_slots);
    DNP_REQUIRE(N >= 0 && P >= 0, "negative size");
    DNP_REQUIRE(0 <= p_begin && p_begin <= p_end && p_end <= P, "bad patch range [%lld,%lld) of %lld",
                (long long)p_begin, (long long)p_end, (long long)P);
    if (N == 0 || p_begin == p_end) return DNP_OK;
    DNP_REQUIRE(pts && patch_off && point_patch && dE, "NULL pointer");   // patch_idx may be NULL (contiguous)
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    const bool scalar_path = !patch_idx && eps > 0.0;
    DNP_REQUIRE(!w_part || scalar_path, "w_part needs the patch-sorted layout (patch_idx == NULL) and eps > 0");
    const int64_t K = p_end - p_begin;
    const int64_t n_tiles = ceil_div(N, (int64_t)64 * kPatchScalarKT64);
    DNP_REQUIRE(!patch_order || (scalar_path && K <= 65535), "patch_order needs the patch-sorted layout, eps > 0 and at most 65535 patches");
    for (int64_t k0 = 0; k0 < K; k0 += 65535) {
        const int64_t kn = (K - k0 < 65535) ? (K - k0) : 65535;
        PairArgs<double, double> pa{};
        pa.src = pts; pa.ld_src = ld_pts; pa.src_idx = patch_idx;
        pa.tgt = pts; pa.ld_tgt = ld_pts; pa.tgt_idx = nullptr; pa.T = N;
        pa.chunk_off_dev = patch_off; pa.chunk_base = p_begin + k0; pa.tgt_group = point_patch;
        pa.eps = eps; pa.partial = dE + k0 * N * 3;
        pa.chunk_perm = patch_order;
        const bool tabled = scalar_path && patch_box && tile_box && far_threshold_d2(eps, kFarRatio64) > 0.0;
        pa.far_d2 = tabled ? far_threshold_d2(eps, kFarRatio64) : 0.0;
        pa.chunk_box = tabled ? patch_box : nullptr;
        pa.tile_box = tabled ? tile_box : nullptr;
        pa.w_part = w_part ? w_part + k0 * n_tiles * w_slots : nullptr;
        const hipStream_t st = (hipStream_t)stream;
#ifdef DNP_BOUNDS
        pa.bnd = PairBounds{};
        pa.bnd.n_chunk_off = P + 1; pa.bnd.n_tgt_group = N; pa.bnd.n_w_part = w_part ? kn * n_tiles * w_slots : 0;
        pa.bnd.n_chunk_box = tabled ? P : 0; pa.bnd.n_tile_box = tabled ? n_tiles : 0;
        pa.bnd.n_partial = kn * N * 3; pa.bnd.n_src_rows = N; pa.bnd.n_chunk_perm = patch_order ? kn : 0; pa.bnd.err = bounds_err_buffer();
#endif
        if (scalar_path) {
            const dim3 sgrid((unsigned)ceil_div(N, (int64_t)kTabledWaves * 64 * kPatchScalarKT64), (unsigned)kn);
#define DNP_LAUNCH_F64(FARF, WP)                                                                                              \
    hipLaunchKernelGGL((pair_kernel_scalar<double, double, kField, kPatchScalarKT64, kFast, FARF, FARF, FARF, WP, 1, kTabledWaves>), \
                       sgrid, dim3(kTabledWaves * 64), 0, st, pa)
            const int wp = w_part ? w_slots : 0;
            if (tabled) { if (wp == 3) DNP_LAUNCH_F64(true, 3); else if (wp == 2) DNP_LAUNCH_F64(true, 2); else DNP_LAUNCH_F64(true, 0); }
            else { if (wp == 3) DNP_LAUNCH_F64(false, 3); else if (wp == 2) DNP_LAUNCH_F64(false, 2); else DNP_LAUNCH_F64(false, 0); }
#undef DNP_LAUNCH_F64
        } else {
            const dim3 grid((unsigned)ceil_div(N, (int64_t)kBlock), (unsigned)kn);     // KT = 1: 116 VGPRs (KT = 4: 220, two wavefronts per SIMD)
            if (eps > 0.0)
                hipLaunchKernelGGL((pair_kernel<double, double, kField, 1, kFast>), grid, dim3(kBlock), 0, st, pa);
            else if (eps == 0.0)
                hipLaunchKernelGGL((pair_kernel<double, double, kField, 1, kNanCoinc>), grid, dim3(kBlock), 0, st, pa);
            else
                hipLaunchKernelGGL((pair_kernel<double, double, kField, 1, kRobust>), grid, dim3(kBlock), 0, st, pa);
        }
        DNP_CHECK_HIP(hipGetLastError());
    }
    return DNP_OK;
}

int dnp_patch_fields_tiled_f64(const double* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off,
                               const int64_t* patch_idx, int64_t P, const int64_t* point_patch,
                               const double* patch_box, const double* tile_box,
                               int64_t p_begin, int64_t p_end, double eps, double* dE, double* w_part, int w_slots, void* stream) {
    return patch_fields_f64(pts, N, ld_pts, patch_off, patch_idx, P, point_patch, patch_box, tile_box, p_begin, p_end, eps, dE, w_part,
                            w_slots, nullptr, stream);
}

int dnp_patch_fields_ordered_f64(const double* pts, int64_t N, int64_t ld_pts, const int64_t* patch_off, int64_t P,
                                 const int64_t* point_patch, const double* patch_box, const double* tile_box, int64_t p_begin,
                                 int64_t p_end, const int32_t* patch_order, double eps, double* dE, double* w_part, int w_slots,
                                 void* stream) {
    return patch_fields_f64(pts, N, ld_pts, patch_off, nullptr, P, point_patch, patch_box, tile_box, p_begin, p_end, eps, dE, w_part,
                            w_slots, patch_order, stream);
}

int dnp_interactions_f32(const float* dE, int64_t K, int64_t N, const float* pts, int64_t ld_pts,
                         const int64_t* patch_off, const int64_t* patch_idx, int64_t P, double* W,
                         void* stream) {
    clear_error();
    DNP_REQUIRE(K >= 0 && N >= 0 && P >= 0, "negative size");
    if (K == 0 || P == 0) return DNP_OK;
    DNP_REQUIRE(dE && pts && patch_off && W, "NULL pointer");             // patch_idx may be NULL (contiguous)
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    DNP_REQUIRE(K <= 65535, "K=%lld slabs exceed one launch (65535)", (long long)K);
    hipLaunchKernelGGL(interactions_kernel<float>, dim3((unsigned)P, (unsigned)K), dim3(256), 0, (hipStream_t)stream, dE, N,
                       pts, ld_pts, patch_off, patch_idx, P, W);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_interactions_f64(const double* dE, int64_t K, int64_t N, const double* pts, int64_t ld_pts,
                         const int64_t* patch_off, const int64_t* patch_idx, int64_t P, double* W,
                         void* stream) {
    clear_error();
    DNP_REQUIRE(K >= 0 && N >= 0 && P >= 0, "negative size");
    if (K == 0 || P == 0) return DNP_OK;
    DNP_REQUIRE(dE && pts && patch_off && W, "NULL pointer");             // patch_idx may be NULL (contiguous)
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    DNP_REQUIRE(K <= 65535, "K=%lld slabs exceed one launch (65535)", (long long)K);
    hipLaunchKernelGGL(interactions_kernel<double>, dim3((unsigned)P, (unsigned)K), dim3(256), 0, (hipStream_t)stream, dE, N,
                       pts, ld_pts, patch_off, patch_idx, P, W);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_combine_fields_f32(const float* dE, int64_t K, int64_t N, const float* coef, const int64_t* slab,
                           int64_t n, float* E, int accumulate, void* stream) {
    clear_error();
    DNP_REQUIRE(K >= 0 && N >= 0 && n >= 0, "negative size");
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(E, "NULL E");
    DNP_REQUIRE(n == 0 || (dE && coef && slab), "NULL pointer");
    const int64_t N3 = N * 3;
    hipLaunchKernelGGL(combine_kernel, dim3((unsigned)ceil_div(N3, 256)), dim3(256), 0, (hipStream_t)stream, dE, N3,
                       coef, slab, n, E, accumulate);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

#ifdef DNP_STAMP
int dnp_debug_set_stamps_patch(void* p) { return dnp::set_stamps_here((unsigned long long*)p) == hipSuccess ? 0 : -4; }
#endif

}  // extern "C"
