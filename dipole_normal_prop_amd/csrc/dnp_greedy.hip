// dnp_greedy.hip - K4: field_utils.strongest_field_propagation_points (field_utils.py:353-388)
// as ONE persistent launch instead of N-1 rounds of ~15 torch ops with three host syncs each.
//
// Single-workgroup form (N <= 512*PPT): 512 threads = 8 waves = 2 per SIMD, so every lane may
// hold up to 24 points (x, n, E: 9 floats each) in its 256-VGPR budget.  One step =
//   (1) every lane scans its unvisited points for max |E.n|            (registers only)
//   (2) wave argmax by DPP shuffles, 8 wave results through LDS, ONE barrier per step
//       (slots double-buffered by step parity)
//   (3) every wave reads the winner's row with wave-uniform loads - pts[] is read-only during
//       the loop: a point's normal flips at most once, when it is chosen, and is never read by
//       anybody else afterwards, so the owner just remembers the flip and writes it at the end
//   (4) every lane adds the winner's dipole field to its points.
// Ties in |E.n| go to the smallest point index, as torch.argmax over the index-ordered
// unvisited subset does (field_utils.py:372-373).
//
// The per-pair arithmetic here uses IEEE sqrt/div in the reference's own op order
// (field_utils.py:96-109) rather than the rsq/rcp form of pair_kernel.h: a step is latency
// bound (one barrier + one dependent row fetch), not ALU bound, and the greedy order is a
// chaotic function of E, so staying as close as possible to the reference's rounding is worth
// more than the ~20 saved instructions.
#include "dnp_common.h"

// no fma contraction in this file: the step arithmetic mirrors the reference's separately
// rounded torch ops
#pragma clang fp contract(off)

namespace dnp {

constexpr int kGreedyThreads = 512;

struct Best {
    float a;      // |interaction|
    float v;      // signed interaction
    int idx;      // point index (INT_MAX = none)
};

__device__ __forceinline__ Best better(const Best& p, const Best& q) {
    // larger |v| wins; ties -> smaller index
    const bool take_q = (q.a > p.a) || (q.a == p.a && q.idx < p.idx);
    return take_q ? q : p;
}

__device__ __forceinline__ void add_dipole_field(float sx, float sy, float sz, float px, float py, float pz,
                                                 float x, float y, float z, float eps, float& ex, float& ey,
                                                 float& ez) {
    // one source -> one target, the leaf of field_utils.py:96-109 verbatim in IEEE fp32
    const float rx = sx - x, ry = sy - y, rz = sz - z;
    const float d2 = rx * rx + ry * ry + rz * rz;
    const float nrm = __builtin_sqrtf(d2);
    float fx = 0.f, fy = 0.f, fz = 0.f;
    if (nrm != 0.f) {
        const float ux = rx / nrm, uy = ry / nrm, uz = rz / nrm;
        const float c = 3.f * (px * ux + py * uy + pz * uz);
        fx = c * ux - px; fy = c * uy - py; fz = c * uz - pz;
    }
    const float den = nrm * nrm * nrm + eps;
    fx = fx / den; fy = fy / den; fz = fz / den;
    // E_total = E.sum(dim=0) * -1 ; Inf/NaN -> 0 (per call)
    fx = -fx; fy = -fy; fz = -fz;
    if (!__builtin_isfinite(fx)) fx = 0.f;
    if (!__builtin_isfinite(fy)) fy = 0.f;
    if (!__builtin_isfinite(fz)) fz = 0.f;
    ex += fx; ey += fy; ez += fz;
}

template <int PPT>
__global__ __launch_bounds__(kGreedyThreads) void point_greedy_kernel(float* __restrict__ pts, int64_t N,
                                                                      int64_t ld, int start, float eps, int diffuse,
                                                                      int64_t* __restrict__ order_out,
                                                                      float* __restrict__ E_out) {
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kGreedyThreads / 64;
    __shared__ Best slots[2][kWaves];

    float x[PPT], y[PPT], z[PPT], nx[PPT], ny[PPT], nz[PPT], ex[PPT], ey[PPT], ez[PPT];
    unsigned visited = 0, flipped = 0, valid = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = (int64_t)k * kGreedyThreads + tid;
        x[k] = y[k] = z[k] = nx[k] = ny[k] = nz[k] = 0.f;
        ex[k] = ey[k] = ez[k] = 0.f;
        if (i < N) {
            const float* p = pts + i * ld;
            x[k] = p[0]; y[k] = p[1]; z[k] = p[2]; nx[k] = p[3]; ny[k] = p[4]; nz[k] = p[5];
            valid |= 1u << k;
        }
    }
    visited = ~valid;  // slots past N never take part

    int cur = start;          // wave-uniform
    float cur_sign = 1.f;     // the start point is not flipped
    for (int64_t step = 0; step < N; ++step) {
        // (3)+(4): add the field of point `cur` (with its possibly flipped normal) to every point
        {
            const float* p = pts + (int64_t)cur * ld;
            const float sx = p[0], sy = p[1], sz = p[2];
            const float px = p[3] * cur_sign, py = p[4] * cur_sign, pz = p[5] * cur_sign;
            const int ck = cur / kGreedyThreads, ct = cur - ck * kGreedyThreads;
            if (tid == ct) {
                visited |= 1u << ck;
                if (cur_sign < 0.f) flipped |= 1u << ck;
            }
            if (order_out && tid == 0) order_out[step] = cur;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool self = (tid == ct) && (k == ck);   // E[~(indx == i)]: the source itself is skipped
                if (((valid >> k) & 1u) && !self)
                    add_dipole_field(sx, sy, sz, px, py, pz, x[k], y[k], z[k], eps, ex[k], ey[k], ez[k]);
            }
        }
        if (step + 1 == N) break;

        // (1) local scan
        Best b{-1.f, 0.f, 0x7fffffff};
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!((visited >> k) & 1u)) {
                // an unvisited point has not been flipped: its normal is the input normal
                const float v = ex[k] * nx[k] + ey[k] * ny[k] + ez[k] * nz[k];
                const Best c{__builtin_fabsf(v), v, k * kGreedyThreads + tid};
                b = better(b, c);
            }
        }
        // (2) wave argmax, then across waves
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            Best o;
            o.a = __shfl_xor(b.a, off, 64);
            o.v = __shfl_xor(b.v, off, 64);
            o.idx = __shfl_xor(b.idx, off, 64);
            b = better(b, o);
        }
        const int par = (int)(step & 1);
        if (lane == 0) slots[par][wave] = b;
        __syncthreads();
        Best g = slots[par][0];
#pragma unroll
        for (int w = 1; w < kWaves; ++w) g = better(g, slots[par][w]);
        cur = __builtin_amdgcn_readfirstlane(g.idx);   // wave-uniform: the row fetch becomes scalar loads
        cur_sign = (g.v < 0.f) ? -1.f : 1.f;   // `if interaction[max] < 0: flip`
    }

    // epilogue: write flips, optional diffuse sign pass (field_utils.py:382-385), E_out
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = (int64_t)k * kGreedyThreads + tid;
        if (i < N) {
            float s = ((flipped >> k) & 1u) ? -1.f : 1.f;
            float a = nx[k] * s, b2 = ny[k] * s, c = nz[k] * s;
            if (diffuse) {
                const float v = ex[k] * a + ey[k] * b2 + ez[k] * c;
                const float sg = (v > 0.f) ? 1.f : -1.f;
                a *= sg; b2 *= sg; c *= sg;
            }
            float* p = pts + i * ld;
            p[3] = a; p[4] = b2; p[5] = c;
            if (E_out) { E_out[i * 3 + 0] = ex[k]; E_out[i * 3 + 1] = ey[k]; E_out[i * 3 + 2] = ez[k]; }
        }
    }
}

}  // namespace dnp

using namespace dnp;

extern "C" {

size_t dnp_point_greedy_workspace_bytes(int64_t N) {
    (void)N;
    return 256;  // the single-workgroup form keeps all state in registers / LDS
}

int dnp_point_greedy_max_points(void) { return kGreedyThreads * 24; }

int dnp_point_greedy_f32(float* pts, int64_t N, int64_t ld_pts, int64_t start, float eps, int diffuse,
                         int64_t* order_out, float* E_out, void* workspace, size_t workspace_bytes,
                         void* stream) {
    (void)workspace; (void)workspace_bytes;
    clear_error();
    DNP_REQUIRE(N >= 0, "negative N");
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(pts, "NULL pts");
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    DNP_REQUIRE(start >= 0 && start < N, "starting_point %lld out of range [0,%lld)", (long long)start, (long long)N);
    DNP_REQUIRE(N <= (int64_t)kGreedyThreads * 24,
                "N=%lld exceeds the %d points of the single-workgroup persistent kernel", (long long)N,
                kGreedyThreads * 24);
    hipStream_t st = (hipStream_t)stream;
#define DNP_LAUNCH_GREEDY(P)                                                                                   \
    hipLaunchKernelGGL((point_greedy_kernel<P>), dim3(1), dim3(kGreedyThreads), 0, st, pts, N, ld_pts, (int)start, \
                       eps, diffuse, order_out, E_out)
    if (N <= kGreedyThreads * 2) DNP_LAUNCH_GREEDY(2);
    else if (N <= kGreedyThreads * 4) DNP_LAUNCH_GREEDY(4);
    else if (N <= kGreedyThreads * 8) DNP_LAUNCH_GREEDY(8);
    else if (N <= kGreedyThreads * 16) DNP_LAUNCH_GREEDY(16);
    else DNP_LAUNCH_GREEDY(24);
#undef DNP_LAUNCH_GREEDY
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

}  // extern "C"
