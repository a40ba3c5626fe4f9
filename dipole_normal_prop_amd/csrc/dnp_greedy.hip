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
#include <stdlib.h>

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


// ---- multi-workgroup form (N > 512*24) -----------------------------------------------------------------
// G workgroups (one per CU, all co-resident), each owning a contiguous slice of the cloud in registers.
// Per step every workgroup publishes its local winner as ONE naturally aligned 8-byte granule
//   { float signed_interaction ; uint32 (step & 0xfff) << 20 | point_index }          (index < 2^20)
// with a relaxed agent-scope store (payload and tag travel together, so no release/acquire pair is needed:
// MI355X_MICROARCH.md "granule"), and wave 0 of every workgroup polls the G granules of the step with
// relaxed agent-scope loads (they bypass the per-CU L1) until all carry the step's tag - an all-gather,
// not a barrier.  Slots are double-buffered by step parity: a workgroup writes its step-(n+1) slot only
// after it has read every step-n slot, i.e. after everybody has finished reading the step-(n-1) slots it
// overwrites.  pts[] is read-only during the loop (see the single-workgroup form), so the winner's row is
// fetched with plain loads.  Every spin is bounded: on timeout the workgroup raises status[0] and every
// workgroup leaves the loop.
constexpr unsigned kTagShift = 20;
constexpr unsigned kIdxMask = (1u << kTagShift) - 1;
constexpr int kMaxGroups = 256;

struct MultiArgs {
    float* pts; int64_t N; int64_t ld; int start; float eps; int diffuse;
    int64_t* order_out; float* E_out;
    unsigned long long* slots;   // [2][kMaxGroups]
    int* status;                 // [0] = 0 ok, 1 timeout
    int per_group;               // points per workgroup (multiple of 512)
};

template <int PPT>
__global__ __launch_bounds__(kGreedyThreads) void point_greedy_multi_kernel(const MultiArgs a) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kGreedyThreads / 64;
    const int G = gridDim.x, g = blockIdx.x;
    const int64_t base = (int64_t)g * a.per_group;
    __shared__ Best slots_l[kWaves];
    __shared__ Best global_best;
    __shared__ int abort_flag;

    float x[PPT], y[PPT], z[PPT], nx[PPT], ny[PPT], nz[PPT], ex[PPT], ey[PPT], ez[PPT];
    unsigned visited = 0, flipped = 0, valid = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = base + (int64_t)k * kGreedyThreads + tid;
        x[k] = y[k] = z[k] = nx[k] = ny[k] = nz[k] = 0.f;
        ex[k] = ey[k] = ez[k] = 0.f;
        if (i < a.N && i < base + a.per_group) {
            const float* p = a.pts + i * a.ld;
            x[k] = p[0]; y[k] = p[1]; z[k] = p[2]; nx[k] = p[3]; ny[k] = p[4]; nz[k] = p[5];
            valid |= 1u << k;
        }
    }
    visited = ~valid;
    if (tid == 0) abort_flag = 0;
    __syncthreads();

    int cur = a.start;
    float cur_sign = 1.f;
    for (int64_t step = 0; step < a.N; ++step) {
        {
            const float* p = a.pts + (int64_t)cur * a.ld;
            const float sx = p[0], sy = p[1], sz = p[2];
            const float px = p[3] * cur_sign, py = p[4] * cur_sign, pz = p[5] * cur_sign;
            const int64_t rel = (int64_t)cur - base;
            const bool mine = rel >= 0 && rel < a.per_group;
            const int ck = mine ? (int)(rel / kGreedyThreads) : -1, ct = mine ? (int)(rel - (int64_t)ck * kGreedyThreads) : -1;
            if (tid == ct) {
                visited |= 1u << ck;
                if (cur_sign < 0.f) flipped |= 1u << ck;
            }
            if (a.order_out && g == 0 && tid == 0) a.order_out[step] = cur;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool self = (tid == ct) && (k == ck);
                if (((valid >> k) & 1u) && !self)
                    add_dipole_field(sx, sy, sz, px, py, pz, x[k], y[k], z[k], a.eps, ex[k], ey[k], ez[k]);
            }
        }
        if (step + 1 == a.N) break;

        Best b{-1.f, 0.f, 0x7fffffff};
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!((visited >> k) & 1u)) {
                const float v = ex[k] * nx[k] + ey[k] * ny[k] + ez[k] * nz[k];
                const Best c{__builtin_fabsf(v), v, (int)(base + (int64_t)k * kGreedyThreads + tid)};
                b = better(b, c);
            }
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            Best o;
            o.a = __shfl_xor(b.a, off, 64);
            o.v = __shfl_xor(b.v, off, 64);
            o.idx = __shfl_xor(b.idx, off, 64);
            b = better(b, o);
        }
        if (lane == 0) slots_l[wave] = b;
        __syncthreads();
        if (wave == 0) {
            Best wg = slots_l[0];
#pragma unroll
            for (int w = 1; w < kWaves; ++w) wg = better(wg, slots_l[w]);
            const unsigned tag = (unsigned)(step & 0xfff);
            unsigned long long* row = a.slots + (size_t)(step & 1) * kMaxGroups;
            if (lane == 0) {
                // a workgroup with nothing left publishes index kIdxMask (never a real point: N < 2^20)
                const unsigned idx = (wg.idx == 0x7fffffff) ? kIdxMask : (unsigned)wg.idx;
                const unsigned long long gran = ((unsigned long long)((tag << kTagShift) | idx) << 32) |
                                                (unsigned long long)__builtin_bit_cast(unsigned, wg.v);
                __hip_atomic_store(row + g, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // all-gather: lane l polls groups l, l+64, ...
            Best best{-1.f, 0.f, 0x7fffffff};
            bool timed_out = false;
            for (int q = lane; q < G; q += 64) {
                unsigned long long gran;
                unsigned spins = 0;
                for (;;) {
                    gran = __hip_atomic_load(row + q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    if ((unsigned)(gran >> (32 + kTagShift)) == tag) break;
                    if (++spins > (1u << 22) ||
                        __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        timed_out = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (timed_out) break;
                const unsigned idx = (unsigned)(gran >> 32) & kIdxMask;
                if (idx != kIdxMask) {
                    const float v = __builtin_bit_cast(float, (unsigned)(gran & 0xffffffffu));
                    best = better(best, Best{__builtin_fabsf(v), v, (int)idx});
                }
            }
            if (__any(timed_out)) {
                if (lane == 0) {
                    __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    abort_flag = 1;
                }
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                Best o;
                o.a = __shfl_xor(best.a, off, 64);
                o.v = __shfl_xor(best.v, off, 64);
                o.idx = __shfl_xor(best.idx, off, 64);
                best = better(best, o);
            }
            if (lane == 0) global_best = best;
        }
        __syncthreads();
        if (abort_flag) break;
        const Best gb = global_best;
        cur = __builtin_amdgcn_readfirstlane(gb.idx);
        cur_sign = (gb.v < 0.f) ? -1.f : 1.f;
        // global_best / slots_l are rewritten only after the next step's first barrier
    }

#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = base + (int64_t)k * kGreedyThreads + tid;
        if ((valid >> k) & 1u) {
            float s = ((flipped >> k) & 1u) ? -1.f : 1.f;
            float c0 = nx[k] * s, c1 = ny[k] * s, c2 = nz[k] * s;
            if (a.diffuse) {
                const float v = ex[k] * c0 + ey[k] * c1 + ez[k] * c2;
                const float sg = (v > 0.f) ? 1.f : -1.f;
                c0 *= sg; c1 *= sg; c2 *= sg;
            }
            float* p = a.pts + i * a.ld;
            p[3] = c0; p[4] = c1; p[5] = c2;
            if (a.E_out) { a.E_out[i * 3 + 0] = ex[k]; a.E_out[i * 3 + 1] = ey[k]; a.E_out[i * 3 + 2] = ez[k]; }
        }
    }
}

}  // namespace dnp

using namespace dnp;

extern "C" {

size_t dnp_point_greedy_workspace_bytes(int64_t N) {
    (void)N;
    // [0,256): status word (+ padding); then the 2 x 256 eight-byte granule slots of the multi-workgroup form
    return 256 + 2 * kMaxGroups * sizeof(unsigned long long);
}

int dnp_point_greedy_max_points(void) { return (int)kIdxMask; }   // index field of the granule: N < 2^20

int dnp_point_greedy_f32(float* pts, int64_t N, int64_t ld_pts, int64_t start, float eps, int diffuse,
                         int64_t* order_out, float* E_out, void* workspace, size_t workspace_bytes,
                         void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0, "negative N");
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(pts, "NULL pts");
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    DNP_REQUIRE(start >= 0 && start < N, "starting_point %lld out of range [0,%lld)", (long long)start, (long long)N);
    DNP_REQUIRE(N < (int64_t)kIdxMask, "N=%lld exceeds the %u points of the persistent per-point kernel",
                (long long)N, kIdxMask - 1);
    hipStream_t st = (hipStream_t)stream;
    // Form selection: one workgroup keeps everything in registers but pays ~0.9 us per point-per-lane and step
    // (IEEE div/sqrt chain); one workgroup per CU pays ~7 us per step for the granule all-gather.  Measured
    // crossover ~3000 points (ok.xyz, 10 000 points: 20.6 vs 7.0 us/step).  DNP_GREEDY_FORCE_MULTI=1 / =0
    // force the multi- / single-workgroup form where it applies (tests).
    const char* force = getenv("DNP_GREEDY_FORCE_MULTI");
    bool multi = N > 2048;
    if (force && force[0] == '1') multi = true;
    if (force && force[0] == '0' && N <= (int64_t)kGreedyThreads * 24) multi = false;
    if (!multi) {
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

    const size_t need = dnp_point_greedy_workspace_bytes(N);
    if (!workspace || workspace_bytes < need) {
        set_error("workspace of %zu bytes required, %zu given", need, workspace ? workspace_bytes : (size_t)0);
        return DNP_EWORKSPACE;
    }
    int dev = 0, cus = 0;
    DNP_CHECK_HIP(hipGetDevice(&dev));
    DNP_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    int groups = cus < kMaxGroups ? cus : kMaxGroups;          // one resident workgroup per CU
    const char* genv = getenv("DNP_GREEDY_GROUPS");
    if (genv && atoi(genv) > 0 && atoi(genv) < groups) groups = atoi(genv);
    int64_t per = ceil_div(ceil_div(N, (int64_t)groups), (int64_t)kGreedyThreads) * kGreedyThreads;
    groups = (int)ceil_div(N, per);
    const int ppt = (int)(per / kGreedyThreads);
    DNP_REQUIRE(ppt <= 24, "N=%lld needs %d points per lane on %d CUs (max 24)", (long long)N, ppt, groups);
    // tags start at step 0: every slot must hold a tag that no early step uses
    DNP_CHECK_HIP(hipMemsetAsync(workspace, 0xff, need, st));
    DNP_CHECK_HIP(hipMemsetAsync(workspace, 0, 256, st));
    MultiArgs ma{pts, N, ld_pts, (int)start, eps, diffuse, order_out, E_out,
                 (unsigned long long*)((char*)workspace + 256), (int*)workspace, (int)per};
#define DNP_LAUNCH_MULTI(P) \
    hipLaunchKernelGGL((point_greedy_multi_kernel<P>), dim3(groups), dim3(kGreedyThreads), 0, st, ma)
    if (ppt <= 1) DNP_LAUNCH_MULTI(1);
    else if (ppt <= 2) DNP_LAUNCH_MULTI(2);
    else if (ppt <= 4) DNP_LAUNCH_MULTI(4);
    else if (ppt <= 8) DNP_LAUNCH_MULTI(8);
    else if (ppt <= 16) DNP_LAUNCH_MULTI(16);
    else DNP_LAUNCH_MULTI(24);
#undef DNP_LAUNCH_MULTI
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

}  // extern "C"
