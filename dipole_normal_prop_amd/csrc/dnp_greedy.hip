// dnp_greedy.hip - K4: field_utils.strongest_field_propagation_points (field_utils.py:353-388)
// as ONE persistent launch instead of N-1 rounds of ~15 torch ops with three host syncs each.
//
// Single-workgroup form (N <= 512*PPT): 512 threads = 8 waves = 2 per SIMD, so every lane may
// hold up to 24 points (x, n, E: 9 floats each) in its 256-VGPR budget.  One step =
//   (1) every lane scans its unvisited points for max |E.n|            (registers only)
//   (2) every lane folds (|E.n|, index, sign) into one 64-bit key and does ONE ds_max_u64 on a word in LDS,
//       ONE barrier per step (three words in rotation)
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

// Candidate key: one 64-bit word whose unsigned order is the selection order of the greedy step -
//   bits 63..32  |interaction| as its IEEE bit pattern (monotone for non-negative floats)
//   bits 31..1   0x7fffffff - point index   (ties in |interaction| go to the smallest index, as torch.argmax)
//   bit  0       1 when the interaction is negative
// so a workgroup's winner is ONE ds_max_u64 per lane instead of a 6-step shuffle tree over three values.
__device__ __forceinline__ unsigned long long candidate_key(float v, int idx) {
    const unsigned absbits = __builtin_bit_cast(unsigned, __builtin_fabsf(v));
    const unsigned low = ((0x7fffffffu - (unsigned)idx) << 1) | (v < 0.f ? 1u : 0u);
    return ((unsigned long long)absbits << 32) | low;
}
constexpr unsigned long long kNoCandidate = 0ull;   // below every real key (index field of a real key is > 0)

template <int PPT>
__global__ __launch_bounds__(kGreedyThreads) void point_greedy_kernel(float* __restrict__ pts, int64_t N,
                                                                      int64_t ld, int start, float eps, int diffuse,
                                                                      int64_t* __restrict__ order_out,
                                                                      float* __restrict__ E_out) {
    const int tid = threadIdx.x;
    __shared__ unsigned long long best_key[3];      // rotating, see the step loop
    if (tid == 0) best_key[0] = best_key[1] = best_key[2] = kNoCandidate;
    __syncthreads();

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

        // (1) local scan -> (2) one LDS atomic max of the packed candidate key per lane, ONE barrier per step
        unsigned long long key = kNoCandidate;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!((visited >> k) & 1u)) {
                // an unvisited point has not been flipped: its normal is the input normal
                const float v = ex[k] * nx[k] + ey[k] * ny[k] + ez[k] * nz[k];
                const unsigned long long c = candidate_key(v, k * kGreedyThreads + tid);
                key = c > key ? c : key;
            }
        }
        // three words in rotation: word (step+1)%3 was last READ right after the barrier of step-2, i.e. before
        // every thread's arrival at the barrier of step-1, so thread 0 may clear it now; it is next written
        // after this step's barrier
        const int par = (int)(step % 3);
        if (tid == 0) best_key[(par + 1) % 3] = kNoCandidate;
        if (key != kNoCandidate) atomicMax(&best_key[par], key);
        __syncthreads();
        const unsigned long long gk = best_key[par];
        cur = __builtin_amdgcn_readfirstlane((int)(0x7fffffffu - (unsigned)((gk & 0xffffffffull) >> 1)));
        cur_sign = (gk & 1ull) ? -1.f : 1.f;                 // `if interaction[max] < 0: flip`
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
    const int G = gridDim.x, g = blockIdx.x;
    const int64_t base = (int64_t)g * a.per_group;
    __shared__ unsigned long long local_key[2];     // per step parity
    __shared__ float win_row[8];                    // winner's (x, y, z, nx, ny, nz, signed interaction)
    __shared__ int win_idx;
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
    if (tid == 0) {
        abort_flag = 0;
        local_key[0] = local_key[1] = kNoCandidate;
        const float* p = a.pts + (int64_t)a.start * a.ld;
        for (int c = 0; c < 6; ++c) win_row[c] = p[c];
        win_row[6] = 1.f;                            // the start point is not flipped
        win_idx = a.start;
    }
    __syncthreads();

    for (int64_t step = 0; step < a.N; ++step) {
        // the chosen point's row comes from LDS: wave 0 fetched it while it was polling (or the prologue did)
        const int cur = win_idx;
        const float cur_sign = (win_row[6] < 0.f) ? -1.f : 1.f;     // `if interaction[max] < 0: flip`
        {
            const float sx = win_row[0], sy = win_row[1], sz = win_row[2];
            const float px = win_row[3] * cur_sign, py = win_row[4] * cur_sign, pz = win_row[5] * cur_sign;
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

        // local winner: one LDS atomic max per lane that has a candidate
        unsigned long long key = kNoCandidate;
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!((visited >> k) & 1u)) {
                const float v = ex[k] * nx[k] + ey[k] * ny[k] + ez[k] * nz[k];
                const unsigned long long c = candidate_key(v, (int)(base + (int64_t)k * kGreedyThreads + tid));
                key = c > key ? c : key;
            }
        }
        const int par = (int)(step & 1);
        if (key != kNoCandidate) atomicMax(&local_key[par], key);
        __syncthreads();                              // (also orders this step's win_row reads before its rewrite)
        if (wave == 0) {
            const unsigned tag = (unsigned)(step & 0xfff);
            unsigned long long* row = a.slots + (size_t)par * kMaxGroups;
            if (lane == 0) {
                const unsigned long long lk = local_key[par];
                local_key[par] = kNoCandidate;        // ready for step + 2 (step + 1 uses the other word)
                // granule: the key's low word carries index and sign; its high word (|v| bits) keeps the top 20
                // bits for |v| ... no: the granule must also carry the step tag, so it is re-packed:
                //   { float signed_interaction ; tag << 20 | index }   (index kIdxMask = no candidate)
                unsigned idx = kIdxMask;
                float v = 0.f;
                if (lk != kNoCandidate) {
                    idx = 0x7fffffffu - (unsigned)((lk & 0xffffffffull) >> 1);
                    v = __builtin_bit_cast(float, (unsigned)(lk >> 32));
                    if (lk & 1ull) v = -v;
                }
                const unsigned long long gran = ((unsigned long long)((tag << kTagShift) | idx) << 32) |
                                                (unsigned long long)__builtin_bit_cast(unsigned, v);
                __hip_atomic_store(row + g, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // all-gather: lane l polls groups l, l+64, ...; as soon as a group's candidate is known its row is
            // requested, so the winner's row is already on its way when the argmax is done
            unsigned long long best = kNoCandidate;
            float r0 = 0.f, r1 = 0.f, r2 = 0.f, r3 = 0.f, r4 = 0.f, r5 = 0.f;
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
                    const unsigned long long c = candidate_key(v, (int)idx);
                    if (c > best) {
                        best = c;
                        const float* p = a.pts + (int64_t)idx * a.ld;    // read-only during the loop
                        r0 = p[0]; r1 = p[1]; r2 = p[2]; r3 = p[3]; r4 = p[4]; r5 = p[5];
                    }
                }
            }
            if (__any(timed_out)) {
                if (lane == 0) {
                    __hip_atomic_store(a.status, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                    abort_flag = 1;
                }
            }
            // wave argmax of the 64-bit key (two dwords per step instead of three values)
            unsigned long long wbest = best;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const unsigned long long o = __shfl_xor(wbest, off, 64);
                wbest = o > wbest ? o : wbest;
            }
            if (best == wbest && best != kNoCandidate) {          // exactly one lane: keys are unique per point
                win_row[0] = r0; win_row[1] = r1; win_row[2] = r2; win_row[3] = r3; win_row[4] = r4; win_row[5] = r5;
                win_row[6] = (best & 1ull) ? -1.f : 1.f;
                win_idx = (int)(0x7fffffffu - (unsigned)((best & 0xffffffffull) >> 1));
            }
        }
        __syncthreads();
        if (abort_flag) break;
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
    if (workspace && workspace_bytes >= 256)      // status word: 0 = ok (only the multi-workgroup form can set it)
        DNP_CHECK_HIP(hipMemsetAsync(workspace, 0, 256, st));
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
