// dnp_greedy.hip - K4: field_utils.strongest_field_propagation_points (field_utils.py:353-388)
// as ONE persistent launch instead of N-1 rounds of ~15 torch ops with three host syncs each.
// Templated on the cloud's precision F: float for the file-based callers, double for the socket path
// (util.py:71-77 feeds float64 clouds, which the reference then propagates in fp64).
//
// Single-workgroup form: 512 threads = 8 waves = 2 per SIMD, every lane holds up to PPT points
// (x, n, E: 9 values each) in registers.  One step =
//   (1) every lane scans its unvisited points for max |E.n|            (registers only)
//   (2) workgroup argmax: fp32 folds (|E.n|, index, sign) into one 64-bit key and does ONE ds_max_u64 per lane;
//       fp64 (the magnitude alone needs 64 bits) runs a wave butterfly and leaves one candidate per wave in LDS.
//       ONE barrier per step either way.
//   (3) every wave reads the winner's row with wave-uniform loads.  pts[] is never written by these kernels:
//       the oriented normals go to a scratch array and a second, stream-ordered kernel copies them into pts
//       once every workgroup has left the loop (no workgroup may still be fetching a candidate's row then).
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

#ifndef DNP_K4_THREADS      // A/B builds (tools/gpu_k4_ab.py): threads per workgroup of both forms
#define DNP_K4_THREADS 512
#endif
constexpr int kGreedyThreads = DNP_K4_THREADS;
constexpr int kGreedyWaves = kGreedyThreads / 64;
// Threads per workgroup of the MULTI-workgroup form (round 5, tools/gpu_k4_ab.py on ok.xyz, 10 000 points): a step is [every thread
// adds the winner's field to its point(s): the IEEE div / sqrt chain] + [workgroup argmax] + [all-gather of the workgroups'
// candidates].  256 threads = one wavefront per SIMD for the chain, 40 workgroups to gather from.  fp64 (a chain several times
// longer): 24.1 ms against 32.6 with 512 threads (two wavefronts per SIMD), 27.8 with 128, 35 with 320 / 384 (uneven wavefronts
// per SIMD).  fp32: with the granule slots one cache line apart (slot_stride below) 22.3 ms against 25.0 with 512 threads; with
// the dense slots of rounds 1-4 the 40 writers of 256-thread workgroups cost more than the chain gained (30.9 against 26.2 ms).
#ifndef DNP_K4_MULTI_THREADS_F32
#define DNP_K4_MULTI_THREADS_F32 256
#endif
#ifndef DNP_K4_MULTI_THREADS_F64
#define DNP_K4_MULTI_THREADS_F64 256
#endif
template <typename F> constexpr int multi_threads() { return sizeof(F) == 8 ? DNP_K4_MULTI_THREADS_F64 : DNP_K4_MULTI_THREADS_F32; }
static_assert(DNP_K4_MULTI_THREADS_F32 <= kGreedyThreads && DNP_K4_MULTI_THREADS_F64 <= kGreedyThreads, "the argmax slots are sized by kGreedyWaves");

template <typename F> __device__ __forceinline__ F ieee_sqrt(F x);
template <> __device__ __forceinline__ float ieee_sqrt<float>(float x) { return __builtin_sqrtf(x); }
template <> __device__ __forceinline__ double ieee_sqrt<double>(double x) { return __builtin_sqrt(x); }

template <typename F>
__device__ __forceinline__ void add_dipole_field(F sx, F sy, F sz, F px, F py, F pz, F x, F y, F z, F eps, F& ex,
                                                 F& ey, F& ez) {
    // one source -> one target, the leaf of field_utils.py:96-109 verbatim in IEEE arithmetic
    const F rx = sx - x, ry = sy - y, rz = sz - z;
    const F d2 = rx * rx + ry * ry + rz * rz;
    const F nrm = ieee_sqrt<F>(d2);
    F fx = F(0), fy = F(0), fz = F(0);
    if (nrm != F(0)) {
        const F ux = rx / nrm, uy = ry / nrm, uz = rz / nrm;
        const F c = F(3) * (px * ux + py * uy + pz * uz);
        fx = c * ux - px; fy = c * uy - py; fz = c * uz - pz;
    }
    const F den = nrm * nrm * nrm + eps;
    fx = fx / den; fy = fy / den; fz = fz / den;
    // E_total = E.sum(dim=0) * -1 ; Inf/NaN -> 0 (per call)
    fx = -fx; fy = -fy; fz = -fz;
    if (!__builtin_isfinite(fx)) fx = F(0);
    if (!__builtin_isfinite(fy)) fy = F(0);
    if (!__builtin_isfinite(fz)) fz = F(0);
    ex += fx; ey += fy; ez += fz;
}

// ---- candidate keys: an order in which "greater" = chosen first ------------------------------------------------
//   magnitude   |interaction| as its IEEE bit pattern (monotone for non-negative values)
//   low word    (0x7fffffff - point index) << 1 | (interaction < 0)     ties go to the smallest index, as torch.argmax
// fp32 packs both into ONE 64-bit word (so a workgroup's winner is one ds_max_u64 per lane); fp64 needs 96 bits.
constexpr unsigned kLowNone = 0u;   // the low word of a real key is >= 2 (index < 2^30)

__device__ __forceinline__ unsigned low_word(int idx, bool neg) {
    return ((0x7fffffffu - (unsigned)idx) << 1) | (neg ? 1u : 0u);
}
__device__ __forceinline__ int low_index(unsigned low) { return (int)(0x7fffffffu - (low >> 1)); }

template <typename F> struct Key;
template <> struct Key<float> {
    unsigned long long k;
    static __device__ __forceinline__ Key none() { return Key{0ull}; }
    static __device__ __forceinline__ Key make(float v, int idx) {
        const unsigned a = __builtin_bit_cast(unsigned, __builtin_fabsf(v));
        return Key{((unsigned long long)a << 32) | low_word(idx, v < 0.f)};
    }
    __device__ __forceinline__ bool valid() const { return k != 0ull; }
    __device__ __forceinline__ bool beats(const Key& o) const { return k > o.k; }
    __device__ __forceinline__ unsigned low() const { return (unsigned)(k & 0xffffffffull); }
    __device__ __forceinline__ float signed_value() const {
        const float a = __builtin_bit_cast(float, (unsigned)(k >> 32));
        return (k & 1ull) ? -a : a;
    }
    __device__ __forceinline__ Key xor_lane(int off) const { return Key{(unsigned long long)__shfl_xor(k, off, 64)}; }
};
template <> struct Key<double> {
    unsigned long long a;
    unsigned lo;
    static __device__ __forceinline__ Key none() { return Key{0ull, kLowNone}; }
    static __device__ __forceinline__ Key make(double v, int idx) {
        return Key{__builtin_bit_cast(unsigned long long, __builtin_fabs(v)), low_word(idx, v < 0.0)};
    }
    __device__ __forceinline__ bool valid() const { return lo != kLowNone; }
    __device__ __forceinline__ bool beats(const Key& o) const { return a > o.a || (a == o.a && lo > o.lo); }
    __device__ __forceinline__ unsigned low() const { return lo; }
    __device__ __forceinline__ double signed_value() const {
        const double m = __builtin_bit_cast(double, a);
        return (lo & 1u) ? -m : m;
    }
    __device__ __forceinline__ Key xor_lane(int off) const {
        return Key{(unsigned long long)__shfl_xor(a, off, 64), (unsigned)__shfl_xor(lo, off, 64)};
    }
};

template <int CTRL>
__device__ __forceinline__ unsigned dpp_u32(unsigned v) {
    // every lane has a source lane in these forms: no `old` value to keep (update_dpp(v, v, ...) costs a register copy and a wait
    // state per moved word - round 5, found in the patch greedy loop's ISA)
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, 0xf, 0xf, true);
}
template <int CTRL>
__device__ __forceinline__ unsigned long long dpp_u64(unsigned long long v) {
    return ((unsigned long long)dpp_u32<CTRL>((unsigned)(v >> 32)) << 32) | dpp_u32<CTRL>((unsigned)(v & 0xffffffffull));
}
template <int CTRL> __device__ __forceinline__ Key<float> key_dpp(const Key<float>& k) { return Key<float>{dpp_u64<CTRL>(k.k)}; }
template <int CTRL> __device__ __forceinline__ Key<double> key_dpp(const Key<double>& k) {
    return Key<double>{dpp_u64<CTRL>(k.a), dpp_u32<CTRL>(k.lo)};
}

// the wave's best key as a wave-uniform value: four butterfly rounds on DPP (quad_perm lane^1, lane^2; row_half_mirror;
// row_mirror) leave every 16-lane row's best in all of its lanes; the rows are folded with row_bcast:15 (rows 0 / 2 into
// rows 1 / 3) and row_bcast:31 (into rows 2 / 3) - all register to register; rounds 1-3 had crossed the rows through
// ds_bpermute, an LDS round trip per round on a step that is one latency chain - and lane 63 is read with v_readlane.
#ifndef DNP_K4_ROW_BCAST     // 0: A/B builds with the ds_bpermute rounds
#define DNP_K4_ROW_BCAST 1
#endif
template <int CTRL, int ROWMASK>
__device__ __forceinline__ unsigned dpp_u32_rows(unsigned v) {
    // rows the mask leaves out keep whatever the destination register held: only lane 63 is read afterwards, and what reaches it
    // (row 3 <- row 2's lane 47, then row 3 <- row 1's lane 31) is a written or an untouched key
    return (unsigned)__builtin_amdgcn_mov_dpp((int)v, CTRL, ROWMASK, 0xf, false);
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ Key<float> key_dpp_rows(const Key<float>& k) {
    return Key<float>{((unsigned long long)dpp_u32_rows<CTRL, ROWMASK>((unsigned)(k.k >> 32)) << 32) | dpp_u32_rows<CTRL, ROWMASK>((unsigned)(k.k & 0xffffffffull))};
}
template <int CTRL, int ROWMASK> __device__ __forceinline__ Key<double> key_dpp_rows(const Key<double>& k) {
    return Key<double>{((unsigned long long)dpp_u32_rows<CTRL, ROWMASK>((unsigned)(k.a >> 32)) << 32) | dpp_u32_rows<CTRL, ROWMASK>((unsigned)(k.a & 0xffffffffull)),
                       dpp_u32_rows<CTRL, ROWMASK>(k.lo)};
}
__device__ __forceinline__ unsigned readlane63(unsigned v) { return (unsigned)__builtin_amdgcn_readlane((int)v, 63); }
__device__ __forceinline__ Key<float> key_lane63(const Key<float>& k) {
    return Key<float>{((unsigned long long)readlane63((unsigned)(k.k >> 32)) << 32) | readlane63((unsigned)(k.k & 0xffffffffull))};
}
__device__ __forceinline__ Key<double> key_lane63(const Key<double>& k) {
    return Key<double>{((unsigned long long)readlane63((unsigned)(k.a >> 32)) << 32) | readlane63((unsigned)(k.a & 0xffffffffull)), readlane63(k.lo)};
}
template <typename F>
__device__ __forceinline__ Key<F> wave_best(Key<F> k) {
    Key<F> o = key_dpp<0xB1>(k); if (o.beats(k)) k = o;
    o = key_dpp<0x4E>(k); if (o.beats(k)) k = o;
    o = key_dpp<0x141>(k); if (o.beats(k)) k = o;
    o = key_dpp<0x140>(k); if (o.beats(k)) k = o;
#if DNP_K4_ROW_BCAST
    o = key_dpp_rows<0x142, 0xa>(k); if (o.beats(k)) k = o;
    o = key_dpp_rows<0x143, 0xc>(k); if (o.beats(k)) k = o;
    return key_lane63(k);
#else
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) {
        o = k.xor_lane(off);
        if (o.beats(k)) k = o;
    }
    return k;
#endif
}

// LDS of the workgroup argmax.  fp32: three words in rotation (word (step+1)%3 was last read right after the
// barrier of step-2, i.e. before every thread's arrival at the barrier of step-1, so thread 0 may clear it during
// step; it is next written after this step's barrier).  fp64: one slot per wave, two sets by step parity (a set is
// completely rewritten before its barrier, and its readers have passed the next barrier before that happens).
// Workgroup argmax, two forms.  0 (shipped): wave butterfly + one slot per wave in LDS.  1 (round 1, fp32 only): one
// ds_max_u64 per lane on a single word - 512 atomics on one address serialise: 4.5 us per step on ok.xyz against
// 3.1 us for the butterfly (tools/gpu_k4_ab.py, profiles/r02_greedy_time.txt).
#ifndef DNP_K4_ATOMIC
#define DNP_K4_ATOMIC 0
#endif
template <typename F> struct ArgmaxLds {
    unsigned long long word[3];                      // fp32 atomic form
    unsigned long long a[2][kGreedyWaves];           // butterfly form: one candidate per wave, two sets by step parity
    unsigned lo[2][kGreedyWaves];
};
template <typename F> constexpr bool use_atomic_argmax() { return sizeof(F) == 4 && DNP_K4_ATOMIC; }

template <typename F>
__device__ __forceinline__ void argmax_init(ArgmaxLds<F>& s) {
    if constexpr (use_atomic_argmax<F>()) {
        if (threadIdx.x == 0) s.word[0] = s.word[1] = s.word[2] = 0ull;
    }
}

template <typename F> __device__ __forceinline__ void key_store(ArgmaxLds<F>& s, int par, int wave, const Key<F>& k);
template <> __device__ __forceinline__ void key_store<float>(ArgmaxLds<float>& s, int par, int wave, const Key<float>& k) {
    s.a[par][wave] = k.k;
}
template <> __device__ __forceinline__ void key_store<double>(ArgmaxLds<double>& s, int par, int wave, const Key<double>& k) {
    s.a[par][wave] = k.a; s.lo[par][wave] = k.lo;
}
template <typename F> __device__ __forceinline__ Key<F> key_load(const ArgmaxLds<F>& s, int par, int q);
template <> __device__ __forceinline__ Key<float> key_load<float>(const ArgmaxLds<float>& s, int par, int q) {
    return Key<float>{s.a[par][q]};
}
template <> __device__ __forceinline__ Key<double> key_load<double>(const ArgmaxLds<double>& s, int par, int q) {
    return Key<double>{s.a[par][q], s.lo[par][q]};
}

// every thread contributes `mine`; returns the workgroup's best key to every thread.  Contains ONE barrier.
// Atomic form (fp32): three words in rotation - word (step+1)%3 was last read right after the barrier of step-2,
// i.e. before every thread's arrival at the barrier of step-1, so thread 0 may clear it during step; it is next
// written after this step's barrier.  Butterfly form: a set of wave slots is completely rewritten before its
// barrier, and its readers have passed the next barrier before that happens, so two sets by step parity suffice.
template <typename F, int WAVES = kGreedyWaves>
__device__ __forceinline__ Key<F> workgroup_best(ArgmaxLds<F>& s, Key<F> mine, int64_t step) {
    if constexpr (use_atomic_argmax<F>()) {
        const int par = (int)(step % 3);
        if (threadIdx.x == 0) s.word[(par + 1) % 3] = 0ull;
        if (mine.valid()) atomicMax(&s.word[par], mine.k);
        __syncthreads();
        return Key<float>{s.word[par]};
    } else {
        const int par = (int)(step & 1), lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        const Key<F> w = wave_best<F>(mine);
        if (lane == 0) key_store<F>(s, par, wave, w);
        __syncthreads();
        Key<F> best = Key<F>::none();
#pragma unroll
        for (int q = 0; q < WAVES; ++q) {
            const Key<F> c = key_load<F>(s, par, q);
            if (c.beats(best)) best = c;
        }
        return best;
    }
}

template <typename F, int PPT>
__global__ __launch_bounds__(kGreedyThreads) void point_greedy_kernel(const F* __restrict__ pts, int64_t N,
                                                                      int64_t ld, int start, F eps, int diffuse,
                                                                      int64_t* __restrict__ order_out,
                                                                      F* __restrict__ E_out,
                                                                      F* __restrict__ n_out) {
    const int tid = threadIdx.x;
    __shared__ ArgmaxLds<F> am;
    argmax_init<F>(am);
    __syncthreads();

    F x[PPT], y[PPT], z[PPT], nx[PPT], ny[PPT], nz[PPT], ex[PPT], ey[PPT], ez[PPT];
    unsigned visited = 0, flipped = 0, valid = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = (int64_t)k * kGreedyThreads + tid;
        x[k] = y[k] = z[k] = nx[k] = ny[k] = nz[k] = F(0);
        ex[k] = ey[k] = ez[k] = F(0);
        if (i < N) {
            const F* p = pts + i * ld;
            x[k] = p[0]; y[k] = p[1]; z[k] = p[2]; nx[k] = p[3]; ny[k] = p[4]; nz[k] = p[5];
            valid |= 1u << k;
        }
    }
    visited = ~valid;  // slots past N never take part

    int cur = start;          // wave-uniform
    F cur_sign = F(1);        // the start point is not flipped
    for (int64_t step = 0; step < N; ++step) {
        // (3)+(4): add the field of point `cur` (with its possibly flipped normal) to every point
        {
            const F* p = pts + (int64_t)cur * ld;
            const F sx = p[0], sy = p[1], sz = p[2];
            const F px = p[3] * cur_sign, py = p[4] * cur_sign, pz = p[5] * cur_sign;
            const int ck = cur / kGreedyThreads, ct = cur - ck * kGreedyThreads;
            if (tid == ct) {
                visited |= 1u << ck;
                if (cur_sign < F(0)) flipped |= 1u << ck;
            }
            if (order_out && tid == 0) order_out[step] = cur;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool self = (tid == ct) && (k == ck);   // E[~(indx == i)]: the source itself is skipped
                if (((valid >> k) & 1u) && !self)
                    add_dipole_field<F>(sx, sy, sz, px, py, pz, x[k], y[k], z[k], eps, ex[k], ey[k], ez[k]);
            }
        }
        if (step + 1 == N) break;

        // (1) local scan -> (2) workgroup argmax, ONE barrier per step
        Key<F> key = Key<F>::none();
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!((visited >> k) & 1u)) {
                // an unvisited point has not been flipped: its normal is the input normal
                const F v = ex[k] * nx[k] + ey[k] * ny[k] + ez[k] * nz[k];
                const Key<F> c = Key<F>::make(v, k * kGreedyThreads + tid);
                if (c.beats(key)) key = c;
            }
        }
        const Key<F> gk = workgroup_best<F>(am, key, step);
        cur = __builtin_amdgcn_readfirstlane(low_index(gk.low()));
        cur_sign = (gk.low() & 1u) ? F(-1) : F(1);           // `if interaction[max] < 0: flip`
    }

    // epilogue: flips, optional diffuse sign pass (field_utils.py:382-385), E_out
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = (int64_t)k * kGreedyThreads + tid;
        if (i < N) {
            const F s = ((flipped >> k) & 1u) ? F(-1) : F(1);
            F a = nx[k] * s, b2 = ny[k] * s, c = nz[k] * s;
            if (diffuse) {
                const F v = ex[k] * a + ey[k] * b2 + ez[k] * c;
                const F sg = (v > F(0)) ? F(1) : F(-1);
                a *= sg; b2 *= sg; c *= sg;
            }
            n_out[i * 3 + 0] = a; n_out[i * 3 + 1] = b2; n_out[i * 3 + 2] = c;
            if (E_out) { E_out[i * 3 + 0] = ex[k]; E_out[i * 3 + 1] = ey[k]; E_out[i * 3 + 2] = ez[k]; }
        }
    }
}

// stream-ordered behind the persistent kernel: pts[:, 3:6] = n_out - unless the persistent kernel gave up
// (status[0] != 0: a workgroup timed out waiting for its peers; n_out then holds a partial propagation and pts must
// stay the caller's input, because the host falls back to step-wise launches on it)
template <typename F>
__global__ __launch_bounds__(256) void store_normals_kernel(F* __restrict__ pts, int64_t ld,
                                                            const F* __restrict__ n_out, int64_t N,
                                                            const int* __restrict__ status) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N || status[0] != 0) return;
    F* p = pts + i * ld;
    p[3] = n_out[i * 3 + 0]; p[4] = n_out[i * 3 + 1]; p[5] = n_out[i * 3 + 2];
}


// ---- multi-workgroup form --------------------------------------------------------------------------------------
// G workgroups (one per CU, all co-resident: the launcher checks the grid against the occupancy query), each owning a contiguous slice of the cloud in registers.
// Per step every workgroup publishes its local winner as naturally aligned 8-byte granules
//   fp32:  { float signed_interaction ; uint32 (step & 0xfff) << 20 | point_index }          (index < 2^20)
//   fp64:  two granules { high / low 32 bits of the signed interaction ; the same tag|index word }
// with relaxed agent-scope stores (payload and tag travel together, so no release/acquire pair is needed:
// MI355X_MICROARCH.md "granule"), and wave 0 of every workgroup polls the G granules of the step with
// relaxed agent-scope loads (they bypass the per-CU L1) until all carry the step's tag - an all-gather,
// not a barrier.  Slots are double-buffered by step parity: a workgroup writes its step-(n+1) slot only
// after it has read every step-n slot, i.e. after everybody has finished reading the step-(n-1) slots it
// overwrites.  Every spin is bounded: on timeout the workgroup raises status[0] and every workgroup leaves the loop.
constexpr unsigned kTagShift = 20;
constexpr unsigned kIdxMask = (1u << kTagShift) - 1;
constexpr int kMaxGroups = 256;
// 8-byte words per workgroup in a row of granule slots: its granule word(s) side by side, then padding.  Round 5 (tools/gpu_k4_ab.py,
// ok.xyz): fp32 granules one 8-byte word apart shared three cache lines between 40 writers on eight XCDs - with 128 bytes per
// workgroup (16) and 256-thread workgroups the 10 000 steps take 22.3 ms against 26.2 (512 threads, dense: the round-4 form; dense
// with 256 threads 30.9); 32 or 64 words level, 4 words 23.0, 2 words 25.7.  fp64 (two words per workgroup, now adjacent instead of
// two rows 2 KB apart) is best dense - 23.4 ms against 25.2 with any padding.
#ifndef DNP_K4_SLOT_PAD_F32
#define DNP_K4_SLOT_PAD_F32 16
#endif
#ifndef DNP_K4_SLOT_PAD_F64
#define DNP_K4_SLOT_PAD_F64 2
#endif
template <typename F> constexpr int slot_stride() { return sizeof(F) == 8 ? DNP_K4_SLOT_PAD_F64 : DNP_K4_SLOT_PAD_F32; }
constexpr int kMaxSlotStride = DNP_K4_SLOT_PAD_F32 > DNP_K4_SLOT_PAD_F64 ? DNP_K4_SLOT_PAD_F32 : DNP_K4_SLOT_PAD_F64;
static_assert(DNP_K4_SLOT_PAD_F32 >= 1 && DNP_K4_SLOT_PAD_F64 >= 2, "a workgroup's granule words sit side by side");

template <typename F>
struct MultiArgs {
    const F* pts; int64_t N; int64_t ld; int start; F eps; int diffuse;
    int64_t* order_out; F* E_out; F* n_out;
    unsigned long long* slots;   // [2 parities][words per group][kMaxGroups]
    int* status;                 // [0] = 0 ok, 1 timeout
    int per_group;               // points per workgroup (multiple of 512)
#ifdef DNP_K4_STATS              // instrumented builds only (tools/gpu_k4_spread.py): where does a step's time go, and at what clock
    unsigned long long* stats;   // [N][4] per step, by workgroup 0: {wall clock at the step's end (100 MHz), shader clock counter, spins of its slowest polling lane, wall clock when it had published};
#endif                           // then [G][2] per workgroup: {HW_ID, XCC_ID}
};

template <typename F> struct Granule;
template <> struct Granule<float> {
    static constexpr int kWords = 1;
    static __device__ __forceinline__ unsigned long long pack(float v, unsigned tagidx, int) {
        return ((unsigned long long)tagidx << 32) | (unsigned long long)__builtin_bit_cast(unsigned, v);
    }
    static __device__ __forceinline__ float value(const unsigned long long* g) {
        return __builtin_bit_cast(float, (unsigned)(g[0] & 0xffffffffull));
    }
};
template <> struct Granule<double> {
    static constexpr int kWords = 2;
    static __device__ __forceinline__ unsigned long long pack(double v, unsigned tagidx, int w) {
        const unsigned long long bits = __builtin_bit_cast(unsigned long long, v);
        const unsigned half = w == 0 ? (unsigned)(bits >> 32) : (unsigned)(bits & 0xffffffffull);
        return ((unsigned long long)tagidx << 32) | (unsigned long long)half;
    }
    static __device__ __forceinline__ double value(const unsigned long long* g) {
        return __builtin_bit_cast(double, ((g[0] & 0xffffffffull) << 32) | (g[1] & 0xffffffffull));
    }
};

template <typename F, int PPT>
__global__ __launch_bounds__(multi_threads<F>()) void point_greedy_multi_kernel(const MultiArgs<F> a) {
    constexpr int kT = multi_threads<F>();                 // threads of this form's workgroups (by precision, see above)
    constexpr int kW = Granule<F>::kWords;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int G = gridDim.x, g = blockIdx.x;
    const int64_t base = (int64_t)g * a.per_group;
    __shared__ ArgmaxLds<F> am;
    __shared__ F win_row[8];                        // winner's (x, y, z, nx, ny, nz, sign)
    __shared__ int win_idx;
    __shared__ int abort_flag;

    F x[PPT], y[PPT], z[PPT], nx[PPT], ny[PPT], nz[PPT], ex[PPT], ey[PPT], ez[PPT];
    unsigned visited = 0, flipped = 0, valid = 0;
#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = base + (int64_t)k * kT + tid;
        x[k] = y[k] = z[k] = nx[k] = ny[k] = nz[k] = F(0);
        ex[k] = ey[k] = ez[k] = F(0);
        if (i < a.N && i < base + a.per_group) {
            const F* p = a.pts + i * a.ld;
            x[k] = p[0]; y[k] = p[1]; z[k] = p[2]; nx[k] = p[3]; ny[k] = p[4]; nz[k] = p[5];
            valid |= 1u << k;
        }
    }
    visited = ~valid;
    argmax_init<F>(am);
#ifdef DNP_K4_STATS
    if (tid == 0 && a.stats) {
        unsigned h, x;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %1, hwreg(HW_REG_XCC_ID)" : "=s"(h), "=s"(x));
        a.stats[(size_t)a.N * 4 + (size_t)g * 2] = h;
        a.stats[(size_t)a.N * 4 + (size_t)g * 2 + 1] = x;
    }
#endif
    if (tid == 0) {
        abort_flag = 0;
        const F* p = a.pts + (int64_t)a.start * a.ld;
        for (int c = 0; c < 6; ++c) win_row[c] = p[c];
        win_row[6] = F(1);                           // the start point is not flipped
        win_idx = a.start;
    }
    __syncthreads();

    for (int64_t step = 0; step < a.N; ++step) {
        // the chosen point's row comes from LDS: wave 0 fetched it while it was polling (or the prologue did)
        const int cur = win_idx;
        const F cur_sign = (win_row[6] < F(0)) ? F(-1) : F(1);     // `if interaction[max] < 0: flip`
        {
            const F sx = win_row[0], sy = win_row[1], sz = win_row[2];
            const F px = win_row[3] * cur_sign, py = win_row[4] * cur_sign, pz = win_row[5] * cur_sign;
            const int64_t rel = (int64_t)cur - base;
            const bool mine = rel >= 0 && rel < a.per_group;
            const int ck = mine ? (int)(rel / kT) : -1, ct = mine ? (int)(rel - (int64_t)ck * kT) : -1;
            if (tid == ct) {
                visited |= 1u << ck;
                if (cur_sign < F(0)) flipped |= 1u << ck;
            }
            // (this per-step global store is not on the critical path: a build without it runs the 10 000 steps of ok.xyz in 28.34
            // against 28.52 ms, round 4)
            if (a.order_out && g == 0 && tid == 0) a.order_out[step] = cur;
#pragma unroll
            for (int k = 0; k < PPT; ++k) {
                const bool self = (tid == ct) && (k == ck);
                if (((valid >> k) & 1u) && !self)
                    add_dipole_field<F>(sx, sy, sz, px, py, pz, x[k], y[k], z[k], a.eps, ex[k], ey[k], ez[k]);
            }
        }
        if (step + 1 == a.N) break;

        // local winner (the barrier inside also orders this step's win_row reads before its rewrite)
        Key<F> key = Key<F>::none();
#pragma unroll
        for (int k = 0; k < PPT; ++k) {
            if (!((visited >> k) & 1u)) {
                const F v = ex[k] * nx[k] + ey[k] * ny[k] + ez[k] * nz[k];
                const Key<F> c = Key<F>::make(v, (int)(base + (int64_t)k * kT + tid));
                if (c.beats(key)) key = c;
            }
        }
        const Key<F> lk = workgroup_best<F, kT / 64>(am, key, step);
        const int par = (int)(step & 1);
        if (wave == 0) {
            const unsigned tag = (unsigned)(step & 0xfff);
            constexpr int kStride = slot_stride<F>();                       // 8-byte words per workgroup: its granule word(s), then padding
            unsigned long long* row = a.slots + (size_t)par * kMaxGroups * kStride;
            if (lane < kW) {
                // re-pack the key as granule(s): { value bits ; tag << 20 | index }   (index kIdxMask = no candidate)
                unsigned idx = kIdxMask;
                F v = F(0);
                if (lk.valid()) { idx = (unsigned)low_index(lk.low()); v = lk.signed_value(); }
                const unsigned long long gran = Granule<F>::pack(v, (tag << kTagShift) | idx, lane);
                __hip_atomic_store(row + (size_t)g * kStride + lane, gran, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            // all-gather: lane l polls groups l, l+64, ...; as soon as a group's candidate is known its row is
            // requested, so the winner's row is already on its way when the argmax is done
            Key<F> best = Key<F>::none();
            F r0 = F(0), r1 = F(0), r2 = F(0), r3 = F(0), r4 = F(0), r5 = F(0);
            bool timed_out = false;
#ifdef DNP_K4_STATS
            unsigned long long k4_spins = 0, k4_t_pub = 0;
            asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(k4_t_pub));     // published: the wait for the others starts
#endif
            for (int q = lane; q < G; q += 64) {
                unsigned long long gran[kW];
                unsigned spins = 0;
                for (;;) {
                    bool ready = true;
#pragma unroll
                    for (int w = 0; w < kW; ++w) {
                        gran[w] = __hip_atomic_load(row + (size_t)q * kStride + w, __ATOMIC_RELAXED,
                                                    __HIP_MEMORY_SCOPE_AGENT);
                        ready = ready && ((unsigned)(gran[w] >> (32 + kTagShift)) == tag);
                    }
                    if (ready) break;
                    if (++spins > (1u << 22) ||
                        __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != 0) {
                        timed_out = true;
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
                if (timed_out) break;
#ifdef DNP_K4_STATS
                k4_spins += spins;
#endif
                const unsigned idx = (unsigned)(gran[0] >> 32) & kIdxMask;
                if (idx != kIdxMask) {
                    const Key<F> c = Key<F>::make(Granule<F>::value(gran), (int)idx);
                    if (c.beats(best)) {
                        best = c;
                        const F* p = a.pts + (int64_t)idx * a.ld;       // pts is never written by this kernel
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
#ifdef DNP_K4_STATS
            {
                unsigned long long smax = k4_spins;          // the slowest lane's spins: how long this workgroup waited for the others
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) { const unsigned long long o = __shfl_xor(smax, off, 64); smax = o > smax ? o : smax; }
                if (g == 0 && lane == 0 && a.stats) {
                    unsigned long long t_wall, t_core;
                    asm volatile("s_memrealtime %0\n\ts_memtime %1\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_wall), "=s"(t_core));
                    unsigned long long* q = a.stats + (size_t)step * 4;
                    q[0] = t_wall; q[1] = t_core; q[2] = smax; q[3] = k4_t_pub;
                }
            }
#endif
            const Key<F> wbest = wave_best<F>(best);
            if (best.valid() && !best.beats(wbest) && !wbest.beats(best)) {      // exactly one lane: keys are unique per point
                win_row[0] = r0; win_row[1] = r1; win_row[2] = r2; win_row[3] = r3; win_row[4] = r4; win_row[5] = r5;
                win_row[6] = (best.low() & 1u) ? F(-1) : F(1);
                win_idx = low_index(best.low());
            }
        }
        __syncthreads();
        if (abort_flag) break;
    }

#pragma unroll
    for (int k = 0; k < PPT; ++k) {
        const int64_t i = base + (int64_t)k * kT + tid;
        if ((valid >> k) & 1u) {
            const F s = ((flipped >> k) & 1u) ? F(-1) : F(1);
            F c0 = nx[k] * s, c1 = ny[k] * s, c2 = nz[k] * s;
            if (a.diffuse) {
                const F v = ex[k] * c0 + ey[k] * c1 + ez[k] * c2;
                const F sg = (v > F(0)) ? F(1) : F(-1);
                c0 *= sg; c1 *= sg; c2 *= sg;
            }
            a.n_out[i * 3 + 0] = c0; a.n_out[i * 3 + 1] = c1; a.n_out[i * 3 + 2] = c2;
            if (a.E_out) { a.E_out[i * 3 + 0] = ex[k]; a.E_out[i * 3 + 1] = ey[k]; a.E_out[i * 3 + 2] = ez[k]; }
        }
    }
}

// points per lane the two precisions can hold in a 256-VGPR budget (9 values per point)
template <typename F> struct GreedyCap;
template <> struct GreedyCap<float> { static constexpr int kMaxPPT = 20; };
template <> struct GreedyCap<double> { static constexpr int kMaxPPT = 8; };

constexpr size_t kGreedyHeader = 256;                                         // status word (+ padding)
constexpr size_t kGreedySlots = (size_t)2 * kMaxGroups * kMaxSlotStride * sizeof(unsigned long long);   // 2 parities x groups x words per group

#ifdef DNP_K4_STATS
static unsigned long long* g_k4_stats = nullptr;      // set by dnp_debug_set_k4_stats: [N][4] + [groups][2] words (device)
#endif

template <typename F>
static int run_point_greedy(F* pts, int64_t N, int64_t ld_pts, int64_t start, F eps, int diffuse, int64_t* order_out,
                            F* E_out, int form, int max_groups, void* workspace, size_t workspace_bytes,
                            hipStream_t st) {
    constexpr int kMaxPPT = GreedyCap<F>::kMaxPPT;
    clear_error();
    DNP_REQUIRE(N >= 0, "negative N");
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(pts, "NULL pts");
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    DNP_REQUIRE(start >= 0 && start < N, "starting_point %lld out of range [0,%lld)", (long long)start, (long long)N);
    DNP_REQUIRE(N < (int64_t)kIdxMask, "N=%lld exceeds the %u points of the persistent per-point kernel",
                (long long)N, kIdxMask - 1);
    DNP_REQUIRE(form >= 0 && form <= 3, "form=%d (0 auto, 1 single workgroup, 2 one workgroup per CU, 3 = 2 with the "
                "time-out raised before the launch: test hook for the abort path)", form);
    const size_t need = kGreedyHeader + kGreedySlots + (size_t)N * 3 * sizeof(F);
    if (!workspace || workspace_bytes < need) {
        set_error("workspace of %zu bytes required, %zu given", need, workspace ? workspace_bytes : (size_t)0);
        return DNP_EWORKSPACE;
    }
    int* status = (int*)workspace;
    unsigned long long* slots = (unsigned long long*)((char*)workspace + kGreedyHeader);
    F* n_out = (F*)((char*)workspace + kGreedyHeader + kGreedySlots);
    // Form selection: one workgroup keeps everything in registers but pays for every point per lane and step
    // (IEEE div/sqrt chain: 1.1 us per step at 256 points, 2.4 at 2048, 4.2 at 4096 in fp32); one workgroup per CU
    // pays 2.2-2.4 us per step (3 in fp64) for the granule all-gather whatever the size.  Measured crossover
    // (tools/gpu_k4_forms.py): just below 2048 points in fp32, ~1500 in fp64.
    // (round 5, after the multi-workgroup form's step went to 2.2-2.3 us in both precisions: fp64 crosses at ~1024 points - 1280:
    // 2.59 against 2.29 us per step, 1536: 2.90 against 2.34 -, fp32 still just below 1800)
    bool multi = (form >= 2) || (form == 0 && N > (sizeof(F) == 8 ? 1024 : 1792));
    if (form == 1) DNP_REQUIRE(N <= (int64_t)kGreedyThreads * kMaxPPT, "N=%lld exceeds the %d points of the single-workgroup form",
                               (long long)N, kGreedyThreads * kMaxPPT);
    if (N > (int64_t)kGreedyThreads * kMaxPPT) multi = true;
    DNP_CHECK_HIP(hipMemsetAsync(workspace, 0, kGreedyHeader, st));   // status word: 0 = ok (only the multi form sets it)
    if (!multi) {
#define DNP_LAUNCH_GREEDY(P)                                                                                     \
    hipLaunchKernelGGL((point_greedy_kernel<F, P>), dim3(1), dim3(kGreedyThreads), 0, st, (const F*)pts, N, ld_pts, \
                       (int)start, eps, diffuse, order_out, E_out, n_out)
        if (N <= kGreedyThreads * 2) DNP_LAUNCH_GREEDY(2);
        else if (N <= kGreedyThreads * 4) DNP_LAUNCH_GREEDY(4);
        else if (N <= kGreedyThreads * 8) DNP_LAUNCH_GREEDY(8);
        else if constexpr (kMaxPPT >= 20) {
            if (N <= kGreedyThreads * 16) DNP_LAUNCH_GREEDY(16);
            else DNP_LAUNCH_GREEDY(20);
        }
#undef DNP_LAUNCH_GREEDY
        DNP_CHECK_HIP(hipGetLastError());
    } else {
        int dev = 0, cus = 0;
        DNP_CHECK_HIP(hipGetDevice(&dev));
        DNP_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
        int groups = cus < kMaxGroups ? cus : kMaxGroups;          // one resident workgroup per CU
        if (max_groups > 0 && max_groups < groups) groups = max_groups;
        constexpr int kT = multi_threads<F>();
        int64_t per = ceil_div(ceil_div(N, (int64_t)groups), (int64_t)kT) * kT;
        groups = (int)ceil_div(N, per);
        const int ppt = (int)(per / kT);
        DNP_REQUIRE(ppt <= kMaxPPT, "N=%lld needs %d points per lane on %d CUs (max %d)", (long long)N, ppt, groups,
                    kMaxPPT);
        // tags start at step 0: every slot must hold a tag that no early step uses
        DNP_CHECK_HIP(hipMemsetAsync(slots, 0xff, kGreedySlots, st));
        // form 3: the status word says "timed out" before the kernel starts - the first workgroup that has to wait for a
        // granule leaves, then all do; what a time-out in the field looks like, on demand (tests)
        if (form == 3) DNP_CHECK_HIP(hipMemsetAsync(status, 1, 1, st));
        MultiArgs<F> ma{pts, N, ld_pts, (int)start, eps, diffuse, order_out, E_out, n_out, slots, status, (int)per};
#ifdef DNP_K4_STATS
        ma.stats = g_k4_stats;
#endif
        // Co-residency: the all-gather may only wait for workgroups that are on the chip.  One workgroup per CU at
        // most, and the occupancy query must confirm that a CU holds one (the check hipLaunchCooperativeKernel would
        // make; the cooperative launch itself is avoided because rocprofv3 (ROCm 7.2) crashes at exit in a process
        // that used it).  Every spin in the kernel is bounded besides.
#define DNP_LAUNCH_MULTI(P)                                                                                         \
    do {                                                                                                            \
        int per_cu = 0;                                                                                             \
        DNP_CHECK_HIP(hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, point_greedy_multi_kernel<F, P>,        \
                                                                   kT, 0));                                         \
        DNP_REQUIRE(per_cu >= 1 && groups <= cus, "%d workgroups of the per-point kernel cannot be co-resident on " \
                    "%d CUs (%d per CU)", groups, cus, per_cu);                                                     \
        hipLaunchKernelGGL((point_greedy_multi_kernel<F, P>), dim3(groups), dim3(kT), 0, st, ma);                   \
        DNP_CHECK_HIP(hipGetLastError());                                                                           \
    } while (0)
        if (ppt <= 1) DNP_LAUNCH_MULTI(1);
        else if (ppt <= 2) DNP_LAUNCH_MULTI(2);
        else if (ppt <= 4) DNP_LAUNCH_MULTI(4);
        else if (ppt <= 8) DNP_LAUNCH_MULTI(8);
        else if constexpr (kMaxPPT >= 20) {
            if (ppt <= 16) DNP_LAUNCH_MULTI(16);
            else DNP_LAUNCH_MULTI(20);
        }
#undef DNP_LAUNCH_MULTI
    }
    hipLaunchKernelGGL((store_normals_kernel<F>), dim3((unsigned)ceil_div(N, 256)), dim3(256), 0, st, pts, ld_pts,
                       (const F*)n_out, N, (const int*)status);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

}  // namespace dnp

using namespace dnp;

extern "C" {

size_t dnp_point_greedy_workspace_bytes(int64_t N, int elem_size) {
    if (N < 0) N = 0;
    // [0,256): status word (+ padding); the granule slots of the multi-workgroup form; the oriented normals
    return kGreedyHeader + kGreedySlots + (size_t)N * 3 * (size_t)(elem_size == 8 ? 8 : 4);
}

int dnp_point_greedy_max_points(void) { return (int)kIdxMask; }   // index field of the granule: N < 2^20

#ifdef DNP_K4_STATS
int dnp_debug_set_k4_stats(void* p) { dnp::g_k4_stats = (unsigned long long*)p; return 0; }
#endif

int dnp_point_greedy_f32(float* pts, int64_t N, int64_t ld_pts, int64_t start, float eps, int diffuse,
                         int64_t* order_out, float* E_out, int form, int max_groups, void* workspace,
                         size_t workspace_bytes, void* stream) {
    return run_point_greedy<float>(pts, N, ld_pts, start, eps, diffuse, order_out, E_out, form, max_groups, workspace,
                                   workspace_bytes, (hipStream_t)stream);
}

int dnp_point_greedy_f64(double* pts, int64_t N, int64_t ld_pts, int64_t start, double eps, int diffuse,
                         int64_t* order_out, double* E_out, int form, int max_groups, void* workspace,
                         size_t workspace_bytes, void* stream) {
    return run_point_greedy<double>(pts, N, ld_pts, start, eps, diffuse, order_out, E_out, form, max_groups,
                                    workspace, workspace_bytes, (hipStream_t)stream);
}

}  // extern "C"
