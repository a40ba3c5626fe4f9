// pair_kernel.h - the all-pairs dipole kernel for CDNA4 (gfx950), shared by the field_grad /
// potential entry points (dnp_field.hip) and the batched per-patch entry point (dnp_patch.hip).
//
// Work decomposition
//   grid.x = target tiles (BLOCK threads x KT targets per lane), grid.y = source chunks.
//   A block streams its chunk's sources through LDS in tiles of BLOCK rows (double-buffered:
//   the next tile's global loads are in flight while the current tile is consumed) and every
//   lane accumulates KT targets in registers.  All 64 lanes of a wave read the SAME LDS
//   address per source (hardware broadcast, conflict-free): one ds_read_b128 + one ds_read_b64
//   feed KT x 28 VALU issue slots.  Output: partial[chunk][target][NC], reduced by a second,
//   tiny kernel - deterministic (no float atomics), and the natural place for the reference's
//   per-leaf Inf/NaN filter (field_utils.py:110-115).
//
// Arithmetic per pair (field mode), 33 flops in 24 full-rate + 2 half-rate VALU instructions:
//   r = x_s - x_t; d2 = r.r; inv = rsq(d2); d = d2*inv; w = rcp(d2*d + eps)
//   a = (p.r) * inv^2 * w;   A += a*r;  B += w*p;      E = -(3A - B)
// which is field_utils.py:96-109 with r^ = r*inv folded in.  |r| == 0 is handled by
// substituting d2 := 1e30 (one v_cmp + one v_cndmask): then d2*d overflows to +inf, w = 0,
// a = 0 and the pair contributes exactly 0, as `E[zero_mask] = 0` does.
//
// Accumulation: fp32 inside a run of FLUSH sources, fp64 across runs (one cvt+fma per FLUSH
// sources, free) - the reference's own sum is a cascade sum (torch CPU), so a plain fp32 chain
// over 10^5 terms would not stay within 1e-5 of it.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace dnp {

constexpr int kBlock = 256;       // threads per workgroup (4 waves, one per SIMD)
constexpr int kFlush = 128;       // fp32 chain length before spilling into the fp64 sums
constexpr int kMaxChunks = 128;   // by-value chunk table entries per launch

enum PairMode { kField = 0, kPotential = 1 };

template <typename F>
struct PairArgs {
    const F* src;            // [*, ld_src] rows (x,y,z,px,py,pz)
    int64_t ld_src;
    const int64_t* src_idx;  // row gather for sources, or nullptr
    const F* tgt;            // [*, ld_tgt] rows (x,y,z,...)
    int64_t ld_tgt;
    const int64_t* tgt_idx;  // row gather for targets, or nullptr
    int64_t T;
    const int64_t* chunk_off_dev;  // device CSR offsets (patch mode), or nullptr
    int64_t chunk_base;            // first chunk handled by this launch (index into chunk_off_dev)
    const int64_t* tgt_group;      // per target-row group id; rows whose group == chunk id get 0
    F eps;
    F* partial;              // [gridDim.y][T][NC]
    int32_t chunk_off[kMaxChunks + 1];  // by-value CSR offsets when chunk_off_dev == nullptr
};

// ---- per-pair arithmetic ----------------------------------------------------------------
template <typename F> struct Math;
template <> struct Math<float> {
    static __device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static constexpr float kHuge = 1e30f;   // (1e30)^1.5 overflows fp32 -> w = rcp(inf) = 0
    static constexpr float kFar = 1e18f;    // padding source position: d2 = 3e36 (finite), d^3 = inf
};
template <> struct Math<double> {
    static __device__ __forceinline__ double rsq(double x) { return 1.0 / __builtin_sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static constexpr double kHuge = 1e250;  // (1e250)^1.5 overflows fp64
    static constexpr double kFar = 1e150;   // d2 = 3e300 finite, d^3 = inf
};

template <typename F, bool NAN_COINC>
__device__ __forceinline__ void pair_field(F sx, F sy, F sz, F px, F py, F pz, F tx, F ty, F tz, F eps,
                                           F& ax, F& ay, F& az, F& bx, F& by, F& bz) {
    using M = Math<F>;
    const F rx = sx - tx, ry = sy - ty, rz = sz - tz;
    const F d2 = M::fma(rz, rz, M::fma(ry, ry, rx * rx));
    const bool coinc = (d2 == F(0));
    const F d2m = coinc ? M::kHuge : d2;
    const F inv = M::rsq(d2m);
    const F d = d2m * inv;
    F w = M::rcp(M::fma(d2m, d, eps));
    if (NAN_COINC) w = coinc ? __builtin_nanf("") : w;   // eps == 0: the reference's 0/0
    const F pr = M::fma(pz, rz, M::fma(py, ry, px * rx));
    const F a = pr * (inv * inv * w);
    ax = M::fma(a, rx, ax);
    ay = M::fma(a, ry, ay);
    az = M::fma(a, rz, az);
    bx = M::fma(w, px, bx);
    by = M::fma(w, py, by);
    bz = M::fma(w, pz, bz);
}

template <typename F>
__device__ __forceinline__ void pair_potential(F sx, F sy, F sz, F px, F py, F pz, F tx, F ty, F tz, F& phi) {
    using M = Math<F>;
    const F rx = sx - tx, ry = sy - ty, rz = sz - tz;
    const F d2 = M::fma(rz, rz, M::fma(ry, ry, rx * rx));
    const F inv = M::rsq(d2);                       // d2 == 0 -> inf, pr == 0 -> 0*inf = NaN (as 0/0)
    const F pr = M::fma(pz, rz, M::fma(py, ry, px * rx));
    phi = M::fma(pr, inv * inv * inv, phi);
}

// LDS image of one staged source row: two 16-byte slots, (x,y,z,px) and (py,pz,-,-).
template <typename F> struct Vec4 { F x, y, z, w; };

template <typename F, int MODE, int KT, bool NAN_COINC>
__global__ __launch_bounds__(kBlock) void pair_kernel(const PairArgs<F> a) {
    using M = Math<F>;
    constexpr int NC = (MODE == kField) ? 3 : 1;
    __shared__ __attribute__((aligned(16))) Vec4<F> lds[2][kBlock][2];

    const int tid = threadIdx.x;
    const int64_t chunk = blockIdx.y;
    int64_t s_begin, s_end;
    if (a.chunk_off_dev) {
        s_begin = a.chunk_off_dev[a.chunk_base + chunk];
        s_end = a.chunk_off_dev[a.chunk_base + chunk + 1];
    } else {
        s_begin = a.chunk_off[chunk];
        s_end = a.chunk_off[chunk + 1];
    }
    const int64_t tile_base = (int64_t)blockIdx.x * (kBlock * KT);

    // ---- this lane's targets ---------------------------------------------------------------
    F tx[KT], ty[KT], tz[KT];
    int64_t trow[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int64_t t = tile_base + k * kBlock + tid;
        trow[k] = -1;
        tx[k] = ty[k] = tz[k] = F(0);
        if (t < a.T) {
            const int64_t row = a.tgt_idx ? a.tgt_idx[t] : t;
            trow[k] = row;
            const F* p = a.tgt + row * a.ld_tgt;
            tx[k] = p[0]; ty[k] = p[1]; tz[k] = p[2];
        }
    }

    double acc[KT][NC];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[k][c] = 0.0;

    // ---- staging helpers -------------------------------------------------------------------
    F g0, g1, g2, g3, g4, g5;
    auto load_row = [&](int64_t s) {
        if (s < s_end) {
            const int64_t row = a.src_idx ? a.src_idx[s] : s;
            const F* p = a.src + row * a.ld_src;
            g0 = p[0]; g1 = p[1]; g2 = p[2]; g3 = p[3]; g4 = p[4]; g5 = p[5];
        } else {  // padding row: infinitely far, zero dipole -> contributes exactly 0
            g0 = g1 = g2 = M::kFar; g3 = g4 = g5 = F(0);
        }
    };
    auto store_row = [&](int buf) {
        lds[buf][tid][0] = Vec4<F>{g0, g1, g2, g3};
        lds[buf][tid][1] = Vec4<F>{g4, g5, F(0), F(0)};
    };

    const int64_t n_src = s_end - s_begin;
    const int64_t n_tiles = (n_src + kBlock - 1) / kBlock;
    if (n_tiles > 0) {
        load_row(s_begin + tid);
        store_row(0);
    }
    __syncthreads();

    for (int64_t it = 0; it < n_tiles; ++it) {
        const int buf = (int)(it & 1);
        const int64_t tile_s = s_begin + it * kBlock;
        if (it + 1 < n_tiles) load_row(tile_s + kBlock + tid);   // in flight during the compute below
        int n_here = (int)((s_end - tile_s) < kBlock ? (s_end - tile_s) : kBlock);
        n_here = (n_here + 3) & ~3;                                // rows past s_end are padding rows

        for (int j0 = 0; j0 < n_here; j0 += kFlush) {
            const int j1 = (j0 + kFlush < n_here) ? j0 + kFlush : n_here;
            if (MODE == kField) {
                F A[KT][3], B[KT][3];
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int c = 0; c < 3; ++c) A[k][c] = B[k][c] = F(0);
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {
                    const Vec4<F> s0 = lds[buf][j][0];
                    const Vec4<F> s1 = lds[buf][j][1];
#pragma unroll
                    for (int k = 0; k < KT; ++k)
                        pair_field<F, NAN_COINC>(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx[k], ty[k], tz[k], a.eps,
                                                 A[k][0], A[k][1], A[k][2], B[k][0], B[k][1], B[k][2]);
                }
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int c = 0; c < 3; ++c) acc[k][c] += 3.0 * (double)A[k][c] - (double)B[k][c];
            } else {
                F P[KT];
#pragma unroll
                for (int k = 0; k < KT; ++k) P[k] = F(0);
#pragma unroll 4
                for (int j = j0; j < j1; ++j) {
                    const Vec4<F> s0 = lds[buf][j][0];
                    const Vec4<F> s1 = lds[buf][j][1];
#pragma unroll
                    for (int k = 0; k < KT; ++k)
                        pair_potential<F>(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx[k], ty[k], tz[k], P[k]);
                }
#pragma unroll
                for (int k = 0; k < KT; ++k) acc[k][0] += (double)P[k];
            }
        }
        if (it + 1 < n_tiles) store_row(buf ^ 1);
        __syncthreads();
    }

    // ---- epilogue: partial[chunk][t][:] -----------------------------------------------------
    const int64_t chunk_id = a.chunk_base + chunk;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int64_t t = tile_base + k * kBlock + tid;
        if (t < a.T) {
            bool excluded = false;
            if (a.tgt_group) excluded = (a.tgt_group[trow[k]] == chunk_id);
            F* o = a.partial + ((int64_t)chunk * a.T + t) * NC;
#pragma unroll
            for (int c = 0; c < NC; ++c) {
                const double v = (MODE == kField) ? -acc[k][c] : acc[k][c];
                o[c] = excluded ? F(0) : (F)v;
            }
        }
    }
}

}  // namespace dnp
