// dnp_xie.hip - the fork's "xie" pair functions:
//   xie_field / xie_intersaction  (field_utils.py:431-469, :509-519)  -> dnp_xie_pairs_f32/_f64
//   their knn_mask > 0 branch (field_utils.py:451-460, 467-468)        -> dnp_xie_knn_f32/_f64 + dnp_xie_pairs_knn_f32/_f64
//   xie_propagation_points_in_order (field_utils.py:569-605)           -> dnp_xie_order_f32/_f64, its diffuse pass (:597-603)
//                                                                          -> dnp_xie_rowdots_f32/_f64
//
// xie_pairs: the per-pair "reflected normal"  ref[t][s] = (n_s - C (n_s.r^) r^) / |r|^3,  r = x_s - x_t,
// not divided when |r| == 0 (so a coincident pair yields n_s itself), and its projection on the target
// normal M[t][s] = ref[t][s] . n_t with NaN/Inf -> 0.  The OUTPUT is the whole T x S matrix (4 or 12 bytes per pair):
// lanes run along the source index so that every store instruction writes 256 contiguous bytes of a row, targets are
// staged through LDS and broadcast.  The arithmetic is the reference's IEEE op order bit for bit for normal-range operands
// (sqrt, divisions, no fma contraction; see Recip for the one caveat) instead of the rsq/rcp chain of pair_kernel.h: the ordered propagation takes sign decisions on row sums of
// this matrix.  Measured (profiles/r03_xie_time.txt): the tensor form is bound by its HBM writes (4.5-5.6 TB/s), the
// matrix form by that arithmetic (2 TB/s written, 510 Gpairs/s) - see Recip and ieee_sqrt below for what it costs.
//
// xie_order: for each of R visiting orders, N sequential steps  inter[idx] = sum_j M[idx][j] * w[j];
// w[idx] = inter[idx] < 0 ? -1 : +1  (w starts at 0, so only visited points contribute).  One persistent
// workgroup per order.  The order is GIVEN, so nothing but w[idx] links a step to the next: the weights live in registers
// (thread t owns columns t, t + 1024, ...), the NEXT row is fetched while the current one is reduced, and every thread folds
// the 16 wave partials itself - one barrier per step (round 3; up to 16 384 points, beyond that the plain form with the
// weights in memory and two barriers).  Same products, same fp64 additions in the same order as the plain form.
#include "dnp_common.h"

#pragma clang fp contract(off)

namespace dnp {

constexpr int kXieBlock = 256;
constexpr int kXieTargets = 64;   // targets staged per workgroup

template <typename F>
struct XieArgs {
    const F* src; int64_t S; int64_t ld_src;
    const F* tgt; int64_t T; int64_t ld_tgt;
    F C;
    int vector_out;
    F* out;
    const double* kth_d2;     // KNN form: per source, squared distance and index of its k-th nearest target (xie_knn_kernel)
    const int64_t* kth_idx;
};

// squared distance of the kNN mask: fp64 on the exact coordinates, as the reference's KDTree sees them (field_utils.py:453-458:
// the tree is built on the numpy view of the cloud and computes in double).  ONE function for the selection and for the mask
// test of the pair kernel: both must see the same bits (the file compiles with fp contraction off).
template <typename F>
__device__ __forceinline__ double knn_d2(F sx, F sy, F sz, F tx, F ty, F tz) {
    const double dx = (double)sx - (double)tx, dy = (double)sy - (double)ty, dz = (double)sz - (double)tz;
    return (dx * dx + dy * dy) + dz * dz;
}

// correctly rounded square root in the operand's own precision (__builtin_sqrt on a float is the DOUBLE root: v_rsq_f64 and
// eight fp64 fmas per pair until round 3 - the same bits, since a double root rounded to float is the correctly rounded
// float root, at several times the cost)
__device__ inline float ieee_sqrt(float x) { return __builtin_sqrtf(x); }
__device__ inline double ieee_sqrt(double x) { return __builtin_sqrt(x); }

// IEEE division by a SHARED denominator.  hipcc expands a / b in fp32 to div_scale, rcp, one Newton step on the reciprocal,
// q0 = a r, two residual corrections (the last one in div_fmas) and div_fixup - 11 instructions and a transcendental per
// quotient, and the pair body divides three numbers by |R| and three by |R|^3.  Outside the exponent extremes (where
// div_scale rescales and div_fixup patches) that expansion IS  r = rcp(b); r += r (1 - b r);  q = a r;  q += r (a - b q)
// twice - so the refined reciprocal is computed once per denominator and every quotient costs five instructions, with the
// same bits as the compiler's division WHEN DENOMINATOR, NUMERATOR AND QUOTIENT ARE IN THE NORMAL RANGE (checked against a
// -DDNP_XIE_IEEE_DIV=1 build on 10^8 pairs, tools/gpu_xie_time.py).  The kernel guards the denominator (|R| outside [1e-10,
// 1e10] takes the compiler's division); a DENORMAL numerator or quotient - a coordinate difference or a normal component
// below 1e-38, which fp32 coordinates in the unit box cannot produce (their differences are 0 or >= 2^-25 apart) - skips
// div_scale / div_fixup here and may round differently from a / b: the bit-for-bit statement is for normal-range operands.
// fp64: the same idea on the compiler's fp64 expansion (two Newton steps, one residual correction), guard |R| in [1e-100, 1e100].
#ifndef DNP_XIE_IEEE_DIV
#define DNP_XIE_IEEE_DIV 0
#endif
template <typename F>
struct Recip {
    static constexpr bool kShared = !DNP_XIE_IEEE_DIV;
    F b, r;
    __device__ explicit Recip(F den) : b(den), r(F(0)) {
        if constexpr (sizeof(F) == 4 && !DNP_XIE_IEEE_DIV) {
            const float r0 = __builtin_amdgcn_rcpf(den);
            r = __builtin_fmaf(__builtin_fmaf(-den, r0, 1.0f), r0, r0);
        } else if constexpr (sizeof(F) == 8 && !DNP_XIE_IEEE_DIV) {
            // fp64 (round 5): hipcc's a / b is div_scale x 2, v_rcp_f64, TWO Newton steps on the reciprocal, q0 = a r, one residual
            // correction in div_fmas, div_fixup - 13 instructions and a transcendental per quotient; the refined reciprocal once
            // per denominator leaves three instructions per quotient, the same bits for normal-range operands (checked against a
            // -DDNP_XIE_IEEE_DIV=1 build on 10^8 pairs, tools/gpu_xie_time.py)
            const double r0 = __builtin_amdgcn_rcp(den);
            const double r1 = __builtin_fma(r0, __builtin_fma(-den, r0, 1.0), r0);
            r = __builtin_fma(r1, __builtin_fma(-den, r1, 1.0), r1);
        }
    }
    __device__ F divide(F a) const {
        if constexpr (sizeof(F) == 4 && !DNP_XIE_IEEE_DIV) {
            float q = a * r;
            q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
            q = __builtin_fmaf(__builtin_fmaf(-b, q, a), r, q);
            return q;
        } else if constexpr (sizeof(F) == 8 && !DNP_XIE_IEEE_DIV) {
            const double q = a * r;
            return __builtin_fma(__builtin_fma(-b, q, a), r, q);
        } else {
            return a / b;
        }
    }
};

template <typename F, bool VEC, bool KNN = false>
__global__ __launch_bounds__(kXieBlock) void xie_pairs_kernel(const XieArgs<F> a) {
    __shared__ __attribute__((aligned(16))) F tl[kXieTargets][8];       // rows of 8: one wide LDS read per half row
    const int tid = threadIdx.x;
    const int64_t s = (int64_t)blockIdx.x * kXieBlock + tid;
    const int64_t t0 = (int64_t)blockIdx.y * kXieTargets;
    const int nt = (int)((a.T - t0) < kXieTargets ? (a.T - t0) : kXieTargets);
    for (int i = tid; i < nt * 6; i += kXieBlock) {
        const int r = i / 6, c = i - r * 6;
        tl[r][c] = a.tgt[(t0 + r) * a.ld_tgt + c];
    }
    __syncthreads();
    if (s >= a.S) return;
    const F* ps = a.src + s * a.ld_src;
    const F sx = ps[0], sy = ps[1], sz = ps[2], nx = ps[3], ny = ps[4], nz = ps[5];
    double kd = 0.0;
    int64_t ki = 0;
    if constexpr (KNN) { kd = a.kth_d2[s]; ki = a.kth_idx[s]; }
    F* po = a.out + (t0 * a.S + s) * (VEC ? 3 : 1);                  // one 64-bit address, then a constant stride per target
    const int64_t step = a.S * (VEC ? 3 : 1);
    for (int r = 0; r < nt; ++r, po += step) {
        const F rx = sx - tl[r][0], ry = sy - tl[r][1], rz = sz - tl[r][2];      // R = source - target
        const F nrm = ieee_sqrt(rx * rx + ry * ry + rz * rz);
        // the quotients of the general case, computed for every lane (IEEE divisions: the reference's op order); a lane
        // with |R| == 0 - the diagonal of a self matrix - takes n_s instead: R_unit = 0 there and nothing is divided
        const F n3 = nrm * nrm * nrm;
        const bool coincident = nrm == F(0);
        F ux, uy, uz, d, fx, fy, fz;
        constexpr F kLoR = sizeof(F) == 4 ? F(1e-10) : F(1e-100), kHiR = sizeof(F) == 4 ? F(1e10) : F(1e100);
        if (Recip<F>::kShared && !(nrm >= kLoR && nrm <= kHiR) && !coincident) {
            // |R| at an exponent extreme (never on a cloud in the unit box): |R|^3 may be denormal or overflow, where the
            // compiler's division rescales its operands - take that division itself
            ux = rx / nrm; uy = ry / nrm; uz = rz / nrm;
            d = a.C * (nx * ux + ny * uy + nz * uz);
            fx = (nx - d * ux) / n3; fy = (ny - d * uy) / n3; fz = (nz - d * uz) / n3;
        } else {
            const Recip<F> by_nrm(nrm), by_n3(n3);
            ux = by_nrm.divide(rx); uy = by_nrm.divide(ry); uz = by_nrm.divide(rz);
            d = a.C * (nx * ux + ny * uy + nz * uz);
            fx = coincident ? nx : by_n3.divide(nx - d * ux);
            fy = coincident ? ny : by_n3.divide(ny - d * uy);
            fz = coincident ? nz : by_n3.divide(nz - d * uz);
        }
        if constexpr (KNN) {
            // `ref_normal_s *= tree_mask[:, :, None]` (field_utils.py:467-468): target t0 + r is among the k nearest targets of
            // this source iff (d2, index) <= the k-th pair; the others are multiplied by 0 (an Inf / NaN entry turns NaN, as there)
            const double d2 = knn_d2<F>(sx, sy, sz, tl[r][0], tl[r][1], tl[r][2]);
            const bool in = d2 < kd || (d2 == kd && t0 + r <= ki);
            if (!in) { fx = fx * F(0); fy = fy * F(0); fz = fz * F(0); }
        }
        if constexpr (VEC) {
            po[0] = fx; po[1] = fy; po[2] = fz;
        } else {
            F v = fx * tl[r][3] + fy * tl[r][4] + fz * tl[r][5];
            if (!__builtin_isfinite(v)) v = F(0);           // intersaction[isnan / isinf] = 0
            po[0] = v;
        }
    }
}

// ---- kNN selection for the masked forms (field_utils.py:451-460: KDTree(targets).query(sources, k)) ----------------------
// One wavefront per source; the wave keeps the 64 smallest (d2, index) pairs seen so far SORTED ACROSS ITS LANES (lane i = the
// (i+1)-th smallest), and the value in lane m-1 (m = min(k, 64)) is the bar a candidate has to beat.  Targets are staged
// through LDS in tiles shared by the workgroup's four sources; every lane computes one distance per iteration, a ballot names
// the lanes under the bar (after the first few hundred targets: rarely any - about k ln(T / k) insertions per source in all)
// and the wave inserts those one at a time: the lanes behind the insertion point take their left neighbour's pair (DPP
// wave_shr:1, register to register).  Everything about an insertion is wave-uniform, so no lane idles through another lane's
// work.  (The first form of this kernel kept a sorted list per LANE: nearly every iteration had some lane inserting, and the
// whole wave walked its compare-and-shift chain - 807 us at N = 10 000, k = 20 against this form's figure in
// profiles/r05_xie_time.txt.)  k > 64: further passes, each restricted to pairs behind the last one taken.  The result is the
// k-th pair per source; ties at equal distance go to the lower target index (the tree's choice among equidistant targets is an
// implementation detail; fp64 distances of distinct points rarely tie).
constexpr int kKnnBlock = 256, kKnnTile = 1024;

// lane i takes lane i-1's value, lane 0 takes `first` (DPP wave_shr:1 - gfx9 only, like the row_bcast folds of wave_sum_f64)
__device__ __forceinline__ int wave_shr1_i32(int v, int first) {
    return __builtin_amdgcn_update_dpp(first, v, 0x138, 0xf, 0xf, false);
}
__device__ __forceinline__ double wave_shr1_f64(double v, double first) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v), f = __builtin_bit_cast(unsigned long long, first);
    const unsigned lo = (unsigned)wave_shr1_i32((int)(unsigned)(b & 0xffffffffull), (int)(unsigned)(f & 0xffffffffull));
    const unsigned hi = (unsigned)wave_shr1_i32((int)(unsigned)(b >> 32), (int)(unsigned)(f >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double readlane_f64(double v, int l) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffull), l);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), l);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <typename F>
__global__ __launch_bounds__(kKnnBlock) void xie_knn_kernel(const F* __restrict__ src, int64_t S, int64_t ld_src,
                                                            const F* __restrict__ tgt, int64_t T, int64_t ld_tgt, int k,
                                                            double* __restrict__ kth_d2, int64_t* __restrict__ kth_idx) {
    __shared__ F tl[3][kKnnTile];                                     // coordinate-major: lane l reads tl[c][j + l], no conflicts
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int64_t s = (int64_t)blockIdx.x * (kKnnBlock / 64) + wave;
    const bool live = s < S;                                           // dead waves keep the barriers company
    F sx = F(0), sy = F(0), sz = F(0);
    if (live) { sx = src[s * ld_src]; sy = src[s * ld_src + 1]; sz = src[s * ld_src + 2]; }
    double prev_d = -1.0;                                              // every pair is behind (-1, -1): d2 >= 0
    int prev_i = -1;
    for (int remaining = k; remaining > 0;) {                          // the same count in every wave of the workgroup
        const int m = remaining < 64 ? remaining : 64;
        double ld = __builtin_huge_val();                              // the wave's sorted list, one pair per lane
        int li = 0x7fffffff;
        double bar = __builtin_huge_val();                             // lane m-1's distance (wave-uniform)
        for (int64_t t0 = 0; t0 < T; t0 += kKnnTile) {
            const int nt = (int)((T - t0) < kKnnTile ? (T - t0) : kKnnTile);
            __syncthreads();                                           // the previous tile has been consumed
            for (int i = threadIdx.x; i < nt * 3; i += kKnnBlock) {
                const int r = i / 3, c = i - r * 3;
                tl[c][r] = tgt[(t0 + r) * ld_tgt + c];
            }
            __syncthreads();
            if (!live) continue;
            for (int j0 = 0; j0 < nt; j0 += 64) {
                const int j = j0 + lane;
                double d = __builtin_huge_val();
                if (j < nt) d = knn_d2<F>(sx, sy, sz, tl[0][j], tl[1][j], tl[2][j]);
                const int t = (int)(t0 + j);
                const bool behind = d > prev_d || (d == prev_d && t > prev_i);
                // strict: targets arrive in index order, so an equal distance with a (necessarily higher) index stays out
                unsigned long long todo = __ballot(behind && d < bar);
                while (todo) {
                    const int l = (int)__builtin_ctzll(todo);
                    todo &= todo - 1;
                    const double cd = readlane_f64(d, l);
                    if (!(cd < bar)) continue;                         // the bar has come down since the ballot
                    const int ci = (int)(t0 + j0) + l;
                    const bool gt = ld > cd;                           // a suffix of the lanes (the list is sorted)
                    const double sd = wave_shr1_f64(ld, -1.0);         // lane 0 has no left neighbour: -1 is never > cd
                    const int si = wave_shr1_i32(li, -1);
                    const bool first = !(sd > cd);                     // ... the first lane of the suffix takes the candidate
                    li = gt ? (first ? ci : si) : li;
                    ld = gt ? (first ? cd : sd) : ld;
                    bar = readlane_f64(ld, m - 1);
                }
            }
        }
        if (live) {
            prev_d = bar;
            prev_i = __builtin_amdgcn_readlane(li, m - 1);
        }
        remaining -= m;
    }
    if (live && lane == 0) { kth_d2[s] = prev_d; kth_idx[s] = prev_i; }
}

constexpr int kOrderThreads = 1024;

// Sum of a double over the 64 lanes of a wavefront as a wave-uniform value, register to register: four butterfly rounds on DPP
// (quad_perm lane ^ 1, lane ^ 2, row_half_mirror, row_mirror) leave every 16-lane row's sum in all of its lanes, row_bcast:15 /
// row_bcast:31 fold the four rows into lane 63, v_readlane makes it uniform.  The association is the balanced binary tree over the
// lanes in index order (IEEE addition commutes, so the mirrored operand orders do not matter): v = v[0::2] + v[1::2], six times.
// Rounds 3-4 reduced with six __shfl_down steps - six dependent ds_bpermute round trips, ~0.3 us of a 1.5-us step.
// (The moves carry no `old` operand - a register copy and a wait state per moved word otherwise: in the row_bcast rounds the rows
// the mask leaves out add whatever the destination register held; only lane 63 is read, and what reaches it - row 3 += row 2's lane
// 47, then row 3 += row 1's lane 31 - was written or is untouched input.)
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ double xie_dpp_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b & 0xffffffffull), CTRL, ROWMASK, 0xf, ROWMASK == 0xf);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xf, ROWMASK == 0xf);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_sum_f64(double v) {
    v += xie_dpp_f64<0xB1>(v);                          // quad_perm [1,0,3,2]
    v += xie_dpp_f64<0x4E>(v);                          // quad_perm [2,3,0,1]
    v += xie_dpp_f64<0x141>(v);                         // row_half_mirror
    v += xie_dpp_f64<0x140>(v);                         // row_mirror: every lane of a row holds the row's sum
    v += xie_dpp_f64<0x142, 0xa>(v);                    // row_bcast:15 into rows 1 and 3
    v += xie_dpp_f64<0x143, 0xc>(v);                    // row_bcast:31 into rows 2 and 3: lane 63 = (r3 + r2) + (r1 + r0)
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffull), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

// F = the matrix's precision (float; double for float64 clouds, round 5): products rounded in F as the reference's
// interaction_mat[idx] * weights, row sums in fp64, the sum rounded to F before its sign is taken.
template <typename F, int VEC> struct DotVec;
template <> struct DotVec<float, 4> { using T = float4; };
template <> struct DotVec<double, 2> { using T = double2; };
template <> struct DotVec<float, 1> { using T = float; };
template <> struct DotVec<double, 1> { using T = double; };
template <typename V, int VEC> __device__ __forceinline__ auto dot_elem(const V& v, int e) {
    if constexpr (VEC == 1) return v;
    else if constexpr (VEC == 2) return e == 0 ? v.x : v.y;
    else return e == 0 ? v.x : e == 1 ? v.y : e == 2 ? v.z : v.w;
}

// VEC (both kernels): a thread owns its columns in groups of VEC consecutive ones - column j belongs to thread (j / VEC) % 1024 -, so
// that the register form can fetch a row with 16-byte loads (VEC = 4 floats / 2 doubles when N is a multiple of VEC, else 1);
// a thread adds its columns in ascending order.  The plain form uses the same ownership, so both forms sum in the same order.
template <typename F, int VEC>
__global__ __launch_bounds__(kOrderThreads) void xie_order_kernel(const F* __restrict__ M, int64_t N,
                                                                 const int64_t* __restrict__ order,
                                                                 F* __restrict__ weights,
                                                                 F* __restrict__ inter, const int* __restrict__ only_if) {
    if (only_if && !only_if[blockIdx.x]) return;           // blocked form (below): only the rows it left to this kernel
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t* ord = order + (int64_t)blockIdx.x * N;
    F* w = weights + (int64_t)blockIdx.x * N;
    F* out = inter + (int64_t)blockIdx.x * N;
    __shared__ double part[kOrderThreads / 64];
    for (int64_t j = tid; j < N; j += kOrderThreads) { w[j] = F(0); out[j] = F(0); }
    __syncthreads();
    for (int64_t i = 0; i < N; ++i) {
        const int64_t idx = ord[i];
        const F* row = M + idx * N;
        double s = 0.0;
        using V = typename DotVec<F, VEC>::T;            // (N % VEC == 0: a group of VEC columns lies inside the row as a whole)
        for (int64_t j0 = (int64_t)tid * VEC; j0 < N; j0 += (int64_t)kOrderThreads * VEC) {
            const V rv = *reinterpret_cast<const V*>(row + j0), wv = *reinterpret_cast<const V*>(w + j0);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += (double)(dot_elem<V, VEC>(rv, e) * dot_elem<V, VEC>(wv, e));
        }
        s = wave_sum_f64(s);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        if (tid == 0) {
            double tot = 0.0;
            for (int k = 0; k < kOrderThreads / 64; ++k) tot += part[k];
            const F v = (F)tot;
            out[idx] = v;
            w[idx] = (v < F(0)) ? F(-1) : F(1);
        }
        __syncthreads();   // w[idx] is visible to the whole workgroup before the next row is weighted
    }
}

// VPT columns per thread (N <= 1024 * VPT).  DEPTH rows are in flight: the order is GIVEN, so every future row's address is known,
// and a row of a 400 MB matrix visited in a scattered order comes from HBM (~2 us) - with ONE row ahead (rounds 3-4: cur / nxt)
// a step could hide only its own ~0.5 us of reduction behind that; round 5 keeps a ring of DEPTH rows (3 in fp32, 2 in fp64:
// the 128-VGPR budget of a 1024-thread workgroup), refilled slot by slot behind the step that consumed it.  Same products,
// same fp64 additions in the same order as before (pinned bit for bit against the numpy statement of that order).
#ifndef DNP_XIE_DEPTH
#define DNP_XIE_DEPTH 3
#endif
template <typename F, int VPT, int DEPTH, int VEC>
__global__ __launch_bounds__(kOrderThreads) void xie_order_reg_kernel(const F* __restrict__ M, int64_t N,
                                                                     const int64_t* __restrict__ order,
                                                                     F* __restrict__ weights,
                                                                     F* __restrict__ inter, const int* __restrict__ only_if) {
    if (only_if && !only_if[blockIdx.x]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t* ord = order + (int64_t)blockIdx.x * N;
    F* out = inter + (int64_t)blockIdx.x * N;
    __shared__ double part[2][kOrderThreads / 64];
    F w[VPT], buf[DEPTH][VPT];
#pragma unroll
    for (int k = 0; k < VPT; ++k) w[k] = F(0);
    // interactions = torch.zeros(T, N) in the reference (field_utils.py:581): an order row that is not a full permutation
    // (a repeated index) leaves entries unvisited, and they must read 0, not whatever the caller's buffer held (round-3
    // advisor: field_utils passes torch.empty).  The stores of the loop below come after the loop's first barrier.
    for (int64_t j = tid; j < N; j += kOrderThreads) out[j] = F(0);
    static_assert(VPT % VEC == 0, "whole groups of VEC columns per thread");
    using V = typename DotVec<F, VEC>::T;
    // slot k of a thread = column (k / VEC) * (VEC * 1024) + VEC * tid + k % VEC
    auto column = [&](int k) -> int64_t { return (int64_t)(k / VEC) * (VEC * kOrderThreads) + (int64_t)VEC * tid + (k % VEC); };
    auto fetch = [&](int64_t idx, F (&dst)[VPT]) {
        const F* row = M + idx * N;
#pragma unroll
        for (int g = 0; g < VPT / VEC; ++g) {              // clamped, unconditional: all loads in flight at once (N % VEC == 0:
            const int64_t j = column(g * VEC);             // a group lies inside the row or behind it as a whole)
            const V v = *reinterpret_cast<const V*>(row + (j < N ? j : N - VEC));
#pragma unroll
            for (int e = 0; e < VEC; ++e) dst[g * VEC + e] = dot_elem<V, VEC>(v, e);
        }
    };
#pragma unroll
    for (int d = 0; d < DEPTH; ++d)
        if (d < N) fetch(ord[d], buf[d]);
    for (int64_t i0 = 0; i0 < N; i0 += DEPTH) {
#pragma unroll
        for (int d = 0; d < DEPTH; ++d) {                  // slot d holds row i0 + d (static register indexing)
            const int64_t i = i0 + d;
            if (i < N) {                                    // uniform over the workgroup
                const int64_t idx = ord[i];
                double s = 0.0;
#pragma unroll
                for (int k = 0; k < VPT; ++k)
                    if (column(k) < N) s += (double)(buf[d][k] * w[k]);
                s = wave_sum_f64(s);
                double* p = part[i & 1];
                if (lane == 0) p[wave] = s;
                __syncthreads();                           // the only barrier of the step (partials double-buffered by parity)
                double tot = 0.0;
#pragma unroll
                for (int k = 0; k < kOrderThreads / 64; ++k) tot += p[k];
                const F v = (F)tot;
                const F sign = (v < F(0)) ? F(-1) : F(1);
                if (tid == (int)((idx / VEC) % kOrderThreads)) {
                    const int kk = (int)(idx / (VEC * kOrderThreads)) * VEC + (int)(idx % VEC);
#pragma unroll
                    for (int k = 0; k < VPT; ++k) w[k] = (k == kk) ? sign : w[k];
                }
                if (tid == 0) out[idx] = v;
                if (i + DEPTH < N) fetch(ord[i + DEPTH], buf[d]);   // refill this slot: in flight under the next DEPTH - 1 steps
            }
        }
    }
    F* wout = weights + (int64_t)blockIdx.x * N;
#pragma unroll
    for (int k = 0; k < VPT; ++k) {
        const int64_t j = column(k);
        if (j < N) wout[j] = w[k];
    }
}

// ---- the diffuse pass of the ordered propagation (field_utils.py:597-603): out[r][i] = sum_j M[i][j] * w[r][j] for the R final
// weight vectors at once - one pass over the N x N matrix (HBM-bound: 4 N^2 bytes), up to kDotOrders weight vectors per pass.
// A wavefront takes kDotRows consecutive matrix rows: the weights of a column are loaded once and used for all of them (the
// first form - one row per wavefront - issued one load of the matrix and R loads of weights per element and reached 0.8 TB/s:
// bound by load issue, not by HBM; profiles/SUMMARY_r05.md).  Products rounded in F, sums in fp64 in a fixed order (per lane its
// columns lane, lane + 64, ... ascending, then the butterfly over the lanes), the result rounded to F.  (Rounds 1-4 ran this
// as a torch matmul - the one rocBLAS call on the path.)
constexpr int kDotOrders = 5;           // the callers' `times` is odd: 1 or 5 visiting orders - one pass over the matrix for up to 5
// VEC = elements per lane and load (16-byte loads when N is a multiple of VEC and the buffers are 16-byte aligned: 146 -> ~100 us
// at N = 10^4 in fp32, profiles/r05_xie_time.txt; VEC = 1 is the general form).  A lane's columns: VEC consecutive ones per block
// of 64 VEC, blocks ascending - a fixed order for given N and VEC.
template <typename F, int VEC>
__global__ __launch_bounds__(256) void xie_rowdots_kernel(const F* __restrict__ M, int64_t N, const F* __restrict__ weights,
                                                          int64_t R, int64_t r0, F* __restrict__ out) {
    using V = typename DotVec<F, VEC>::T;
    constexpr int kDotRows = sizeof(F) == 4 ? 8 : 4;      // 40 / 20 fp64 sums per lane
    const int lane = threadIdx.x & 63;
    const int64_t i0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * kDotRows;
    if (i0 >= N) return;
    const int nr = (int)((R - r0) < kDotOrders ? (R - r0) : kDotOrders);
    double s[kDotRows][kDotOrders];
#pragma unroll
    for (int q = 0; q < kDotRows; ++q)
#pragma unroll
        for (int r = 0; r < kDotOrders; ++r) s[q][r] = 0.0;
    for (int64_t j = (int64_t)lane * VEC; j < N; j += 64 * VEC) {
        V w[kDotOrders], m[kDotRows];
#pragma unroll
        for (int r = 0; r < kDotOrders; ++r)              // clamped: all loads unconditional
            w[r] = *reinterpret_cast<const V*>(weights + (r0 + (r < nr ? r : 0)) * N + j);
#pragma unroll
        for (int q = 0; q < kDotRows; ++q) m[q] = *reinterpret_cast<const V*>(M + (i0 + q < N ? i0 + q : N - 1) * N + j);
#pragma unroll
        for (int e = 0; e < VEC; ++e)
#pragma unroll
            for (int q = 0; q < kDotRows; ++q)
#pragma unroll
                for (int r = 0; r < kDotOrders; ++r) s[q][r] += (double)(dot_elem<V, VEC>(m[q], e) * dot_elem<V, VEC>(w[r], e));
    }
#pragma unroll
    for (int q = 0; q < kDotRows; ++q)
#pragma unroll
        for (int r = 0; r < kDotOrders; ++r) {
            for (int off = 32; off > 0; off >>= 1) s[q][r] += __shfl_xor(s[q][r], off, 64);
            if (lane == 0 && r < nr && i0 + q < N) out[(r0 + r) * N + i0 + q] = (F)s[q][r];
        }
}

// ---- the ordered propagation in BLOCKS (round 5) --------------------------------------------------------------------
// The visiting order is GIVEN, so the loop  inter[idx_t] = sum_j M[idx_t][j] w[j];  w[idx_t] = sign  is a forward substitution
// with a sign in it, and it blocks like one.  For a block of kBlkSteps consecutive steps:
//   rows kernel  (the whole chip, HBM-bound): base[t] = sum_j M[idx_t][j] w[j] over the weights decided BEFORE the block - one
//                wavefront per step streams its matrix row -, and the block's own kBlkSteps x kBlkSteps corner
//                sub[s][t] = M[idx_t][idx_s] is gathered from the rows while they are hot;
//   solve kernel (one wavefront per visiting order): the 256 dependent steps on registers - lane l owns steps 4l..4l+3, a
//                step reads its sum with v_readlane, takes the sign and adds its column of the corner to the 256 pending sums
//                (one fused multiply-add per owned step, the column prefetched 8 steps ahead): ~20 ns per step where the
//                row-per-step kernels above pay one matrix row fetch by one CU and a workgroup barrier (1.24 us at N = 10^4).
// Same products (M's precision, weights +-1) and fp64 sums; the ORDER of the fp64 additions differs from the row-per-step
// kernels (columns before the block in lane order, then the block's own steps in visiting order), so `inter` agrees with them
// to fp64 rounding, not bit for bit.  An order row that is not a permutation of 0..N-1 (a repeated index re-decides a point:
// the recurrence then needs weight DIFFERENCES) is found by a check kernel and left to the row-per-step kernel: same results
// as ever for such rows, no host round trip.
constexpr int kBlkSteps = 256;

__global__ __launch_bounds__(256) void xie_perm_check_kernel(const int64_t* __restrict__ order, int64_t N, int64_t R,
                                                             unsigned* __restrict__ marks, int* __restrict__ flags) {
    const int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (i >= R * N) return;
    const int64_t r = i / N, idx = order[i];
    if (idx < 0 || idx >= N) { flags[r] = 1; return; }
    if (atomicAdd(&marks[r * N + idx], 1u) != 0u) flags[r] = 1;
}

constexpr int kBlkAhead = 8;                               // corner columns the solve kernel keeps in flight
constexpr int kBlkCols = kBlkSteps + kBlkAhead;            // ... and reads past the last step without a clamp: the corner has that many
template <typename F, int VEC>
__global__ __launch_bounds__(256) void xie_block_rows_kernel(const F* __restrict__ M, int64_t N, const int64_t* __restrict__ order,
                                                             const F* __restrict__ weights, int64_t t0, int nb,
                                                             const int* __restrict__ flags, double* __restrict__ base,
                                                             F* __restrict__ sub) {
    using V = typename DotVec<F, VEC>::T;
    const int r = blockIdx.y, t = blockIdx.x;              // one workgroup per step of the block: four wavefronts share its matrix row
    if (flags[r]) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int64_t* ord = order + (int64_t)r * N + t0;
    const F* row = M + ord[t] * N;
    const F* w = weights + (int64_t)r * N;
    __shared__ double part[4];
    if (t0 > 0) {                                           // the first block starts from all-zero weights
        double s = 0.0;
#pragma unroll 4
        for (int64_t j = (int64_t)tid * VEC; j < N; j += 256 * VEC) {      // a thread adds its columns in ascending order
            const V m = *reinterpret_cast<const V*>(row + j), wv = *reinterpret_cast<const V*>(w + j);
#pragma unroll
            for (int e = 0; e < VEC; ++e) s += (double)(dot_elem<V, VEC>(m, e) * dot_elem<V, VEC>(wv, e));
        }
        s = wave_sum_f64(s);
        if (lane == 0) part[wave] = s;
        __syncthreads();
        if (tid == 0) base[(int64_t)r * kBlkSteps + t] = ((part[0] + part[1]) + part[2]) + part[3];
    } else if (tid == 0) {
        base[(int64_t)r * kBlkSteps + t] = 0.0;
    }
    // the block's own corner: sub[s][t] = M[idx_t][idx_s] = what step s adds to the sum of a LATER step t; 0 for t <= s, so the
    // solve kernel can add whole columns and still end with every step's sum as it stood when the step was taken
    F* col = sub + (int64_t)r * kBlkCols * kBlkSteps + t;
    for (int q = tid; q < nb; q += 256) col[(int64_t)q * kBlkSteps] = t > q ? row[ord[q]] : F(0);
}

// Four consecutive entries of a corner column, fetched by hand: the solve loop keeps kAhead columns in flight ACROSS its back edge,
// and for loads the compiler can see its s_waitcnt at the loop header was vmcnt(0) - a full drain every kAhead steps, 250 cycles
// per step.  Loads issued from inline asm are invisible to that pass; the loop states its own waits (wait_oldest), with the loaded
// registers passed THROUGH the wait so that nothing that reads them can be scheduled in front of it.
typedef float xie_v4f __attribute__((ext_vector_type(4)));
typedef double xie_v2d __attribute__((ext_vector_type(2)));
template <typename F> struct Col4;
template <> struct Col4<float> {
    static constexpr int kLoads = 1;
    xie_v4f v;
    __device__ __forceinline__ void issue(const float* p) { asm volatile("global_load_dwordx4 %0, %1, off" : "=&v"(v) : "v"(p) : "memory"); }
    template <int YOUNGER> __device__ __forceinline__ void wait_oldest() { asm volatile("s_waitcnt vmcnt(%1)" : "+v"(v) : "n"(YOUNGER)); }
    __device__ __forceinline__ double get(int q) const { return (double)v[q]; }
};
template <> struct Col4<double> {
    static constexpr int kLoads = 2;
    xie_v2d lo, hi;
    __device__ __forceinline__ void issue(const double* p) {
        asm volatile("global_load_dwordx4 %0, %2, off\n\tglobal_load_dwordx4 %1, %2, off offset:16" : "=&v"(lo), "=&v"(hi) : "v"(p) : "memory");
    }
    template <int YOUNGER> __device__ __forceinline__ void wait_oldest() { asm volatile("s_waitcnt vmcnt(%2)" : "+v"(lo), "+v"(hi) : "n"(YOUNGER)); }
    __device__ __forceinline__ double get(int q) const { return q < 2 ? lo[q] : hi[q - 2]; }
};

template <typename F>
__global__ __launch_bounds__(64) void xie_block_solve_kernel(const int64_t* __restrict__ order, int64_t N, int64_t t0, int nb,
                                                             const int* __restrict__ flags, const double* __restrict__ base,
                                                             const F* __restrict__ sub, F* __restrict__ weights,
                                                             F* __restrict__ inter) {
    const int r = blockIdx.x;
    if (flags[r]) return;
    const int lane = threadIdx.x;
    constexpr int kOwn = kBlkSteps / 64, kAhead = kBlkAhead;   // steps per lane (4 lane + q), columns in flight
    static_assert(kOwn == 4 && kAhead % kOwn == 0, "the unrolled step loop assumes 4 steps per lane");
    double pend[kOwn];
#pragma unroll
    for (int q = 0; q < kOwn; ++q) {
        const int u = kOwn * lane + q;
        pend[q] = u < nb ? base[(int64_t)r * kBlkSteps + u] : 0.0;
    }
    // the compiler's own loads (base) have landed before the first hand-issued one: its counts and the loop's never mix
    asm volatile("s_waitcnt vmcnt(0)" : "+v"(pend[0]), "+v"(pend[1]), "+v"(pend[2]), "+v"(pend[3]));
    // column s of the corner, this lane's four entries.  A load is issued for EVERY step of the main loop (the corner has kAhead
    // columns to spare), so the in-flight count the waits rely on is exact.
    const F* next = sub + (int64_t)r * kBlkCols * kBlkSteps + kOwn * lane;
    Col4<F> ring[kAhead];
#pragma unroll
    for (int d = 0; d < kAhead; ++d) { ring[d].issue(next); next += kBlkSteps; }
    auto step = [&](int t, int d) {
        // the step's sum as it stands: nothing is added to it any more (columns are 0 up to and including their own step)
        const double v = readlane_f64(pend[d % kOwn], t / kOwn);
        const double wd = (F)v < F(0) ? -1.0 : 1.0;        // the sum rounded to M's precision before its sign is taken
        // M[idx_u][idx_t] * w is exact in M's precision (w = +-1): one fused multiply-add in fp64 = that product added in fp64
#pragma unroll
        for (int q = 0; q < kOwn; ++q) pend[q] = __builtin_fma(ring[d].get(q), wd, pend[q]);
    };
    int tb = 0;
    for (; tb + kAhead <= nb; tb += kAhead) {
#pragma unroll
        for (int d = 0; d < kAhead; ++d) {
            ring[d].template wait_oldest<(kAhead - 1) * Col4<F>::kLoads>();
            step(tb + d, d);
            ring[d].issue(next);
            next += kBlkSteps;
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");        // the tail (nb % kAhead steps) and the stores below: everything has landed
#pragma unroll
    for (int d = 0; d < kAhead; ++d)
        if (tb + d < nb) {                                  // wave-uniform
            ring[d].template wait_oldest<0>();
            step(tb + d, d);
        }
    const int64_t* ord = order + (int64_t)r * N + t0;
#pragma unroll
    for (int q = 0; q < kOwn; ++q) {
        const int u = kOwn * lane + q;
        if (u < nb) {
            const int64_t idx = ord[u];
            const F vf = (F)pend[q];
            weights[(int64_t)r * N + idx] = vf < F(0) ? F(-1) : F(1);
            inter[(int64_t)r * N + idx] = vf;
        }
    }
}

template <typename F>
static int run_xie_pairs(const F* src, int64_t S, int64_t ld_src, const F* tgt, int64_t T, int64_t ld_tgt, F C,
                         int vector_out, F* out, hipStream_t stream, const double* kth_d2 = nullptr,
                         const int64_t* kth_idx = nullptr) {
    clear_error();
    DNP_REQUIRE(S >= 0 && T >= 0, "negative size");
    if (S == 0 || T == 0) return DNP_OK;
    DNP_REQUIRE(src && tgt && out, "NULL pointer");
    DNP_REQUIRE(ld_src >= 6 && ld_tgt >= 6, "xie pairs need 6-column sources and targets");
    const int64_t gy = ceil_div(T, (int64_t)kXieTargets);
    DNP_REQUIRE(gy <= 65535, "T=%lld exceeds %d targets per launch", (long long)T, 65535 * kXieTargets);
    DNP_REQUIRE((kth_d2 == nullptr) == (kth_idx == nullptr), "kth_d2 and kth_idx come together");
    XieArgs<F> a{src, S, ld_src, tgt, T, ld_tgt, C, vector_out, out, kth_d2, kth_idx};
    const dim3 grid((unsigned)ceil_div(S, (int64_t)kXieBlock), (unsigned)gy);
    if (kth_d2) {
        if (vector_out) hipLaunchKernelGGL((xie_pairs_kernel<F, true, true>), grid, dim3(kXieBlock), 0, stream, a);
        else hipLaunchKernelGGL((xie_pairs_kernel<F, false, true>), grid, dim3(kXieBlock), 0, stream, a);
    } else {
        if (vector_out) hipLaunchKernelGGL((xie_pairs_kernel<F, true>), grid, dim3(kXieBlock), 0, stream, a);
        else hipLaunchKernelGGL((xie_pairs_kernel<F, false>), grid, dim3(kXieBlock), 0, stream, a);
    }
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

template <typename F>
static int run_xie_knn(const F* src, int64_t S, int64_t ld_src, const F* tgt, int64_t T, int64_t ld_tgt, int64_t k,
                       double* kth_d2, int64_t* kth_idx, hipStream_t stream) {
    clear_error();
    DNP_REQUIRE(S >= 0 && T >= 0, "negative size");
    DNP_REQUIRE(k >= 1 && k <= T, "k=%lld outside 1..T=%lld (the caller clamps: min(len(targets), knn_mask))", (long long)k, (long long)T);
    if (S == 0) return DNP_OK;
    DNP_REQUIRE(src && tgt && kth_d2 && kth_idx, "NULL pointer");
    DNP_REQUIRE(ld_src >= 3 && ld_tgt >= 3, "kNN needs 3 coordinate columns");
    DNP_REQUIRE(T <= INT32_MAX, "T=%lld exceeds 2^31-1 targets", (long long)T);
    const dim3 grid((unsigned)ceil_div(S, (int64_t)(kKnnBlock / 64)));
    hipLaunchKernelGGL((xie_knn_kernel<F>), grid, dim3(kKnnBlock), 0, stream, src, S, ld_src, tgt, T, ld_tgt, (int)k, kth_d2, kth_idx);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

}  // namespace dnp

using namespace dnp;

template <typename F>
static int run_xie_order(const F* M, int64_t N, const int64_t* order, int64_t R, F* weights, F* inter, hipStream_t stream,
                         const int* only_if = nullptr) {
    clear_error();
    DNP_REQUIRE(N >= 0 && R >= 0, "negative size");
    if (N == 0 || R == 0) return DNP_OK;
    DNP_REQUIRE(M && order && weights && inter, "NULL pointer");
#ifndef DNP_XIE_ORDER_PLAIN      // A/B builds: the plain form for every size
#define DNP_XIE_ORDER_PLAIN 0
#endif
    // register form: the weights of a thread's columns live in VGPRs (1024 threads per workgroup: <= 128 VGPRs each, so
    // fp64 stops at 12 columns per thread)
    constexpr bool f64 = sizeof(F) == 8;
    constexpr int kDepth = f64 ? 2 : DNP_XIE_DEPTH;        // rows in flight (VGPR budget: 128 per thread)
    constexpr int kBig = f64 ? 12 : 16;                    // columns per thread of the large register form
    constexpr int kVec = f64 ? 2 : 4;                      // 16-byte row loads when the rows allow (N % kVec == 0, M 16-byte aligned)
    const bool wide = N % kVec == 0 && (reinterpret_cast<uintptr_t>(M) & 15) == 0 && (reinterpret_cast<uintptr_t>(weights) & 15) == 0;
    const dim3 grid((unsigned)R), block(kOrderThreads);
#define DNP_XIE_LAUNCH(KERNEL) hipLaunchKernelGGL((KERNEL), grid, block, 0, stream, M, N, order, weights, inter, only_if)
    if (!DNP_XIE_ORDER_PLAIN && N <= 4 * kOrderThreads) {
        if (wide) DNP_XIE_LAUNCH((xie_order_reg_kernel<F, 4, f64 ? 3 : 4, kVec>));
        else DNP_XIE_LAUNCH((xie_order_reg_kernel<F, 4, f64 ? 3 : 4, 1>));
    } else if (!DNP_XIE_ORDER_PLAIN && N <= kBig * kOrderThreads) {
        if (wide) DNP_XIE_LAUNCH((xie_order_reg_kernel<F, kBig, kDepth, kVec>));
        else DNP_XIE_LAUNCH((xie_order_reg_kernel<F, kBig, kDepth, 1>));
    } else {
        if (wide) DNP_XIE_LAUNCH((xie_order_kernel<F, kVec>));
        else DNP_XIE_LAUNCH((xie_order_kernel<F, 1>));
    }
#undef DNP_XIE_LAUNCH
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

static size_t xie_order_workspace(int64_t N, int64_t R, size_t elem) {
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    return pad((size_t)R * sizeof(int)) + pad((size_t)R * (size_t)N * sizeof(unsigned)) + pad((size_t)R * kBlkSteps * sizeof(double)) +
           pad((size_t)R * kBlkCols * kBlkSteps * elem);
}

#ifndef DNP_XIE_BLOCKED_FROM     // points from which the blocked form is used (below: the row-per-step kernels)
#define DNP_XIE_BLOCKED_FROM 512
#endif
template <typename F>
static int run_xie_order_blocked(const F* M, int64_t N, const int64_t* order, int64_t R, F* weights, F* inter, void* workspace,
                                 size_t workspace_bytes, hipStream_t stream) {
    clear_error();
    DNP_REQUIRE(N >= 0 && R >= 0, "negative size");
    if (N == 0 || R == 0) return DNP_OK;
    DNP_REQUIRE(M && order && weights && inter, "NULL pointer");
    if (N < DNP_XIE_BLOCKED_FROM) return run_xie_order<F>(M, N, order, R, weights, inter, stream);
    DNP_REQUIRE(R <= 65535, "R=%lld visiting orders exceed one launch", (long long)R);
    const size_t need = xie_order_workspace(N, R, sizeof(F));
    if (!workspace || workspace_bytes < need) {
        set_error("workspace of %zu bytes required, %zu given", need, workspace ? workspace_bytes : (size_t)0);
        return DNP_EWORKSPACE;
    }
    auto pad = [](size_t b) { return (b + 255) & ~(size_t)255; };
    char* p = (char*)workspace;
    int* flags = (int*)p; p += pad((size_t)R * sizeof(int));
    unsigned* marks = (unsigned*)p; p += pad((size_t)R * (size_t)N * sizeof(unsigned));
    double* base = (double*)p; p += pad((size_t)R * kBlkSteps * sizeof(double));
    F* sub = (F*)p;
    DNP_CHECK_HIP(hipMemsetAsync(workspace, 0, pad((size_t)R * sizeof(int)) + pad((size_t)R * (size_t)N * sizeof(unsigned)), stream));
    DNP_CHECK_HIP(hipMemsetAsync(weights, 0, (size_t)R * (size_t)N * sizeof(F), stream));     // rows kernel reads them from block 1 on
    hipLaunchKernelGGL(xie_perm_check_kernel, dim3((unsigned)ceil_div(R * N, (int64_t)256)), dim3(256), 0, stream, order, N, R, marks, flags);
    constexpr int kVec = sizeof(F) == 8 ? 2 : 4;
    const bool wide = N % kVec == 0 && (reinterpret_cast<uintptr_t>(M) & 15) == 0 && (reinterpret_cast<uintptr_t>(weights) & 15) == 0;
    for (int64_t t0 = 0; t0 < N; t0 += kBlkSteps) {
        const int nb = (int)((N - t0) < kBlkSteps ? (N - t0) : kBlkSteps);
        const dim3 grid((unsigned)nb, (unsigned)R);
        if (wide) hipLaunchKernelGGL((xie_block_rows_kernel<F, kVec>), grid, dim3(256), 0, stream, M, N, order, weights, t0, nb, flags, base, sub);
        else hipLaunchKernelGGL((xie_block_rows_kernel<F, 1>), grid, dim3(256), 0, stream, M, N, order, weights, t0, nb, flags, base, sub);
        hipLaunchKernelGGL((xie_block_solve_kernel<F>), dim3((unsigned)R), dim3(64), 0, stream, order, N, t0, nb, flags, base, sub, weights, inter);
    }
    DNP_CHECK_HIP(hipGetLastError());
    // rows that are not permutations: the row-per-step kernel, which zeroes and fills their outputs itself
    return run_xie_order<F>(M, N, order, R, weights, inter, stream, flags);
}

template <typename F>
static int run_xie_rowdots(const F* M, int64_t N, const F* weights, int64_t R, F* out, hipStream_t stream) {
    clear_error();
    DNP_REQUIRE(N >= 0 && R >= 0, "negative size");
    if (N == 0 || R == 0) return DNP_OK;
    DNP_REQUIRE(M && weights && out, "NULL pointer");
    constexpr int kVec = sizeof(F) == 4 ? 4 : 2;
    const bool wide = N % kVec == 0 && (reinterpret_cast<uintptr_t>(M) & 15) == 0 && (reinterpret_cast<uintptr_t>(weights) & 15) == 0;
    const dim3 grid((unsigned)ceil_div(N, (int64_t)4 * (sizeof(F) == 4 ? 8 : 4)));
    for (int64_t r0 = 0; r0 < R; r0 += kDotOrders) {
        if (wide) hipLaunchKernelGGL((xie_rowdots_kernel<F, kVec>), grid, dim3(256), 0, stream, M, N, weights, R, r0, out);
        else hipLaunchKernelGGL((xie_rowdots_kernel<F, 1>), grid, dim3(256), 0, stream, M, N, weights, R, r0, out);
    }
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

extern "C" {

int dnp_xie_pairs_f32(const float* src, int64_t S, int64_t ld_src, const float* tgt, int64_t T, int64_t ld_tgt,
                      float C, int vector_out, float* out, void* stream) {
    return run_xie_pairs<float>(src, S, ld_src, tgt, T, ld_tgt, C, vector_out, out, (hipStream_t)stream);
}

int dnp_xie_pairs_f64(const double* src, int64_t S, int64_t ld_src, const double* tgt, int64_t T, int64_t ld_tgt,
                      double C, int vector_out, double* out, void* stream) {
    return run_xie_pairs<double>(src, S, ld_src, tgt, T, ld_tgt, C, vector_out, out, (hipStream_t)stream);
}

int dnp_xie_knn_f32(const float* src, int64_t S, int64_t ld_src, const float* tgt, int64_t T, int64_t ld_tgt, int64_t k,
                    double* kth_d2, int64_t* kth_idx, void* stream) {
    return run_xie_knn<float>(src, S, ld_src, tgt, T, ld_tgt, k, kth_d2, kth_idx, (hipStream_t)stream);
}

int dnp_xie_knn_f64(const double* src, int64_t S, int64_t ld_src, const double* tgt, int64_t T, int64_t ld_tgt, int64_t k,
                    double* kth_d2, int64_t* kth_idx, void* stream) {
    return run_xie_knn<double>(src, S, ld_src, tgt, T, ld_tgt, k, kth_d2, kth_idx, (hipStream_t)stream);
}

int dnp_xie_pairs_knn_f32(const float* src, int64_t S, int64_t ld_src, const float* tgt, int64_t T, int64_t ld_tgt,
                          float C, int vector_out, const double* kth_d2, const int64_t* kth_idx, float* out, void* stream) {
    clear_error();
    DNP_REQUIRE(kth_d2 && kth_idx, "NULL kNN thresholds");
    return run_xie_pairs<float>(src, S, ld_src, tgt, T, ld_tgt, C, vector_out, out, (hipStream_t)stream, kth_d2, kth_idx);
}

int dnp_xie_pairs_knn_f64(const double* src, int64_t S, int64_t ld_src, const double* tgt, int64_t T, int64_t ld_tgt,
                          double C, int vector_out, const double* kth_d2, const int64_t* kth_idx, double* out, void* stream) {
    clear_error();
    DNP_REQUIRE(kth_d2 && kth_idx, "NULL kNN thresholds");
    return run_xie_pairs<double>(src, S, ld_src, tgt, T, ld_tgt, C, vector_out, out, (hipStream_t)stream, kth_d2, kth_idx);
}

int dnp_xie_order_f32(const float* M, int64_t N, const int64_t* order, int64_t R, float* weights, float* inter,
                      void* stream) {
    return run_xie_order<float>(M, N, order, R, weights, inter, (hipStream_t)stream);
}

int dnp_xie_order_f64(const double* M, int64_t N, const int64_t* order, int64_t R, double* weights, double* inter,
                      void* stream) {
    return run_xie_order<double>(M, N, order, R, weights, inter, (hipStream_t)stream);
}

size_t dnp_xie_order_workspace_bytes(int64_t N, int64_t R, int elem_bytes) {
    if (N <= 0 || R <= 0 || N < DNP_XIE_BLOCKED_FROM) return 256;
    return xie_order_workspace(N, R, (size_t)(elem_bytes == 8 ? 8 : 4));
}

int dnp_xie_order_blocked_f32(const float* M, int64_t N, const int64_t* order, int64_t R, float* weights, float* inter,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return run_xie_order_blocked<float>(M, N, order, R, weights, inter, workspace, workspace_bytes, (hipStream_t)stream);
}

int dnp_xie_order_blocked_f64(const double* M, int64_t N, const int64_t* order, int64_t R, double* weights, double* inter,
                              void* workspace, size_t workspace_bytes, void* stream) {
    return run_xie_order_blocked<double>(M, N, order, R, weights, inter, workspace, workspace_bytes, (hipStream_t)stream);
}

int dnp_xie_rowdots_f32(const float* M, int64_t N, const float* weights, int64_t R, float* out, void* stream) {
    return run_xie_rowdots<float>(M, N, weights, R, out, (hipStream_t)stream);
}

int dnp_xie_rowdots_f64(const double* M, int64_t N, const double* weights, int64_t R, double* out, void* stream) {
    return run_xie_rowdots<double>(M, N, weights, R, out, (hipStream_t)stream);
}

}  // extern "C"
