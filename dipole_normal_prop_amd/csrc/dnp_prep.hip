// dnp_prep.hip - the steps directly in front of / behind the pair kernels in the patch drivers, moved off the
// host:
//   dnp_patch_pca_*    per-patch mean + 3x3 covariance + eigen-decomposition in fp64, one workgroup per patch
//                      (util.pca_eigen_values util.py:495-500, the start-patch rule field_utils.py:230-233 /
//                      :303-306, inference_utils.fix_n_filter :52-71, util.orient_center util.py:39-44)
//   dnp_patch_greedy   the greedy loop of field_utils.py:314-324 / :242-254 on the P x P interaction matrix,
//                      one wavefront, no host round trip
//   dnp_merge_cells    the order-dependent merge of small voxel cells (util.merge_nodes util.py:448-492);
//                      sequential by definition, so it is a HOST function on the cell table (a few thousand rows)
#include <math.h>

#include <unordered_map>
#include <vector>

#include "dnp_common.h"

namespace dnp {

// ---- deterministic block sum of NV doubles per thread (256 threads): wave tree, then the 4 waves in order ----
template <int NV>
__device__ __forceinline__ void block_sum(double (&v)[NV], double* lds /* [4][NV] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NV; ++c)
        for (int off = 32; off > 0; off >>= 1) v[c] += __shfl_down(v[c], off, 64);
    __syncthreads();                                   // previous users of lds are done
    if (lane == 0)
#pragma unroll
        for (int c = 0; c < NV; ++c) lds[wave * NV + c] = v[c];
    __syncthreads();
#pragma unroll
    for (int c = 0; c < NV; ++c) v[c] = (lds[0 * NV + c] + lds[1 * NV + c]) + (lds[2 * NV + c] + lds[3 * NV + c]);
}

// cyclic Jacobi on a symmetric 3x3 (fp64): a -> diag, v -> eigenvectors in columns
__device__ inline void jacobi3(double a[3][3], double v[3][3]) {
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) v[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 32; ++sweep) {
        const double off = fabs(a[0][1]) + fabs(a[0][2]) + fabs(a[1][2]);
        const double diag = fabs(a[0][0]) + fabs(a[1][1]) + fabs(a[2][2]);
        if (off == 0.0 || off <= 1e-300 || off < 1e-22 * diag) break;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (a[p][q] == 0.0) continue;
                const double theta = (a[q][q] - a[p][p]) / (2.0 * a[p][q]);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
                const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                const double apq = a[p][q];
                a[p][p] -= t * apq;
                a[q][q] += t * apq;
                a[p][q] = a[q][p] = 0.0;
                const int r = 3 - p - q;
                const double arp = a[r][p], arq = a[r][q];
                a[r][p] = a[p][r] = c * arp - s * arq;
                a[r][q] = a[q][r] = s * arp + c * arq;
                for (int k = 0; k < 3; ++k) {
                    const double vkp = v[k][p], vkq = v[k][q];
                    v[k][p] = c * vkp - s * vkq;
                    v[k][q] = s * vkp + c * vkq;
                }
            }
    }
}

template <typename F>
__global__ __launch_bounds__(256) void patch_pca_kernel(const F* __restrict__ pts, int64_t ld,
                                                        const int64_t* __restrict__ off,
                                                        const int64_t* __restrict__ idx, double* __restrict__ mean,
                                                        double* __restrict__ evals, double* __restrict__ evecs) {
    __shared__ double red[4 * 6];
    __shared__ double mu[3];
    const int64_t p = blockIdx.x, lo = off[p], hi = off[p + 1];
    const double n = (double)(hi - lo);
    double s[3] = {0.0, 0.0, 0.0};
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const F* r = pts + (idx ? idx[i] : i) * ld;
        s[0] += (double)r[0]; s[1] += (double)r[1]; s[2] += (double)r[2];
    }
    block_sum<3>(s, red);
    if (threadIdx.x == 0)
        for (int c = 0; c < 3; ++c) mu[c] = (hi > lo) ? s[c] / n : 0.0;
    __syncthreads();
    const double mx = mu[0], my = mu[1], mz = mu[2];
    double m2[6] = {0, 0, 0, 0, 0, 0};                  // xx xy xz yy yz zz about the mean
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const F* r = pts + (idx ? idx[i] : i) * ld;
        const double x = (double)r[0] - mx, y = (double)r[1] - my, z = (double)r[2] - mz;
        m2[0] += x * x; m2[1] += x * y; m2[2] += x * z; m2[3] += y * y; m2[4] += y * z; m2[5] += z * z;
    }
    block_sum<6>(m2, red);
    if (threadIdx.x != 0) return;
    double a[3][3], v[3][3];
    const double inv = (hi > lo) ? 1.0 / n : 0.0;
    a[0][0] = m2[0] * inv; a[0][1] = a[1][0] = m2[1] * inv; a[0][2] = a[2][0] = m2[2] * inv;
    a[1][1] = m2[3] * inv; a[1][2] = a[2][1] = m2[4] * inv; a[2][2] = m2[5] * inv;
    jacobi3(a, v);
    int o[3] = {0, 1, 2};                               // ascending eigenvalues
    for (int i = 0; i < 2; ++i)
        for (int j = 0; j < 2 - i; ++j)
            if (a[o[j]][o[j]] > a[o[j + 1]][o[j + 1]]) { const int t = o[j]; o[j] = o[j + 1]; o[j + 1] = t; }
    for (int c = 0; c < 3; ++c) mean[p * 3 + c] = mu[c];
    for (int k = 0; k < 3; ++k) {
        const int col = o[k];
        evals[p * 3 + k] = a[col][col];
        // sign convention (an eigenvector's sign is arbitrary; LAPACK's is an implementation detail):
        // the component of largest magnitude is positive, ties to the lowest index
        int big = 0;
        for (int c = 1; c < 3; ++c)
            if (fabs(v[c][col]) > fabs(v[big][col])) big = c;
        const double sg = v[big][col] < 0.0 ? -1.0 : 1.0;
        for (int c = 0; c < 3; ++c) evecs[p * 9 + c * 3 + k] = sg * v[c][col];   // [P][row c][eigen k] as torch's v
    }
}

// ---- greedy loop on W: one wavefront, lane l owns patches l, l+64, ... (EPL per lane) -------------------------
// I_j = sum_{k visited} sigma_k W[k][j]; pick the first maximum of |I_j| over the unvisited patches in patch
// order (torch.argmax over the reference's `remaining` list, which stays in patch order), flip when I_j < 0.
// A step is a dependent chain (argmax -> row fetch -> add), so its latency is the whole cost: the wave argmax
// runs its four intra-row rounds on DPP (quad_perm / row_half_mirror / row_mirror: register-to-register) and only
// the two cross-row rounds through ds_bpermute; the winner's signed value is fetched with v_readlane.
template <int CTRL>
__device__ __forceinline__ int dpp_i32(int v) {
    return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false);
}
template <int CTRL>
__device__ __forceinline__ double dpp_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)dpp_i32<CTRL>((int)(unsigned)(b & 0xffffffffull));
    const unsigned hi = (unsigned)dpp_i32<CTRL>((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void take_better(double& bv, int& bj, double ov, int oj) {
    if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; }
}
// (bv, bj) -> the wave's (max bv, smallest bj among equals) in every lane
__device__ __forceinline__ void wave_argmax(double& bv, int& bj) {
    take_better(bv, bj, dpp_f64<0xB1>(bv), dpp_i32<0xB1>(bj));     // quad_perm [1,0,3,2]: lane ^ 1
    take_better(bv, bj, dpp_f64<0x4E>(bv), dpp_i32<0x4E>(bj));     // quad_perm [2,3,0,1]: lane ^ 2
    take_better(bv, bj, dpp_f64<0x141>(bv), dpp_i32<0x141>(bj));   // row_half_mirror: quads of an 8-lane half swap
    take_better(bv, bj, dpp_f64<0x140>(bv), dpp_i32<0x140>(bj));   // row_mirror: halves of a 16-lane row swap
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1) take_better(bv, bj, __shfl_xor(bv, off, 64), __shfl_xor(bj, off, 64));
}

// the wave's maximum of bv as a wave-uniform value (values only: three instructions per round instead of ten).  Four
// butterfly rounds on DPP leave every 16-lane row's maximum in all of its lanes; the rows are then folded with row_bcast:15
// (lane 15 of rows 0 / 2 into rows 1 / 3) and row_bcast:31 (lane 31 into rows 2 / 3) - register to register, where the two
// ds_bpermute rounds of an all-lanes butterfly cost an LDS round trip each - and lane 63 is read with v_readlane.
template <int CTRL, int ROWMASK>
__device__ __forceinline__ double dpp_f64_rows(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(b & 0xffffffffull), (int)(unsigned)(b & 0xffffffffull), CTRL, ROWMASK, 0xf, false);
    const unsigned hi = (unsigned)__builtin_amdgcn_update_dpp((int)(unsigned)(b >> 32), (int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xf, false);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
// v_max_f64 as ONE instruction: __builtin_fmax on a value that came through a DPP move (an integer bit cast) is preceded by a
// canonicalising v_max_f64 x, x per operand and a register copy for the DPP's `old` operand - seven instructions per butterfly
// round where three do (the operands here are |I| >= 0, -1 for "none" or +inf: never NaN)
__device__ __forceinline__ double max_f64_raw(double a, double b) {
    double d;
    asm("v_max_f64 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
    return d;
}
// DPP move of a double where every lane has a source lane (quad_perm / row_mirror forms): no `old` value to preserve
template <int CTRL, int ROWMASK = 0xf>
__device__ __forceinline__ double dpp_f64_all(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b & 0xffffffffull), CTRL, ROWMASK, 0xf, ROWMASK == 0xf);
    const unsigned hi = (unsigned)__builtin_amdgcn_mov_dpp((int)(unsigned)(b >> 32), CTRL, ROWMASK, 0xf, ROWMASK == 0xf);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ double wave_max_f64(double v) {
    v = max_f64_raw(v, dpp_f64_all<0xB1>(v));
    v = max_f64_raw(v, dpp_f64_all<0x4E>(v));
    v = max_f64_raw(v, dpp_f64_all<0x141>(v));
    v = max_f64_raw(v, dpp_f64_all<0x140>(v));
    // the four row maxima meet in lane 63: row_bcast:15 (lane 15 of rows 0 / 2 into rows 1 / 3), row_bcast:31 (lane 31 into rows 2 / 3).
    // The rows a move does not write are left with whatever the destination register held - only lane 63 is read, and what feeds
    // it (row 3 <- row 2's lane 47 in the first move, row 3 <- row 1's lane 31 in the second) is written or untouched input.
    v = max_f64_raw(v, dpp_f64_all<0x142, 0xa>(v));
    v = max_f64_raw(v, dpp_f64_all<0x143, 0xc>(v));
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b & 0xffffffffull), 63);
    const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(b >> 32), 63);
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}

template <int EPL>
__global__ __launch_bounds__(64) void patch_greedy_kernel(const double* __restrict__ W, int P,
                                                          const int64_t* __restrict__ start_ptr,
                                                          int64_t* __restrict__ order, double* __restrict__ sigma,
                                                          double* __restrict__ chosen) {
    // the trace stays in LDS until the end: a global store inside the loop would sit in front of the next row's
    // loads in the wave's vmcnt queue, and every step would wait for a store round trip to memory
    __shared__ int order_s[64 * EPL];
    __shared__ double chosen_s[64 * EPL];
    const int lane = threadIdx.x;
    double inter[EPL];
    unsigned long long visited = 0, negative = 0;        // bit e <-> patch e*64 + lane
    int cur = (int)start_ptr[0];
    if (cur < 0 || cur >= P) cur = 0;
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        inter[e] = 0.0;
        if (e * 64 + lane >= P) visited |= 1ull << e;
    }
#ifndef DNP_PG_PRETOUCH   // 0: A/B builds without the pass over W in front of the loop (tools/gpu_pg_time.py)
#define DNP_PG_PRETOUCH 1
#endif
    if (DNP_PG_PRETOUCH && (int64_t)P * P * 8 <= (2 << 20)) {
        // W was written by other CUs a moment ago: a row load of the loop is served by the memory-side cache (~230 ns) until this
        // XCD's L2 has the line (~85 ns).  One pass over the matrix puts all of it there.  (16-byte loads when W is 16-byte
        // aligned - the ABI promises a const double*, i.e. 8 -, 8-byte loads otherwise; the sums only keep the loads alive.)
        double t = 0.0;
        if ((reinterpret_cast<uintptr_t>(W) & 15) == 0) {
            const double2* w2 = reinterpret_cast<const double2*>(W);
            const int64_t n2 = (int64_t)P * P / 2;
#pragma unroll 16
            for (int64_t i = lane; i < n2; i += 64) { const double2 v = w2[i]; t += v.x + v.y; }
        } else {
            const int64_t n1 = (int64_t)P * P;
#pragma unroll 16
            for (int64_t i = lane; i < n1; i += 64) t += W[i];
        }
        asm volatile("" ::"v"(t));                       // a sink: the loads stay, nothing is stored
    }
    double s = 1.0;                                       // the start patch is not flipped
    // The row of the patch a step adds is requested the moment that patch is known - before the winner's signed value is
    // extracted and the trace is written - so that the fetch (~0.1 us from this XCD's L2) runs under those ~70
    // instructions instead of behind them (round 5; DNP_PG_EARLY_ROW=0 builds keep the fetch at the top of the step).
#ifndef DNP_PG_EARLY_ROW
#define DNP_PG_EARLY_ROW 1
#endif
    double r[EPL];
    auto fetch_row = [&](int patch) {
        const double* row = W + (int64_t)patch * P;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {                  // unconditional (clamped) loads: all EPL in flight at once -
            const int j = e * 64 + lane;                 // a predicated load per entry would be EPL serial round trips
            r[e] = row[j < P ? j : P - 1];
        }
    };
    if (DNP_PG_EARLY_ROW) fetch_row(cur);
    for (int step = 0; step < P; ++step) {
        if (lane == 0) order_s[step] = cur;
        if ((cur & 63) == lane) {
            visited |= 1ull << (cur >> 6);
            if (s < 0.0) negative |= 1ull << (cur >> 6);
        }
        if (!DNP_PG_EARLY_ROW) fetch_row(cur);
#pragma unroll
        for (int e = 0; e < EPL; ++e) inter[e] += s * r[e];   // s = +-1: the product is exact.  (Entries beyond P add the clamped
        //                                                       load's value: they are "visited" from the start and never read.)
        if (step + 1 == P) break;
        // first maximum of |I_j| in patch order; a NaN counts as the maximum, as in torch.argmax
        double bv = -1.0;
        int bj = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < EPL; ++e) {                  // ascending j within the lane
            // min(|I|, +inf): v_min_f64 returns the operand that is not a NaN, so a NaN |I| becomes +inf in one instruction
            double a;
            asm("v_min_f64 %0, |%1|, %2" : "=v"(a) : "v"(inter[e]), "v"(__builtin_huge_val()));
            if (!((visited >> e) & 1ull) && a > bv) { bv = a; bj = e * 64 + lane; }
        }
#ifndef DNP_PG_FAST_ARGMAX   // 0: A/B builds that keep the (value, index) butterfly on every step
#define DNP_PG_FAST_ARGMAX 1
#endif
        // The step is one dependent chain, so the argmax's length is paid 255 times: first the wave's maximum VALUE alone
        // (no index travels, no tie rule: 3 instructions per butterfly round), then a ballot of the lanes that hold it.  One
        // such lane (what happens but for exact fp64 ties): its candidate is the winner.  Several: the full (value, index)
        // butterfly with the first-in-patch-order rule - the same result either way.
        bool decided = false;
        if (DNP_PG_FAST_ARGMAX) {
            const double m = wave_max_f64(bv);           // no NaN here: a NaN |I| was made +inf above, "none" is -1
            const unsigned long long tied = __ballot(bv == m && bj != 0x7fffffff);
            if (__builtin_popcountll(tied) == 1) {
                cur = __builtin_amdgcn_readlane(bj, (int)__builtin_ctzll(tied));
                decided = true;
            }
        }
        if (!decided) {
            wave_argmax(bv, bj);
            cur = __builtin_amdgcn_readfirstlane(bj);    // wave-uniform
        }
        if (DNP_PG_EARLY_ROW) {
            fetch_row(cur);
            __builtin_amdgcn_sched_barrier(0);           // the requests stay HERE, in front of the extraction below
        }
        // the winner's signed interaction lives in lane cur & 63, entry cur >> 6
        double mine = 0.0;
#pragma unroll
        for (int e = 0; e < EPL; ++e) mine = (e == (cur >> 6)) ? inter[e] : mine;
        const unsigned long long mb = __builtin_bit_cast(unsigned long long, mine);
        const unsigned lo = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mb & 0xffffffffull), cur & 63);
        const unsigned hi = (unsigned)__builtin_amdgcn_readlane((int)(unsigned)(mb >> 32), cur & 63);
        const double bi = __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
        s = (bi < 0.0) ? -1.0 : 1.0;                     // `if interaction[max] < 0: flip`
        if (lane == 0) chosen_s[step] = bi;
    }
    __syncthreads();
    for (int i = lane; i < P; i += 64) {
        order[i] = order_s[i];
        if (i + 1 < P) chosen[i] = chosen_s[i];
    }
#pragma unroll
    for (int e = 0; e < EPL; ++e) {
        const int j = e * 64 + lane;
        if (j < P) sigma[j] = ((negative >> e) & 1ull) ? -1.0 : 1.0;
    }
}

// ---- the same loop for more patches than one wavefront holds: ONE workgroup of 1024 threads, thread t owns patches
// t, t + 1024, ... (EPT per thread); one barrier per step.  Every wave reduces its candidates as above, carrying the
// signed interaction along, and leaves {|I|, j, I} in LDS (two sets by step parity: a set is rewritten only after
// the barrier that follows its last read); every thread then folds the 16 records in wave order, so the first
// maximum in patch order wins exactly as in the one-wavefront kernel.
constexpr int kGreedyBlock = 1024;
__device__ __forceinline__ void take_better3(double& bv, int& bj, double& sv, double ov, int oj, double osv) {
    if (ov > bv || (ov == bv && oj < bj)) { bv = ov; bj = oj; sv = osv; }
}
// the same inside one 16-lane row only (the four DPP rounds)
__device__ __forceinline__ void row_argmax3(double& bv, int& bj, double& sv) {
    take_better3(bv, bj, sv, dpp_f64<0xB1>(bv), dpp_i32<0xB1>(bj), dpp_f64<0xB1>(sv));
    take_better3(bv, bj, sv, dpp_f64<0x4E>(bv), dpp_i32<0x4E>(bj), dpp_f64<0x4E>(sv));
    take_better3(bv, bj, sv, dpp_f64<0x141>(bv), dpp_i32<0x141>(bj), dpp_f64<0x141>(sv));
    take_better3(bv, bj, sv, dpp_f64<0x140>(bv), dpp_i32<0x140>(bj), dpp_f64<0x140>(sv));
}
__device__ __forceinline__ double readfirstlane_f64(double v) {
    const unsigned long long b = __builtin_bit_cast(unsigned long long, v);
    const unsigned lo = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b & 0xffffffffull));
    const unsigned hi = (unsigned)__builtin_amdgcn_readfirstlane((int)(unsigned)(b >> 32));
    return __builtin_bit_cast(double, ((unsigned long long)hi << 32) | lo);
}
__device__ __forceinline__ void wave_argmax3(double& bv, int& bj, double& sv) {
    take_better3(bv, bj, sv, dpp_f64<0xB1>(bv), dpp_i32<0xB1>(bj), dpp_f64<0xB1>(sv));
    take_better3(bv, bj, sv, dpp_f64<0x4E>(bv), dpp_i32<0x4E>(bj), dpp_f64<0x4E>(sv));
    take_better3(bv, bj, sv, dpp_f64<0x141>(bv), dpp_i32<0x141>(bj), dpp_f64<0x141>(sv));
    take_better3(bv, bj, sv, dpp_f64<0x140>(bv), dpp_i32<0x140>(bj), dpp_f64<0x140>(sv));
#pragma unroll
    for (int off = 16; off <= 32; off <<= 1)
        take_better3(bv, bj, sv, __shfl_xor(bv, off, 64), __shfl_xor(bj, off, 64), __shfl_xor(sv, off, 64));
}

// The trace (order, chosen) goes through a two-block LDS ring of 64 steps that the LAST wave copies out every 64
// steps: a global store inside the loop sits in front of the next row's loads in the storing wave's vmcnt queue and
// costs a store round trip per step (3.0 against 1.3 us per step measured).
template <int EPT>
__global__ __launch_bounds__(kGreedyBlock) void patch_greedy_block_kernel(const double* __restrict__ W, int P,
                                                                          const int64_t* __restrict__ start_ptr,
                                                                          int64_t* __restrict__ order,
                                                                          double* __restrict__ sigma,
                                                                          double* __restrict__ chosen) {
    constexpr int kWaves = kGreedyBlock / 64;
    __shared__ double best_v[2][kWaves], best_s[2][kWaves];
    __shared__ int best_j[2][kWaves];
    __shared__ int order_r[2][64];
    __shared__ double chosen_r[2][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double inter[EPT];
    unsigned visited = 0, negative = 0;                   // bit e <-> patch e*1024 + tid
    int cur = (int)start_ptr[0];
    if (cur < 0 || cur >= P) cur = 0;
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        inter[e] = 0.0;
        if (e * kGreedyBlock + tid >= P) visited |= 1u << e;
    }
    double s = 1.0;                                       // the start patch is not flipped
    int step = 0;
    for (; step < P; ++step) {
        if (tid == 0) order_r[(step >> 6) & 1][step & 63] = cur;
        if ((cur & (kGreedyBlock - 1)) == tid) {
            visited |= 1u << (cur / kGreedyBlock);
            if (s < 0.0) negative |= 1u << (cur / kGreedyBlock);
        }
        const double* row = W + (int64_t)cur * P;
        double r[EPT];
#pragma unroll
        for (int e = 0; e < EPT; ++e) {                   // unconditional (clamped) loads, all in flight at once
            const int j = e * kGreedyBlock + tid;
            r[e] = row[j < P ? j : P - 1];
        }
#pragma unroll
        for (int e = 0; e < EPT; ++e)
            if (e * kGreedyBlock + tid < P) inter[e] += s * r[e];
        if (step + 1 == P) break;
        double bv = -1.0, sv = 0.0;
        int bj = 0x7fffffff;
#pragma unroll
        for (int e = 0; e < EPT; ++e) {                   // ascending j within the thread
            double a = fabs(inter[e]);
            if (a != a) a = __builtin_huge_val();         // a NaN counts as the maximum, as in torch.argmax
            if (!((visited >> e) & 1u) && a > bv) { bv = a; bj = e * kGreedyBlock + tid; sv = inter[e]; }
        }
        wave_argmax3(bv, bj, sv);
        const int par = step & 1;
        if (lane == 0) { best_v[par][wave] = bv; best_j[par][wave] = bj; best_s[par][wave] = sv; }
        __syncthreads();
        // every trace entry of the steps before this one is in LDS now: the last wave copies a finished block out
        if (wave == kWaves - 1 && step > 0 && (step & 63) == 0) {
            const int blk = (step >> 6) - 1;
            order[blk * 64 + lane] = order_r[blk & 1][lane];
            chosen[blk * 64 + lane] = chosen_r[blk & 1][lane];
        }
        // lanes 0..15 of every wave take one record each (3 LDS reads per wave - every thread folding all 16 records
        // itself is 48 LDS reads per thread and made the step LDS-issue bound: 3.1 us), four DPP rounds inside that
        // 16-lane row leave the block's winner in lane 0, read out with v_readfirstlane
        bv = (lane < kWaves) ? best_v[par][lane & (kWaves - 1)] : -1.0;
        bj = (lane < kWaves) ? best_j[par][lane & (kWaves - 1)] : 0x7fffffff;
        sv = (lane < kWaves) ? best_s[par][lane & (kWaves - 1)] : 0.0;
        row_argmax3(bv, bj, sv);
        cur = __builtin_amdgcn_readfirstlane(bj);         // the same in every thread
        sv = readfirstlane_f64(sv);
        s = (sv < 0.0) ? -1.0 : 1.0;                      // `if interaction[max] < 0: flip`
        if (tid == 0) chosen_r[(step >> 6) & 1][step & 63] = sv;
    }
    __syncthreads();
    // the last block, and the one before it (its in-loop copy is triggered by a step of the last block that may not
    // have come; both ring halves are still intact, so copying it again is harmless).  order: P entries, chosen: P - 1
    if (wave == kWaves - 1) {
        const int last = (P - 1) >> 6;
        for (int blk = (last > 0 ? last - 1 : 0); blk <= last; ++blk) {
            const int i = blk * 64 + lane;
            if (i < P) order[i] = order_r[blk & 1][lane];
            if (i < P - 1) chosen[i] = chosen_r[blk & 1][lane];
        }
    }
#pragma unroll
    for (int e = 0; e < EPT; ++e) {
        const int j = e * kGreedyBlock + tid;
        if (j < P) sigma[j] = ((negative >> e) & 1u) ? -1.0 : 1.0;
    }
}

// ---- the patch-sorted working layout in ONE launch: swork[i] = pts[idx[i]], sorted_patch[i] = p for the rows
// i in [off[p], off[p+1]) of every patch p (one workgroup per patch); replaces a repeat_interleave (three launches), a
// row gather and their temporaries in the drivers' set-up
template <typename F>
__global__ __launch_bounds__(256) void patch_layout_kernel(const F* __restrict__ pts, int64_t ld,
                                                           const int64_t* __restrict__ off,
                                                           const int64_t* __restrict__ idx, F* __restrict__ swork,
                                                           int64_t* __restrict__ sorted_patch) {
    const int64_t p = blockIdx.x, lo = off[p], hi = off[p + 1];
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        const F* r = pts + idx[i] * ld;
        F* o = swork + i * 6;
        o[0] = r[0]; o[1] = r[1]; o[2] = r[2]; o[3] = r[3]; o[4] = r[4]; o[5] = r[5];
        sorted_patch[i] = p;
    }
}

// ---- the tail of the batched patch drivers in ONE launch (field_utils.py:322-323 flips, :337-342 diffuse sign pass,
// :344-346 weight un-scaling), on the patch-sorted working cloud: for sorted row t
//   n = work[t].n * sigma[patch(t)]                                    (patch(t) < 0: not in any patch, untouched)
//   diffuse and listed[patch(t)]:  n *= ((float)E64[t] . n > 0) ? +1 : -1     (fp32 products, summed in order)
//   weights:  n /= w[t]
//   out[perm[t]] (row stride ld_out, columns 3..5) = n        - the caller's point order and tensor
// F = the working cloud's precision (float / double: the field is rounded to it before the dot, as the reference's E is
// held in the cloud's dtype), OUT = the caller's tensor.
template <typename F, typename OUT>
__global__ __launch_bounds__(256) void patch_finish_kernel(const F* __restrict__ work, int64_t ld, int64_t N,
                                                           const int64_t* __restrict__ point_patch,
                                                           const double* __restrict__ sigma,
                                                           const double* __restrict__ E64,
                                                           const unsigned char* __restrict__ listed,
                                                           const F* __restrict__ weights,
                                                           const int64_t* __restrict__ perm, OUT* __restrict__ out,
                                                           int64_t ld_out) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= N) return;
    const F* r = work + t * ld;
    F n0 = r[3], n1 = r[4], n2 = r[5];
    const int64_t k = point_patch[t];
    if (k >= 0) {
        const F sg = (F)sigma[k];
        n0 *= sg; n1 *= sg; n2 *= sg;
        if (E64 && (!listed || listed[k])) {
            const F e0 = (F)E64[t * 3 + 0], e1 = (F)E64[t * 3 + 1], e2 = (F)E64[t * 3 + 2];
            const F dot = (e0 * n0 + e1 * n1) + e2 * n2;       // products rounded separately, added in order (contraction off)
            const F s = dot > F(0) ? F(1) : F(-1);
            n0 *= s; n1 *= s; n2 *= s;
        }
    }
    if (weights) { const F w = weights[t]; n0 = n0 / w; n1 = n1 / w; n2 = n2 / w; }
    OUT* o = out + (perm ? perm[t] : t) * ld_out;
    o[3] = (OUT)n0; o[4] = (OUT)n1; o[5] = (OUT)n2;
}

// ---- the tail of the representatives driver for the points that are NOT representatives, in ONE launch
// (field_utils.py:251-252: a flipped patch flips its rest points too; :273-276: every non-representative point then takes the
// sign of the field E of all representatives): for entry i of patch p's rest list, row = rest_idx[i],
//   n = work[row].n * sigma[p];   n *= (E[i] . n > 0) ? +1 : -1      (products rounded separately, added in order)
// in place.  One workgroup per patch; every point is listed once (the callers' partition).  Replaces a segmented count, a
// parity, two gathers and five elementwise torch launches (~14 launches, round 5: profiles/r05_config3_kernels.txt).
template <typename F>
__global__ __launch_bounds__(256) void rest_finish_kernel(F* __restrict__ work, int64_t ld, const int64_t* __restrict__ rest_off,
                                                          const int64_t* __restrict__ rest_idx, const double* __restrict__ sigma,
                                                          const F* __restrict__ E) {
#pragma clang fp contract(off)
    const int64_t p = blockIdx.x, lo = rest_off[p], hi = rest_off[p + 1];
    const F sg = (F)sigma[p];
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) {
        F* r = work + rest_idx[i] * ld;
        F n0 = r[3] * sg, n1 = r[4] * sg, n2 = r[5] * sg;
        const F dot = (E[i * 3 + 0] * n0 + E[i * 3 + 1] * n1) + E[i * 3 + 2] * n2;
        const F s = dot > F(0) ? F(1) : F(-1);
        r[3] = n0 * s; r[4] = n1 * s; r[5] = n2 * s;
    }
}

// E64[t][c] (+)= sum_k sigma[k] * dE[k][t][c] over the K slabs held here, accumulated in fp64 in slab order.
// sigma is +-1, so every product is exact; a fp64 sum of a few thousand fp32 values is independent of the
// order to ~1e-16 relative, hence the same on one GPU and on eight.
template <typename F>
__global__ __launch_bounds__(256) void combine_signed_kernel(const F* __restrict__ dE, int64_t K, int64_t N3,
                                                             const double* __restrict__ sigma,
                                                             double* __restrict__ E, int accumulate) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= N3) return;
    double e = accumulate ? E[i] : 0.0;
    int64_t k = 0;
    for (; k + 4 <= K; k += 4) {                      // 4 independent loads in flight, added in slab order
        const F v0 = dE[(k + 0) * N3 + i], v1 = dE[(k + 1) * N3 + i], v2 = dE[(k + 2) * N3 + i],
                v3 = dE[(k + 3) * N3 + i];
        e += sigma[k + 0] * (double)v0; e += sigma[k + 1] * (double)v1;
        e += sigma[k + 2] * (double)v2; e += sigma[k + 3] * (double)v3;
    }
    for (; k < K; ++k) e += sigma[k] * (double)dE[k * N3 + i];
    E[i] = e;
}

}  // namespace dnp

using namespace dnp;

extern "C" {

static int patch_pca_check(const void* pts, int64_t ld, const int64_t* off, int64_t P, double* mean, double* evals,
                           double* evecs) {
    clear_error();
    DNP_REQUIRE(P >= 0, "negative P");
    if (P == 0) return 1;
    DNP_REQUIRE(pts && off && mean && evals && evecs, "NULL pointer");
    DNP_REQUIRE(ld >= 3, "ld_pts=%lld < 3", (long long)ld);
    DNP_REQUIRE(P <= INT32_MAX, "P too large");
    return DNP_OK;
}

int dnp_patch_pca_f32(const float* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                      int64_t P, double* mean, double* evals, double* evecs, void* stream) {
    const int rc = patch_pca_check(pts, ld_pts, patch_off, P, mean, evals, evecs);
    if (rc != DNP_OK) return rc > 0 ? DNP_OK : rc;
    hipLaunchKernelGGL((patch_pca_kernel<float>), dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pts, ld_pts,
                       patch_off, patch_idx, mean, evals, evecs);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_pca_f64(const double* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                      int64_t P, double* mean, double* evals, double* evecs, void* stream) {
    const int rc = patch_pca_check(pts, ld_pts, patch_off, P, mean, evals, evecs);
    if (rc != DNP_OK) return rc > 0 ? DNP_OK : rc;
    hipLaunchKernelGGL((patch_pca_kernel<double>), dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pts, ld_pts,
                       patch_off, patch_idx, mean, evals, evecs);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_greedy_max_patches(void) { return kGreedyBlock * 16; }

int dnp_patch_greedy(const double* W, int64_t P, const int64_t* start, int64_t* order, double* sigma,
                     double* chosen, void* stream) {
    clear_error();
    DNP_REQUIRE(P >= 0, "negative P");
    if (P == 0) return DNP_OK;
    DNP_REQUIRE(W && start && order && sigma, "NULL pointer");
    DNP_REQUIRE(P == 1 || chosen, "NULL chosen");
    DNP_REQUIRE(P <= dnp_patch_greedy_max_patches(), "P=%lld exceeds the %d patches of the device greedy loop",
                (long long)P, dnp_patch_greedy_max_patches());
    const hipStream_t st = (hipStream_t)stream;
#define DNP_LAUNCH_PG(E) \
    hipLaunchKernelGGL((patch_greedy_kernel<E>), dim3(1), dim3(64), 0, st, W, (int)P, start, order, sigma, chosen)
    // one wavefront up to 2048 patches (0.8 us per step at 256, 2.25 at 2048, 4.3 at 4096: 64 row loads per lane),
    // the one-workgroup kernel above (2.4-2.5 us per step up to 4096, 2.9 at 8192, 4.6 at 16384; tools/gpu_pg_time.py)
#ifndef DNP_PG_BLOCK_FROM        // A/B builds move the crossover
#define DNP_PG_BLOCK_FROM (64 * 32)
#endif
    if (P > DNP_PG_BLOCK_FROM) {
        if (P <= kGreedyBlock * 8)
            hipLaunchKernelGGL((patch_greedy_block_kernel<8>), dim3(1), dim3(kGreedyBlock), 0, st, W, (int)P, start, order, sigma, chosen);
        else
            hipLaunchKernelGGL((patch_greedy_block_kernel<16>), dim3(1), dim3(kGreedyBlock), 0, st, W, (int)P, start, order, sigma, chosen);
    } else if (P <= 64 * 2) DNP_LAUNCH_PG(2);      // (round 5: 2, 6, 12, 24 - a step's argmax scans EPL entries per lane and the
    else if (P <= 64 * 4) DNP_LAUNCH_PG(4);          // winner's value is picked out of EPL registers: P = 369 ran as EPL = 8)
    else if (P <= 64 * 6) DNP_LAUNCH_PG(6);
    else if (P <= 64 * 8) DNP_LAUNCH_PG(8);
    else if (P <= 64 * 12) DNP_LAUNCH_PG(12);
    else if (P <= 64 * 16) DNP_LAUNCH_PG(16);
    else if (P <= 64 * 24) DNP_LAUNCH_PG(24);
    else if (P <= 64 * 32) DNP_LAUNCH_PG(32);
    else DNP_LAUNCH_PG(64);
#undef DNP_LAUNCH_PG
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_combine_signed_f32(const float* dE, int64_t K, int64_t N, const double* sigma, int64_t P, int64_t p_lo,
                           double* E, int accumulate, void* stream) {
    clear_error();
    DNP_REQUIRE(K >= 0 && N >= 0 && P >= 0, "negative size");
    DNP_REQUIRE(p_lo >= 0 && p_lo + K <= P, "slabs [%lld,%lld) outside the %lld patches", (long long)p_lo,
                (long long)(p_lo + K), (long long)P);
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(E, "NULL E");
    DNP_REQUIRE(K == 0 || (dE && sigma), "NULL pointer");
    const int64_t N3 = N * 3;
    hipLaunchKernelGGL(combine_signed_kernel<float>, dim3((unsigned)ceil_div(N3, 256)), dim3(256), 0, (hipStream_t)stream,
                       dE, K, N3, sigma ? sigma + p_lo : nullptr, E, accumulate);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_combine_signed_f64(const double* dE, int64_t K, int64_t N, const double* sigma, int64_t P, int64_t p_lo,
                           double* E, int accumulate, void* stream) {
    clear_error();
    DNP_REQUIRE(K >= 0 && N >= 0 && P >= 0, "negative size");
    DNP_REQUIRE(p_lo >= 0 && p_lo + K <= P, "slabs [%lld,%lld) outside the %lld patches", (long long)p_lo,
                (long long)(p_lo + K), (long long)P);
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(E, "NULL E");
    DNP_REQUIRE(K == 0 || (dE && sigma), "NULL pointer");
    const int64_t N3 = N * 3;
    hipLaunchKernelGGL(combine_signed_kernel<double>, dim3((unsigned)ceil_div(N3, 256)), dim3(256), 0, (hipStream_t)stream,
                       dE, K, N3, sigma ? sigma + p_lo : nullptr, E, accumulate);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_layout_f32(const float* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                         int64_t P, float* swork, int64_t* sorted_patch, void* stream) {
    clear_error();
    DNP_REQUIRE(P >= 0, "negative P");
    if (P == 0) return DNP_OK;
    DNP_REQUIRE(pts && patch_off && patch_idx && swork && sorted_patch, "NULL pointer");
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    hipLaunchKernelGGL(patch_layout_kernel<float>, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pts, ld_pts, patch_off,
                       patch_idx, swork, sorted_patch);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_layout_f64(const double* pts, int64_t ld_pts, const int64_t* patch_off, const int64_t* patch_idx,
                         int64_t P, double* swork, int64_t* sorted_patch, void* stream) {
    clear_error();
    DNP_REQUIRE(P >= 0, "negative P");
    if (P == 0) return DNP_OK;
    DNP_REQUIRE(pts && patch_off && patch_idx && swork && sorted_patch, "NULL pointer");
    DNP_REQUIRE(ld_pts >= 6, "ld_pts=%lld < 6", (long long)ld_pts);
    hipLaunchKernelGGL(patch_layout_kernel<double>, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, pts, ld_pts, patch_off,
                       patch_idx, swork, sorted_patch);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_finish_f32(const float* work, int64_t ld_work, int64_t N, const int64_t* point_patch,
                         const double* sigma, const double* E64, const unsigned char* listed, const float* weights,
                         const int64_t* perm, void* out, int64_t ld_out, int out_is_f64, void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0, "negative N");
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(work && point_patch && sigma && out, "NULL pointer");
    DNP_REQUIRE(ld_work >= 6 && ld_out >= 6, "row stride < 6");
    const dim3 grid((unsigned)ceil_div(N, 256));
    if (out_is_f64)
        hipLaunchKernelGGL((patch_finish_kernel<float, double>), grid, dim3(256), 0, (hipStream_t)stream, work, ld_work, N,
                           point_patch, sigma, E64, listed, weights, perm, (double*)out, ld_out);
    else
        hipLaunchKernelGGL((patch_finish_kernel<float, float>), grid, dim3(256), 0, (hipStream_t)stream, work, ld_work, N,
                           point_patch, sigma, E64, listed, weights, perm, (float*)out, ld_out);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_patch_finish_f64(const double* work, int64_t ld_work, int64_t N, const int64_t* point_patch,
                         const double* sigma, const double* E64, const unsigned char* listed, const double* weights,
                         const int64_t* perm, void* out, int64_t ld_out, int out_is_f64, void* stream) {
    clear_error();
    DNP_REQUIRE(N >= 0, "negative N");
    if (N == 0) return DNP_OK;
    DNP_REQUIRE(work && point_patch && sigma && out, "NULL pointer");
    DNP_REQUIRE(ld_work >= 6 && ld_out >= 6, "row stride < 6");
    const dim3 grid((unsigned)ceil_div(N, 256));
    if (out_is_f64)
        hipLaunchKernelGGL((patch_finish_kernel<double, double>), grid, dim3(256), 0, (hipStream_t)stream, work, ld_work, N,
                           point_patch, sigma, E64, listed, weights, perm, (double*)out, ld_out);
    else
        hipLaunchKernelGGL((patch_finish_kernel<double, float>), grid, dim3(256), 0, (hipStream_t)stream, work, ld_work, N,
                           point_patch, sigma, E64, listed, weights, perm, (float*)out, ld_out);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_rest_finish_f32(float* work, int64_t ld_work, const int64_t* rest_off, const int64_t* rest_idx, int64_t P,
                        const double* sigma, const float* E, void* stream) {
    clear_error();
    DNP_REQUIRE(P >= 0, "negative P");
    if (P == 0) return DNP_OK;
    DNP_REQUIRE(work && rest_off && rest_idx && sigma && E, "NULL pointer");
    DNP_REQUIRE(ld_work >= 6, "row stride < 6");
    hipLaunchKernelGGL(rest_finish_kernel<float>, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, work, ld_work, rest_off,
                       rest_idx, sigma, E);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

int dnp_rest_finish_f64(double* work, int64_t ld_work, const int64_t* rest_off, const int64_t* rest_idx, int64_t P,
                        const double* sigma, const double* E, void* stream) {
    clear_error();
    DNP_REQUIRE(P >= 0, "negative P");
    if (P == 0) return DNP_OK;
    DNP_REQUIRE(work && rest_off && rest_idx && sigma && E, "NULL pointer");
    DNP_REQUIRE(ld_work >= 6, "row stride < 6");
    hipLaunchKernelGGL(rest_finish_kernel<double>, dim3((unsigned)P), dim3(256), 0, (hipStream_t)stream, work, ld_work, rest_off,
                       rest_idx, sigma, E);
    DNP_CHECK_HIP(hipGetLastError());
    return DNP_OK;
}

// ---- host: merge of small voxel cells ---------------------------------------------------------------------------
static int merge_cells_body(const int32_t* cell_ijk, const int64_t* cell_size, int64_t C, int64_t min_patch,
                            int64_t* seq, int64_t* seq_off, int64_t* n_out, int32_t* sweeps_out) {
    clear_error();
    DNP_REQUIRE(C >= 0, "negative C");
    DNP_REQUIRE(seq_off && n_out, "NULL output pointer");
    DNP_REQUIRE(C == 0 || (cell_ijk && cell_size && seq), "NULL pointer");
    // voxel -> live cell that currently contains it.  Keys are packed 21 bits per axis (shifted by +1 so that the
    // -1 neighbours of index 0 stay non-negative).
    auto key = [](int64_t i, int64_t j, int64_t k) { return ((i + 1) << 42) | ((j + 1) << 21) | (k + 1); };
    std::unordered_map<int64_t, int32_t> owner;
    owner.reserve((size_t)C * 2);
    std::vector<std::vector<int32_t>> members((size_t)C);     // original cells of a live cell, concatenation order
    std::vector<int64_t> size((size_t)C);
    for (int64_t c = 0; c < C; ++c) {
        const int32_t* v = cell_ijk + c * 3;
        DNP_REQUIRE(v[0] >= 0 && v[1] >= 0 && v[2] >= 0 && v[0] < (1 << 20) && v[1] < (1 << 20) && v[2] < (1 << 20),
                    "cell %lld has a coordinate outside [0, 2^20)", (long long)c);
        owner[key(v[0], v[1], v[2])] = (int32_t)c;
        members[(size_t)c].push_back((int32_t)c);
        size[(size_t)c] = cell_size[c];
    }
    int sweeps = 0;
    bool again = true;
    while (again && sweeps < 10) {                           // max_recursive_merges = 10
        again = false;
        ++sweeps;
        for (int64_t i = 0; i < C; ++i) {
            auto& mine = members[(size_t)i];
            if (mine.empty() || size[(size_t)i] >= min_patch) continue;
            // find_dij: the LAST (highest index) other live cell with a voxel in the 26-neighbourhood of one of ours
            int32_t target = -1;
            for (int32_t c : mine) {
                const int32_t* v = cell_ijk + (int64_t)c * 3;
                for (int di = -1; di <= 1; ++di)
                    for (int dj = -1; dj <= 1; ++dj)
                        for (int dk = -1; dk <= 1; ++dk) {
                            auto it = owner.find(key(v[0] + di, v[1] + dj, v[2] + dk));
                            if (it != owner.end() && it->second != (int32_t)i && it->second > target) target = it->second;
                        }
            }
            if (target < 0) continue;
            auto& theirs = members[(size_t)target];
            for (int32_t c : mine) {
                const int32_t* v = cell_ijk + (int64_t)c * 3;
                owner[key(v[0], v[1], v[2])] = target;
                theirs.push_back(c);
            }
            size[(size_t)target] += size[(size_t)i];
            size[(size_t)i] = 0;
            mine.clear();
            if (size[(size_t)target] < min_patch) again = true;
        }
    }
    int64_t n = 0, pos = 0;
    seq_off[0] = 0;
    for (int64_t i = 0; i < C; ++i) {
        if (members[(size_t)i].empty() || size[(size_t)i] < min_patch) continue;
        for (int32_t c : members[(size_t)i]) seq[pos++] = c;
        seq_off[++n] = pos;
    }
    *n_out = n;
    if (sweeps_out) *sweeps_out = sweeps;
    return DNP_OK;
}

// the merge allocates (hash map, per-cell member lists): nothing may leave the boundary as an exception
int dnp_merge_cells(const int32_t* cell_ijk, const int64_t* cell_size, int64_t C, int64_t min_patch,
                    int64_t* seq, int64_t* seq_off, int64_t* n_out, int32_t* sweeps_out) {
    return guarded("dnp_merge_cells", [&]() { return merge_cells_body(cell_ijk, cell_size, C, min_patch, seq, seq_off, n_out, sweeps_out); });
}

}  // extern "C"
