// dnp_field.hip - K1 dnp_field_grad_* and K2 dnp_potential_* entry points.
//
// Host side: derive the reference's recursion leaves (field_utils.py:73-94: the source range is
// halved at int(n/2) until <= max_pts), cut every leaf into enough chunks to fill 256 CUs, launch
// pair_kernel over (target tiles x chunks) and then reduce_kernel, which sums the chunks of
// each leaf in fp64, zeroes non-finite leaf components (field_utils.py:110-115 / :53-54) and
// adds the leaves.
#include <math.h>

#include <algorithm>
#include <utility>
#include <vector>

#include "dnp_common.h"
#include "pair_kernel.h"

namespace dnp {

#ifndef DNP_MINCHUNK_CAP
#define DNP_MINCHUNK_CAP 512
#endif
// shortest chunk a leaf is cut into.  Scalar kernel: 64 sources, growing to 512 for very large source sets (S / 512)
// so that the by-value chunk table (512 entries) suffices for one round per ~262 000 sources.  LDS kernel: S / 64
// clamped to [64, 512] - it stages 256-row tiles, and a chunk of less than two tiles has no load/compute overlap.
#ifndef DNP_MINCHUNK_DIV
#define DNP_MINCHUNK_DIV 512
#endif
#ifndef DNP_LDS_MINCHUNK_DIV
#define DNP_LDS_MINCHUNK_DIV 64
#endif
static inline int64_t min_chunk(int64_t S, bool scalar_kernel) {
    const int64_t m = S / (scalar_kernel ? DNP_MINCHUNK_DIV : DNP_LDS_MINCHUNK_DIV);
    return m < 64 ? 64 : (m > DNP_MINCHUNK_CAP ? DNP_MINCHUNK_CAP : m);
}
constexpr int kKTScalar = 2, kKTLds = 4;
#ifndef DNP_KT_TILES      // A/B builds (tools/gpu_k1_ab.py K1_SHAPES)
#define DNP_KT_TILES 8
#endif
// ... used once that still leaves >= 8 target tiles of the scalar kernel (T >= 4096) / 16 of the LDS kernel (T >= 16 384).
// Round 5: 16 -> 8 for the scalar kernel - the representatives driver's final field (93 411 sources x 6589 targets) 393 -> 375 us,
// 100 000 x 4096 and 30 000 x 7000 level; 4 costs the latter 20 % (profiles/r05_kt_tiles_ab.txt)
constexpr int64_t kTilesForLarge = DNP_KT_TILES;
#ifndef DNP_LDS_KT_TILES
#define DNP_LDS_KT_TILES 16
#endif
constexpr int64_t kTilesForLargeLds = DNP_LDS_KT_TILES;
// workgroups an LDS-kernel launch aims at (small problems).  8192 = round 1's rule
#ifndef DNP_LDS_WANT_BLOCKS
#define DNP_LDS_WANT_BLOCKS 8192
#endif
#ifndef DNP_LDS_TAPER      // tapered chunk lengths in LDS-kernel launches too (0: equal chunks)
#define DNP_LDS_TAPER 1
#endif
constexpr size_t kSlabCap = (size_t)1 << 30;  // bytes of partial slab per round
struct Plan {
    // one entry per round; every round is a run of whole leaves
    struct Round {
        std::vector<int32_t> chunk_off;  // n_chunks+1 source offsets
        std::vector<int32_t> leaf_first; // n_leaves+1 indices into chunk_off
    };
    std::vector<Round> rounds;
    int64_t max_chunks = 0;  // largest n_chunks over the rounds
    int kt = 1;              // targets per lane (1 for small target sets: twice the workgroups)
};

static void split_leaves(int64_t lo, int64_t hi, int64_t max_pts, std::vector<int64_t>& cuts) {
    // field_utils.py:79-82: mid = int(S/2); halves are summed.  Leaves are emitted left to right.
    if (max_pts > 0 && hi - lo > max_pts) {
        const int64_t mid = lo + (hi - lo) / 2;
        split_leaves(lo, mid, max_pts, cuts);
        split_leaves(mid, hi, max_pts, cuts);
    } else {
        cuts.push_back(hi);
    }
}

#ifndef DNP_K1_FAR        // far-field chain in the generic entry points (pays only for spatially sorted clouds)
#define DNP_K1_FAR 1
#endif

// Number of source chunks for one launch, from the sweep in profiles/r02_k1_planning.txt (tools/gpu_k1_ab.py, MI355X):
// short chunks win - ~256 sources per chunk, and at least ~4096 workgroups when the target set is small - as long
// as the fp64 partial slab (24 B per target and chunk, written once and read once) stays around 300 MB:
//   fandisk 11 031^2: 93-186 chunks 90 us (32 chunks 106 us);   30 000^2: 186 chunks 476 us (46 chunks 515 us);
//   100 000^2: 128 chunks 4.82 ms (42 chunks 5.01 ms, the 1 GB slab of 390 chunks would cost 0.4 ms).
// make_plan clips the answer by the shortest chunk it allows and by the by-value chunk table.
constexpr int64_t kChunkSources = 256, kWantBlocks = 4096, kShortestUseful = 128;
constexpr size_t kSlabTarget = (size_t)320 << 20;
// Source split (round 3, far-field launches of the scalar kernel): the four wavefronts of a workgroup share one target
// tile and take a quarter of the chunk each, with their own far-field decision - chunks of ~kSplitChunkSources sources
// then have the item length and the far-test granularity of chunks a quarter as long, and a partial slab a quarter the size.
#ifndef DNP_K1_SPLIT
#define DNP_K1_SPLIT 4
#endif
#ifndef DNP_K1_SPLIT_CHUNK
#define DNP_K1_SPLIT_CHUNK 1536
#endif
constexpr int64_t kSplitChunkSources = DNP_K1_SPLIT_CHUNK;
#ifndef DNP_K1_SPLIT_CHUNK_SHORT
#define DNP_K1_SPLIT_CHUNK_SHORT 768
#endif
#ifndef DNP_K1_SPLIT_FILL
#define DNP_K1_SPLIT_FILL 8192
#endif
constexpr int64_t kSplitChunkShort = DNP_K1_SPLIT_CHUNK_SHORT, kSplitFillBlocks = DNP_K1_SPLIT_FILL;
static int64_t choose_chunks(int64_t S, int64_t T, int64_t t_tiles, int64_t n_leaves, int nc, bool scalar_kernel, int split = 1) {
    if (scalar_kernel && split > 1) {
        int64_t want = ceil_div(S, kSplitChunkSources);
        // few target tiles (the representatives driver's final field: 93 411 sources x 6589 targets = 52 tiles): chunks down to
        // kSplitChunkShort sources while the launch stays below kSplitFillBlocks workgroups (profiles/r05_rest_field_ab.txt)
        if (t_tiles * want < kSplitFillBlocks) {
            int64_t fill = kSplitFillBlocks / t_tiles;
            if (fill > ceil_div(S, kSplitChunkShort)) fill = ceil_div(S, kSplitChunkShort);
            if (want < fill) want = fill;
        }
        const int64_t slab = (int64_t)(kSlabTarget / ((size_t)(T > 0 ? T : 1) * nc * sizeof(double)));
        if (want > slab) want = slab;
        return want < n_leaves ? n_leaves : want;
    }
    if (!scalar_kernel) {        // LDS kernel (small problems, gathered sources): ~8192 workgroups, the round-1 rule
        const int64_t want = DNP_LDS_WANT_BLOCKS >= 8192 ? ceil_div((int64_t)DNP_LDS_WANT_BLOCKS, t_tiles)
                                                         : (DNP_LDS_WANT_BLOCKS / t_tiles > 0 ? DNP_LDS_WANT_BLOCKS / t_tiles : 1);
        return want < n_leaves ? n_leaves : want;
    }
    int64_t want = S / kChunkSources;
    int64_t fill = ceil_div(kWantBlocks, t_tiles);          // small target sets: more workgroups, but not chunks so
    if (fill > S / kShortestUseful) fill = S / kShortestUseful;   // short that the workgroup prologue dominates
    if (want < fill) want = fill;
    const int64_t slab = (int64_t)(kSlabTarget / ((size_t)(T > 0 ? T : 1) * nc * sizeof(double)));
    if (want > slab) want = slab;
    if (want < n_leaves) want = n_leaves;
    return want;
}

static Plan make_plan(int64_t S, int64_t T, int64_t max_pts, int nc, size_t elem, bool scalar_kernel, int split = 1) {
    Plan plan;
    const int kKTLarge = scalar_kernel ? kKTScalar : kKTLds;
    std::vector<int64_t> cuts;  // leaf end offsets
    if (S > 0) split_leaves(0, S, max_pts, cuts);
    // small problems are latency bound: prefer many short workgroups (1 target per lane, chunks down to 64
    // sources); large ones amortise the source fetch over 2 (scalar kernel) / 4 (LDS kernel) targets per lane
    plan.kt = (T >= (int64_t)kBlock * kKTLarge * (scalar_kernel ? kTilesForLarge : kTilesForLargeLds)) ? kKTLarge : 1;
#ifdef DNP_FORCE_KT       // planning experiments only
    plan.kt = DNP_FORCE_KT;
#endif
    const int64_t t_tiles = ceil_div(T > 0 ? T : 1, (int64_t)(kBlock / split) * plan.kt);
    const int64_t n_leaves = (int64_t)cuts.size();
    // chunk cap per round from the slab budget (at least one)
    int64_t cap = (int64_t)(kSlabCap / ((size_t)(T > 0 ? T : 1) * nc * elem));
    if (cap > kMaxChunks) cap = kMaxChunks;
    if (cap < 1) cap = 1;
    int64_t want = choose_chunks(S, T, t_tiles, n_leaves, nc, scalar_kernel, split);
#ifdef DNP_FORCE_CHUNKS   // planning experiments only
    want = DNP_FORCE_CHUNKS;
#endif

    if (want < n_leaves) want = n_leaves;
    // keep all leaves in ONE round whenever they fit: a round is two launches
    if (n_leaves <= cap && want > cap) want = cap;

    // hand the chunks to the leaves in proportion to their length (largest remainder), at least one each
    std::vector<int64_t> per_leaf((size_t)n_leaves, 1);
    {
        int64_t lo = 0, given = 0;
        std::vector<std::pair<double, int64_t>> frac;
        for (int64_t l = 0; l < n_leaves; ++l) {
            const double share = (double)want * (double)(cuts[l] - lo) / (double)S;
            int64_t m = (int64_t)share;
            if (m < 1) m = 1;
            per_leaf[(size_t)l] = m;
            given += m;
            frac.push_back({share - (double)(int64_t)share, l});
            lo = cuts[l];
        }
        std::sort(frac.begin(), frac.end(), [](const std::pair<double, int64_t>& x, const std::pair<double, int64_t>& y) {
            return x.first > y.first || (x.first == y.first && x.second < y.second);
        });
        for (size_t i = 0; given < want && i < frac.size(); ++i, ++given) per_leaf[(size_t)frac[i].second] += 1;
    }

    // TAPER (round 3): workgroups are dispatched in chunk order, and a launch ends with every SIMD running its last
    // wavefront alone, latency bound (the timeline of a fandisk launch, profiles/r03_timeline.txt: workgroups of equal
    // work live 22 to 84 us, the chip is below half full for the last 35 us of an 85 us launch).  So the chunks of the
    // LAST leaf of a launch are not equal: the first 60 % of them keep full length, the rest shrink linearly to a
    // quarter - the last workgroups to start are the shortest, as in longest-processing-time-first scheduling.  Same
    // number of chunks (same partial slab), the same sums up to the order of the fp64 chunk additions.
#ifndef DNP_TAPER
#define DNP_TAPER 1
#endif
#ifndef DNP_TAPER_KNEE       // fraction of the chunks that keep full length / relative length of the last chunk
#define DNP_TAPER_KNEE 0.6
#endif
#ifndef DNP_TAPER_MIN
#define DNP_TAPER_MIN 0.25
#endif
    auto taper_weight = [scalar_kernel](int64_t i, int64_t m) -> double {      // relative length of chunk i of m
        const double knee = DNP_TAPER_KNEE * (double)m;
        if (!DNP_TAPER || (!scalar_kernel && !DNP_LDS_TAPER) || m < 8 || (double)i < knee) return 1.0;
        return 1.0 - (1.0 - DNP_TAPER_MIN) * ((double)i - knee) / ((double)m - knee);
    };
    Plan::Round cur;
    cur.chunk_off.push_back(0);
    cur.leaf_first.push_back(0);
    int64_t lo = 0;
    for (int64_t l = 0; l < n_leaves; ++l) {
        const int64_t hi = cuts[l];
        const int64_t len = hi - lo;
        int64_t m = per_leaf[(size_t)l];
        const int64_t m_max = len / min_chunk(S, scalar_kernel) > 0 ? len / min_chunk(S, scalar_kernel) : 1;
        if (m > m_max) m = m_max;
        if (m > cap) m = cap;
        if (m < 1) m = 1;
        if ((int64_t)cur.chunk_off.size() - 1 + m > cap && cur.chunk_off.size() > 1) {
            plan.rounds.push_back(cur);                 // start a new round with this leaf
            cur = Plan::Round();
            cur.chunk_off.push_back((int32_t)lo);
            cur.leaf_first.push_back(0);
        }
        if (l + 1 == n_leaves) {                     // the launch's tail: tapered chunk lengths
            double total = 0.0, run = 0.0;
            for (int64_t i = 0; i < m; ++i) total += taper_weight(i, m);
            for (int64_t i = 1; i <= m; ++i) {
                run += taper_weight(i - 1, m);
                int64_t cut = (i == m) ? hi : lo + (int64_t)((double)len * run / total);
                if (split > 1 && i != m) cut = lo + (cut - lo) / (kFlush * split) * (kFlush * split);   // equal parts per wavefront
                cur.chunk_off.push_back((int32_t)(cut > cur.chunk_off.back() ? cut : cur.chunk_off.back()));
            }
        } else {
            for (int64_t i = 1; i <= m; ++i) {
                int64_t cut = lo + len * i / m;
                if (split > 1 && i != m) cut = lo + (cut - lo) / (kFlush * split) * (kFlush * split);
                cur.chunk_off.push_back((int32_t)cut);
            }
        }
        cur.leaf_first.push_back((int32_t)cur.chunk_off.size() - 1);
        lo = hi;
    }
    if (cur.chunk_off.size() > 1) plan.rounds.push_back(cur);
    for (auto& r : plan.rounds)
        if ((int64_t)r.chunk_off.size() - 1 > plan.max_chunks) plan.max_chunks = (int64_t)r.chunk_off.size() - 1;
    return plan;
}

static size_t plan_workspace(const Plan& p, int64_t T, int nc, size_t elem) {
    size_t b = (size_t)p.max_chunks * (size_t)(T > 0 ? T : 0) * nc * elem;
    return (b + 255) & ~(size_t)255;
}

// ---- reduce: out[t][c] (+)= sum_leaves filter( sum_{chunks in leaf} partial[chunk][t][c] ) ----
template <typename F>
struct ReduceArgs {
    const double* partial;  // [n_chunks][T][NC] chunk sums, kept in fp64
    int64_t T;
    const int64_t* tgt_idx; // for out_scatter
    F* out;
    int64_t ld_out;
    int out_scatter;
    int accumulate;
    int* nonfinite;         // [2] counters of inf / nan leaf components zeroed, or nullptr
    int n_leaves;
    int32_t leaf_first[kMaxChunks + 1];
};

template <typename F, int NC>
__global__ __launch_bounds__(256) void reduce_kernel(const ReduceArgs<F> a) {
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;  // element (t, c)
    if (i >= a.T * NC) return;
    const int64_t t = i / NC;
    const int c = (int)(i - t * NC);
    const int64_t stride = a.T * NC;
    double total = 0.0;
    for (int l = 0; l < a.n_leaves; ++l) {
        double s = 0.0;
        const int c1 = a.leaf_first[l + 1];
        // 8 independent loads in flight, added in chunk order (the sum does not depend on the batching)
        for (int ch = a.leaf_first[l]; ch < c1; ch += 8) {
            double v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = (ch + u < c1) ? a.partial[(int64_t)(ch + u) * stride + i] : 0.0;
#pragma unroll
            for (int u = 0; u < 8; ++u) s += v[u];
        }
        const F f = (F)s;
        // E_total[E_total.isinf()] = 0; E_total[E_total.isnan()] = 0   (per leaf, per component)
        if (__builtin_isfinite(f)) total += (double)f;
        else if (a.nonfinite) atomicAdd(a.nonfinite + (f != f ? 1 : 0), 1);
    }
    const int64_t row = a.out_scatter ? a.tgt_idx[t] : t;
    F* o = a.out + row * a.ld_out + c;
    const F v = (F)total;
    *o = a.accumulate ? (F)(*o + v) : v;
}

// ---- reduce with the tail of field_utils.reference_field (field_utils.py:188-201) fused in: one thread per target row.
//   kTailNormalise  out[t] = (x, y, z, E / |E|)   (rows with |E| == 0 keep E)                    :191-194, 3-column targets
//   kTailSign       n_t *= (E . n_t >= 0 ? +1 : -1), in place on the target rows' columns 3..5   :195-199, 6-column targets
// The field itself is summed exactly as reduce_kernel does (per leaf: chunk sums in fp64 in chunk order, rounded to F,
// non-finite components zeroed and counted; leaves added in fp64).  The per-point products are rounded separately and
// added left to right, as torch's (E * n).sum(dim=-1) does - no fma contraction: a sign decision hangs on it.
enum RefTail { kTailNormalise = 1, kTailSign = 2 };

template <typename F, int TAIL>
__global__ __launch_bounds__(256) void reduce_rows_kernel(const ReduceArgs<F> a, F* __restrict__ tgt, int64_t ld_tgt) {
#pragma clang fp contract(off)
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= a.T) return;
    const int64_t stride = a.T * 3;
    F e[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        double total = 0.0;
        for (int l = 0; l < a.n_leaves; ++l) {
            double s = 0.0;
            const int c1 = a.leaf_first[l + 1];
            for (int ch = a.leaf_first[l]; ch < c1; ch += 8) {
                double v[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) v[u] = (ch + u < c1) ? a.partial[(int64_t)(ch + u) * stride + t * 3 + c] : 0.0;
#pragma unroll
                for (int u = 0; u < 8; ++u) s += v[u];
            }
            const F f = (F)s;
            if (__builtin_isfinite(f)) total += (double)f;
            else if (a.nonfinite) atomicAdd(a.nonfinite + (f != f ? 1 : 0), 1);
        }
        e[c] = (F)total;
    }
    F* row = tgt + t * ld_tgt;
    if (TAIL == kTailNormalise) {
        F* o = a.out + t * a.ld_out;
        const F len = __builtin_sqrt((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);      // IEEE sqrt / division
        const F den = (len != F(0)) ? len : F(1);
        o[0] = row[0]; o[1] = row[1]; o[2] = row[2];
        o[3] = e[0] / den; o[4] = e[1] / den; o[5] = e[2] / den;
    } else {
        const F d = (e[0] * row[3] + e[1] * row[4]) + e[2] * row[5];
        const F sg = (d >= F(0)) ? F(1) : F(-1);                                      // `>=`: reference_field, not the drivers' `>`
        row[3] = row[3] * sg; row[4] = row[4] * sg; row[5] = row[5] * sg;
    }
}

template <typename F, int MODE>
static int run_pairs_body(const F* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                     const F* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                     F eps, int64_t max_pts, F* out, int64_t ld_out, int out_scatter, int accumulate,
                     int* nonfinite, int* nonfinite_host, void* workspace, size_t workspace_bytes, hipStream_t stream,
                     int tail = 0, F* tgt_rw = nullptr) {
    constexpr int NC = (MODE == kField) ? 3 : 1;
    clear_error();
    DNP_REQUIRE(S >= 0 && T >= 0, "negative size S=%lld T=%lld", (long long)S, (long long)T);
    DNP_REQUIRE(S <= INT32_MAX, "S=%lld exceeds the 2^31-1 source rows one call supports", (long long)S);
    if (T == 0) {
        if (nonfinite && nonfinite_host)
            DNP_CHECK_HIP(hipMemcpyAsync(nonfinite_host, nonfinite, 3 * sizeof(int), hipMemcpyDeviceToHost, stream));
        return DNP_OK;
    }
    DNP_REQUIRE(tgt && (out || tail == kTailSign), "NULL tgt/out pointer");
    DNP_REQUIRE(S == 0 || src, "NULL src pointer");
    DNP_REQUIRE(ld_src >= 6 || S == 0, "ld_src=%lld < 6", (long long)ld_src);
    DNP_REQUIRE(ld_tgt >= 3, "ld_tgt=%lld < 3", (long long)ld_tgt);
    DNP_REQUIRE(ld_out >= (MODE == kField ? 3 : 1), "ld_out=%lld too small", (long long)ld_out);
    DNP_REQUIRE(!out_scatter || tgt_idx, "out_scatter requires tgt_idx");

    if (S == 0 && tail != 0) {  // empty sum: E = 0 -> the tails see a zero field (normals (0,0,0) / sign +1)
        ReduceArgs<F> ra{};
        ra.partial = nullptr; ra.T = T; ra.tgt_idx = nullptr; ra.out = out; ra.ld_out = ld_out; ra.n_leaves = 0;
        if (tail == kTailNormalise)
            hipLaunchKernelGGL((reduce_rows_kernel<F, kTailNormalise>), dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, stream, ra, tgt_rw, ld_tgt);
        else
            hipLaunchKernelGGL((reduce_rows_kernel<F, kTailSign>), dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, stream, ra, tgt_rw, ld_tgt);
        DNP_CHECK_HIP(hipGetLastError());
        return DNP_OK;
    }
    if (S == 0) {  // empty sum: zeros (the reference's sum over an empty dim)
        if (nonfinite && nonfinite_host)
            DNP_CHECK_HIP(hipMemcpyAsync(nonfinite_host, nonfinite, 3 * sizeof(int), hipMemcpyDeviceToHost, stream));
        if (!accumulate) {
            ReduceArgs<F> ra{};
            ra.partial = nullptr; ra.T = T; ra.tgt_idx = tgt_idx; ra.out = out; ra.ld_out = ld_out;
            ra.out_scatter = out_scatter; ra.accumulate = 0; ra.nonfinite = nullptr; ra.n_leaves = 0;
            const int64_t n = T * NC;
            hipLaunchKernelGGL((reduce_kernel<F, NC>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, ra);
            DNP_CHECK_HIP(hipGetLastError());
        }
        return DNP_OK;
    }

    // contiguous source rows go through the scalar unit, gathered rows through LDS (pair_kernel.h)
#ifndef DNP_K1_FORCE_LDS   // A/B builds only
#define DNP_K1_FORCE_LDS 0
#endif
    // Small problems are latency bound and the LDS kernel's cooperative staging wins there (fandisk 11 031^2: 84 us
    // against 95 us, 3000^2: 20 against 29 us); from ~5 10^8 pairs on the scalar kernel is level or ahead
    // (30 000^2: 473 against 496 us) and is the one that can take the far-field chain on sorted clouds
    // (profiles/r02_k1_planning.txt).
#ifndef DNP_K1_SCALAR_FROM   // pairs from which the scalar-unit kernel takes over / may use the far chain (A/B builds move them)
#define DNP_K1_SCALAR_FROM 5e8
#endif
#ifndef DNP_K1_FAR_FROM
#define DNP_K1_FAR_FROM 1e9
#endif
#ifndef DNP_K1_FAR_FROM_WIDE  // ... and from here already when there are at least twice as many sources as targets (round 5: the
#define DNP_K1_FAR_FROM_WIDE 5e8   // representatives driver's final field 93 411 x 6589: 397 -> 371 us, 100 000 x 8000: 496 -> 458 us,
#endif                        // 50 000 x 12 000: 365 -> 350 us; 30 000^2 on an unsorted cloud would lose 1.5 % - profiles/r05_rest_field_ab.txt)
    const bool scalar_kernel = (src_idx == nullptr) && !DNP_K1_FORCE_LDS && (double)S * (double)T >= DNP_K1_SCALAR_FROM;
    // far-field launches (fp32, >= 10^9 pairs, eps in range) of the scalar kernel split their work items by source
    const double n_pairs = (double)S * (double)T;
    const bool far_size = n_pairs >= DNP_K1_FAR_FROM || (n_pairs >= DNP_K1_FAR_FROM_WIDE && S >= 2 * T);
    const bool far_candidate = sizeof(F) == 4 && DNP_K1_FAR && scalar_kernel && MODE == kField && eps > F(0) &&
                               far_size && far_threshold_d2((double)eps) > 0.0;
    const int split = (far_candidate && T >= (int64_t)kBlock * kKTScalar * kTilesForLarge) ? DNP_K1_SPLIT : 1;
    const Plan plan = make_plan(S, T, max_pts, NC, sizeof(double), scalar_kernel, split);
    const size_t need = plan_workspace(plan, T, NC, sizeof(double));
    if (!workspace || workspace_bytes < need) {
        set_error("workspace of %zu bytes required, %zu given", need, workspace ? workspace_bytes : (size_t)0);
        return DNP_EWORKSPACE;
    }
    const int64_t t_tiles = ceil_div(T, (int64_t)(kBlock / split) * plan.kt);
    DNP_REQUIRE(t_tiles <= INT32_MAX, "T too large");
    if (tail != 0 && plan.rounds.size() != 1) {
        set_error("reference_field tail: %zu rounds of chunks (S=%lld, T=%lld) - call dnp_field_grad and finish on the caller's side",
                  plan.rounds.size(), (long long)S, (long long)T);
        return DNP_EINVAL;
    }

    bool first = true;
    for (const auto& r : plan.rounds) {
        const int n_chunks = (int)r.chunk_off.size() - 1;
        PairArgs<F, double> pa{};
        pa.src = src; pa.ld_src = ld_src; pa.src_idx = src_idx;
        pa.tgt = tgt; pa.ld_tgt = ld_tgt; pa.tgt_idx = tgt_idx; pa.T = T;
        pa.chunk_off_dev = nullptr; pa.chunk_base = 0; pa.tgt_group = nullptr;
        pa.eps = eps; pa.partial = (double*)workspace;
        // one chunk that is also the only leaf of the only round: the pair kernel writes the final rows
        const bool direct = plan.rounds.size() == 1 && n_chunks == 1 && tail == 0;
        pa.out = direct ? out : nullptr; pa.ld_out = ld_out; pa.out_scatter = out_scatter; pa.accumulate = accumulate;
        pa.nonfinite = nonfinite;
        // the far-field chain only pays for spatially sorted clouds; its per-workgroup set-up (box scan + barrier) is
        // noise for big problems and a measurable 5-10 % for small ones (fandisk): off below 10^9 pairs.  (Round 3
        // measured the boxes from two pre-kernels instead - a chunk table and a target-tile table as in patch mode:
        // 4452.6 against 4448.3 us at 100 000^2 on the patch-sorted cloud, no gain: with 780-source chunks the scan is
        // already amortised and the two extra launches cost what the tables save; profiles/r03_k1_tables_ab.txt.)
        pa.far_d2 = (sizeof(F) == 8 ? n_pairs >= DNP_K1_FAR_FROM : far_size) ? (F)far_threshold_d2((double)eps, sizeof(F) == 8 ? kFarRatio64 : kFarRatio) : F(0);
        for (int i = 0; i <= n_chunks; ++i) pa.chunk_off[i] = r.chunk_off[i];
#ifdef DNP_BOUNDS
        pa.bnd = PairBounds{};
        pa.bnd.n_partial = (int64_t)(workspace_bytes / sizeof(double)); pa.bnd.n_src_rows = src_idx ? INT64_MAX : S;
        pa.bnd.err = bounds_err_buffer();
#endif
        const dim3 grid((unsigned)t_tiles, (unsigned)n_chunks);
        // eps > 0 (what every caller of the reference passes): the short chain; otherwise the explicit one
        const int variant = (MODE != kField) ? kFast : (eps > F(0) ? kFast : (eps == F(0) ? kNanCoinc : kRobust));
#define DNP_LAUNCH_PAIR(KT, V) \
    hipLaunchKernelGGL((pair_kernel<F, double, MODE, KT, V>), grid, dim3(kBlock), 0, stream, pa)
#define DNP_LAUNCH_SCALAR(KT, V)                                                                                  \
    do {                                                                                                          \
        if (sizeof(F) == 4 && DNP_K1_FAR && pa.far_d2 > F(0) && split == 4 && KT == kKTScalar && V == kFast &&   \
            MODE == kField)                                                                                       \
            hipLaunchKernelGGL((pair_kernel_scalar<F, double, kField, kKTScalar, kFast, (sizeof(F) == 4 && DNP_K1_FAR), \
                                                   false, false, 0, (sizeof(F) == 4 && DNP_K1_FAR) ? 4 : 1>),  \
                               grid, dim3(kBlock), 0, stream, pa);                                                \
        else if (sizeof(F) == 4 && DNP_K1_FAR && pa.far_d2 > F(0))                                                \
            hipLaunchKernelGGL((pair_kernel_scalar<F, double, MODE, KT, V, (sizeof(F) == 4 && DNP_K1_FAR)>), grid, \
                               dim3(kBlock), 0, stream, pa);                                                      \
        else if (sizeof(F) == 8 && DNP_K1_FAR && pa.far_d2 > F(0) && KT == kKTScalar && V == kFast && MODE == kField) \
            /* fp64 (round 5): the one-tier far chain, chunk boxes scanned by the workgroup, no source split */     \
            hipLaunchKernelGGL((pair_kernel_scalar<F, double, kField, kKTScalar, kFast, (sizeof(F) == 8 && DNP_K1_FAR)>), grid, \
                               dim3(kBlock), 0, stream, pa);                                                      \
        else                                                                                                      \
            hipLaunchKernelGGL((pair_kernel_scalar<F, double, MODE, KT, V, false>), grid, dim3(kBlock), 0, stream, \
                               pa);                                                                               \
    } while (0)
        if (scalar_kernel) {
            if (plan.kt == kKTScalar) {
                if (variant == kFast) DNP_LAUNCH_SCALAR(kKTScalar, kFast);
                else if (variant == kNanCoinc) DNP_LAUNCH_SCALAR(kKTScalar, kNanCoinc);
                else DNP_LAUNCH_SCALAR(kKTScalar, kRobust);
            } else {
                if (variant == kFast) DNP_LAUNCH_SCALAR(1, kFast);
                else if (variant == kNanCoinc) DNP_LAUNCH_SCALAR(1, kNanCoinc);
                else DNP_LAUNCH_SCALAR(1, kRobust);
            }
        } else if (plan.kt == kKTLds) {
            if (variant == kFast) DNP_LAUNCH_PAIR(kKTLds, kFast);
            else if (variant == kNanCoinc) DNP_LAUNCH_PAIR(kKTLds, kNanCoinc);
            else DNP_LAUNCH_PAIR(kKTLds, kRobust);
        } else {
            if (variant == kFast) DNP_LAUNCH_PAIR(1, kFast);
            else if (variant == kNanCoinc) DNP_LAUNCH_PAIR(1, kNanCoinc);
            else DNP_LAUNCH_PAIR(1, kRobust);
        }
#undef DNP_LAUNCH_SCALAR
#undef DNP_LAUNCH_PAIR
        DNP_CHECK_HIP(hipGetLastError());
        if (direct) break;
#ifdef DNP_SKIP_REDUCE     // experiment builds only (wrong results): the call WITHOUT its second pass - the upper bound of what
        continue;          // an in-kernel reduction by the last-arriving workgroup could save (profiles/r04_inkernel_reduce_bound.txt)
#endif

        ReduceArgs<F> ra{};
        ra.partial = (const double*)workspace; ra.T = T; ra.tgt_idx = tgt_idx; ra.out = out; ra.ld_out = ld_out;
        ra.out_scatter = out_scatter; ra.accumulate = (accumulate || !first) ? 1 : 0; ra.nonfinite = nonfinite;
        ra.n_leaves = (int)r.leaf_first.size() - 1;
        for (int i = 0; i <= ra.n_leaves; ++i) ra.leaf_first[i] = r.leaf_first[i];
        const int64_t n = T * NC;
        if (tail != 0 && MODE == kField) {
            // reference_field: a plan of one round (checked above), field summed and consumed in one pass
            if (tail == kTailNormalise)
                hipLaunchKernelGGL((reduce_rows_kernel<F, kTailNormalise>), dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, stream, ra, tgt_rw, ld_tgt);
            else
                hipLaunchKernelGGL((reduce_rows_kernel<F, kTailSign>), dim3((unsigned)ceil_div(T, 256)), dim3(256), 0, stream, ra, tgt_rw, ld_tgt);
        } else {
            hipLaunchKernelGGL((reduce_kernel<F, NC>), dim3((unsigned)ceil_div(n, 256)), dim3(256), 0, stream, ra);
        }
        DNP_CHECK_HIP(hipGetLastError());
        first = false;
    }
    // the counters (and the caller's "landed" stamp behind them) travel to pinned host memory behind the kernels
    if (nonfinite && nonfinite_host)
        DNP_CHECK_HIP(hipMemcpyAsync(nonfinite_host, nonfinite, 3 * sizeof(int), hipMemcpyDeviceToHost, stream));
    return DNP_OK;
}

// the planner allocates (std::vector): no exception may leave the entry points (dnp_common.h, guarded)
template <typename F, int MODE>
static int run_pairs(const F* src, int64_t S, int64_t ld_src, const int64_t* src_idx, const F* tgt, int64_t T, int64_t ld_tgt,
                     const int64_t* tgt_idx, F eps, int64_t max_pts, F* out, int64_t ld_out, int out_scatter, int accumulate,
                     int* nonfinite, int* nonfinite_host, void* workspace, size_t workspace_bytes, hipStream_t stream,
                     int tail = 0, F* tgt_rw = nullptr) {
    return guarded("field / potential launch", [&]() {
        return run_pairs_body<F, MODE>(src, S, ld_src, src_idx, tgt, T, ld_tgt, tgt_idx, eps, max_pts, out, ld_out, out_scatter,
                                       accumulate, nonfinite, nonfinite_host, workspace, workspace_bytes, stream, tail, tgt_rw);
    });
}

// bytes of partial slab a call of this shape can ask for: the maximum over every plan run_pairs may choose - the scalar
// kernel unsplit and source-split (the split plan's chunk rule is its own; round-3 advisor: nothing guaranteed that its
// chunk count stayed below the unsplit plan's) and the LDS kernel
static size_t workspace_for(int64_t S, int64_t T, int64_t max_pts, int nc) {
    size_t best = 256;
    for (int variant = 0; variant < 3; ++variant) {
        const bool scalar = variant != 2;
        const int split = variant == 1 ? DNP_K1_SPLIT : 1;
        const size_t b = plan_workspace(make_plan(S, T, max_pts, nc, sizeof(double), scalar, split), T, nc, sizeof(double));
        best = b > best ? b : best;
    }
    return best;
}

}  // namespace dnp

using namespace dnp;

extern "C" {

size_t dnp_field_grad_workspace_bytes(int64_t S, int64_t T, int64_t max_pts) {
    if (S <= 0 || T <= 0) return 256;
    // chunk sums are kept in fp64 for both precisions; 0 = the planner itself ran out of host memory (dnp_last_error)
    size_t bytes = 0;
    const int rc = guarded("dnp_field_grad_workspace_bytes", [&]() { bytes = workspace_for(S, T, max_pts, 3); return (int)DNP_OK; });
    return rc == DNP_OK ? bytes : 0;
}

size_t dnp_potential_workspace_bytes(int64_t S, int64_t T, int64_t max_pts) {
    if (S <= 0 || T <= 0) return 256;
    size_t bytes = 0;
    const int rc = guarded("dnp_potential_workspace_bytes", [&]() { bytes = workspace_for(S, T, max_pts, 1); return (int)DNP_OK; });
    return rc == DNP_OK ? bytes : 0;
}

int dnp_field_grad_f32(const float* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                       const float* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                       float eps, int64_t max_pts, float* out, int64_t ld_out, int out_scatter,
                       int accumulate, int32_t* nonfinite, int32_t* nonfinite_host, void* workspace,
                       size_t workspace_bytes, void* stream) {
    return run_pairs<float, kField>(src, S, ld_src, src_idx, tgt, T, ld_tgt, tgt_idx, eps, max_pts, out, ld_out,
                                    out_scatter, accumulate, nonfinite, nonfinite_host, workspace, workspace_bytes,
                                    (hipStream_t)stream);
}

int dnp_field_grad_f64(const double* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                       const double* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                       double eps, int64_t max_pts, double* out, int64_t ld_out, int out_scatter,
                       int accumulate, int32_t* nonfinite, int32_t* nonfinite_host, void* workspace,
                       size_t workspace_bytes, void* stream) {
    return run_pairs<double, kField>(src, S, ld_src, src_idx, tgt, T, ld_tgt, tgt_idx, eps, max_pts, out, ld_out,
                                     out_scatter, accumulate, nonfinite, nonfinite_host, workspace, workspace_bytes,
                                     (hipStream_t)stream);
}

int dnp_reference_field_f32(const float* src, int64_t S, int64_t ld_src, float* tgt, int64_t T, int64_t ld_tgt,
                            int form, float eps, int64_t max_pts, float* out, int64_t ld_out, int32_t* nonfinite,
                            void* workspace, size_t workspace_bytes, void* stream) {
    clear_error();
    DNP_REQUIRE(form == kTailNormalise || form == kTailSign, "form=%d (1 = 3-column targets -> out[T,6], 2 = 6-column targets, in place)", form);
    DNP_REQUIRE(form == kTailNormalise ? (out && ld_out >= 6) : ld_tgt >= 6, "form %d needs %s", form,
                form == kTailNormalise ? "out[T, >=6]" : "6-column target rows");
    return run_pairs<float, kField>(src, S, ld_src, nullptr, tgt, T, ld_tgt, nullptr, eps, max_pts, out, ld_out, 0, 0, nonfinite,
                                    nullptr, workspace, workspace_bytes, (hipStream_t)stream, form, tgt);
}

int dnp_reference_field_f64(const double* src, int64_t S, int64_t ld_src, double* tgt, int64_t T, int64_t ld_tgt,
                            int form, double eps, int64_t max_pts, double* out, int64_t ld_out, int32_t* nonfinite,
                            void* workspace, size_t workspace_bytes, void* stream) {
    clear_error();
    DNP_REQUIRE(form == kTailNormalise || form == kTailSign, "form=%d (1 = 3-column targets -> out[T,6], 2 = 6-column targets, in place)", form);
    DNP_REQUIRE(form == kTailNormalise ? (out && ld_out >= 6) : ld_tgt >= 6, "form %d needs %s", form,
                form == kTailNormalise ? "out[T, >=6]" : "6-column target rows");
    return run_pairs<double, kField>(src, S, ld_src, nullptr, tgt, T, ld_tgt, nullptr, eps, max_pts, out, ld_out, 0, 0, nonfinite,
                                     nullptr, workspace, workspace_bytes, (hipStream_t)stream, form, tgt);
}

int dnp_potential_f32(const float* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                      const float* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                      int64_t max_pts, float* out, int64_t ld_out,
                      void* workspace, size_t workspace_bytes, void* stream) {
    return run_pairs<float, kPotential>(src, S, ld_src, src_idx, tgt, T, ld_tgt, tgt_idx, 0.f, max_pts, out, ld_out,
                                        0, 0, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

int dnp_potential_f64(const double* src, int64_t S, int64_t ld_src, const int64_t* src_idx,
                      const double* tgt, int64_t T, int64_t ld_tgt, const int64_t* tgt_idx,
                      int64_t max_pts, double* out, int64_t ld_out,
                      void* workspace, size_t workspace_bytes, void* stream) {
    return run_pairs<double, kPotential>(src, S, ld_src, src_idx, tgt, T, ld_tgt, tgt_idx, 0.0, max_pts, out, ld_out,
                                         0, 0, nullptr, nullptr, workspace, workspace_bytes, (hipStream_t)stream);
}

#ifdef DNP_STAMP
int dnp_debug_set_stamps_field(void* p) { return dnp::set_stamps_here((unsigned long long*)p) == hipSuccess ? 0 : -4; }
#endif

}  // extern "C"
