// pair_kernel.h - the all-pairs dipole kernels for CDNA4 (gfx950), shared by the field_grad / potential entry
// points (dnp_field.hip) and the batched per-patch entry point (dnp_patch.hip).  Two kernels, same arithmetic:
//
// pair_kernel_scalar - sources through the SCALAR unit (contiguous source rows: every driver's hot case)
//   A source is the same for all 64 lanes of a wave, so it is fetched with s_load (wave-uniform address, scalar
//   cache -> SGPRs) and every VALU instruction takes its source operand from an SGPR: no staging, no barrier, no
//   ds_read, no VGPRs for the source row.  KT = 2 targets per lane in registers (57 VGPRs -> 8 waves per SIMD), a
//   wave owns 128 consecutive targets.  With FAR, a wave whose targets' bounding box is far from the chunk's box
//   runs the chunk through the one-transcendental far-field chain (pair_field_far); the drivers sort the cloud by
//   patch, so ~2/3 of the (wave, patch) combinations of the 100k / 256-patch workload qualify.  Measured on
//   MI355X (tools/gpu_ab_far.py, interleaved in one process): 4.37 ms per 10^10 pairs against 4.72 ms for the
//   LDS kernel below.  Why this wins although the instruction count is the same: the per-pair chain is a serial
//   dependency chain (hipcc schedules one pair after the other) and at the LDS kernel's 3 waves per SIMD the far
//   chain's 12-deep dependency could not be covered (the same far chain inside the LDS kernel was 8 % SLOWER than
//   the exact one); at 8 waves per SIMD it is 10 % faster.
//
// pair_kernel - sources staged through LDS (row gathers: src_idx != NULL, where a scalar load per gathered row
//   would be two dependent scalar round trips)
//   grid.x = target tiles (BLOCK threads x KT targets per lane; KT = 4 for large target sets, 1 for small
//   ones - swept on gfx950: 4 targets x 2 accumulator sets (134 VGPRs, 3 waves/SIMD) beats 2 x 4 (120 VGPRs,
//   4 waves) by 2 %, 6 x 2 and 4 x 4 lose to register pressure), grid.y = source chunks.
//   A block streams its chunk's sources through LDS in tiles of BLOCK rows (double-buffered:
//   the next tile's global loads are in flight while the current tile is consumed) and every
//   lane accumulates KT targets in registers.  All 64 lanes of a wave read the SAME LDS
//   address per source (hardware broadcast, conflict-free): one ds_read_b128 + one ds_read_b64
//   feed KT x 21 VALU instructions.
//
// Both: grid = (target tiles) x (source chunks); output partial[chunk][target][NC], reduced by a second,
//   tiny kernel - deterministic (no float atomics), and the natural place for the reference's
//   per-leaf Inf/NaN filter (field_utils.py:110-115) - or, when the plan is a single chunk, the final
//   rows directly (a.out).
//
// Arithmetic per pair (field mode): 19 full-rate + 2 quarter-rate (v_sqrt, v_rcp) VALU instructions
//   r = x_s - x_t; d2 = r.r; w = 1/(|r|^3 + eps); a = (p.r) w / |r|^2;  A += a*r;  B += w*p;  E = -(3A - B)
// which is field_utils.py:96-109 with r^ = r/|r| folded in (see pair_field for the exact chain and
// for how |r| == 0 contributes exactly 0).  Measured on gfx950 (tools/ubench_*.hip, profiles/r01_ubench_*):
// v_fma_f32 issues at 2.2 cycles per wave64, v_pk_fma_f32 at 4.1 (no gain from packing), v_rsq/v_rcp/v_sqrt
// at 8 back to back and ~13 when mixed with FMAs, so the exact chain costs ~70 cycles per 64 pairs per SIMD and
// the FP32 vector ALU is the roofline (DESIGN.md section 4 lists the alternatives that were measured).
//
// Accumulation: fp32 inside a run of kFlush = 32 sources (128 in the patch mode of the scalar kernel, see
// kFlushScalar), spread over kSets = 2 interleaved accumulator sets (chains of 16 adds), fp64 across runs (one
// cvt+fma per run, ~2 % of the issue slots; 64-source runs were 1.4 % faster and left the worst row of the
// random-dipole cloud G11 at 7.3e-6 of |E| instead of 4.5e-6 - margin towards the 1e-5 bound comes first) - the
// reference's own sum is a cascade sum (torch CPU); a plain fp32 chain over 10^5 terms would not
// stay within 1e-5 of it, and with 128-long chains the accumulation error still dominated the
// per-term rounding on cancellation-heavy rows (tools/gpu_accuracy.py).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

// experiment switches that produce WRONG or UNSAFE kernels may only be combined with the check build (round-4 advisor)
#if defined(DNP_BUG_A8D48F5) && !defined(DNP_BOUNDS)
#error "DNP_BUG_A8D48F5 re-enables the out-of-bounds tile_box read of round 3: only together with -DDNP_BOUNDS (tests/test_gpu_bounds.py)"
#endif
#if defined(DNP_SKIP_REDUCE) && !defined(DNP_EXPERIMENT)
#error "DNP_SKIP_REDUCE drops the second pass of field_grad (wrong results): timing experiments only, pass -DDNP_EXPERIMENT with it"
#endif

namespace dnp {

#ifdef DNP_STAMP   // timeline builds only (tools/gpu_timeline.py); the product library has none of this.  Every
                   // WAVEFRONT leaves four 64-bit words in a buffer set through dnp_debug_set_stamps_*: [0] its start time
                   // (LDS kernel only), [1] its end time (100 MHz wall clock), [2] HW_ID | XCC_ID << 32 - the SIMD it ran on.
                   // Everything goes through the scalar unit (s_memrealtime / s_getreg -> s_store, written back at once).
                   // pair_kernel_scalar has NO start stamp: ANY instruction with a side effect in its prologue - vector
                   // store, scalar store, a bare s_memrealtime kept in SGPRs - moves hipcc's allocation from 61 to 72-78
                   // VGPRs (two wavefronts per SIMD less: that timeline was not the product's).  With end stamps alone the
                   // allocation is the product's, and because a SIMD's slots are refilled at once while work is queued,
                   // "wavefronts of SIMD x that end after t" IS its occupancy for every t after the last dispatch.
static __device__ unsigned long long* g_stamps = nullptr;    // one per translation unit (no relocatable device code)
static inline hipError_t set_stamps_here(unsigned long long* p) {
    return hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &p, sizeof(p));
}
__device__ inline unsigned long long* stamp_slot() {         // wave-uniform by construction; readfirstlane makes hipcc believe it
    const unsigned long long q = (unsigned long long)(g_stamps + 4 * (((size_t)blockIdx.y * gridDim.x + blockIdx.x) * 4 + (threadIdx.x >> 6)));
    return (unsigned long long*)(((unsigned long long)(unsigned)__builtin_amdgcn_readfirstlane((int)(q >> 32)) << 32) |
                                 (unsigned)__builtin_amdgcn_readfirstlane((int)q));
}
#define DNP_STAMP_SLOT_() stamp_slot()
#define DNP_STAMP_BEGIN()                                                                                 \
    do {                                                                                                  \
        unsigned long long t_;                                                                            \
        unsigned long long* q_ = DNP_STAMP_SLOT_();                                                       \
        asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t_));                               \
        asm volatile("s_store_dwordx2 %0, %1, 0x0\n\ts_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(t_), "s"(q_)); \
    } while (0)
#define DNP_STAMP_END()                                                                                   \
    do {                                                                                                  \
        unsigned long long t_;                                                                            \
        unsigned h_, x_;                                                                                  \
        unsigned long long* q_ = DNP_STAMP_SLOT_();                                                       \
        asm volatile("s_memrealtime %0\n\ts_getreg_b32 %1, hwreg(HW_REG_HW_ID)\n\ts_getreg_b32 %2, hwreg(HW_REG_XCC_ID)\n\t" \
                     "s_waitcnt lgkmcnt(0)" : "=s"(t_), "=s"(h_), "=s"(x_));                              \
        asm volatile("s_store_dwordx2 %0, %3, 0x8\n\ts_store_dword %1, %3, 0x10\n\ts_store_dword %2, %3, 0x14\n\t" \
                     "s_waitcnt lgkmcnt(0)\n\ts_dcache_wb" ::"s"(t_), "s"(h_), "s"(x_), "s"(q_));        \
    } while (0)
#define DNP_STAMP_BEGIN_SCALAR() do { } while (0)
#else
#define DNP_STAMP_BEGIN() do { } while (0)
#define DNP_STAMP_BEGIN_SCALAR() do { } while (0)
#define DNP_STAMP_END() do { } while (0)
#endif

#ifndef DNP_BLOCK      // A/B builds only (64 / 128: finer dispatch granularity for the scalar kernel, whose wavefronts do not cooperate)
#define DNP_BLOCK 256
#endif
constexpr int kBlock = DNP_BLOCK;       // threads per workgroup (4 waves, one per SIMD)
#ifndef DNP_FLUSH
#define DNP_FLUSH 32
#endif
constexpr int kFlush = DNP_FLUSH;  // sources between two spills of the fp32 sums into the fp64 sums
#ifndef DNP_FLUSH_SCALAR
#define DNP_FLUSH_SCALAR 128
#endif
// the same for the scalar-unit kernel in PATCH mode (fp32 slabs): 128 (fp32 chains of 64 adds per set) measured 2.1 %
// faster than 64 on the bench workload (4.283 against 4.376 ms, profiles/r02_ab_scalar_forms.txt) at 1.9e-8 instead of
// 1.3e-8 median error of the summed slabs.  The generic entry points (fp64 partial slabs) keep the short kFlush: on
// random-dipole clouds, where rows are cancellation residues, 128 raised the worst row from 5.5e-6 (at 64) to 8.4e-6
// of |E| (N = 40 000) - inside 1e-5, but margin better kept.
constexpr int kFlushScalar = DNP_FLUSH_SCALAR;
#ifndef DNP_SETS
#define DNP_SETS 2
#endif
constexpr int kSets = DNP_SETS;   // interleaved fp32 accumulator sets (chain length kFlush / kSets); 2 or 4
#ifndef DNP_UNROLL
#define DNP_UNROLL DNP_SETS
#endif
constexpr int kUnroll = DNP_UNROLL;  // sources per inner-loop iteration (multiple of kSets; source u -> set u % kSets)
constexpr int kMaxChunks = 512;   // by-value chunk table entries per launch (2 KB of kernarg)
// bytes of one record of the exchange buffer (PairArgs::xch_ticket): a 128-byte line for the arrival counter + the run terms
constexpr int64_t xch_item_bytes(int ss, int kt, int nc) { return 128 + (int64_t)ss * kt * nc * 64 * (int64_t)sizeof(double); }
// (Round 5 built a second form - records of 8 run slots, split items of patches with 513..1024 points evaluated by TWO
// four-wavefront workgroups, the last of eight arrivers adding the slots in run order; bit-identical, fuzzed - and measured
// it: never ahead of leaving such a patch to one wavefront, +9 % on launches whose split patches all fit four runs (the
// second workgroups start, read their range and leave).  Not kept: profiles/r05_xch_eight_wavefronts.patch, r05_rank_share_partitions.txt.)

enum PairMode { kField = 0, kPotential = 1 };

// -DDNP_BOUNDS builds (tools/bin/libdnp_bounds.so: tests/test_gpu_bounds.py and tools/gpu_fuzz.py run against it): every read
// or write of a table whose index does not come from the lane's own bounds test - the wave-uniform tables (chunk offsets,
// chunk / target-tile boxes, the group of a tile's first row), the per-tile partials, the exchange records, the partial slab -
// is checked against the table's LENGTH, which the launchers pass in PairArgs::bnd; an index outside is COUNTED in a device
// error word (dnp_debug_bounds_errors) and clamped, so the check build reports where the product build would have read past
// a table - a fault only when the table happens to end on a page boundary (round 3's fuzz seed 7: the last workgroup's
// target-less wavefronts read the tile-box table up to 72 bytes past its end).  The product build has none of this.
struct PairBounds {
    int64_t n_chunk_off;     // entries of chunk_off_dev (patch mode: P + 1)
    int64_t n_chunk_box;     // rows of chunk_box
    int64_t n_tile_box;      // rows of tile_box
    int64_t n_tgt_group;     // entries of tgt_group
    int64_t n_w_part;        // doubles of w_part
    int64_t n_partial;       // elements of partial
    int64_t n_xch_items;     // records of the exchange buffer
    int64_t n_src_rows;      // rows of src
    int64_t n_chunk_perm;    // entries of chunk_perm
    unsigned int* err;       // [16] counters: [0..7] out-of-bounds accesses by table (kBnd*), [8] tiles that break WPART's precondition
};
enum { kBndChunkOff = 0, kBndChunkBox = 1, kBndTileBox = 2, kBndTgtGroup = 3, kBndWPart = 4, kBndPartial = 5, kBndXch = 6, kBndSrc = 7,
       kBndGroups = 8 };
#ifdef DNP_BOUNDS
__device__ inline int64_t bounds_checked(int64_t idx, int64_t len, int code, unsigned int* err) {
    if (idx >= 0 && idx < len) return idx;
    if (err) atomicAdd(err + code, 1u);
    return idx < 0 || len <= 0 ? 0 : len - 1;
}
#define DNP_BND(idx, field, code) bounds_checked((int64_t)(idx), a.bnd.field, code, a.bnd.err)
#else
#define DNP_BND(idx, field, code) (idx)
#endif

template <typename F, typename PT = F>
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
    PT* partial;             // [gridDim.y][T][NC]; PT = double keeps the chunk sums unrounded
    // direct epilogue (one chunk == one leaf): write the final rows here instead of a partial slab
    F* out;                  // nullptr = write `partial`
    int64_t ld_out;
    int out_scatter;         // row = tgt_idx[t]
    int accumulate;          // out += result
    int* nonfinite;          // [2] counters of inf / nan leaf components zeroed (direct epilogue), or nullptr
    const F* chunk_box;      // [chunks][6] bounding boxes (lo xyz, hi xyz) of the chunks' sources, or nullptr: found by the workgroup
    F far_d2;                // scalar kernel, FAR: squared box distance beyond which the one-transcendental chain runs
    const F* tile_box;       // scalar kernel, TBOX: [ceil(T / (64 KT))][6] boxes of the wavefronts' target tiles (dnp_tile_boxes_f32)
    double* w_part;          // scalar kernel, WPART: [gridDim.y][ceil(T / (64 KT))][WPART] per-(slab, tile) interaction partials
    int split_from;          // scalar kernel, XCH: chunks [split_from, n) of the launch are evaluated with the source split
    int n_chunks;            // scalar kernel, XCH: chunks of the launch (its grid is 1-D)
    // XCH: the exchange buffer, one record per (split chunk, target tile) item: [arrival counter, padded to a 128-byte line]
    // [SS][KT * NC][64] run terms (fp64).  A record's place depends on the item's index only, so launches of any size keep
    // their counters in the same words: zero before the first launch, left zero by every launch.
    unsigned int* xch_ticket;  // = the buffer; item i's counter is xch_ticket[i * xch_item_bytes / 4]
    double* xch_terms;         // = the buffer + 128 bytes; item i's terms start at xch_terms[i * xch_item_bytes / 8]
    // scalar kernel: launch row -> chunk of the launch (a permutation of 0..n-1), or nullptr = identity.  Workgroups are
    // dispatched in launch order, so the LAST rows decide how a launch ends: the drivers put the longest patches first
    // (longest-processing-time-first; round 5, profiles/r05_tail_sweep.txt).  Slabs, partials and boxes stay indexed by chunk.
    const int32_t* chunk_perm;
#ifdef DNP_BOUNDS
    PairBounds bnd;
#endif
    int32_t chunk_off[kMaxChunks + 1];  // by-value CSR offsets when chunk_off_dev == nullptr
};

// ---- per-pair arithmetic ----------------------------------------------------------------
template <typename F> struct Math;
template <> struct Math<float> {
    static __device__ __forceinline__ float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
    static __device__ __forceinline__ float rcp(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
    // the kFast chain's root and reciprocal (fp32: the hardware's own, 1 ulp; fp64: refined, see Math<double>)
    static __device__ __forceinline__ float sqrt_fast(float x) { return __builtin_amdgcn_sqrtf(x); }
    static __device__ __forceinline__ float rcp_fast(float x) { return __builtin_amdgcn_rcpf(x); }
    static __device__ __forceinline__ float rsq_fast(float x) { return __builtin_amdgcn_rsqf(x); }
    static constexpr float kTiny = 1.17549435e-38f;   // FLT_MIN
    static __device__ __forceinline__ float fma(float a, float b, float c) { return __builtin_fmaf(a, b, c); }
    static constexpr float kHuge = 1e30f;   // (1e30)^1.5 overflows fp32 -> w = rcp(inf) = 0
    static constexpr float kFar = 1e18f;    // padding source position: d2 = 3e36 (finite), d^3 = inf
};
template <> struct Math<double> {
    static __device__ __forceinline__ double rsq(double x) { return 1.0 / __builtin_sqrt(x); }
    static __device__ __forceinline__ double rcp(double x) { return 1.0 / x; }
    static __device__ __forceinline__ double sqrt(double x) { return __builtin_sqrt(x); }
    // Round 5: the fp64 kFast chain and the potential no longer go through the compiler's IEEE sqrt / division (a v_rsq_f64 or
    // v_rcp_f64 + ~12 fp64 instructions each: the 100 000^2 field took 14.5 ms, profiles/r05_f64_time.txt).  v_rsq_f64 /
    // v_rcp_f64 are 2^29-ulp approximations (relative error <= 2^-23, the ISA manual's figure); ONE third-order step takes that
    // to ~2^-69 - below half an ulp, so the result is the rounding of the exact value but for the last-place ties a fused
    // evaluation cannot see:
    //   rsq:  e = 1 - x y^2,  y' = y (1 + e/2 + 3 e^2/8)        5 fp64 instructions + the transcendental
    //   rcp:  e = 1 - x r,    r' = r (1 + e + e^2)              3 + 1
    // (tools/ubench_f64.hip: v_fma_f64 4.3 cycles per wave64 instruction at 8 wavefronts per SIMD, the fp64 transcendentals 16.2;
    // 28 fma + rsq + rcp per pair mixes at 5.0 cycles per instruction.)  Arguments must be finite and positive: 0, inf and NaN
    // come back as NaN (0 * inf in the step), which is why sqrt_fast clamps its argument and the padding rows sit at 1e50.
    static __device__ __forceinline__ double rsq_fast(double x) {
        const double y = __builtin_amdgcn_rsq(x);
        const double t = x * y;
        const double e = __builtin_fma(-t, y, 1.0);
        const double p = __builtin_fma(0.375, e, 0.5);
        return __builtin_fma(y * e, p, y);
    }
    static __device__ __forceinline__ double sqrt_fast(double x) {     // sqrt(0) = 0 exactly (the coincident pair of kFast)
        return x * rsq_fast(__builtin_fmax(x, 2.2250738585072014e-308));
    }
    static __device__ __forceinline__ double rcp_fast(double x) {
        const double r = __builtin_amdgcn_rcp(x);
        const double e = __builtin_fma(-x, r, 1.0);
        return __builtin_fma(r, __builtin_fma(e, e, e), r);
    }
    static constexpr double kTiny = 2.2250738585072014e-308;   // DBL_MIN
    static __device__ __forceinline__ double fma(double a, double b, double c) { return __builtin_fma(a, b, c); }
    static constexpr double kHuge = 1e250;  // (1e250)^1.5 overflows fp64
    static constexpr double kFar = 1e50;    // padding source position: every intermediate of the refined chains stays finite
                                            // (d2 = 3e100, d2 (|r|^3 + eps) = 1.5e251), the zero dipole makes the pair contribute 0
};

// Variants of the per-pair chain (template parameter V of pair_field / pair_kernel):
//   kFast    eps > 0 (every caller of the reference): 19 full-rate + 2 quarter-rate instructions
//              d = sqrt(d2); den = d2*d + eps; g = d2*den + FLT_MIN; q = rcp(g); w = d2*q; a = (p.r)*q
//            q = 1/(|r|^2 (|r|^3+eps)), so w = 1/(|r|^3+eps) and a = (p.r)/(|r|^2 (|r|^3+eps)).
//            Adding FLT_MIN changes no g >= 2^-102 (it is below half an ulp) and makes g > 0 when
//            |r| == 0: then q is finite, w = 0*q = 0 and a = 0*q = 0 - the pair contributes exactly 0,
//            as `E[zero_mask] = 0` followed by /(0+eps) does (field_utils.py:106-108), without a
//            compare/select.
//   kRobust  any eps (also negative): explicit |r| == 0 test, rsq/rcp chain, 22 + 2 instructions.
//   kNanCoinc  eps == 0: as kRobust, and a coincident pair yields NaN like the reference's 0/0
//            (the leaf filter then zeroes the row, field_utils.py:112-115).
enum PairVariant { kFast = 0, kRobust = 1, kNanCoinc = 2 };

template <typename F, int V>
__device__ __forceinline__ void pair_field(F sx, F sy, F sz, F px, F py, F pz, F tx, F ty, F tz, F eps,
                                           F& ax, F& ay, F& az, F& bx, F& by, F& bz) {
    using M = Math<F>;
    const F rx = sx - tx, ry = sy - ty, rz = sz - tz;
    const F d2 = M::fma(rz, rz, M::fma(ry, ry, rx * rx));
    const F pr = M::fma(pz, rz, M::fma(py, ry, px * rx));
    F w, a;
    if (V == kFast) {
        const F d = M::sqrt_fast(d2);
        const F den = M::fma(d2, d, eps);
        const F q = M::rcp_fast(M::fma(d2, den, M::kTiny));
        w = d2 * q;
        a = pr * q;
    } else {
        const bool coinc = (d2 == F(0));
        const F d2m = coinc ? M::kHuge : d2;
        const F inv = M::rsq(d2m);
        const F d = d2m * inv;
        w = M::rcp(M::fma(d2m, d, eps));
        if (V == kNanCoinc) w = coinc ? F(__builtin_nanf("")) : w;
        a = pr * (inv * inv * w);
    }
    ax = M::fma(a, rx, ax);
    ay = M::fma(a, ry, ay);
    az = M::fma(a, rz, az);
    bx = M::fma(w, px, bx);
    by = M::fma(w, py, by);
    bz = M::fma(w, pz, bz);
}

// Far-field chain (eps > 0, every pair of the wave farther apart than far_d2 = (eps / kFarRatio)^(2/3), i.e.
// e = eps / |r|^3 < kFarRatio): ONE transcendental instead of two.
//   u = rsq(d2); u3 = u^3 = 1/|r|^3;  w = 1/(|r|^3 + eps) = u3 / (1 + e) = u3 (1 - e + e^2 - ...),  e = eps u3
// truncated after e^2: relative error e^3 < kFarRatio^3 = 5.1e-7 for the nearest far pairs and falling with |r|^-9
// (kFarRatio 2e-3 -> 4e-3 in round 2: 1.3 % faster; 4e-3 -> 8e-3 in round 3: another 0.9 %, 4.081 against 4.120 ms
// same-box; the error of the summed slabs against fp64 unchanged both times at 1.9e-8 median / 5.8e-8 max, a slab row
// at the threshold distance carries up to 5e-7 of systematic error - the bound is 1e-5; 1.2e-2 would give 1.2 % for
// 1.7e-6: profiles/r03_ab_far_ratio.txt).  22 full-rate + 1
// quarter-rate instructions against 19 + 2 for the exact chain (a transcendental costs ~13 issue cycles when
// mixed with FMAs).  No coincident pair can be in a far tile.
#ifndef DNP_FAR_RATIO
#define DNP_FAR_RATIO 8e-3
#endif
constexpr double kFarRatio = DNP_FAR_RATIO;
// Host side: the squared box distance from which the far chains run; 0 switches them off.  They need a normal-range
// threshold: for a tiny eps (< 1e-30) far_d2 would admit pairs whose u^3 = rsq(d2)^3 overflows fp32 (then e = eps*inf
// and w = NaN where the exact chain returns a finite 1/(|r|^3+eps)), so such calls keep the exact chain.
constexpr double kFarMinEps = 1e-30;
// fp64 (round 5): ONE far tier with the series taken to e^4 - 1/(1 + e) = 1 - e + e^2 - e^3 + e^4 - ..., truncation e^5 < 7.8e-17
// (below half an fp64 ulp) for e = eps / |r|^3 < kFarRatio64 = 6e-4, i.e. boxes farther apart than 0.255 for eps = 1e-5: the
// fp64 chain's second transcendental (v_rcp_f64, 16 cycles + 3 refinement instructions) becomes 4 fused multiply-adds.
constexpr double kFarRatio64 = 6e-4;
static inline double far_threshold_d2(double eps, double ratio = kFarRatio) {
    return eps >= kFarMinEps ? __builtin_pow(eps / ratio, 2.0 / 3.0) : 0.0;
}

template <typename F>
__device__ __forceinline__ void pair_field_far(F sx, F sy, F sz, F px, F py, F pz, F tx, F ty, F tz, F eps,
                                               F& ax, F& ay, F& az, F& bx, F& by, F& bz) {
    using M = Math<F>;
    const F rx = sx - tx, ry = sy - ty, rz = sz - tz;
    const F d2 = M::fma(rz, rz, M::fma(ry, ry, rx * rx));
    const F pr = M::fma(pz, rz, M::fma(py, ry, px * rx));
    const F u = M::rsq_fast(d2);                    // (fp32: the hardware's v_rsq_f32; fp64: v_rsq_f64 + one third-order step)
    const F u2 = u * u;
    const F u3 = u2 * u;
    const F e = eps * u3;
    F w;
    if constexpr (sizeof(F) == 8) {                 // fp64: the series to e^4 (see kFarRatio64)
        const F t = M::fma(e, M::fma(e, M::fma(e, e - F(1), F(1)), F(-1)), F(1));
        w = u3 * t;
    } else {
        w = M::fma(u3, M::fma(e, e, -e), u3);
    }
    const F a = pr * (w * u2);
    ax = M::fma(a, rx, ax);
    ay = M::fma(a, ry, ay);
    az = M::fma(a, rz, az);
    bx = M::fma(w, px, bx);
    by = M::fma(w, py, by);
    bz = M::fma(w, pz, bz);
}

// Second tier (e = eps / |r|^3 < kFar2Ratio): the e^2 term is dropped as well - w = u3 (1 - e), one instruction less
// (21 full-rate + 1 transcendental).  kFar2Ratio was 2.4e-4 in round 2 (e^2 below one fp32 ulp); round 3 put it at 7e-4,
// where the dropped e^2 <= 4.9e-7 matches the first tier's e^3 <= 5.1e-7: 4.070 against 4.131 ms same-box (-1.5 %),
// summed-slab error against fp64 unchanged (profiles/r03_ab_far_ratio.txt; 1e-3 would give -1.8 % for e^2 <= 1e-6).
#ifndef DNP_FAR2      // 1: second tier on (4.187 -> 4.096 ms per launch on the bench workload, summed-slab accuracy unchanged)
#define DNP_FAR2 1
#endif
constexpr double kFar2Ratio = 7e-4;
template <typename F>
__device__ __forceinline__ void pair_field_far2(F sx, F sy, F sz, F px, F py, F pz, F tx, F ty, F tz, F eps,
                                                F& ax, F& ay, F& az, F& bx, F& by, F& bz) {
    using M = Math<F>;
    const F rx = sx - tx, ry = sy - ty, rz = sz - tz;
    const F d2 = M::fma(rz, rz, M::fma(ry, ry, rx * rx));
    const F pr = M::fma(pz, rz, M::fma(py, ry, px * rx));
    const F u = M::rsq(d2);
    const F u2 = u * u;
    const F u3 = u2 * u;
    const F e = eps * u3;
    const F w = M::fma(-e, u3, u3);
    const F a = pr * (w * u2);
    ax = M::fma(a, rx, ax);
    ay = M::fma(a, ry, ay);
    az = M::fma(a, rz, az);
    bx = M::fma(w, px, bx);
    by = M::fma(w, py, by);
    bz = M::fma(w, pz, bz);
}

template <typename F>
__device__ __forceinline__ F wave_min(F v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const F o = __shfl_xor(v, off, 64); v = o < v ? o : v; }
    return v;
}
template <typename F>
__device__ __forceinline__ F wave_max(F v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) { const F o = __shfl_xor(v, off, 64); v = o > v ? o : v; }
    return v;
}

template <typename F>
__device__ __forceinline__ void pair_potential(F sx, F sy, F sz, F px, F py, F pz, F tx, F ty, F tz, F& phi) {
    using M = Math<F>;
    const F rx = sx - tx, ry = sy - ty, rz = sz - tz;
    const F d2 = M::fma(rz, rz, M::fma(ry, ry, rx * rx));
    const F inv = M::rsq_fast(d2);                  // d2 == 0 -> inf (fp64: NaN), pr == 0 -> 0*inf = NaN (as 0/0)
    const F pr = M::fma(pz, rz, M::fma(py, ry, px * rx));
    phi = M::fma(pr, inv * inv * inv, phi);
}

// LDS image of one staged source row: two 16-byte slots, (x,y,z,px) and (py,pz,-,-).
template <typename F> struct Vec4 { F x, y, z, w; };

template <typename F, typename PT, int MODE, int KT, int V>
__global__ __launch_bounds__(kBlock) void pair_kernel(const PairArgs<F, PT> a) {
    using M = Math<F>;
    constexpr int NC = (MODE == kField) ? 3 : 1;
    __shared__ __attribute__((aligned(16))) Vec4<F> lds[2][kBlock][2];
    DNP_STAMP_BEGIN();

    const int tid = threadIdx.x;
    const int64_t chunk = blockIdx.y;
    int64_t s_begin, s_end;
    if (a.chunk_off_dev) {
        s_begin = a.chunk_off_dev[DNP_BND(a.chunk_base + chunk, n_chunk_off, kBndChunkOff)];
        s_end = a.chunk_off_dev[DNP_BND(a.chunk_base + chunk + 1, n_chunk_off, kBndChunkOff)];
    } else {
        s_begin = a.chunk_off[chunk];
        s_end = a.chunk_off[chunk + 1];
    }
#ifdef DNP_BOUNDS   // the source range comes from a caller's table: it must lie inside the source rows
    if (s_begin < 0 || s_begin > s_end || s_end > a.bnd.n_src_rows) {
        if (a.bnd.err && (threadIdx.x & 63) == 0) atomicAdd(a.bnd.err + kBndSrc, 1u);
        s_begin = s_end = 0;
    }
#endif
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
    // (a staged row travels BY VALUE: with six scalars captured by reference hipcc kept two of them in scratch memory - a
    // store -> load round trip through the private segment in every workgroup's prologue and once per tile, and a kernel
    // that needs a private segment at all; found in round 4 by tools/isa_resources.py's scratch column)
    struct Staged { Vec4<F> lo, hi; };
    auto load_row = [&](int64_t s) -> Staged {
        Staged r;
        r.lo = Vec4<F>{M::kFar, M::kFar, M::kFar, F(0)};      // padding row: infinitely far, zero dipole -> contributes exactly 0
        r.hi = Vec4<F>{F(0), F(0), F(0), F(0)};
        if (s < s_end) {
            const int64_t row = a.src_idx ? a.src_idx[s] : s;
            const F* p = a.src + row * a.ld_src;
            r.lo = Vec4<F>{p[0], p[1], p[2], p[3]};
            r.hi = Vec4<F>{p[4], p[5], F(0), F(0)};
        }
        return r;
    };
    auto store_row = [&](int buf, const Staged& r) {
        lds[buf][tid][0] = r.lo;
        lds[buf][tid][1] = r.hi;
    };
    Staged staged;

    const int64_t n_src = s_end - s_begin;
    const int64_t n_tiles = (n_src + kBlock - 1) / kBlock;
    if (n_tiles > 0) {
        staged = load_row(s_begin + tid);
        store_row(0, staged);
    }
    __syncthreads();

    for (int64_t it = 0; it < n_tiles; ++it) {
        const int buf = (int)(it & 1);
        const int64_t tile_s = s_begin + it * kBlock;
        if (it + 1 < n_tiles) staged = load_row(tile_s + kBlock + tid);   // in flight during the compute below
        int n_here = (int)((s_end - tile_s) < kBlock ? (s_end - tile_s) : kBlock);
        n_here = (n_here + 3) & ~3;                                // rows past s_end are padding rows (kUnroll | 4)

        for (int j0 = 0; j0 < n_here; j0 += kFlush) {
            const int j1 = (j0 + kFlush < n_here) ? j0 + kFlush : n_here;   // multiples of kSets
            if (MODE == kField) {
                // kSets interleaved accumulator sets: source j goes to set j % kSets, so an fp32 chain is
                // kFlush / kSets = 16 adds long (no extra loop instructions, only registers)
                F A[kSets][KT][3], B[kSets][KT][3];
#pragma unroll
                for (int u = 0; u < kSets; ++u)
#pragma unroll
                    for (int k = 0; k < KT; ++k)
#pragma unroll
                        for (int c = 0; c < 3; ++c) A[u][k][c] = B[u][k][c] = F(0);
                for (int j = j0; j < j1; j += kUnroll) {
#pragma unroll
                    for (int u = 0; u < kUnroll; ++u) {
                        const Vec4<F> s0 = lds[buf][j + u][0];
                        const Vec4<F> s1 = lds[buf][j + u][1];
                        constexpr int kS = kSets;
#pragma unroll
                        for (int k = 0; k < KT; ++k)
                            pair_field<F, V>(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx[k], ty[k], tz[k], a.eps,
                                             A[u % kS][k][0], A[u % kS][k][1], A[u % kS][k][2], B[u % kS][k][0],
                                             B[u % kS][k][1], B[u % kS][k][2]);
                    }
                }
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int c = 0; c < 3; ++c) {
                        const F as = (kSets == 4) ? (A[0][k][c] + A[1][k][c]) + (A[2 % kSets][k][c] + A[3 % kSets][k][c])
                                                  : (A[0][k][c] + A[1][k][c]);
                        const F bs = (kSets == 4) ? (B[0][k][c] + B[1][k][c]) + (B[2 % kSets][k][c] + B[3 % kSets][k][c])
                                                  : (B[0][k][c] + B[1][k][c]);
                        acc[k][c] += 3.0 * (double)as - (double)bs;
                    }
            } else {
                F P[kSets][KT];
#pragma unroll
                for (int u = 0; u < kSets; ++u)
#pragma unroll
                    for (int k = 0; k < KT; ++k) P[u][k] = F(0);
                for (int j = j0; j < j1; j += kSets) {
#pragma unroll
                    for (int u = 0; u < kSets; ++u) {
                        const Vec4<F> s0 = lds[buf][j + u][0];
                        const Vec4<F> s1 = lds[buf][j + u][1];
#pragma unroll
                        for (int k = 0; k < KT; ++k)
                            pair_potential<F>(s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, tx[k], ty[k], tz[k], P[u][k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < KT; ++k)
                    acc[k][0] += (kSets == 4) ? (double)((P[0][k] + P[1][k]) + (P[2 % kSets][k] + P[3 % kSets][k]))
                                              : (double)(P[0][k] + P[1][k]);
            }
        }
        if (it + 1 < n_tiles) store_row(buf ^ 1, staged);
        __syncthreads();
    }

    // ---- epilogue: partial[chunk][t][:] -----------------------------------------------------
    const int64_t chunk_id = a.chunk_base + chunk;
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int64_t t = tile_base + k * kBlock + tid;
        if (t < a.T) {
            bool excluded = false;
            if (a.tgt_group) excluded = (a.tgt_group[DNP_BND(trow[k], n_tgt_group, kBndTgtGroup)] == chunk_id);
            if (a.out) {
                // single chunk, single leaf: the second pass would only filter and copy - do it here
                F* o = a.out + (a.out_scatter ? trow[k] : t) * a.ld_out;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    F v = (F)((MODE == kField) ? -acc[k][c] : acc[k][c]);
                    if (!__builtin_isfinite(v)) {                    // field_utils.py:110-115 / :53-54
                        if (a.nonfinite) atomicAdd(a.nonfinite + (v != v ? 1 : 0), 1);
                        v = F(0);
                    }
                    o[c] = a.accumulate ? (F)(o[c] + v) : v;
                }
            } else {
                PT* o = a.partial + DNP_BND(((int64_t)chunk * a.T + t) * NC, n_partial - (NC - 1), kBndPartial);
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    PT v = (PT)((MODE == kField) ? -acc[k][c] : acc[k][c]);
                    // patch mode: a slab is the complete dE of one field_grad call (one leaf), so the reference's
                    // Inf/NaN zeroing (field_utils.py:114-115) applies here; otherwise reduce_kernel applies it
                    if (a.tgt_group && !__builtin_isfinite(v)) v = PT(0);
                    o[c] = excluded ? PT(0) : v;
                }
            }
        }
    }
    DNP_STAMP_END();
}

// ---- pair_kernel_scalar: sources through the scalar unit (see the header comment); contiguous source rows only
// (a.src_idx is ignored: the launchers send gathered sources to pair_kernel) -------------------------------------
// NOTE ON THIS KERNEL'S SOURCE FORM.  hipcc's placement of the s_load / s_waitcnt pairs in the two inner loops is
// very sensitive to how the loop is written: semantically identical variants of scalar_field_run (row gather ternary
// removed; loads hoisted in front of the group; explicit next-group prefetch) measured 4.56-5.14 ms per 10^10 pairs
// against 4.36-4.42 ms for the form below in interleaved same-process A/B runs (profiles/r02_ab_scalar_forms.txt).
// Re-run tools/gpu_ab_far.py after ANY edit here.  (One edit that did pay, late in round 2: the 32-bit trip count below.)
// FAR: a wave whose target box is farther than sqrt(far_d2) from the box of the chunk's sources runs the whole chunk
// through pair_field_far (one decision per wave and chunk).  BOX: the chunks' boxes come from a.chunk_box (the patch
// drivers compute them once per cloud, dnp_patch_boxes_f32); otherwise the workgroup finds its chunk's box itself
// (a scan of the chunk, 36 ds_bpermute and a barrier per workgroup: 1.1 % of the bench launch).
#ifndef DNP_LOOP32     // 1: 32-bit trip count for the group loop - with the 64-bit `s + kUnroll <= run_end` form the compare ran on
#define DNP_LOOP32 1  // the VALU (no scalar 64-bit order compare on gfx950): 2 of 94 VALU instructions per group, 2.6 % of a launch
#endif
template <typename F, int KT, int V, int FARCHAIN>
__device__ __forceinline__ void scalar_field_run(const F* __restrict__ src, const int64_t* __restrict__ sidx, int64_t ld,
                                               int64_t& s, int64_t run_end, const F (&tx)[KT], const F (&ty)[KT],
                                               const F (&tz)[KT], F eps, double (&acc)[KT][3]) {
    F A[kSets][KT][3], B[kSets][KT][3];
#pragma unroll
    for (int u = 0; u < kSets; ++u)
#pragma unroll
        for (int k = 0; k < KT; ++k)
#pragma unroll
            for (int c = 0; c < 3; ++c) A[u][k][c] = B[u][k][c] = F(0);
    auto one = [&](int64_t row, int set) {
        const F* p = src + row * ld;                                    // uniform address: scalar loads
        const F sx = p[0], sy = p[1], sz = p[2], px = p[3], py = p[4], pz = p[5];
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            if (FARCHAIN == 2)
                pair_field_far2<F>(sx, sy, sz, px, py, pz, tx[k], ty[k], tz[k], eps, A[set][k][0], A[set][k][1],
                                   A[set][k][2], B[set][k][0], B[set][k][1], B[set][k][2]);
            else if (FARCHAIN == 1)
                pair_field_far<F>(sx, sy, sz, px, py, pz, tx[k], ty[k], tz[k], eps, A[set][k][0], A[set][k][1],
                                  A[set][k][2], B[set][k][0], B[set][k][1], B[set][k][2]);
            else
                pair_field<F, V>(sx, sy, sz, px, py, pz, tx[k], ty[k], tz[k], eps, A[set][k][0], A[set][k][1],
                                 A[set][k][2], B[set][k][0], B[set][k][1], B[set][k][2]);
        }
    };
#if DNP_LOOP32
    const int n_groups = (int)((run_end - s) / kUnroll);            // <= 64: a 32-bit trip count keeps the compare scalar
    for (int g = 0; g < n_groups; ++g, s += kUnroll) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) one(sidx ? sidx[s + u] : s + u, u % kSets);
    }
#else
    for (; s + kUnroll <= run_end; s += kUnroll) {
#pragma unroll
        for (int u = 0; u < kUnroll; ++u) one(sidx ? sidx[s + u] : s + u, u % kSets);
    }
#endif
    for (; s < run_end; ++s) one(sidx ? sidx[s] : s, 0);
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const F as = (kSets == 4) ? (A[0][k][c] + A[1][k][c]) + (A[2 % kSets][k][c] + A[3 % kSets][k][c])
                                      : (A[0][k][c] + A[1][k][c]);
            const F bs = (kSets == 4) ? (B[0][k][c] + B[1][k][c]) + (B[2 % kSets][k][c] + B[3 % kSets][k][c])
                                      : (B[0][k][c] + B[1][k][c]);
            acc[k][c] += 3.0 * (double)as - (double)bs;
        }
}

// TBOX: the box of a wavefront's 64 KT targets comes from a.tile_box (one table per cloud) instead of 36 cross-lane
// min / max steps per (wavefront, chunk).  WPART = 2 or 3 (patch mode, cloud sorted by patch, every tile inside <= WPART
// groups; 0 = off): the epilogue also leaves sum_t dE[t] . n_t of the tile's targets, split by group (2: the tile's first
// group / the other one; 3, round 5: first / first + 1 / the rest - patches of 100..127 points, the reference's own grid
// partitions, put three groups into a 128-row tile), in a.w_part - the patch interaction matrix W then needs no second pass
// over the slabs (dnp_interactions_from_tiles).
//
// SS (1, 2, 4): SOURCE SPLIT.  With SS = 1 the wavefronts of a workgroup own one target tile each and run the whole
// chunk; with SS > 1 a tile's chunk is evaluated by SS wavefronts and the partial sums are added IN A FIXED ORDER - the
// same fp32 runs and the same fp64 additions whatever SS, so results do not depend on it, and the SS = 1 loop keeps its
// form (a first version that rewrote the run loop around quarter cuts cost the SS = 1 path 1 % through nothing but the
// rewritten loop; profiles/r03_ab_source_split.txt).  A work item is up to SS times shorter, so the end of a launch - the
// last items on a chip that is emptying - shrinks with it, for more wavefronts with the same prologue.  Two forms:
//   * fp64 partial slabs (the generic entry points, far-field launches): the chunk is cut into SS equal PARTS, the four
//     wavefronts of a workgroup take one each with their own far-field decision, the part sums meet in LDS and wavefront 0
//     adds them in part order ("kParts" below).
//   * XCH (round 4; fp32 slabs, the patch drivers' tabled form): the split WITHOUT LDS, for the LAST chunks of a launch.
//     Wavefront i of a split item runs the chunk's i-th RUN of 128 sources (chunks of more than SS runs are left to
//     wavefront 0), writes its run term to the exchange buffer with write-through (sc1) stores, waits for them
//     (vmcnt(0)) and draws a ticket from the item's arrival counter with one agent-scope atomic add; the wavefront that
//     draws the last ticket acquires, reads the SS terms back, adds them in run order and runs the epilogue; the others
//     are done.  No LDS, no barrier, nobody waits, correct for any placement of the wavefronts over CUs and XCDs
//     (MI355X_MICROARCH.md, price list rows dequeue / splitk-seam: sc1 slab stores + a ticket + the last arriver).  The
//     launch is a 1-D grid: first the chunks [0, split_from) unsplit - WAVES wavefronts per workgroup on WAVES target
//     tiles -, then the split chunks, one (tile, run) item per wavefront; workgroups start in grid order, so the launch's
//     last resident set consists of items a third as long and the chip drains in a third of the time
//     (profiles/r03_timeline.txt: 49 us of a 0.56 ms launch were drain).  Round 3 built this tail with the run terms in
//     LDS: a kernel whose wavefronts share nothing pays 2.4 % for merely CARRYING 9 KB of LDS and a barrier
//     (profiles/r03_tail_fill.txt), so that tail lost above 44 patches per launch; this one pays at every size
//     (profiles/r04_xch_ab.txt: -11 % at 16 patches, -3.7 % at 32 - a rank's share of eight -, -0.5 % at 128) and the hand-off
//     itself - stores, wait, ticket, acquire, 24 loads - is 3 % of a split item.
// WAVES: wavefronts per workgroup (launch with WAVES * 64 threads).  The wavefronts of the tabled kernel do not cooperate
// (SS = 1 or XCH, boxes from tables), so the workgroup is only the unit of dispatch: with the XCD-aware tile mapping
// below, 2 wavefronts measured 4.049 ms on the bench launch against 4.074 with 4 (a CU takes a new pair of wavefronts as
// soon as two slots are free) and 4.060 with 1 (profiles/r03_ab_block.txt).  The XCH launches use 4: a tile's four runs
// then start together on one CU, and split items in workgroups of 1 or 2 wavefronts cost +25 % instead of +4 %
// (profiles/r04_xch_ab.txt).
template <typename F, typename PT, int MODE, int KT, int V, bool FAR = false, bool BOX = false, bool TBOX = false,
          int WPART = 0, int SS = 1, int WAVES = kBlock / 64, bool XCH = false>
__global__ __launch_bounds__(WAVES * 64) void pair_kernel_scalar(const PairArgs<F, PT> a) {
    using M = Math<F>;
    constexpr int NC = (MODE == kField) ? 3 : 1;
    constexpr bool kFarPath = FAR && MODE == kField && V == kFast;
    static_assert(SS == 1 || SS == 2 || SS == 4, "source split: 1, 2 or 4 wavefronts per target tile");
    static_assert(WPART == 0 || WPART == 2 || WPART == 3, "interaction partials: off, or 2 / 3 group slots per tile");
    static_assert(SS == 1 || MODE == kField, "the source split is built for the field mode");
    static_assert(XCH || (WAVES % SS == 0 && WAVES >= SS), "a workgroup holds whole target tiles");
    constexpr int kTG = XCH ? WAVES : WAVES / SS;           // target tiles per workgroup (XCH: of an unsplit workgroup)
    static_assert(!XCH || (SS > 1 && BOX && TBOX && sizeof(PT) == 4), "XCH: the tabled patch-mode kernel (no LDS, no barrier)");
    static_assert(XCH || SS == 1 || sizeof(PT) == 8, "fp32 slabs split their items through the exchange buffer (XCH)");
    DNP_STAMP_BEGIN_SCALAR();
    __shared__ F chunk_box[WAVES][6];
    constexpr bool kLdsSplit = SS > 1 && !XCH;              // the split forms whose run terms meet in LDS
    __shared__ double split_terms[kLdsSplit ? SS - 1 : 1][kLdsSplit ? kTG : 1][kLdsSplit ? KT : 1][kLdsSplit ? NC : 1][kLdsSplit ? 64 : 1];
    const int tid = threadIdx.x;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);   // an SGPR: everything derived from it stays wave-uniform,
    int tg = XCH ? wave : wave / SS, sp = XCH ? 0 : wave % SS;   // above all the source range (scalar loads); tile / source part
    int64_t chunk = blockIdx.y;
    int64_t xch_item = 0;                                   // XCH: (split chunk, target tile) index of this wavefront's item
    // XCD-aware tile mapping.  Workgroups are dealt round-robin over the 8 XCDs in launch order (x fastest: XCD = linear
    // id mod 8 - observed behaviour, MI355X_MICROARCH.md; only speed depends on it).  With a grid width that is not a
    // multiple of 8 a target tile would wander over the XCDs from one chunk row to the next, and every XCD's L2 would
    // fetch every target row and take part in every output line: with 391-wide grids (two-wavefront workgroups) the
    // kernel's L2 fetches were 148 MB per launch against 17 MB (profiles/r03_ab_block.txt).  So inside each aligned
    // group of 8 x-blocks the tile index is rotated by the row's offset: tile t always runs on XCD t mod 8.
#ifndef DNP_XCD_MAP
#define DNP_XCD_MAP 1
#endif
    unsigned bx = blockIdx.x;
    bool row_split = SS > 1;
    int ktg = kTG;                                           // target tiles of this workgroup
    if constexpr (XCH) {
        const unsigned n_tiles = (unsigned)((a.T + 64 * KT - 1) / (64 * KT));
        const unsigned gxa = (n_tiles + WAVES - 1) / WAVES;
        const unsigned unsplit_blocks = (unsigned)a.split_from * gxa;
        const unsigned b = blockIdx.x;
        if (b < unsplit_blocks) {                               // the product's unsplit form: WAVES tiles per workgroup
            const unsigned row = b / gxa;
            bx = b - row * gxa;
            if (DNP_XCD_MAP && bx < (gxa & ~7u)) bx = (bx & ~7u) | ((bx + row * gxa) & 7u);
            chunk = row;
            row_split = false;
        } else {                                                // one (tile, run) item per wavefront
            // Workgroups go to the XCDs round-robin by their linear id, and a tile's runs are NOT equal work (a 390-point
            // patch is 128 + 128 + 128 + 6 sources): with the workgroups of a tile on consecutive ids every XCD saw one kind
            // of run only - even XCDs runs 0-1, odd XCDs runs 2-3 - and the split patches took 1.3 times as long as they
            // should (profiles/r04_xch_ab.txt: all patches split +24 % with two-wavefront workgroups, +5 % with four).  So
            // inside a patch the ids are dealt XCD-first: id = (8 j + x), tile = 8 (j / h) + x, runs WAVES (j % h) ..., with
            // h = SS / WAVES workgroups per tile - a tile's workgroups share an XCD and every XCD gets every kind of run.
            static_assert(SS % WAVES == 0, "XCH: whole workgroups per tile");
            static_assert(SS % WAVES == 0, "XCH: whole workgroups per tile");
            constexpr unsigned kH = SS / WAVES;
            const unsigned n_t8 = (n_tiles + 7u) & ~7u;         // tiles padded to whole groups of 8 (the surplus ones exit)
            const unsigned c = b - unsplit_blocks;
            const unsigned row = c / (n_t8 * kH);
            const unsigned r = c - row * (n_t8 * kH);
            const unsigned j = r >> 3;
            const unsigned tile = (j / kH) * 8u + (r & 7u);
            if (tile >= n_tiles) return;
            sp = (int)((j % kH) * WAVES) + wave;
            bx = tile;
            chunk = (int64_t)a.split_from + row;
            ktg = 1;
            tg = 0;
            xch_item = (int64_t)row * n_tiles + tile;
        }
    } else {
        if (DNP_XCD_MAP && bx < (gridDim.x & ~7u)) bx = (bx & ~7u) | ((bx + blockIdx.y * gridDim.x) & 7u);
    }
    if (a.chunk_perm) chunk = a.chunk_perm[DNP_BND(chunk, n_chunk_perm, kBndChunkOff)];     // launch row -> chunk
    int64_t s_begin, s_end;
    if (a.chunk_off_dev) {
        s_begin = a.chunk_off_dev[DNP_BND(a.chunk_base + chunk, n_chunk_off, kBndChunkOff)];
        s_end = a.chunk_off_dev[DNP_BND(a.chunk_base + chunk + 1, n_chunk_off, kBndChunkOff)];
    } else {
        s_begin = a.chunk_off[chunk];
        s_end = a.chunk_off[chunk + 1];
    }
#ifdef DNP_BOUNDS   // the source range comes from a caller's table: it must lie inside the source rows
    if (s_begin < 0 || s_begin > s_end || s_end > a.bnd.n_src_rows) {
        if (a.bnd.err && (threadIdx.x & 63) == 0) atomicAdd(a.bnd.err + kBndSrc, 1u);
        s_begin = s_end = 0;
    }
#endif
    const int64_t tile_base = (int64_t)bx * (ktg * 64 * KT);
    F tx[KT], ty[KT], tz[KT];
    int64_t trow[KT];
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int64_t t = tile_base + (int64_t)tg * (64 * KT) + k * 64 + (tid & 63);   // a tile = 64 KT consecutive targets
        trow[k] = -1;
        tx[k] = ty[k] = tz[k] = F(0);
        if (t < a.T) {
            const int64_t row = a.tgt_idx ? a.tgt_idx[t] : t;
            trow[k] = row;
            const F* p = a.tgt + row * a.ld_tgt;
            tx[k] = p[0]; ty[k] = p[1]; tz[k] = p[2];
        }
    }
    const F* __restrict__ src = a.src;
    const int64_t* __restrict__ sidx = a.src_idx;
    const int64_t ld = a.ld_src;

    // SS > 1 with fp64 partial slabs (the generic entry points): the chunk is cut into SS contiguous PARTS of equal
    // length (a multiple of kFlush), wavefront sp takes part sp WITH ITS OWN far-field decision from its own part's box -
    // a chunk can then be SS times longer (a partial slab SS times smaller) at the same item length and the same
    // granularity of the far test.  (fp32 slabs, patch mode: run i of the chunk to wavefront i, see further down.)
    constexpr bool kParts = SS > 1 && sizeof(PT) == 8;
    int64_t part_lo = s_begin, part_hi = s_end;
    if constexpr (kParts) {
        const int64_t len = s_end - s_begin;
        const int64_t part = ((len + SS - 1) / SS + kFlush - 1) / kFlush * kFlush;
        part_lo = s_begin + (int64_t)sp * part;
        part_lo = part_lo < s_end ? part_lo : s_end;
        part_hi = part_lo + part < s_end ? part_lo + part : s_end;
    }

    int far_chunk = 0;
    if (kFarPath && a.far_d2 > F(0)) {     // far_d2 <= 0: the launcher switched the far machinery off (small problems)
        // box of the chunk's sources (given, or found by the workgroup) and of this wave's targets
        constexpr bool given = BOX;          // a compile-time choice: a run-time branch here cost 19 VGPRs (occupancy 8 -> 6)
        static_assert(!(kParts && BOX), "source parts find their own boxes");
        F plo[3] = {M::kHuge, M::kHuge, M::kHuge}, phi[3] = {-M::kHuge, -M::kHuge, -M::kHuge};   // kParts: this part's box
        if constexpr (kParts) {
            for (int64_t q = part_lo + (tid & 63); q < part_hi; q += 64) {
                const F* p = src + q * ld;
#pragma unroll
                for (int c = 0; c < 3; ++c) { plo[c] = p[c] < plo[c] ? p[c] : plo[c]; phi[c] = p[c] > phi[c] ? p[c] : phi[c]; }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { plo[c] = wave_min<F>(plo[c]); phi[c] = wave_max<F>(phi[c]); }
        } else if constexpr (!given) {
            F lo[3] = {M::kHuge, M::kHuge, M::kHuge}, hi[3] = {-M::kHuge, -M::kHuge, -M::kHuge};
            for (int64_t q = s_begin + tid; q < s_end; q += WAVES * 64) {
                const F* p = src + (sidx ? sidx[q] : q) * ld;
#pragma unroll
                for (int c = 0; c < 3; ++c) { lo[c] = p[c] < lo[c] ? p[c] : lo[c]; hi[c] = p[c] > hi[c] ? p[c] : hi[c]; }
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) { lo[c] = wave_min<F>(lo[c]); hi[c] = wave_max<F>(hi[c]); }
            if ((tid & 63) == 0)
#pragma unroll
                for (int c = 0; c < 3; ++c) { chunk_box[tid >> 6][c] = lo[c]; chunk_box[tid >> 6][3 + c] = hi[c]; }
        }
        F tlo[3] = {M::kHuge, M::kHuge, M::kHuge}, thi[3] = {-M::kHuge, -M::kHuge, -M::kHuge};
        if constexpr (TBOX) {
            // (the last workgroup's wavefronts may lie past the last tile: they have no targets, but the table has no
            // entry for them either - round 3's first form read up to 72 bytes past its end, found by tools/gpu_fuzz.py
            // as a memory access fault when the table ended on a page boundary)
            const int64_t n_tiles = (a.T + 64 * KT - 1) / (64 * KT);
            int64_t wave_tile = (int64_t)bx * ktg + __builtin_amdgcn_readfirstlane(tg);
#ifndef DNP_BUG_A8D48F5   // (defined only by the regression build of tests/test_gpu_bounds.py: the kernel as it was before the fix)
            wave_tile = wave_tile < n_tiles ? wave_tile : n_tiles - 1;
#else
            (void)n_tiles;
#endif
            const F* tb = a.tile_box + DNP_BND(wave_tile, n_tile_box, kBndTileBox) * 6;    // wave-uniform: scalar loads
#pragma unroll
            for (int c = 0; c < 3; ++c) { tlo[c] = tb[c]; thi[c] = tb[3 + c]; }
        } else {
#pragma unroll
            for (int k = 0; k < KT; ++k)
                if (trow[k] >= 0) {
                    tlo[0] = tx[k] < tlo[0] ? tx[k] : tlo[0]; thi[0] = tx[k] > thi[0] ? tx[k] : thi[0];
                    tlo[1] = ty[k] < tlo[1] ? ty[k] : tlo[1]; thi[1] = ty[k] > thi[1] ? ty[k] : thi[1];
                    tlo[2] = tz[k] < tlo[2] ? tz[k] : tlo[2]; thi[2] = tz[k] > thi[2] ? tz[k] : thi[2];
                }
#pragma unroll
            for (int c = 0; c < 3; ++c) { tlo[c] = wave_min<F>(tlo[c]); thi[c] = wave_max<F>(thi[c]); }
        }
        F d2box = F(0);
        if constexpr (kParts) {
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                F gap = plo[c] - thi[c];
                const F gap2 = tlo[c] - phi[c];
                gap = gap2 > gap ? gap2 : gap;
                gap = gap > F(0) ? gap : F(0);
                d2box = M::fma(gap, gap, d2box);
            }
        } else if constexpr (given) {
            const F* b = a.chunk_box + DNP_BND(a.chunk_base + chunk, n_chunk_box, kBndChunkBox) * 6;   // wave-uniform: scalar loads
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                F gap = b[c] - thi[c];
                const F gap2 = tlo[c] - b[3 + c];
                gap = gap2 > gap ? gap2 : gap;
                gap = gap > F(0) ? gap : F(0);
                d2box = M::fma(gap, gap, d2box);
            }
        } else {
            __syncthreads();
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                F slo = chunk_box[0][c], shi = chunk_box[0][3 + c];
#pragma unroll
                for (int w = 1; w < WAVES; ++w) {
                    slo = chunk_box[w][c] < slo ? chunk_box[w][c] : slo;
                    shi = chunk_box[w][3 + c] > shi ? chunk_box[w][3 + c] : shi;
                }
                F gap = slo - thi[c];
                const F gap2 = tlo[c] - shi;
                gap = gap2 > gap ? gap2 : gap;
                gap = gap > F(0) ? gap : F(0);
                d2box = M::fma(gap, gap, d2box);
            }
        }
#if DNP_FAR2
#ifndef DNP_FAR2_SCALE   // (kFarRatio / kFar2Ratio)^(2/3); A/B builds that move DNP_FAR_RATIO pass the matching value
#define DNP_FAR2_SCALE 5.0736
        static_assert(kFarRatio == 8e-3 && kFar2Ratio == 7e-4, "DNP_FAR2_SCALE is (kFarRatio / kFar2Ratio)^(2/3) = 5.0736 for these");
#endif
        constexpr F kFar2Scale = (F)DNP_FAR2_SCALE;
        if constexpr (sizeof(F) == 8) far_chunk = __builtin_amdgcn_readfirstlane((int)(d2box > a.far_d2));     // fp64: one tier
        else far_chunk = __builtin_amdgcn_readfirstlane((int)(d2box > a.far_d2) + (int)(d2box > a.far_d2 * kFar2Scale));
#else
        far_chunk = __builtin_amdgcn_readfirstlane((int)(d2box > a.far_d2));
#endif
    }

    double acc[KT][NC];
#pragma unroll
    for (int k = 0; k < KT; ++k)
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[k][c] = 0.0;

    // ---- the chunk's runs: kRun sources each from the chunk's start; with SS > 1 (patch mode: kRun = 128) run i of a
    // chunk of <= SS runs belongs to source part i, longer chunks to part 0 alone.  (Balanced runs - a 390-source chunk
    // as 98, 98, 98, 96 instead of 128, 128, 128, 6, so that no wavefront of a split item idles - made the split pay up
    // to 32 chunks per launch instead of 16 but cost the SS = 1 path 1 %: profiles/r03_ab_source_split.txt.) ----------
    constexpr int kRun = (sizeof(PT) == 4) ? kFlushScalar : kFlush;
    constexpr int64_t run_len = kRun;
    int64_t s = s_begin;                                    // wave-uniform
    int64_t part_end = s_end;
    bool exchange = false;                                  // SS > 1: this chunk's runs are spread over the source parts
    if constexpr (kParts) {
        exchange = true;
        s = part_lo;
        part_end = part_hi;
    } else if constexpr (SS > 1) {
        exchange = row_split && (s_end - s_begin) <= (int64_t)SS * run_len;
        if (exchange) {
            s = s_begin + (int64_t)sp * run_len;
            part_end = (s + run_len < s_end) ? s + run_len : s_end;
        } else if (sp != 0) {
            s = s_end;                                      // a long chunk: part 0 runs all of it
        }
    }
    while (s < part_end) {
        const int64_t run_end = (s + run_len < part_end) ? s + run_len : part_end;
        if constexpr (MODE == kField) {
#if DNP_FAR2
            if (kFarPath && far_chunk == 2) scalar_field_run<F, KT, V, 2>(src, sidx, ld, s, run_end, tx, ty, tz, a.eps, acc);
            else
#endif
            if (kFarPath && far_chunk) scalar_field_run<F, KT, V, 1>(src, sidx, ld, s, run_end, tx, ty, tz, a.eps, acc);
            else scalar_field_run<F, KT, V, 0>(src, sidx, ld, s, run_end, tx, ty, tz, a.eps, acc);
        } else {
            F P[kSets][KT];
#pragma unroll
            for (int u = 0; u < kSets; ++u)
#pragma unroll
                for (int k = 0; k < KT; ++k) P[u][k] = F(0);
            for (; s + kSets <= run_end; s += kSets) {
#pragma unroll
                for (int u = 0; u < kSets; ++u) {
                    const F* p = src + (sidx ? sidx[s + u] : s + u) * ld;
                    const F sx = p[0], sy = p[1], sz = p[2], px = p[3], py = p[4], pz = p[5];
#pragma unroll
                    for (int k = 0; k < KT; ++k) pair_potential<F>(sx, sy, sz, px, py, pz, tx[k], ty[k], tz[k], P[u][k]);
                }
            }
            for (; s < run_end; ++s) {
                const F* p = src + (sidx ? sidx[s] : s) * ld;
                const F sx = p[0], sy = p[1], sz = p[2], px = p[3], py = p[4], pz = p[5];
#pragma unroll
                for (int k = 0; k < KT; ++k) pair_potential<F>(sx, sy, sz, px, py, pz, tx[k], ty[k], tz[k], P[0][k]);
            }
#pragma unroll
            for (int k = 0; k < KT; ++k)
                acc[k][0] += (kSets == 4) ? (double)((P[0][k] + P[1][k]) + (P[2 % kSets][k] + P[3 % kSets][k]))
                                          : (double)(P[0][k] + P[1][k]);
        }
    }
    if constexpr (XCH) {
        if (row_split) {
            if (!exchange) {
                if (sp != 0) return;                                // a long chunk: part 0 has run all of it
            } else {
                // publish this run's term (a wavefront whose run lies behind the chunk's end publishes zeros): write-through stores
                // (512 contiguous bytes per instruction), then the ticket
                constexpr int64_t kItemDoubles = xch_item_bytes(SS, KT, NC) / 8, kItemWords = xch_item_bytes(SS, KT, NC) / 4;
                xch_item = DNP_BND(xch_item, n_xch_items, kBndXch);
                unsigned int* ticket = a.xch_ticket + xch_item * kItemWords;
                double* slot = a.xch_terms + xch_item * kItemDoubles + (sp * (KT * NC)) * 64 + (tid & 63);
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int c = 0; c < NC; ++c)
                        __hip_atomic_store(slot + (k * NC + c) * 64, acc[k][c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                // ORDERING (round-4 advisor): the hand-off uses relaxed agent-scope atomics only - in the language's memory model the
                // last arriver's acquire fence pairs with no release.  It is correct on THIS target because (1) the run terms are
                // stored write-through (agent-scope atomic stores compile to global_store ... sc1: they do not stay dirty in this
                // XCD's L2) and (2) on gfx9 vmcnt counts stores, so s_waitcnt vmcnt(0) below means the stores have been
                // acknowledged by the memory side before the ticket is drawn.  A release on the ticket would add a buffer_wbl2 - a
                // write-back of the whole L2's dirty lines (the other wavefronts' slab rows) per split item.  Both facts are pinned:
                // the static_assert here, and tests/test_isa_budget.py checks the emitted ISA (sc1 stores, vmcnt(0) before the atomic).
#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__) && !defined(__gfx90a__)
                static_assert(sizeof(F) == 0, "the exchange hand-off relies on gfx9 semantics: sc1 write-through stores + vmcnt counting stores");
#endif
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");    // this wavefront's stores have left before it signals
                unsigned drawn = 0;
                if ((tid & 63) == 0)
                    drawn = __hip_atomic_fetch_add(ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                drawn = (unsigned)__builtin_amdgcn_readfirstlane((int)drawn);
                if (drawn != SS - 1) return;                        // somebody else arrives last and finishes the item
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                if ((tid & 63) == 0)                                // re-armed for the next launch on this buffer
                    __hip_atomic_store(ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                const double* terms = a.xch_terms + xch_item * kItemDoubles + (tid & 63);
#pragma unroll
                for (int k = 0; k < KT; ++k) {
                    // one target's SS * NC loads in flight at a time: all KT * SS * NC at once (what hipcc schedules by itself)
                    // cost 66 VGPRs, i.e. one wavefront per SIMD for the WHOLE kernel
                    if (k > 0) __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int c = 0; c < NC; ++c) {                   // run order: ((t0 + t1) + t2) + t3, the unsplit form's additions
                        double sum = __hip_atomic_load(terms + (k * NC + c) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
                        for (int q = 1; q < SS; ++q)
                            sum += __hip_atomic_load(terms + ((q * KT + k) * NC + c) * 64, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        acc[k][c] = sum;
                    }
                }
            }
        }
    } else if (SS > 1 && row_split) {
        // run i's term travels on its own and part 0 adds the terms in run order: exactly the sums of the SS = 1 form
        if (exchange && sp != 0) {
#pragma unroll
            for (int k = 0; k < KT; ++k)
#pragma unroll
                for (int c = 0; c < NC; ++c) split_terms[sp - 1][tg][k][c][tid & 63] = acc[k][c];
        }
        __syncthreads();
        if (sp != 0) return;                                    // wavefront 0 of the tile finishes
        if (exchange) {
#pragma unroll
            for (int q = 1; q < SS; ++q)
#pragma unroll
                for (int k = 0; k < KT; ++k)
#pragma unroll
                    for (int c = 0; c < NC; ++c) acc[k][c] += split_terms[q - 1][tg][k][c][tid & 63];
        }
    }

#ifdef DNP_DUMMY_LDS   // experiment builds only (profiles/r03_tail_fill.txt): what does it cost a kernel whose wavefronts share nothing to
                       // CARRY an LDS allocation and a barrier (1: never executed - a.T < 0 is false; 2: executed by every workgroup)
    if (DNP_DUMMY_LDS == 2 || a.T < 0) {
#ifndef DNP_DUMMY_LDS_DOUBLES
#define DNP_DUMMY_LDS_DOUBLES 1152
#endif
        __shared__ double dummy_lds[DNP_DUMMY_LDS_DOUBLES];
        dummy_lds[tid % DNP_DUMMY_LDS_DOUBLES] = acc[0][0];
        __syncthreads();
        if (dummy_lds[(tid + 1) % (DNP_DUMMY_LDS_DOUBLES < WAVES * 64 ? DNP_DUMMY_LDS_DOUBLES : WAVES * 64)] == 1.2345e300) acc[0][0] += 1.0;
    }
#endif
    const int64_t chunk_id = a.chunk_base + chunk;
    double w_first = 0.0, w_other = 0.0;                  // WPART: this lane's share of the tile's interaction sums
    double w_second = 0.0;                                 // WPART == 3: the group behind the tile's first one
    int64_t g_first = 0;
    if constexpr (WPART != 0) {                             // group of the tile's first target (wave-uniform)
        const int row0 = __builtin_amdgcn_readfirstlane((int)trow[0]);
        g_first = row0 >= 0 ? a.tgt_group[DNP_BND(row0, n_tgt_group, kBndTgtGroup)] : -2;
    }
#pragma unroll
    for (int k = 0; k < KT; ++k) {
        const int64_t t = tile_base + (int64_t)tg * (64 * KT) + k * 64 + (tid & 63);
        if (t < a.T) {
            bool excluded = false;
            int64_t grp = -1;
            if (a.tgt_group) { grp = a.tgt_group[DNP_BND(trow[k], n_tgt_group, kBndTgtGroup)]; excluded = (grp == chunk_id); }
            if (a.out) {
                F* o = a.out + (a.out_scatter ? trow[k] : t) * a.ld_out;
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    F v = (F)((MODE == kField) ? -acc[k][c] : acc[k][c]);
                    if (!__builtin_isfinite(v)) {
                        if (a.nonfinite) atomicAdd(a.nonfinite + (v != v ? 1 : 0), 1);
                        v = F(0);
                    }
                    o[c] = a.accumulate ? (F)(o[c] + v) : v;
                }
            } else {
                PT* o = a.partial + DNP_BND(((int64_t)chunk * a.T + t) * NC, n_partial - (NC - 1), kBndPartial);
                PT vv[NC];
#pragma unroll
                for (int c = 0; c < NC; ++c) {
                    PT v = (PT)((MODE == kField) ? -acc[k][c] : acc[k][c]);
                    if (a.tgt_group && !__builtin_isfinite(v)) v = PT(0);
                    vv[c] = excluded ? PT(0) : v;
                    o[c] = vv[c];
                }
                if constexpr (WPART != 0 && MODE == kField) {
                    // per-point dot in fp32 like (E[patch] * pts[patch, 3:]).sum(dim=-1) (field_utils.py:316), patch sums in fp64
                    // (products rounded separately, added left to right - no fma contraction: W from this epilogue, W from
                    // interactions_kernel and torch's elementwise product + sum must not depend on a compiler's choice)
                    // (F = double, the fp64 drivers of round 5: the dot in fp64, as the reference's on a float64 cloud)
                    const F* n = a.tgt + trow[k] * a.ld_tgt + 3;
                    F d;
                    {
#pragma clang fp contract(off)
                        d = ((F)vv[0] * n[0] + (F)vv[1] * n[1]) + (F)vv[2] * n[2];
                    }
                    if constexpr (WPART == 3) {
                        if (grp == g_first) w_first += (double)d; else if (grp == g_first + 1) w_second += (double)d; else w_other += (double)d;
                    } else {
                        if (grp == g_first) w_first += (double)d; else w_other += (double)d;
                    }
                }
            }
        }
    }
#ifdef DNP_BOUNDS    // WPART's precondition: the tile's rows take at most WPART group values (else w_other mixes patches)
    if constexpr (WPART != 0 && MODE == kField) {
        auto named = [&](int64_t gq) { return gq == g_first || (WPART == 3 && gq == g_first + 1); };
        int64_t other = -9;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int64_t t = tile_base + (int64_t)tg * (64 * KT) + k * 64 + (tid & 63);
            if (t < a.T) { const int64_t gq = a.tgt_group[DNP_BND(trow[k], n_tgt_group, kBndTgtGroup)]; if (!named(gq)) other = gq; }
        }
        const int64_t omax = wave_max<int64_t>(other);
        bool bad = false;
#pragma unroll
        for (int k = 0; k < KT; ++k) {
            const int64_t t = tile_base + (int64_t)tg * (64 * KT) + k * 64 + (tid & 63);
            if (t < a.T) { const int64_t gq = a.tgt_group[DNP_BND(trow[k], n_tgt_group, kBndTgtGroup)]; bad |= (!named(gq) && gq != omax); }
        }
        if (__any(bad) && (tid & 63) == 0 && a.bnd.err) atomicAdd(a.bnd.err + kBndGroups, 1u);   // the CALLER's error, not an access
    }
#endif
    if constexpr (WPART != 0 && MODE == kField) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {             // fixed butterfly: the sums do not depend on anything but the tile
            w_first += __shfl_xor(w_first, off, 64);
            w_other += __shfl_xor(w_other, off, 64);
            if constexpr (WPART == 3) w_second += __shfl_xor(w_second, off, 64);
        }
        const int64_t n_tiles = (a.T + 64 * KT - 1) / (64 * KT);
        const int64_t wave_tile = (int64_t)bx * ktg + tg;
        if ((tid & 63) == 0 && wave_tile < n_tiles) {
            double* wp = a.w_part + DNP_BND(((int64_t)chunk * n_tiles + wave_tile) * WPART, n_w_part - (WPART - 1), kBndWPart);
            wp[0] = w_first;
            if constexpr (WPART == 3) { wp[1] = w_second; wp[2] = w_other; } else { wp[1] = w_other; }
        }
    }
    DNP_STAMP_END();
}

}  // namespace dnp
