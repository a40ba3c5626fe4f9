// ubench_mfma_mix.hip - can the f32 MFMA pipe take the dot products of the far-field pair chain off the vector ALU?
// One "step" = 512 pairs per wavefront.  Instruction streams per step (per lane: 8 pairs):
//   valu     the shipped far chain: 8 x (22 full-rate + 1 v_rsq_f32)              = 176 fma-class + 8 rsq
//   mfma     4 x v_mfma_f32_16x16x4_f32 (|s'|^2 - 2 s'.t' and p.s' - p.t' for 8 sources x 16 targets each)
//            + 8 x (15 full-rate + 1 v_rsq_f32)                                   = 120 fma-class + 8 rsq + 4 mfma
//   mfma_only / valu120 : the two halves of `mfma` alone
// at 2, 4 and 8 wavefronts per SIMD; prints cycles per step per SIMD (all wavefronts of a SIMD together).
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
#define F(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define F8 F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define F7 F(0) F(1) F(2) F(3) F(4) F(5) F(6)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)
// 22 fma + 1 rsq, and 15 fma + 1 rsq
#define PAIR23 F8 F8 F(0) F(1) F(2) F(3) F(4) F(5) T(7)
#define PAIR16 F8 F7 T(7)

template <int KIND>
__global__ void bench(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 1e-3f;
    f32x4 d0 = {0, 0, 0, 0}, d1 = d0, d2 = d0, d3 = d0;
    float av = a0 * 0.5f, bv = a1 * 0.25f;
    bf16x8 ab, bb;
    for (int j = 0; j < 8; ++j) { ab[j] = (__bf16)(a0 + j); bb[j] = (__bf16)(a1 - j); }
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {
            asm volatile(PAIR23 PAIR23 PAIR23 PAIR23 PAIR23 PAIR23 PAIR23 PAIR23 OPS);
        } else {
            if (KIND == 1 || KIND == 2) {
                d0 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, d3, 0, 0, 0);
            }
            if (KIND == 4 || KIND == 5) {       // the same with ONE bf16 16x16x32 MFMA per 16-target block (fp32 split in 3 bf16)
                d0 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d0, 0, 0, 0);
                d1 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d1, 0, 0, 0);
                d2 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d2, 0, 0, 0);
                d3 = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab, bb, d3, 0, 0, 0);
            }
            if (KIND == 1 || KIND == 3 || KIND == 4) {
                asm volatile(PAIR16 PAIR16 PAIR16 PAIR16 PAIR16 PAIR16 PAIR16 PAIR16 OPS);
            }
            if (KIND == 1 || KIND == 4) {   // the chain consumes the MFMA results: tie them in (two adds per step)
                a0 += d0[0] * 1e-30f; a1 += d1[1] * 1e-30f; a2 += d2[2] * 1e-30f; a3 += d3[3] * 1e-30f;
            }
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + d0[0] + d1[1] + d2[2] + d3[3];
    if (s == 123.456f) out[0] = s;
}

template <int KIND>
static void run(const char* name) {
    float* out; (void)hipMalloc(&out, 4);
    const int iters = 20000;
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int blocks = wps == 8 ? 512 : 256, threads = wps == 8 ? 1024 : 64 * 4 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        bench<KIND><<<blocks, threads>>>(out, iters);
        (void)hipEventRecord(e0);
        bench<KIND><<<blocks, threads>>>(out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        double ns_step = ms * 1e6 / ((double)iters * wps);   // per step per SIMD
        printf("%-26s wps=%d  %8.3f ms   %.1f ns per 512-pair step per SIMD  (= %.0f cycles at 2.3 GHz, %.3f cycles per pair)\n", name, wps, ms,
               ns_step, ns_step * 2.3, ns_step * 2.3 / 512);
    }
    (void)hipFree(out);
}

int main() {
    run<0>("valu: 8x(22 fma,rsq)");
    run<1>("mfma: 4 mfma + 8x(15 fma,rsq)");
    run<2>("mfma_only: 4 mfma");
    run<3>("valu120: 8x(15 fma,rsq)");
    run<4>("bf16: 4 mfma16x16x32 + 8x(15 fma,rsq)");
    run<5>("bf16_only: 4 mfma16x16x32");
    return 0;
}
