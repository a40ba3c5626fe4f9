// ubench_trans.hip - what does a quarter-rate transcendental cost when mixed with full-rate VALU work?
// Patterns of 56 v_fma_f32 + 8 v_rsq_f32 per 64 instructions with the 8 transcendentals spread
// (1 per 7 fma), paired, in fours, or all grouped; plus sqrt/rcp variants and a 19:2 mix.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define F(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define S(i) "v_sqrt_f32 %" #i ", %" #i "\n"
#define R(i) "v_rcp_f32 %" #i ", %" #i "\n"
#define F7 F(0) F(1) F(2) F(3) F(4) F(5) F(6)
#define F8 F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)

template <int KIND>
__global__ void bench(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {         // spread: (7 fma, 1 rsq) x 8
            asm volatile(F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) OPS);
        } else if (KIND == 1) {  // pairs: (14 fma, 2 rsq) x 4
            asm volatile(F7 F7 T(7) T(6) F7 F7 T(7) T(6) F7 F7 T(7) T(6) F7 F7 T(7) T(6) OPS);
        } else if (KIND == 2) {  // fours: (28 fma, 4 rsq) x 2
            asm volatile(F7 F7 F7 F7 T(7) T(6) T(5) T(4) F7 F7 F7 F7 T(7) T(6) T(5) T(4) OPS);
        } else if (KIND == 3) {  // grouped: 56 fma, 8 rsq
            asm volatile(F7 F7 F7 F7 F7 F7 F7 F7 T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) OPS);
        } else if (KIND == 4) {  // 64 fma (baseline)
            asm volatile(F8 F8 F8 F8 F8 F8 F8 F8 OPS);
        } else if (KIND == 5) {  // pair-kernel mix: (19 fma, sqrt, rcp) x 3 + 1 fma  = 64
            asm volatile(F7 F7 F(0) F(1) F(2) F(3) F(4) S(7) R(6) F7 F7 F(0) F(1) F(2) F(3) F(4) S(7) R(6)
                         F7 F7 F(0) F(1) F(2) F(3) F(4) S(7) R(6) F(0) OPS);
        } else if (KIND == 6) {  // same mix, transcendentals grouped by six: 57 fma, 6 trans, 1 fma
            asm volatile(F7 F7 F7 F7 F7 F7 F7 F8 S(7) S(6) S(5) R(4) R(3) R(2) F(0) OPS);
        } else if (KIND == 7) {  // 64 rsq
            asm volatile(T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
                         T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
                         T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
                         T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7) OPS);
        }
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
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
        // ns per 64-instruction group per SIMD (all waves of the SIMD together issue wps groups per iteration)
        double ns_group = ms * 1e6 / ((double)iters * wps);
        printf("%-22s wps=%d  %8.3f ms   %.2f ns per 64-instr group per SIMD  (= %.1f cycles at 2.3 GHz)\n", name, wps, ms, ns_group, ns_group * 2.3);
    }
    (void)hipFree(out);
}

int main() {
    run<4>("64 fma");
    run<7>("64 rsq");
    run<0>("8x(7 fma,1 rsq)");
    run<1>("4x(14 fma,2 rsq)");
    run<2>("2x(28 fma,4 rsq)");
    run<3>("56 fma, 8 rsq");
    run<5>("3x(19 fma,sqrt,rcp)+1");
    run<6>("58 fma, 3 sqrt+3 rcp");
    return 0;
}
