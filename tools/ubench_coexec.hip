// ubench_coexec.hip - do transcendental instructions of one wave overlap with FMAs of another wave on the same
// SIMD?  8 waves per SIMD; in the "split" run the even waves execute only v_fma_f32 and the odd waves only
// v_rsq_f32 (same instruction counts per wave as in the pure runs with 4 waves per SIMD).
// overlap  -> t(split) ~ max(t_fma4, t_rsq4);   serialised -> t(split) ~ t_fma4 + t_rsq4
#include <hip/hip_runtime.h>
#include <stdio.h>
#define F(i) "v_fma_f32 %" #i ", %" #i ", %8, %9\n"
#define T(i) "v_rsq_f32 %" #i ", %" #i "\n"
#define G8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)
// mode 0: every wave fma; 1: every wave rsq; 2: even waves fma, odd waves rsq (wave index within the SIMD)
__global__ void bench(float* out, int iters, int mode) {
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f, c = 1e-3f;
    const int wave = threadIdx.x >> 6;                 // waves go to SIMDs round-robin: wave>>2 = index on its SIMD
    const bool do_rsq = (mode == 1) || (mode == 2 && ((wave >> 2) & 1));
    if (do_rsq) {
        for (int i = 0; i < iters; ++i) asm volatile(G8(T) G8(T) G8(T) G8(T) G8(T) G8(T) G8(T) G8(T) OPS);
    } else {
        for (int i = 0; i < iters; ++i) asm volatile(G8(F) G8(F) G8(F) G8(F) G8(F) G8(F) G8(F) G8(F) OPS);
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456f) out[0] = s;
}
static float run(int threads, int blocks, int mode, int iters) {
    float* out; (void)hipMalloc(&out, 4);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    bench<<<blocks, threads>>>(out, iters, mode);
    (void)hipEventRecord(e0);
    bench<<<blocks, threads>>>(out, iters, mode);
    (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipFree(out);
    return ms;
}
int main() {
    const int iters = 20000;
    // 4 waves per SIMD (1024 threads per block, 1 block per CU would be 4 waves/SIMD) - pure runs
    float f4 = run(1024, 256, 0, iters), r4 = run(1024, 256, 1, iters);
    // 8 waves per SIMD (2 blocks per CU): pure and split
    float f8 = run(1024, 512, 0, iters), r8 = run(1024, 512, 1, iters), s8 = run(1024, 512, 2, iters);
    printf("4 waves/SIMD: all-fma %.3f ms, all-rsq %.3f ms\n", f4, r4);
    printf("8 waves/SIMD: all-fma %.3f ms, all-rsq %.3f ms, split (4 fma + 4 rsq waves) %.3f ms\n", f8, r8, s8);
    printf("split vs sum of the 4-wave pure runs: %.3f vs %.3f (serialised) or %.3f (perfect overlap)\n", s8, f4 + r4, r4 > f4 ? r4 : f4);
    return 0;
}
