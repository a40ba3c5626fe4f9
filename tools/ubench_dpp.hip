// ubench_dpp.hip - issue rate of DPP-modified VOP2 ops (row_ror) vs plain VOP2 on gfx950, and of LDS f64 atomics.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define P(i) "v_fmac_f32_e32 %" #i ", %8, %9\n"
#define D(i) "v_fmac_f32_dpp %" #i ", %8, %9 row_ror:1 row_mask:0xf bank_mask:0xf\n"
#define A(i) "v_add_f32_dpp %" #i ", %8, %9 row_ror:3 row_mask:0xf bank_mask:0xf\n"
#define S(i) "v_sub_f32_dpp %" #i ", %8, %9 row_ror:5 row_mask:0xf bank_mask:0xf\n"
#define G8(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)
template <int KIND> __global__ void bench(float* out, int iters) {
    float a0 = threadIdx.x * 1e-3f + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const float m = 0.999f + threadIdx.x * 1e-6f, c = 1e-3f;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(G8(P) G8(P) G8(P) G8(P) G8(P) G8(P) G8(P) G8(P) OPS);
        else if (KIND == 1) asm volatile(G8(D) G8(D) G8(D) G8(D) G8(D) G8(D) G8(D) G8(D) OPS);
        else if (KIND == 2) asm volatile(G8(A) G8(A) G8(A) G8(A) G8(A) G8(A) G8(A) G8(A) OPS);
        else if (KIND == 3) asm volatile(G8(P) G8(D) G8(P) G8(S) G8(P) G8(A) G8(P) G8(D) OPS);   // half DPP
    }
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456f) out[0] = s;
}
template <int KIND> static void run(const char* name) {
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
        double ns_group = ms * 1e6 / ((double)iters * wps);
        printf("%-26s wps=%d  %8.3f ms   %.2f ns per 64-instr group per SIMD (%.2f cycles/instr at 2.3 GHz)\n", name, wps, ms, ns_group, ns_group * 2.3 / 64);
    }
}
int main() {
    run<0>("64 v_fmac_f32");
    run<1>("64 v_fmac_f32_dpp row_ror");
    run<2>("64 v_add_f32_dpp row_ror");
    run<3>("32 plain + 32 dpp");
    return 0;
}
