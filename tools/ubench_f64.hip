// ubench_f64.hip - fp64 vector issue rates on gfx950 (design input for the fp64 pair chain, round 5).
// Cycles per wave64 instruction per SIMD for v_fma_f64 / v_mul_f64 / v_add_f64, the fp64 transcendentals
// (v_rsq_f64, v_rcp_f64, v_sqrt_f64) back to back, and the mixes the fp64 pair chain would issue
// (28 fma-class + rsq + rcp per pair), at 2 / 4 / 8 wavefronts per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_f64.hip -o /tmp/ubench_f64 && /tmp/ubench_f64
#include <hip/hip_runtime.h>
#include <stdio.h>

#define F(i) "v_fma_f64 %" #i ", %" #i ", %8, %9\n"
#define M(i) "v_mul_f64 %" #i ", %" #i ", %8\n"
#define A(i) "v_add_f64 %" #i ", %" #i ", %9\n"
#define T(i) "v_rsq_f64 %" #i ", %" #i "\n"
#define R(i) "v_rcp_f64 %" #i ", %" #i "\n"
#define S(i) "v_sqrt_f64 %" #i ", %" #i "\n"
#define F7 F(0) F(1) F(2) F(3) F(4) F(5) F(6)
#define F8 F(0) F(1) F(2) F(3) F(4) F(5) F(6) F(7)
#define M8 M(0) M(1) M(2) M(3) M(4) M(5) M(6) M(7)
#define A8 A(0) A(1) A(2) A(3) A(4) A(5) A(6) A(7)
#define T8 T(0) T(1) T(2) T(3) T(4) T(5) T(6) T(7)
#define R8 R(0) R(1) R(2) R(3) R(4) R(5) R(6) R(7)
#define S8 S(0) S(1) S(2) S(3) S(4) S(5) S(6) S(7)
#define OPS : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c)

template <int KIND>
__global__ void bench(double* out, int iters) {
    double a0 = threadIdx.x * 1e-3 + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    const double m = 0.999, c = 1e-3;
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) asm volatile(F8 F8 F8 F8 F8 F8 F8 F8 OPS);                 // 64 fma
        else if (KIND == 1) asm volatile(M8 M8 M8 M8 M8 M8 M8 M8 OPS);            // 64 mul
        else if (KIND == 2) asm volatile(A8 A8 A8 A8 A8 A8 A8 A8 OPS);            // 64 add
        else if (KIND == 3) asm volatile(T8 T8 T8 T8 T8 T8 T8 T8 OPS);            // 64 rsq
        else if (KIND == 4) asm volatile(R8 R8 R8 R8 R8 R8 R8 R8 OPS);            // 64 rcp
        else if (KIND == 5) asm volatile(S8 S8 S8 S8 S8 S8 S8 S8 OPS);            // 64 sqrt
        else if (KIND == 6)                                                       // pair-chain mix: 2 x (28 fma, rsq, rcp) + 4 fma = 64
            asm volatile(F7 F7 F7 F7 T(7) R(6) F7 F7 F7 F7 T(7) R(6) F(0) F(1) F(2) F(3) OPS);
        else if (KIND == 7)                                                       // 56 fma + 8 rsq, spread
            asm volatile(F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) F7 T(7) OPS);
    }
    double s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7;
    if (s == 123.456) out[0] = s;
}

// fp32 instructions beside fp64 ones: do the 32-bit integer / fp32 ops a fp64 chain needs (v_cndmask, address math) issue at
// the fp32 rate between fp64 fmas?  32 v_fma_f64 + 32 v_fma_f32
__global__ void bench_mixed32(double* out, int iters) {
    double a0 = threadIdx.x * 1e-3 + 1, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3;
    float b0 = threadIdx.x * 1e-3f + 1, b1 = b0 + 1, b2 = b0 + 2, b3 = b0 + 3;
    const double m = 0.999, c = 1e-3;
    const float mf = 0.999f, cf = 1e-3f;
    for (int i = 0; i < iters; ++i) {
#define X "v_fma_f64 %0, %0, %8, %9\n v_fma_f32 %4, %4, %10, %11\n v_fma_f64 %1, %1, %8, %9\n v_fma_f32 %5, %5, %10, %11\n" \
          "v_fma_f64 %2, %2, %8, %9\n v_fma_f32 %6, %6, %10, %11\n v_fma_f64 %3, %3, %8, %9\n v_fma_f32 %7, %7, %10, %11\n"
        asm volatile(X X X X X X X X : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(b0), "+v"(b1), "+v"(b2), "+v"(b3)
                     : "v"(m), "v"(c), "v"(mf), "v"(cf));
#undef X
    }
    double s = a0 + a1 + a2 + a3 + (double)(b0 + b1 + b2 + b3);
    if (s == 123.456) out[0] = s;
}

template <typename K>
static void run_kernel(const char* name, K kernel) {
    double* out; (void)hipMalloc(&out, 8);
    const int iters = 10000;
    for (int wps = 2; wps <= 8; wps *= 2) {
        const int blocks = wps == 8 ? 512 : 256, threads = wps == 8 ? 1024 : 64 * 4 * wps;
        hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
        kernel<<<blocks, threads>>>(out, iters);
        (void)hipEventRecord(e0);
        kernel<<<blocks, threads>>>(out, iters);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1);
        const double ns_group = ms * 1e6 / ((double)iters * wps);    // per 64-instruction group per SIMD
        printf("%-26s wps=%d  %8.3f ms   %7.2f ns per 64-instr group per SIMD  = %6.2f cycles per instruction at 2.4 GHz\n", name, wps, ms,
               ns_group, ns_group * 2.4 / 64.0);
    }
    (void)hipFree(out);
}

int main() {
    hipDeviceProp_t p; (void)hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs=%d  clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run_kernel("v_fma_f64", bench<0>);
    run_kernel("v_mul_f64", bench<1>);
    run_kernel("v_add_f64", bench<2>);
    run_kernel("v_rsq_f64", bench<3>);
    run_kernel("v_rcp_f64", bench<4>);
    run_kernel("v_sqrt_f64", bench<5>);
    run_kernel("2x(28fma,rsq,rcp)+4fma", bench<6>);
    run_kernel("8x(7fma,rsq)", bench<7>);
    run_kernel("32 fma_f64 + 32 fma_f32", bench_mixed32);
    printf("peak check: 256 CUs x 4 SIMDs x 64 lanes x 2 flop x 2.4 GHz / (cycles per v_fma_f64) = FP64 vector TFLOP/s; 78.6 T <=> 4.0 cycles\n");
    return 0;
}
