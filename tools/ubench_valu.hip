// ubench_valu.hip - VALU issue-rate microbenchmark for gfx950 (design input for pair_kernel.h).
// Measures wave-instructions per cycle per SIMD for v_fma_f32, v_pk_fma_f32, v_rsq_f32,
// v_rcp_f32, v_mov_b32 and a broadcast ds_read_b128 + fma mix, at 1/2/4/8 waves per SIMD.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o /tmp/ubench && /tmp/ubench
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef float f2 __attribute__((ext_vector_type(2)));

#define REP8(x) x x x x x x x x

template <int KIND>
__global__ void bench(float* out, int iters, long long* cyc) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1, a2 = a0 + 2, a3 = a0 + 3, a4 = a0 + 4, a5 = a0 + 5, a6 = a0 + 6, a7 = a0 + 7;
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, p4 = {a1, a0}, p5 = {a3, a2}, p6 = {a5, a4}, p7 = {a7, a6};
    const float m = 0.999f, c = 1e-3f;
    const f2 pm = {m, m}, pc = {c, c};
    __shared__ float4 lds[64];
    if (threadIdx.x < 64) lds[threadIdx.x] = make_float4(m, c, m, c);
    __syncthreads();
    long long t0 = clock64();
    for (int i = 0; i < iters; ++i) {
        if (KIND == 0) {  // 64 independent-ish v_fma_f32 (8 chains)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_fma_f32 %3, %3, %8, %9\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (KIND == 1) {  // 64 v_pk_fma_f32
            REP8(asm volatile("v_pk_fma_f32 %0, %0, %8, %9\n v_pk_fma_f32 %1, %1, %8, %9\n v_pk_fma_f32 %2, %2, %8, %9\n v_pk_fma_f32 %3, %3, %8, %9\n"
                              "v_pk_fma_f32 %4, %4, %8, %9\n v_pk_fma_f32 %5, %5, %8, %9\n v_pk_fma_f32 %6, %6, %8, %9\n v_pk_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm), "v"(pc));)
        } else if (KIND == 2) {  // 64 v_rsq_f32
            REP8(asm volatile("v_rsq_f32 %0, %0\n v_rsq_f32 %1, %1\n v_rsq_f32 %2, %2\n v_rsq_f32 %3, %3\n"
                              "v_rsq_f32 %4, %4\n v_rsq_f32 %5, %5\n v_rsq_f32 %6, %6\n v_rsq_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 3) {  // 64 v_rcp_f32
            REP8(asm volatile("v_rcp_f32 %0, %0\n v_rcp_f32 %1, %1\n v_rcp_f32 %2, %2\n v_rcp_f32 %3, %3\n"
                              "v_rcp_f32 %4, %4\n v_rcp_f32 %5, %5\n v_rcp_f32 %6, %6\n v_rcp_f32 %7, %7\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 4) {  // 64 v_mov_b32 (chain of copies between two sets)
            REP8(asm volatile("v_mov_b32 %0, %1\n v_mov_b32 %1, %2\n v_mov_b32 %2, %3\n v_mov_b32 %3, %4\n"
                              "v_mov_b32 %4, %5\n v_mov_b32 %5, %6\n v_mov_b32 %6, %7\n v_mov_b32 %7, %0\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));)
        } else if (KIND == 5) {  // mix: 7 fma : 1 rsq (64 instr)
            REP8(asm volatile("v_fma_f32 %0, %0, %8, %9\n v_fma_f32 %1, %1, %8, %9\n v_fma_f32 %2, %2, %8, %9\n v_rsq_f32 %3, %3\n"
                              "v_fma_f32 %4, %4, %8, %9\n v_fma_f32 %5, %5, %8, %9\n v_fma_f32 %6, %6, %8, %9\n v_fma_f32 %7, %7, %8, %9\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(m), "v"(c));)
        } else if (KIND == 6) {  // 64 v_pk_mul_f32
            REP8(asm volatile("v_pk_mul_f32 %0, %0, %8\n v_pk_mul_f32 %1, %1, %8\n v_pk_mul_f32 %2, %2, %8\n v_pk_mul_f32 %3, %3, %8\n"
                              "v_pk_mul_f32 %4, %4, %8\n v_pk_mul_f32 %5, %5, %8\n v_pk_mul_f32 %6, %6, %8\n v_pk_mul_f32 %7, %7, %8\n"
                              : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3), "+v"(p4), "+v"(p5), "+v"(p6), "+v"(p7) : "v"(pm));)
        } else if (KIND == 7) {  // 48 fma + 8 broadcast ds_read_b128 + 8 broadcast ds_read_b64 (pair-kernel-like ratio)
            float4 q; float2 r;
            REP8(asm volatile("ds_read_b128 %8, %12\n ds_read_b64 %9, %12 offset:16\n"
                              "v_fma_f32 %0, %0, %10, %11\n v_fma_f32 %1, %1, %10, %11\n v_fma_f32 %2, %2, %10, %11\n"
                              "v_fma_f32 %3, %3, %10, %11\n v_fma_f32 %4, %4, %10, %11\n v_fma_f32 %5, %5, %10, %11\n"
                              "s_waitcnt lgkmcnt(0)\n"
                              : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7), "=&v"(q), "=&v"(r)
                              : "v"(m), "v"(c), "v"(0));
                 a6 += q.x; a7 += r.x;)
        }
    }
    long long t1 = clock64();
    float s = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p0.y + p1.x + p1.y + p2.x + p2.y + p3.x + p3.y + p4.x + p5.x + p6.x + p7.y;
    if (s == 123.456f) out[0] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
}

template <int KIND>
static void run(const char* name, int instr_per_iter) {
    float* out; long long* cyc;
    hipMalloc(&out, 4); hipMalloc(&cyc, 8);
    const int iters = 20000;
    for (int wps = 1; wps <= 8; wps *= 2) {
        const int threads = 64 * 4 * wps;  // one block per CU, wps waves per SIMD
        if (threads > 1024) {  // 8 waves/SIMD = 2 blocks of 1024
            hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
            bench<KIND><<<512, 1024>>>(out, iters, cyc);
            hipEventRecord(e0);
            bench<KIND><<<512, 1024>>>(out, iters, cyc);
            hipEventRecord(e1); hipEventSynchronize(e1);
            float ms; hipEventElapsedTime(&ms, e0, e1);
            long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            double wave_instr = (double)iters * instr_per_iter * 8;  // per SIMD
            printf("%-14s wps=%d  %.3f ms  clock64=%lld  instr/cyc/SIMD(clock64)=%.3f  eff.clk=%.2f GHz  Ginstr/s/SIMD=%.2f\n", name, wps, ms, c,
                   wave_instr / (double)c, (double)c / (ms * 1e6), wave_instr / (ms * 1e6));
            continue;
        }
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        bench<KIND><<<256, threads>>>(out, iters, cyc);
        hipEventRecord(e0);
        bench<KIND><<<256, threads>>>(out, iters, cyc);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        long long c; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
        double wave_instr = (double)iters * instr_per_iter * wps;  // per SIMD
        printf("%-14s wps=%d  %.3f ms  clock64=%lld  instr/cyc/SIMD(clock64)=%.3f  eff.clk=%.2f GHz  Ginstr/s/SIMD=%.2f\n", name, wps, ms, c,
               wave_instr / (double)c, (double)c / (ms * 1e6), wave_instr / (ms * 1e6));
    }
    hipFree(out); hipFree(cyc);
}

int main() {
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    printf("device %s  CUs=%d  clock=%d kHz\n", p.gcnArchName, p.multiProcessorCount, p.clockRate);
    run<0>("v_fma_f32", 64);
    run<1>("v_pk_fma_f32", 64);
    run<6>("v_pk_mul_f32", 64);
    run<2>("v_rsq_f32", 64);
    run<3>("v_rcp_f32", 64);
    run<4>("v_mov_b32", 64);
    run<5>("7fma:1rsq", 64);
    run<7>("6fma+2dsread", 48);
    return 0;
}
