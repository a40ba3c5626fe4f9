// ubench_dispatch.hip - how fast does the MI355X start workgroups?  (Round 4: the split tail through 2-wavefront workgroups
// cost +24 %, through 4-wavefront workgroups +5 %, with the same wavefronts doing the same work - the dispatcher, not the
// arithmetic.)  Launches W wavefronts in workgroups of B threads, every wavefront busy for ~`work` fma-loop iterations, and
// prints the launch time by HIP events (median of 20) for B = 64 .. 1024: with work = 0 the time IS the dispatch time.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_dispatch.hip -o tools/bin/ubench_dispatch && tools/bin/ubench_dispatch
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <algorithm>
#include <vector>

__global__ void busy(float* out, int iters) {
    float a = threadIdx.x * 1e-3f, b = 1.0001f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 64; ++u) a = __builtin_fmaf(a, b, 1e-7f);
    }
    if (a == 123.456f) out[0] = a;      // never true: keeps the loop
}

int main() {
    float* d;
    (void)hipMalloc(&d, 4);
    hipEvent_t e0, e1;
    (void)hipEventCreate(&e0);
    (void)hipEventCreate(&e1);
    const long waves_list[] = {8192, 32768, 131072};
    const int iters_list[] = {0, 16, 64, 256};          // x 64 fma per lane: 0, ~1k, ~4k, ~16k VALU instructions per wavefront
    for (int it = 0; it < 4; ++it)
        for (int wi = 0; wi < 3; ++wi) {
            const long waves = waves_list[wi];
            printf("iters %3d (%5d fma / wavefront), %6ld wavefronts:", iters_list[it], iters_list[it] * 64, waves);
            for (int B = 64; B <= 1024; B *= 2) {
                const long wgs = waves * 64 / B;
                for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(busy, dim3((unsigned)wgs), dim3(B), 0, 0, d, iters_list[it]);
                (void)hipDeviceSynchronize();
                std::vector<float> ts;
                for (int r = 0; r < 20; ++r) {
                    (void)hipEventRecord(e0, 0);
                    hipLaunchKernelGGL(busy, dim3((unsigned)wgs), dim3(B), 0, 0, d, iters_list[it]);
                    (void)hipEventRecord(e1, 0);
                    (void)hipEventSynchronize(e1);
                    float ms;
                    (void)hipEventElapsedTime(&ms, e0, e1);
                    ts.push_back(ms * 1e3f);
                }
                std::sort(ts.begin(), ts.end());
                printf("  B=%4d %8.1f us (%5.1f WG/us)", B, ts[10], wgs / ts[10]);
            }
            printf("\n");
        }
    return 0;
}
