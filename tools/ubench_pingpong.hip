// ubench_pingpong.hip - latency of an 8-byte {data, tag} hand-off between two workgroups on the SAME XCD (plain store,
// sc1 load: the line stays in the XCD's L2) against two workgroups on DIFFERENT XCDs (sc1 store, sc1 load), placement
// verified with HW_REG_XCC_ID.  What a same-XCD placement of the per-point kernel's all-gather could buy.
//   hipcc -O3 --offload-arch=gfx950 tools/ubench_pingpong.hip -o tools/bin/ubench_pingpong && tools/bin/ubench_pingpong
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__device__ __forceinline__ unsigned xcc_id() {
    unsigned x;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
    return x & 0xf;
}

// roles: the first workgroup that finds the wanted XCD for role 0 / role 1 takes it; everybody else exits
template <bool SAME>
__global__ __launch_bounds__(64) void pingpong(unsigned long long* slots, int* roles, int rounds, long long* cycles) {
    __shared__ int role_s;
    if (threadIdx.x == 0) {
        const unsigned x = xcc_id();
        int role = -1;
        if (SAME) {
            if (x == 0) role = atomicAdd(&roles[0], 1);                 // two workgroups of XCD 0
        } else {
            if (x == 0) role = (atomicAdd(&roles[0], 1) == 0) ? 0 : -1;  // one of XCD 0 ...
            if (x == 1) role = (atomicAdd(&roles[1], 1) == 0) ? 1 : -1;  // ... and one of XCD 1
        }
        role_s = (role == 0 || role == 1) ? role : -1;
    }
    __syncthreads();
    const int role = role_s;
    if (role < 0 || threadIdx.x != 0) return;
    unsigned long long* mine = slots + role * 32;        // 256 B apart
    unsigned long long* theirs = slots + (1 - role) * 32;
    const long long t0 = clock64();
    unsigned spins = 0;
    for (int r = 1; r <= rounds; ++r) {
        if (role == 0) {
            if (SAME) *(volatile unsigned long long*)mine = (unsigned long long)r;
            else __hip_atomic_store(mine, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)r)
                if (++spins > (1u << 28)) return;
        } else {
            while (__hip_atomic_load(theirs, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) != (unsigned long long)r)
                if (++spins > (1u << 28)) return;
            if (SAME) *(volatile unsigned long long*)mine = (unsigned long long)r;
            else __hip_atomic_store(mine, (unsigned long long)r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    if (role == 0) cycles[0] = clock64() - t0;
}

int main() {
    unsigned long long* slots; int* roles; long long* cyc;
    hipMalloc(&slots, 4096); hipMalloc(&roles, 64); hipMalloc(&cyc, 8);
    const int rounds = 200000;
    for (int mode = 0; mode < 2; ++mode) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(slots, 0, 4096); hipMemset(roles, 0, 64); hipMemset(cyc, 0, 8);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            if (mode == 0) hipLaunchKernelGGL(pingpong<true>, dim3(64), dim3(64), 0, 0, slots, roles, rounds, cyc);
            else hipLaunchKernelGGL(pingpong<false>, dim3(64), dim3(64), 0, 0, slots, roles, rounds, cyc);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            long long c = 0; hipMemcpy(&c, cyc, 8, hipMemcpyDeviceToHost);
            printf("%s  round trip %.3f us (wall %.1f ms / %d rounds), %lld clock64 ticks per round\n",
                   mode == 0 ? "same XCD, plain store + sc1 load " : "two XCDs, sc1 store + sc1 load  ", ms * 1e3 / rounds, ms, rounds,
                   c / rounds);
        }
    }
    return 0;
}
