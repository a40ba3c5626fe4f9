// count_far_forms.hip - instruction count of the far-field pair chain in two algebraic forms (compile only:
//   hipcc -O3 --offload-arch=gfx950 -fno-slp-vectorize -S tools/count_far_forms.hip -o - | tools/count via the
//   script in DESIGN.md section 4; no GPU needed).  Both kernels: KT = 2 targets per lane in registers, sources from a
//   wave-uniform address (scalar loads, SGPR operands), two accumulator sets, as pair_kernel_scalar's far loop.
//   form_direct    r = s - t;  d2 = r.r;  pr = p.r;  u = rsq(d2) ... ; A += a r;  B += w p          (shipped)
//   form_expanded  sources pre-centred on the patch (s' = s - c, per-source |s'|^2 and p.s' precomputed, row of 8 floats),
//                  targets centred per (wave, chunk): d2 = |s'|^2 + |t'|^2 - 2 s'.t',  pr = p.s' - p.t',
//                  sum a r = sum a s' - t' sum a   (VERDICT r02 item 5 / DESIGN r02 "distance by expansion")
#include <hip/hip_runtime.h>

__device__ __forceinline__ void tail(float d2, float pr, float eps, float& w, float& a) {
    const float u = __builtin_amdgcn_rsqf(d2);
    const float u2 = u * u, u3 = u2 * u, e = eps * u3;
    w = __builtin_fmaf(u3, __builtin_fmaf(e, e, -e), u3);
    a = pr * (w * u2);
}

extern "C" __global__ __launch_bounds__(256) void form_direct(const float* __restrict__ src, int n, const float* __restrict__ tgt,
                                                              float eps, float* __restrict__ out) {
    float tx[2], ty[2], tz[2], A[2][2][3] = {}, B[2][2][3] = {};
    for (int k = 0; k < 2; ++k) { const float* t = tgt + (threadIdx.x + 64 * k) * 3; tx[k] = t[0]; ty[k] = t[1]; tz[k] = t[2]; }
    for (int s = 0; s + 2 <= n; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* p = src + (s + u) * 6;
            const float sx = p[0], sy = p[1], sz = p[2], px = p[3], py = p[4], pz = p[5];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float rx = sx - tx[k], ry = sy - ty[k], rz = sz - tz[k];
                const float d2 = __builtin_fmaf(rz, rz, __builtin_fmaf(ry, ry, rx * rx));
                const float pr = __builtin_fmaf(pz, rz, __builtin_fmaf(py, ry, px * rx));
                float w, a;
                tail(d2, pr, eps, w, a);
                A[u][k][0] = __builtin_fmaf(a, rx, A[u][k][0]); A[u][k][1] = __builtin_fmaf(a, ry, A[u][k][1]); A[u][k][2] = __builtin_fmaf(a, rz, A[u][k][2]);
                B[u][k][0] = __builtin_fmaf(w, px, B[u][k][0]); B[u][k][1] = __builtin_fmaf(w, py, B[u][k][1]); B[u][k][2] = __builtin_fmaf(w, pz, B[u][k][2]);
            }
        }
    }
    for (int k = 0; k < 2; ++k)
        for (int c = 0; c < 3; ++c) out[(threadIdx.x + 64 * k) * 3 + c] = 3.f * (A[0][k][c] + A[1][k][c]) - (B[0][k][c] + B[1][k][c]);
}

extern "C" __global__ __launch_bounds__(256) void form_expanded(const float* __restrict__ src8, int n, const float* __restrict__ tgt,
                                                                float eps, float* __restrict__ out) {
    // src8 row: (-2 s'x, -2 s'y, -2 s'z, |s'|^2, px, py, pz, p.s');  tgt already centred: (t'x, t'y, t'z), t2 = |t'|^2
    float tx[2], ty[2], tz[2], t2[2], A[2][2][3] = {}, B[2][2][3] = {}, SA[2][2] = {};
    for (int k = 0; k < 2; ++k) {
        const float* t = tgt + (threadIdx.x + 64 * k) * 3; tx[k] = t[0]; ty[k] = t[1]; tz[k] = t[2];
        t2[k] = tx[k] * tx[k] + ty[k] * ty[k] + tz[k] * tz[k];
    }
    for (int s = 0; s + 2 <= n; s += 2) {
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const float* p = src8 + (s + u) * 8;
            const float mx = p[0], my = p[1], mz = p[2], s2 = p[3], px = p[4], py = p[5], pz = p[6], ps = p[7];
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                const float d2 = __builtin_fmaf(mz, tz[k], __builtin_fmaf(my, ty[k], __builtin_fmaf(mx, tx[k], s2 + t2[k])));
                const float pr = ps - __builtin_fmaf(pz, tz[k], __builtin_fmaf(py, ty[k], px * tx[k]));
                float w, a;
                tail(d2, pr, eps, w, a);
                A[u][k][0] = __builtin_fmaf(a, mx, A[u][k][0]); A[u][k][1] = __builtin_fmaf(a, my, A[u][k][1]); A[u][k][2] = __builtin_fmaf(a, mz, A[u][k][2]);
                SA[u][k] += a;
                B[u][k][0] = __builtin_fmaf(w, px, B[u][k][0]); B[u][k][1] = __builtin_fmaf(w, py, B[u][k][1]); B[u][k][2] = __builtin_fmaf(w, pz, B[u][k][2]);
            }
        }
    }
    for (int k = 0; k < 2; ++k) {
        const float t[3] = {tx[k], ty[k], tz[k]};
        for (int c = 0; c < 3; ++c)     // sum a r = -0.5 sum a (-2 s') - t' sum a
            out[(threadIdx.x + 64 * k) * 3 + c] = 3.f * (-0.5f * (A[0][k][c] + A[1][k][c]) - t[c] * (SA[0][k] + SA[1][k])) - (B[0][k][c] + B[1][k][c]);
    }
}
