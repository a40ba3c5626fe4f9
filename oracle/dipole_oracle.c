/* dipole_oracle.c - CPU oracle (fp64 arbiter) for the dipole field / potential.
 * TEST INFRASTRUCTURE, NOT PRODUCT CODE: only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this; the product path never links or calls it.
 *
 * Plain-C restatement of the reference leaf arithmetic (field_utils.py:96-109 for the field,
 * field_utils.py:46-51 for the potential), evaluated in double precision pair by pair with one
 * OpenMP thread per target row, so that full-size clouds (10^4..10^5 points) can be checked in
 * seconds.  Parity status: PINNED - tests/test_oracle_golden.py compares it with the fp64 runs of
 * the reference stored in tests/golden/ (G1, G2, G5, G11, G12).
 *
 * The Inf/NaN -> 0 leaf filter of the reference (field_utils.py:110-115) is applied per source
 * leaf [leaf_off[l], leaf_off[l+1]) exactly as the recursion would.
 *
 *   gcc -O2 -fopenmp -shared -fPIC oracle/dipole_oracle.c -o oracle/libdipole_oracle.so -lm
 */
#include <math.h>
#include <stdint.h>

/* src: [S, lds] rows (x,y,z,px,py,pz) ; tgt: [T, ldt] rows (x,y,z,..) ; out: [T,3] */
void oracle_field_grad_f64(const double* src, int64_t S, int64_t lds, const double* tgt, int64_t T, int64_t ldt,
                           double eps, const int64_t* leaf_off, int64_t n_leaves, double* out) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < T; ++t) {
        const double tx = tgt[t * ldt + 0], ty = tgt[t * ldt + 1], tz = tgt[t * ldt + 2];
        double tot[3] = {0.0, 0.0, 0.0};
        for (int64_t l = 0; l < n_leaves; ++l) {
            double e[3] = {0.0, 0.0, 0.0};
            for (int64_t s = leaf_off[l]; s < leaf_off[l + 1]; ++s) {
                const double* r = src + s * lds;
                const double dx = r[0] - tx, dy = r[1] - ty, dz = r[2] - tz;   /* R = x_s - x_t */
                const double dist = sqrt(dx * dx + dy * dy + dz * dz);
                double cx = 0.0, cy = 0.0, cz = 0.0;
                if (dist != 0.0) {                                             /* zero_mask */
                    const double ux = dx / dist, uy = dy / dist, uz = dz / dist;
                    const double along = 3.0 * (r[3] * ux + r[4] * uy + r[5] * uz);
                    cx = along * ux - r[3]; cy = along * uy - r[4]; cz = along * uz - r[5];
                }
                const double den = dist * dist * dist + eps;
                e[0] += cx / den; e[1] += cy / den; e[2] += cz / den;
            }
            for (int c = 0; c < 3; ++c) {
                double v = -e[c];
                if (isnan(v) || isinf(v)) v = 0.0;
                tot[c] += v;
            }
        }
        out[t * 3 + 0] = tot[0]; out[t * 3 + 1] = tot[1]; out[t * 3 + 2] = tot[2];
    }
}

void oracle_potential_f64(const double* src, int64_t S, int64_t lds, const double* tgt, int64_t T, int64_t ldt,
                          const int64_t* leaf_off, int64_t n_leaves, double* out) {
#pragma omp parallel for schedule(static)
    for (int64_t t = 0; t < T; ++t) {
        const double tx = tgt[t * ldt + 0], ty = tgt[t * ldt + 1], tz = tgt[t * ldt + 2];
        double tot = 0.0;
        for (int64_t l = 0; l < n_leaves; ++l) {
            double phi = 0.0;
            for (int64_t s = leaf_off[l]; s < leaf_off[l + 1]; ++s) {
                const double* r = src + s * lds;
                const double dx = r[0] - tx, dy = r[1] - ty, dz = r[2] - tz;
                const double dist = sqrt(dx * dx + dy * dy + dz * dz);
                phi += (r[3] * dx + r[4] * dy + r[5] * dz) / (dist * dist * dist);   /* 0/0 -> NaN */
            }
            if (isnan(phi) || isinf(phi)) phi = 0.0;
            tot += phi;
        }
        out[t] = tot;
    }
}
