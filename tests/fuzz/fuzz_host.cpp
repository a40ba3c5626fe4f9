// AddressSanitizer / UBSan fuzz of the HOST functions of libdnp (CPU build of csrc/dnp_io.hip and csrc/dnp_prep.hip;
// GPU sanitizers are not available on the pool).  Built and run by tests/test_host_sanitized.py.
//   dnp_xyz_parse_f32 / dnp_xyz_format_f32: random bytes, random number-alphabet text and mutated well-formed rows in
//     exact-size heap buffers (an over-read trips ASan); every accepted text must survive format -> parse unchanged.
//   dnp_merge_cells: random voxel sets against a brute-force restatement of the merge rule (include/dnp.h); bad tables
//     (coordinates outside [0, 2^20), NULL pointers, negative counts) are refused with DNP_EINVAL.
//   the extern "C" contract (round 4): with operator new made to FAIL after a random number of allocations, dnp_merge_cells
//     and the launch planner behind dnp_field_grad_workspace_bytes / dnp_potential_workspace_bytes return DNP_ENOMEM / 0 -
//     no exception leaves the library; the planner is also run over random shapes (its index arithmetic under ASan).
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <new>
#include <random>
#include <string>
#include <vector>

#include "dnp.h"

// ---- allocation-failure injection: the g_fail_in-th operator new from now on throws std::bad_alloc (0 = never) ------
static long g_fail_in = 0;
void* operator new(size_t n) {
    if (g_fail_in > 0 && --g_fail_in == 0) throw std::bad_alloc();
    void* p = malloc(n ? n : 1);
    if (!p) throw std::bad_alloc();
    return p;
}
void* operator new[](size_t n) { return operator new(n); }
void operator delete(void* p) noexcept { free(p); }
void operator delete[](void* p) noexcept { free(p); }
void operator delete(void* p, size_t) noexcept { free(p); }
void operator delete[](void* p, size_t) noexcept { free(p); }

static int fuzz_text(std::mt19937_64& rng, int iterations) {
    const char alphabet[] = "0123456789.eE+- \n\t\rnaif_x";
    long parsed = 0, rejected = 0;
    for (int it = 0; it < iterations; ++it) {
        std::string s(rng() % 200, ' ');
        const int mode = (int)(rng() % 3);
        if (mode == 0) {
            for (auto& c : s) c = alphabet[rng() % (sizeof(alphabet) - 1)];
        } else if (mode == 1) {
            for (auto& c : s) c = (char)(rng() & 0xff);
        } else {
            s.clear();
            const int rows = (int)(rng() % 6), cols = (rng() & 1) ? 3 : 6;
            for (int r = 0; r < rows; ++r) {
                for (int c = 0; c < cols; ++c) {
                    char b[40];
                    snprintf(b, sizeof b, "%.9g", (double)(int64_t)(rng() % 2000000) / 1e3 - 1e3);
                    s += b;
                    if (c + 1 < cols) s += " ";
                }
                s += "\n";
            }
            if (!s.empty() && (rng() & 3) == 0) s[rng() % s.size()] = alphabet[rng() % (sizeof(alphabet) - 1)];
        }
        char* txt = (char*)malloc(s.size() ? s.size() : 1);
        memcpy(txt, s.data(), s.size());
        int32_t ncol = 0;
        const int64_t max_rows = (int64_t)(rng() % 8);
        float* out = (float*)malloc(sizeof(float) * 6 * (size_t)(max_rows ? max_rows : 1));
        const int64_t n = dnp_xyz_parse_f32(txt, (int64_t)s.size(), out, max_rows, &ncol);
        if (n > 0) {
            ++parsed;
            const int64_t cap = dnp_xyz_format_bound(n, ncol);
            char* o = (char*)malloc((size_t)cap);
            const int64_t w = dnp_xyz_format_f32(out, n, ncol, o, cap);
            if (w < 0 || w > cap) { printf("format wrote %lld of %lld\n", (long long)w, (long long)cap); return 1; }
            std::vector<float> back((size_t)(n * ncol));
            int32_t nc2 = 0;
            const int64_t n2 = dnp_xyz_parse_f32(o, w, back.data(), n, &nc2);
            if (n2 != n || nc2 != ncol || memcmp(back.data(), out, sizeof(float) * (size_t)(n * ncol)) != 0) {
                printf("format -> parse changed the values\n");
                return 1;
            }
            free(o);
        } else {
            ++rejected;
        }
        free(out);
        free(txt);
    }
    std::vector<float> ext;                                   // every bit pattern class through the formatter
    for (int i = 0; i < 40000; ++i) {
        const uint32_t b = (uint32_t)rng();
        float f;
        memcpy(&f, &b, 4);
        ext.push_back(f);
    }
    const int64_t cap = dnp_xyz_format_bound((int64_t)ext.size() / 4, 4);
    char* o = (char*)malloc((size_t)cap);
    const int64_t w = dnp_xyz_format_f32(ext.data(), (int64_t)ext.size() / 4, 4, o, cap);
    free(o);
    printf("text: %ld accepted, %ld rejected / empty; %lld of %lld bytes for random bit patterns\n", parsed, rejected,
           (long long)w, (long long)cap);
    if (!(w > 0 && w <= cap)) return 1;
    // texts long enough for the threaded paths (blocks of lines on several host threads): format -> parse round trip with
    // blank lines and CRLF sprinkled in, a row-capacity error, and an irregular line deep inside a late block
    for (int round = 0; round < 6; ++round) {
        const int cols = (round & 1) ? 3 : 6;
        const int64_t rows = 20000 + (int64_t)(rng() % 30000);
        std::vector<float> v((size_t)(rows * cols));
        for (auto& x : v) x = (float)((double)(int64_t)(rng() % 2000000) / 1e3 - 1e3) * ((rng() & 7) ? 1.f : 1e-9f);
        const int64_t c2 = dnp_xyz_format_bound(rows, cols);
        std::vector<char> txt((size_t)c2);
        const int64_t w2 = dnp_xyz_format_f32(v.data(), rows, cols, txt.data(), c2);
        if (w2 <= 0 || w2 > c2) { printf("long format wrote %lld of %lld\n", (long long)w2, (long long)c2); return 1; }
        std::string loose;
        for (int64_t i = 0; i < w2; ++i) {
            if (txt[(size_t)i] == '\n' && (rng() % 50) == 0) loose += (rng() & 1) ? "\r\n\n" : "\n \n";
            else loose += txt[(size_t)i];
        }
        const std::string* forms[2] = {nullptr, &loose};
        for (const std::string* t : forms) {
            const char* p = t ? t->data() : txt.data();
            const int64_t len = t ? (int64_t)t->size() : w2;
            std::vector<float> back((size_t)(rows * cols));
            int32_t nc = 0;
            const int64_t n = dnp_xyz_parse_f32(p, len, back.data(), rows, &nc);
            if (n != rows || nc != cols || memcmp(back.data(), v.data(), sizeof(float) * v.size()) != 0) {
                printf("long round trip: %lld rows of %lld, %d columns\n", (long long)n, (long long)rows, nc);
                return 1;
            }
            if (dnp_xyz_parse_f32(p, len, back.data(), rows - 1, &nc) != DNP_EWORKSPACE) { printf("row capacity not enforced\n"); return 1; }
        }
        std::string bad(txt.data(), (size_t)w2);
        bad[bad.size() - 1 - (size_t)(rng() % 2000)] = 'x';
        int32_t nc = 0;
        std::vector<float> back((size_t)(rows * cols));
        if (dnp_xyz_parse_f32(bad.data(), (int64_t)bad.size(), back.data(), rows, &nc) != -2) { printf("irregular tail accepted\n"); return 1; }
    }
    return 0;
}

// brute-force statement of the merge rule: live cells own voxel lists; adjacency by scanning every voxel pair
static void merge_brute(const std::vector<int32_t>& ijk, const std::vector<int64_t>& sz, int64_t min_patch,
                        std::vector<int64_t>& seq, std::vector<int64_t>& off) {
    const int64_t C = (int64_t)sz.size();
    std::vector<std::vector<int32_t>> members((size_t)C);
    std::vector<int64_t> size(sz);
    for (int64_t c = 0; c < C; ++c) members[(size_t)c].push_back((int32_t)c);
    auto touches = [&](int32_t a, int32_t b) {
        for (int d = 0; d < 3; ++d)
            if (abs(ijk[(size_t)a * 3 + d] - ijk[(size_t)b * 3 + d]) > 1) return false;
        return true;
    };
    bool again = true;
    for (int sweep = 0; again && sweep < 10; ++sweep) {
        again = false;
        for (int64_t i = 0; i < C; ++i) {
            if (members[(size_t)i].empty() || size[(size_t)i] >= min_patch) continue;
            int64_t target = -1;
            for (int64_t j = 0; j < C; ++j) {
                if (j == i || members[(size_t)j].empty()) continue;
                bool adj = false;
                for (int32_t a : members[(size_t)i])
                    for (int32_t b : members[(size_t)j]) adj = adj || touches(a, b);
                if (adj) target = j;                          // the last match wins
            }
            if (target < 0) continue;
            for (int32_t a : members[(size_t)i]) members[(size_t)target].push_back(a);
            size[(size_t)target] += size[(size_t)i];
            size[(size_t)i] = 0;
            members[(size_t)i].clear();
            if (size[(size_t)target] < min_patch) again = true;
        }
    }
    seq.clear();
    off.assign(1, 0);
    for (int64_t i = 0; i < C; ++i) {
        if (members[(size_t)i].empty() || size[(size_t)i] < min_patch) continue;
        for (int32_t a : members[(size_t)i]) seq.push_back(a);
        off.push_back((int64_t)seq.size());
    }
}

static int fuzz_merge(std::mt19937_64& rng, int iterations) {
    for (int it = 0; it < iterations; ++it) {
        const int side = 2 + (int)(rng() % 5);
        std::vector<int32_t> ijk;
        std::vector<int64_t> sz;
        for (int i = 0; i < side; ++i)                        // (i, j, k)-lexicographic order, some cells empty
            for (int j = 0; j < side; ++j)
                for (int k = 0; k < side; ++k)
                    if (rng() % 3) {
                        ijk.insert(ijk.end(), {i, j, k});
                        sz.push_back((int64_t)(rng() % 40));
                    }
        const int64_t C = (int64_t)sz.size(), min_patch = (int64_t)(rng() % 60);
        // exact-size heap buffers
        int32_t* h_ijk = (int32_t*)malloc(sizeof(int32_t) * 3 * (size_t)(C ? C : 1));
        int64_t* h_sz = (int64_t*)malloc(sizeof(int64_t) * (size_t)(C ? C : 1));
        int64_t* h_seq = (int64_t*)malloc(sizeof(int64_t) * (size_t)(C ? C : 1));
        int64_t* h_off = (int64_t*)malloc(sizeof(int64_t) * (size_t)(C + 1));
        memcpy(h_ijk, ijk.data(), sizeof(int32_t) * ijk.size());
        memcpy(h_sz, sz.data(), sizeof(int64_t) * sz.size());
        int64_t n = -1;
        int32_t sweeps = -1;
        const int rc = dnp_merge_cells(h_ijk, h_sz, C, min_patch, h_seq, h_off, &n, &sweeps);
        std::vector<int64_t> seq, off;
        merge_brute(ijk, sz, min_patch, seq, off);
        bool ok = rc == 0 && n == (int64_t)off.size() - 1 && sweeps >= 1 && sweeps <= 10;
        for (size_t q = 0; ok && q < off.size(); ++q) ok = h_off[q] == off[q];
        for (size_t q = 0; ok && q < seq.size(); ++q) ok = h_seq[q] == seq[q];
        free(h_ijk); free(h_sz); free(h_seq); free(h_off);
        if (!ok) { printf("merge mismatch at iteration %d (C=%lld, min_patch=%lld)\n", it, (long long)C, (long long)min_patch); return 1; }
    }
    printf("merge: %d random voxel sets equal the brute-force rule\n", iterations);
    return 0;
}

static int fuzz_merge_refusals() {
    int32_t ijk[6] = {0, 0, 0, 1, 0, 0};
    int64_t sz[2] = {5, 5}, seq[2], off[3], n = -7;
    int bad = 0;
    bad += dnp_merge_cells(ijk, sz, -1, 10, seq, off, &n, nullptr) != DNP_EINVAL;
    bad += dnp_merge_cells(nullptr, sz, 2, 10, seq, off, &n, nullptr) != DNP_EINVAL;
    bad += dnp_merge_cells(ijk, sz, 2, 10, seq, nullptr, &n, nullptr) != DNP_EINVAL;
    ijk[4] = 1 << 20;
    bad += dnp_merge_cells(ijk, sz, 2, 10, seq, off, &n, nullptr) != DNP_EINVAL;
    ijk[4] = -1;
    bad += dnp_merge_cells(ijk, sz, 2, 10, seq, off, &n, nullptr) != DNP_EINVAL;
    bad += strlen(dnp_last_error()) == 0;
    ijk[4] = 0;
    bad += dnp_merge_cells(ijk, sz, 0, 10, nullptr, off, &n, nullptr) != DNP_OK || n != 0;     // an empty table is fine
    if (bad) { printf("merge: %d refusal checks failed\n", bad); return 1; }
    printf("merge: bad tables are refused\n");
    return 0;
}

static int fuzz_alloc_failures(std::mt19937_64& rng, int iterations) {
    // a fixed table of 64 cells; operator new fails at a random point of the call
    std::vector<int32_t> ijk;
    std::vector<int64_t> sz;
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 4; ++k) { ijk.insert(ijk.end(), {i, j, k}); sz.push_back(3 + (i + j + k) % 5); }
    std::vector<int64_t> seq(sz.size()), off(sz.size() + 1);
    int enomem = 0, ok = 0;
    for (int it = 0; it < iterations; ++it) {
        int64_t n = -1;
        g_fail_in = 1 + (long)(rng() % 200);
        const int rc = dnp_merge_cells(ijk.data(), sz.data(), (int64_t)sz.size(), 12, seq.data(), off.data(), &n, nullptr);
        g_fail_in = 0;
        if (rc == DNP_ENOMEM) ++enomem;
        else if (rc == DNP_OK) ++ok;
        else { printf("alloc: dnp_merge_cells returned %d under allocation failure\n", rc); return 1; }
    }
    int zero = 0, sized = 0;
    for (int it = 0; it < iterations; ++it) {
        const int64_t S = 1 + (int64_t)(rng() % 400000), T = 1 + (int64_t)(rng() % 400000);
        g_fail_in = 1 + (long)(rng() % 12);
        const size_t a = (it & 1) ? dnp_field_grad_workspace_bytes(S, T, 15000) : dnp_potential_workspace_bytes(S, T, (it & 2) ? 0 : 15000);
        g_fail_in = 0;
        if (a == 0) { ++zero; if (strlen(dnp_last_error()) == 0) { printf("alloc: no message with a 0 answer\n"); return 1; } }
        else ++sized;
    }
    if (enomem == 0 || zero == 0) { printf("alloc: the injection never hit (%d, %d)\n", enomem, zero); return 1; }
    printf("alloc: %d merges / %d planner calls answered DNP_ENOMEM / 0 under allocation failure (%d / %d completed); nothing escaped\n",
           enomem, zero, ok, sized);
    // the planner over random shapes, no failures: every answer is a plausible size
    for (int it = 0; it < iterations * 4; ++it) {
        const int64_t S = (int64_t)(rng() % 3000000), T = (int64_t)(rng() % 3000000);
        const int64_t max_pts = (rng() & 3) ? 15000 : (int64_t)(rng() % 40000) - 5;
        const size_t a = dnp_field_grad_workspace_bytes(S, T, max_pts), b = dnp_potential_workspace_bytes(S, T, max_pts);
        if (a < 256 || b < 256 || a > ((size_t)5 << 30) || b > ((size_t)5 << 30)) {
            printf("planner: workspace %zu / %zu for S=%lld T=%lld max_pts=%lld\n", a, b, (long long)S, (long long)T, (long long)max_pts);
            return 1;
        }
    }
    printf("planner: %d random shapes sized\n", iterations * 4);
    return 0;
}

int main(int argc, char** argv) {
    const int scale = argc > 1 ? atoi(argv[1]) : 1;
    std::mt19937_64 rng(1234);
    if (fuzz_text(rng, 40000 * scale)) return 1;
    if (fuzz_merge(rng, 300 * scale)) return 1;
    if (fuzz_merge_refusals()) return 1;
    if (fuzz_alloc_failures(rng, 300 * scale)) return 1;
    return 0;
}
