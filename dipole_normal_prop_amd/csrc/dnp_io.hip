// dnp_io.hip - the '.xyz' text format on the host side of the path (util.xyz2tensor util.py:53-69,
// util.export_pc util.py:46-51).  HOST functions, host pointers, no device work: with the fields at
// milliseconds, formatting 600 000 numbers with Python's str(float) (0.25 s) and parsing them back (0.15 s) was
// what an orient_large run on 100 k points spent its time on.
#include <charconv>
#include <cmath>
#include <cstring>
#include <thread>
#include <vector>

#include "dnp_common.h"

namespace dnp {

// Python's repr(float) / str(float) of a double: shortest digits that round-trip, fixed notation for
// 1e-4 <= |x| < 1e16 (with ".0" appended to an integral value), otherwise d[.ddd]e+XX with at least two exponent
// digits; "inf", "-inf", "nan".  Returns the number of characters written (no terminator).
static inline int py_repr(double v, char* out) {
    if (std::isnan(v)) { memcpy(out, "nan", 3); return 3; }
    if (std::isinf(v)) { if (v < 0) { memcpy(out, "-inf", 4); return 4; } memcpy(out, "inf", 3); return 3; }
    char sci[40];
    const auto res = std::to_chars(sci, sci + sizeof(sci), v, std::chars_format::scientific);   // [-]d[.ddd]e[+-]XX
    const char* p = sci;
    int n = 0;
    if (*p == '-') { out[n++] = '-'; ++p; }
    char digits[24];
    int nd = 0;
    digits[nd++] = *p++;
    if (*p == '.') { ++p; while (*p != 'e') digits[nd++] = *p++; }
    ++p;                                                     // 'e'
    int e10 = 0;
    std::from_chars(p + ((*p == '+') ? 1 : 0), res.ptr, e10);
    if (e10 < -4 || e10 >= 16) {                             // exponent form
        out[n++] = digits[0];
        if (nd > 1) { out[n++] = '.'; memcpy(out + n, digits + 1, (size_t)(nd - 1)); n += nd - 1; }
        out[n++] = 'e';
        out[n++] = e10 < 0 ? '-' : '+';
        int a = e10 < 0 ? -e10 : e10;
        char eb[8];
        int ne = 0;
        do { eb[ne++] = (char)('0' + a % 10); a /= 10; } while (a);
        if (ne < 2) eb[ne++] = '0';
        while (ne) out[n++] = eb[--ne];
        return n;
    }
    if (e10 < 0) {                                           // 0.000ddd
        out[n++] = '0'; out[n++] = '.';
        for (int z = 0; z < -e10 - 1; ++z) out[n++] = '0';
        memcpy(out + n, digits, (size_t)nd); n += nd;
        return n;
    }
    const int int_digits = e10 + 1;                          // digits before the point
    if (nd <= int_digits) {
        memcpy(out + n, digits, (size_t)nd); n += nd;
        for (int z = nd; z < int_digits; ++z) out[n++] = '0';
        out[n++] = '.'; out[n++] = '0';
    } else {
        memcpy(out + n, digits, (size_t)int_digits); n += int_digits;
        out[n++] = '.';
        memcpy(out + n, digits + int_digits, (size_t)(nd - int_digits)); n += nd - int_digits;
    }
    return n;
}

// host threads for the text format: 600 000 numbers are 22 ms to format and 16 ms to parse on one core - with the fields at
// 5 ms that was four fifths of a file-to-file orient_pointcloud call on 100 k points.  At most 8 threads (a process
// under a CPU quota still sees every core of the box), one block per ~4096 rows.
#ifndef DNP_IO_THREADS
#define DNP_IO_THREADS 8
#endif
constexpr int64_t kIoThreads = DNP_IO_THREADS, kIoRowsPerBlock = 4096;
static inline int64_t io_blocks(int64_t n_rows) {
    int64_t b = n_rows / kIoRowsPerBlock;
    const int64_t hw = (int64_t)std::thread::hardware_concurrency();
    const int64_t cap = hw > 0 && hw < kIoThreads ? hw : kIoThreads;
    return b < 1 ? 1 : (b > cap ? cap : b);
}
template <typename Fn>
static void run_blocks(int64_t blocks, Fn&& fn) {
    if (blocks <= 1) { fn((int64_t)0); return; }
    std::vector<std::thread> pool;
    int64_t inline_from = blocks;                       // blocks [inline_from, blocks) run here when no thread could be started
    for (int64_t b = 1; b < blocks; ++b) {
        try {
            pool.emplace_back([&fn, b] { fn(b); });
        } catch (...) {                                  // std::system_error (thread limit): nothing may leave an extern "C" entry
            inline_from = b;
            break;
        }
    }
    fn((int64_t)0);
    for (int64_t b = inline_from; b < blocks; ++b) fn(b);
    for (auto& t : pool) t.join();
}

// One block of whole lines [p, end): rows appended to `vals` (cols values each), cols fixed by the block's first row (or
// given).  Returns the rows parsed, -2 for text that is not regular.
static int64_t parse_lines(const char* p, const char* end, std::vector<float>* vals, float* direct, int64_t max_rows, int* cols_io) {
    auto is_space = [](char c) { return c == ' ' || c == '\t' || c == '\r' || c == '\v' || c == '\f'; };
    int64_t rows = 0;
    int cols = *cols_io;
    while (p < end) {
        const char* eol = (const char*)memchr(p, '\n', (size_t)(end - p));
        if (!eol) eol = end;
        const char* a = p;
        const char* b = eol;
        while (a < b && is_space(*a)) ++a;
        while (b > a && is_space(b[-1])) --b;
        p = eol + 1;
        if (a == b) continue;                                                  // blank line: one empty token, ignored
        float v6[6];
        int k = 0;
        while (a < b) {
            const char* tok_end = (const char*)memchr(a, ' ', (size_t)(b - a));
            if (!tok_end) tok_end = b;
            if (tok_end == a || k == 6) return -2;                             // double space / too many columns
            const char* q = a;
            if (*q == '+' && q + 1 < tok_end && (q[1] == '.' || (q[1] >= '0' && q[1] <= '9'))) ++q;   // float("+1.5")
            double v = 0.0;
            const auto res = std::from_chars(q, tok_end, v);
            if (res.ec != std::errc() || res.ptr != tok_end) return -2;        // underscores, stray characters, ...
            if (v != v) return -2;                                             // 'nan' lines are dropped: line-by-line path
            v6[k++] = (float)v;
            a = tok_end < b ? tok_end + 1 : b;
            if (tok_end < b && a == b) return -2;                              // trailing separator inside the line
        }
        if (k != 3 && k != 6) return -2;
        if (cols == 0) cols = k;
        if (k != cols) return -2;
        if (vals) {
            vals->insert(vals->end(), v6, v6 + cols);
        } else {
            if (rows >= max_rows) return DNP_EWORKSPACE;
            if (direct) memcpy(direct + rows * cols, v6, sizeof(float) * (size_t)cols);
        }
        ++rows;
    }
    *cols_io = cols;
    return rows;
}

}  // namespace dnp

using namespace dnp;

extern "C" {

int64_t dnp_xyz_format_bound(int64_t n_rows, int64_t n_cols) { return n_rows * n_cols * 26 + 1; }

int64_t dnp_xyz_format_f32(const float* rows, int64_t n_rows, int64_t n_cols, char* out, int64_t cap) {
    clear_error();
    if (n_rows < 0 || n_cols < 0 || (n_rows > 0 && n_cols > 0 && (!rows || !out))) {
        set_error("bad arguments to dnp_xyz_format_f32");
        return DNP_EINVAL;
    }
    if (n_rows > 0 && n_cols == 0) { set_error("rows without columns"); return DNP_EINVAL; }
    if (cap < dnp_xyz_format_bound(n_rows, n_cols)) {
        set_error("output buffer of %lld bytes, %lld needed", (long long)cap, (long long)dnp_xyz_format_bound(n_rows, n_cols));
        return DNP_EWORKSPACE;
    }
    // blocks of rows on up to kIoThreads host threads: every block is formatted at the place its bound reserves in `out`
    // (disjoint by construction), then the blocks are moved down to close the gaps - byte for byte the serial text
    const int64_t blocks = io_blocks(n_rows);
    int64_t len[kIoThreads] = {0};
    auto format_block = [&](int64_t b) {
        const int64_t r0 = n_rows * b / blocks, r1 = n_rows * (b + 1) / blocks;
        char* o = out + r0 * n_cols * 26;
        int64_t n = 0;
        for (int64_t r = r0; r < r1; ++r) {
            if (r) o[n++] = '\n';
            for (int64_t c = 0; c < n_cols; ++c) {
                if (c) o[n++] = ' ';
                n += py_repr((double)rows[r * n_cols + c], o + n);   // str(v) of the Python float = the float32 as a double
            }
        }
        len[b] = n;
    };
    run_blocks(blocks, format_block);
    int64_t n = len[0];
    for (int64_t b = 1; b < blocks; ++b) {
        memmove(out + n, out + (n_rows * b / blocks) * n_cols * 26, (size_t)len[b]);
        n += len[b];
    }
    return n;
}

// Regular '.xyz' text: every line that is not blank after stripping holds the same number (3 or 6) of
// single-space separated numbers; no line contains "nan".  Returns the number of rows parsed into out[max_rows, *ncol]
// (each value parsed as a double, then rounded to float - what float(c) followed by torch.tensor(..., float32) does),
// -2 when the text is not of that form (the caller then takes the line-by-line path that defines the semantics),
// DNP_EWORKSPACE when max_rows is too small.
int64_t dnp_xyz_parse_f32(const char* txt, int64_t len, float* out, int64_t max_rows, int32_t* ncol) {
    clear_error();
    if (len < 0 || !ncol || (len > 0 && !txt)) { set_error("bad arguments to dnp_xyz_parse_f32"); return DNP_EINVAL; }
    // blocks of whole lines on up to kIoThreads host threads (cut at the newline behind every len / blocks bytes), each into
    // its own vector; the rows are then laid end to end - the same rows in the same order as one pass, and the same
    // verdict: every block regular with the same number of columns
    const int64_t blocks = io_blocks(len / 48);                                // ~48 bytes per 6-column row
    if (blocks <= 1) {
        int cols = 0;
        const int64_t rows = parse_lines(txt, txt + len, nullptr, out, max_rows, &cols);
        if (rows == DNP_EWORKSPACE) { set_error("more than %lld rows", (long long)max_rows); return rows; }
        if (rows >= 0) *ncol = cols;
        return rows;
    }
    try {
    std::vector<const char*> cut((size_t)blocks + 1, txt + len);
    cut[0] = txt;
    for (int64_t b = 1; b < blocks; ++b) {
        const char* from = txt + len * b / blocks;
        if (from < cut[(size_t)b - 1]) from = cut[(size_t)b - 1];
        const char* nl = (const char*)memchr(from, '\n', (size_t)(txt + len - from));
        cut[(size_t)b] = nl ? nl + 1 : txt + len;
    }
    std::vector<std::vector<float>> vals((size_t)blocks);
    std::vector<int64_t> rows((size_t)blocks, 0);
    std::vector<int> cols((size_t)blocks, 0);
    run_blocks(blocks, [&](int64_t b) {
        try {
            vals[(size_t)b].reserve((size_t)((cut[(size_t)b + 1] - cut[(size_t)b]) / 8 + 6));
            rows[(size_t)b] = parse_lines(cut[(size_t)b], cut[(size_t)b + 1], &vals[(size_t)b], nullptr, 0, &cols[(size_t)b]);
        } catch (...) {                                  // out of memory in a worker: the caller's line-by-line path decides
            rows[(size_t)b] = -2;
        }
    });
    int64_t total = 0;
    int c = 0;
    for (int64_t b = 0; b < blocks; ++b) {
        if (rows[(size_t)b] < 0) return -2;
        if (rows[(size_t)b] == 0) continue;                                    // a block of blank lines
        if (c == 0) c = cols[(size_t)b];
        if (cols[(size_t)b] != c) return -2;
        total += rows[(size_t)b];
    }
    if (total > max_rows) { set_error("more than %lld rows", (long long)max_rows); return DNP_EWORKSPACE; }
    if (out) {
        int64_t at = 0;
        for (int64_t b = 0; b < blocks; ++b) {
            if (!vals[(size_t)b].empty()) memcpy(out + at, vals[(size_t)b].data(), vals[(size_t)b].size() * sizeof(float));
            at += (int64_t)vals[(size_t)b].size();
        }
    }
    *ncol = c;
    return total;
    } catch (...) {                                      // allocation failure: not a verdict on the text
        return -2;
    }
}

}  // extern "C"
