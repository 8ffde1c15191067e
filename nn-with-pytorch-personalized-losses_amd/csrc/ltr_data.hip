// ltr_data.hip -- the data path in front of the slate pipeline.
//
//   (1) LETOR / svmlight text -> packed fp32 rows.  The reference parses with sklearn and then walks every
//       document in Python (utils/dataset.py:34-69: `data[0][i].toarray().reshape(-1).astype(np.float32)` per
//       document, joblib-cached because it is slow).  Here: mmap + N host threads, two passes (count lines and
//       feature-id range; fill), straight into the packed [n_docs][F] layout the kernels stream.  HOST code.
//   (2) Per-epoch shuffle.  The reference gathers the whole training tensor every epoch
//       (`X_train = X_train[idx]`, main_batch_execution.py:112-117).  ltr_gather_rows_f32 is that gather on the
//       device: one coalesced 16-byte-per-lane row copy per destination row, HBM-bound (reads + writes every byte
//       once).  The slate pipeline itself can also take the permutation directly (see ltr_fused_step's callers),
//       in which case nothing is copied at all.
#include "../../include/ltr_mi355x.h"
#include <hip/hip_runtime.h>

#include <fcntl.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <thread>
#include <vector>

namespace {

// ------------------------------------------------------------------------------------------ device gather
// dst[r][:] = src[idx[r]][:], rows of `row_f4` float4s.  One workgroup handles kRowsPerBlock destination rows;
// lanes sweep a row in 16-byte pieces (fully coalesced on both sides).  Negative indices count from the end like torch
// indexing (idx + src_rows); an index still out of range yields a ZERO row (torch raises a device-side assert there).
constexpr int kGatherThreads = 256;
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(kGatherThreads)
gather_rows_kernel(const f32x4 *__restrict__ src, const int64_t *__restrict__ idx, int64_t n_rows, int64_t src_rows,
                   int row_f4, f32x4 *__restrict__ dst) {
    const int64_t total = n_rows * row_f4;
    for (int64_t e = (int64_t)blockIdx.x * kGatherThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kGatherThreads) {
        const int64_t r = e / row_f4;
        const int c = (int)(e - r * row_f4);
        int64_t s = idx[r];
        if (s < 0) s += src_rows;
        dst[e] = s >= 0 && s < src_rows ? __builtin_nontemporal_load(src + s * row_f4 + c) : f32x4{0.f, 0.f, 0.f, 0.f};
    }
}

// Wide rows (a whole query: 128 x 136 floats = 69 632 B): one workgroup moves a 16 KiB piece of ONE row -- the row index is read
// once per workgroup (scalar), every lane streams 4 x 16 B with no per-element division, loads issued before the stores.
constexpr int kGatherPiece = kGatherThreads * 4;      // float4s per workgroup
__global__ void __launch_bounds__(kGatherThreads)
gather_rows_wide_kernel(const f32x4 *__restrict__ src, const int64_t *__restrict__ idx, int64_t n_rows, int64_t src_rows, int row_f4,
                        int pieces, f32x4 *__restrict__ dst) {
    for (int64_t b = blockIdx.x; b < n_rows * pieces; b += gridDim.x) {
        const int64_t r = b / pieces;
        const int piece = (int)(b - r * pieces);
        int64_t srow = idx[r];
        if (srow < 0) srow += src_rows;
        const bool ok = srow >= 0 && srow < src_rows;
        const f32x4 *sp = src + (ok ? srow : 0) * row_f4;
        f32x4 *dp = dst + r * row_f4;
        const int c0 = piece * kGatherPiece + threadIdx.x;
        f32x4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k * kGatherThreads < row_f4) v[k] = ok ? __builtin_nontemporal_load(sp + c0 + k * kGatherThreads) : f32x4{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (c0 + k * kGatherThreads < row_f4) __builtin_nontemporal_store(v[k], dp + c0 + k * kGatherThreads);
    }
}

__global__ void __launch_bounds__(kGatherThreads)
gather_rows_scalar_kernel(const float *__restrict__ src, const int64_t *__restrict__ idx, int64_t n_rows, int64_t src_rows,
                          int row_f, float *__restrict__ dst) {
    const int64_t total = n_rows * row_f;
    for (int64_t e = (int64_t)blockIdx.x * kGatherThreads + threadIdx.x; e < total; e += (int64_t)gridDim.x * kGatherThreads) {
        const int64_t r = e / row_f;
        const int c = (int)(e - r * row_f);
        int64_t s = idx[r];
        if (s < 0) s += src_rows;
        dst[e] = s >= 0 && s < src_rows ? src[s * row_f + c] : 0.f;
    }
}

// ------------------------------------------------------------------------------------------ svmlight parser
struct Mapped {
    const char *p = nullptr;
    size_t n = 0;
    int fd = -1;
    ~Mapped() {
        if (p && n) munmap(const_cast<char *>(p), n);
        if (fd >= 0) close(fd);
    }
    int open_file(const char *path) {
        fd = open(path, O_RDONLY);
        if (fd < 0) return LTR_ERR_IO;
        struct stat st;
        if (fstat(fd, &st) != 0) return LTR_ERR_IO;
        n = (size_t)st.st_size;
        if (n == 0) return LTR_OK;
        void *m = mmap(nullptr, n, PROT_READ, MAP_PRIVATE, fd, 0);
        if (m == MAP_FAILED) {
            n = 0;
            return LTR_ERR_IO;
        }
        p = static_cast<const char *>(m);
        return LTR_OK;
    }
};

// byte ranges [b, e) of `parts` chunks, each starting at the beginning of a line
std::vector<std::pair<size_t, size_t>> split_lines(const Mapped &m, int parts) {
    std::vector<std::pair<size_t, size_t>> out;
    size_t b = 0;
    for (int k = 1; k <= parts && b < m.n; ++k) {
        size_t e = k == parts ? m.n : std::max(b, m.n / (size_t)parts * (size_t)k);
        if (e > 0 && e < m.n) {      // move to the first line start at or after e
            const void *nl = memchr(m.p + e - 1, '\n', m.n - (e - 1));
            e = nl ? (size_t)(static_cast<const char *>(nl) - m.p) + 1 : m.n;
        }
        if (e > b) out.emplace_back(b, e);
        b = e;
    }
    return out;
}

inline bool blank_or_comment(const char *s, const char *e) {
    while (s < e && (*s == ' ' || *s == '\t' || *s == '\r')) ++s;
    return s >= e || *s == '#';
}

struct ScanResult {
    int64_t docs = 0;
    int32_t min_fid = 0x7fffffff, max_fid = -1;
    int err = 0;
};

// Bounded numeric parsing: the mapping is NOT NUL-terminated and strtod/strtol skip '\n' as white space, so every token
// is delimited first ([s, t): up to the next blank / '#' / ':' / line end e), copied into a NUL-terminated stack buffer
// and must be consumed ENTIRELY by the conversion.  Empty values ("2:" at a line end), over-long tokens and non-finite
// values (nan, inf, 1e400) are parse errors, as is anything that would read past the line.
inline const char *token_end(const char *s, const char *e, bool stop_at_colon) {
    while (s < e && *s != ' ' && *s != '\t' && *s != '\r' && *s != '\n' && *s != '#' && !(stop_at_colon && *s == ':')) ++s;
    return s;
}

inline bool parse_f64(const char *s, const char *t, double *out) {
    char buf[64];
    const size_t n = (size_t)(t - s);
    if (n == 0 || n >= sizeof(buf)) return false;
    memcpy(buf, s, n);
    buf[n] = 0;
    char *q = nullptr;
    const double v = strtod(buf, &q);
    if (q != buf + n || !(v - v == 0.0)) return false;        // whole token consumed; finite
    if (buf[0] == ' ' || buf[0] == '\t') return false;
    *out = v;
    return true;
}

inline bool parse_i64(const char *s, const char *t, long long *out) {
    if (s >= t || t - s > 19) return false;
    bool neg = false;
    if (*s == '-' || *s == '+') {
        neg = *s == '-';
        if (++s >= t) return false;
    }
    long long v = 0;
    for (; s < t; ++s) {
        if (*s < '0' || *s > '9') return false;
        const int dgt = *s - '0';
        if (v > (0x7fffffffffffffffLL - dgt) / 10) return false;      // 19-digit tokens can pass LLONG_MAX: reject, never overflow
        v = v * 10 + dgt;
    }
    *out = neg ? -v : v;
    return true;
}

// Parse one line [s, e) (e = the '\n' or the end of the mapping; never dereferenced).  X == nullptr: scan only
// (feature-id range).  Returns 0, or LTR_ERR_PARSE.
inline int parse_line(const char *s, const char *e, int n_features, int fid_base, float *X, double *y, int64_t *qid,
                      int32_t *min_fid, int32_t *max_fid) {
    while (s < e && (*s == ' ' || *s == '\t')) ++s;
    const char *t = token_end(s, e, false);
    double label;
    if (!parse_f64(s, t, &label)) return LTR_ERR_PARSE;
    if (y) *y = label;
    s = t;
    if (qid) *qid = -1;
    while (s < e) {
        while (s < e && (*s == ' ' || *s == '\t' || *s == '\r')) ++s;
        if (s >= e || *s == '#') break;
        if (e - s > 4 && s[0] == 'q' && s[1] == 'i' && s[2] == 'd' && s[3] == ':') {
            t = token_end(s + 4, e, false);
            long long v;
            if (!parse_i64(s + 4, t, &v)) return LTR_ERR_PARSE;
            if (qid) *qid = (int64_t)v;
            s = t;
            continue;
        }
        t = token_end(s, e, true);
        long long fid;
        if (t >= e || *t != ':' || !parse_i64(s, t, &fid)) return LTR_ERR_PARSE;
        s = t + 1;
        t = token_end(s, e, false);
        double v;
        if (!parse_f64(s, t, &v)) return LTR_ERR_PARSE;           // "k:" with nothing behind it is an error, like sklearn
        s = t;
        if (fid < 0 || fid > 0x7ffffff0) return LTR_ERR_PARSE;
        if (min_fid && (int32_t)fid < *min_fid) *min_fid = (int32_t)fid;
        if (max_fid && (int32_t)fid > *max_fid) *max_fid = (int32_t)fid;
        if (X) {
            const long long c = fid - fid_base;
            if (c < 0 || c >= n_features) return LTR_ERR_PARSE;
            X[c] = (float)v;                     // float64 parse, then the reference's astype(np.float32) (dataset.py:63)
        }
    }
    return 0;
}

template <class F>
int for_each_line(const char *b, const char *e, F f) {
    while (b < e) {
        const char *nl = static_cast<const char *>(memchr(b, '\n', (size_t)(e - b)));
        const char *le = nl ? nl : e;
        if (!blank_or_comment(b, le))
            if (int rc = f(b, le)) return rc;
        b = nl ? nl + 1 : e;
    }
    return 0;
}

int clamp_threads(int n) {
    if (n <= 0) n = (int)std::thread::hardware_concurrency();
    return n < 1 ? 1 : (n > 64 ? 64 : n);
}

}  // namespace

extern "C" {

int ltr_gather_rows_f32(const float *src, int64_t src_rows, const int64_t *idx, int64_t n_rows, int64_t row_floats,
                        float *dst, void *stream) {
    if (!src || !idx || !dst) return LTR_ERR_NULL;
    if (n_rows < 0 || src_rows < 0 || row_floats < 1 || row_floats > (1 << 28)) return LTR_ERR_SHAPE;
    if (n_rows == 0) return LTR_OK;
    const bool vec = row_floats % 4 == 0 && ((uintptr_t)src & 15u) == 0 && ((uintptr_t)dst & 15u) == 0;
    const int64_t total = vec ? n_rows * (row_floats / 4) : n_rows * row_floats;
    int64_t blocks = (total + kGatherThreads - 1) / kGatherThreads;
    if (blocks > 256 * 32) blocks = 256 * 32;                      // 32 workgroups per CU, grid-stride beyond
    if (vec && row_floats / 4 >= kGatherPiece / 2) {
        const int row_f4 = (int)(row_floats / 4), pieces = (row_f4 + kGatherPiece - 1) / kGatherPiece;
        int64_t wg = n_rows * pieces;
        if (wg > 256 * 64) wg = 256 * 64;
        hipLaunchKernelGGL(gather_rows_wide_kernel, dim3((unsigned)wg), dim3(kGatherThreads), 0, (hipStream_t)stream,
                           reinterpret_cast<const f32x4 *>(src), idx, n_rows, src_rows, row_f4, pieces, reinterpret_cast<f32x4 *>(dst));
    } else if (vec)
        hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(kGatherThreads), 0, (hipStream_t)stream,
                           reinterpret_cast<const f32x4 *>(src), idx, n_rows, src_rows, (int)(row_floats / 4),
                           reinterpret_cast<f32x4 *>(dst));
    else
        hipLaunchKernelGGL(gather_rows_scalar_kernel, dim3((unsigned)blocks), dim3(kGatherThreads), 0, (hipStream_t)stream, src,
                           idx, n_rows, src_rows, (int)row_floats, dst);
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? LTR_OK : (int)e;
}

int ltr_svmlight_scan(const char *path, int64_t *n_docs, int32_t *min_feature_id, int32_t *max_feature_id, int n_threads) {
    if (!path || !n_docs || !min_feature_id || !max_feature_id) return LTR_ERR_NULL;
    Mapped m;
    if (int rc = m.open_file(path)) return rc;
    const auto parts = split_lines(m, clamp_threads(n_threads));
    std::vector<ScanResult> res(parts.size());
    std::vector<std::thread> th;
    for (size_t t = 0; t < parts.size(); ++t)
        th.emplace_back([&, t] {
            ScanResult &r = res[t];
            r.err = for_each_line(m.p + parts[t].first, m.p + parts[t].second, [&](const char *b, const char *e) {
                ++r.docs;
                return parse_line(b, e, 0, 0, nullptr, nullptr, nullptr, &r.min_fid, &r.max_fid);
            });
        });
    for (auto &t : th) t.join();
    int64_t docs = 0;
    int32_t lo = 0x7fffffff, hi = -1;
    for (const auto &r : res) {
        if (r.err) return r.err;
        docs += r.docs;
        lo = std::min(lo, r.min_fid);
        hi = std::max(hi, r.max_fid);
    }
    *n_docs = docs;
    *min_feature_id = hi < 0 ? 0 : lo;
    *max_feature_id = hi;
    return LTR_OK;
}

int ltr_svmlight_load(const char *path, int64_t n_docs, int n_features, int feature_id_base, float *X, double *y,
                      int64_t *qid, int n_threads) {
    if (!path || !X || !y || !qid) return LTR_ERR_NULL;
    if (n_docs < 0 || n_features < 1) return LTR_ERR_SHAPE;
    Mapped m;
    if (int rc = m.open_file(path)) return rc;
    const auto parts = split_lines(m, clamp_threads(n_threads));
    // pass 1: lines per chunk -> row offsets
    std::vector<int64_t> count(parts.size(), 0);
    {
        std::vector<std::thread> th;
        for (size_t t = 0; t < parts.size(); ++t)
            th.emplace_back([&, t] {
                for_each_line(m.p + parts[t].first, m.p + parts[t].second, [&](const char *, const char *) {
                    ++count[t];
                    return 0;
                });
            });
        for (auto &t : th) t.join();
    }
    std::vector<int64_t> first(parts.size() + 1, 0);
    for (size_t t = 0; t < parts.size(); ++t) first[t + 1] = first[t] + count[t];
    if (first[parts.size()] != n_docs) return LTR_ERR_SHAPE;
    std::vector<int> err(parts.size(), 0);
    std::vector<std::thread> th;
    for (size_t t = 0; t < parts.size(); ++t)
        th.emplace_back([&, t] {
            int64_t row = first[t];
            err[t] = for_each_line(m.p + parts[t].first, m.p + parts[t].second, [&](const char *b, const char *e) {
                float *xr = X + (size_t)row * n_features;
                memset(xr, 0, sizeof(float) * (size_t)n_features);                   // absent features are 0 (sparse format)
                const int rc = parse_line(b, e, n_features, feature_id_base, xr, y + row, qid + row, nullptr, nullptr);
                ++row;
                return rc;
            });
        });
    for (auto &t : th) t.join();
    for (int e : err)
        if (e) return e;
    return LTR_OK;
}

}  // extern "C"
