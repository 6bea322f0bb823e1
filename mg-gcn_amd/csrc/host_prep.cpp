// host_prep.cpp -- host-side graph preprocessing behind the C ABI.
//
// The reference does this work on the host too, with parallel-STL loops over
// managed memory: csr_matrix::normalize (src/matrix.hpp:340-390), ::transpose
// (src/matrix.hpp:392-453), the P x P block split of dist_row_csr_matrix
// (src/dist_matrix.hpp:215-259) and dn_matrix::init (src/matrix.hpp:539-545).
// Here it is plain C++17 + std::thread (no TBB, no OpenMP runtime to clash with
// the host application's), deterministic by construction: every thread owns a
// contiguous row range and partial results are combined in thread order, so the
// output equals the serial algorithm's regardless of the thread count.
#include <algorithm>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <random>
#include <thread>
#include <vector>

#include "mggcn.h"

#define MGGCN_API extern "C" __attribute__((visibility("default")))

namespace {

// per_thread_bytes: scratch each extra thread costs (histograms); the total is
// capped at 2 GiB so 100 M-vertex graphs do not multiply their footprint by T.
unsigned host_threads(size_t work, size_t per_thread_bytes = 0) {
    unsigned hw = std::thread::hardware_concurrency();
    if (const char *s = std::getenv("MGGCN_HOST_THREADS")) hw = (unsigned)std::strtoul(s, nullptr, 10);
    if (hw < 1) hw = 1;
    if (hw > 64) hw = 64;
    // below ~1M non-zeros per thread the thread start-up costs more than it saves (MGGCN_HOST_THREADS_MIN_NNZ: the
    // sanitizer tests thread small matrices)
    size_t per_thread = (size_t)1 << 20;
    if (const char *s = std::getenv("MGGCN_HOST_THREADS_MIN_NNZ")) per_thread = std::max<size_t>(1, std::strtoull(s, nullptr, 10));
    const unsigned by_work = (unsigned)std::min<size_t>(64, std::max<size_t>(1, work / per_thread));
    unsigned T = std::min(hw, by_work);
    if (per_thread_bytes) T = std::min<unsigned>(T, (unsigned)std::max<size_t>(1, (2ull << 30) / per_thread_bytes));
    return T;
}

// Cuts [0, n) rows into T ranges with about equal non-zeros (indptr is the prefix sum).
std::vector<uint32_t> balanced_row_cuts(const uint32_t *indptr, uint32_t n, unsigned T) {
    std::vector<uint32_t> cut(T + 1, n);
    cut[0] = 0;
    const uint64_t base = indptr[0], total = (uint64_t)indptr[n] - base;
    for (unsigned t = 1; t < T; t++) {
        const uint64_t target = base + total * t / T;
        cut[t] = (uint32_t)(std::lower_bound(indptr, indptr + n, (uint32_t)target) - indptr);
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    return cut;
}

template <typename F>
void run_threads(unsigned T, F &&f) {
    if (T <= 1) { f(0u); return; }
    std::vector<std::thread> th;
    th.reserve(T);
    for (unsigned t = 0; t < T; t++) th.emplace_back(f, t);
    for (auto &x : th) x.join();
}

// q has nq+1 entries; returns j with q[j] <= col < q[j+1]
inline uint32_t col_block(const uint32_t *q, uint32_t nq, uint32_t col) {
    return (uint32_t)(std::upper_bound(q + 1, q + nq, col) - (q + 1));
}

}  // namespace

MGGCN_API void mggcn_csr_normalize_host(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr,
                                        const uint32_t *indices, float *values, int axis) {
    if (!n_rows) return;
    const size_t nnz = (size_t)indptr[n_rows] - indptr[0];
    const unsigned T = host_threads(nnz, axis ? (size_t)n_cols * sizeof(float) : 0);
    const auto cut = balanced_row_cuts(indptr, n_rows, T);
    if (!axis) {
        run_threads(T, [&](unsigned t) {
            for (uint32_t v = cut[t]; v < cut[t + 1]; v++) {
                float sum = 0.f;
                for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) sum += values[e];
                for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) values[e] /= sum;
            }
        });
        return;
    }
    // Column sums.  The serial reference order adds the entries of a column in
    // increasing row order; thread t covers a contiguous row range, so combining the
    // per-thread partial sums in thread order reproduces a fixed (thread-count
    // dependent only in rounding) result.  T == 1 is exactly the serial loop.
    std::vector<std::vector<float>> part(T, std::vector<float>(n_cols, 0.f));
    run_threads(T, [&](unsigned t) {
        float *deg = part[t].data();
        for (uint32_t v = cut[t]; v < cut[t + 1]; v++)
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) deg[indices[e]] += values[e];
    });
    std::vector<float> &deg = part[0];
    for (unsigned t = 1; t < T; t++)
        for (uint32_t c = 0; c < n_cols; c++) deg[c] += part[t][c];
    run_threads(T, [&](unsigned t) {
        for (uint32_t v = cut[t]; v < cut[t + 1]; v++)
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) values[e] /= deg[indices[e]];
    });
}

MGGCN_API void mggcn_csr_transpose_host(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr,
                                        const uint32_t *indices, const float *values, uint32_t *t_indptr,
                                        uint32_t *t_indices, float *t_values) {
    std::memset(t_indptr, 0, ((size_t)n_cols + 1) * sizeof(uint32_t));
    if (!n_rows) return;
    const size_t nnz = (size_t)indptr[n_rows] - indptr[0];
    const unsigned T = host_threads(nnz, (size_t)n_cols * sizeof(uint32_t));
    const auto cut = balanced_row_cuts(indptr, n_rows, T);
    // counting sort with per-thread column histograms: thread t's entries of column c
    // go after those of threads < t, which is the serial (row-ascending) order.
    std::vector<std::vector<uint32_t>> cnt(T, std::vector<uint32_t>(n_cols, 0u));
    run_threads(T, [&](unsigned t) {
        uint32_t *c = cnt[t].data();
        for (uint32_t v = cut[t]; v < cut[t + 1]; v++)
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) c[indices[e]]++;
    });
    uint32_t run = 0;
    for (uint32_t c = 0; c < n_cols; c++) {
        t_indptr[c] = run;
        for (unsigned t = 0; t < T; t++) {
            const uint32_t k = cnt[t][c];
            cnt[t][c] = run;          // becomes thread t's write cursor for column c
            run += k;
        }
    }
    t_indptr[n_cols] = run;
    run_threads(T, [&](unsigned t) {
        uint32_t *cur = cnt[t].data();
        for (uint32_t v = cut[t]; v < cut[t + 1]; v++)
            for (uint32_t e = indptr[v]; e < indptr[v + 1]; e++) {
                const uint32_t at = cur[indices[e]]++;
                t_indices[at] = v;
                t_values[at] = values[e];
            }
    });
}

MGGCN_API void mggcn_csr_block_split_count_host(const uint32_t *indptr, const uint32_t *indices,
                                                uint32_t row_begin, uint32_t row_end, const uint32_t *q,
                                                uint32_t nq, uint32_t *blk_indptr) {
    const uint32_t rows = row_end - row_begin;
    std::memset(blk_indptr, 0, (size_t)nq * (rows + 1) * sizeof(uint32_t));
    if (!rows) return;
    const size_t nnz = (size_t)indptr[row_end] - indptr[row_begin];
    const unsigned T = host_threads(nnz);
    const auto cut = balanced_row_cuts(indptr + row_begin, rows, T);
    run_threads(T, [&](unsigned t) {
        for (uint32_t r = cut[t]; r < cut[t + 1]; r++)
            for (uint32_t k = indptr[row_begin + r]; k < indptr[row_begin + r + 1]; k++)
                blk_indptr[(size_t)col_block(q, nq, indices[k]) * (rows + 1) + r + 1]++;
    });
    for (uint32_t j = 0; j < nq; j++) {
        uint32_t *p = blk_indptr + (size_t)j * (rows + 1);
        for (uint32_t r = 0; r < rows; r++) p[r + 1] += p[r];
    }
}

MGGCN_API void mggcn_csr_block_split_fill_host(const uint32_t *indptr, const uint32_t *indices,
                                               const float *values, uint32_t row_begin, uint32_t row_end,
                                               const uint32_t *q, uint32_t nq, const uint32_t *blk_indptr,
                                               uint32_t *const *blk_indices, float *const *blk_values) {
    const uint32_t rows = row_end - row_begin;
    if (!rows) return;
    const size_t nnz = (size_t)indptr[row_end] - indptr[row_begin];
    const unsigned T = host_threads(nnz);
    const auto cut = balanced_row_cuts(indptr + row_begin, rows, T);
    run_threads(T, [&](unsigned t) {
        std::vector<uint32_t> fill(nq);
        for (uint32_t r = cut[t]; r < cut[t + 1]; r++) {
            std::fill(fill.begin(), fill.end(), 0u);
            for (uint32_t k = indptr[row_begin + r]; k < indptr[row_begin + r + 1]; k++) {
                const uint32_t j = col_block(q, nq, indices[k]);
                const uint32_t at = blk_indptr[(size_t)j * (rows + 1) + r] + fill[j]++;
                blk_indices[j][at] = indices[k] - q[j];
                blk_values[j][at] = values[k];
            }
        }
    });
}

MGGCN_API void mggcn_init_uniform_host(float *buffer, size_t n_rows, size_t n_cols, float gain) {
    // std::default_random_engine + uniform_real_distribution<float> are the
    // reference's own (libstdc++) generators: seed 99, restarted for every tensor.
    if (gain < 0.f) gain = (float)std::sqrt(2 / (1 + 0.01 * 0.01));
    std::default_random_engine gen(99);
    gain *= std::sqrt(3.0 / n_rows);
    std::uniform_real_distribution<float> uni(-gain, gain);
    for (size_t i = 0; i < n_rows * n_cols; i++) buffer[i] = uni(gen);
}
