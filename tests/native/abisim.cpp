// tests/native/abisim.cpp -- include/mggcn.h on the stream model of tests/native/hipsim/: every entry point of the C ABI that the
// C++ host layer (mg-gcn_amd/host/*.hpp) calls, as a plain CPU loop enqueued on a modelled stream.  TEST INFRASTRUCTURE ONLY: it
// exists so that the host layer's multi-GPU schedule -- P enqueue threads, compute / communication / copying streams, the event
// edges between them, the double-buffered exchange, deferred releases -- can run on the CPU with every rank on a "device" of its
// own, under ThreadSanitizer and under adversarial stream schedules (tests/native/Makefile: host_dist_sim).  Nothing here is a
// product path and nothing of it is linked into the product; numerics are those of straightforward loops (the comparisons of
// host/tests/test_dist.cpp are distributed against single-device on the SAME backend, and threads against one thread bit for bit).
//
// Modelled faithfully: stream order, event record / wait (capture at call time), asynchronous copies (a host SOURCE is consumed at
// the call, like a pageable hipMemcpyAsync; device memory is read and written when the operation executes), synchronisation
// (a drain of everything enqueued, in the schedule chosen by HIPSIM_POLICY / HIPSIM_SEED).  Fresh device memory is NaN-filled.
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <memory>
#include <mutex>
#include <vector>

#include "hipsim.h"
#include "mggcn.h"

namespace {
thread_local int t_dev = 0;
std::once_flag g_once;

void setup() {
    std::call_once(g_once, [] {
        const char *p = std::getenv("HIPSIM_POLICY"), *s = std::getenv("HIPSIM_SEED");
        hipsim_set_schedule(s ? std::strtoull(s, nullptr, 10) : 1, (hipsim_policy)(p ? std::atoi(p) : 0));
        if (const char *c = std::getenv("HIPSIM_DROP_CLASS")) hipsim_drop_class(std::atoi(c));     // mutation runs: one kind of wait ignored
        if (std::getenv("HIPSIM_STATS"))                                                           // operations the run enqueued (a proxy for API calls)
            std::atexit([] { std::fprintf(stderr, "[hipsim] operations executed %llu, of them event waits asked for %llu\n",
                                          (unsigned long long)hipsim_ops_executed(), (unsigned long long)hipsim_waits_seen()); });
    });
}
hipStream_t st(mggcn_stream_t s) { return reinterpret_cast<hipStream_t>(s); }
// The product's streams are hipStreamNonBlocking: the NULL stream orders nothing against them.  Work on the null stream (the host
// layer's upload / download helpers, each followed by a synchronize of the null stream) therefore runs on the spot, WITHOUT
// draining anything else -- a caller that leans on an implicit synchronisation gets stale or half-written data here too.
template <typename F> void launch(mggcn_stream_t s, F &&f) {
    if (!s) { f(); return; }
    hipsim_enqueue(st(s), std::forward<F>(f));
}
inline float lrelu(float x, float a) { const float y = a * x; return x > y ? x : y; }
inline float at(const float *A, size_t ld, int trans, size_t i, size_t k) { return trans ? A[k * ld + i] : A[i * ld + k]; }   // op(A)[i, k]
}  // namespace

extern "C" {

int mggcn_abi_version(void) { return MGGCN_ABI_VERSION; }
int mggcn_device_count(void) { return 8; }
void mggcn_set_device(int device) { setup(); t_dev = device; (void)hipSetDevice(device); }
int mggcn_get_device(void) { return t_dev; }
void mggcn_device_synchronize(void) { setup(); hipsim_drain(); }
mggcn_stream_t mggcn_stream_create(int) { setup(); return hipsim_stream_create(t_dev); }
void mggcn_stream_destroy(mggcn_stream_t) {}
void mggcn_stream_release_scratch(mggcn_stream_t) {}
void mggcn_stream_synchronize(mggcn_stream_t s) { if (s) hipsim_drain(); }
mggcn_event_t mggcn_event_create(void) { setup(); hipEvent_t e; (void)hipEventCreateWithFlags(&e, 0); return e; }
void mggcn_event_destroy(mggcn_event_t) {}
void mggcn_event_record(mggcn_event_t e, mggcn_stream_t s) { (void)hipEventRecord(reinterpret_cast<hipEvent_t>(e), st(s)); }
void mggcn_stream_wait_event(mggcn_stream_t s, mggcn_event_t e) { (void)hipStreamWaitEvent(st(s), reinterpret_cast<hipEvent_t>(e), 0); }
void mggcn_event_synchronize(mggcn_event_t) { hipsim_drain(); }
float mggcn_event_elapsed_ms(mggcn_event_t, mggcn_event_t) { return 0.f; }

void *mggcn_malloc(size_t bytes) {
    if (!bytes) return nullptr;
    void *p = std::malloc(bytes);
    std::memset(p, 0xFF, bytes);                       // NaN floats, huge indices: a read before the first write shows
    return p;
}
void mggcn_free(void *p) { std::free(p); }
void *mggcn_malloc_host(size_t bytes) { return bytes ? std::calloc(1, bytes) : nullptr; }
void mggcn_free_host(void *p) { std::free(p); }
void mggcn_memcpy_h2d(void *dst, const void *src, size_t bytes, mggcn_stream_t s) {
    if (!bytes) return;
    auto staged = std::make_shared<std::vector<char>>((const char *)src, (const char *)src + bytes);   // pageable source: consumed now
    launch(s, [dst, staged] { std::memcpy(dst, staged->data(), staged->size()); });
}
void mggcn_memcpy_d2h(void *dst, const void *src, size_t bytes, mggcn_stream_t s) {
    if (bytes) launch(s, [=] { std::memcpy(dst, src, bytes); });
}
void mggcn_memcpy_d2d(void *dst, const void *src, size_t bytes, mggcn_stream_t s) {
    if (bytes) launch(s, [=] { std::memmove(dst, src, bytes); });
}
void mggcn_memset_zero(void *dst, size_t bytes, mggcn_stream_t s) {
    if (bytes) launch(s, [=] { std::memset(dst, 0, bytes); });
}

// ---- SpMM: the "plan" holds nothing, the call walks the CSR arrays it is handed -----------------
struct mggcn_spmm_plan { uint32_t n_rows, n_cols; };
mggcn_spmm_plan *mggcn_spmm_plan_create_for(uint32_t n_rows, uint32_t n_cols, const uint32_t *, const uint32_t *, const float *, uint32_t, uint32_t) {
    return new mggcn_spmm_plan{n_rows, n_cols};
}
mggcn_spmm_plan *mggcn_spmm_plan_create(uint32_t n_rows, uint32_t n_cols, const uint32_t *ip, const uint32_t *ix, const float *v, uint32_t d) {
    return mggcn_spmm_plan_create_for(n_rows, n_cols, ip, ix, v, d, d);
}
void mggcn_spmm_plan_destroy(mggcn_spmm_plan *p) { delete p; }
void mggcn_spmm_plan_concurrent_builders(uint32_t) {}
void mggcn_spmm_plan_reserved_cus(uint32_t) {}
uint32_t mggcn_spmm_plan_num_items(const mggcn_spmm_plan *) { return 0; }
uint32_t mggcn_spmm_plan_num_split_rows(const mggcn_spmm_plan *) { return 0; }
uint32_t mggcn_spmm_plan_num_sweep_tasks(const mggcn_spmm_plan *) { return 0; }
uint32_t mggcn_spmm_plan_num_launches(const mggcn_spmm_plan *, uint32_t) { return 1; }
size_t mggcn_spmm_plan_bytes(const mggcn_spmm_plan *) { return 0; }
uint32_t mggcn_spmm_plan_num_slices(const mggcn_spmm_plan *) { return 0; }
int mggcn_spmm_plan_describe(const mggcn_spmm_plan *, char *out, size_t cap) { return std::snprintf(out, cap, "model"); }
void mggcn_debug_occupy_cus(mggcn_stream_t, uint32_t, uint32_t, const uint32_t *) {}
uint32_t mggcn_spmm_plan_read_stamps(const mggcn_spmm_plan *, uint32_t, uint64_t *, uint32_t) { return 0; }

void mggcn_spmm_csr_f32(mggcn_stream_t s, const mggcn_spmm_plan *, uint32_t n_rows, uint32_t, const uint32_t *indptr, const uint32_t *indices,
                        const float *values, const float *B, size_t ldb, float *C, size_t ldc, uint32_t d, float alpha, float beta,
                        uint32_t flags, float slope) {
    launch(s, [=] {
        std::vector<float> acc(d);
        for (uint32_t r = 0; r < n_rows; r++) {
            std::fill(acc.begin(), acc.end(), 0.f);
            for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++)
                for (uint32_t c = 0; c < d; c++) acc[c] = std::fmaf(values[e], B[(size_t)indices[e] * ldb + c], acc[c]);
            for (uint32_t c = 0; c < d; c++) {
                float o = alpha * acc[c];
                if (beta != 0.f) o = std::fmaf(beta, C[(size_t)r * ldc + c], o);
                if (flags & MGGCN_SPMM_LEAKY_RELU) o = lrelu(o, slope);
                C[(size_t)r * ldc + c] = o;
            }
        }
    });
}

// ---- GEMM family ---------------------------------------------------------------------------------
size_t mggcn_gemm_workspace_bytes(int, int, uint32_t, uint32_t, uint32_t) { return 0; }
size_t mggcn_gemm_tn_colsum_workspace_bytes(uint32_t, uint32_t, uint32_t) { return 0; }

static void gemm_any(mggcn_stream_t s, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, float alpha, const float *A, size_t lda,
                     const float *B, size_t ldb, float beta, float *C, size_t ldc, const float *bias, const float *Z, size_t ldz, float slope) {
    launch(s, [=] {
        for (uint32_t i = 0; i < M; i++)
            for (uint32_t j = 0; j < N; j++) {
                float acc = 0.f;
                for (uint32_t k = 0; k < K; k++) acc = std::fmaf(at(A, lda, ta, i, k), tb ? B[(size_t)j * ldb + k] : B[(size_t)k * ldb + j], acc);
                float o = alpha * acc;
                if (bias) o += bias[j];
                else if (Z) o *= Z[(size_t)i * ldz + j] > 0.f ? 1.f : slope;
                else if (beta != 0.f) o = std::fmaf(beta, C[(size_t)i * ldc + j], o);
                C[(size_t)i * ldc + j] = o;
            }
    });
}
void mggcn_gemm_f32(mggcn_stream_t s, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, float alpha, const float *A, size_t lda,
                    const float *B, size_t ldb, float beta, float *C, size_t ldc, void *, size_t) {
    gemm_any(s, ta, tb, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, nullptr, nullptr, 0, 0.f);
}
void mggcn_gemm_bias_f32(mggcn_stream_t s, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, float alpha, const float *A, size_t lda,
                         const float *B, size_t ldb, const float *bias, float *C, size_t ldc, void *, size_t) {
    gemm_any(s, ta, tb, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, bias, nullptr, 0, 0.f);
}
void mggcn_gemm_lrelu_bwd_f32(mggcn_stream_t s, int ta, int tb, uint32_t M, uint32_t N, uint32_t K, float alpha, const float *A, size_t lda,
                              const float *B, size_t ldb, const float *Z, size_t ldz, float slope, float *C, size_t ldc, void *, size_t) {
    gemm_any(s, ta, tb, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, nullptr, Z, ldz, slope);
}
void mggcn_gemm_tn_colsum_f32(mggcn_stream_t s, uint32_t M, uint32_t N, uint32_t K, float alpha, const float *A, size_t lda, const float *B,
                              size_t ldb, float *C, size_t ldc, float *colsum, void *, size_t) {
    gemm_any(s, 1, 0, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, nullptr, nullptr, 0, 0.f);
    launch(s, [=] {
        for (uint32_t j = 0; j < N; j++) {
            float acc = 0.f;
            for (uint32_t k = 0; k < K; k++) acc += B[(size_t)k * ldb + j];
            colsum[j] = alpha * acc;
        }
    });
}

// ---- element-wise / row kernels --------------------------------------------------------------------
void mggcn_leaky_relu_forward_f32(mggcn_stream_t s, const float *in, float *out, size_t n, float a) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) out[i] = lrelu(in[i], a); });
}
void mggcn_leaky_relu_backward_f32(mggcn_stream_t s, const float *in, const float *G_in, float *G_out, size_t n, float a) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) G_out[i] = in[i] > 0.f ? G_in[i] : a * G_in[i]; });
}
void mggcn_broadcast_rows_f32(mggcn_stream_t s, const float *row, float *mat, size_t n, size_t m, int discard) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) mat[i] = discard ? row[i % m] : mat[i] + row[i % m]; });
}
void mggcn_scale_rows_f32(mggcn_stream_t s, float *mat, const float *scalar, size_t n, size_t m) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) mat[i] /= scalar[i / m]; });
}
void mggcn_max_rows_f32(mggcn_stream_t s, const float *mat, float *maxs, size_t n, size_t m) {
    launch(s, [=] {
        for (size_t r = 0; r < n / m; r++) {
            float b = mat[r * m];
            for (size_t c = 1; c < m; c++) b = std::max(b, mat[r * m + c]);
            maxs[r] = b;
        }
    });
}
void mggcn_max_row_indices_f32(mggcn_stream_t s, const float *mat, int32_t *maxs, size_t n, size_t m) {
    launch(s, [=] {
        for (size_t r = 0; r < n / m; r++) {
            size_t b = 0;
            for (size_t c = 1; c < m; c++) if (mat[r * m + c] > mat[r * m + b]) b = c;
            maxs[r] = (int32_t)b;
        }
    });
}
void mggcn_index_log_rows_f32(mggcn_stream_t s, const float *mat, const int32_t *idx, float *values, size_t n, size_t m) {
    launch(s, [=] { for (size_t r = 0; r < n / m; r++) values[r] = std::log(mat[r * m + (size_t)idx[r]]); });
}
void mggcn_add_indexed_rows_f32(mggcn_stream_t s, float *mat, const int32_t *idx, float alpha, size_t n, size_t m) {
    launch(s, [=] { for (size_t r = 0; r < n / m; r++) mat[r * m + (size_t)idx[r]] += alpha; });
}
void mggcn_is_equal_i32(mggcn_stream_t s, const int32_t *a, const int32_t *b, float *out, size_t n) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) out[i] = a[i] == b[i] ? 1.f : 0.f; });
}
void mggcn_subtract_rows_exp_f32(mggcn_stream_t s, const float *mat, const float *scalar, float *out, size_t n, size_t m) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) out[i] = std::exp(mat[i] - scalar[i / m]); });
}
void mggcn_axpby_f32(mggcn_stream_t s, const float *A, float *B, float alpha, float beta, size_t n) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) B[i] = alpha * A[i] + beta * B[i]; });
}
void mggcn_aaxpby_f32(mggcn_stream_t s, const float *A, float *B, float alpha, float beta, size_t n) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) B[i] = alpha * A[i] * A[i] + beta * B[i]; });
}
void mggcn_adam_final_f32(mggcn_stream_t s, float *p, const float *m, const float *v, float lr, float c1, float c2, float eps, size_t n) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) p[i] -= (lr / c1) * m[i] / (std::sqrt(v[i] / c2) + eps); });
}
void mggcn_axpy_f32(mggcn_stream_t s, const float *A, float *B, float alpha, size_t n) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) B[i] += alpha * A[i]; });
}
void mggcn_scale_mat_f32(mggcn_stream_t s, float *mat, float x, size_t n) {
    launch(s, [=] { for (size_t i = 0; i < n; i++) mat[i] *= x; });
}
void mggcn_abssum_f32(mggcn_stream_t s, const float *A, size_t n, float *result) {
    launch(s, [=] { float acc = 0.f; for (size_t i = 0; i < n; i++) acc += std::fabs(A[i]); *result = acc; });
}
void mggcn_gather_rows_f32(mggcn_stream_t s, const float *src, size_t ld_src, const uint32_t *idx, size_t k, uint32_t d, float *dst, size_t ld_dst) {
    launch(s, [=] { for (size_t r = 0; r < k; r++) std::memcpy(dst + r * ld_dst, src + (size_t)idx[r] * ld_src, d * sizeof(float)); });
}
void mggcn_softmax_xent_fused_from_f32(mggcn_stream_t s, const float *logits, float *G, const int32_t *Y, size_t n_rows, size_t m,
                                       float grad_scale, float *sums) {
    launch(s, [=] {
        float loss = 0.f, correct = 0.f;
        std::vector<float> o(m);
        for (size_t r = 0; r < n_rows; r++) {
            const float *h = logits + r * m;
            size_t arg = 0;
            for (size_t c = 1; c < m; c++) if (h[c] > h[arg]) arg = c;
            float sum = 0.f;
            for (size_t c = 0; c < m; c++) { o[c] = std::exp(h[c] - h[arg]); sum += o[c]; }
            const size_t y = (size_t)Y[r];
            loss += std::fabs(std::log(o[y] / sum));
            correct += y == arg ? 1.f : 0.f;
            for (size_t c = 0; c < m; c++) G[r * m + c] = (o[c] / sum - (c == y ? 1.f : 0.f)) * grad_scale;
        }
        sums[0] += loss;
        sums[1] += correct;
    });
}
void mggcn_softmax_xent_fused_f32(mggcn_stream_t s, float *H, const int32_t *Y, size_t n_rows, size_t m, float grad_scale, float *sums) {
    mggcn_softmax_xent_fused_from_f32(s, H, H, Y, n_rows, m, grad_scale, sums);
}
static void adam_one(float *p, float *g, float *m, float *v, size_t n, float lr, float b1, float b2, float wd, float c1, float c2, float eps) {
    for (size_t i = 0; i < n; i++) {
        g[i] += wd * p[i];
        m[i] = b1 * m[i] + (1.f - b1) * g[i];
        v[i] = b2 * v[i] + (1.f - b2) * g[i] * g[i];
        p[i] -= (lr / c1) * m[i] / (std::sqrt(v[i] / c2) + eps);
    }
}
void mggcn_adam_fused_f32(mggcn_stream_t s, float *p, float *g, float *m, float *v, float lr, float b1, float b2, float wd, float c1,
                          float c2, float eps, size_t n) {
    launch(s, [=] { adam_one(p, g, m, v, n, lr, b1, b2, wd, c1, c2, eps); });
}
uint32_t mggcn_adam_multi_blocks(uint64_t size) { return (uint32_t)std::max<uint64_t>(1, (size + 1023) / 1024); }
void mggcn_adam_multi_f32(mggcn_stream_t s, const mggcn_adam_tensor *table, uint32_t n_tensors, uint32_t, float lr, float b1, float b2,
                          float c1, float c2, float eps) {
    launch(s, [=] {
        for (uint32_t t = 0; t < n_tensors; t++)
            adam_one(table[t].param, table[t].grad, table[t].m, table[t].v, (size_t)table[t].size, lr, b1, b2, table[t].weight_decay, c1, c2, eps);
    });
}

}  // extern "C"
