// gemm_timeline.hip -- where a wave of gemm_mfma_kernel spends its cycles.
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I mg-gcn_amd/csrc profiles/experiments/gemm_timeline.hip \
//         -L mg-gcn_amd/lib -lmggcn_hip -Wl,-rpath,$PWD/mg-gcn_amd/lib -o profiles/experiments/_proto/gemm_timeline     (see gemm_timeline.sh)
//
// Includes the PRODUCT kernel source with its measurement hooks defined: lane 0 of every wave of a few workgroups
// writes (s_memtime, phase id) pairs; the host prints the mean cycles between consecutive phase marks of the K loop
//   1 step top | 2 global loads of tile t+2 issued | 3 first 8 k-pairs of MFMAs issued | 4 tile t+1 written to LDS |
//   5 last 8 k-pairs issued | 6 past the workgroup barrier
// s_memtime returns through lgkmcnt, so every mark also drains the wave's outstanding LDS reads -- the marks sit where
// the kernel waits for them anyway (before the MFMAs that consume them) except 2 and 4.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#include <cmath>

#ifndef NO_MARKS
constexpr int kPhases = 10, kStampBlocks = 64, kStampStride = 29;
__device__ unsigned long long *g_stamps;       // [marked wave][phase id] -> cycles spent reaching that mark, [..][kPhases + id] -> visits
// The marks accumulate in scalar registers (s_memtime is wave-uniform) and are written once, at mark 8.
#define MGGCN_GEMM_STAMP_DECL                                                                                              \
    unsigned long long st_sum[kPhases] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; unsigned st_cnt[kPhases] = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0}; \
    unsigned long long st_prev = 0;                                                                                         \
    const bool stamp_on = blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x % kStampStride == 0 && blockIdx.x / kStampStride < kStampBlocks
#define MGGCN_GEMM_STAMP(id)                                                                                              \
    do {                                                                                                                  \
        const unsigned long long st_now = __builtin_readcyclecounter();                                                    \
        st_sum[id] += st_now - st_prev; st_cnt[id]++; st_prev = st_now;                                                     \
        if ((id) == 8 && stamp_on && (threadIdx.x & 63) == 0) {                                                            \
            unsigned long long *o = g_stamps + ((blockIdx.x / kStampStride) * 8 + (threadIdx.x >> 6)) * 2 * kPhases;       \
            for (int q = 0; q < kPhases; q++) { o[q] = st_sum[q]; o[kPhases + q] = st_cnt[q]; }                            \
        }                                                                                                                 \
    } while (0)

#else
constexpr int kPhases = 10, kStampBlocks = 64;
__device__ unsigned long long *g_stamps;
#endif

#include "gemm.hip"

int main(int argc, char **argv) {
    const uint32_t M = argc > 1 ? atoi(argv[1]) : 232968, K = argc > 2 ? atoi(argv[2]) : 608, N = argc > 5 ? atoi(argv[5]) : 128;
    const int trans_a = argc > 3 ? atoi(argv[3]) : 0;           // 1: A is stored [K x M] (the X^T G product)
    float *A, *B, *C;
    const size_t na = (size_t)M * K, nb = (size_t)K * N, nc = (size_t)M * N;
    hipMalloc(&A, na * 4); hipMalloc(&B, nb * 4); hipMalloc(&C, nc * 4);
    {   // random operands: with all-zero inputs the same kernel runs ~20 % faster (344 vs 424 us on [n x 608].[608 x 128]) --
        // the matrix pipe's timing does not depend on the data, its power draw and hence the sustained clock do
        std::vector<float> h(std::max(na, nb));
        unsigned long long x = 88172645463325252ull;
        for (auto &v : h) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; v = (float)((double)(x >> 11) / 9007199254740992.0 * 2.0 - 1.0); }
        if (getenv("GEMM_ZEROS")) std::fill(h.begin(), h.end(), 0.f);
        if (getenv("GEMM_NORMAL"))                          // standard normal (Box-Muller), what gemm_shapes.py feeds
            for (size_t i = 0; i + 1 < h.size(); i += 2) {
                const double u = (h[i] + 1.0) * 0.5 + 1e-12, w = (h[i + 1] + 1.0) * 3.14159265358979;
                const double r = std::sqrt(-2.0 * std::log(u));
                h[i] = (float)(r * std::cos(w)); h[i + 1] = (float)(r * std::sin(w));
            }
        const size_t b_off = getenv("GEMM_NORMAL") ? (std::max(na, nb) - nb) : 0;   // B from the other end of the stream
        hipMemcpy(A, h.data(), na * 4, hipMemcpyHostToDevice); hipMemcpy(B, h.data() + b_off, nb * 4, hipMemcpyHostToDevice);
    }
    unsigned long long *d_st; const size_t ns = (size_t)kStampBlocks * 8 * 2 * kPhases;
    hipMalloc(&d_st, ns * 8); hipMemset(d_st, 0, ns * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), &d_st, sizeof(d_st));
    const size_t wsb = mggcn_gemm_workspace_bytes(trans_a, 0, M, N, K);
    void *ws = nullptr; if (wsb) hipMalloc(&ws, wsb);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = argc > 4 ? atoi(argv[4]) : 1;
    for (int rep = 0; rep < 3; rep++) {
        hipMemset(d_st, 0, ns * 8);
        hipEventRecord(e0);
        for (int q = 0; q < reps; q++)
            mggcn_gemm_f32(nullptr, trans_a, 0, M, N, K, 1.f, A, trans_a ? M : K, B, N, 0.f, C, N, ws, wsb);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= reps;
    printf("M=%u N=%u K=%u trans_a=%d: %.1f us per call (%s)\n", M, N, K, trans_a, ms * 1e3, hipGetErrorString(hipGetLastError()));
#ifdef NO_MARKS
    return 0;
#endif
    std::vector<unsigned long long> st(ns);
    hipMemcpy(st.data(), d_st, ns * 8, hipMemcpyDeviceToHost);
    static const char *what[kPhases] = {"(kernel entry)", "loop overhead after the barrier -> step top", "address math + global loads of tile t+2 issued",
        "k-pairs 0-7: LDS reads + 16 MFMAs issued", "tile t+1: wait for its loads, k-edge mask, ds_writes issued", "k-pairs 8-15: LDS reads + 16 MFMAs issued",
        "workgroup barrier", "loop exit", "epilogue (C stores issued)", "prologue: tiles 0 and 1 requested, tile 0 landed and in LDS, barrier"};
    double sum[kPhases] = {0}, cnt[kPhases] = {0}; long waves = 0;
    for (size_t w = 0; w < (size_t)kStampBlocks * 8; w++) {
        const unsigned long long *o = &st[w * 2 * kPhases];
        if (!o[kPhases + 8]) continue;
        waves++;
        for (int q = 1; q < kPhases; q++) { sum[q] += (double)o[q]; cnt[q] += (double)o[kPhases + q]; }
    }
    double total = 0; for (int q = 1; q < kPhases; q++) total += sum[q];
    printf("%ld waves marked; mean cycles from kernel entry to the last mark: %.0f\n", waves, total / waves);
    for (int q = 1; q < kPhases; q++)
        printf("  -> %d : mean %8.0f cycles x %5.1f per wave = %5.1f %% of the wave   %s\n", q, cnt[q] ? sum[q] / cnt[q] : 0.0, cnt[q] / waves,
               100.0 * sum[q] / total, what[q]);
    return 0;
}
