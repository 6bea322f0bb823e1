// elementwise.hip -- the element-wise / row / loss / optimiser kernels of one epoch.
//
// One entry point per live launcher of the reference's device TU
// (src/cuda_utils.cu:229-390, kernels :12-227).  The reference launches
// min(ceil(size/1024),1280) x 1024 threads and walks rows with ONE THREAD PER ROW
// (serial, uncoalesced loop over the m columns: max_rows, max_row_indices, ...).
// Here: streaming kernels move 16 B per lane (float4) when the buffer allows it,
// row kernels give each row to a group of lanes inside a wave64 and reduce with
// DPP/LDS-crossbar shuffles, grids are sized for 256 CUs.  All of these are HBM
// passes over [n x m] fp32; none is reshaped into a GEMM.
#include <algorithm>

#include <map>
#include <vector>
#include <type_traits>
#include <mutex>
#include <utility>

#include "common.h"

namespace {

__device__ __forceinline__ float lrelu(float x, float slope) {
    const float y = slope * x;
    return x > y ? x : y;
}

// ---- streaming helpers ----------------------------------------------------
template <typename F>
__global__ __launch_bounds__(256) void map1_kernel(const float *__restrict__ in, float *__restrict__ out,
                                                   size_t size, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride) out[i] = f(in[i]);
}

template <typename F>
__global__ __launch_bounds__(256) void map1_vec4_kernel(const float4 *__restrict__ in,
                                                        float4 *__restrict__ out, size_t size4, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size4; i += stride) {
        float4 x = in[i];
        x.x = f(x.x); x.y = f(x.y); x.z = f(x.z); x.w = f(x.w);
        out[i] = x;
    }
}

template <typename F>
__global__ __launch_bounds__(256) void map2_kernel(const float *__restrict__ a, const float *__restrict__ b,
                                                   float *__restrict__ out, size_t size, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride)
        out[i] = f(a[i], b[i]);
}

template <typename F>
__global__ __launch_bounds__(256) void map2_vec4_kernel(const float4 *__restrict__ a,
                                                        const float4 *__restrict__ b,
                                                        float4 *__restrict__ out, size_t size4, F f) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size4; i += stride) {
        const float4 x = a[i], y = b[i];
        float4 o;
        o.x = f(x.x, y.x); o.y = f(x.y, y.y); o.z = f(x.z, y.z); o.w = f(x.w, y.w);
        out[i] = o;
    }
}

template <typename F>
void launch_map1(hipStream_t st, const float *in, float *out, size_t size, F f) {
    if (!size) return;
    if (size % 4 == 0 && aligned16(in) && aligned16(out)) {
        hipLaunchKernelGGL(map1_vec4_kernel<F>, dim3(stream_grid(size / 4)), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(in), reinterpret_cast<float4 *>(out), size / 4, f);
    } else {
        hipLaunchKernelGGL(map1_kernel<F>, dim3(stream_grid(size)), dim3(256), 0, st, in, out, size, f);
    }
    MGGCN_CHECK_LAUNCH();
}

template <typename F>
void launch_map2(hipStream_t st, const float *a, const float *b, float *out, size_t size, F f) {
    if (!size) return;
    if (size % 4 == 0 && aligned16(a) && aligned16(b) && aligned16(out)) {
        hipLaunchKernelGGL(map2_vec4_kernel<F>, dim3(stream_grid(size / 4)), dim3(256), 0, st,
                           reinterpret_cast<const float4 *>(a), reinterpret_cast<const float4 *>(b),
                           reinterpret_cast<float4 *>(out), size / 4, f);
    } else {
        hipLaunchKernelGGL(map2_kernel<F>, dim3(stream_grid(size)), dim3(256), 0, st, a, b, out, size, f);
    }
    MGGCN_CHECK_LAUNCH();
}

struct LreluFwd { float s; __device__ float operator()(float x) const { return lrelu(x, s); } };
struct LreluBwd { float s; __device__ float operator()(float in, float g) const { return in > 0.f ? g : s * g; } };
struct Axpby { float a, b; __device__ float operator()(float x, float y) const { return a * x + b * y; } };
struct Aaxpby { float a, b; __device__ float operator()(float x, float y) const { return a * x * x + b * y; } };
struct Axpy { float a; __device__ float operator()(float x, float y) const { return fmaf(a, x, y); } };
struct Scal { float a; __device__ float operator()(float x) const { return x * a; } };

// ---- row-indexed streaming kernels ----------------------------------------
// mat[i] (= | +=) row[i % m]          reference src/cuda_utils.cu:40-51
__global__ __launch_bounds__(256) void broadcast_rows_kernel(const float *__restrict__ row,
                                                             float *__restrict__ mat, size_t size,
                                                             size_t m, int discard) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride) {
        const float r = row[i % m];
        mat[i] = discard ? r : mat[i] + r;
    }
}

// mat[i] /= scalar[i / m]             reference src/cuda_utils.cu:75-79
__global__ __launch_bounds__(256) void scale_rows_kernel(float *__restrict__ mat,
                                                         const float *__restrict__ scalar, size_t size,
                                                         size_t m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride)
        mat[i] /= scalar[i / m];
}

// out[i] = exp(mat[i] - scalar[i / m]) reference src/cuda_utils.cu:192-200
__global__ __launch_bounds__(256) void subtract_rows_exp_kernel(const float *__restrict__ mat,
                                                                const float *__restrict__ scalar,
                                                                float *__restrict__ out, size_t size,
                                                                size_t m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride)
        out[i] = expf(mat[i] - scalar[i / m]);
}

// ---- one wave64 per row ---------------------------------------------------
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v = fmaxf(v, __shfl_xor(v, off));
    return v;
}
__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int off = 32; off; off >>= 1) v += __shfl_xor(v, off);
    return v;
}

// row max                              reference src/cuda_utils.cu:95-104
__global__ __launch_bounds__(256) void max_rows_kernel(const float *__restrict__ mat,
                                                       float *__restrict__ maxs, size_t n_rows, size_t m) {
    const int lane = threadIdx.x & 63;
    const size_t wstride = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n_rows; r += wstride) {
        float mx = -INFINITY;
        for (size_t c = lane; c < m; c += 64) mx = fmaxf(mx, mat[r * m + c]);
        mx = wave_max(mx);
        if (lane == 0) maxs[r] = mx;
    }
}

// argmax, first maximum wins           reference src/cuda_utils.cu:119-133
// Wave-wide reductions on the DPP path (no LDS): four in-row steps (quad swaps, half-row and row mirrors) leave every
// lane of a 16-lane row with its row's result, four v_readlane bring the row results together.  hipcc lowers
// __shfl_xor to ds_bpermute_b32, an LDS-crossbar instruction: the fused loss kernel spent its time there (~30 per row).
template <typename Op>
__device__ __forceinline__ float wave_reduce_dpp(float v, Op op) {
    auto dpp = [](float x, auto ctrl) {
        return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = op(v, dpp(v, std::integral_constant<int, 0xB1>{}));     // quad_perm [1,0,3,2]
    v = op(v, dpp(v, std::integral_constant<int, 0x4E>{}));     // quad_perm [2,3,0,1]
    v = op(v, dpp(v, std::integral_constant<int, 0x141>{}));    // row_half_mirror
    v = op(v, dpp(v, std::integral_constant<int, 0x140>{}));    // row_mirror
    // (the builtin is typed int: a float argument would be CONVERTED, not re-interpreted)
    auto lane_f = [](float x, int l) { return __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x), l)); };
    const float r0 = lane_f(v, 0), r1 = lane_f(v, 16), r2 = lane_f(v, 32), r3 = lane_f(v, 48);
    return op(op(r0, r1), op(r2, r3));
}
__device__ __forceinline__ float wave_sum_dpp(float v) { return wave_reduce_dpp(v, [](float a, float b) { return a + b; }); }
__device__ __forceinline__ float wave_max_dpp(float v) { return wave_reduce_dpp(v, [](float a, float b) { return fmaxf(a, b); }); }

__device__ __forceinline__ void argmax_combine(float &v, uint32_t &i, float ov, uint32_t oi) {
    if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; }
}
__global__ __launch_bounds__(256) void max_row_indices_kernel(const float *__restrict__ mat,
                                                              int32_t *__restrict__ maxs, size_t n_rows,
                                                              size_t m) {
    const int lane = threadIdx.x & 63;
    const size_t wstride = ((size_t)gridDim.x * blockDim.x) >> 6;
    for (size_t r = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r < n_rows; r += wstride) {
        float mx = -INFINITY;
        uint32_t idx = 0xFFFFFFFFu;
        for (size_t c = lane; c < m; c += 64) {
            const float x = mat[r * m + c];
            if (x > mx) { mx = x; idx = (uint32_t)c; }   // strict >: earlier column of this lane wins
        }
#pragma unroll
        for (int off = 32; off; off >>= 1) {
            const float ov = __shfl_xor(mx, off);
            const uint32_t oi = __shfl_xor(idx, off);
            argmax_combine(mx, idx, ov, oi);
        }
        // all -inf / NaN row: the reference's strict `max < x` never fires -> index 0
        if (lane == 0) maxs[r] = idx == 0xFFFFFFFFu ? 0 : (int32_t)idx;
    }
}

// values[r] = log(mat[r, indices[r]])  reference src/cuda_utils.cu:142-150
__global__ __launch_bounds__(256) void index_log_rows_kernel(const float *__restrict__ mat,
                                                             const int32_t *__restrict__ indices,
                                                             float *__restrict__ values, size_t n_rows,
                                                             size_t m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride)
        values[r] = logf(mat[r * m + (size_t)indices[r]]);
}

// mat[r, indices[r]] += alpha           reference src/cuda_utils.cu:159-164
__global__ __launch_bounds__(256) void add_indexed_rows_kernel(float *__restrict__ mat,
                                                               const int32_t *__restrict__ indices,
                                                               float alpha, size_t n_rows, size_t m) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t r = (size_t)blockIdx.x * blockDim.x + threadIdx.x; r < n_rows; r += stride)
        mat[r * m + (size_t)indices[r]] += alpha;
}

// out[i] = (a[i] == b[i])               reference src/cuda_utils.cu:180-184
__global__ __launch_bounds__(256) void is_equal_kernel(const int32_t *__restrict__ a,
                                                       const int32_t *__restrict__ b,
                                                       float *__restrict__ out, size_t size) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride)
        out[i] = (float)(a[i] == b[i]);
}

// param -= step * m / (sqrt(v / c2) + eps)   reference src/cuda_utils.cu:208-218
__global__ __launch_bounds__(256) void adam_final_kernel(float *__restrict__ param,
                                                         const float *__restrict__ m,
                                                         const float *__restrict__ v, float step, float c2,
                                                         float eps, size_t size) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride)
        param[i] -= step * m[i] / (sqrtf(v[i] / c2) + eps);
}

// ---- |x| sum: fixed-order two-level reduction (reproducible) ---------------
constexpr unsigned kAsumBlocks = 1024;
constexpr unsigned kXentBlocks = kNumCU * 8;               // grid cap of the fused loss (stream_grid)
constexpr unsigned kScratchFloats = 2 * kXentBlocks;        // abssum partials / the fused loss's (loss, correct) pairs

__global__ __launch_bounds__(256) void abssum_partial_kernel(const float *__restrict__ A, size_t size,
                                                             float *__restrict__ partial) {
    __shared__ float wsum[4];
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    float s = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride) s += fabsf(A[i]);
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) partial[blockIdx.x] = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

__global__ __launch_bounds__(256) void abssum_final_kernel(const float *__restrict__ partial, unsigned n,
                                                           float *__restrict__ result) {
    __shared__ float wsum[4];
    float s = 0.f;
    for (unsigned i = threadIdx.x; i < n; i += 256) s += partial[i];
    s = wave_sum(s);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) *result = (wsum[0] + wsum[1]) + (wsum[2] + wsum[3]);
}

// scratch for the abssum partials: one per (device, stream), allocated on first use (never on a captured
// launch path: the host layers call abssum once during warm-up).  Per STREAM, not per device: several
// contexts may drive one GPU at once (a dist_context whose ranks share a device, two models on two
// streams) and two sums in flight on different streams must not share their partials.
std::mutex g_scratch_mu;
std::map<std::pair<int, hipStream_t>, float *> g_scratch;

float *abssum_scratch(hipStream_t st) {
    int dev = 0;
    MGGCN_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lock(g_scratch_mu);
    float *&p = g_scratch[{dev, st}];
    if (!p) MGGCN_CHECK_HIP(hipMalloc(&p, kScratchFloats * sizeof(float)));
    return p;
}

// ---- fused softmax + cross-entropy + argmax + gradient ---------------------
// One wave64 per row; the row's m <= 64*K logits live in K registers per lane.
constexpr int kXentMaxPerLane = 16;  // m <= 1024

// R rows are in flight per wave (their loads issued together): with one row at a time every row is a dependent
// HBM round trip -- load, six shuffle steps, store -- and the pass was latency-bound at 0.7 TB/s (r01: 110 us for
// the 76 MB of the [233 k x 41] logits); four rows in flight hide it.
// (src and dst may be the same matrix: a row is read whole before any of it is written)
template <int K, int R>
__global__ __launch_bounds__(256) void softmax_xent_fused_kernel(const float *src, float *dst,
                                                                 const int32_t *__restrict__ Y,
                                                                 size_t n_rows, size_t m, float grad_scale,
                                                                 float *__restrict__ partials) {
    __shared__ float s_loss[4], s_acc[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const size_t wstride = ((size_t)gridDim.x * blockDim.x) >> 6;
    float loss_acc = 0.f, corr_acc = 0.f;
    for (size_t r0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 6; r0 < n_rows; r0 += wstride * R) {
        float x[R][K];
        int32_t y[R];
#pragma unroll
        for (int q = 0; q < R; q++) {                     // all loads first
            const size_t r = r0 + (size_t)q * wstride;
            const bool live = r < n_rows;                 // wave-uniform
            y[q] = live ? Y[r] : 0;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const size_t c = (size_t)lane + 64u * k;
                x[q][k] = (live && c < m) ? src[r * m + c] : -INFINITY;
            }
        }
#pragma unroll
        for (int q = 0; q < R; q++) {
            const size_t r = r0 + (size_t)q * wstride;
            if (r >= n_rows) break;                       // wave-uniform
            // row maximum (wave-uniform) and the FIRST column holding it (strict `<` of the reference, cuda_utils.cu:126)
            float mx = -INFINITY;
#pragma unroll
            for (int k = 0; k < K; k++) mx = fmaxf(mx, x[q][k]);
            mx = wave_max_dpp(mx);
            uint32_t idx = 0xFFFFFFFFu;
#pragma unroll
            for (int k = K - 1; k >= 0; k--) {
                const unsigned long long hit = __ballot(x[q][k] == mx && (size_t)lane + 64u * k < m);
                if (hit) idx = (uint32_t)__builtin_ctzll(hit) + 64u * (uint32_t)k;        // lower k = lower columns: last writer wins
            }
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < K; k++) {
                const size_t c = (size_t)lane + 64u * k;
                x[q][k] = c < m ? expf(x[q][k] - mx) : 0.f;
                sum += x[q][k];
            }
            sum = wave_sum_dpp(sum);
            float py = 0.f;                                    // p_y: read from the lane that holds column y
#pragma unroll
            for (int k = 0; k < K; k++) {
                const size_t c = (size_t)lane + 64u * k;
                if (c < m) {
                    const float o = x[q][k] / sum;
                    const bool hit = (int32_t)c == y[q];
                    dst[r * m + c] = (hit ? o - 1.f : o) * grad_scale;
                    x[q][k] = o;
                }
            }
            if (y[q] >= 0 && (size_t)y[q] < m) {               // wave-uniform
#pragma unroll
                for (int k = 0; k < K; k++)
                    if ((y[q] >> 6) == k)
                        py = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, x[q][k]), y[q] & 63));
            }
            if (lane == 0) {
                loss_acc += fabsf(logf(py));
                // argmax of the softmax output == argmax of the logits (exp is monotone);
                // a row whose maximum never beat -inf reports index 0 like the reference
                corr_acc += ((idx == 0xFFFFFFFFu ? 0 : (int32_t)idx) == y[q]) ? 1.f : 0.f;
            }
        }
    }
    if (lane == 0) { s_loss[wid] = loss_acc; s_acc[wid] = corr_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {              // one (loss, correct) pair per workgroup, summed by xent_final_kernel
        partials[2 * blockIdx.x + 0] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        partials[2 * blockIdx.x + 1] = (s_acc[0] + s_acc[1]) + (s_acc[2] + s_acc[3]);
    }
}

// m <= 64 (the logits layer: 41 classes, 48 at P = 8): one row per 16-LANE GROUP, KE = ceil(m / 16) logits per lane, so a
// wave works on four rows at once and every reduction is four DPP rotations inside the 16-lane row (row_ror 8, 4, 2, 1:
// a butterfly -- both lanes of a pair add the same two numbers, so all 16 lanes end with the same bits), no v_readlane,
// no cross-row step.  The wave-per-row form above keeps 41 of 64 lanes busy and spends a full wave reduction per row:
// 105 us for the [233 k x 41] logits (0.7 TB/s); this one issues a quarter of the instructions per row.
template <typename T, typename Op>
__device__ __forceinline__ T row16_reduce(T v, Op op) {
    auto ror = [](T x, auto ctrl) {
        return __builtin_bit_cast(T, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), decltype(ctrl)::value, 0xF, 0xF, true));
    };
    v = op(v, ror(v, std::integral_constant<int, 0x128>{}));    // row_ror:8
    v = op(v, ror(v, std::integral_constant<int, 0x124>{}));    // row_ror:4
    v = op(v, ror(v, std::integral_constant<int, 0x122>{}));    // row_ror:2
    v = op(v, ror(v, std::integral_constant<int, 0x121>{}));    // row_ror:1
    return v;
}

template <int KE, int R>
__global__ __launch_bounds__(256) void softmax_xent_rows16_kernel(const float *src, float *dst,
                                                                  const int32_t *__restrict__ Y, size_t n_rows,
                                                                  uint32_t m, float grad_scale, float *__restrict__ partials) {
    __shared__ float s_loss[4], s_acc[4];
    const int lane = threadIdx.x & 63, wid = threadIdx.x >> 6;
    const uint32_t sub = lane & 15;
    const size_t gstride = ((size_t)gridDim.x * blockDim.x) >> 4;           // 16-lane groups in the grid
    const size_t g0 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) >> 4;
    const size_t w0 = g0 & ~(size_t)3;                                       // first group of this wave
    float loss_acc = 0.f, corr_acc = 0.f;
    for (size_t base = 0; w0 + base < n_rows; base += gstride * R) {         // wave-uniform trip count
        float x[R][KE];
        int32_t y[R];
#pragma unroll
        for (int q = 0; q < R; q++) {                                        // all loads first
            const size_t r = g0 + base + (size_t)q * gstride;
            const bool live = r < n_rows;
            y[q] = live ? Y[r] : 0;
#pragma unroll
            for (int k = 0; k < KE; k++) {
                const uint32_t c = sub + 16u * k;
                x[q][k] = (live && c < m) ? src[r * m + c] : -INFINITY;
            }
        }
#pragma unroll
        for (int q = 0; q < R; q++) {
            const size_t r = g0 + base + (size_t)q * gstride;
            const bool live = r < n_rows;
            float mx = x[q][0];
#pragma unroll
            for (int k = 1; k < KE; k++) mx = fmaxf(mx, x[q][k]);
            mx = row16_reduce(mx, [](float a, float b) { return fmaxf(a, b); });
            // FIRST column holding the maximum (strict `<` of the reference, cuda_utils.cu:126); none (all -inf / NaN) -> 0
            uint32_t idx = 0xFFFFFFFFu;
#pragma unroll
            for (int k = KE - 1; k >= 0; k--)
                if (x[q][k] == mx && sub + 16u * k < m) idx = sub + 16u * k;
            idx = row16_reduce(idx, [](uint32_t a, uint32_t b) { return a < b ? a : b; });
            float sum = 0.f;
#pragma unroll
            for (int k = 0; k < KE; k++) {
                x[q][k] = sub + 16u * k < m ? expf(x[q][k] - mx) : 0.f;
                sum += x[q][k];
            }
            sum = row16_reduce(sum, [](float a, float b) { return a + b; });
            float py = 0.f;                                                  // p_y: exactly one lane / slot holds column y
#pragma unroll
            for (int k = 0; k < KE; k++) {
                const uint32_t c = sub + 16u * k;
                const float o = x[q][k] / sum;
                const bool hit = (int32_t)c == y[q];
                if (live && c < m) dst[r * m + c] = (hit ? o - 1.f : o) * grad_scale;
                if (hit && c < m) py = o;
            }
            py = row16_reduce(py, [](float a, float b) { return a + b; });   // the others are exact zeros
            if (live && sub == 0) {
                loss_acc += fabsf(logf(py));
                corr_acc += ((idx == 0xFFFFFFFFu ? 0 : (int32_t)idx) == y[q]) ? 1.f : 0.f;
            }
        }
    }
    loss_acc = wave_sum_dpp(loss_acc);
    corr_acc = wave_sum_dpp(corr_acc);
    if (lane == 0) { s_loss[wid] = loss_acc; s_acc[wid] = corr_acc; }
    __syncthreads();
    if (threadIdx.x == 0) {              // one (loss, correct) pair per workgroup, summed by xent_final_kernel
        partials[2 * blockIdx.x + 0] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        partials[2 * blockIdx.x + 1] = (s_acc[0] + s_acc[1]) + (s_acc[2] + s_acc[3]);
    }
}

// sums[0] += sum of the workgroups' loss terms, sums[1] += their correct counts, in a fixed order.  (First version: two
// float atomics per workgroup on the same two addresses -- 4096 device-scope read-modify-writes in a row were most of the
// pass: 61 us for any m <= 64; and the two scalars depended on arrival order in their last bits.)
__global__ __launch_bounds__(256) void xent_final_kernel(const float *__restrict__ partials, unsigned n_blocks,
                                                         float *__restrict__ sums) {
    __shared__ float w[2][4];
    float l = 0.f, a = 0.f;
    for (unsigned i = threadIdx.x; i < n_blocks; i += 256) { l += partials[2 * i]; a += partials[2 * i + 1]; }
    l = wave_sum(l); a = wave_sum(a);
    if ((threadIdx.x & 63) == 0) { w[0][threadIdx.x >> 6] = l; w[1][threadIdx.x >> 6] = a; }
    __syncthreads();
    if (threadIdx.x == 0) {
        sums[0] += (w[0][0] + w[0][1]) + (w[0][2] + w[0][3]);
        sums[1] += (w[1][0] + w[1][1]) + (w[1][2] + w[1][3]);
    }
}

// ---- fused Adam -------------------------------------------------------------
__global__ __launch_bounds__(256) void adam_fused_kernel(float *__restrict__ p, float *__restrict__ g,
                                                         float *__restrict__ m, float *__restrict__ v,
                                                         float step, float b1, float b2, float wd, float c2,
                                                         float eps, size_t size) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < size; i += stride) {
        const float pi = p[i];
        const float gi = fmaf(wd, pi, g[i]);                 // axpy(W, G_W, wd)     gcn.hpp:163
        const float mi = (1.f - b1) * gi + b1 * m[i];        // axpby                gcn.hpp:164
        const float vi = (1.f - b2) * gi * gi + b2 * v[i];   // aaxpby               gcn.hpp:166
        g[i] = gi; m[i] = mi; v[i] = vi;
        p[i] = pi - step * mi / (sqrtf(vi / c2) + eps);      // adam_final           gcn.hpp:168
    }
}

// one launch for EVERY parameter tensor of the model: block b works on 1024 elements of the tensor whose
// [first_block, first_block + blocks) range holds b (a model has ~8 tensors: linear search)
__device__ __forceinline__ void adam_one(float *p, float *g, float *m, float *v, size_t i, float step, float b1,
                                         float b2, float wd, float c2, float eps) {
    const float pi = p[i];
    const float gi = fmaf(wd, pi, g[i]);
    const float mi = (1.f - b1) * gi + b1 * m[i];
    const float vi = (1.f - b2) * gi * gi + b2 * v[i];
    g[i] = gi; m[i] = mi; v[i] = vi;
    p[i] = pi - step * mi / (sqrtf(vi / c2) + eps);
}

__global__ __launch_bounds__(256) void adam_multi_kernel(const mggcn_adam_tensor *__restrict__ table, uint32_t n_tensors,
                                                         float step, float b1, float b2, float c2, float eps) {
    uint32_t t = 0;
    while (t + 1 < n_tensors && table[t + 1].first_block <= blockIdx.x) t++;
    const mggcn_adam_tensor T = table[t];
    const size_t base = (size_t)(blockIdx.x - T.first_block) * 1024;
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const size_t i = base + (size_t)k * 256 + threadIdx.x;
        if (i < T.size) adam_one(T.param, T.grad, T.m, T.v, i, step, b1, b2, T.weight_decay, c2, eps);
    }
}

}  // namespace

// ============================ C ABI =========================================
MGGCN_API void mggcn_leaky_relu_forward_f32(mggcn_stream_t stream, const float *in, float *out,
                                            size_t size, float alpha) {
    launch_map1(as_stream(stream), in, out, size, LreluFwd{alpha});
}

MGGCN_API void mggcn_leaky_relu_backward_f32(mggcn_stream_t stream, const float *in, const float *G_in,
                                             float *G_out, size_t size, float alpha) {
    launch_map2(as_stream(stream), in, G_in, G_out, size, LreluBwd{alpha});
}

MGGCN_API void mggcn_broadcast_rows_f32(mggcn_stream_t stream, const float *row, float *mat, size_t size,
                                        size_t m, int discard) {
    if (!size) return;
    MGGCN_REQUIRE(m > 0, "row width must be positive");
    hipLaunchKernelGGL(broadcast_rows_kernel, dim3(stream_grid(size)), dim3(256), 0, as_stream(stream), row,
                       mat, size, m, discard);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_scale_rows_f32(mggcn_stream_t stream, float *mat, const float *scalar, size_t size,
                                    size_t m) {
    if (!size) return;
    MGGCN_REQUIRE(m > 0, "row width must be positive");
    hipLaunchKernelGGL(scale_rows_kernel, dim3(stream_grid(size)), dim3(256), 0, as_stream(stream), mat,
                       scalar, size, m);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_max_rows_f32(mggcn_stream_t stream, const float *mat, float *maxs, size_t size,
                                  size_t m) {
    if (!size) return;
    MGGCN_REQUIRE(m > 0 && size % m == 0, "size must be n_rows * m");
    const size_t n_rows = size / m;
    hipLaunchKernelGGL(max_rows_kernel, dim3(stream_grid(n_rows * 64)), dim3(256), 0, as_stream(stream), mat,
                       maxs, n_rows, m);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_max_row_indices_f32(mggcn_stream_t stream, const float *mat, int32_t *maxs,
                                         size_t size, size_t m) {
    if (!size) return;
    MGGCN_REQUIRE(m > 0 && size % m == 0, "size must be n_rows * m");
    const size_t n_rows = size / m;
    hipLaunchKernelGGL(max_row_indices_kernel, dim3(stream_grid(n_rows * 64)), dim3(256), 0,
                       as_stream(stream), mat, maxs, n_rows, m);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_index_log_rows_f32(mggcn_stream_t stream, const float *mat, const int32_t *indices,
                                        float *values, size_t size, size_t m) {
    if (!size) return;
    MGGCN_REQUIRE(m > 0 && size % m == 0, "size must be n_rows * m");
    const size_t n_rows = size / m;
    hipLaunchKernelGGL(index_log_rows_kernel, dim3(stream_grid(n_rows)), dim3(256), 0, as_stream(stream),
                       mat, indices, values, n_rows, m);
    MGGCN_CHECK_LAUNCH();
}

namespace {
// Halo pack: dst[k, :] = src[idx[k], :].  One wave64 per gathered row, float4 lanes where the
// pitches allow; a pure HBM stream (the rows a peer rank needs of this rank's shard).
__global__ __launch_bounds__(256) void gather_rows_kernel(const float *__restrict__ src, size_t ld_src,
                                                          const uint32_t *__restrict__ idx, size_t n_idx, uint32_t d,
                                                          float *__restrict__ dst, size_t ld_dst, bool vec) {
    const int lane = threadIdx.x & 63;
    const size_t waves = (size_t)gridDim.x * (blockDim.x >> 6);
    for (size_t k = (size_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6); k < n_idx; k += waves) {
        const float *s = src + (size_t)idx[k] * ld_src;
        float *o = dst + k * ld_dst;
        if (vec) {
            for (uint32_t c = lane * 4; c < d; c += 256)
                *reinterpret_cast<float4 *>(o + c) = *reinterpret_cast<const float4 *>(s + c);
        } else {
            for (uint32_t c = lane; c < d; c += 64) o[c] = s[c];
        }
    }
}
}  // namespace

MGGCN_API void mggcn_gather_rows_f32(mggcn_stream_t stream, const float *src, size_t ld_src, const uint32_t *indices,
                                     size_t n_indices, uint32_t d, float *dst, size_t ld_dst) {
    if (!n_indices || !d) return;
    MGGCN_REQUIRE(src && indices && dst && ld_src >= d && ld_dst >= d, "gather_rows: bad operand");
    const bool vec = d % 4 == 0 && ld_src % 4 == 0 && ld_dst % 4 == 0 && aligned16(src) && aligned16(dst);
    const size_t blocks = std::min<size_t>((n_indices + 3) / 4, (size_t)kNumCU * 16);
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, as_stream(stream), src, ld_src, indices,
                       n_indices, d, dst, ld_dst, vec);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_add_indexed_rows_f32(mggcn_stream_t stream, float *mat, const int32_t *indices,
                                          float alpha, size_t size, size_t m) {
    if (!size) return;
    MGGCN_REQUIRE(m > 0 && size % m == 0, "size must be n_rows * m");
    const size_t n_rows = size / m;
    hipLaunchKernelGGL(add_indexed_rows_kernel, dim3(stream_grid(n_rows)), dim3(256), 0, as_stream(stream),
                       mat, indices, alpha, n_rows, m);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_is_equal_i32(mggcn_stream_t stream, const int32_t *a, const int32_t *b, float *out,
                                  size_t size) {
    if (!size) return;
    hipLaunchKernelGGL(is_equal_kernel, dim3(stream_grid(size)), dim3(256), 0, as_stream(stream), a, b, out,
                       size);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_subtract_rows_exp_f32(mggcn_stream_t stream, const float *mat, const float *scalar,
                                           float *out, size_t size, size_t m) {
    if (!size) return;
    MGGCN_REQUIRE(m > 0, "row width must be positive");
    hipLaunchKernelGGL(subtract_rows_exp_kernel, dim3(stream_grid(size)), dim3(256), 0, as_stream(stream),
                       mat, scalar, out, size, m);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_axpby_f32(mggcn_stream_t stream, const float *A, float *B, float alpha, float beta,
                               size_t size) {
    launch_map2(as_stream(stream), A, B, B, size, Axpby{alpha, beta});
}

MGGCN_API void mggcn_aaxpby_f32(mggcn_stream_t stream, const float *A, float *B, float alpha, float beta,
                                size_t size) {
    launch_map2(as_stream(stream), A, B, B, size, Aaxpby{alpha, beta});
}

MGGCN_API void mggcn_adam_final_f32(mggcn_stream_t stream, float *param, const float *m, const float *v,
                                    float lr, float c1, float c2, float eps, size_t size) {
    if (!size) return;
    hipLaunchKernelGGL(adam_final_kernel, dim3(stream_grid(size)), dim3(256), 0, as_stream(stream), param, m,
                       v, lr / c1, c2, eps, size);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_axpy_f32(mggcn_stream_t stream, const float *A, float *B, float alpha, size_t size) {
    launch_map2(as_stream(stream), A, B, B, size, Axpy{alpha});
}

MGGCN_API void mggcn_scale_mat_f32(mggcn_stream_t stream, float *mat, float scalar, size_t size) {
    launch_map1(as_stream(stream), mat, mat, size, Scal{scalar});
}

// The reduction scratch above belongs to a (device, stream) pair: mggcn_stream_destroy releases it with the stream;
// a host layer whose streams come from elsewhere (torch) calls this when it drops a stream, so that a recycled
// stream handle never inherits a buffer another stream may still be using, and nothing accumulates.
MGGCN_API void mggcn_stream_release_scratch(mggcn_stream_t stream) {
    // by STREAM alone, whatever device is current on the calling thread (a stream handle belongs to one device; a context
    // may be dropped from a thread that has another device current -- the entry must still be found, or a recycled handle
    // inherits the buffer)
    std::vector<std::pair<int, float *>> mine;
    {
        std::lock_guard<std::mutex> lock(g_scratch_mu);
        for (auto it = g_scratch.begin(); it != g_scratch.end();)
            if (it->first.second == as_stream(stream)) { mine.push_back({it->first.first, it->second}); it = g_scratch.erase(it); }
            else ++it;
    }
    if (mine.empty()) return;
    int prev = 0;
    MGGCN_CHECK_HIP(hipGetDevice(&prev));
    for (const auto &m : mine) {
        MGGCN_CHECK_HIP(hipSetDevice(m.first));
        MGGCN_CHECK_HIP(hipStreamSynchronize(as_stream(stream)));      // nobody may still be summing into it
        MGGCN_CHECK_HIP(hipFree(m.second));
    }
    MGGCN_CHECK_HIP(hipSetDevice(prev));
}

MGGCN_API void mggcn_abssum_f32(mggcn_stream_t stream, const float *A, size_t size, float *result_device) {
    MGGCN_REQUIRE(result_device != nullptr, "null result pointer");
    float *scratch = abssum_scratch(as_stream(stream));
    const unsigned blocks = std::min<unsigned>(kAsumBlocks, stream_grid(size ? size : 1));
    hipLaunchKernelGGL(abssum_partial_kernel, dim3(blocks), dim3(256), 0, as_stream(stream), A, size, scratch);
    MGGCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(abssum_final_kernel, dim3(1), dim3(256), 0, as_stream(stream), scratch, blocks,
                       result_device);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_softmax_xent_fused_from_f32(mggcn_stream_t stream, const float *logits, float *G, const int32_t *Y,
                                                 size_t n_rows, size_t m, float grad_scale, float *sums_device) {
    if (!n_rows) return;
    MGGCN_REQUIRE(m > 0 && m <= 64u * kXentMaxPerLane, "fused loss supports 1 <= m <= 1024 classes");
    MGGCN_REQUIRE(logits != nullptr && G != nullptr && Y != nullptr && sums_device != nullptr, "fused loss: null operand");
    hipStream_t st = as_stream(stream);
    const dim3 block(256);
    float *partials = abssum_scratch(st);        // per (device, stream); stream order keeps its users apart
    if (m <= 64) {                               // one row per 16-lane group
        const dim3 grid(stream_grid(n_rows * 16));
#define MGGCN_XENT16(KE)                                                                                   \
    hipLaunchKernelGGL((softmax_xent_rows16_kernel<KE, 4>), grid, block, 0, st, logits, G, Y, n_rows, (uint32_t)m, \
                       grad_scale, partials)
        if (m <= 16) MGGCN_XENT16(1);
        else if (m <= 32) MGGCN_XENT16(2);
        else if (m <= 48) MGGCN_XENT16(3);
        else MGGCN_XENT16(4);
#undef MGGCN_XENT16
        MGGCN_CHECK_LAUNCH();
        hipLaunchKernelGGL(xent_final_kernel, dim3(1), dim3(256), 0, st, partials, grid.x, sums_device);
        MGGCN_CHECK_LAUNCH();
        return;
    }
    // (tried on the wave-per-row form: two workgroups per CU to thin out the two contended scalar atomics at the end of
    //  every workgroup -- 105 -> 145 us: the pass wants the occupancy)
    const dim3 grid(stream_grid(n_rows * 64));
#define MGGCN_XENT(K, R)                                                                              \
    hipLaunchKernelGGL((softmax_xent_fused_kernel<K, R>), grid, block, 0, st, logits, G, Y, n_rows, m, grad_scale, \
                       partials)
    if (m <= 128) MGGCN_XENT(2, 4);
    else if (m <= 256) MGGCN_XENT(4, 2);
    else if (m <= 512) MGGCN_XENT(8, 1);
    else MGGCN_XENT(16, 1);
#undef MGGCN_XENT
    MGGCN_CHECK_LAUNCH();
    hipLaunchKernelGGL(xent_final_kernel, dim3(1), dim3(256), 0, st, partials, grid.x, sums_device);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_softmax_xent_fused_f32(mggcn_stream_t stream, float *H, const int32_t *Y,
                                            size_t n_rows, size_t m, float grad_scale, float *sums_device) {
    mggcn_softmax_xent_fused_from_f32(stream, H, H, Y, n_rows, m, grad_scale, sums_device);
}

MGGCN_API uint32_t mggcn_adam_multi_blocks(uint64_t size) { return (uint32_t)((size + 1023) / 1024); }

MGGCN_API void mggcn_adam_multi_f32(mggcn_stream_t stream, const mggcn_adam_tensor *table_device, uint32_t n_tensors,
                                    uint32_t total_blocks, float lr, float beta1, float beta2, float c1, float c2,
                                    float eps) {
    if (!n_tensors || !total_blocks) return;
    MGGCN_REQUIRE(table_device != nullptr, "mggcn_adam_multi_f32 needs the tensor table");
    hipLaunchKernelGGL(adam_multi_kernel, dim3(total_blocks), dim3(256), 0, as_stream(stream), table_device, n_tensors,
                       lr / c1, beta1, beta2, c2, eps);
    MGGCN_CHECK_LAUNCH();
}

MGGCN_API void mggcn_adam_fused_f32(mggcn_stream_t stream, float *param, float *grad, float *m, float *v,
                                    float lr, float beta1, float beta2, float weight_decay, float c1,
                                    float c2, float eps, size_t size) {
    if (!size) return;
    hipLaunchKernelGGL(adam_fused_kernel, dim3(stream_grid(size)), dim3(256), 0, as_stream(stream), param,
                       grad, m, v, lr / c1, beta1, beta2, weight_decay, c2, eps, size);
    MGGCN_CHECK_LAUNCH();
}
