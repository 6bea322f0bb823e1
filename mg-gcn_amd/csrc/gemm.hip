// gemm.hip -- dense fp32 GEMM on the gfx950 matrix cores (the only MFMA path).
//
// Stands in for matmul(context, dn A, dn B, dn C, alpha, beta, A_T, B_T) =
// cublasSgemm with swapped operands (reference src/cuda_utils.hpp:149-172):
//     C = alpha * op(A) * op(B) + beta * C,   everything row-major.
//
// The shapes of a GCN epoch are tall and skinny:
//   forward   H[n x in] . W[in x out]              n = 233 k, in <= 608, out <= 128
//   backward  G_W = X^T[in x n] . G[n x out]       K = n  -> split over K
//             G_out = G[n x out] . W^T[out x in]
//             G_b = 1^T[1 x n] . G[n x out]
// so the kernel is a 128 x {128|64} output tile per workgroup with the whole of N
// in one or two tiles, K streamed in 32-deep steps through LDS, and a split-K
// grid dimension for the tall reductions.  Arithmetic is
// v_mfma_f32_32x32x2_f32: f32 in / f32 accumulate, bitwise a k-ordered fmaf chain
// (no TF32-style truncation exists on gfx950), i.e. at least as accurate as the
// reference's cublasSgemm.
//
// Tile anatomy (256 threads = 4 wave64):
//   BN = 128: waves 2(M) x 2(N), each 64 x 64 = 2x2 MFMA blocks  (64 acc VGPRs)
//   BN =  64: waves 4(M) x 1(N), each 32 x 64 = 1x2 MFMA blocks  (32 acc VGPRs)
//   LDS: As[32][BM+pad] + Bs[32][BN+pad] fp32, k-major so the MFMA operand read
//        (lane l -> row l&31, k l>>5) is a conflict-free ds_read_b32.
//   Global->LDS goes through registers, prefetched one K-step ahead.
//   Operands whose K index is the contiguous one (A not transposed, B transposed)
//   are transposed on the LDS write; the others are copied with 16-byte stores.
// Split-K partial tiles land in a caller-provided slab and are summed in split
// order by a second kernel (reproducible; no float atomics).
#include <cstdlib>

#include "common.h"

namespace {

typedef float f32x16 __attribute__((ext_vector_type(16)));

constexpr int BM = 128;
constexpr int BK = 32;

// what happens to alpha * (A B) on its way to C besides beta * C
struct Epilogue {
    const float *bias = nullptr;     // + 1 bias^T            (mggcn_gemm_bias_f32)
    const float *mask = nullptr;     // .* (Z > 0 ? 1 : slope) (mggcn_gemm_lrelu_bwd_f32)
    size_t ldz = 0;
    float slope = 0.f;
    float *colsum = nullptr;         // also 1^T op(B) -> colsum[N]   (mggcn_gemm_tn_colsum_f32; B stored [K][N])
    int prio = 0;                    // experiment (MGGCN_GEMM_PRIO): 1 = memory phases at s_setprio 1, 2 = MFMA phase at 1
};

__device__ __forceinline__ float epilogue_value(const Epilogue &e, float av, float beta, const float *cp, size_t row, size_t col) {
    // bias: the row vector the reference broadcasts into C before its beta = 1 sgemm (src/gcn.hpp:116-123);
    // same single rounding as fmaf(1, bias, alpha*v)
    if (e.bias) return fmaf(1.f, e.bias[col], av);
    // mask: leaky_relu_backward of the consumer (src/cuda_utils.cu:33-38) on the value a beta = 0 GEMM would store
    if (e.mask) return e.mask[row * e.ldz + col] > 0.f ? av : e.slope * av;
    return beta == 0.f ? av : fmaf(beta, *cp, av);
}

// Loads 4 consecutive elements along the contiguous dimension, zero-filled
// outside [0, limit).  vec: the whole operand is 16-byte aligned with ld % 4 == 0.
__device__ __forceinline__ float4 load4_guard(const float *__restrict__ p, long long first,
                                              long long limit, bool vec) {
    float4 r = make_float4(0.f, 0.f, 0.f, 0.f);
    if (first + 3 < limit && vec) {
        r = *reinterpret_cast<const float4 *>(p + first);
    } else {
        if (first + 0 < limit) r.x = p[first + 0];
        if (first + 1 < limit) r.y = p[first + 1];
        if (first + 2 < limit) r.z = p[first + 2];
        if (first + 3 < limit) r.w = p[first + 3];
    }
    return r;
}

// One operand's tile staging.  R = rows of the tile in the M (or N) direction.
// KCONTIG: the stored matrix is [R-dim][K] (K contiguous) -> transpose into Xs[k][r].
// else   : the stored matrix is [K][R-dim] (R contiguous) -> straight copy.
template <int R, bool KCONTIG, int NT>
struct Stager {
    static constexpr int PAD = KCONTIG ? 1 : 4;
    static constexpr int LD = R + PAD;
    static constexpr int SEGS = R * BK / 4 / NT;    // float4 segments per thread
    static_assert(SEGS >= 1 && R * BK / 4 % NT == 0, "tile does not divide over the threads");
    float4 reg[SEGS];

    __device__ __forceinline__ void load(const float *__restrict__ X, size_t ld, long long r0,
                                         long long r_limit, long long k0, long long k_limit, bool vec,
                                         int tid) {
#pragma unroll
        for (int s = 0; s < SEGS; s++) {
            const int f = tid + NT * s;
            if (KCONTIG) {
                const int r = f / (BK / 4), kq = f % (BK / 4);
                const long long row = r0 + r;
                reg[s] = row < r_limit ? load4_guard(X + (size_t)row * ld, k0 + kq * 4, k_limit, vec)
                                       : make_float4(0.f, 0.f, 0.f, 0.f);
            } else {
                const int k = f / (R / 4), q = f % (R / 4);
                const long long kk = k0 + k;
                reg[s] = kk < k_limit ? load4_guard(X + (size_t)kk * ld, r0 + q * 4, r_limit, vec)
                                      : make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    }

    __device__ __forceinline__ void store(float *__restrict__ Xs, int tid) const {
#pragma unroll
        for (int s = 0; s < SEGS; s++) {
            const int f = tid + NT * s;
            if (KCONTIG) {
                const int r = f / (BK / 4), kq = f % (BK / 4);
                Xs[(kq * 4 + 0) * LD + r] = reg[s].x;
                Xs[(kq * 4 + 1) * LD + r] = reg[s].y;
                Xs[(kq * 4 + 2) * LD + r] = reg[s].z;
                Xs[(kq * 4 + 3) * LD + r] = reg[s].w;
            } else {
                const int k = f / (R / 4), q = f % (R / 4);
                *reinterpret_cast<float4 *>(Xs + k * LD + q * 4) = reg[s];
            }
        }
    }
};

// NT = 256: 4 waves, each 64x64 (BN = 128) or 32x64 (BN = 64) of the tile -- 64 / 32 accumulator registers, ~170 VGPRs,
//           two workgroups = 2 waves per SIMD.
// NT = 512: 8 waves, each 32x64 (BN = 128) or 32x32 (BN = 64) -- half the accumulators per wave, ~90 VGPRs, two
//           workgroups = 4 waves per SIMD: the matrix pipe of a SIMD keeps running while some of its waves sit at
//           the workgroup barrier or wait for their LDS stores (r01 counters on [n x 608].[608 x 128]: MFMA pipe 55 %
//           busy, waves parked on s_waitcnt / s_barrier 39 % of their cycles with two waves per SIMD).
// The K loop is double-buffered in LDS: tile k+1 is written to the other buffer after the MFMAs of tile k, ONE
// barrier per K-step (the single-buffer loop needed two and serialised store -> compute).
template <bool A_KCONTIG, bool B_KCONTIG, int BN, int NT>
__global__ __launch_bounds__(NT) void gemm_mfma_kernel(
    uint32_t M, uint32_t N, uint32_t K, float alpha, const float *__restrict__ A, size_t lda,
    const float *__restrict__ B, size_t ldb, float beta, float *__restrict__ C, size_t ldc,
    float *__restrict__ slab, uint32_t k_chunk, bool a_vec, bool b_vec, const Epilogue epi) {
    constexpr int NW = NT / 64;
    constexpr int WN = (BN == 128 || NW == 8) ? 2 : 1;   // waves along N
    constexpr int WM = NW / WN;                     // waves along M
    constexpr int MI = BM / WM / 32;                // MFMA blocks per wave along M
    constexpr int NI = BN / WN / 32;                // along N
    static_assert(MI >= 1 && NI >= 1, "wave tile smaller than one MFMA block");
    using StA = Stager<BM, A_KCONTIG, NT>;
    using StB = Stager<BN, B_KCONTIG, NT>;
    __shared__ __attribute__((aligned(16))) float As[2][BK * StA::LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * StB::LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    const int wm = wid / WN, wn = wid % WN;
    const long long m0 = (long long)blockIdx.x * BM, n0 = (long long)blockIdx.y * BN;
    const long long k_begin = (long long)blockIdx.z * k_chunk;
    const long long k_end = min((long long)K, k_begin + (long long)k_chunk);

    f32x16 acc[MI][NI];
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++)
#pragma unroll
            for (int r = 0; r < 16; r++) acc[i][j][r] = 0.f;

    StA sa;
    StB sb;
    // column sums of B (G_b = 1^T G riding on G_W = X^T G): the workgroups of the FIRST M-tile add up the B tiles
    // they stage anyway -- thread tid always holds columns (tid % (BN/4))*4 .. +3 of its k rows
    const bool do_colsum = !B_KCONTIG && epi.colsum != nullptr && blockIdx.x == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_colsum = [&]() {
        if constexpr (!B_KCONTIG) {
            if (do_colsum) {
#pragma unroll
                for (int q = 0; q < StB::SEGS; q++) {
                    csum.x += sb.reg[q].x; csum.y += sb.reg[q].y; csum.z += sb.reg[q].z; csum.w += sb.reg[q].w;
                }
            }
        }
    };
    if (k_begin < k_end) {
        sa.load(A, lda, m0, M, k_begin, k_end, a_vec, tid);
        sb.load(B, ldb, n0, N, k_begin, k_end, b_vec, tid);
        add_colsum();
        sa.store(As[0], tid);
        sb.store(Bs[0], tid);
    }
    __syncthreads();
    const int l31 = lane & 31, lhi = lane >> 5;
    int cur = 0;
    for (long long k0 = k_begin; k0 < k_end; k0 += BK, cur ^= 1) {
        const bool more = k0 + BK < k_end;
        if (epi.prio == 1) __builtin_amdgcn_s_setprio(1);
        if (more) {                      // global loads of the next K-step fly under the MFMAs
            sa.load(A, lda, m0, M, k0 + BK, k_end, a_vec, tid);
            sb.load(B, ldb, n0, N, k0 + BK, k_end, b_vec, tid);
        }
        if (epi.prio == 1) __builtin_amdgcn_s_setprio(0);
        if (epi.prio == 2) __builtin_amdgcn_s_setprio(1);
        const float *as = As[cur], *bs = Bs[cur];
#pragma unroll
        for (int kk = 0; kk < BK / 2; kk++) {
            float a[MI], b[NI];
#pragma unroll
            for (int i = 0; i < MI; i++)
                a[i] = as[(kk * 2 + lhi) * StA::LD + wm * (BM / WM) + i * 32 + l31];
#pragma unroll
            for (int j = 0; j < NI; j++)
                b[j] = bs[(kk * 2 + lhi) * StB::LD + wn * (BN / WN) + j * 32 + l31];
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[i], b[j], acc[i][j], 0, 0, 0);
        }
        if (epi.prio == 2) __builtin_amdgcn_s_setprio(0);
        if (epi.prio == 1) __builtin_amdgcn_s_setprio(1);
        if (more) {                      // the other buffer: nobody reads it until the barrier below
            add_colsum();
            sa.store(As[cur ^ 1], tid);
            sb.store(Bs[cur ^ 1], tid);
        }
        if (epi.prio == 1) __builtin_amdgcn_s_setprio(0);
        __syncthreads();
    }
    if constexpr (!B_KCONTIG) {
        if (do_colsum) {                 // block-uniform.  Fold the NT / (BN/4) threads of a column group in order.
            float4 *red = reinterpret_cast<float4 *>(As[0]);          // all tiles are consumed: reuse the LDS
            red[tid] = csum;
            __syncthreads();
            if (tid < BN / 4) {
                float4 t = red[tid];
                for (int g = 1; g < NT / (BN / 4); g++) {
                    const float4 o = red[tid + g * (BN / 4)];
                    t.x += o.x; t.y += o.y; t.z += o.z; t.w += o.w;
                }
                const float tv[4] = {t.x, t.y, t.z, t.w};
                // split-K: row M of this slice's slab (reduced with the rest); otherwise straight to the result
                float *dst = gridDim.z > 1 ? slab + ((size_t)blockIdx.z * (M + 1) + M) * N : epi.colsum;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    const long long col = n0 + tid * 4 + c;
                    if (col < N) dst[col] = gridDim.z > 1 ? tv[c] : alpha * tv[c];
                }
            }
        }
    }

    // epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5)
    const bool to_slab = gridDim.z > 1;
#pragma unroll
    for (int i = 0; i < MI; i++)
#pragma unroll
        for (int j = 0; j < NI; j++) {
            const long long col = n0 + wn * (BN / WN) + j * 32 + l31;
#pragma unroll
            for (int r = 0; r < 16; r++) {
                const long long row = m0 + wm * (BM / WM) + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (row < M && col < N) {
                    const float v = acc[i][j][r];
                    if (to_slab) {
                        slab[((size_t)blockIdx.z * (M + (epi.colsum ? 1 : 0)) + row) * N + col] = v;
                    } else {
                        float *cp = C + (size_t)row * ldc + col;
                        *cp = epilogue_value(epi, alpha * v, beta, cp, (size_t)row, (size_t)col);
                    }
                }
            }
        }
}

// C = alpha * sum_z slab[z] + beta * C.  256 threads = 32 outputs x 8 split-groups: group g adds
// slabs g, g+8, g+16, ... (independent loads, unrolled), the 8 group sums are folded through
// LDS in group order -> a fixed summation tree, bitwise reproducible.
__global__ __launch_bounds__(256) void gemm_splitk_reduce_kernel(const float *__restrict__ slab,
                                                                 uint32_t splits, uint32_t M, uint32_t N,
                                                                 float alpha, float beta,
                                                                 float *__restrict__ C, size_t ldc,
                                                                 const Epilogue epi) {
    __shared__ float part[8][33];
    // with a column-sum row the slabs hold M + 1 rows; row M goes to epi.colsum instead of C
    const uint32_t total = (M + (epi.colsum ? 1u : 0u)) * N;   // < 2^32: checked by the launcher
    const uint32_t lane = threadIdx.x & 31, grp = threadIdx.x >> 5;
    for (uint32_t base = blockIdx.x * 32; base < total; base += gridDim.x * 32) {
        const uint32_t i = base + lane;
        float s = 0.f;
        if (i < total) {
#pragma unroll 4
            for (uint32_t z = grp; z < splits; z += 8) s += slab[(size_t)z * total + i];
        }
        part[grp][lane] = s;
        __syncthreads();
        if (grp == 0 && i < total) {
            float t = part[0][lane];
#pragma unroll
            for (int g = 1; g < 8; g++) t += part[g][lane];
            if (i / N >= M) {
                epi.colsum[i % N] = alpha * t;
            } else {
                float *cp = C + (size_t)(i / N) * ldc + (i % N);
                *cp = epilogue_value(epi, alpha * t, beta, cp, (size_t)(i / N), (size_t)(i % N));
            }
        }
        __syncthreads();
    }
}

// Very thin op(A) (M <= 4 rows, not transposed): G_b = 1^T G (reference src/gcn.hpp:131) and
// friends.  An MFMA tile would waste 124 of its 128 rows; this is a plain HBM stream of B:
// thread (tx, ty) owns columns {n0+tx, n0+tx+64} and the k's congruent to ty (mod 4) of the
// block's K-chunk; the four ty partials are folded through LDS in order and the chunk's
// [M x N] partial goes to the split-K slab.
template <int MROWS>
__global__ __launch_bounds__(256) void gemm_thin_kernel(uint32_t N, uint32_t K, const float *__restrict__ A,
                                                        size_t lda, const float *__restrict__ B, size_t ldb,
                                                        float *__restrict__ slab, uint32_t k_chunk, uint32_t M) {
    __shared__ float red[4][MROWS][128];
    const uint32_t tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const uint32_t n0 = blockIdx.x * 128;
    const uint32_t c0 = n0 + tx, c1 = n0 + tx + 64;
    const uint32_t k_begin = blockIdx.y * k_chunk, k_end = min(K, k_begin + k_chunk);
    float acc0[MROWS], acc1[MROWS];
#pragma unroll
    for (int m = 0; m < MROWS; m++) { acc0[m] = 0.f; acc1[m] = 0.f; }
    for (uint32_t k = k_begin + ty; k < k_end; k += 4) {
        const float b0 = c0 < N ? B[(size_t)k * ldb + c0] : 0.f;
        const float b1 = c1 < N ? B[(size_t)k * ldb + c1] : 0.f;
#pragma unroll
        for (int m = 0; m < MROWS; m++) {
            const float a = (uint32_t)m < M ? A[(size_t)m * lda + k] : 0.f;
            acc0[m] = fmaf(a, b0, acc0[m]);
            acc1[m] = fmaf(a, b1, acc1[m]);
        }
    }
#pragma unroll
    for (int m = 0; m < MROWS; m++) { red[ty][m][tx] = acc0[m]; red[ty][m][tx + 64] = acc1[m]; }
    __syncthreads();
    if (ty == 0) {
#pragma unroll
        for (int m = 0; m < MROWS; m++) {
            if ((uint32_t)m >= M) break;
            const float s0 = ((red[0][m][tx] + red[1][m][tx]) + red[2][m][tx]) + red[3][m][tx];
            const float s1 = ((red[0][m][tx + 64] + red[1][m][tx + 64]) + red[2][m][tx + 64]) + red[3][m][tx + 64];
            float *out = slab + ((size_t)blockIdx.y * M + m) * N;
            if (c0 < N) out[c0] = s0;
            if (c1 < N) out[c1] = s1;
        }
    }
}

// C = beta * C (K == 0 degenerate case)
__global__ __launch_bounds__(256) void gemm_scale_c_kernel(uint32_t M, uint32_t N, float beta,
                                                           float *__restrict__ C, size_t ldc) {
    const size_t total = (size_t)M * N;
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        float *cp = C + (i / N) * ldc + (i % N);
        *cp = beta == 0.f ? 0.f : beta * *cp;
    }
}

struct Split {
    uint32_t splits;
    uint32_t k_chunk;   // multiple of BK
};

constexpr uint32_t kThinRows = 4;

// threads per workgroup of the MFMA kernel (see gemm_mfma_kernel); MGGCN_GEMM_THREADS = 256 | 512 for experiments
inline int gemm_threads() {
    static const int nt = [] {
        const char *e = std::getenv("MGGCN_GEMM_THREADS");
        const int v = e ? std::atoi(e) : 512;
        return v == 256 ? 256 : 512;
    }();
    return nt;
}

inline bool use_thin(int trans_a, int trans_b, uint32_t M, uint32_t K) {
    return !trans_a && !trans_b && M <= kThinRows && K >= 4096;
}

// thin path: ~1024 K-chunks of a multiple of 64
Split choose_split_thin(uint32_t K) {
    uint32_t chunk = (K + 1023) / 1024;
    chunk = (chunk + 63) / 64 * 64;
    return {(K + chunk - 1) / chunk, chunk};
}

// Tall reductions: fill the resident slots (2 workgroups per CU: 168-172 VGPRs) with K-slices, and
// never exceed them -- 515 blocks on 512 slots run a second round for three blocks
// (G_W = X^T G, M = 608: 103 slices 687 us, 102 slices 501 us; profiles/experiments/gemm_splits.py).
Split choose_split(uint32_t M, uint32_t N, uint32_t K) {
    const uint32_t bn = N > 64 ? 128 : 64;
    const uint64_t tiles = (uint64_t)((M + BM - 1) / BM) * ((N + bn - 1) / bn);
    const uint32_t k_steps = (K + BK - 1) / BK;
    uint32_t splits = 1;
    if (tiles < (uint64_t)kNumCU && k_steps >= 16) {
        const uint32_t want = (uint32_t)std::max<uint64_t>(1, (2ull * kNumCU) / tiles);
        splits = std::min<uint32_t>(want, k_steps / 8);   // at least 8 K-steps per slice
        splits = std::max<uint32_t>(splits, 1u);
    }
    if (const char *e = std::getenv("MGGCN_GEMM_SPLITS"))          // tuning knob (profiles/experiments/gemm_splits.py)
        splits = std::max<uint32_t>(1u, std::min<uint32_t>((uint32_t)std::strtoul(e, nullptr, 10), k_steps));
    const uint32_t steps_per = (k_steps + splits - 1) / splits;
    splits = (k_steps + steps_per - 1) / steps_per;
    return {splits ? splits : 1u, steps_per * BK};
}

}  // namespace

MGGCN_API size_t mggcn_gemm_tn_colsum_workspace_bytes(uint32_t M, uint32_t N, uint32_t K) {
    if (!M || !N || !K) return 0;
    const Split s = choose_split(M, N, K);
    return s.splits > 1 ? (size_t)s.splits * (M + 1) * N * sizeof(float) : 0;
}

MGGCN_API size_t mggcn_gemm_workspace_bytes(int trans_a, int trans_b, uint32_t M, uint32_t N, uint32_t K) {
    (void)trans_a; (void)trans_b;
    if (!M || !N || !K) return 0;
    if (use_thin(trans_a, trans_b, M, K)) return (size_t)choose_split_thin(K).splits * M * N * sizeof(float);
    const Split s = choose_split(M, N, K);
    return s.splits > 1 ? (size_t)s.splits * M * N * sizeof(float) : 0;
}

namespace {
void gemm_dispatch(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N, uint32_t K, float alpha,
                   const float *A, size_t lda, const float *B, size_t ldb, float beta, float *C, size_t ldc,
                   void *workspace, size_t workspace_bytes, const Epilogue &epi);
}

MGGCN_API void mggcn_gemm_f32(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N,
                              uint32_t K, float alpha, const float *A, size_t lda, const float *B,
                              size_t ldb, float beta, float *C, size_t ldc, void *workspace,
                              size_t workspace_bytes) {
    gemm_dispatch(stream, trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, beta, C, ldc, workspace, workspace_bytes,
                  Epilogue{});
}

MGGCN_API void mggcn_gemm_bias_f32(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N,
                                   uint32_t K, float alpha, const float *A, size_t lda, const float *B,
                                   size_t ldb, const float *bias, float *C, size_t ldc, void *workspace,
                                   size_t workspace_bytes) {
    MGGCN_REQUIRE(bias != nullptr, "mggcn_gemm_bias_f32 needs a bias row");
    if (M && N && !K) {          // degenerate: C = 1 bias^T
        MGGCN_REQUIRE(ldc == N, "K == 0 with a bias needs a dense C");
        mggcn_broadcast_rows_f32(stream, bias, C, (size_t)M * N, N, 1);
        return;
    }
    Epilogue e;
    e.bias = bias;
    gemm_dispatch(stream, trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, workspace, workspace_bytes, e);
}

MGGCN_API void mggcn_gemm_tn_colsum_f32(mggcn_stream_t stream, uint32_t M, uint32_t N, uint32_t K, float alpha,
                                        const float *A, size_t lda, const float *B, size_t ldb, float *C, size_t ldc,
                                        float *colsum, void *workspace, size_t workspace_bytes) {
    MGGCN_REQUIRE(colsum != nullptr, "mggcn_gemm_tn_colsum_f32 needs the column-sum output");
    if (!N) return;
    if (!K) {
        MGGCN_CHECK_HIP(hipMemsetAsync(colsum, 0, (size_t)N * sizeof(float), as_stream(stream)));
        if (M) gemm_dispatch(stream, 1, 0, M, N, 0, alpha, A, lda, B, ldb, 0.f, C, ldc, workspace, workspace_bytes, Epilogue{});
        return;
    }
    MGGCN_REQUIRE(M > 0, "mggcn_gemm_tn_colsum_f32: M == 0");
    Epilogue e;
    e.colsum = colsum;
    gemm_dispatch(stream, 1, 0, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, workspace, workspace_bytes, e);
}

MGGCN_API void mggcn_gemm_lrelu_bwd_f32(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N,
                                        uint32_t K, float alpha, const float *A, size_t lda, const float *B,
                                        size_t ldb, const float *Z, size_t ldz, float slope, float *C, size_t ldc,
                                        void *workspace, size_t workspace_bytes) {
    MGGCN_REQUIRE(Z != nullptr && ldz >= N, "mggcn_gemm_lrelu_bwd_f32 needs the activation matrix Z (ldz >= N)");
    MGGCN_REQUIRE(K > 0 || !M || !N, "mggcn_gemm_lrelu_bwd_f32: K == 0");
    Epilogue e;
    e.mask = Z;
    e.ldz = ldz;
    e.slope = slope;
    gemm_dispatch(stream, trans_a, trans_b, M, N, K, alpha, A, lda, B, ldb, 0.f, C, ldc, workspace, workspace_bytes, e);
}

namespace {
void gemm_dispatch(mggcn_stream_t stream, int trans_a, int trans_b, uint32_t M, uint32_t N, uint32_t K, float alpha,
                   const float *A, size_t lda, const float *B, size_t ldb, float beta, float *C, size_t ldc,
                   void *workspace, size_t workspace_bytes, const Epilogue &epi_in) {
    if (!M || !N) return;
    Epilogue epi = epi_in;
    static const int prio_env = [] { const char *e = std::getenv("MGGCN_GEMM_PRIO"); return e ? std::atoi(e) : 0; }();
    epi.prio = prio_env;
    hipStream_t st = as_stream(stream);
    MGGCN_REQUIRE(C != nullptr && ldc >= N, "bad C / ldc");
    if (!K) {
        hipLaunchKernelGGL(gemm_scale_c_kernel, dim3(stream_grid((size_t)M * N)), dim3(256), 0, st, M, N, beta,
                           C, ldc);
        MGGCN_CHECK_LAUNCH();
        return;
    }
    MGGCN_REQUIRE(A != nullptr && B != nullptr, "null operand");
    MGGCN_REQUIRE(lda >= (trans_a ? M : K), "lda smaller than the stored row of A");
    MGGCN_REQUIRE(ldb >= (trans_b ? K : N), "ldb smaller than the stored row of B");
    MGGCN_REQUIRE((uint64_t)M * N < (1ull << 31), "output too large");
    if (use_thin(trans_a, trans_b, M, K)) {
        const Split sp = choose_split_thin(K);
        MGGCN_REQUIRE(workspace != nullptr && workspace_bytes >= (size_t)sp.splits * M * N * sizeof(float),
                      "thin GEMM needs the workspace reported by mggcn_gemm_workspace_bytes");
        float *slab = static_cast<float *>(workspace);
        hipLaunchKernelGGL((gemm_thin_kernel<kThinRows>), dim3((N + 127) / 128, sp.splits), dim3(256), 0, st, N, K, A,
                           lda, B, ldb, slab, sp.k_chunk, M);
        MGGCN_CHECK_LAUNCH();
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(std::min<unsigned>((M * N + 31) / 32, 2048u)), dim3(256), 0,
                           st, slab, sp.splits, M, N, alpha, beta, C, ldc, epi);
        MGGCN_CHECK_LAUNCH();
        return;
    }
    const Split sp = choose_split(M, N, K);
    float *slab = nullptr;
    const uint32_t slab_rows = M + (epi.colsum ? 1u : 0u);
    if (sp.splits > 1) {
        MGGCN_REQUIRE(workspace != nullptr && workspace_bytes >= (size_t)sp.splits * slab_rows * N * sizeof(float),
                      "split-K GEMM needs the workspace reported by mggcn_gemm_[tn_colsum_]workspace_bytes");
        slab = static_cast<float *>(workspace);
    }
    const bool a_vec = aligned16(A) && lda % 4 == 0;
    const bool b_vec = aligned16(B) && ldb % 4 == 0;
    const bool a_kc = !trans_a, b_kc = trans_b != 0;
    const int bn = N > 64 ? 128 : 64;
    const dim3 grid((M + BM - 1) / BM, (N + bn - 1) / bn, sp.splits);

    const int nt = gemm_threads();
    const dim3 blk(nt);
#define MGGCN_GEMM_LAUNCH(AK, BKC, BNV, NTV)                                                               \
    hipLaunchKernelGGL((gemm_mfma_kernel<AK, BKC, BNV, NTV>), grid, blk, 0, st, M, N, K, alpha, A, lda, B, ldb, \
                       beta, C, ldc, slab, sp.k_chunk, a_vec, b_vec, epi)
#define MGGCN_GEMM_LAUNCH_NT(AK, BKC, BNV) \
    do { if (nt == 512) MGGCN_GEMM_LAUNCH(AK, BKC, BNV, 512); else MGGCN_GEMM_LAUNCH(AK, BKC, BNV, 256); } while (0)
    if (bn == 128) {
        if (a_kc && b_kc) MGGCN_GEMM_LAUNCH_NT(true, true, 128);
        else if (a_kc) MGGCN_GEMM_LAUNCH_NT(true, false, 128);
        else if (b_kc) MGGCN_GEMM_LAUNCH_NT(false, true, 128);
        else MGGCN_GEMM_LAUNCH_NT(false, false, 128);
    } else {
        if (a_kc && b_kc) MGGCN_GEMM_LAUNCH_NT(true, true, 64);
        else if (a_kc) MGGCN_GEMM_LAUNCH_NT(true, false, 64);
        else if (b_kc) MGGCN_GEMM_LAUNCH_NT(false, true, 64);
        else MGGCN_GEMM_LAUNCH_NT(false, false, 64);
    }
#undef MGGCN_GEMM_LAUNCH_NT
#undef MGGCN_GEMM_LAUNCH
    MGGCN_CHECK_LAUNCH();
    if (sp.splits > 1) {
        hipLaunchKernelGGL(gemm_splitk_reduce_kernel, dim3(std::min<unsigned>((slab_rows * N + 31) / 32, 2048u)), dim3(256), 0,
                           st, slab, sp.splits, M, N, alpha, beta, C, ldc, epi);
        MGGCN_CHECK_LAUNCH();
    }
}
}  // namespace
