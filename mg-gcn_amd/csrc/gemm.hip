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
// Tile anatomy (512 threads = 8 wave64, two workgroups per CU = 4 waves per SIMD, <= 128 VGPRs):
//   BN = 128: waves 4(M) x 2(N), each 32 x 64 = 1x2 MFMA blocks  (32 acc VGPRs)
//   BN =  64: waves 4(M) x 2(N), each 32 x 32 = 1x1 MFMA block   (16 acc VGPRs)
//   LDS: two buffers of As[32][BM+pad] + Bs[32][BN+pad] fp32, k-major so the MFMA operand read
//        (lane l -> row l&31, k l>>5) is a conflict-free ds_read_b32; ONE barrier per K-step.
//   Global->LDS goes through two register sets: tile t+2 is requested while tile t is multiplied and
//   tile t+1 (requested a step earlier) is written to the other LDS buffer between the two halves of
//   the step's MFMAs.  Loads are branch-free (clamped addresses; see Stager).
//   Operands whose K index is the contiguous one (A not transposed, B transposed)
//   are transposed on the LDS write; the others are copied with 16-byte stores.
// Split-K partial tiles land in a caller-provided slab and are summed in split
// order by a second kernel (reproducible; no float atomics).
// Measured (profiles/r02_gemm_summary.md): 101 TF on both K = 608 products of the Reddit epoch with
// the clock warm, MFMA pipe 70 % busy at the ~2.0 GHz the chip holds under this load.
#include <cstdlib>
#include <type_traits>

#include "common.h"

// Measurement hooks of profiles/experiments/gemm_timeline.hip (which defines them and includes this file); compiled
// out of the library.
#ifndef MGGCN_GEMM_STAMP
#define MGGCN_GEMM_STAMP_DECL
#define MGGCN_GEMM_STAMP(id)
#endif
#ifndef MGGCN_GEMM_QUARTERS        // 1: memory work dealt out between four quarters of a K-step's MFMAs; 0: two halves
#define MGGCN_GEMM_QUARTERS 0
#endif
#ifndef MGGCN_GEMM_SETPRIO         // experiment: 1 = a wave's load / LDS-store sections run at s_setprio 1, its MFMA runs at 0; 2 = the reverse
#define MGGCN_GEMM_SETPRIO 0
#endif

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
};

__device__ __forceinline__ float epilogue_value(const Epilogue &e, float av, float beta, const float *cp, size_t row, size_t col) {
    // bias: the row vector the reference broadcasts into C before its beta = 1 sgemm (src/gcn.hpp:116-123);
    // same single rounding as fmaf(1, bias, alpha*v)
    if (e.bias) return fmaf(1.f, e.bias[col], av);
    // mask: leaky_relu_backward of the consumer (src/cuda_utils.cu:33-38) on the value a beta = 0 GEMM would store
    if (e.mask) return e.mask[row * e.ldz + col] > 0.f ? av : e.slope * av;
    return beta == 0.f ? av : fmaf(beta, *cp, av);
}

// One operand's tile staging.  R = rows of the tile in the M (or N) direction.
// KCONTIG: the stored matrix is [R-dim][K] (K contiguous) -> transpose into Xs[k][r].
// else   : the stored matrix is [K][R-dim] (R contiguous) -> straight copy.
// VEC    : the operand is 16-byte aligned with ld % 4 == 0 -> one 16-byte load per segment; else four 4-byte loads.
//
// The loads are BRANCH-FREE and never leave the matrix:
//  * rows / columns of the tile beyond the matrix edge (r >= r_limit) read a CLAMPED address and keep what they get:
//    such a row of the A tile (column of the B tile) only feeds output rows (columns) that the epilogue never stores,
//    so no mask is needed -- and with VEC a float4 that straddles r_limit still lies inside its row (ld % 4 == 0);
//  * k beyond k_end reads a clamped address too and is ZEROED afterwards (mask_k_edge) -- only in the one K-step that
//    contains the edge, a block-uniform branch around register-only code.
// The first version branched per segment between a vector path, four guarded scalar loads and zero fill, all writing the
// same registers: the compiler put s_waitcnt vmcnt(0) in front of loads, kept both paths' addresses live across the K
// loop and spilled at the 128-VGPR budget -- and a scratch reload waits for EVERY outstanding global load
// (profiles/r02_gemm_summary.md).
template <int R, bool KCONTIG, int NT, bool VEC>
struct Stager {
    static constexpr int PAD = KCONTIG ? 1 : 4;
    static constexpr int LD = R + PAD;
    static constexpr int SEGS = R * BK / 4 / NT;    // float4 segments per thread
    static_assert(SEGS >= 1 && R * BK / 4 % NT == 0, "tile does not divide over the threads");
    float4 reg[SEGS];

    // the loop-invariant half of a thread's addresses
    struct Source {
        const float *X;
        size_t ld;
        long long r_limit;
        long long r[SEGS];           // KCONTIG: the (clamped) matrix row of segment s; else: its first matrix column
        __device__ __forceinline__ Source(const float *X_, size_t ld_, long long r0, long long r_limit_, int tid)
            : X(X_), ld(ld_), r_limit(r_limit_) {
#pragma unroll
            for (int s = 0; s < SEGS; s++) {
                const int f = tid + NT * s;
                if (KCONTIG) {
                    const long long row = r0 + f / (BK / 4);
                    r[s] = row < r_limit ? row : r_limit - 1;
                } else {
                    const long long col = r0 + (f % (R / 4)) * 4;
                    r[s] = (VEC && col >= r_limit) ? 0 : col;
                }
            }
        }
    };

    __device__ __forceinline__ void load(const Source &src, long long k0, long long k_end, int tid) {
#pragma unroll
        for (int s = 0; s < SEGS; s++) {
            const int f = tid + NT * s;
            if (KCONTIG) {
                const long long k = k0 + (f % (BK / 4)) * 4;
                const float *row = src.X + (size_t)src.r[s] * src.ld;
                if (VEC) {
                    reg[s] = *reinterpret_cast<const float4 *>(row + (k < k_end ? k : 0));
                } else {
                    const long long last = k_end - 1;
                    reg[s].x = row[k + 0 < k_end ? k + 0 : last];
                    reg[s].y = row[k + 1 < k_end ? k + 1 : last];
                    reg[s].z = row[k + 2 < k_end ? k + 2 : last];
                    reg[s].w = row[k + 3 < k_end ? k + 3 : last];
                }
            } else {
                const long long k = k0 + f / (R / 4);
                const float *row = src.X + (size_t)(k < k_end ? k : k_end - 1) * src.ld;
                if (VEC) {
                    reg[s] = *reinterpret_cast<const float4 *>(row + src.r[s]);
                } else {
                    const long long last = src.r_limit - 1, c = src.r[s];
                    reg[s].x = row[c + 0 < src.r_limit ? c + 0 : last];
                    reg[s].y = row[c + 1 < src.r_limit ? c + 1 : last];
                    reg[s].z = row[c + 2 < src.r_limit ? c + 2 : last];
                    reg[s].w = row[c + 3 < src.r_limit ? c + 3 : last];
                }
            }
        }
    }

    // zero what load() fetched for k >= k_end (call when k0 + BK > k_end)
    __device__ __forceinline__ void mask_k_edge(long long k0, long long k_end, int tid) {
#pragma unroll
        for (int s = 0; s < SEGS; s++) {
            const int f = tid + NT * s;
            if (KCONTIG) {
                const long long k = k0 + (f % (BK / 4)) * 4;
                reg[s].x = k + 0 < k_end ? reg[s].x : 0.f;
                reg[s].y = k + 1 < k_end ? reg[s].y : 0.f;
                reg[s].z = k + 2 < k_end ? reg[s].z : 0.f;
                reg[s].w = k + 3 < k_end ? reg[s].w : 0.f;
            } else {
                if (k0 + f / (R / 4) >= k_end) reg[s] = make_float4(0.f, 0.f, 0.f, 0.f);
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

// NT = 512 threads: 8 waves, each 32x64 (BN = 128) or 32x32 (BN = 64), <= 128 VGPRs, two workgroups = 4 waves per
// SIMD: the matrix pipe of a SIMD keeps running while some of its waves sit at the workgroup barrier or in their
// load / LDS-store sections.  (The 256-thread layout of round 1 -- 4 waves of 64x64, ~170 VGPRs, 2 waves per SIMD --
// measured 1.83 ms for the epoch's shapes against 1.54: profiles/experiments/gemm_shapes_r02_threads{256,512}.log.)
// The K loop is double-buffered in LDS with ONE barrier per K-step.
template <bool A_KCONTIG, bool B_KCONTIG, int BN, int NT, bool VEC>
__global__ __launch_bounds__(NT, 4) void gemm_mfma_kernel(   // 4 waves per SIMD: <= 128 VGPRs
    uint32_t M, uint32_t N, uint32_t K, float alpha, const float *__restrict__ A, size_t lda,
    const float *__restrict__ B, size_t ldb, float beta, float *__restrict__ C, size_t ldc,
    float *__restrict__ slab, uint32_t k_chunk, const Epilogue epi) {
    // VEC: BOTH operands are 16-byte aligned with a leading dimension that is a multiple of 4 (every GEMM of the hidden
    // layers).  A template parameter, not a run-time flag: with the element-wise path compiled into the same kernel its
    // 4 x the addresses stayed live across the K loop and the kernel spilled at its 128-VGPR budget.
    constexpr int NW = NT / 64;
    constexpr int WN = (BN == 128 || NW == 8) ? 2 : 1;   // waves along N
    constexpr int WM = NW / WN;                     // waves along M
    constexpr int MI = BM / WM / 32;                // MFMA blocks per wave along M
    constexpr int NI = BN / WN / 32;                // along N
    static_assert(MI >= 1 && NI >= 1, "wave tile smaller than one MFMA block");
    using StA = Stager<BM, A_KCONTIG, NT, VEC>;
    using StB = Stager<BN, B_KCONTIG, NT, VEC>;
    __shared__ __attribute__((aligned(16))) float As[2][BK * StA::LD];
    __shared__ __attribute__((aligned(16))) float Bs[2][BK * StB::LD];

    const int tid = threadIdx.x, lane = tid & 63, wid = tid >> 6;
    MGGCN_GEMM_STAMP_DECL;
    MGGCN_GEMM_STAMP(0);
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

    // Two register sets: while the MFMAs of tile t run, the loads of tile t+2 are in flight and tile t+1 (loaded during
    // tile t-1, landed by now) is written to the other LDS buffer IN THE MIDDLE of the MFMA sequence -- the matrix pipe
    // keeps executing the queued MFMAs while the wave issues its ds_writes.  (One set, stores after the last MFMA: every
    // wave of the workgroup sat in the store / barrier phase at the same time and the pipe idled -- one workgroup alone
    // on a CU reached 61 TF, two 75: profiles/r02_gemm_summary.md.)
    StA sa[2];
    StB sb[2];
    // column sums of B (G_b = 1^T G riding on G_W = X^T G): the workgroups of the FIRST M-tile add up the B tiles
    // they stage anyway -- thread tid always holds columns (tid % (BN/4))*4 .. +3 of its k rows
    const bool do_colsum = !B_KCONTIG && epi.colsum != nullptr && blockIdx.x == 0;
    float4 csum = make_float4(0.f, 0.f, 0.f, 0.f);
    auto add_colsum = [&](const StB &t) {
        if constexpr (!B_KCONTIG) {
            if (do_colsum) {
#pragma unroll
                for (int q = 0; q < StB::SEGS; q++) {
                    csum.x += t.reg[q].x; csum.y += t.reg[q].y; csum.z += t.reg[q].z; csum.w += t.reg[q].w;
                }
            }
        }
    };
    const int l31 = lane & 31, lhi = lane >> 5;
    const typename StA::Source src_a(A, lda, m0, M, tid);
    const typename StB::Source src_b(B, ldb, n0, N, tid);
    auto load_a = [&](StA &t, long long k0) { t.load(src_a, k0, k_end, tid); };
    auto load_b = [&](StB &t, long long k0) { t.load(src_b, k0, k_end, tid); };
    // tile k0 is about to go to LDS: zero its k >= k_end part (block-uniform, at most once per workgroup)
    auto finish = [&](StA &ta, StB &tb, long long k0) {
        if (k0 + BK > k_end) {
            ta.mask_k_edge(k0, k_end, tid);
            tb.mask_k_edge(k0, k_end, tid);
        }
    };
    // operands of k-pair kk+1 are read while the MFMAs of kk issue (two register sets, spelled out: left to itself the
    // compiler re-used one set and waited out the LDS latency in front of every MFMA pair)
    auto read_ab = [&](const float *as, const float *bs, int kk, float (&a)[MI], float (&b)[NI]) {
#pragma unroll
        for (int i = 0; i < MI; i++) a[i] = as[(kk * 2 + lhi) * StA::LD + wm * (BM / WM) + i * 32 + l31];
#pragma unroll
        for (int j = 0; j < NI; j++) b[j] = bs[(kk * 2 + lhi) * StB::LD + wn * (BN / WN) + j * 32 + l31];
    };
    auto mfma_range = [&](const float *as, const float *bs, auto kk_lo, auto kk_hi) {
        constexpr int LO = decltype(kk_lo)::value, HI = decltype(kk_hi)::value;
        float a[2][MI], b[2][NI];
        read_ab(as, bs, LO, a[0], b[0]);
#pragma unroll
        for (int kk = LO; kk < HI; kk++) {
            const int c = (kk - LO) & 1;
            if (kk + 1 < HI) read_ab(as, bs, kk + 1, a[c ^ 1], b[c ^ 1]);
#pragma unroll
            for (int i = 0; i < MI; i++)
#pragma unroll
                for (int j = 0; j < NI; j++)
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(a[c][i], b[c][j], acc[i][j], 0, 0, 0);
        }
    };
    using q0 = std::integral_constant<int, 0>;
    using q1 [[maybe_unused]] = std::integral_constant<int, BK / 8>;     // quarters: MGGCN_GEMM_QUARTERS builds only
    using q2 = std::integral_constant<int, BK / 4>;
    using q3 [[maybe_unused]] = std::integral_constant<int, 3 * BK / 8>;
    using q4 = std::integral_constant<int, BK / 2>;
    // one K-step: tile (k0) is in LDS buffer CUR; set NXT holds tile k0 + BK (in flight or landed); set CUR is free.
    // The memory work of the step is dealt out between the four quarters of the MFMA sequence, so a wave never leaves
    // the matrix pipe without queued work for long: it issues a few loads / LDS stores while its last MFMAs execute.
    auto step = [&](long long k0, auto cur_c) {
        constexpr int CUR = decltype(cur_c)::value, NXT = CUR ^ 1;
        const bool more2 = k0 + 2 * BK < k_end, more1 = k0 + BK < k_end;
        MGGCN_GEMM_STAMP(1);
#if MGGCN_GEMM_QUARTERS
        if (more2) load_a(sa[CUR], k0 + 2 * BK);       // tile t+2 -> the register set tile t came from
        MGGCN_GEMM_STAMP(2);
        mfma_range(As[CUR], Bs[CUR], q0{}, q1{});
        if (more2) load_b(sb[CUR], k0 + 2 * BK);
        mfma_range(As[CUR], Bs[CUR], q1{}, q2{});
        MGGCN_GEMM_STAMP(3);
        if (more1) {                                   // tile t+1 -> the other LDS buffer (nobody reads it until the barrier below)
            finish(sa[NXT], sb[NXT], k0 + BK);
            sa[NXT].store(As[NXT], tid);
        }
        MGGCN_GEMM_STAMP(4);
        mfma_range(As[CUR], Bs[CUR], q2{}, q3{});
        if (more1) {
            add_colsum(sb[NXT]);
            sb[NXT].store(Bs[NXT], tid);
        }
        mfma_range(As[CUR], Bs[CUR], q3{}, q4{});
#else
        if (MGGCN_GEMM_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (MGGCN_GEMM_SETPRIO == 2) __builtin_amdgcn_s_setprio(0);
        if (more2) {
            load_a(sa[CUR], k0 + 2 * BK);
            load_b(sb[CUR], k0 + 2 * BK);
        }
        MGGCN_GEMM_STAMP(2);
        if (MGGCN_GEMM_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (MGGCN_GEMM_SETPRIO == 2) __builtin_amdgcn_s_setprio(1);
        mfma_range(As[CUR], Bs[CUR], q0{}, q2{});
        MGGCN_GEMM_STAMP(3);
        if (MGGCN_GEMM_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
        if (MGGCN_GEMM_SETPRIO == 2) __builtin_amdgcn_s_setprio(0);
        if (more1) {
            finish(sa[NXT], sb[NXT], k0 + BK);
            add_colsum(sb[NXT]);
            sa[NXT].store(As[NXT], tid);
            sb[NXT].store(Bs[NXT], tid);
        }
        MGGCN_GEMM_STAMP(4);
        if (MGGCN_GEMM_SETPRIO == 1) __builtin_amdgcn_s_setprio(0);
        if (MGGCN_GEMM_SETPRIO == 2) __builtin_amdgcn_s_setprio(1);
        mfma_range(As[CUR], Bs[CUR], q2{}, q4{});
        if (MGGCN_GEMM_SETPRIO == 1) __builtin_amdgcn_s_setprio(1);
#endif
        MGGCN_GEMM_STAMP(5);
        __syncthreads();
        MGGCN_GEMM_STAMP(6);
    };
    if (k_begin < k_end) {
        load_a(sa[0], k_begin);
        load_b(sb[0], k_begin);
        if (k_begin + BK < k_end) {
            load_a(sa[1], k_begin + BK);
            load_b(sb[1], k_begin + BK);
        }
        finish(sa[0], sb[0], k_begin);
        add_colsum(sb[0]);
        sa[0].store(As[0], tid);
        sb[0].store(Bs[0], tid);
    }
    __syncthreads();
    MGGCN_GEMM_STAMP(9);
    for (long long k0 = k_begin; k0 < k_end; k0 += 2 * BK) {
        step(k0, std::integral_constant<int, 0>{});
        if (k0 + BK < k_end) step(k0 + BK, std::integral_constant<int, 1>{});
    }
    if constexpr (!B_KCONTIG) {
        if (do_colsum) {                 // block-uniform.  Fold the NT / (BN/4) threads of a column group in order.
            float4 *red = reinterpret_cast<float4 *>(As[0]);          // all tiles are consumed: reuse the LDS
            red[tid] = csum;
            __syncthreads();
            if (tid < BN / 4) {
                float4 t = red[tid];
#pragma unroll 4                          // fully unrolled (32 float4 live at BN = 64) this fold alone spilled to scratch
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

    MGGCN_GEMM_STAMP(7);
    // epilogue.  C/D layout of 32x32 MFMA: col = lane&31, row = (r&3) + 8*(r>>2) + 4*(lane>>5).
    // The kind of epilogue and "is this tile interior" are block-uniform and become COMPILE-TIME parameters of the code
    // that runs (eight straight-line variants): per 32x32 block the 16 auxiliary reads (old C for beta != 0, Z for the
    // leaky-ReLU mask) are issued together, then 16 stores back to back; an element's address is the lane's block corner
    // plus a wave-uniform offset.  History: (1) epilogue_value() per element -- a 64-bit multiply, three branches and, with
    // a mask or beta, a memory round trip per element; (2) the kind as a run-time value inside the unrolled loops -- the
    // compiler merged the paths' outstanding loads and put s_waitcnt vmcnt(0) around EVERY store: 32 dependent memory
    // round trips per wave, 24-30 k cycles, 45 % of a wave's life at K = 128 (gemm_timeline, r02).
    const bool to_slab = gridDim.z > 1;
    float *out = to_slab ? slab + (size_t)blockIdx.z * (M + (epi.colsum ? 1 : 0)) * N : C;
    const size_t ldo = to_slab ? (size_t)N : ldc;
    enum { kPlain, kBias, kMask, kBeta };
    const int kind = to_slab ? kPlain : epi.bias ? kBias : epi.mask ? kMask : beta != 0.f ? kBeta : kPlain;
    const float scale = to_slab ? 1.f : alpha;
    auto emit = [&](auto kind_c, auto full_c) {
        constexpr int KIND = decltype(kind_c)::value;
        constexpr bool FULL = decltype(full_c)::value;          // every row and column of the tile is inside C
#pragma unroll
        for (int i = 0; i < MI; i++)
#pragma unroll
            for (int j = 0; j < NI; j++) {
                const long long col = n0 + wn * (BN / WN) + j * 32 + l31;
                const long long row0 = m0 + wm * (BM / WM) + i * 32 + 4 * lhi;
                const bool col_ok = FULL || col < (long long)N;
                const long long rows_left = (long long)M - row0;      // element r is in range iff its row offset < rows_left
                float *corner = out + (size_t)row0 * ldo + col;      // (only dereferenced where the element exists)
                float aux[16];
                if constexpr (KIND == kMask || KIND == kBeta) {
                    const float *src = KIND == kMask ? epi.mask + (size_t)row0 * epi.ldz + col : corner;
                    const size_t lds = KIND == kMask ? epi.ldz : ldo;
#pragma unroll
                    for (int r = 0; r < 16; r++) {
                        const int ro = (r & 3) + 8 * (r >> 2);
                        aux[r] = (FULL || (col_ok && ro < rows_left)) ? src[(size_t)ro * lds] : 0.f;
                    }
                }
                float bias = 0.f;
                if constexpr (KIND == kBias) bias = col_ok ? epi.bias[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 16; r++) {
                    const int ro = (r & 3) + 8 * (r >> 2);
                    const float av = scale * acc[i][j][r];
                    float v = av;
                    // bias: the row vector the reference broadcasts into C before its beta = 1 sgemm (src/gcn.hpp:116-123),
                    // same single rounding as fmaf(1, bias, alpha*v); mask: leaky_relu_backward of the consumer
                    // (src/cuda_utils.cu:33-38) on the value a beta = 0 GEMM would store
                    if constexpr (KIND == kBias) v = fmaf(1.f, bias, av);
                    if constexpr (KIND == kMask) v = aux[r] > 0.f ? av : epi.slope * av;
                    if constexpr (KIND == kBeta) v = fmaf(beta, aux[r], av);
                    if (FULL || (col_ok && ro < rows_left)) corner[(size_t)ro * ldo] = v;
                }
            }
    };
    using std::integral_constant;
    const bool tile_full = m0 + BM <= (long long)M && n0 + BN <= (long long)N;      // block-uniform
    if (tile_full) {
        if (kind == kPlain) emit(integral_constant<int, kPlain>{}, std::true_type{});
        else if (kind == kBias) emit(integral_constant<int, kBias>{}, std::true_type{});
        else if (kind == kMask) emit(integral_constant<int, kMask>{}, std::true_type{});
        else emit(integral_constant<int, kBeta>{}, std::true_type{});
    } else {
        if (kind == kPlain) emit(integral_constant<int, kPlain>{}, std::false_type{});
        else if (kind == kBias) emit(integral_constant<int, kBias>{}, std::false_type{});
        else if (kind == kMask) emit(integral_constant<int, kMask>{}, std::false_type{});
        else emit(integral_constant<int, kBeta>{}, std::false_type{});
    }
    MGGCN_GEMM_STAMP(8);
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
// Tile width: 128 columns per workgroup when N allows -- unless that tiling leaves most of the chip idle and the product is
// too short for split-K to make up for it (K < 16 steps): a rank's share at P = 8 of the hidden-layer products,
// [29 121 x 128] . [128 x 128], is 228 workgroups of 512 threads at two per CU -- 114 of 256 CUs.  64-wide tiles double the
// workgroups (the A tile is read twice, from L2).  The single-GPU shapes (M = 232 968: 1 821 tiles) are not touched.
uint32_t pick_bn(uint32_t M, uint32_t N, uint32_t K) {
    if (N <= 64) return 64;
    static const bool always_wide = [] { const char *e = std::getenv("MGGCN_GEMM_WIDE_TILES"); return e && std::atoi(e) != 0; }();   // experiment knob
    if (always_wide) return 128;
    const uint64_t tiles128 = (uint64_t)((M + BM - 1) / BM) * ((N + 127) / 128);
    const uint32_t k_steps = (K + BK - 1) / BK;
    return (tiles128 < (uint64_t)kNumCU && k_steps < 16) ? 64 : 128;
}

Split choose_split(uint32_t M, uint32_t N, uint32_t K) {
    const uint32_t bn = pick_bn(M, N, K);
    const uint64_t tiles = (uint64_t)((M + BM - 1) / BM) * ((N + bn - 1) / bn);
    const uint32_t k_steps = (K + BK - 1) / BK;
    uint32_t splits = 1;
    if (tiles < (uint64_t)kNumCU && k_steps >= 16) {
        const uint32_t want = (uint32_t)std::max<uint64_t>(1, (2ull * kNumCU) / tiles);
        splits = std::min<uint32_t>(want, k_steps / 8);   // at least 8 K-steps per slice
        splits = std::max<uint32_t>(splits, 1u);
    }
    // tuning knob (profiles/experiments/gemm_splits.py), read ONCE: this runs on every GEMM dispatch and workspace query
    static const uint32_t forced = [] { const char *e = std::getenv("MGGCN_GEMM_SPLITS"); return e ? (uint32_t)std::strtoul(e, nullptr, 10) : 0u; }();
    if (forced) splits = std::max<uint32_t>(1u, std::min<uint32_t>(forced, k_steps));
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
    // 16-byte loads only when the operand's contiguous extent is itself a multiple of 4: with ld % 4 == 0 but an odd
    // extent (a sub-view with ld > extent) the float4 that straddles the last stored element of the LAST row would read
    // up to 12 bytes past a (rows - 1) * ld + extent allocation -- the ABI only demands ld >= extent
    const bool a_vec = aligned16(A) && lda % 4 == 0 && (trans_a ? M : K) % 4 == 0;
    const bool b_vec = aligned16(B) && ldb % 4 == 0 && (trans_b ? K : N) % 4 == 0;
    const bool a_kc = !trans_a, b_kc = trans_b != 0;
    const int bn = (int)pick_bn(M, N, K);
    const dim3 grid((M + BM - 1) / BM, (N + bn - 1) / bn, sp.splits);

    const dim3 blk(512);
    // experiment knob: extra dynamic LDS per workgroup (bytes) to cap the workgroups resident per CU
    static const unsigned dyn_lds = [] { const char *e = std::getenv("MGGCN_GEMM_DYN_LDS"); return e ? (unsigned)std::atoi(e) : 0u; }();
#define MGGCN_GEMM_LAUNCH(AK, BKC, BNV, VECV)                                                             \
    hipLaunchKernelGGL((gemm_mfma_kernel<AK, BKC, BNV, 512, VECV>), grid, blk, dyn_lds, st, M, N, K, alpha, A, lda, B, \
                       ldb, beta, C, ldc, slab, sp.k_chunk, epi)
#define MGGCN_GEMM_LAUNCH_NT(AK, BKC, BNV) \
    do { if (a_vec && b_vec) MGGCN_GEMM_LAUNCH(AK, BKC, BNV, true); else MGGCN_GEMM_LAUNCH(AK, BKC, BNV, false); } while (0)
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
