// spmm_sweep.hip -- column-panel "sweep" SpMM for gfx950: the L2-resident gather.
//
// Why: the row-per-wave kernel (spmm.hip) already runs at the Infinity-Cache gather
// rate (8.7 TB/s of 512-B row fetches on the Reddit shape, rocprof r01: 50 GB leave
// the XCD L2s per launch against 1.16 GB of algorithmic bytes, L2 hit rate 17 %).
// The same kernel with every column index confined to a 2 MiB window of B runs 2.9x
// faster (22 TB/s, profiles/experiments/l2_window.py).  B as a whole (119 MB) cannot
// live in a 4 MiB L2 -- but a PANEL of it can, if every wave on the chip walks the
// columns in the same order at the same pace.
//
// How: at plan time the matrix is re-cut into equal-work tasks, one per wave64:
//   * a task owns <= RW output rows (heavy rows are first cut into slices, as in the
//     row-split plan; slices get partial-sum slots combined in a fixed order);
//   * tasks are balanced by non-zero count (longest-processing-time greedy), so every
//     wave has the same amount of work;
//   * a task's non-zeros are stored as ONE stream sorted by (column panel, row): the
//     wave sweeps the column space panel by panel.  Entry = 8 bytes:
//     {run_start:1 | row_local:4 | column:27, value}.
//   * the wave keeps its RW accumulator rows in REGISTERS (16 rows x 4 planes); a run of
//     entries with the same row is summed in a small accumulator and added to its row at
//     the run boundary through the VGPR index mode (wave-uniform row number): no LDS, no
//     atomics, fixed order -> bitwise reproducible.
// Because all resident waves start together (one launch per "round" of resident
// tasks), have equal work and see the same column distribution, they cross each
// panel at about the same time: the panel is pulled from the Infinity Cache once per
// XCD and then hit in L2 by ~500 waves.  Placement is a speed matter only; results
// never depend on it.
//
// Kernels in this file (dispatch: sweep_launch):
//   spmm_sweep_pair_kernel<FAST>  d % 4 == 0, d >= 96: two 512-byte rows per buffer_load_dwordx4; <true> when the row pitch
//                                 is a power of two (d = 128 contiguous): four instructions fewer per pair
//   spmm_sweep_quad_lds_kernel<L> d <= 64 (plan built with a width hint): 64/L rows per load,
//                                 entries staged through LDS
//   spmm_sweep_kernel<1|2>        every other width / alignment: one row per load
//   sweep_repack_kernel           B -> 64-byte-multiple pitch for the narrow form
//   sweep_combine_kernel          partial rows of heavy (sliced) rows, fixed order
#include <algorithm>
#include <cmath>
#include <cstring>
#include <queue>
#include <thread>
#include <type_traits>
#include <vector>

#include "plan_host.h"
#include "spmm_internal.h"

// packed-entry constants, task / split-row records and the host passes of the plan builder: plan_host.h (free of HIP, so
// that the sanitizers can run it on the CPU)
using namespace mggcn_plan;

// (hipcc 7.2's __builtin_amdgcn_raw_buffer_load_b64/_b128 lower to a single dword load
// splatted over the result -- checked in the IR -- so the LLVM intrinsics are bound directly.)
typedef float f32x2_t __attribute__((ext_vector_type(2)));
__device__ f32x2_t mggcn_buffer_load_v2f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v2f32");
__device__ float mggcn_buffer_load_f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.f32");
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ f32x4_t mggcn_buffer_load_v4f32(__amdgpu_buffer_rsrc_t rsrc, int voffset, int soffset, int aux)
    __asm("llvm.amdgcn.raw.ptr.buffer.load.v4f32");

// cache-policy bits of the pair kernel's row gathers (experiment, profiles/experiments/gather_aux.sh): gfx940+ encoding
// 1 = sc0, 2 = nt, 16 = sc1
#ifndef MGGCN_GATHER_AUX
#define MGGCN_GATHER_AUX 0
#endif

namespace {

// Priority rotation.  The four waves that share a SIMD are served oldest first: measured per hardware wave slot
// (= blockIdx / 256: blocks are dealt one per CU per "layer"; profiles/experiments/wave_spread.py) equal tasks took
// 305 / 308 / 317 / 330 us -- the youngest wave 8 % slower all launch long, so the waves spread over +-6 % of the
// column slice (two panels: the resident window outgrows the L2) and every launch ended in an 8 % tail.  Every
// 2^shift entries each wave moves one priority level on (level = slot + phase mod 4): every wave spends the same share
// of the launch at every level -> 299 / 298 / 296 / 300 us, d = 128 SpMM 2.59 -> 2.44 ms forward, 2.73 -> 2.52 backward
// (profiles/experiments/prio_rotate_r02.log).  Speed only: results never depend on it.
__device__ __forceinline__ void rotate_priority(uint32_t slot, uint32_t phase) {
    switch ((slot + phase) & 3u) {
        case 0: __builtin_amdgcn_s_setprio(0); break;
        case 1: __builtin_amdgcn_s_setprio(1); break;
        case 2: __builtin_amdgcn_s_setprio(2); break;
        default: __builtin_amdgcn_s_setprio(3); break;
    }
}

__device__ __forceinline__ float lrelu(float x, float slope) {
    const float y = slope * x;
    return x > y ? x : y;
}

template <int VEC> struct Vec;
template <> struct Vec<1> {
    float v;
    __device__ __forceinline__ void zero() { v = 0.f; }
    __device__ __forceinline__ void fma(float s, const Vec &b) { v = fmaf(s, b.v, v); }
    __device__ __forceinline__ void add(const Vec &b) { v += b.v; }
    __device__ __forceinline__ static Vec load(const float *p) { Vec r; r.v = *p; return r; }
    __device__ __forceinline__ void store(float *p) const { *p = v; }
    template <typename F> __device__ __forceinline__ void map(const float *c, F f) { v = f(v, c[0]); }
};
template <> struct Vec<2> {
    float2 v;
    __device__ __forceinline__ void zero() { v = make_float2(0.f, 0.f); }
    __device__ __forceinline__ void fma(float s, const Vec &b) { v.x = fmaf(s, b.v.x, v.x); v.y = fmaf(s, b.v.y, v.y); }
    __device__ __forceinline__ void add(const Vec &b) { v.x += b.v.x; v.y += b.v.y; }
    __device__ __forceinline__ static Vec load(const float *p) { Vec r; r.v = *reinterpret_cast<const float2 *>(p); return r; }
    __device__ __forceinline__ void store(float *p) const { *reinterpret_cast<float2 *>(p) = v; }
    template <typename F> __device__ __forceinline__ void map(const float *c, F f) { v.x = f(v.x, c[0]); v.y = f(v.y, c[1]); }
};

// Entry batches are 8 entries = 64 bytes, 64-byte aligned (the plan pads every task's
// stream to a multiple of 8 with zero-valued entries), so one s_load_dwordx16 brings a
// batch straight into scalar registers: no VGPRs, no cross-lane broadcast, and -- because
// scalar loads count on lgkmcnt, not vmcnt -- the HBM-latency entry stream never sits in
// front of the L2-latency row gathers in the in-order vector-memory queue.
struct EntryBatch {
    uint4 q[4];
    __device__ __forceinline__ uint32_t pk(int u) const {
        const uint4 &x = q[u >> 1];
        return (u & 1) ? x.z : x.x;
    }
    __device__ __forceinline__ float val(int u) const {
        const uint4 &x = q[u >> 1];
        return __builtin_bit_cast(float, (u & 1) ? x.w : x.y);
    }
};

__device__ __forceinline__ EntryBatch load_batch(const uint2 *__restrict__ entries, uint32_t e) {
    const uint4 *p = reinterpret_cast<const uint4 *>(entries + e);   // wave-uniform address
    EntryBatch r;
    r.q[0] = p[0]; r.q[1] = p[1]; r.q[2] = p[2]; r.q[3] = p[3];
    return r;
}

// The wave's RW = 16 accumulator rows live in REGISTERS: one 16-wide vector register
// group per feature plane (plane k, element r = feature k of row r for this lane).  A run
// of entries with the same row is summed in `acc`; at the run boundary `acc` is added to
// element `row` of each plane.  `row` is wave-uniform, so the compiler indexes the VGPR
// group through the gfx9 index mode (s_set_gpr_idx_on): two moves per plane, no branch
// tree, no LDS, no memory wait on the fold.  (Measured alternatives: LDS float atomics
// serialise per lane, ~150 cycles per ds_add_f32 wave op; LDS read-modify-write puts an
// lgkmcnt wait on every run boundary.)
typedef float f32x16 __attribute__((ext_vector_type(16)));

// NOTE: the planes must be separate LOCAL vector variables.  Wrapped in a struct (or an
// array of vectors) the compiler keeps them in scratch memory (192 B) instead of using the
// index mode -- checked in the ISA.
#define MGGCN_ROWS_GET(R) (VEC == 2 ? make_float2(p0[R], p1[R]) : make_float2(p0[R], 0.f))

// ---- accumulator planes in RESERVED registers (float4 kernels) -------------------------------
// The float4 kernels keep their 4 x 16 row accumulators in v[64:127], outside the register
// allocator (the kernels are compiled with amdgpu_num_vgpr(64); the asm clobber lists make the
// kernel's VGPR count 128 = four waves per SIMD, the residency the sweep wants anyway).  A run is
// summed in compiler-managed registers (`acc`, packed FMAs, freely scheduled); at the run boundary
// it is added to row `cur_row` of every plane with the index mode on the add itself:
//     s_set_gpr_idx_on row, gpr_idx(SRC0,DST);  v_add_f32 v64, v64, acc.x;  ... x4 ;  s_set_gpr_idx_off
// six instructions.  The compiler's own lowering of `p[cur_row] += acc` on a vector variable is
// seven per plane (on / mov / off / add / on / mov / off): it cost 0.22 ms of the 2.9 ms d = 128
// SpMM at 8192-row panels and 0.44 ms at 4096 (diag run with the fold skipped, r01).  (Indexing
// the FMAs themselves -- no `acc`, no fold -- was slower: an on/off pair around every group of
// four FMAs and no packed FMA, 2.97 ms.)
#define MGGCN_PLANE_CLOBBERS "v64","v65","v66","v67","v68","v69","v70","v71","v72","v73","v74","v75","v76","v77","v78","v79","v80","v81","v82","v83","v84","v85","v86","v87","v88","v89","v90","v91","v92","v93","v94","v95","v96","v97","v98","v99","v100","v101","v102","v103","v104","v105","v106","v107","v108","v109","v110","v111","v112","v113","v114","v115","v116","v117","v118","v119","v120","v121","v122","v123","v124","v125","v126","v127"
#define MGGCN_PLANES_ZERO() asm volatile("v_mov_b32 v64, 0\n\tv_mov_b32 v65, 0\n\tv_mov_b32 v66, 0\n\tv_mov_b32 v67, 0\n\tv_mov_b32 v68, 0\n\tv_mov_b32 v69, 0\n\tv_mov_b32 v70, 0\n\tv_mov_b32 v71, 0\n\tv_mov_b32 v72, 0\n\tv_mov_b32 v73, 0\n\tv_mov_b32 v74, 0\n\tv_mov_b32 v75, 0\n\tv_mov_b32 v76, 0\n\tv_mov_b32 v77, 0\n\tv_mov_b32 v78, 0\n\tv_mov_b32 v79, 0\n\tv_mov_b32 v80, 0\n\tv_mov_b32 v81, 0\n\tv_mov_b32 v82, 0\n\tv_mov_b32 v83, 0\n\tv_mov_b32 v84, 0\n\tv_mov_b32 v85, 0\n\tv_mov_b32 v86, 0\n\tv_mov_b32 v87, 0\n\tv_mov_b32 v88, 0\n\tv_mov_b32 v89, 0\n\tv_mov_b32 v90, 0\n\tv_mov_b32 v91, 0\n\tv_mov_b32 v92, 0\n\tv_mov_b32 v93, 0\n\tv_mov_b32 v94, 0\n\tv_mov_b32 v95, 0\n\tv_mov_b32 v96, 0\n\tv_mov_b32 v97, 0\n\tv_mov_b32 v98, 0\n\tv_mov_b32 v99, 0\n\tv_mov_b32 v100, 0\n\tv_mov_b32 v101, 0\n\tv_mov_b32 v102, 0\n\tv_mov_b32 v103, 0\n\tv_mov_b32 v104, 0\n\tv_mov_b32 v105, 0\n\tv_mov_b32 v106, 0\n\tv_mov_b32 v107, 0\n\tv_mov_b32 v108, 0\n\tv_mov_b32 v109, 0\n\tv_mov_b32 v110, 0\n\tv_mov_b32 v111, 0\n\tv_mov_b32 v112, 0\n\tv_mov_b32 v113, 0\n\tv_mov_b32 v114, 0\n\tv_mov_b32 v115, 0\n\tv_mov_b32 v116, 0\n\tv_mov_b32 v117, 0\n\tv_mov_b32 v118, 0\n\tv_mov_b32 v119, 0\n\tv_mov_b32 v120, 0\n\tv_mov_b32 v121, 0\n\tv_mov_b32 v122, 0\n\tv_mov_b32 v123, 0\n\tv_mov_b32 v124, 0\n\tv_mov_b32 v125, 0\n\tv_mov_b32 v126, 0\n\tv_mov_b32 v127, 0" ::: MGGCN_PLANE_CLOBBERS)
// (s_set_gpr_idx_on writes M0.  hipcc rejects "m0" on a clobber list as a reserved register, so the
//  contract is guarded at build time instead: tests/test_abi_host.py fails if any compiler-generated
//  instruction of these kernels names m0 -- on gfx9 plain LDS / buffer ops do not use it.)
#define MGGCN_PLANES_FOLD(ROW, ACC)                                                             \
    asm volatile("s_set_gpr_idx_on %0, gpr_idx(SRC0,DST)\n\t"                                   \
                 "v_add_f32 v64, v64, %1\n\t"                                                   \
                 "v_add_f32 v80, v80, %2\n\t"                                                   \
                 "v_add_f32 v96, v96, %3\n\t"                                                   \
                 "v_add_f32 v112, v112, %4\n\t"                                                 \
                 "s_set_gpr_idx_off"                                                             \
                 :: "s"(ROW), "v"((ACC)[0]), "v"((ACC)[1]), "v"((ACC)[2]), "v"((ACC)[3])         \
                 : MGGCN_PLANE_CLOBBERS)
#define MGGCN_PLANES_GET(X0, X1, X2, X3, R0, R1, R2, R3)                                        \
    asm volatile("v_mov_b32 %0, " R0 "\n\tv_mov_b32 %1, " R1 "\n\tv_mov_b32 %2, " R2 "\n\tv_mov_b32 %3, " R3 \
                 : "=v"(X0), "=v"(X1), "=v"(X2), "=v"(X3) :: MGGCN_PLANE_CLOBBERS)
#define MGGCN_PLANES_EMIT_ONE(EMIT, R, R0, R1, R2, R3)                                          \
    { float x0_, x1_, x2_, x3_; MGGCN_PLANES_GET(x0_, x1_, x2_, x3_, R0, R1, R2, R3); EMIT(R, x0_, x1_, x2_, x3_); }
// EMIT4(R0, x[4][4]) for R0 = 0, 4, 8, 12: rows R0..R0+3, x[k][c] = plane c of row R0 + k
#define MGGCN_PLANES_EMIT_FOURS(EMIT4) \
    { float x_[4][4]; MGGCN_PLANES_GET(x_[0][0], x_[0][1], x_[0][2], x_[0][3], "v64", "v80", "v96", "v112"); MGGCN_PLANES_GET(x_[1][0], x_[1][1], x_[1][2], x_[1][3], "v65", "v81", "v97", "v113"); MGGCN_PLANES_GET(x_[2][0], x_[2][1], x_[2][2], x_[2][3], "v66", "v82", "v98", "v114"); MGGCN_PLANES_GET(x_[3][0], x_[3][1], x_[3][2], x_[3][3], "v67", "v83", "v99", "v115"); EMIT4(0, x_); } \
    { float x_[4][4]; MGGCN_PLANES_GET(x_[0][0], x_[0][1], x_[0][2], x_[0][3], "v68", "v84", "v100", "v116"); MGGCN_PLANES_GET(x_[1][0], x_[1][1], x_[1][2], x_[1][3], "v69", "v85", "v101", "v117"); MGGCN_PLANES_GET(x_[2][0], x_[2][1], x_[2][2], x_[2][3], "v70", "v86", "v102", "v118"); MGGCN_PLANES_GET(x_[3][0], x_[3][1], x_[3][2], x_[3][3], "v71", "v87", "v103", "v119"); EMIT4(4, x_); } \
    { float x_[4][4]; MGGCN_PLANES_GET(x_[0][0], x_[0][1], x_[0][2], x_[0][3], "v72", "v88", "v104", "v120"); MGGCN_PLANES_GET(x_[1][0], x_[1][1], x_[1][2], x_[1][3], "v73", "v89", "v105", "v121"); MGGCN_PLANES_GET(x_[2][0], x_[2][1], x_[2][2], x_[2][3], "v74", "v90", "v106", "v122"); MGGCN_PLANES_GET(x_[3][0], x_[3][1], x_[3][2], x_[3][3], "v75", "v91", "v107", "v123"); EMIT4(8, x_); } \
    { float x_[4][4]; MGGCN_PLANES_GET(x_[0][0], x_[0][1], x_[0][2], x_[0][3], "v76", "v92", "v108", "v124"); MGGCN_PLANES_GET(x_[1][0], x_[1][1], x_[1][2], x_[1][3], "v77", "v93", "v109", "v125"); MGGCN_PLANES_GET(x_[2][0], x_[2][1], x_[2][2], x_[2][3], "v78", "v94", "v110", "v126"); MGGCN_PLANES_GET(x_[3][0], x_[3][1], x_[3][2], x_[3][3], "v79", "v95", "v111", "v127"); EMIT4(12, x_); }
// EMIT(R, x0, x1, x2, x3) for R = 0..15, the plane registers spelled out
#define MGGCN_PLANES_EMIT_ALL(EMIT) \
    MGGCN_PLANES_EMIT_ONE(EMIT, 0, "v64", "v80", "v96", "v112") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 1, "v65", "v81", "v97", "v113") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 2, "v66", "v82", "v98", "v114") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 3, "v67", "v83", "v99", "v115") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 4, "v68", "v84", "v100", "v116") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 5, "v69", "v85", "v101", "v117") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 6, "v70", "v86", "v102", "v118") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 7, "v71", "v87", "v103", "v119") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 8, "v72", "v88", "v104", "v120") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 9, "v73", "v89", "v105", "v121") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 10, "v74", "v90", "v106", "v122") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 11, "v75", "v91", "v107", "v123") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 12, "v76", "v92", "v108", "v124") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 13, "v77", "v93", "v109", "v125") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 14, "v78", "v94", "v110", "v126") \
    MGGCN_PLANES_EMIT_ONE(EMIT, 15, "v79", "v95", "v111", "v127")

template <int VEC> __device__ __forceinline__ Vec<VEC> vec_from2(float2 x);
template <> __device__ __forceinline__ Vec<1> vec_from2<1>(float2 x) { Vec<1> r; r.v = x.x; return r; }
template <> __device__ __forceinline__ Vec<2> vec_from2<2>(float2 x) { Vec<2> r; r.v = x; return r; }
template <int VEC> __device__ __forceinline__ float2 vec_to2(const Vec<VEC> &v);
template <> __device__ __forceinline__ float2 vec_to2<1>(const Vec<1> &v) { return make_float2(v.v, 0.f); }
template <> __device__ __forceinline__ float2 vec_to2<2>(const Vec<2> &v) { return v.v; }

// Row gathers go through a buffer descriptor over B: the per-entry address is then ONE
// scalar multiply (column * row bytes -> soffset) and no vector ALU work at all
// (buffer_load_dwordx2 v, v_lane_offset, s[rsrc], s_row_offset offen).  The kernel is
// otherwise bound by scalar-instruction issue, not by memory (rocprof r01: 13 SALU
// instructions per non-zero in the first version).
template <int VEC> struct BufLoad;
template <> struct BufLoad<1> {
    __device__ __forceinline__ static Vec<1> load(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
        Vec<1> r;
        r.v = mggcn_buffer_load_f32(rsrc, (int)voff, (int)soff, 0);
        return r;
    }
};
template <> struct BufLoad<2> {
    __device__ __forceinline__ static Vec<2> load(__amdgpu_buffer_rsrc_t rsrc, uint32_t voff, uint32_t soff) {
        const f32x2_t x = mggcn_buffer_load_v2f32(rsrc, (int)voff, (int)soff, 0);
        Vec<2> r;
        r.v = make_float2(x[0], x[1]);
        return r;
    }
};

template <int VEC>
__global__ __launch_bounds__(64 * kWavesPerBlock) void spmm_sweep_kernel(
    const SweepTask *__restrict__ tasks, uint32_t task0, uint32_t n_launch,
    const uint2 *__restrict__ entries, const uint32_t *__restrict__ task_rows,
    const float *__restrict__ B, uint32_t b_bytes, uint32_t row_bytes, float *__restrict__ C, size_t ldc,
    float *__restrict__ partial, uint32_t d, float alpha, float beta, uint32_t flags, float slope) {
    static_assert(kRW == 16, "the accumulator planes hold 16 rows");
    constexpr int TILE = 64 * VEC;
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t local = blockIdx.x * kWavesPerBlock + wib;
    if (local >= n_launch) return;                    // no barriers: waves are independent
    const uint32_t t = task0 + local;
    const SweepTask task = tasks[t];
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, b_bytes, 0x00020000);

    for (uint32_t col0 = 0; col0 < d; col0 += TILE) {
        const uint32_t col = col0 + lane * VEC;
        const bool active = col < d;                  // VEC == 2 is only used with even d
        const uint32_t lane_off = (active ? col : 0) * 4u;   // idle lanes re-read column 0 (never stored)
        f32x16 p0, p1;
#pragma unroll
        for (int r = 0; r < 16; r++) { p0[r] = 0.f; p1[r] = 0.f; }
        uint32_t cur_row = 0;                         // row 0 is a valid sink for an empty prefix
        Vec<VEC> acc; acc.zero();

        if (task.beg < task.end) {
            // NOTE: keep this loop body inline.  Wrapping it in a lambda that captures p0/p1 by
            // reference sends the accumulators to scratch memory (measured: 2.0 -> 14 ms at d = 41).
            // Two SGPR batch sets used alternately (a macro, not a lambda): this loop is bound by
            // scalar issue, and copying the 16 batch registers every 8 entries costs 2 SALU per entry.
#define MGGCN_SWEEP_BODY(cur)                                                                     \
            {                                                                                     \
                Vec<VEC> b[8];                                                                    \
                _Pragma("unroll") for (int u = 0; u < 8; u++)                                     \
                    b[u] = BufLoad<VEC>::load(rsrc, lane_off, (cur.pk(u) & kColMask) * row_bytes); \
                _Pragma("unroll") for (int u = 0; u < 8; u++) {                                   \
                    const uint32_t pk = cur.pk(u);                                                \
                    if (pk & kRunFlag) { /* first entry of a (panel,row) run */                   \
                        const float2 a2 = vec_to2<VEC>(acc);                                      \
                        p0[cur_row] += a2.x;                                                      \
                        if (VEC == 2) p1[cur_row] += a2.y;                                        \
                        acc.zero();                                                               \
                        cur_row = (pk >> kColBits) & (kRW - 1);                                   \
                    }                                                                             \
                    acc.fma(cur.val(u), b[u]);                                                    \
                }                                                                                 \
            }
            const uint32_t last = task.end - 8;
            EntryBatch A = load_batch(entries, task.beg);
            for (uint32_t e = task.beg;; e += 16) {
                const EntryBatch Bn = load_batch(entries, min(e + 8, last));
                MGGCN_SWEEP_BODY(A)
                if (e + 8 >= task.end) break;
                A = load_batch(entries, min(e + 16, last));
                MGGCN_SWEEP_BODY(Bn)
                if (e + 16 >= task.end) break;
            }
#undef MGGCN_SWEEP_BODY
        }
        {
            const float2 a2 = vec_to2<VEC>(acc);
            p0[cur_row] += a2.x;
            if (VEC == 2) p1[cur_row] += a2.y;
        }
        if (active) {
            // static row numbers only: the accumulators stay in registers
            auto emit = [&](uint32_t r, Vec<VEC> s) {
                if (r >= task.n_rows) return;
                const uint32_t dst = task_rows[(size_t)t * kRW + r];
                if (dst & kSlotFlag) {
                    s.store(partial + (size_t)(dst & ~kSlotFlag) * d + col);
                    return;
                }
                float *cp = C + (size_t)dst * ldc + col;
                // epilogue: alpha, beta (C is only read when beta != 0), optional leaky-ReLU
                Vec<VEC> c0; c0.zero();
                if (beta != 0.f) c0 = Vec<VEC>::load(cp);
                s.map(reinterpret_cast<const float *>(&c0), [=](float x, float c) {
                    float o = alpha * x;
                    if (beta != 0.f) o = fmaf(beta, c, o);
                    return (flags & MGGCN_SPMM_LEAKY_RELU) ? lrelu(o, slope) : o;
                });
                s.store(cp);
            };
            emit(0, vec_from2<VEC>(MGGCN_ROWS_GET(0)));   emit(1, vec_from2<VEC>(MGGCN_ROWS_GET(1)));
            emit(2, vec_from2<VEC>(MGGCN_ROWS_GET(2)));   emit(3, vec_from2<VEC>(MGGCN_ROWS_GET(3)));
            emit(4, vec_from2<VEC>(MGGCN_ROWS_GET(4)));   emit(5, vec_from2<VEC>(MGGCN_ROWS_GET(5)));
            emit(6, vec_from2<VEC>(MGGCN_ROWS_GET(6)));   emit(7, vec_from2<VEC>(MGGCN_ROWS_GET(7)));
            emit(8, vec_from2<VEC>(MGGCN_ROWS_GET(8)));   emit(9, vec_from2<VEC>(MGGCN_ROWS_GET(9)));
            emit(10, vec_from2<VEC>(MGGCN_ROWS_GET(10))); emit(11, vec_from2<VEC>(MGGCN_ROWS_GET(11)));
            emit(12, vec_from2<VEC>(MGGCN_ROWS_GET(12))); emit(13, vec_from2<VEC>(MGGCN_ROWS_GET(13)));
            emit(14, vec_from2<VEC>(MGGCN_ROWS_GET(14))); emit(15, vec_from2<VEC>(MGGCN_ROWS_GET(15)));
        }
    }
}

// ---------------------------------------------------------------------------------------
// float4 "pair" form (d % 4 == 0, 16-byte aligned rows): the 8-byte gathers of the kernel
// above top out at ~17.5 TB/s even with every access hitting L2 (8-B accesses run at
// 0.54-0.70x the 16-B rate, MI355X_MICROARCH.md), the 16-byte row-split kernel reaches 22.
// Here a 128-wide row is 32 lanes x float4 and ONE buffer_load_dwordx4 fetches the two rows
// of an entry pair, one per half-wave (the plan made every run even and ordered each pair by
// column, so the upper half's extra offset is non-negative).  Both halves accumulate partial
// sums of the SAME output row (the run's row is still wave-uniform -> index-mode fold); the
// halves are added once per task at write-out.
// ---------------------------------------------------------------------------------------
// FAST (row pitch a power of two >= 512 bytes, i.e. d = 128 contiguous): three scalar and two vector instructions fewer per pair.
//   * offset = packed word x pitch WITHOUT masking the flag / row bits off first: they sit at bit 27 and up, the pitch is a
//     multiple of 32, so they leave the 32-bit product;
//   * the upper half's extra offset is a multiple of the pitch and the lane's own offset is below it: OR instead of ADD, fused
//     with the AND of the half mask (v_and_or_b32);
//   * the half's value through v_bfi_b32 (one move + one select instead of three instructions).
// The per-wave instruction chain is on the critical path of this kernel (profiles/experiments/split_pairs_r03.log: six
// more instructions per pair cost 16 % per batch).
template <bool FAST>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_num_vgpr(64))) void spmm_sweep_pair_kernel(
    const SweepTask *__restrict__ tasks, uint32_t task0, uint32_t n_launch,
    const uint2 *__restrict__ entries, const uint32_t *__restrict__ task_rows,
    const float *__restrict__ B, uint32_t b_bytes, uint32_t row_bytes, float *__restrict__ C, size_t ldc,
    float *__restrict__ partial, uint32_t d, float alpha, float beta, uint32_t flags, float slope,
    unsigned long long *__restrict__ stamps, uint32_t wave_stride) {
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t local = blockIdx.x * kWavesPerBlock + wib;
    if (local >= n_launch) return;                    // no barriers: waves are independent
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, b_bytes, 0x00020000);
    const uint32_t sub = lane & 31;
    const uint32_t hmask = (lane & 32) ? 0xFFFFFFFFu : 0u;     // upper half-wave takes the pair's 2nd entry
    const uint32_t prio_slot = blockIdx.x / ((flags >> kNumCuPos) & kNumCuMask);   // blocks are dealt one per CU per "layer": layer = hardware wave slot
    const uint32_t prio_shift = (flags >> kPrioShiftPos) & 15u, prio_mask = (1u << prio_shift) - 1u;

    // one task per wave per launch by default (wave_stride >= n_launch); MGGCN_SPMM_TASKS_PER_WAVE > 1 lets a wave walk
    // several tasks of the round table in one launch (experiment: fewer launch tails against longer unsynchronised runs)
    for (uint32_t tl = local; tl < n_launch; tl += wave_stride) {
    const uint32_t t = task0 + tl;
    const SweepTask task = tasks[t];
    // diagnostics (MGGCN_SPMM_STAMPS=1, never in a timed run): start / end of every wave on the 100 MHz constant clock
    const unsigned long long t_start = stamps ? __builtin_amdgcn_s_memrealtime() : 0ull;

    for (uint32_t col0 = 0; col0 < d; col0 += 128) {
        const uint32_t col = col0 + sub * 4;
        const bool active = col < d;
        const uint32_t lane_off = (active ? col : 0) * 4u;
        MGGCN_PLANES_ZERO();
        uint32_t cur_row = 0;
        f32x4_t acc = {0.f, 0.f, 0.f, 0.f};
        const uint32_t my_dst = task_rows[(size_t)t * kRW + (lane & 15)];   // row table, read under the main loop

        if (task.beg < task.end) {
            EntryBatch cur = load_batch(entries, task.beg);
            for (uint32_t e = task.beg; e < task.end; e += 8) {
                if ((flags & kFlagPrioRotate) && ((e - task.beg) & prio_mask) == 0)
                    rotate_priority(prio_slot, (e - task.beg) >> prio_shift);
                const uint32_t e_next = e + 8 < task.end ? e + 8 : e;
                const EntryBatch nxt = load_batch(entries, e_next);
                f32x4_t b[4];
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t off_a = (cur.pk(2 * u) & kColMask) * row_bytes;
                    const uint32_t off_b = (cur.pk(2 * u + 1) & kColMask) * row_bytes;     // >= off_a
                    if constexpr (FAST) {
                        const uint32_t fa = cur.pk(2 * u) * row_bytes, fb = cur.pk(2 * u + 1) * row_bytes;
                        const uint32_t voff = lane_off | ((fb - fa) & hmask);
                        b[u] = mggcn_buffer_load_v4f32(rsrc, (int)voff, (int)fa, MGGCN_GATHER_AUX);
                    } else {
                        const uint32_t voff = lane_off + ((off_b - off_a) & hmask);
                        b[u] = mggcn_buffer_load_v4f32(rsrc, (int)voff, (int)off_a, MGGCN_GATHER_AUX);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t pk = cur.pk(2 * u);
                    if (pk & kRunFlag) {                                   // first pair of a (panel,row) run
                        MGGCN_PLANES_FOLD(cur_row, acc);
                        acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                        cur_row = (pk >> kColBits) & (kRW - 1);
                    }
                    const uint32_t va = __builtin_bit_cast(uint32_t, cur.val(2 * u));
                    const uint32_t vb = __builtin_bit_cast(uint32_t, cur.val(2 * u + 1));
                    float v;
                    if constexpr (FAST) {
                        uint32_t sel;
                        asm("v_mov_b32 %0, %1\n\tv_bfi_b32 %0, %2, %3, %0" : "=&v"(sel) : "s"(va), "v"(hmask), "s"(vb));
                        v = __builtin_bit_cast(float, sel);
                    } else {
                        v = __builtin_bit_cast(float, (va & ~hmask) | (vb & hmask));
                    }
                    // (starting the new run from the product v * b instead of zeroing acc, and dropping the clamp of the
                    //  prefetch index, were measured on top of FAST: +2 % and +-0: profiles/experiments/fast_pairs_r03.log)
                    acc[0] = fmaf(v, b[u][0], acc[0]);
                    acc[1] = fmaf(v, b[u][1], acc[1]);
                    acc[2] = fmaf(v, b[u][2], acc[2]);
                    acc[3] = fmaf(v, b[u][3], acc[3]);
                }
                cur = nxt;
            }
        }
        MGGCN_PLANES_FOLD(cur_row, acc);

        // Epilogue, four rows at a time, in three phases with nothing else between the memory operations of a phase:
        // (1) the four reads of C (beta != 0: every column slice after the first), (2) all four results, (3) the four stores.
        // vmcnt counts loads AND stores in order: written row by row (read, combine, store, next row) every row was a
        // dependent memory round trip -- and with the stores inside the per-row branches the compiler put s_waitcnt vmcnt(0)
        // in front of each of them even at beta = 0: sixteen serial round trips at the END of every wave, when no other
        // wave is left to hide them (r02: the same pattern cost the GEMM epilogue 45 % of a wave's life at K = 128).
        // (beta != 0 is a compile-time parameter of the code that runs: with the reads of C merely predicated the
        //  compiler still waited vmcnt(0) in front of every group's loads -- i.e. for the previous group's stores)
        // the row table comes out of its VGPR once, before the first store: a v_readlane of a loaded value in front of
        // every group made the compiler wait vmcnt(0) there -- for the previous group's stores
        uint32_t dst_all[kRW];
#pragma unroll
        for (int r = 0; r < kRW; r++) dst_all[r] = __builtin_amdgcn_readlane(my_dst, r);
        auto emit4_any = [&](auto has_beta_c, uint32_t r0, float (&x)[4][4]) {
            constexpr bool HAS_BETA = decltype(has_beta_c)::value;
            if (r0 >= task.n_rows) return;                                   // wave-uniform
#pragma unroll
            for (int k = 0; k < 4; k++)
#pragma unroll
                for (int c = 0; c < 4; c++) x[k][c] += __shfl_xor(x[k][c], 32);
            const bool writer = active && !hmask;
            uint32_t dst[4];
            bool use[4], slot[4];
            float4 c0[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {                                    // (1)
                dst[k] = dst_all[(r0 + k) & 15];                                  // wave-uniform
                use[k] = r0 + k < task.n_rows;
                slot[k] = (dst[k] & kSlotFlag) != 0;
                c0[k] = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (HAS_BETA) {
                    if (use[k] && !slot[k] && writer)
                        c0[k] = *reinterpret_cast<const float4 *>(C + (size_t)dst[k] * ldc + col);
                }
            }
            float4 res[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {                                    // (2)
                float4 f = make_float4(alpha * x[k][0], alpha * x[k][1], alpha * x[k][2], alpha * x[k][3]);
                if constexpr (HAS_BETA) {
                    f.x = fmaf(beta, c0[k].x, f.x); f.y = fmaf(beta, c0[k].y, f.y);
                    f.z = fmaf(beta, c0[k].z, f.z); f.w = fmaf(beta, c0[k].w, f.w);
                }
                if (flags & MGGCN_SPMM_LEAKY_RELU) {
                    f.x = lrelu(f.x, slope); f.y = lrelu(f.y, slope); f.z = lrelu(f.z, slope); f.w = lrelu(f.w, slope);
                }
                // a slice of a heavy row keeps its raw partial sum (combined, scaled and activated by sweep_combine_kernel)
                res[k] = slot[k] ? make_float4(x[k][0], x[k][1], x[k][2], x[k][3]) : f;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {                                    // (3)
                float *p = slot[k] ? partial + (size_t)(dst[k] & ~kSlotFlag) * d + col : C + (size_t)dst[k] * ldc + col;
                if (use[k] && writer) *reinterpret_cast<float4 *>(p) = res[k];
            }
        };
        auto emit4_beta = [&](uint32_t r0, float (&x)[4][4]) { emit4_any(std::true_type{}, r0, x); };
        auto emit4_nobeta = [&](uint32_t r0, float (&x)[4][4]) { emit4_any(std::false_type{}, r0, x); };
        if (beta != 0.f) { MGGCN_PLANES_EMIT_FOURS(emit4_beta) } else { MGGCN_PLANES_EMIT_FOURS(emit4_nobeta) }
    }
    if (stamps) {                                       // wave-uniform
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));      // wave slot / SIMD / CU / SE of this wave
        if (lane == 0) {
            stamps[3 * (size_t)t + 0] = t_start;
            stamps[3 * (size_t)t + 1] = __builtin_amdgcn_s_memrealtime();
            stamps[3 * (size_t)t + 2] = ((unsigned long long)hw << 32) | ((unsigned long long)(blockIdx.x & 0xFFFFF) << 4) | (xcc & 0xF);
        }
    }
    }   // tasks of this wave
}

// ---------------------------------------------------------------------------------------
// float4 narrow-row form (d <= 64; the reference's logits layer has d = 41, 48 at P = 8).
// Measured on this chip, everything hitting L1/L2 (profiles/experiments/narrow_spmm.py,
// l2_window.py): a row-gather wave-instruction is priced by the 128-byte LINES it touches, about
// 2.7 cycles per line per CU (vector-L1 tag/data rate) on top of ~8-11 for the instruction itself:
// dword x 41 lanes (one 164-byte row, 2-3 lines) 11 cycles, dwordx4 over two 512-byte rows (8 lines)
// 22, over four 176-byte rows 26, over sixteen 64-byte rows 62.  The one-column-per-lane kernel
// spends a whole instruction on one 164-byte row (2.12 ms on the Reddit shape, with or without L2
// locality).  Here B is first re-pitched to a multiple of 64 bytes (sweep_repack_kernel: one
// streaming pass of n_cols x d floats; a 164/176-byte pitch puts 37 % of the rows on three lines,
// a 192-byte pitch never more than two), a row is LPE lanes x float4 and ONE buffer_load_dwordx4
// fetches the rows of G = 64 / LPE entries (the plan pads every run to a multiple of G; d = 41:
// LPE = 12, five rows per instruction).  All groups accumulate the SAME output row (wave-uniform
// -> index-mode fold); they are added once per task.  1.63 ms.  (Tried and dropped: 8-byte lanes
// at 4-byte alignment, two rows per instruction -- no faster than one row per instruction.)
// ---------------------------------------------------------------------------------------
// src_row (optional): row r of the copy is row src_row[r] of B -- the plan's internal column permutation (spmm.hip)
__global__ __launch_bounds__(256) void sweep_repack_kernel(const float *__restrict__ B, size_t ldb, uint32_t n,
                                                           uint32_t d, float *__restrict__ out, uint32_t dp,
                                                           const uint32_t *__restrict__ src_row) {
    const uint32_t q4 = dp / 4;
    const size_t total = (size_t)n * q4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const uint32_t r = (uint32_t)(i / q4), c = (uint32_t)(i % q4) * 4;
        const float *src = B + (size_t)(src_row ? src_row[r] : r) * ldb + c;
        float4 v;
        v.x = c + 0 < d ? src[0] : 0.f;
        v.y = c + 1 < d ? src[1] : 0.f;
        v.z = c + 2 < d ? src[2] : 0.f;
        v.w = c + 3 < d ? src[3] : 0.f;
        *reinterpret_cast<float4 *>(out + (size_t)r * dp + c) = v;
    }
}

// Entries are staged through LDS.  (With the entries in scalar registers, as in the kernels
// above, picking "my quarter's entry" out of four SGPRs costs ~16 vector ops per quad: 1.90 ms.)
// The wave copies its stream 128 entries at a time
// (one coalesced global_load_dwordx4, issued a whole chunk ahead, then one ds_write_b128 into a
// wave-private 1 KiB slot) and every lane fetches ITS entry of a quad with one broadcast
// ds_read_b64: no select, no scalar multiply; the run flag / row of the quad's first entry comes
// from lane 0 (v_readfirstlane).  ~5 vector ops per quad -> the kernel is left with the gather
// instructions themselves.
// LPE = lanes per entry (16 B each): 16 -> rows up to 64 floats, 4 entries per instruction;
// 12 -> up to 48 floats, 5 entries (the logits layer: 41 classes, 48 at P = 8); 8 -> 32 floats, 8
// entries; 4 -> 16 floats, 16 entries.  The plan pads runs to G = 64 / LPE entries.
template <int LPE> __device__ __forceinline__ float reduce_groups(float x) {
    if constexpr (LPE == 12) {                      // 5 groups at lanes 0,12,24,36,48
        const float far = __shfl_down(x, 48);       // group 4 -> lanes 0..11
        x += __shfl_down(x, 24);                    // groups 2,3 -> 0,1
        x += __shfl_down(x, 12);                    // group 1 -> 0
        return x + far;
    } else {
#pragma unroll
        for (int o = LPE; o < 64; o <<= 1) x += __shfl_xor(x, o);
        return x;
    }
}

template <int LPE>
__global__ __launch_bounds__(64 * kWavesPerBlock) __attribute__((amdgpu_num_vgpr(64))) void spmm_sweep_quad_lds_kernel(
    const SweepTask *__restrict__ tasks, uint32_t task0, uint32_t n_launch,
    const uint2 *__restrict__ entries, const uint32_t *__restrict__ task_rows,
    const float *__restrict__ B, uint32_t b_bytes, uint32_t row_bytes, float *__restrict__ C, size_t ldc,
    float *__restrict__ partial, uint32_t d, float alpha, float beta, uint32_t flags, float slope) {
    __shared__ uint4 ring[kWavesPerBlock][2][64];          // per wave: two slots of 128 entries
    const int lane = threadIdx.x & 63;
    const uint32_t wib = __builtin_amdgcn_readfirstlane((uint32_t)(threadIdx.x >> 6));
    const uint32_t local = blockIdx.x * kWavesPerBlock + wib;
    if (local >= n_launch) return;                    // no barriers: waves are independent
    const uint32_t t = task0 + local;
    const SweepTask task = tasks[t];
    const __amdgpu_buffer_rsrc_t rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float *>(B), 0, b_bytes, 0x00020000);
    constexpr int G = 64 / LPE;                       // entries per gather instruction
    constexpr int STEP = 4 * G;                       // entries per loop step (four gathers in flight)
    constexpr int CH = (128 / STEP) * STEP;           // entries consumed per 1 KiB LDS chunk
    const uint32_t sub = lane % LPE;
    const bool live = lane < G * LPE;                 // LPE = 12 leaves lanes 60..63 idle
    const uint32_t grp = live ? lane / LPE : G - 1;
    const uint32_t col = sub * 4;
    const bool active = live && col < d;
    const uint32_t lane_off = (active ? col : 0) * 4u;
    MGGCN_PLANES_ZERO();
    uint32_t cur_row = 0;
    f32x4_t acc = {0.f, 0.f, 0.f, 0.f};

    const uint32_t n_ent = task.end - task.beg;           // multiple of STEP
    if (n_ent) {
        // every chunk is loaded as 128 entries (1 KiB) starting at entry c * CH; CH of them are used
        const uint4 *stream = reinterpret_cast<const uint4 *>(entries + task.beg) + lane;   // 2 entries per uint4
        const uint32_t n_chunks = (n_ent + CH - 1) / CH;  // the plan leaves a chunk of slack after the last task
        ring[wib][0][lane] = stream[0];
        uint4 pre = stream[n_chunks > 1 ? CH / 2 : 0];
        const uint32_t prio_slot = blockIdx.x / ((flags >> kNumCuPos) & kNumCuMask);
        const uint32_t prio_shift = (flags >> kPrioShiftPos) & 15u;          // in chunks of CH (~128) entries here
        f32x4_t b[4];
#pragma unroll
        for (int u = 0; u < 4; u++) b[u] = (f32x4_t){0.f, 0.f, 0.f, 0.f};
        for (uint32_t c = 0; c < n_chunks; c++) {
            if ((flags & kFlagPrioRotate) && (c & ((1u << prio_shift) - 1u)) == 0) rotate_priority(prio_slot, c >> prio_shift);
            ring[wib][(c + 1) & 1][lane] = pre;           // chunk c+1 (slot last read during chunk c-1)
            pre = stream[(size_t)(CH / 2) * (c + 2 < n_chunks ? c + 2 : c)];
            __builtin_amdgcn_wave_barrier();
            const uint2 *slot = reinterpret_cast<const uint2 *>(&ring[wib][c & 1][0]) + grp;
            const uint32_t n_steps = min((uint32_t)(CH / STEP), (n_ent - c * CH) / STEP);
            for (uint32_t s = 0; s < n_steps; s++) {
                uint2 ent[4];
#pragma unroll
                for (int u = 0; u < 4; u++) ent[u] = slot[s * STEP + u * G];
                // (b is zeroed ONCE, in front of the chunk loop: lanes past the row never load and keep their zeros --
                //  zeroing it here was sixteen v_mov per step, four per gather)
                if (active) {                                    // lanes past the row issue no load (3 % faster)
#pragma unroll
                    for (int u = 0; u < 4; u++)
                        b[u] = mggcn_buffer_load_v4f32(rsrc, (int)(__umul24(ent[u].x & kColMask, row_bytes) + lane_off), 0, 0);
                }
#pragma unroll
                for (int u = 0; u < 4; u++) {
                    const uint32_t pk = __builtin_amdgcn_readfirstlane(ent[u].x);
                    if (pk & kRunFlag) {                                   // first group of a (panel,row) run
                        MGGCN_PLANES_FOLD(cur_row, acc);
                        acc = (f32x4_t){0.f, 0.f, 0.f, 0.f};
                        cur_row = (pk >> kColBits) & (kRW - 1);
                    }
                    const float v = __builtin_bit_cast(float, ent[u].y);
                    acc[0] = fmaf(v, b[u][0], acc[0]);
                    acc[1] = fmaf(v, b[u][1], acc[1]);
                    acc[2] = fmaf(v, b[u][2], acc[2]);
                    acc[3] = fmaf(v, b[u][3], acc[3]);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    MGGCN_PLANES_FOLD(cur_row, acc);

    // four rows at a time, three phases (reads of C, results, stores) -- see the pair kernel's epilogue
    auto emit4_any = [&](auto has_beta_c, uint32_t r0, float (&x)[4][4]) {
        constexpr bool HAS_BETA = decltype(has_beta_c)::value;
        if (r0 >= task.n_rows) return;                                   // wave-uniform
#pragma unroll
        for (int k = 0; k < 4; k++)
#pragma unroll
            for (int c = 0; c < 4; c++) x[k][c] = reduce_groups<LPE>(x[k][c]);
        const bool writer = active && grp == 0;
        // with reads of C: two rows per pass (the kernel has 64 VGPRs to itself; four rows' old values spilled at LPE = 12)
        constexpr int STEP = HAS_BETA ? 2 : 4;
#pragma unroll
        for (int k0 = 0; k0 < 4; k0 += STEP) {
            uint32_t dst[STEP];
            bool use[STEP], slot[STEP];
            float old[STEP][4];
#pragma unroll
            for (int q = 0; q < STEP; q++) {                                 // (1)
                use[q] = r0 + k0 + q < task.n_rows;
                dst[q] = use[q] ? task_rows[(size_t)t * kRW + r0 + k0 + q] : 0u;
                slot[q] = (dst[q] & kSlotFlag) != 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    old[q][c] = 0.f;
                    if constexpr (HAS_BETA) {
                        if (use[q] && !slot[q] && writer && col + c < d) old[q][c] = C[(size_t)dst[q] * ldc + col + c];
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < STEP; q++)                                   // (2), in place
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    float o = alpha * x[k0 + q][c];
                    if constexpr (HAS_BETA) o = fmaf(beta, old[q][c], o);
                    if (flags & MGGCN_SPMM_LEAKY_RELU) o = lrelu(o, slope);
                    x[k0 + q][c] = slot[q] ? x[k0 + q][c] : o;
                }
#pragma unroll
            for (int q = 0; q < STEP; q++) {                                 // (3)
                float *p = slot[q] ? partial + (size_t)(dst[q] & ~kSlotFlag) * d + col : C + (size_t)dst[q] * ldc + col;
#pragma unroll
                for (int c = 0; c < 4; c++)
                    if (use[q] && writer && col + c < d) p[c] = x[k0 + q][c];
            }
        }
    };
    auto emit4_beta = [&](uint32_t r0, float (&x)[4][4]) { emit4_any(std::true_type{}, r0, x); };
    auto emit4_nobeta = [&](uint32_t r0, float (&x)[4][4]) { emit4_any(std::false_type{}, r0, x); };
    if (beta != 0.f) { MGGCN_PLANES_EMIT_FOURS(emit4_beta) } else { MGGCN_PLANES_EMIT_FOURS(emit4_nobeta) }
}

__global__ __launch_bounds__(256) void sweep_combine_kernel(
    const SweepSplitRow *__restrict__ rows, uint32_t n_split, const float *__restrict__ partial,
    float *__restrict__ C, size_t ldc, uint32_t d, float alpha, float beta, uint32_t flags, float slope) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave =
        __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= n_split) return;
    const SweepSplitRow sr = rows[wave];
    for (uint32_t col = lane; col < d; col += 64) {
        float acc = 0.f;
        for (uint32_t s = 0; s < sr.n_slots; s++) acc += partial[(size_t)(sr.first_slot + s) * d + col];
        float o = alpha * acc;
        float *cp = C + (size_t)sr.row * ldc + col;
        if (beta != 0.f) o = fmaf(beta, *cp, o);
        if (flags & MGGCN_SPMM_LEAKY_RELU) o = lrelu(o, slope);
        *cp = o;
    }
}

}  // namespace

// the host side of a sweep plan (mggcn_plan::SweepHost: tasks, entry stream, row tables, knobs) + its device image
struct SweepPlan {
    uint32_t n_rows = 0, n_cols = 0, max_d = 0;
    uint32_t n_tasks = 0, round_tasks = 0, n_split_rows = 0, n_slots = 0;
    uint32_t run_pad = 2;          // every (panel,row) run is a multiple of this many entries
    uint32_t lpe = 0;              // narrow form: lanes per entry of the gather kernel (run_pad = 64 / lpe); 0 = wide form
    SweepTask *d_tasks = nullptr;
    uint2 *d_entries = nullptr;
    uint32_t *d_task_rows = nullptr;
    SweepSplitRow *d_split = nullptr;
    float *d_partial = nullptr;
    unsigned long long *d_stamps = nullptr;   // diagnostics: 3 words per task (MGGCN_SPMM_STAMPS=1)
    size_t bytes = 0;
    // tuning knobs, read from the environment ONCE when the plan is built (never on the launch path)
    uint32_t panel_rows = 0, num_cu = 0;
    uint64_t n_entries = 0;        // padded entry stream length
    uint32_t prio_bits_wide = 0, prio_bits_narrow = 0;   // kFlagPrioRotate | shift << kPrioShiftPos | CUs << kNumCuPos
    uint32_t tasks_per_wave = 1;
    bool allow_quad = true, allow_vec4 = true, fast_pairs = true;
};

// compute units of the current device (a launch round is sized to the resident set; a partitioned device has fewer)
static uint32_t device_compute_units() {
    int dev = 0, cus = 0;
    MGGCN_CHECK_HIP(hipGetDevice(&dev));
    MGGCN_CHECK_HIP(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev));
    return (uint32_t)std::max(cus, 1);
}

SweepPlan *sweep_plan_build(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr,
                            const uint32_t *indices, const float *values, uint32_t max_d, bool force,
                            uint32_t d_hint, bool hot_columns) {
    static_assert(sizeof(Entry) == sizeof(uint2), "the entry stream is uploaded as is");
    SweepHost h;
    if (!sweep_build_host(n_rows, n_cols, indptr, indices, values, max_d, force, d_hint, hot_columns, device_compute_units(), h))
        return nullptr;
    auto *p = new SweepPlan;
    p->n_rows = h.n_rows; p->n_cols = h.n_cols; p->max_d = h.max_d;
    p->n_tasks = h.n_tasks; p->round_tasks = h.round_tasks; p->run_pad = h.run_pad; p->lpe = h.lpe;
    p->n_split_rows = (uint32_t)h.split_rows.size(); p->n_slots = h.n_slots;
    p->panel_rows = h.panel_rows; p->n_entries = h.n_entries; p->num_cu = h.num_cu;
    p->prio_bits_wide = h.prio_bits_wide; p->prio_bits_narrow = h.prio_bits_narrow; p->tasks_per_wave = h.tasks_per_wave;
    p->allow_quad = h.allow_quad; p->allow_vec4 = h.allow_vec4; p->fast_pairs = h.fast_pairs;
    const size_t tb = h.tasks.size() * sizeof(SweepTask), eb = h.entries.size() * sizeof(uint2);
    const size_t rb = h.task_rows.size() * sizeof(uint32_t), sb = h.split_rows.size() * sizeof(SweepSplitRow);
    const size_t pb = (size_t)h.n_slots * max_d * sizeof(float);
    MGGCN_CHECK_HIP(hipMalloc(&p->d_tasks, tb));
    MGGCN_CHECK_HIP(hipMemcpy(p->d_tasks, h.tasks.data(), tb, hipMemcpyHostToDevice));
    MGGCN_CHECK_HIP(hipMalloc(&p->d_entries, eb + 2048));          // + a chunk: the LDS-staged kernel reads whole 1 KiB chunks
    MGGCN_CHECK_HIP(hipMemset(reinterpret_cast<char *>(p->d_entries) + eb, 0, 2048));
    if (eb) MGGCN_CHECK_HIP(hipMemcpy(p->d_entries, h.entries.data(), eb, hipMemcpyHostToDevice));
    MGGCN_CHECK_HIP(hipMalloc(&p->d_task_rows, rb));
    MGGCN_CHECK_HIP(hipMemcpy(p->d_task_rows, h.task_rows.data(), rb, hipMemcpyHostToDevice));
    if (sb) {
        MGGCN_CHECK_HIP(hipMalloc(&p->d_split, sb));
        MGGCN_CHECK_HIP(hipMemcpy(p->d_split, h.split_rows.data(), sb, hipMemcpyHostToDevice));
    }
    if (pb) MGGCN_CHECK_HIP(hipMalloc(&p->d_partial, pb));
    if (env_u32("MGGCN_SPMM_STAMPS", 0u)) {
        MGGCN_CHECK_HIP(hipMalloc(&p->d_stamps, (size_t)h.n_tasks * 3 * sizeof(unsigned long long)));
        MGGCN_CHECK_HIP(hipMemset(p->d_stamps, 0, (size_t)h.n_tasks * 3 * sizeof(unsigned long long)));
    }
    p->bytes = tb + eb + rb + sb + pb;
    return p;
}

void sweep_plan_destroy(SweepPlan *p) {
    if (!p) return;
    if (p->d_tasks) MGGCN_CHECK_HIP(hipFree(p->d_tasks));
    if (p->d_entries) MGGCN_CHECK_HIP(hipFree(p->d_entries));
    if (p->d_task_rows) MGGCN_CHECK_HIP(hipFree(p->d_task_rows));
    if (p->d_split) MGGCN_CHECK_HIP(hipFree(p->d_split));
    if (p->d_partial) MGGCN_CHECK_HIP(hipFree(p->d_partial));
    if (p->d_stamps) MGGCN_CHECK_HIP(hipFree(p->d_stamps));
    delete p;
}

size_t sweep_plan_bytes(const SweepPlan *p) { return p ? p->bytes : 0; }

// one line per slice plan for MGGCN_SPMM_PLAN_LOG / mggcn_spmm_plan_describe
int sweep_plan_describe(const SweepPlan *p, char *out, size_t cap) {
    if (!p) return 0;
    return std::snprintf(out, cap, "tasks=%u rounds=%u panel_rows=%u run_pad=%u lpe=%u entries=%llu split_rows=%u slots=%u",
                         p->n_tasks, (p->n_tasks + p->round_tasks - 1) / p->round_tasks, p->panel_rows, p->run_pad, p->lpe,
                         (unsigned long long)p->n_entries, p->n_split_rows, p->n_slots);
}
uint32_t sweep_plan_tasks(const SweepPlan *p) { return p ? p->n_tasks : 0; }
uint32_t sweep_plan_split_rows(const SweepPlan *p) { return p ? p->n_split_rows : 0; }

uint32_t sweep_plan_read_stamps(const SweepPlan *p, unsigned long long *host_out, uint32_t capacity_tasks) {
    if (!p || !p->d_stamps) return 0;
    const uint32_t n = std::min(p->n_tasks, capacity_tasks);
    MGGCN_CHECK_HIP(hipDeviceSynchronize());
    MGGCN_CHECK_HIP(hipMemcpy(host_out, p->d_stamps, (size_t)n * 3 * sizeof(unsigned long long), hipMemcpyDeviceToHost));
    return n;
}

uint32_t sweep_plan_launches(const SweepPlan *p, uint32_t d) {
    if (!p) return 0;
    const bool quad = p->lpe && d <= 4 * p->lpe;
    const bool vec4 = p->run_pad % 2 == 0 && d >= 96 && d % 4 == 0;
    const bool vec2 = d > 64 && d % 2 == 0;
    const uint32_t tpw = (vec4 && !quad) ? p->tasks_per_wave : 1u;
    const uint32_t per_launch = (vec2 || vec4 || quad) ? p->round_tasks * tpw : std::max(p->round_tasks, p->num_cu * 6u * kWavesPerBlock);
    return (p->n_tasks + per_launch - 1) / per_launch + (p->n_split_rows ? 1u : 0u);
}

bool sweep_supports(const SweepPlan *p, uint32_t d, size_t ldb, size_t ldc, const void *B, const void *C) {
    if (!p) return false;
    if (p->n_slots && d > p->max_d) return false;
    // the row gathers address B through a 32-bit buffer descriptor
    if ((uint64_t)p->n_cols * ldb * sizeof(float) > 0xFFFFFFFFull) return false;
    (void)ldc; (void)B; (void)C;
    return true;
}

bool sweep_wants_repack(const SweepPlan *p, uint32_t d, size_t ldb, const void *B) {
    if (!p || !p->lpe || d > 4 * p->lpe || !p->allow_quad) return false;
    // The gather path is priced per 128-byte line touched (~2.7 cycles each, narrow_spmm.py: 16 rows of
    // 64 B per instruction cost 62 cycles, 4 rows of 176 B 26): a 176-byte pitch puts 37 % of the rows
    // on three lines, a pitch that is a multiple of 64 B never more than two.
    return !(ldb % 16 == 0 && (reinterpret_cast<uintptr_t>(B) & 63u) == 0);
}

void sweep_repack(hipStream_t st, const float *B, size_t ldb, uint32_t n_cols, uint32_t d, float *out, uint32_t dp,
                  const uint32_t *src_row) {
    const size_t total = (size_t)n_cols * (dp / 4);
    const unsigned grid = (unsigned)std::min<size_t>((total + 255) / 256, (size_t)kNumCU * 16);
    hipLaunchKernelGGL(sweep_repack_kernel, dim3(std::max(grid, 1u)), dim3(256), 0, st, B, ldb, n_cols, d, out, dp, src_row);
    MGGCN_CHECK_LAUNCH();
}

void sweep_launch(hipStream_t st, const SweepPlan *p, const float *B, size_t ldb, float *C, size_t ldc,
                  uint32_t d, float alpha, float beta, uint32_t flags, float slope) {
    // narrow rows on a quad-padded stream: B must be 16-byte pitched (the caller re-pitches it
    // with sweep_repack when sweep_wants_repack says so)
    const bool quad = p->lpe && d <= 4 * p->lpe && ldb % 4 == 0 && aligned16(B) && p->allow_quad;
    // float4 pair form: 16-byte aligned rows of >= 96 columns (narrower rows would idle most of
    // a half-wave); float2 lanes need 8-byte aligned rows; otherwise one column per lane
    const bool vec4 = p->run_pad % 2 == 0 && d >= 96 && d % 4 == 0 && ldb % 4 == 0 && ldc % 4 == 0 && aligned16(B) && aligned16(C) &&
                      p->allow_vec4;
    const bool vec2 = d > 64 && d % 2 == 0 && ldb % 2 == 0 && ldc % 2 == 0 &&
                      (reinterpret_cast<uintptr_t>(B) & 7u) == 0 && (reinterpret_cast<uintptr_t>(C) & 7u) == 0;
    const uint32_t b_bytes = (uint32_t)((uint64_t)p->n_cols * ldb * sizeof(float));
    const uint32_t row_bytes = (uint32_t)(ldb * sizeof(float));
    // One launch = one round of RESIDENT tasks for the float4 kernels (128 VGPRs: four blocks of four
    // waves per CU; six blocks per launch left a third of the waves queued behind the others and out
    // of step: d = 41 1.64 / 1.71 ms against 1.52 / 1.55).  The one-column-per-lane kernels (56 VGPRs)
    // are instruction-bound and take six blocks per CU (d = 41: 2.4 -> 2.0 ms).
    const uint32_t tpw = (vec4 && !quad) ? p->tasks_per_wave : 1u;
    const uint32_t per_launch = (vec2 || vec4 || quad) ? p->round_tasks * tpw : std::max(p->round_tasks, p->num_cu * 6u * kWavesPerBlock);
    // the knobs (priority rotation, tasks per wave, kernel gates) were read once, when the plan was built
    const uint32_t wide_flags = (flags & 0xFFu) | p->prio_bits_wide;
    const uint32_t narrow_flags = (flags & 0xFFu) | p->prio_bits_narrow;
    for (uint32_t t0 = 0; t0 < p->n_tasks; t0 += per_launch) {
        const uint32_t n_launch = std::min(per_launch, p->n_tasks - t0);
        const uint32_t n_waves = tpw > 1 ? std::min(n_launch, p->round_tasks) : n_launch;
        const dim3 grid((n_waves + kWavesPerBlock - 1) / kWavesPerBlock), block(64 * kWavesPerBlock);
#define MGGCN_LAUNCH_NARROW(L)                                                                             \
    hipLaunchKernelGGL((spmm_sweep_quad_lds_kernel<L>), grid, block, 0, st, p->d_tasks, t0, n_launch, p->d_entries, \
                       p->d_task_rows, B, b_bytes, row_bytes, C, ldc, p->d_partial, d, alpha, beta, narrow_flags, slope)
        if (quad) {
            if (p->lpe == 4) MGGCN_LAUNCH_NARROW(4);
            else if (p->lpe == 8) MGGCN_LAUNCH_NARROW(8);
            else if (p->lpe == 12) MGGCN_LAUNCH_NARROW(12);
            else MGGCN_LAUNCH_NARROW(16);
        }
#undef MGGCN_LAUNCH_NARROW
        else if (vec4 && row_bytes >= 512u && (row_bytes & (row_bytes - 1u)) == 0 && p->fast_pairs)   // power-of-two pitch
            hipLaunchKernelGGL((spmm_sweep_pair_kernel<true>), grid, block, 0, st, p->d_tasks, t0, n_launch,
                               p->d_entries, p->d_task_rows, B, b_bytes, row_bytes, C, ldc, p->d_partial, d,
                               alpha, beta, wide_flags, slope, p->d_stamps, tpw > 1 ? p->round_tasks : n_launch);
        else if (vec4)
            hipLaunchKernelGGL((spmm_sweep_pair_kernel<false>), grid, block, 0, st, p->d_tasks, t0, n_launch,
                               p->d_entries, p->d_task_rows, B, b_bytes, row_bytes, C, ldc, p->d_partial, d,
                               alpha, beta, wide_flags, slope, p->d_stamps, tpw > 1 ? p->round_tasks : n_launch);
        else if (vec2)
            hipLaunchKernelGGL((spmm_sweep_kernel<2>), grid, block, 0, st, p->d_tasks, t0, n_launch,
                               p->d_entries, p->d_task_rows, B, b_bytes, row_bytes, C, ldc, p->d_partial, d,
                               alpha, beta, flags, slope);
        else
            hipLaunchKernelGGL((spmm_sweep_kernel<1>), grid, block, 0, st, p->d_tasks, t0, n_launch,
                               p->d_entries, p->d_task_rows, B, b_bytes, row_bytes, C, ldc, p->d_partial, d,
                               alpha, beta, flags, slope);
        MGGCN_CHECK_LAUNCH();
    }
    if (p->n_split_rows) {
        hipLaunchKernelGGL(sweep_combine_kernel, dim3((p->n_split_rows + 3) / 4), dim3(256), 0, st, p->d_split,
                           p->n_split_rows, p->d_partial, C, ldc, d, alpha, beta, flags, slope);
        MGGCN_CHECK_LAUNCH();
    }
}
