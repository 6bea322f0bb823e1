// spmm.hip -- CSR SpMM  C = alpha*A*B + beta*C  for gfx950 (MI355X), fp32.
//
// Stands in for cusparseSpMM as the reference calls it (src/cuda_utils.hpp:15-32:
// op N/N, CSR u32/u32/f32, row-major dense operands) and for its workspace query
// (src/cuda_utils.hpp:94-102).  There is no reference kernel source: the
// arithmetic lived inside cuSPARSE.  Written from the math.
//
// Shape of the work (Reddit): 233 k rows, 115 M non-zeros, mean
// degree 493, maximum ~21 k, 128-wide fp32 feature rows (512 B).  Per SpMM the
// non-zero stream is 0.92 GB and B is 119 MB -- B sits in the 256 MiB Infinity
// Cache, so the kernel is a *row gather* at L2 / Infinity-Cache rate (59.8 GB of
// 512-B row fetches), not an HBM stream.  Design consequences:
//
//  * one wave64 per work item (a row, or a <=kSplit-long slice of a heavy row);
//    a 128-wide row is covered by 32 lanes x float4, so one wave-instruction
//    (global_load_dwordx4, 1 KiB) fetches TWO neighbour rows, fully coalesced;
//    8 such loads (16 rows, 8 KiB) are in flight per wave before the first FMA.
//  * column indices / values are fetched 64 at a time with one coalesced load per
//    wave and handed to the row loads through the LDS crossbar (ds_bpermute).
//  * heavy-tailed degrees: at ~0.5 us per non-zero per wave a 21 k-row would run
//    for ~10 ms on its own, longer than the rest of the kernel.  The plan cuts such
//    rows into slices, sorts all items longest-first (LPT) and sums the slices'
//    partial rows in a fixed order in a second, tiny kernel -> bitwise
//    reproducible, no float atomics.
//  * fp32 FMA accumulation in registers; alpha/beta/leaky-ReLU fused in the
//    epilogue; beta == 0 never reads C.
#include <algorithm>
#include <functional>
#include <numeric>
#include <string>
#include <thread>
#include <vector>

#include "common.h"
#include <chrono>
#include <cstdio>

#include "plan_host.h"
#include "spmm_internal.h"

// work items / split rows of the row-split form and the host passes of the plan builder: plan_host.h (free of HIP)
using mggcn_plan::env_u32;
using mggcn_plan::kNoSlot;
using mggcn_plan::SplitRow;
using mggcn_plan::SpmmItem;

namespace {

__device__ __forceinline__ float lrelu(float x, float slope) {
    const float y = slope * x;
    return x > y ? x : y;  // max(x, slope*x), reference src/cuda_utils.cu:26-31
}

__device__ __forceinline__ float4 fma4(float s, float4 b, float4 a) {
    a.x = fmaf(s, b.x, a.x);
    a.y = fmaf(s, b.y, a.y);
    a.z = fmaf(s, b.z, a.z);
    a.w = fmaf(s, b.w, a.w);
    return a;
}

// ---------------------------------------------------------------------------
// Vector path: d % 4 == 0, 16-byte aligned rows.  LPR lanes cover one feature row
// with float4 each (LPR*4 columns per pass); G = 64/LPR neighbour rows are
// fetched by one wave-instruction.  Wider d loops over column tiles.
// ---------------------------------------------------------------------------
template <int LPR, int UNROLL, bool HAS_ITEMS>
__global__ __launch_bounds__(256) void spmm_vec4_kernel(
    const SpmmItem *__restrict__ items, uint32_t n_items, const uint32_t *__restrict__ indptr,
    const uint32_t *__restrict__ indices, const float *__restrict__ values,
    const float *__restrict__ B, size_t ldb, float *__restrict__ C, size_t ldc,
    float *__restrict__ partial, uint32_t d, float alpha, float beta, uint32_t flags, float slope) {
    constexpr int G = 64 / LPR;
    const int lane = threadIdx.x & 63;
    const uint32_t wave =
        __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= n_items) return;

    uint32_t row, beg, end, slot;
    if (HAS_ITEMS) {
        const SpmmItem it = items[wave];
        row = it.row; beg = it.beg; end = it.end; slot = it.slot;
    } else {
        row = wave; beg = indptr[wave]; end = indptr[wave + 1]; slot = kNoSlot;
    }
    const int sub = lane % LPR;
    const int grp = lane / LPR;

    for (uint32_t col0 = 0; col0 < d; col0 += LPR * 4) {
        const uint32_t col = col0 + sub * 4;
        const bool active = col < d;
        const float *__restrict__ Bc = B + col;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);

        // software pipeline over 64-entry chunks of the row: the next chunk's
        // (index, value) pair is in flight while the current chunk's rows are gathered
        uint32_t nxt_c = 0;
        float nxt_v = 0.f;
        if (beg + lane < end) { nxt_c = indices[beg + lane]; nxt_v = values[beg + lane]; }
        for (uint32_t base = beg; base < end; base += 64) {
            const uint32_t my_c = nxt_c;
            const float my_v = nxt_v;
            const uint32_t e_next = base + 64 + lane;
            nxt_c = 0; nxt_v = 0.f;
            if (e_next < end) { nxt_c = indices[e_next]; nxt_v = values[e_next]; }
            const uint32_t cnt = min(64u, end - base);
            for (uint32_t j = 0; j < cnt; j += G * UNROLL) {
                float4 b[UNROLL];
                float v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    const uint32_t src = j + u * G + grp;   // < 64 + G*UNROLL
                    const uint32_t c = __shfl(my_c, src & 63);
                    const float vv = __shfl(my_v, src & 63);
                    const bool ok = active && src < cnt;
                    v[u] = ok ? vv : 0.f;
                    b[u] = ok ? *reinterpret_cast<const float4 *>(Bc + (size_t)c * ldb)
                              : make_float4(0.f, 0.f, 0.f, 0.f);
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) acc = fma4(v[u], b[u], acc);
            }
        }
        // fold the G lane groups (fixed order -> reproducible)
#pragma unroll
        for (int off = LPR; off < 64; off <<= 1) {
            acc.x += __shfl_xor(acc.x, off);
            acc.y += __shfl_xor(acc.y, off);
            acc.z += __shfl_xor(acc.z, off);
            acc.w += __shfl_xor(acc.w, off);
        }
        if (grp == 0 && active) {
            if (slot == kNoSlot) {
                float4 o = make_float4(alpha * acc.x, alpha * acc.y, alpha * acc.z, alpha * acc.w);
                float4 *cp = reinterpret_cast<float4 *>(C + (size_t)row * ldc + col);
                if (beta != 0.f) {
                    const float4 c0 = *cp;
                    o.x = fmaf(beta, c0.x, o.x); o.y = fmaf(beta, c0.y, o.y);
                    o.z = fmaf(beta, c0.z, o.z); o.w = fmaf(beta, c0.w, o.w);
                }
                if (flags & MGGCN_SPMM_LEAKY_RELU) {
                    o.x = lrelu(o.x, slope); o.y = lrelu(o.y, slope);
                    o.z = lrelu(o.z, slope); o.w = lrelu(o.w, slope);
                }
                *cp = o;
            } else {
                *reinterpret_cast<float4 *>(partial + (size_t)slot * d + col) = acc;
            }
        }
    }
}

// ---------------------------------------------------------------------------
// Scalar path: any d / any alignment (the reference's d = 41 logits layer has
// 164-byte rows).  One lane per column, one neighbour row per wave-instruction;
// index/value broadcast through v_readlane (wave-uniform -> scalar registers).
// ---------------------------------------------------------------------------
template <int UNROLL, bool HAS_ITEMS>
__global__ __launch_bounds__(256) void spmm_scalar_kernel(
    const SpmmItem *__restrict__ items, uint32_t n_items, const uint32_t *__restrict__ indptr,
    const uint32_t *__restrict__ indices, const float *__restrict__ values,
    const float *__restrict__ B, size_t ldb, float *__restrict__ C, size_t ldc,
    float *__restrict__ partial, uint32_t d, float alpha, float beta, uint32_t flags, float slope) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave =
        __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= n_items) return;
    uint32_t row, beg, end, slot;
    if (HAS_ITEMS) {
        const SpmmItem it = items[wave];
        row = it.row; beg = it.beg; end = it.end; slot = it.slot;
    } else {
        row = wave; beg = indptr[wave]; end = indptr[wave + 1]; slot = kNoSlot;
    }
    for (uint32_t col0 = 0; col0 < d; col0 += 64) {
        const uint32_t col = col0 + lane;
        const bool active = col < d;
        const float *__restrict__ Bc = B + (active ? col : 0);
        float acc = 0.f;
        uint32_t nxt_c = 0;
        float nxt_v = 0.f;
        if (beg + lane < end) { nxt_c = indices[beg + lane]; nxt_v = values[beg + lane]; }
        for (uint32_t base = beg; base < end; base += 64) {
            const uint32_t my_c = nxt_c;
            const float my_v = nxt_v;
            const uint32_t e_next = base + 64 + lane;
            nxt_c = 0; nxt_v = 0.f;
            if (e_next < end) { nxt_c = indices[e_next]; nxt_v = values[e_next]; }
            const uint32_t cnt = min(64u, end - base);
            for (uint32_t j = 0; j < cnt; j += UNROLL) {
                float b[UNROLL], v[UNROLL];
#pragma unroll
                for (int u = 0; u < UNROLL; u++) {
                    const uint32_t src = (j + u) & 63;     // wave-uniform
                    const uint32_t c = __builtin_amdgcn_readlane(my_c, src);
                    const float vv = __builtin_bit_cast(
                        float, __builtin_amdgcn_readlane(__builtin_bit_cast(uint32_t, my_v), src));
                    const bool ok = (j + u) < cnt;            // padded lanes carry v = 0, c = 0
                    v[u] = ok ? vv : 0.f;
                    b[u] = (ok && active) ? Bc[(size_t)c * ldb] : 0.f;
                }
#pragma unroll
                for (int u = 0; u < UNROLL; u++) acc = fmaf(v[u], b[u], acc);
            }
        }
        if (active) {
            if (slot == kNoSlot) {
                float o = alpha * acc;
                float *cp = C + (size_t)row * ldc + col;
                if (beta != 0.f) o = fmaf(beta, *cp, o);
                if (flags & MGGCN_SPMM_LEAKY_RELU) o = lrelu(o, slope);
                *cp = o;
            } else {
                partial[(size_t)slot * d + col] = acc;
            }
        }
    }
}

// Sum the slices of every split row in slot order, then the common epilogue.
__global__ __launch_bounds__(256) void spmm_combine_kernel(
    const SplitRow *__restrict__ rows, uint32_t n_split, const float *__restrict__ partial,
    float *__restrict__ C, size_t ldc, uint32_t d, float alpha, float beta, uint32_t flags,
    float slope) {
    const int lane = threadIdx.x & 63;
    const uint32_t wave =
        __builtin_amdgcn_readfirstlane((uint32_t)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));
    if (wave >= n_split) return;
    const SplitRow sr = rows[wave];
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

struct mggcn_spmm_plan {
    uint32_t n_rows = 0, n_cols = 0, max_d = 0;
    uint32_t n_items = 0, n_split_rows = 0, n_slots = 0;
    SpmmItem *d_items = nullptr;
    SplitRow *d_split = nullptr;
    float *d_partial = nullptr;
    size_t bytes = 0;
    int device = 0;
    // column-panel sweep form (large matrices; spmm_sweep.hip): one plan per column SLICE of
    // the matrix, run back to back with beta accumulation (see mggcn_spmm_plan_create)
    std::vector<SweepPlan *> sweeps;
    // narrow form (d_hint <= 64): scratch for B re-pitched to 64-byte-multiple rows, n_cols x bpad_dp floats
    float *d_bpad = nullptr;
    uint32_t bpad_dp = 0;
    // internal column permutation (vertex orders with locality, see mggcn_spmm_plan_create_for): row j of the permuted
    // copy of B is row d_src_row[j] of the caller's B; d_bperm = scratch for that copy in the wide form (the narrow form
    // folds the permutation into its re-pitch pass)
    uint32_t *d_src_row = nullptr;
    float *d_bperm = nullptr;
    // what the plan builder decided (mggcn_spmm_plan_describe)
    uint32_t d_hint = 0;
    int hot_columns = -1;          // -1: not measured (no sweep form considered)
    double hot_share = 0.0;        // share of the non-zeros in the 1 % most popular columns
    double locality = 0.0;         // share of the non-zeros whose column lies in the same 1/32 of the index space as their row
    double mean_run = 0.0;         // mean (panel,row) run length the density gate saw
    double build_s = 0.0;          // host seconds spent in plan_create (sort + upload)
    uint64_t nnz = 0;
};

MGGCN_API mggcn_spmm_plan *mggcn_spmm_plan_create(uint32_t n_rows, uint32_t n_cols,
                                                  const uint32_t *host_indptr,
                                                  const uint32_t *host_indices,
                                                  const float *host_values, uint32_t max_d) {
    return mggcn_spmm_plan_create_for(n_rows, n_cols, host_indptr, host_indices, host_values, max_d, 0);
}

MGGCN_API mggcn_spmm_plan *mggcn_spmm_plan_create_for(uint32_t n_rows, uint32_t n_cols,
                                                      const uint32_t *host_indptr,
                                                      const uint32_t *host_indices,
                                                      const float *host_values, uint32_t max_d,
                                                      uint32_t d_hint) {
    MGGCN_REQUIRE(host_indptr != nullptr || n_rows == 0, "plan needs the host copy of indptr");
    const bool narrow = d_hint >= 1 && d_hint <= 64 && n_cols <= (1u << 24);
    if (!narrow) d_hint = 0;
    MGGCN_REQUIRE(max_d > 0, "max_d must be positive");
    const auto t_build0 = std::chrono::steady_clock::now();
    // one range check for every pass below (they index per-column tables with what they read)
    MGGCN_REQUIRE(!host_indices || mggcn_plan::columns_in_range(n_rows, n_cols, host_indptr, host_indices), "column index out of range");
    // slice length for heavy rows; rows up to 1.5x the slice stay whole
    const uint32_t split = std::max<uint32_t>(64u, env_u32("MGGCN_SPMM_SPLIT", 512u));
    const mggcn_plan::RowSplitHost rs = mggcn_plan::rowsplit_build(n_rows, host_indptr, split);
    const auto &items = rs.items;
    const auto &split_rows = rs.split_rows;
    const uint32_t n_slots = rs.n_slots;

    auto *plan = new mggcn_spmm_plan;
    plan->n_rows = n_rows; plan->n_cols = n_cols; plan->max_d = max_d;
    plan->n_items = (uint32_t)items.size();
    plan->n_split_rows = (uint32_t)split_rows.size();
    plan->n_slots = n_slots;
    MGGCN_CHECK_HIP(hipGetDevice(&plan->device));
    const size_t ib = items.size() * sizeof(SpmmItem), sb = split_rows.size() * sizeof(SplitRow);
    const size_t pb = (size_t)n_slots * max_d * sizeof(float);
    if (ib) {
        MGGCN_CHECK_HIP(hipMalloc(&plan->d_items, ib));
        MGGCN_CHECK_HIP(hipMemcpy(plan->d_items, items.data(), ib, hipMemcpyHostToDevice));
    }
    if (sb) {
        MGGCN_CHECK_HIP(hipMalloc(&plan->d_split, sb));
        MGGCN_CHECK_HIP(hipMemcpy(plan->d_split, split_rows.data(), sb, hipMemcpyHostToDevice));
    }
    if (pb) MGGCN_CHECK_HIP(hipMalloc(&plan->d_partial, pb));
    plan->bytes = ib + sb + pb;
    const char *algo = std::getenv("MGGCN_SPMM_ALGO");
    if (host_indices && host_values && !(algo && std::string(algo) == "rowsplit"))
    {
        // Column slices: the sweep keeps the active part of B in L2 only while the waves stay
        // within a few MiB of each other; their drift grows with the length of the sweep
        // (measured: a 32 MiB extent runs at the in-L2 rate, the full 114 MiB Reddit extent 25 %
        // slower; with the six-instruction run fold 64 MiB slices edge out 32: 2.69 vs 2.73 ms; with the priority
        // rotation of spmm_sweep.hip 64 MiB is the optimum for the backward matrix too: 2.52 -> 2.40 ms).
        // So B is cut into slices of <= MGGCN_SPMM_SLICE_MIB (at 512-byte rows) and
        // C = beta C + alpha sum_s A[:, slice s] B is evaluated slice after slice: every launch
        // boundary re-synchronises the chip.  Costs one extra read+write of C per extra slice.
        const uint32_t hint_bytes = narrow ? (d_hint + 3) / 4 * 16 : 512u;
        // Column popularity.  On the Reddit shape the forward matrix (even rows, power-law column
        // popularity) keeps its hot rows of B in L1/L2 whatever the window and runs best with 6144-row
        // panels and 64 MiB slices (2.69 ms; 2.74 with 4096 / 32); the backward matrix (power-law rows,
        // uniformly popular columns) needs the tighter window: 2.86 ms with 4096 / 32 against 3.19
        // (profiles/experiments/sweep_vs_rowsplit.py, EXP_MATRIX=A).  Measure: share of the non-zeros
        // that sit in the 1 % most popular columns -- and, in the same pass, how much of the matrix sits near its
        // diagonal: a vertex order with locality (an unpermuted community graph), see the column permutation below.
        const mggcn_plan::ColumnStats cs = mggcn_plan::column_stats(n_rows, n_cols, host_indptr, host_indices);
        bool hot_columns = cs.hot_columns;
        plan->locality = cs.locality;
        plan->hot_share = cs.hot_share;
        if (const char *hc = std::getenv("MGGCN_SPMM_HOT_COLUMNS")) hot_columns = std::atoi(hc) != 0;
        const uint64_t slice_rows = std::max<uint64_t>(
            64, env_u32("MGGCN_SPMM_SLICE_ROWS", (uint32_t)(((uint64_t)env_u32("MGGCN_SPMM_SLICE_MIB", 64u) << 20) / hint_bytes)));   // tests set ROWS
        uint32_t S = (uint32_t)std::max<uint64_t>(1, ((uint64_t)n_cols + slice_rows - 1) / slice_rows);
        const uint64_t total_nnz = n_rows ? (uint64_t)host_indptr[n_rows] - host_indptr[0] : 0;
        // The sweep pays only for DENSE rows: its unit of work is a (panel,row) run, and a graph
        // whose rows have fewer non-zeros than there are panels degenerates into one fold per
        // entry.  Measured at the ogbn-products shape (n = 2.45 M, mean degree 51, B = 1.25 GB):
        // row-split 8.2 ms, sweep 15-38 ms (profiles/experiments/products_like.py).  Gate: mean
        // run length >= 2, and never more slices than leave ~64 non-zeros per (row, slice).
        const double avg_deg = n_rows ? (double)total_nnz / n_rows : 0.0;
        const double panel_rows = mggcn_plan::sweep_panel_rows(d_hint, hot_columns);
        const double mean_run = n_cols ? avg_deg * std::min<double>(panel_rows, n_cols) / n_cols : 0.0;
        if (!std::getenv("MGGCN_SPMM_SLICE_ROWS")) S = std::max<uint32_t>(1u, std::min<uint32_t>(S, (uint32_t)(avg_deg / 64.0)));
        plan->hot_columns = hot_columns ? 1 : 0;
        plan->mean_run = mean_run;
        const bool worth_it = total_nnz >= env_u32("MGGCN_SPMM_SWEEP_MIN_NNZ", 1u << 20) &&   // small graphs: row-split is fine
                              mean_run * 10.0 >= env_u32("MGGCN_SPMM_SWEEP_MIN_RUN_X10", 20u);
        // Vertex orders with locality.  The sweep wants every wave of the chip to cross the column panels together, with
        // about the same work in each -- true when a row's columns are spread over the index space (the randomly permuted
        // datasets the reference trains on, test/data/prep.py:87-94).  On an UNPERMUTED community graph a row's entries
        // sit in one panel, every 16-row task is sixteen spikes at sixteen different panels, the waves are never in the
        // same place, and the L2 holds nothing: 4.29 ms for 87 M non-zeros against 2.35 for 115 M on the permuted graph
        // (profiles/experiments/community_r03_*.log).  Detected (>= 8 % of the non-zeros within 1/32 of the diagonal; a
        // permuted graph has 3 %, the same graph numbered by decreasing degree 11 %, 64 contiguous communities 85 %), the plan relabels the COLUMNS by a fixed pseudo-random permutation pi: the entry streams
        // are built on pi(column), and every call first copies B into the plan's scratch in permuted row order (one
        // streaming pass; the narrow form folds it into its re-pitch pass).  MGGCN_SPMM_PERMUTE_COLUMNS = 0 / 1 overrides.
        std::vector<uint32_t> permuted_indices, src_row;
        const uint32_t permute_knob = env_u32("MGGCN_SPMM_PERMUTE_COLUMNS", 2u);
        const bool permute = worth_it && (permute_knob == 1u || (permute_knob == 2u && plan->locality >= 0.08)) && n_cols > 1;
        if (permute) {
            std::vector<uint32_t> pi;
            mggcn_plan::column_permutation(n_cols, pi, src_row);
            mggcn_plan::permute_indices(n_rows, host_indptr, host_indices, pi, permuted_indices);   // indexed like host_indices
            host_indices = permuted_indices.data();
        }
        if (!worth_it) {
            // nothing
        } else if (S <= 1) {
            if (SweepPlan *sp = sweep_plan_build(n_rows, n_cols, host_indptr, host_indices, host_values, max_d, true, d_hint, hot_columns))
                plan->sweeps.push_back(sp);
        } else {
            // bucket the non-zeros by column slice
            const uint32_t width = (n_cols + S - 1) / S;
            mggcn_plan::SliceBuckets sl = mggcn_plan::slice_buckets(n_rows, S, width, host_indptr, host_indices, host_values);
            bool ok = true;
            for (uint32_t k = 0; k < S && ok; k++) {
                if (sl.ixs[k].empty()) continue;            // e.g. a rank's own column range in its "remote" matrix
                SweepPlan *sp = sweep_plan_build(n_rows, n_cols, sl.ips[k].data(), sl.ixs[k].data(), sl.vvs[k].data(), max_d, true, d_hint, hot_columns);
                if (sp) plan->sweeps.push_back(sp); else ok = false;
                std::vector<uint32_t>().swap(sl.ixs[k]);        // release as we go
                std::vector<float>().swap(sl.vvs[k]);
            }
            if (!ok) {                                   // all or nothing: the slices must cover A
                for (auto *sp : plan->sweeps) sweep_plan_destroy(sp);
                plan->sweeps.clear();
            }
        }
        // the permutation's device side only when the sweep form it serves exists (a plan that fell back to the row-split
        // kernels multiplies by the caller's B as it is)
        if (permute && !plan->sweeps.empty()) {
            MGGCN_CHECK_HIP(hipMalloc(&plan->d_src_row, (size_t)n_cols * sizeof(uint32_t)));
            MGGCN_CHECK_HIP(hipMemcpy(plan->d_src_row, src_row.data(), (size_t)n_cols * sizeof(uint32_t), hipMemcpyHostToDevice));
            plan->bytes += (size_t)n_cols * sizeof(uint32_t);
            // scratch for the permuted copy at any width <= max_d (a narrow plan also serves wide calls)
            const size_t bb = (size_t)n_cols * ((max_d + 3) / 4 * 4) * sizeof(float);
            MGGCN_CHECK_HIP(hipMalloc(&plan->d_bperm, bb));
            plan->bytes += bb;
        }
        if (narrow && !plan->sweeps.empty()) {
            plan->bpad_dp = (d_hint + 15) / 16 * 16;       // 64-byte pitch: see sweep_wants_repack
            const size_t bb = (size_t)n_cols * plan->bpad_dp * sizeof(float);
            MGGCN_CHECK_HIP(hipMalloc(&plan->d_bpad, bb));
            plan->bytes += bb;
        }
    }
    plan->d_hint = d_hint;
    plan->nnz = n_rows ? (uint64_t)host_indptr[n_rows] - host_indptr[0] : 0;
    plan->build_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - t_build0).count();
    if (env_u32("MGGCN_SPMM_PLAN_LOG", 0u)) {
        char line[2048];
        mggcn_spmm_plan_describe(plan, line, sizeof line);
        std::fprintf(stderr, "[mggcn plan] %s\n", line);
    }
    return plan;
}

// Human-readable record of the plan builder's decisions (one line; sweep slices separated by " | "): which form,
// the column-popularity measurement and flag, panel rows, slices, lanes per entry, tasks / rounds per slice, padded
// entry stream, device bytes, host build time.  Returns the length written (snprintf semantics).
MGGCN_API int mggcn_spmm_plan_describe(const mggcn_spmm_plan *plan, char *out, size_t cap) {
    if (!plan || !out || !cap) return 0;
    size_t at = 0;
    auto put = [&](const char *fmt, auto... a) {
        if (at + 1 >= cap) return;
        const int k = std::snprintf(out + at, cap - at, fmt, a...);
        if (k > 0) at = std::min(cap - 1, at + (size_t)k);
    };
    put("rows=%u cols=%u nnz=%llu max_d=%u d_hint=%u form=%s hot_share=%.3f hot_columns=%d locality=%.3f permuted=%d mean_run=%.2f slices=%zu bytes=%zu build_s=%.3f",
        plan->n_rows, plan->n_cols, (unsigned long long)plan->nnz, plan->max_d, plan->d_hint,
        plan->sweeps.empty() ? "rowsplit" : (plan->d_bpad ? "sweep-narrow" : "sweep"), plan->hot_share, plan->hot_columns,
        plan->locality, plan->d_src_row ? 1 : 0, plan->mean_run, plan->sweeps.size(), mggcn_spmm_plan_bytes(plan), plan->build_s);
    if (plan->sweeps.empty()) put(" items=%u split_rows=%u", plan->n_items, plan->n_split_rows);
    for (auto *sp : plan->sweeps) {
        char one[256];
        sweep_plan_describe(sp, one, sizeof one);
        put(" | %s", one);
    }
    return (int)at;
}

MGGCN_API void mggcn_spmm_plan_destroy(mggcn_spmm_plan *plan) {
    if (!plan) return;
    for (auto *sp : plan->sweeps) sweep_plan_destroy(sp);
    if (plan->d_items) MGGCN_CHECK_HIP(hipFree(plan->d_items));
    if (plan->d_split) MGGCN_CHECK_HIP(hipFree(plan->d_split));
    if (plan->d_partial) MGGCN_CHECK_HIP(hipFree(plan->d_partial));
    if (plan->d_bpad) MGGCN_CHECK_HIP(hipFree(plan->d_bpad));
    if (plan->d_src_row) MGGCN_CHECK_HIP(hipFree(plan->d_src_row));
    if (plan->d_bperm) MGGCN_CHECK_HIP(hipFree(plan->d_bperm));
    delete plan;
}

MGGCN_API void mggcn_spmm_plan_concurrent_builders(uint32_t n) { mggcn_plan::set_concurrent_builders(n); }
MGGCN_API void mggcn_spmm_plan_reserved_cus(uint32_t n) { mggcn_plan::set_reserved_cus(n); }

MGGCN_API uint32_t mggcn_spmm_plan_num_items(const mggcn_spmm_plan *plan) { return plan->n_items; }
MGGCN_API uint32_t mggcn_spmm_plan_num_split_rows(const mggcn_spmm_plan *plan) { return plan->n_split_rows; }
MGGCN_API size_t mggcn_spmm_plan_bytes(const mggcn_spmm_plan *plan) {
    size_t b = plan->bytes;
    for (auto *sp : plan->sweeps) b += sweep_plan_bytes(sp);
    return b;
}
MGGCN_API uint32_t mggcn_spmm_plan_read_stamps(const mggcn_spmm_plan *plan, uint32_t slice, uint64_t *host_out,
                                               uint32_t capacity_tasks) {
    if (!plan || slice >= plan->sweeps.size()) return 0;
    return sweep_plan_read_stamps(plan->sweeps[slice], reinterpret_cast<unsigned long long *>(host_out), capacity_tasks);
}

MGGCN_API uint32_t mggcn_spmm_plan_num_slices(const mggcn_spmm_plan *plan) { return plan ? (uint32_t)plan->sweeps.size() : 0; }

MGGCN_API uint32_t mggcn_spmm_plan_num_launches(const mggcn_spmm_plan *plan, uint32_t d) {
    // kernel launches one mggcn_spmm_csr_f32 call makes with this plan at width d (aligned operands assumed)
    if (!plan) return 1;
    if (plan->sweeps.empty()) return 1 + (plan->n_split_rows ? 1 : 0);
    uint32_t n = (plan->d_bpad && (d + 15) / 16 * 16 <= plan->bpad_dp && d % 16 != 0) ? 1u : 0u;   // re-pitch pass
    if (plan->d_src_row && !n) n = 1;                    // the permuted copy of B (the narrow form folds it into the re-pitch)
    for (auto *sp : plan->sweeps) n += sweep_plan_launches(sp, d);
    return n;
}

MGGCN_API uint32_t mggcn_spmm_plan_num_sweep_tasks(const mggcn_spmm_plan *plan) {
    uint32_t t = 0;
    for (auto *sp : plan->sweeps) t += sweep_plan_tasks(sp);
    return t;
}

namespace {

template <bool HAS_ITEMS>
void launch_main(hipStream_t st, const mggcn_spmm_plan *plan, uint32_t n_items, const uint32_t *indptr,
                 const uint32_t *indices, const float *values, const float *B, size_t ldb, float *C,
                 size_t ldc, uint32_t d, float alpha, float beta, uint32_t flags, float slope) {
    const SpmmItem *items = HAS_ITEMS ? plan->d_items : nullptr;
    float *partial = HAS_ITEMS ? plan->d_partial : nullptr;
    const unsigned block = 256, waves_per_block = block / 64;
    const unsigned grid = (n_items + waves_per_block - 1) / waves_per_block;
    const bool vec_ok = (d % 4 == 0) && (ldb % 4 == 0) && (ldc % 4 == 0) && aligned16(B) && aligned16(C);
    if (vec_ok && d > 64) {
        hipLaunchKernelGGL((spmm_vec4_kernel<32, 8, HAS_ITEMS>), dim3(grid), dim3(block), 0, st, items,
                           n_items, indptr, indices, values, B, ldb, C, ldc, partial, d, alpha, beta,
                           flags, slope);
    } else if (vec_ok && d > 32) {
        hipLaunchKernelGGL((spmm_vec4_kernel<16, 4, HAS_ITEMS>), dim3(grid), dim3(block), 0, st, items,
                           n_items, indptr, indices, values, B, ldb, C, ldc, partial, d, alpha, beta,
                           flags, slope);
    } else if (vec_ok) {
        hipLaunchKernelGGL((spmm_vec4_kernel<8, 2, HAS_ITEMS>), dim3(grid), dim3(block), 0, st, items,
                           n_items, indptr, indices, values, B, ldb, C, ldc, partial, d, alpha, beta,
                           flags, slope);
    } else {
        hipLaunchKernelGGL((spmm_scalar_kernel<8, HAS_ITEMS>), dim3(grid), dim3(block), 0, st, items,
                           n_items, indptr, indices, values, B, ldb, C, ldc, partial, d, alpha, beta,
                           flags, slope);
    }
    MGGCN_CHECK_LAUNCH();
}

}  // namespace

MGGCN_API void mggcn_spmm_csr_f32(mggcn_stream_t stream, const mggcn_spmm_plan *plan, uint32_t n_rows,
                                  uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices,
                                  const float *values, const float *B, size_t ldb, float *C,
                                  size_t ldc, uint32_t d, float alpha, float beta, uint32_t flags,
                                  float slope) {
    if (n_rows == 0 || d == 0) return;
    MGGCN_REQUIRE(indptr && B && C, "null operand");
    MGGCN_REQUIRE(ldb >= d && ldc >= d, "leading dimension smaller than the feature width");
    MGGCN_REQUIRE((const void *)B != (const void *)C, "C must not alias B");
    MGGCN_REQUIRE((uint64_t)n_cols * ldb < (1ull << 40), "B too large");
    hipStream_t st = as_stream(stream);
    if (plan) {
        MGGCN_REQUIRE(plan->n_rows == n_rows && plan->n_cols == n_cols, "plan built for another matrix");
        MGGCN_REQUIRE(d <= plan->max_d || plan->n_slots == 0, "feature width exceeds the plan's max_d");
        if (!plan->sweeps.empty() && sweep_supports(plan->sweeps[0], d, ldb, ldc, B, C)) {
            const size_t S = plan->sweeps.size();
            const uint32_t dp = (d + 15) / 16 * 16;
            if (plan->d_bpad && dp <= plan->bpad_dp && (plan->d_src_row || sweep_wants_repack(plan->sweeps[0], d, ldb, B))) {
                sweep_repack(st, B, ldb, n_cols, d, plan->d_bpad, dp, plan->d_src_row);     // 64-byte pitched (and permuted) copy of B
                B = plan->d_bpad;
                ldb = dp;
            } else if (plan->d_src_row) {                                  // wide form on a permuted plan: B' = B[src_row]
                const uint32_t d4 = (d + 3) / 4 * 4;
                MGGCN_REQUIRE(plan->d_bperm != nullptr && d4 <= (plan->max_d + 3) / 4 * 4, "feature width exceeds the permuted plan's scratch");
                sweep_repack(st, B, ldb, n_cols, d, plan->d_bperm, d4, plan->d_src_row);
                B = plan->d_bperm;
                ldb = d4;
            }
            for (size_t k = 0; k < S; k++)       // beta only once, the fused activation only on the full sum
                sweep_launch(st, plan->sweeps[k], B, ldb, C, ldc, d, alpha, k == 0 ? beta : 1.f,
                             k + 1 == S ? flags : 0u, slope);
            return;
        }
        launch_main<true>(st, plan, plan->n_items, indptr, indices, values, B, ldb, C, ldc, d, alpha,
                          beta, flags, slope);
        if (plan->n_split_rows) {
            const unsigned grid = (plan->n_split_rows + 3) / 4;
            hipLaunchKernelGGL(spmm_combine_kernel, dim3(grid), dim3(256), 0, st, plan->d_split,
                               plan->n_split_rows, plan->d_partial, C, ldc, d, alpha, beta, flags, slope);
            MGGCN_CHECK_LAUNCH();
        }
    } else {
        launch_main<false>(st, nullptr, n_rows, indptr, indices, values, B, ldb, C, ldc, d, alpha, beta,
                           flags, slope);
    }
}
