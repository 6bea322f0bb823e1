// plan_host.cpp -- host passes of the SpMM plan builders (see plan_host.h).  No HIP in this file:
// it is also compiled by plain g++ under the sanitizers (`make sanitize`).
#include "plan_host.h"

#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <queue>
#include <thread>
#include <utility>

namespace mggcn_plan {

namespace {

// fail-fast like MGGCN_REQUIRE (common.h); only ever called on the thread that entered the builder
void require(bool ok, const char *what) {
    if (ok) return;
    std::fprintf(stderr, "MGGCN precondition failed in the plan builder: %s\n", what);
    std::exit(EXIT_FAILURE);
}

std::atomic<unsigned> g_concurrent_builders{1};
std::atomic<unsigned> g_reserved_cus{0};

// matrices below this many non-zeros are handled by the calling thread alone (MGGCN_HOST_THREADS_MIN_NNZ: tests)
uint64_t thread_threshold(uint64_t dflt) {
    const char *s = std::getenv("MGGCN_HOST_THREADS_MIN_NNZ");
    return (s && *s) ? std::strtoull(s, nullptr, 10) : dflt;
}

// passes over the rows of A, cut into ranges of about equal non-zeros, one std::thread each.  fn(thread, row_begin, row_end).
unsigned row_pass_threads(uint32_t n_rows, const uint32_t *indptr, unsigned cap = 32) {
    const uint64_t nnz = n_rows ? (uint64_t)indptr[n_rows] - indptr[0] : 0;
    return nnz > thread_threshold(1u << 22) ? host_threads(cap) : 1u;
}

template <typename F>
void rows_parallel(uint32_t n_rows, const uint32_t *indptr, unsigned T, F &&fn) {
    const uint64_t nnz = n_rows ? (uint64_t)indptr[n_rows] - indptr[0] : 0;
    T = std::max(1u, T);
    std::vector<uint32_t> cut(T + 1, n_rows);
    cut[0] = 0;
    for (unsigned t = 1; t < T; t++) {
        const uint64_t target = indptr[0] + nnz * t / T;
        cut[t] = (uint32_t)(std::lower_bound(indptr, indptr + n_rows, (uint32_t)target) - indptr);
        if (cut[t] < cut[t - 1]) cut[t] = cut[t - 1];
    }
    if (T == 1) { fn(0u, 0u, n_rows); return; }
    std::vector<std::thread> th;
    for (unsigned t = 0; t < T; t++) th.emplace_back(fn, t, cut[t], cut[t + 1]);
    for (auto &x : th) x.join();
}

struct VRow {
    uint32_t row, beg, end, dst, step;      // entries beg, beg + step, beg + 2 step, ... < end
    uint32_t len() const { return (end - beg + step - 1) / step; }
};

}  // namespace

uint32_t env_u32(const char *name, uint32_t dflt) {
    const char *s = std::getenv(name);
    if (!s || !*s) return dflt;
    return (uint32_t)std::strtoul(s, nullptr, 10);
}

void set_reserved_cus(unsigned n) { g_reserved_cus.store(n, std::memory_order_relaxed); }

void set_concurrent_builders(unsigned n) { g_concurrent_builders.store(std::max(1u, n), std::memory_order_relaxed); }

unsigned host_threads(unsigned cap) {
    if (const char *s = std::getenv("MGGCN_HOST_THREADS")) return std::max(1u, std::min(64u, (unsigned)std::strtoul(s, nullptr, 10)));
    const unsigned hw = std::max(1u, std::thread::hardware_concurrency());
    return std::max(1u, std::min(cap, hw / g_concurrent_builders.load(std::memory_order_relaxed)));
}

// ---- row-split form ----------------------------------------------------------------------------
RowSplitHost rowsplit_build(uint32_t n_rows, const uint32_t *indptr, uint32_t split) {
    RowSplitHost out;
    out.items.reserve((size_t)n_rows + 1024);
    for (uint32_t r = 0; r < n_rows; r++) {
        const uint32_t b = indptr[r], e = indptr[r + 1];
        require(e >= b, "indptr must be non-decreasing");
        const uint32_t len = e - b;
        if (len <= split + split / 2) {
            out.items.push_back({r, b, e, kNoSlot});
        } else {
            const uint32_t parts = (len + split - 1) / split;
            out.split_rows.push_back({r, out.n_slots, parts, 0});
            for (uint32_t k = 0; k < parts; k++) {
                const uint32_t kb = b + (uint32_t)((uint64_t)len * k / parts);
                const uint32_t ke = b + (uint32_t)((uint64_t)len * (k + 1) / parts);
                out.items.push_back({r, kb, ke, out.n_slots++});
            }
        }
    }
    // longest first (LPT); ties keep row order so neighbouring waves touch neighbouring C rows
    std::stable_sort(out.items.begin(), out.items.end(), [](const SpmmItem &a, const SpmmItem &b) { return (a.end - a.beg) > (b.end - b.beg); });
    return out;
}

// ---- whole-matrix passes -------------------------------------------------------------------------
bool columns_in_range(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices) {
    if (!n_rows || !indices) return true;
    const unsigned T = row_pass_threads(n_rows, indptr);
    std::vector<unsigned char> bad(T, 0);
    rows_parallel(n_rows, indptr, T, [&](unsigned t, uint32_t r0, uint32_t r1) {
        unsigned char b = 0;
        for (uint64_t e = indptr[r0]; e < indptr[r1]; e++) b |= (unsigned char)(indices[e] >= n_cols);
        bad[t] = b;
    });
    for (unsigned char b : bad) if (b) return false;
    return true;
}

ColumnStats column_stats(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices) {
    ColumnStats st;
    if (n_cols < 100 || !indices || !n_rows) return st;
    const uint64_t nz0 = indptr[0], nz1 = indptr[n_rows];
    const uint32_t gr = std::max<uint32_t>(1u, (n_rows + 31) / 32), gc = std::max<uint32_t>(1u, (n_cols + 31) / 32);
    // one counter array of n_cols words per thread: keep all of them together under 256 MiB (papers100M has 111 M
    // columns -- 32 threads, four builders side by side would be 57 GB of counters)
    unsigned T = row_pass_threads(n_rows, indptr);
    T = (unsigned)std::max<uint64_t>(1, std::min<uint64_t>(T, (64ull << 20) / std::max<uint32_t>(n_cols, 1u)));
    std::vector<std::vector<uint32_t>> ccs(T, std::vector<uint32_t>(n_cols, 0u));
    std::vector<uint64_t> nears(T, 0);
    rows_parallel(n_rows, indptr, T, [&](unsigned t, uint32_t r0, uint32_t r1) {
        uint32_t *cnt = ccs[t].data();
        uint64_t near_t = 0;
        for (uint32_t r = r0; r < r1; r++) {
            const uint32_t g = r / gr;
            for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++) {
                const uint32_t c = indices[e];
                if (c < n_cols) { cnt[c]++; near_t += (c / gc == g); }
            }
        }
        nears[t] = near_t;
    });
    std::vector<uint32_t> &cc = ccs[0];
    uint64_t near = nears[0];
    for (unsigned t = 1; t < T; t++) {
        near += nears[t];
        for (uint32_t c = 0; c < n_cols; c++) cc[c] += ccs[t][c];
    }
    st.locality = nz1 > nz0 ? (double)near / (double)(nz1 - nz0) : 0.0;
    const size_t top = std::max<size_t>(1, n_cols / 100);
    std::nth_element(cc.begin(), cc.begin() + top, cc.end(), std::greater<uint32_t>());
    uint64_t hot = 0;
    for (size_t k = 0; k < top; k++) hot += cc[k];
    st.hot_columns = (double)hot >= 0.05 * (double)(nz1 - nz0);
    st.hot_share = nz1 > nz0 ? (double)hot / (double)(nz1 - nz0) : 0.0;
    return st;
}

void column_permutation(uint32_t n_cols, std::vector<uint32_t> &pi, std::vector<uint32_t> &src_row) {
    pi.resize(n_cols);
    src_row.resize(n_cols);
    for (uint32_t c = 0; c < n_cols; c++) pi[c] = c;
    uint64_t x = 0x9E3779B97F4A7C15ull;                    // fixed-seed Fisher-Yates (splitmix64)
    for (uint32_t c = n_cols ? n_cols - 1 : 0; c > 0; c--) {
        x += 0x9E3779B97F4A7C15ull;
        uint64_t z = x;
        z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
        z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
        z ^= z >> 31;
        std::swap(pi[c], pi[(uint32_t)(z % (c + 1))]);
    }
    for (uint32_t c = 0; c < n_cols; c++) src_row[pi[c]] = c;
}

void permute_indices(uint32_t n_rows, const uint32_t *indptr, const uint32_t *indices, const std::vector<uint32_t> &pi,
                     std::vector<uint32_t> &out) {
    out.resize(n_rows ? indptr[n_rows] : 0);
    rows_parallel(n_rows, indptr, row_pass_threads(n_rows, indptr), [&](unsigned, uint32_t r0, uint32_t r1) {
        for (uint64_t e = indptr[r0]; e < indptr[r1]; e++) out[e] = pi[indices[e]];
    });
}

SliceBuckets slice_buckets(uint32_t n_rows, uint32_t S, uint32_t width, const uint32_t *indptr, const uint32_t *indices,
                           const float *values) {
    // two passes over A whatever the slice count
    SliceBuckets b;
    b.ips.assign(S, std::vector<uint32_t>((size_t)n_rows + 1, 0u));
    const unsigned T = row_pass_threads(n_rows, indptr);
    rows_parallel(n_rows, indptr, T, [&](unsigned, uint32_t r0, uint32_t r1) {       // per-row counts: no sharing
        for (uint32_t r = r0; r < r1; r++)
            for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++) b.ips[indices[e] / width][r + 1]++;
    });
    b.ixs.resize(S);
    b.vvs.resize(S);
    for (uint32_t k = 0; k < S; k++) {
        for (uint32_t r = 0; r < n_rows; r++) b.ips[k][r + 1] += b.ips[k][r];
        b.ixs[k].resize(b.ips[k][n_rows]);
        b.vvs[k].resize(b.ips[k][n_rows]);
    }
    rows_parallel(n_rows, indptr, T, [&](unsigned, uint32_t r0, uint32_t r1) {       // every row knows its offsets
        std::vector<uint32_t> pos(S);
        for (uint32_t r = r0; r < r1; r++) {
            for (uint32_t k = 0; k < S; k++) pos[k] = b.ips[k][r];
            for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++) {
                const uint32_t k = indices[e] / width;
                b.ixs[k][pos[k]] = indices[e];
                b.vvs[k][pos[k]] = values[e];
                pos[k]++;
            }
        }
    });
    return b;
}

// ---- sweep form ------------------------------------------------------------------------------------
uint32_t sweep_lanes_per_entry(uint32_t d_hint) {
    const uint32_t need = (d_hint + 3) / 4;          // float4 lanes that cover a row
    return need <= 4 ? 4u : need <= 8 ? 8u : need <= 12 ? 12u : 16u;
}

// hot_columns: a few columns carry much of the matrix (see mggcn_spmm_plan_create_for)
uint32_t sweep_panel_rows(uint32_t d_hint, bool hot_columns) {
    if (d_hint >= 1 && d_hint <= 64) {
        // 1.5 MiB of B per panel at the 64-byte-multiple pitch: 8192 rows at d = 41 (best of 8192 /
        // 16384 / 32768 on both Reddit matrices), 24576 at d = 16 (16384 beat 8192 there)
        const uint32_t pitch = (d_hint + 15) / 16 * 64;
        const uint32_t rows = std::max(1024u, (3u << 19) / pitch / 1024u * 1024u);
        return std::max<uint32_t>(64u, env_u32("MGGCN_SPMM_PANEL_ROWS_NARROW", rows));
    }
    // Round 1 (no priority rotation): 6144 rows for the forward matrix (hot columns), 4096 for the backward one.  With the
    // waves of a SIMD equalised (rotate_priority) 4096-row panels (2 MiB) win on both: forward 2.45 -> 2.42 ms, and the
    // backward matrix no longer needs 32-MiB slices (profiles/experiments/retune_after_rotation_r02.log).
    (void)hot_columns;
    return std::max<uint32_t>(64u, env_u32("MGGCN_SPMM_PANEL_ROWS", 4096u));
}

bool sweep_build_host(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices, const float *values,
                      uint32_t max_d, bool force, uint32_t d_hint, bool hot_columns, uint32_t num_cu, SweepHost &P) {
    if (!n_rows || !indices || !values) return false;
    num_cu = std::max(1u, std::min(num_cu, kNumCuMask));
    // narrow form (quad kernel): runs padded to 4 entries, tasks to 16, wider panels (the L2
    // window is counted in bytes: a 176-byte row lets three times as many rows stay resident)
    const bool narrow = d_hint >= 1 && d_hint <= 64 && n_cols <= (1u << 24);
    // lanes per gathered row / entries per gather instruction: the fewest lanes that cover the row
    // give the most rows per instruction, but every run is padded to G entries -- on a matrix with
    // many short runs (power-law rows) a smaller G wins.  Decided after the counting pass below.
    uint32_t lpe = !narrow ? 0u : sweep_lanes_per_entry(d_hint);
    uint32_t G = narrow ? 64u / lpe : 2u;
    if (n_cols > kColMask) return false;                            // column does not fit the packed entry
    static_assert(kRW == 16, "4 row bits in the packed entry");
    const uint64_t nnz = (uint64_t)indptr[n_rows] - indptr[0];
    if (!force && nnz < env_u32("MGGCN_SPMM_SWEEP_MIN_NNZ", 1u << 20)) return false;   // small graphs: row-split is fine
    const uint32_t panel_rows = sweep_panel_rows(d_hint, hot_columns);

    // resident waves per launch ("round").  Registers would admit 6 blocks of 4 waves per CU
    // (56 VGPRs; ~106 SGPRs -> floor(800 / (ceil(sgpr/16)*16 + 16)) = 6, MI355X_MICROARCH.md
    // residency rule), but FEWER waves keep the sweep tighter: the spread of the waves over the
    // column space is what decides the L2 hit rate.  Measured on the Reddit shape, d = 128
    // (profiles/experiments/sweep_vs_rowsplit.py): 2 blocks/CU 3.96 ms, 3 -> 3.34 ms, 4 -> 4.0,
    // 5 -> 4.0, 6 -> 4.4 (row-split kernel: 5.96 ms).  With the float4 pair kernel and 32 MiB
    // column slices (spmm.hip) the optimum moved to 4 blocks/CU, 8192-row panels: 2.83 ms; with
    // the accumulators in reserved registers and the six-instruction fold, 6144-row panels and
    // 64 MiB slices: 2.69 ms (3 blocks/CU 3.11, 5 -> 3.56; the float4 kernels hold 128 VGPRs, so
    // four blocks of four waves is also what fits).
    const uint32_t blocks_per_cu = std::max<uint32_t>(1u, std::min<uint32_t>(env_u32("MGGCN_SPMM_SWEEP_BLOCKS_PER_CU", 4u), 8u));
    // Room for a kernel that SHARES the device with the SpMM -- the channels of a collective (RCCL) kernel: every workgroup of it
    // that sits on a CU displaces one of the four SpMM workgroups there, and a round of EXACTLY the resident set then ends in a
    // second, nearly empty pass.  Measured with a stand-in (profiles/experiments/coresident_r04.log; rank 0's share of the Reddit
    // shape at P = 2, two full rounds per SpMM): 1.29 ms alone, 2.01 ms next to as few as 16 foreign workgroups; with room left,
    // 1.25-1.27 ms next to 16-64 of them.  `reserved` = compute units' worth of wave slots EVERY launch round must leave free
    // (set per plan build by the distributed host layers, mggcn_spmm_plan_reserved_cus; MGGCN_SPMM_RESERVED_CUS overrides; 0 =
    // the device is ours: every single-GPU plan).  It is a minimum, not a cut: a matrix whose tasks leave that room anyway --
    // one round, not full: a rank's pieces at P = 4 / 8 -- keeps the full round size (fewer waves in flight cost 3-6 % when
    // nobody shares the device, rank_epoch_reserve_ab_r04.log); only plans whose rounds would be full are built on smaller rounds.
    const uint32_t reserved = std::min(env_u32("MGGCN_SPMM_RESERVED_CUS", g_reserved_cus.load(std::memory_order_relaxed)), num_cu - 1u);
    const uint32_t full_round = num_cu * blocks_per_cu * kWavesPerBlock, reserved_tasks = reserved * blocks_per_cu * kWavesPerBlock;
    uint32_t round_tasks = full_round;

    // (MGGCN_SPMM_SWEEP_ROWS_PER_TASK caps it for experiments: 8 rows per wave -- twice the launches,
    //  same slices -- ran 2.88 ms against 2.69 at 16, 4 rows 3.00: profiles/experiments/sweep_rows_per_task_r01.log)
    // 1. virtual rows: slices of heavy rows get partial-sum slots
    // rows per task: 16 when there are enough rows to fill a round, fewer for small row blocks
    // (a rank's share at P = 8 has 29 k rows: 16 rows per wave would leave 7 waves per CU)
    const uint32_t cap_limit = std::min<uint32_t>((uint32_t)kRW, std::max(1u, env_u32("MGGCN_SPMM_SWEEP_ROWS_PER_TASK", (uint32_t)kRW)));
    uint32_t cap_rows = 1;
    auto first_cap = [&] { cap_rows = std::max<uint32_t>(1u, std::min<uint32_t>(cap_limit, (n_rows + round_tasks - 1) / round_tasks)); };
    first_cap();
    uint32_t t_est = 0, target = 0, split = 0;
    auto derive_split = [&] {
        t_est = (n_rows + cap_rows - 1) / cap_rows;
        target = (uint32_t)std::max<uint64_t>(1, nnz / t_est);
        split = std::max<uint32_t>(256u, std::min<uint32_t>(env_u32("MGGCN_SPMM_SWEEP_SPLIT", target / 2), 1u << 20));
    };
    derive_split();
    std::vector<VRow> vrows;
    vrows.reserve((size_t)n_rows + 4096);
    std::vector<SweepSplitRow> &split_rows = P.split_rows;
    split_rows.clear();
    uint32_t n_slots = 0;
    uint32_t T = 0;
    std::vector<std::vector<uint32_t>> bins;
    // EXPERIMENT (MGGCN_SPMM_XCD_COLUMNS=1, wide form only): a column partition across the 8 XCDs.
    // Every row is cut into 8 column slices (entries regrouped by slice), every slice of every row gets a partial-sum
    // slot, tasks hold rows of ONE slice, and the task table is laid out so that the workgroups the dispatcher deals to
    // XCD x (block index mod 8 == x) only ever touch slice x of B: each XCD's L2 pulls 1/8 of B per round of resident
    // tasks instead of all of it.  The price: 8 partial rows per output row (written, then summed by
    // sweep_combine_kernel) and 8x as many one-wave tasks.  Kernels unchanged.  profiles/experiments/xcd_columns_r03.log.
    const uint32_t XS = (!narrow && env_u32("MGGCN_SPMM_XCD_COLUMNS", 0u) && n_cols >= 8u * panel_rows) ? 8u : 1u;
    const uint32_t xs_width = (n_cols + XS - 1) / XS;
    std::vector<uint32_t> idx2;
    std::vector<float> val2;
    const uint32_t *ix = indices;
    const float *vv = values;
    auto lpt = [&](const std::vector<uint32_t> &members, uint32_t n_bins, auto &&bin_of) {
        // equal-work bins of <= cap_rows virtual rows: longest first into the lightest bin
        std::vector<uint32_t> order(members);
        std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return vrows[a].len() > vrows[b].len(); });
        std::vector<uint64_t> load(n_bins, 0);
        using HeapItem = std::pair<uint64_t, uint32_t>;        // (load, bin): smallest load first, then lowest id
        std::priority_queue<HeapItem, std::vector<HeapItem>, std::greater<HeapItem>> heap;
        for (uint32_t t = 0; t < n_bins; t++) heap.push({0, t});
        for (uint32_t vi : order) {
            require(!heap.empty(), "sweep plan: task capacity exhausted");
            const HeapItem top = heap.top();
            heap.pop();
            const uint32_t t = top.second;
            auto &bin = bins[bin_of(t)];
            bin.push_back(vi);
            load[t] += vrows[vi].len();
            if (bin.size() < (size_t)cap_rows) heap.push({load[t], t});
        }
    };
    if (XS == 1) {
      for (;;) {
        vrows.clear();
        split_rows.clear();
        n_slots = 0;
        for (uint32_t r = 0; r < n_rows; r++) {
            const uint32_t b = indptr[r], e = indptr[r + 1];
            require(e >= b, "indptr must be non-decreasing");
            const uint32_t len = e - b;
            if (len <= split + split / 2) {
                vrows.push_back({r, b, e, r, 1u});
            } else {
                // A heavy row is cut into INTERLEAVED slices (slice k = entries k, k + parts, k + 2 parts, ...), not into
                // contiguous ranges: a dataset written by scipy / the reference's prep.py, and every transposed matrix, holds
                // its rows sorted by column, so a contiguous slice covers 1 / parts of the column space -- its wave sweeps
                // a few panels only, out of step with every other wave of the chip (the premise of the sweep).  Measured on
                // the symmetric Reddit stand-in: d = 128 SpMM 3.09 ms with contiguous slices against 2.36 with the same
                // rows shuffled (profiles/experiments/symmetric_r03_*.log); interleaved, every slice sees the whole column
                // distribution of its row whatever the order.
                const uint32_t parts = (len + split - 1) / split;
                split_rows.push_back({r, n_slots, parts, 0});
                for (uint32_t k = 0; k < parts; k++) vrows.push_back({r, b + k, e, kSlotFlag | n_slots++, parts});
            }
        }
        // 2. tasks
        T = (uint32_t)((vrows.size() + cap_rows - 1) / cap_rows);
        // A launch round that is slightly over-full (the slices of heavy rows count as rows too) would become TWO rounds of
        // half-length tasks -- every per-task cost twice and a second launch: a rank's backward pieces at P = 8 ran 12 % slower
        // that way once 16 CUs' worth of slots were reserved (profiles/experiments/rank_epoch_reserve_ab_r04.log).  One more
        // row per task keeps it one round.
        if (T > round_tasks && T <= round_tasks + round_tasks / 4 && cap_rows < cap_limit) {
            cap_rows++;
            derive_split();
            continue;
        }
        // the room asked for: one round with at least that many free slots, or every round on the smaller size
        if (reserved_tasks && round_tasks == full_round && (T > full_round || full_round - T < reserved_tasks)) {
            round_tasks = full_round - reserved_tasks;
            first_cap();
            derive_split();
            continue;
        }
        break;
      }
        if (T > round_tasks) {
            T = (T + round_tasks - 1) / round_tasks * round_tasks;
        } else {
            // ONE round: use every wave slot it has (minus the room asked for) -- the kernels are bound by what sixteen waves per
            // CU get through the texture path, and ceil(rows / rows-per-task) tasks left 11 % of the slots of a rank's pieces at
            // P = 8 empty (3 641 of 4 096).  The same rows over more bins: shorter tasks, all of them in flight.
            const uint32_t avail = round_tasks == full_round ? full_round - reserved_tasks : round_tasks;
            T = std::max<uint32_t>(T, std::min<uint32_t>(avail, (uint32_t)vrows.size()));
        }
        T = std::max<uint32_t>(T, 1u);
        bins.resize(T);
        std::vector<uint32_t> all(vrows.size());
        for (size_t i = 0; i < all.size(); i++) all[i] = (uint32_t)i;
        lpt(all, T, [](uint32_t t) { return t; });
    } else {
        // regroup every row's entries by column slice (stable): sub-row (r, s) = [sub[r*XS+s], sub[r*XS+s+1])
        idx2.resize(nnz);
        val2.resize(nnz);
        std::vector<uint32_t> sub((size_t)n_rows * XS + 1, 0u);
        for (uint32_t r = 0; r < n_rows; r++)
            for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++) {
                require(indices[e] < n_cols, "column index out of range");
                sub[(size_t)r * XS + indices[e] / xs_width + 1]++;
            }
        for (size_t k = 0; k + 1 < sub.size(); k++) sub[k + 1] += sub[k];
        {
            std::vector<uint32_t> cur(sub.begin(), sub.end() - 1);
            for (uint32_t r = 0; r < n_rows; r++)
                for (uint32_t e = indptr[r]; e < indptr[r + 1]; e++) {
                    const uint32_t at = cur[(size_t)r * XS + indices[e] / xs_width]++;
                    idx2[at] = indices[e];
                    val2[at] = values[e];
                }
        }
        ix = idx2.data();
        vv = val2.data();
        const uint32_t t_est_x = (uint32_t)(((uint64_t)n_rows * XS + cap_rows - 1) / cap_rows);
        const uint32_t split_x = std::max<uint32_t>(256u, (uint32_t)std::max<uint64_t>(1, nnz / t_est_x) / 2);
        std::vector<std::vector<uint32_t>> by_slice(XS);
        for (uint32_t r = 0; r < n_rows; r++) {
            const uint32_t first = n_slots;
            for (uint32_t sl = 0; sl < XS; sl++) {
                const uint32_t b = sub[(size_t)r * XS + sl], e = sub[(size_t)r * XS + sl + 1], len = e - b;
                if (!len) continue;
                const uint32_t parts = len <= split_x + split_x / 2 ? 1u : (len + split_x - 1) / split_x;
                for (uint32_t k = 0; k < parts; k++) {
                    by_slice[sl].push_back((uint32_t)vrows.size());
                    vrows.push_back({r, b + k, e, kSlotFlag | n_slots++, parts});
                }
            }
            split_rows.push_back({r, first, n_slots - first, 0});        // every row is combined from its slices (0 slots: C = beta C)
        }
        uint32_t Ts = 1;
        for (const auto &m : by_slice) Ts = std::max<uint32_t>(Ts, (uint32_t)((m.size() + cap_rows - 1) / cap_rows));
        const uint32_t per_round = round_tasks / XS;                      // tasks of one slice per launch
        if (Ts > per_round) Ts = (Ts + per_round - 1) / per_round * per_round;
        Ts = (Ts + kWavesPerBlock - 1) / kWavesPerBlock * kWavesPerBlock;     // whole blocks: task q of slice sl lives in block (q / 4) * XS + sl
        T = Ts * XS;
        bins.resize(T);
        // task q of slice sl sits in block (q / 4) * XS + sl: the dispatcher deals block b to XCD b mod 8
        for (uint32_t sl = 0; sl < XS; sl++)
            lpt(by_slice[sl], Ts, [&](uint32_t q) { return ((q / kWavesPerBlock) * XS + sl) * kWavesPerBlock + q % kWavesPerBlock; });
    }
    auto task_col_base = [&](uint32_t t) { return XS == 1 ? 0u : ((t / kWavesPerBlock) % XS) * xs_width; };
    // 3. entry stream per task, sorted by (column panel, local row), original order inside a run.
    //    Every (panel,row) run is padded to an EVEN number of entries (a zero-valued copy of its
    //    last entry) and the two entries of each consecutive pair are ordered by column: the
    //    float4 kernel gathers one pair per instruction, one entry per half-wave.
    const uint32_t n_panels = ((XS == 1 ? n_cols : xs_width) + panel_rows - 1) / panel_rows;
    // threads from 2^19 non-zeros on: a rank's blocks at P = 8 on the Reddit shape hold 1.8-3.2 M each, and the 160 plans of
    // the single-process form were 5 s of one-thread work inside the first epoch
    const unsigned NT = nnz > thread_threshold(1u << 19) ? host_threads(64) : 1u;
    auto run_parallel = [&](auto &&fn) {
        if (NT <= 1) { fn(0u); return; }
        std::vector<std::thread> th;
        for (unsigned i = 0; i < NT; i++) th.emplace_back(fn, i);
        for (auto &x : th) x.join();
    };
    const size_t n_buckets = (size_t)n_panels * kRW;
    // (column indices were range-checked once by the caller: mggcn_plan::columns_in_range)
    auto count_buckets = [&](uint32_t t, std::vector<uint32_t> &cnt) {
        std::fill(cnt.begin(), cnt.end(), 0u);
        const auto &bin = bins[t];
        const uint32_t base = task_col_base(t);
        for (size_t r = 0; r < bin.size(); r++) {
            const VRow &v = vrows[bin[r]];
            for (uint32_t e = v.beg; e < v.end; e += v.step) cnt[(size_t)((ix[e] - base) / panel_rows) * kRW + r]++;
        }
    };
    // candidate forms: this lpe and every wider one (4 -> 16 entries per instruction, 8 -> 8, 12 -> 5, 16 -> 4)
    static const uint32_t kLpes[4] = {4u, 8u, 12u, 16u};
    std::vector<uint32_t> cand;
    if (narrow) {
        const uint32_t forced = env_u32("MGGCN_SPMM_NARROW_LPE", 0u);
        for (uint32_t l : kLpes)
            if (l >= lpe && (!forced || l == forced)) cand.push_back(l);
        if (cand.empty()) cand.push_back(lpe);
    }
    const size_t NC = narrow ? cand.size() : 1;
    std::vector<uint64_t> plen_c((size_t)T * NC, 0);
    run_parallel([&](unsigned tid) {
        std::vector<uint32_t> cnt(n_buckets);
        for (uint32_t t = tid; t < T; t += NT) {
            count_buckets(t, cnt);
            for (size_t k = 0; k < NC; k++) {
                const uint32_t g = narrow ? 64u / cand[k] : G;
                uint64_t len = 0;
                for (uint32_t c : cnt) len += (c + g - 1) / g * g;
                plen_c[(size_t)t * NC + k] = len;
            }
        }
    });
    size_t pick = 0;
    if (narrow) {
        // cost of one gather instruction ~ 3 + 2.7 cycles per 128-byte line touched (fitted:
        // profiles/experiments/narrow_backward.py -- at d = 41 five rows per instruction are no faster
        // than four on the even-row forward matrix, 13 % slower on the power-law backward one; at
        // d = 16 sixteen rows per instruction are 1.5x faster than four); rows are pitched to a
        // multiple of 64 bytes
        const double lines = std::max(1.0, std::ceil(((d_hint + 15) / 16 * 64) / 128.0));
        double best = 0;
        for (size_t k = 0; k < NC; k++) {
            const uint32_t g = 64u / cand[k];
            uint64_t tot = 0;
            for (uint32_t t = 0; t < T; t++) tot += plen_c[(size_t)t * NC + k];
            const double cost = (double)tot * (3.0 + 2.7 * lines * g) / g;
            if (k == 0 || cost < best) { best = cost; pick = k; }
        }
        lpe = cand[pick];
        G = 64u / lpe;
    }
    // task streams are whole steps of the narrow kernel (4 G entries) AND whole 8-entry batches of the others
    const uint32_t batch_pad = !narrow ? 8u : (G == 5u ? 40u : std::max(4u * G, 8u));
    std::vector<uint64_t> plen(T, 0);
    for (uint32_t t = 0; t < T; t++) plen[t] = plen_c[(size_t)t * NC + pick];
    std::vector<uint64_t>().swap(plen_c);
    std::vector<SweepTask> &tasks = P.tasks;
    std::vector<uint32_t> &task_rows = P.task_rows;
    tasks.assign(T, SweepTask{0, 0, 0, 0});
    task_rows.assign((size_t)T * kRW, 0u);
    uint64_t off = 0;
    for (uint32_t t = 0; t < T; t++) {
        tasks[t].beg = (uint32_t)off;
        off += (plen[t] + batch_pad - 1) / batch_pad * batch_pad;   // whole 64-byte batches (two per step in the quad form)
        tasks[t].end = (uint32_t)off;
        tasks[t].n_rows = (uint32_t)bins[t].size();
        tasks[t].pad = 0;
        for (size_t r = 0; r < bins[t].size(); r++) task_rows[(size_t)t * kRW + r] = vrows[bins[t][r]].dst;
    }
    require(off < (1ull << 32), "sweep plan: entry stream exceeds 32-bit offsets");
    std::vector<Entry> &entries = P.entries;
    entries.assign(off, Entry{0u, 0u});
    run_parallel([&](unsigned tid) {
        std::vector<uint32_t> cnt(n_buckets), start(n_buckets + 1), cur(n_buckets);
        for (uint32_t t = tid; t < T; t += NT) {
            count_buckets(t, cnt);
            start[0] = 0;
            for (size_t k = 0; k < n_buckets; k++) start[k + 1] = start[k] + (cnt[k] + G - 1) / G * G;
            std::copy(start.begin(), start.begin() + n_buckets, cur.begin());
            Entry *out = entries.data() + tasks[t].beg;
            const auto &bin = bins[t];
            const uint32_t base = task_col_base(t);
            for (size_t r = 0; r < bin.size(); r++) {
                const VRow &v = vrows[bin[r]];
                for (uint32_t e = v.beg; e < v.end; e += v.step) {
                    const uint32_t c = ix[e];
                    const uint32_t at = cur[(size_t)((c - base) / panel_rows) * kRW + r]++;
                    uint32_t vb;
                    std::memcpy(&vb, &vv[e], 4);
                    out[at] = Entry{(((uint32_t)r & (kRW - 1)) << kColBits) | c, vb};
                }
            }
            for (size_t k = 0; k < n_buckets; k++) {
                if (!cnt[k]) continue;
                const uint32_t s0 = start[k], s1 = start[k + 1];
                for (uint32_t q = s0 + cnt[k]; q < s1; q++) out[q] = Entry{out[s0 + cnt[k] - 1].x, 0u};   // pad the run
                if (G % 2 == 0)
                    for (uint32_t q = s0; q < s1; q += 2)                            // pair: lower column first
                        if ((out[q].x & kColMask) > (out[q + 1].x & kColMask)) std::swap(out[q], out[q + 1]);
                out[s0].x |= kRunFlag;                                               // first entry of the run
            }
            // tail padding up to the 8-entry batch: zero-valued copies of the last entry, no run flag
            const uint32_t real = (uint32_t)plen[t], padded = tasks[t].end - tasks[t].beg;
            for (uint32_t k = real; k < padded; k++) out[k] = Entry{real ? out[real - 1].x & ~kRunFlag : 0u, 0u};
        }
    });

    P.n_rows = n_rows; P.n_cols = n_cols; P.max_d = max_d;
    P.n_tasks = T; P.round_tasks = round_tasks; P.run_pad = G; P.lpe = lpe;
    P.n_slots = n_slots;
    P.panel_rows = panel_rows; P.n_entries = off; P.num_cu = num_cu;
    {   // priority rotation (float4 kernels; see rotate_priority): period 2^8 entries / every chunk of the narrow stream
        const uint32_t rot = (env_u32("MGGCN_SPMM_PRIO_ROTATE", 1u) ? kFlagPrioRotate : 0u) | (num_cu << kNumCuPos);
        P.prio_bits_wide = rot | (std::min(env_u32("MGGCN_SPMM_PRIO_SHIFT", 8u), 15u) << kPrioShiftPos);
        P.prio_bits_narrow = rot | (std::min(env_u32("MGGCN_SPMM_PRIO_SHIFT_NARROW", 1u), 15u) << kPrioShiftPos);
        P.tasks_per_wave = std::max(1u, env_u32("MGGCN_SPMM_TASKS_PER_WAVE", 1u));
        P.allow_quad = env_u32("MGGCN_SPMM_SWEEP_QUAD", 1u) != 0;
        P.allow_vec4 = env_u32("MGGCN_SPMM_SWEEP_VEC4", 1u) != 0;
        P.fast_pairs = env_u32("MGGCN_SPMM_FAST_PAIRS", 1u) != 0;
    }
    return true;
}

}  // namespace mggcn_plan
