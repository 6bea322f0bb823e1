// plan_host_test.cpp -- the host-only code of the engine on the CPU, built by plain g++ under
// AddressSanitizer + UndefinedBehaviorSanitizer and again under ThreadSanitizer
// (tests/native/Makefile: `make sanitize`; driven by tests/test_sanitizers.py).
//
// What runs here is the code the GPU path trusts blindly and that has real host concurrency:
//   * mg-gcn_amd/csrc/plan_host.cpp   the SpMM plan builders' host passes (threaded): row-split items, column
//                                     statistics, column permutation, slice bucketing, LPT task assignment and
//                                     the packed entry streams of the sweep kernels
//   * mg-gcn_amd/csrc/host_prep.cpp   normalize / transpose / P x P block split (threaded)
//   * mg-gcn_amd/host/enqueue.hpp     the per-GPU command queues of the single-process host layer
// checked against the CPU oracle (oracle/mggcn_oracle.c: test infrastructure, linked here and only here):
// the entry streams are DECODED and replayed as an SpMM in double precision -- every non-zero exactly once,
// padding contributes nothing, partial-sum slots add up -- and the structural promises the kernels rely on
// (run flags, run padding, pair order, batch padding, panel order) are asserted entry by entry.
// MGGCN_HOST_THREADS=4 MGGCN_HOST_THREADS_MIN_NNZ=1 make every pass threaded at these small sizes.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <stdexcept>
#include <string>
#include <vector>

#include "enqueue.hpp"
#include "mggcn.h"
#include "plan_host.h"

extern "C" {
void orc_set_num_threads(int n);
void orc_spmm_csr_f64acc(uint32_t n_rows, const uint32_t *indptr, const uint32_t *indices, const float *values, const float *B,
                         size_t ldb, float *C, size_t ldc, uint32_t d, float alpha, float beta);
void orc_csr_normalize(uint32_t n, uint32_t m, const uint32_t *indptr, const uint32_t *indices, float *data, int axis);
void orc_csr_transpose(uint32_t n, uint32_t m, const uint32_t *indptr, const uint32_t *indices, const float *data, uint32_t *t_indptr,
                       uint32_t *t_indices, float *t_data);
void orc_block_split_count(const uint32_t *indptr, const uint32_t *indices, uint32_t row_beg, uint32_t row_end, const uint32_t *q,
                           uint32_t nq, uint32_t *blk_indptr);
void orc_block_split_fill(const uint32_t *indptr, const uint32_t *indices, const float *data, uint32_t row_beg, uint32_t row_end,
                          const uint32_t *q, uint32_t nq, const uint32_t *blk_indptr, uint32_t *const *blk_indices, float *const *blk_data);
}

using namespace mggcn_plan;

static int g_failures = 0;
#define CHECK(cond)                                                                      \
    do {                                                                                 \
        if (!(cond)) {                                                                   \
            if (g_failures < 40) std::fprintf(stderr, "FAILURE: %s at %s:%d\n", #cond, __FILE__, __LINE__); \
            g_failures++;                                                                \
        }                                                                                \
    } while (0)

struct lcg {
    std::uint64_t s;
    std::uint32_t next() { s = s * 6364136223846793005ull + 1442695040888963407ull; return (std::uint32_t)(s >> 33); }
    float unit() { return (float)(next() & 0xFFFFFF) / (float)0x1000000; }
};

struct Csr {
    uint32_t n = 0, m = 0;
    std::vector<uint32_t> ip, ix;
    std::vector<float> v;
};

// kind 0: even rows, random columns; 1: power-law rows (a few giants), empty rows, duplicates; 2: community graph
// (columns near the diagonal), rows SORTED by column; 3: one column only
static Csr make_csr(uint32_t n, uint32_t m, uint32_t mean_deg, int kind, uint64_t seed, bool unit_values) {
    lcg g{seed};
    Csr A;
    A.n = n; A.m = m;
    A.ip.assign(n + 1, 0);
    for (uint32_t r = 0; r < n; r++) {
        uint32_t deg = mean_deg / 2 + g.next() % (mean_deg + 1);
        if (kind == 1) {
            if (r % 7 == 3) deg = 0;
            if (r % 211 == 5) deg = mean_deg * 40;
            if (r % 1009 == 17) deg = mean_deg * 300;
        }
        std::vector<uint32_t> cols;
        for (uint32_t k = 0; k < deg; k++) {
            uint32_t c;
            if (kind == 2) { const uint32_t w = std::max(1u, m / 64); c = (uint32_t)(((uint64_t)r * m / n) / w * w + g.next() % w) % m; }
            else if (kind == 3) c = 0;
            else if (kind == 1 && k % 5 == 0) c = g.next() % std::max(1u, m / 100);       // popular columns
            else c = g.next() % m;
            cols.push_back(c);
        }
        if (kind == 2) std::sort(cols.begin(), cols.end());
        for (uint32_t c : cols) { A.ix.push_back(c); A.v.push_back(unit_values ? 1.f : 0.25f + g.unit()); }
        A.ip[r + 1] = (uint32_t)A.ix.size();
    }
    return A;
}

static std::vector<double> spmm_ref(const Csr &A, const std::vector<float> &B, uint32_t d) {
    std::vector<double> C((size_t)A.n * d, 0.0);
    for (uint32_t r = 0; r < A.n; r++)
        for (uint32_t e = A.ip[r]; e < A.ip[r + 1]; e++)
            for (uint32_t k = 0; k < d; k++) C[(size_t)r * d + k] += (double)A.v[e] * (double)B[(size_t)A.ix[e] * d + k];
    return C;
}

static double max_rel(const std::vector<double> &got, const std::vector<double> &want) {
    double num = 0, den = 0;
    for (size_t i = 0; i < want.size(); i++) { num = std::max(num, std::fabs(got[i] - want[i])); den = std::max(den, std::fabs(want[i])); }
    return num / (den + 1e-300);
}

// ---- row-split form ---------------------------------------------------------------------------------
static void test_rowsplit(const Csr &A, uint32_t split) {
    const RowSplitHost rs = rowsplit_build(A.n, A.ip.data(), split);
    std::vector<uint32_t> covered(A.ix.size(), 0);
    std::map<uint32_t, std::vector<uint32_t>> slots_of_row;
    uint32_t prev_len = 0xFFFFFFFFu;
    for (const auto &it : rs.items) {
        CHECK(it.row < A.n && it.beg >= A.ip[it.row] && it.end <= A.ip[it.row + 1] && it.beg <= it.end);
        CHECK(it.end - it.beg <= prev_len);                      // longest first
        prev_len = it.end - it.beg;
        for (uint32_t e = it.beg; e < it.end; e++) covered[e]++;
        if (it.slot != kNoSlot) slots_of_row[it.row].push_back(it.slot);
    }
    for (uint32_t c : covered) CHECK(c == 1);
    CHECK(rs.split_rows.size() == slots_of_row.size());
    uint32_t slots = 0;
    for (const auto &sr : rs.split_rows) {
        auto &v = slots_of_row[sr.row];
        std::sort(v.begin(), v.end());
        CHECK(v.size() == sr.n_slots && !v.empty() && v.front() == sr.first_slot && v.back() == sr.first_slot + sr.n_slots - 1);
        slots += sr.n_slots;
    }
    CHECK(slots == rs.n_slots);
    // rows without slots appear exactly once
    std::vector<uint32_t> whole(A.n, 0);
    for (const auto &it : rs.items) if (it.slot == kNoSlot) whole[it.row]++;
    for (uint32_t r = 0; r < A.n; r++) CHECK(whole[r] == (slots_of_row.count(r) ? 0u : 1u));
}

// ---- sweep form: decode the entry stream and replay it -----------------------------------------------
static void check_sweep(const Csr &A, const uint32_t *ip, const uint32_t *ix, const float *vv, uint32_t d_hint, uint32_t num_cu,
                        const std::vector<float> &B, uint32_t d, std::vector<double> &C_accum, bool xcd_columns = false) {
    SweepHost h;
    const bool ok = sweep_build_host(A.n, A.m, ip, ix, vv, 128, true, d_hint, true, num_cu, h);
    CHECK(ok || ip[A.n] == ip[0]);                           // an empty matrix has nothing to sweep
    if (!ok) return;
    const bool narrow = d_hint >= 1 && d_hint <= 64;
    CHECK(h.n_tasks == h.tasks.size() && h.task_rows.size() == (size_t)h.n_tasks * kRW && h.n_entries == h.entries.size());
    CHECK(h.round_tasks == num_cu * env_u32("MGGCN_SPMM_SWEEP_BLOCKS_PER_CU", 4u) * kWavesPerBlock);       // nothing reserved in these runs
    CHECK(((h.prio_bits_wide >> kNumCuPos) & kNumCuMask) == num_cu && ((h.prio_bits_narrow >> kNumCuPos) & kNumCuMask) == num_cu);
    CHECK(narrow ? (h.lpe == 4 || h.lpe == 8 || h.lpe == 12 || h.lpe == 16) && h.run_pad == 64 / h.lpe : h.lpe == 0 && h.run_pad == 2);
    const uint32_t G = h.run_pad;
    const uint32_t batch = !narrow ? 8u : (G == 5u ? 40u : std::max(4u * G, 8u));
    std::vector<double> slot_acc((size_t)h.n_slots * d, 0.0);
    std::vector<double> out((size_t)A.n * d, 0.0);
    uint32_t expect_beg = 0;
    uint64_t real_entries = 0;
    for (uint32_t t = 0; t < h.n_tasks; t++) {
        const SweepTask &tk = h.tasks[t];
        CHECK(tk.beg == expect_beg && tk.end >= tk.beg && (tk.end - tk.beg) % batch == 0 && tk.n_rows <= (uint32_t)kRW);
        expect_beg = tk.end;
        // runs: a flagged entry starts a run of ONE row inside ONE panel that lasts until the next flagged entry (the
        // task's tail padding -- unflagged zero-valued copies of the last entry -- extends the last run, which is how the
        // kernels see it too); runs come in (panel, row) order; every run is a multiple of G entries; inside a pair
        // (G even) the lower column comes first
        uint32_t e = tk.beg;
        long prev_key = -1;
        if (tk.end > tk.beg) CHECK(h.entries[tk.beg].x & kRunFlag);
        while (e < tk.end) {
            const Entry &first = h.entries[e];
            // (the experimental XCD column partition counts its panels from the column base of the task's slice)
            const uint32_t xs_base = xcd_columns ? ((t / kWavesPerBlock) % 8u) * ((A.m + 7u) / 8u) : 0u;
            const uint32_t row_local = (first.x >> kColBits) & (kRW - 1), panel = ((first.x & kColMask) - xs_base) / h.panel_rows;
            CHECK(row_local < tk.n_rows);
            const long key = (long)panel * kRW + row_local;
            CHECK(key > prev_key);
            prev_key = key;
            const uint32_t dst = h.task_rows[(size_t)t * kRW + row_local];
            CHECK((dst & kSlotFlag) ? (dst & ~kSlotFlag) < h.n_slots : dst < A.n);
            double *acc = (dst & kSlotFlag) ? &slot_acc[(size_t)(dst & ~kSlotFlag) * d] : &out[(size_t)dst * d];
            uint32_t q = e;
            do {
                const Entry &en = h.entries[q];
                const uint32_t col = en.x & kColMask;
                CHECK(((en.x >> kColBits) & (kRW - 1)) == row_local && (col - xs_base) / h.panel_rows == panel && col < A.m);
                float val;
                std::memcpy(&val, &en.y, 4);
                real_entries += en.y != 0u;
                for (uint32_t k = 0; k < d; k++) acc[k] += (double)val * (double)B[(size_t)col * d + k];
                q++;
            } while (q < tk.end && !(h.entries[q].x & kRunFlag));
            CHECK((q - e) % G == 0);
            if (G % 2 == 0)
                for (uint32_t p2 = e; p2 + 1 < q; p2 += 2) CHECK((h.entries[p2].x & kColMask) <= (h.entries[p2 + 1].x & kColMask));
            e = q;
        }
    }
    CHECK(expect_beg == h.n_entries);
    // slots -> rows, in slot order
    std::vector<char> is_split(A.n, 0);
    uint32_t slots = 0;
    for (const auto &sr : h.split_rows) {
        CHECK(sr.row < A.n && !is_split[sr.row] && sr.first_slot == slots);
        is_split[sr.row] = 1;
        slots += sr.n_slots;
        for (uint32_t s = 0; s < sr.n_slots; s++)
            for (uint32_t k = 0; k < d; k++) out[(size_t)sr.row * d + k] += slot_acc[(size_t)(sr.first_slot + s) * d + k];
    }
    CHECK(slots == h.n_slots);
    (void)real_entries;
    for (size_t i = 0; i < out.size(); i++) C_accum[i] += out[i];
}

static void test_sweep(const Csr &A, uint32_t d_hint, uint32_t num_cu, uint64_t seed) {
    const uint32_t d = 3;
    // the experimental column partition is taken when the plan builder's own gate says so (wide form, >= 8 panels of columns)
    const bool xcd = env_u32("MGGCN_SPMM_XCD_COLUMNS", 0u) && d_hint == 0 && A.m >= 8u * sweep_panel_rows(0, true);
    lcg g{seed};
    std::vector<float> B((size_t)A.m * d);
    for (auto &x : B) x = 2.f * g.unit() - 1.f;
    const auto want = spmm_ref(A, B, d);
    // whole matrix
    std::vector<double> got((size_t)A.n * d, 0.0);
    check_sweep(A, A.ip.data(), A.ix.data(), A.v.data(), d_hint, num_cu, B, d, got, xcd);
    CHECK(max_rel(got, want) < 1e-9);
    // the same through three column slices (what mggcn_spmm_plan_create_for does for wide B): the slices' products add up
    const uint32_t S = 3, width = (A.m + S - 1) / S;
    SliceBuckets sl = slice_buckets(A.n, S, width, A.ip.data(), A.ix.data(), A.v.data());
    std::vector<double> got2((size_t)A.n * d, 0.0);
    size_t total = 0;
    for (uint32_t k = 0; k < S; k++) {
        CHECK(sl.ips[k].size() == (size_t)A.n + 1 && sl.ixs[k].size() == sl.ips[k][A.n] && sl.vvs[k].size() == sl.ixs[k].size());
        for (uint32_t c : sl.ixs[k]) CHECK(c / width == k);
        total += sl.ixs[k].size();
        if (!sl.ixs[k].empty()) check_sweep(A, sl.ips[k].data(), sl.ixs[k].data(), sl.vvs[k].data(), d_hint, num_cu, B, d, got2, xcd);
    }
    CHECK(total == A.ix.size());
    CHECK(max_rel(got2, want) < 1e-9);
    // and on permuted columns against the permuted B
    std::vector<uint32_t> pi, src_row, pix;
    column_permutation(A.m, pi, src_row);
    std::vector<char> seen(A.m, 0);
    for (uint32_t c = 0; c < A.m; c++) { CHECK(pi[c] < A.m && !seen[pi[c]] && src_row[pi[c]] == c); seen[pi[c]] = 1; }
    permute_indices(A.n, A.ip.data(), A.ix.data(), pi, pix);
    std::vector<float> Bp((size_t)A.m * d);
    for (uint32_t j = 0; j < A.m; j++) for (uint32_t k = 0; k < d; k++) Bp[(size_t)j * d + k] = B[(size_t)src_row[j] * d + k];
    std::vector<double> got3((size_t)A.n * d, 0.0);
    check_sweep(A, A.ip.data(), pix.data(), A.v.data(), d_hint, num_cu, Bp, d, got3, xcd);
    CHECK(max_rel(got3, want) < 1e-9);
    // the oracle's own SpMM (fp64 accumulation, rounded once) agrees with the replay at fp32 resolution
    std::vector<float> Co((size_t)A.n * d, 0.f);
    orc_spmm_csr_f64acc(A.n, A.ip.data(), A.ix.data(), A.v.data(), B.data(), d, Co.data(), d, d, 1.f, 0.f);
    double worst = 0, scale = 0;
    for (size_t i = 0; i < Co.size(); i++) { worst = std::max(worst, std::fabs((double)Co[i] - got[i])); scale = std::max(scale, std::fabs(got[i])); }
    CHECK(worst <= 1e-6 * (scale + 1e-30));
}

static void test_column_stats(const Csr &A) {
    CHECK(columns_in_range(A.n, A.m, A.ip.data(), A.ix.data()));
    if (!A.ix.empty() && A.m > 1) {
        Csr bad = A;
        bad.ix[bad.ix.size() / 2] = A.m;                          // one index out of range is seen
        CHECK(!columns_in_range(bad.n, bad.m, bad.ip.data(), bad.ix.data()));
    }
    const ColumnStats cs = column_stats(A.n, A.m, A.ip.data(), A.ix.data());
    if (A.m < 100 || A.ix.empty()) return;
    // serial recount
    std::vector<uint32_t> cnt(A.m, 0);
    uint64_t near = 0;
    const uint32_t gr = std::max(1u, (A.n + 31) / 32), gc = std::max(1u, (A.m + 31) / 32);
    for (uint32_t r = 0; r < A.n; r++)
        for (uint32_t e = A.ip[r]; e < A.ip[r + 1]; e++) { cnt[A.ix[e]]++; near += (A.ix[e] / gc == r / gr); }
    std::sort(cnt.begin(), cnt.end(), std::greater<uint32_t>());
    uint64_t hot = 0;
    for (size_t k = 0; k < std::max<size_t>(1, A.m / 100); k++) hot += cnt[k];
    CHECK(std::fabs(cs.hot_share - (double)hot / (double)A.ix.size()) < 1e-12);
    CHECK(std::fabs(cs.locality - (double)near / (double)A.ix.size()) < 1e-12);
}

// ---- reserved compute units: a minimum of free slots per launch round, not a cut -------------------------------------------
static void test_reserved_cus() {
    const Csr A = make_csr(3000, 3000, 24, 0, 4242, true);
    auto build = [&](uint32_t num_cu, unsigned reserved) {
        set_reserved_cus(reserved);
        SweepHost h;
        CHECK(sweep_build_host(A.n, A.m, A.ip.data(), A.ix.data(), A.v.data(), 128, true, 0, true, num_cu, h));
        set_reserved_cus(0);
        return std::make_pair(h.round_tasks, h.n_tasks);
    };
    // 256 CUs: 3000 rows are ONE round of 4096 slots with a thousand to spare -- the room is there, the round keeps its size
    auto [r0, t0] = build(256, 0);
    auto [r1, t1] = build(256, 12);
    CHECK(r0 == 4096 && r1 == 4096 && t0 == t1 && t1 <= 4096 - 12 * 16);
    // 8 CUs: 128 slots per round, several full rounds -- every round shrinks to (8 - 2) x 16 slots
    auto [r2, t2] = build(8, 0);
    auto [r3, t3] = build(8, 2);
    CHECK(r2 == 128 && t2 % 128 == 0 && r3 == 96 && t3 % 96 == 0);
    // one round that would be nearly full: 33 CUs = 528 slots, 3000 rows at 6 per task = 500 tasks -> 28 free < 32 asked -> smaller rounds
    auto [r4, t4] = build(33, 0);
    auto [r5, t5] = build(33, 2);
    CHECK(r4 == 528 && t4 <= 528 && 528 - t4 < 32 && r5 == 496 && (t5 <= 496 || t5 % 496 == 0));
}

// ---- host_prep against the oracle ----------------------------------------------------------------------
static void test_host_prep(const Csr &A0, uint32_t P) {
    Csr A = A0;
    std::vector<float> want = A.v;
    orc_csr_normalize(A.n, A.m, A.ip.data(), A.ix.data(), want.data(), 1);
    mggcn_csr_normalize_host(A.n, A.m, A.ip.data(), A.ix.data(), A.v.data(), 1);
    double worst = 0;
    for (size_t i = 0; i < want.size(); i++) worst = std::max(worst, std::fabs((double)A.v[i] - want[i]) / (std::fabs((double)want[i]) + 1e-30));
    CHECK(worst <= 2e-6);                                         // thread-order rounding of the column sums only (exact with unit weights)
    std::vector<float> rw = A0.v, rw2 = A0.v;
    orc_csr_normalize(A.n, A.m, A.ip.data(), A.ix.data(), rw.data(), 0);
    mggcn_csr_normalize_host(A.n, A.m, A.ip.data(), A.ix.data(), rw2.data(), 0);
    CHECK(rw.empty() || std::memcmp(rw.data(), rw2.data(), rw.size() * 4) == 0);   // row sums: same order, same bits (NaN rows compare as bytes)
    std::vector<uint32_t> tip(A.m + 1), tix(A.ix.size()), tip2(A.m + 1), tix2(A.ix.size());
    std::vector<float> tv(A.ix.size()), tv2(A.ix.size());
    orc_csr_transpose(A.n, A.m, A.ip.data(), A.ix.data(), A.v.data(), tip.data(), tix.data(), tv.data());
    mggcn_csr_transpose_host(A.n, A.m, A.ip.data(), A.ix.data(), A.v.data(), tip2.data(), tix2.data(), tv2.data());
    CHECK(tip == tip2 && tix == tix2 && (tv.empty() || std::memcmp(tv.data(), tv2.data(), tv.size() * 4) == 0));
    if (A.n != A.m || A.n % P) return;
    std::vector<uint32_t> q(P + 1);
    for (uint32_t i = 0; i <= P; i++) q[i] = (uint32_t)((uint64_t)i * A.n / P);
    for (uint32_t i = 0; i < P; i++) {
        const uint32_t rb = q[i], re = q[i + 1], rows = re - rb;
        std::vector<uint32_t> bip((size_t)P * (rows + 1)), bip2((size_t)P * (rows + 1));
        orc_block_split_count(A.ip.data(), A.ix.data(), rb, re, q.data(), P, bip.data());
        mggcn_csr_block_split_count_host(A.ip.data(), A.ix.data(), rb, re, q.data(), P, bip2.data());
        CHECK(bip == bip2);
        std::vector<std::vector<uint32_t>> bi(P), bi2(P);
        std::vector<std::vector<float>> bv(P), bv2(P);
        std::vector<uint32_t *> pi(P), pi2(P);
        std::vector<float *> pv(P), pv2(P);
        for (uint32_t j = 0; j < P; j++) {
            const uint32_t c = bip[(size_t)j * (rows + 1) + rows];
            bi[j].resize(c); bi2[j].resize(c); bv[j].resize(c); bv2[j].resize(c);
            pi[j] = bi[j].data(); pi2[j] = bi2[j].data(); pv[j] = bv[j].data(); pv2[j] = bv2[j].data();
        }
        orc_block_split_fill(A.ip.data(), A.ix.data(), A.v.data(), rb, re, q.data(), P, bip.data(), pi.data(), pv.data());
        mggcn_csr_block_split_fill_host(A.ip.data(), A.ix.data(), A.v.data(), rb, re, q.data(), P, bip2.data(), pi2.data(), pv2.data());
        for (uint32_t j = 0; j < P; j++) CHECK(bi[j] == bi2[j] && bv[j] == bv2[j]);
    }
}

// ---- enqueue.hpp ----------------------------------------------------------------------------------------
static void test_enqueue() {
    const std::size_t P = 4, N = 5000;
    std::vector<std::vector<int>> log(P);
    std::vector<std::atomic<std::uint64_t>> seq(P);
    for (auto &s : seq) s.store(0);
    std::atomic<int> inits{0};
    {
        mggcn::enqueue_pool pool(P, [&](std::size_t) { inits++; });
        for (std::size_t k = 0; k < N; k++)
            for (std::size_t j = 0; j < P; j++)
                pool.push(j, [&, j, k] {
                    log[j].push_back((int)k);                     // only thread j touches log[j]
                    if (k % 16 == 0) {                            // a cross-rank rendezvous like the peer-copy transport's `ready`
                        seq[j].store(k / 16 + 1, std::memory_order_release);
                        for (std::size_t i = 0; i < P; i++)
                            while (seq[i].load(std::memory_order_acquire) < k / 16 + 1) std::this_thread::yield();
                    }
                });
        pool.drain();
        for (std::size_t j = 0; j < P; j++) {
            CHECK(log[j].size() == N);
            for (std::size_t k = 0; k < log[j].size(); k++) CHECK(log[j][k] == (int)k);      // program order per rank
        }
        // an exception thrown by a command surfaces at the next drain, once, and the queue keeps working
        pool.push(2, [] { throw std::invalid_argument("shape mismatch"); });
        pool.push(2, [&] { log[2].push_back(-1); });
        bool thrown = false;
        try { pool.drain(); } catch (const std::invalid_argument &) { thrown = true; }
        CHECK(thrown && log[2].back() == -1);
        pool.drain();
        // commands still queued when the pool dies are run, not dropped
        for (std::size_t j = 0; j < P; j++) pool.push(j, [&, j] { log[j].push_back(-2); });
    }
    CHECK(inits.load() == (int)P);
    for (std::size_t j = 0; j < P; j++) CHECK(log[j].back() == -2);
}

int main() {
    setenv("MGGCN_HOST_THREADS", "4", 1);
    setenv("MGGCN_HOST_THREADS_MIN_NNZ", "1", 1);
    orc_set_num_threads(1);
    struct shape { uint32_t n, m, deg; int kind; bool unit; };
    const shape shapes[] = {{600, 600, 20, 0, true}, {1500, 1500, 12, 1, true}, {2048, 2048, 30, 2, false}, {64, 5000, 9, 0, false},
                            {3000, 40, 6, 1, false}, {1, 1, 2, 0, true}, {5, 3, 0, 0, true}, {777, 1, 3, 3, true}, {1200, 1200, 8, 1, false}};
    int idx = 0;
    for (const auto &s : shapes) {
        const Csr A = make_csr(s.n, s.m, s.deg, s.kind, 1000 + idx, s.unit);
        const int before = g_failures;
        test_rowsplit(A, 64);
        test_rowsplit(A, 512);
        test_column_stats(A);
        test_host_prep(A, 4);
        for (const uint32_t d_hint : {0u, 41u, 16u, 64u})
            for (const uint32_t num_cu : {256u, 5u}) {
                // small panels and few rows per round so that these small matrices have several panels, rounds and sliced rows
                setenv("MGGCN_SPMM_PANEL_ROWS", "128", 1);
                setenv("MGGCN_SPMM_PANEL_ROWS_NARROW", "256", 1);
                test_sweep(A, d_hint, num_cu, 77 + idx);
            }
        setenv("MGGCN_SPMM_XCD_COLUMNS", "1", 1);                 // the experimental column partition shares the entry format
        setenv("MGGCN_SPMM_PANEL_ROWS", "64", 1);
        test_sweep(A, 0u, 3u, 99 + idx);
        unsetenv("MGGCN_SPMM_XCD_COLUMNS");
        std::printf("%s: plan builders + host prep on %u x %u, kind %d, %zu non-zeros\n", g_failures == before ? "TEST PASSED" : "TEST FAILED",
                    s.n, s.m, s.kind, A.ix.size());
        idx++;
    }
    {
        const int before = g_failures;
        test_reserved_cus();
        std::printf("%s: reserved compute units\n", g_failures == before ? "TEST PASSED" : "TEST FAILED");
    }
    {
        const int before = g_failures;
        test_enqueue();
        std::printf("%s: enqueue queues\n", g_failures == before ? "TEST PASSED" : "TEST FAILED");
    }
    std::printf(g_failures ? "FAILED (%d)\n" : "ALL PASSED\n", g_failures);
    return g_failures ? 1 : 0;
}
