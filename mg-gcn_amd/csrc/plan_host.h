// plan_host.h -- the HOST passes of the SpMM plan builders, free of HIP.
//
// mggcn_spmm_plan_create_for (spmm.hip) = these passes + device allocation and upload.  They live in
// their own translation unit (plan_host.cpp) so that the same code can be compiled with plain g++
// under AddressSanitizer / UndefinedBehaviorSanitizer / ThreadSanitizer and driven on the CPU
// (tests/native/plan_host_test.cpp, `make -C tests/native sanitize`, tests/test_sanitizers.py): the passes are threaded
// (per-thread counters, per-row offsets, one task range per thread) and build the packed entry
// streams every sweep kernel trusts blindly.  Not part of the ABI.
//
// What the reference has in this place: nothing -- cusparseSpMM's workspace query
// (src/cuda_utils.hpp:94-102) hides whatever preprocessing cuSPARSE does.
#pragma once

#include <cstddef>
#include <cstdint>
#include <vector>

namespace mggcn_plan {

// ---- row-split form (spmm.hip) ---------------------------------------------------------------
constexpr uint32_t kNoSlot = 0xFFFFFFFFu;

struct SpmmItem {
    uint32_t row;   // output row
    uint32_t beg;   // first non-zero (absolute offset into indices/values)
    uint32_t end;   // one past the last
    uint32_t slot;  // kNoSlot: write C directly; else: partial-sum slot
};

struct SplitRow {
    uint32_t row;
    uint32_t first_slot;
    uint32_t n_slots;
    uint32_t pad;
};

struct RowSplitHost {
    std::vector<SpmmItem> items;          // longest first; ties in row order
    std::vector<SplitRow> split_rows;
    uint32_t n_slots = 0;
};
// heavy rows (longer than 1.5 x split) are cut into contiguous slices with partial-sum slots
RowSplitHost rowsplit_build(uint32_t n_rows, const uint32_t *indptr, uint32_t split);

// ---- sweep form (spmm_sweep.hip) -------------------------------------------------------------
constexpr int kRW = 16;                 // output rows per wave (4 row bits in the packed entry)
constexpr uint32_t kColBits = 27;       // columns < 134 M (papers100M: 111 M)
constexpr uint32_t kColMask = (1u << kColBits) - 1;
constexpr uint32_t kSlotFlag = 0x80000000u;    // task-row table: the row is a partial-sum slot
constexpr uint32_t kRunFlag = 0x80000000u;     // entry bit 31: first entry of a (panel,row) run; bits 30..27: row
constexpr int kWavesPerBlock = 4;
// internal launch flags (never part of the ABI's flags): bit 8 = rotate the wave priority, bits 12..15 = log2 of the
// rotation period in entries, bits 16..27 = compute units of the device (hardware wave slot of a block = block / CUs)
constexpr uint32_t kFlagPrioRotate = 0x100u;
constexpr uint32_t kPrioShiftPos = 12;
constexpr uint32_t kNumCuPos = 16, kNumCuMask = 0xFFFu;

struct SweepTask {
    uint32_t beg, end;   // entry range
    uint32_t n_rows;     // rows in use (<= kRW)
    uint32_t pad;
};

struct SweepSplitRow {
    uint32_t row, first_slot, n_slots, pad;
};

struct Entry {           // layout of the device's uint2: {run:1 | row:4 | column:27, value bits}
    uint32_t x, y;
};

// everything sweep_plan_build decides and assembles before the first device call
struct SweepHost {
    uint32_t n_rows = 0, n_cols = 0, max_d = 0;
    uint32_t n_tasks = 0, round_tasks = 0, n_slots = 0;
    uint32_t run_pad = 2;          // every (panel,row) run is a multiple of this many entries
    uint32_t lpe = 0;              // narrow form: lanes per entry of the gather kernel (run_pad = 64 / lpe); 0 = wide form
    uint32_t panel_rows = 0;
    uint32_t num_cu = 0;
    uint64_t n_entries = 0;        // padded entry stream length
    // tuning knobs, read from the environment ONCE when the plan is built (never on the launch path)
    uint32_t prio_bits_wide = 0, prio_bits_narrow = 0;   // kFlagPrioRotate | shift << kPrioShiftPos | CUs << kNumCuPos
    uint32_t tasks_per_wave = 1;
    bool allow_quad = true, allow_vec4 = true, fast_pairs = true;
    std::vector<SweepTask> tasks;
    std::vector<Entry> entries;
    std::vector<uint32_t> task_rows;           // n_tasks x kRW: output row or kSlotFlag | slot
    std::vector<SweepSplitRow> split_rows;
};

uint32_t sweep_lanes_per_entry(uint32_t d_hint);                    // 4 / 8 / 12 / 16 float4 lanes per gathered row
uint32_t sweep_panel_rows(uint32_t d_hint, bool hot_columns);       // rows of B per column panel
// false: the matrix is not worth sweeping (tiny, unless force) or does not fit the packed entry.  num_cu = compute
// units of the device the plan is for (a launch round = the resident set: num_cu x blocks per CU x 4 waves).
bool sweep_build_host(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices, const float *values,
                      uint32_t max_d, bool force, uint32_t d_hint, bool hot_columns, uint32_t num_cu, SweepHost &out);

// ---- passes of mggcn_spmm_plan_create_for over the whole matrix ----------------------------------
// every column index < n_cols?  (one threaded read-only pass on behalf of the calling thread, which reports)
bool columns_in_range(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices);

struct ColumnStats {
    double hot_share = 0.0;     // share of the non-zeros in the 1 % most popular columns
    double locality = 0.0;      // share of the non-zeros whose column lies in the same 1/32 of the index space as their row
    bool hot_columns = true;
};
ColumnStats column_stats(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr, const uint32_t *indices);

// fixed pseudo-random relabelling of the columns: pi[c] = new label of column c, src_row[pi[c]] = c
void column_permutation(uint32_t n_cols, std::vector<uint32_t> &pi, std::vector<uint32_t> &src_row);
// out[e] = pi[indices[e]], indexed like `indices` (the offset indptr[0] kept)
void permute_indices(uint32_t n_rows, const uint32_t *indptr, const uint32_t *indices, const std::vector<uint32_t> &pi,
                     std::vector<uint32_t> &out);

// the non-zeros bucketed by column slice of `width` columns: slice k = CSR (ips[k], ixs[k], vvs[k]) over all rows,
// row order and the order inside a row kept
struct SliceBuckets {
    std::vector<std::vector<uint32_t>> ips, ixs;
    std::vector<std::vector<float>> vvs;
};
SliceBuckets slice_buckets(uint32_t n_rows, uint32_t S, uint32_t width, const uint32_t *indptr, const uint32_t *indices,
                           const float *values);

// ---- shared knobs ------------------------------------------------------------------------------
uint32_t env_u32(const char *name, uint32_t dflt);
// threads of a host pass: MGGCN_HOST_THREADS, else the cores of the machine divided by the number of plan builders
// running side by side (set_concurrent_builders; csr_matrix::prebuild_plans runs up to four), capped at `cap`
unsigned host_threads(unsigned cap);
void set_concurrent_builders(unsigned n);
// compute units' worth of wave slots the launch rounds of plans built FROM NOW ON leave free (see sweep_build_host)
void set_reserved_cus(unsigned n);

}  // namespace mggcn_plan
