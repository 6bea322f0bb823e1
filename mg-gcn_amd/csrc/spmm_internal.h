// spmm_internal.h -- shared between spmm.hip (C ABI + row-split kernels) and
// spmm_sweep.hip (column-panel sweep kernels).  Not part of the ABI.
#pragma once

#include "common.h"

struct SweepPlan;   // opaque to spmm.hip

// Builds the panel-swept entry stream for one CSR matrix (host arrays) and uploads it.
// Returns nullptr when the matrix is not worth sweeping (tiny).
// force: build even below the size threshold (later column slices of a sliced matrix)
SweepPlan *sweep_plan_build(uint32_t n_rows, uint32_t n_cols, const uint32_t *indptr,
                            const uint32_t *indices, const float *values, uint32_t max_d, bool force = false,
                            uint32_t d_hint = 0, bool hot_columns = true);
// d_hint in 1..64 builds the narrow ("quad") form: runs padded to four entries, wider panels (plan_host.h:
// sweep_panel_rows, sweep_lanes_per_entry)
// narrow form only: true when B has to be re-pitched to 16-byte rows before sweep_launch
bool sweep_wants_repack(const SweepPlan *p, uint32_t d, size_t ldb, const void *B);
// copy of B at pitch dp (multiple of 4 floats); src_row != nullptr: row r of the copy = row src_row[r] of B
void sweep_repack(hipStream_t st, const float *B, size_t ldb, uint32_t n_cols, uint32_t d, float *out, uint32_t dp,
                  const uint32_t *src_row = nullptr);
void sweep_plan_destroy(SweepPlan *p);
size_t sweep_plan_bytes(const SweepPlan *p);
int sweep_plan_describe(const SweepPlan *p, char *out, size_t cap);   // one line: tasks, rounds, panel rows, lpe, entries
uint32_t sweep_plan_tasks(const SweepPlan *p);
uint32_t sweep_plan_split_rows(const SweepPlan *p);
uint32_t sweep_plan_read_stamps(const SweepPlan *p, unsigned long long *host_out, uint32_t capacity_tasks);
uint32_t sweep_plan_launches(const SweepPlan *p, uint32_t d);   // kernel launches of one sweep_launch at width d
// true if this (d, alignment) combination is served by the sweep kernels
bool sweep_supports(const SweepPlan *p, uint32_t d, size_t ldb, size_t ldc, const void *B, const void *C);
void sweep_launch(hipStream_t st, const SweepPlan *p, const float *B, size_t ldb, float *C, size_t ldc,
                  uint32_t d, float alpha, float beta, uint32_t flags, float slope);
