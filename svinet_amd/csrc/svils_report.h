// svils_report.h -- device side of the pipelined reports (svils_report_enqueue): ONE launch packs what the reference's
// report block reads (src/linksampling.cc:777-786) -- the loop's control block, the likelihood rows recorded since the
// previous report and the community bitmask of the last tagging sweep -- into a staging slot, from where a copy
// stream takes it to pinned host memory while the compute stream goes on sweeping.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "svils_internal.h"

namespace svils {

struct ReportLayout {
  size_t off_rows;     // [max_rows][10] doubles
  size_t off_trows;    // [max_rows][10] doubles: the test rows of the same reports (svils_set_test)
  size_t off_member;   // [n][kw] uint64 (absent when the report carries no communities)
  size_t bytes;
};

// test_likelihood (src/linksampling.cc:1147-1182) behind a sweep's tail: d.vpairs / d.uval / d.nv name the TEST set and
// k_validation has just filled uval.  Writes the row under the number of the validation row this sweep recorded --
// unless the sweep made no report, or the stop rule ended the run in it (the reference exits before test_likelihood).
void launch_test_row(const DeviceState &d, const Params &p, double *ring, uint32_t cap, hipStream_t s);

// out = slot base.  rows [row_first, row_first + row_count) of the ring (capacity rows_cap) in order; nwords = 0: no bitmask.
void launch_report_pack(const void *ctrl, size_t ctrl_bytes, const double *rows, const double *trows, uint32_t rows_cap,
                        uint32_t row_first, uint32_t row_count, const uint64_t *member, size_t nwords, unsigned char *out,
                        const ReportLayout &lay, hipStream_t s);

}  // namespace svils
