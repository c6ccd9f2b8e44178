// svils_report.h -- device side of the pipelined reports (svils_report_enqueue): ONE launch packs what the reference's
// report block reads (src/linksampling.cc:777-786) -- the loop's control block, the likelihood rows recorded since the
// previous report and the community bitmask of the last tagging sweep -- into a staging slot, from where a copy
// stream takes it to pinned host memory while the compute stream goes on sweeping.
#pragma once
#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

namespace svils {

struct ReportLayout {
  size_t off_rows;     // [max_rows][10] doubles
  size_t off_member;   // [n][kw] uint64 (absent when the report carries no communities)
  size_t bytes;
};

// out = slot base.  rows [row_first, row_first + row_count) of the ring (capacity rows_cap) in order; nwords = 0: no bitmask.
void launch_report_pack(const void *ctrl, size_t ctrl_bytes, const double *rows, uint32_t rows_cap, uint32_t row_first,
                        uint32_t row_count, const uint64_t *member, size_t nwords, unsigned char *out,
                        const ReportLayout &lay, hipStream_t s);

}  // namespace svils
