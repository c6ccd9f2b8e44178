// svils_report.hip -- see svils_report.h.  HBM-bound byte moving: 8-byte words, coalesced, grid-stride.
#include "svils_report.h"

namespace svils {

namespace {
__global__ __launch_bounds__(256) void k_report_pack(const unsigned char *ctrl, uint32_t ctrl_bytes, const double *rows,
                                                     uint32_t rows_cap, uint32_t row_first, uint32_t row_count,
                                                     const uint64_t *member, size_t nwords, unsigned char *out,
                                                     size_t off_rows, size_t off_member) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  if (blockIdx.x == 0) {
    if (threadIdx.x < ctrl_bytes) out[threadIdx.x] = ctrl[threadIdx.x];
    double *r = (double *)(out + off_rows);
    for (uint32_t i = threadIdx.x; i < row_count * 10u; i += blockDim.x) {
      const uint32_t slot = (row_first + i / 10u) % rows_cap;   // the ring wraps at rows_cap
      r[i] = rows[(size_t)slot * 10u + i % 10u];
    }
  }
  uint64_t *m = (uint64_t *)(out + off_member);
  for (size_t i = tid; i < nwords; i += nthreads) m[i] = member[i];
}
}  // namespace

void launch_report_pack(const void *ctrl, size_t ctrl_bytes, const double *rows, uint32_t rows_cap, uint32_t row_first,
                        uint32_t row_count, const uint64_t *member, size_t nwords, unsigned char *out,
                        const ReportLayout &lay, hipStream_t s) {
  // enough blocks to stream a large bitmask (n = 1e6, k = 512: 64 MB) at HBM rate, one block for the small cases
  const size_t want = (nwords + 256 * 8 - 1) / (256 * 8);
  const uint32_t blocks = (uint32_t)(want < 1 ? 1 : want > 2048 ? 2048 : want);
  hipLaunchKernelGGL(k_report_pack, dim3(blocks), dim3(256), 0, s, (const unsigned char *)ctrl, (uint32_t)ctrl_bytes, rows, rows_cap,
                     row_first, row_count, member, nwords, out, lay.off_rows, lay.off_member);
}

}  // namespace svils
