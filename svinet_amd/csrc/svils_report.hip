// svils_report.hip -- see svils_report.h.  HBM-bound byte moving: 8-byte words, coalesced, grid-stride.
#include "svils_report.h"

namespace svils {

namespace {
__global__ __launch_bounds__(256) void k_report_pack(const unsigned char *ctrl, uint32_t ctrl_bytes, const double *rows,
                                                     const double *trows, uint32_t rows_cap, uint32_t row_first,
                                                     uint32_t row_count, const uint64_t *member, size_t nwords,
                                                     unsigned char *out, size_t off_rows, size_t off_trows, size_t off_member) {
  const size_t tid = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t nthreads = (size_t)gridDim.x * blockDim.x;
  if (blockIdx.x == 0) {
    if (threadIdx.x < ctrl_bytes) out[threadIdx.x] = ctrl[threadIdx.x];
    double *r = (double *)(out + off_rows);
    for (uint32_t i = threadIdx.x; i < row_count * 10u; i += blockDim.x) {
      const uint32_t slot = (row_first + i / 10u) % rows_cap;   // the ring wraps at rows_cap
      r[i] = rows[(size_t)slot * 10u + i % 10u];
    }
    if (trows) {
      double *t = (double *)(out + off_trows);
      for (uint32_t i = threadIdx.x; i < row_count * 10u; i += blockDim.x) {
        const uint32_t slot = (row_first + i / 10u) % rows_cap;
        t[i] = trows[(size_t)slot * 10u + i % 10u];
      }
    }
  }
  uint64_t *m = (uint64_t *)(out + off_member);
  for (size_t i = tid; i < nwords; i += nthreads) m[i] = member[i];
}
// the reduction of k_row_only (svils_device.hip) over the test pairs, written into the ring
__global__ __launch_bounds__(256) void k_test_row(DeviceState d, Params prm, double *ring, uint32_t cap) {
  const DevCtrl *ctrl = d.ctrl;
  if (ctrl->stopped) return;                            // the run ended in this sweep: exit(0) comes before test_likelihood
  const uint32_t it = ctrl->iter - 1u;                  // the tail has already counted the sweep
  if (it % prm.reportfreq != 0u || ctrl->rows == 0u) return;
  __shared__ double red[2][256];
  __shared__ unsigned int cntz[256];
  double sz = 0.0, so = 0.0;
  unsigned int kz = 0;
  for (uint32_t i = threadIdx.x; i < d.nv; i += blockDim.x) {
    const double u = d.uval[i];
    if (d.vpairs[3 * (size_t)i + 2]) so += u; else { sz += u; kz++; }
  }
  red[0][threadIdx.x] = sz; red[1][threadIdx.x] = so; cntz[threadIdx.x] = kz;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) {
      red[0][threadIdx.x] += red[0][threadIdx.x + o];
      red[1][threadIdx.x] += red[1][threadIdx.x + o];
      cntz[threadIdx.x] += cntz[threadIdx.x + o];
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    double *row = ring + (size_t)((ctrl->rows - 1u) % cap) * 10u;
    const double szeros = red[0][0], sones = red[1][0];
    const uint32_t kzeros = cntz[0], kones = d.nv - cntz[0];
    const double mean0 = szeros / kzeros, mean1 = sones / kones;
    row[0] = (double)it; row[1] = (szeros + sones) / d.nv; row[2] = (double)d.nv;
    row[3] = mean0; row[4] = (double)kzeros; row[5] = mean1; row[6] = (double)kones;
    row[7] = prm.zeros_prob * mean0; row[8] = prm.ones_prob * mean1;
    row[9] = prm.zeros_prob * mean0 + prm.ones_prob * mean1;
  }
}
}  // namespace

void launch_test_row(const DeviceState &d, const Params &p, double *ring, uint32_t cap, hipStream_t s) {
  hipLaunchKernelGGL(k_test_row, dim3(1), dim3(256), 0, s, d, p, ring, cap);
}

void launch_report_pack(const void *ctrl, size_t ctrl_bytes, const double *rows, const double *trows, uint32_t rows_cap,
                        uint32_t row_first, uint32_t row_count, const uint64_t *member, size_t nwords, unsigned char *out,
                        const ReportLayout &lay, hipStream_t s) {
  // enough blocks to stream a large bitmask (n = 1e6, k = 512: 64 MB) at HBM rate, one block for the small cases
  const size_t want = (nwords + 256 * 8 - 1) / (256 * 8);
  const uint32_t blocks = (uint32_t)(want < 1 ? 1 : want > 2048 ? 2048 : want);
  hipLaunchKernelGGL(k_report_pack, dim3(blocks), dim3(256), 0, s, (const unsigned char *)ctrl, (uint32_t)ctrl_bytes, rows, trows,
                     rows_cap, row_first, row_count, member, nwords, out, lay.off_rows, lay.off_trows, lay.off_member);
}

}  // namespace svils
