// micro-benchmark: what a launch costs as a function of the STATIC LDS of its workgroups.
// Found through k_tail: the same code with a 16 KB larger LDS array (21 -> 37 KB per block, 246 blocks of 256 threads) took
// 10-20 us longer per launch (profiles/r07t_ab_tail_parallel_loads.txt).  This maps the curve with a kernel that does nothing
// else: every thread writes one LDS word, a barrier, thread 0 stores one word -- launched back to back NL times on one stream,
// alone and alternating with a small-LDS kernel (what a sweep's launches do), eager and as a hipGraph.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/lds_launch tools/ubench/lds_launch.hip && /tmp/lds_launch
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int BYTES, int NT>
__global__ __launch_bounds__(NT) void k(uint32_t *out) {
  __shared__ uint32_t lds[BYTES / 4];
  lds[(threadIdx.x * 37u) % (BYTES / 4)] = threadIdx.x;
  __syncthreads();
  if (threadIdx.x == 0) out[blockIdx.x] = lds[(blockIdx.x * 101u) % (BYTES / 4)];
}

template <int BYTES, int NT>
static int run(uint32_t *out, hipStream_t s, int grid, int nl) {
  hipEvent_t e0, e1;
  CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
  float ms_alone = 0, ms_alt = 0, ms_graph = 0;
  for (int i = 0; i < 20; ++i) hipLaunchKernelGGL((k<BYTES, NT>), dim3(grid), dim3(NT), 0, s, out);
  CHK(hipEventRecord(e0, s));
  for (int i = 0; i < nl; ++i) hipLaunchKernelGGL((k<BYTES, NT>), dim3(grid), dim3(NT), 0, s, out);
  CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_alone, e0, e1));
  // alternating with a 2 KB kernel of the same shape: per PAIR
  CHK(hipEventRecord(e0, s));
  for (int i = 0; i < nl; ++i) {
    hipLaunchKernelGGL((k<BYTES, NT>), dim3(grid), dim3(NT), 0, s, out);
    hipLaunchKernelGGL((k<2048, 256>), dim3(grid), dim3(256), 0, s, out);
  }
  CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_alt, e0, e1));
  // the alternating pair as a graph of 64 pairs
  hipGraph_t g; hipGraphExec_t ge;
  CHK(hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal));
  for (int i = 0; i < 64; ++i) {
    hipLaunchKernelGGL((k<BYTES, NT>), dim3(grid), dim3(NT), 0, s, out);
    hipLaunchKernelGGL((k<2048, 256>), dim3(grid), dim3(256), 0, s, out);
  }
  CHK(hipStreamEndCapture(s, &g)); CHK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  CHK(hipGraphLaunch(ge, s)); CHK(hipStreamSynchronize(s));
  CHK(hipEventRecord(e0, s));
  for (int i = 0; i < nl / 64; ++i) CHK(hipGraphLaunch(ge, s));
  CHK(hipEventRecord(e1, s)); CHK(hipEventSynchronize(e1)); CHK(hipEventElapsedTime(&ms_graph, e0, e1));
  printf("lds %6d B  threads %4d  grid %5d : alone %6.2f us/launch | alternating with a 2 KB kernel %6.2f us/pair eager, %6.2f us/pair in a graph\n",
         BYTES, NT, grid, ms_alone * 1e3 / nl, ms_alt * 1e3 / nl, ms_graph * 1e3 / (nl / 64 * 64));
  CHK(hipGraphExecDestroy(ge)); CHK(hipGraphDestroy(g));
  return 0;
}

int main(int argc, char **argv) {
  const int nl = argc > 1 ? atoi(argv[1]) : 2048;
  uint32_t *out;
  CHK(hipMalloc(&out, 1 << 20));
  hipStream_t s;
  CHK(hipStreamCreate(&s));
  for (int grid : {246, 1024}) {
    run<2048, 256>(out, s, grid, nl);
    run<16384, 256>(out, s, grid, nl);
    run<24576, 256>(out, s, grid, nl);
    run<31744, 256>(out, s, grid, nl);
    run<32768, 256>(out, s, grid, nl);
    run<33792, 256>(out, s, grid, nl);
    run<40960, 256>(out, s, grid, nl);
    run<65536, 256>(out, s, grid, nl);
    run<65536, 1024>(out, s, grid, nl);
    run<102400, 1024>(out, s, grid, nl);
    run<131072, 1024>(out, s, grid, nl);
    run<163840, 1024>(out, s, grid, nl);
  }
  return 0;
}
