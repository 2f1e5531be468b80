// Diagnostics only (tools/bench_launch_floor.py): one empty kernel behind a C entry point.  Built on demand into tools/_launch_probe.so;
// not part of the shipped C ABI (include/dynllava.h).
#include <hip/hip_runtime.h>
__global__ void launch_probe_kernel() {}
extern "C" int launch_probe(int grid, int block, void* stream) {
  if (grid < 1 || block < 1 || block > 1024) return -1;
  hipLaunchKernelGGL(launch_probe_kernel, dim3(grid), dim3(block), 0, (hipStream_t)stream);
  return hipGetLastError() == hipSuccess ? 0 : -2;
}
