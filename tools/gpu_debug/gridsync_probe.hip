// Ad-hoc measurement: cost of a grid-wide barrier (cooperative launch) on this GPU, 256 blocks x 384 threads.
// hipcc --offload-arch=gfx950 -O3 gridsync_probe.hip -o gridsync_probe && ./gridsync_probe
#include <hip/hip_runtime.h>
#include <hip/hip_cooperative_groups.h>
#include <cstdio>
namespace cg = cooperative_groups;
__global__ void __launch_bounds__(384) k_cg(double *a, int n, int nsync) {
  cg::grid_group g = cg::this_grid();
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  double v = a[i % n];
  for (int s = 0; s < nsync; s++) {
    a[i % n] = v + 1.0;
    g.sync();
    v = a[(i + 4099) % n];
  }
  a[i % n] = v;
}
// hand-written barrier: one agent-scope atomic per block, spin on the counter
__global__ void __launch_bounds__(384) k_own(double *a, int n, int nsync, unsigned *ctr) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const unsigned nb = gridDim.x;
  double v = a[i % n];
  for (int s = 0; s < nsync; s++) {
    a[i % n] = v + 1.0;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      __hip_atomic_fetch_add(ctr, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = nb * (unsigned)(s + 1);
      while (__hip_atomic_load(ctr, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) __builtin_amdgcn_s_sleep(1);
      __threadfence();
    }
    __syncthreads();
    v = a[(i + 4099) % n];
  }
  a[i % n] = v;
}
int main() {
  const int nb = 256, nt = 384, n = nb * nt;
  double *a; unsigned *ctr;
  hipMalloc(&a, n * sizeof(double)); hipMemset(a, 0, n * sizeof(double));
  hipMalloc(&ctr, 4);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int nsync : {1, 101}) {
    for (int rep = 0; rep < 3; rep++) {
      int nn = n, ns = nsync;
      void *args[] = {&a, &nn, &ns};
      hipEventRecord(e0, 0);
      hipError_t r = hipLaunchCooperativeKernel((void *)k_cg, dim3(nb), dim3(nt), args, 0, 0);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("cg::grid.sync  nsync=%3d  rc=%d  %.2f us total\n", nsync, (int)r, ms * 1e3);
    }
    for (int rep = 0; rep < 3; rep++) {
      hipMemset(ctr, 0, 4); hipDeviceSynchronize();
      hipEventRecord(e0, 0);
      hipLaunchKernelGGL(k_own, dim3(nb), dim3(nt), 0, 0, a, n, nsync, ctr);
      hipEventRecord(e1, 0); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      if (rep == 2) printf("own barrier    nsync=%3d        %.2f us total\n", nsync, ms * 1e3);
    }
  }
  return 0;
}
