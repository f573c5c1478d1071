// Ad-hoc measurement: cost of a NEIGHBOUR synchronisation between the blocks of one persistent kernel (what a
// barotropic loop kept inside one launch would pay per sub-step instead of a kernel boundary).  256 blocks x 384
// threads on a 16x16 periodic block grid; per iteration a block writes a 2 KB rim payload, publishes a flag
// (release), waits for the flags of its 8 neighbours (acquire) and reads their payloads.  Spins are capped: the
// probe cannot hang.   hipcc --offload-arch=gfx950 -O3 nbrsync_probe.hip -o nbrsync_probe && ./nbrsync_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#define NBX 16
#define NBY 16
#define PAY 256   // doubles per block
template <int SCOPE>
__global__ void __launch_bounds__(384) k_nbr(double *pay, unsigned *flag, int nsync, unsigned *err, double *out) {
  const int b = blockIdx.x, bx = b % NBX, by = b / NBX, t = threadIdx.x;
  __shared__ int nb[8];
  if (t < 8) {
    const int dx[8] = {1, -1, 0, 0, 1, 1, -1, -1}, dy[8] = {0, 0, 1, -1, 1, -1, 1, -1};
    nb[t] = (bx + dx[t] + NBX) % NBX + ((by + dy[t] + NBY) % NBY) * NBX;
  }
  __syncthreads();
  double acc = 0.0;
  for (int s = 1; s <= nsync; s++) {
    double *mine = pay + ((size_t)(s & 1) * gridDim.x + b) * PAY;     // double-buffered by parity
    if (t < PAY) mine[t] = (double)(b * 1000 + s) + acc * 1e-30;
    __syncthreads();
    if (t == 0) __hip_atomic_store(flag + b * 16, (unsigned)s, __ATOMIC_RELEASE, SCOPE);
    if (t < 8) {
      unsigned spins = 0;
      while (__hip_atomic_load(flag + nb[t] * 16, __ATOMIC_ACQUIRE, SCOPE) < (unsigned)s) {
        if (++spins > 4000000u) { *err = 1; break; }
      }
    }
    __syncthreads();
    if (t < PAY) {
      const double *theirs = pay + ((size_t)(s & 1) * gridDim.x + nb[t & 7]) * PAY;
      const double v = __builtin_nontemporal_load(theirs + t);
      if (v != (double)(nb[t & 7] * 1000 + s) && v - (double)(nb[t & 7] * 1000 + s) > 1e-6) *err = 2;
      acc += v;
    }
  }
  if (t < PAY) out[b * PAY + t] = acc;
}
int main() {
  const int nb = NBX * NBY, nt = 384;
  double *pay, *out; unsigned *flag, *err;
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mem = 0; mem < 2; mem++) {
    if (mem == 0) { hipMalloc(&pay, 2 * nb * PAY * 8); hipMalloc(&flag, nb * 64); }
    else {
      if (hipExtMallocWithFlags((void **)&pay, 2 * nb * PAY * 8, hipDeviceMallocUncached) != hipSuccess ||
          hipExtMallocWithFlags((void **)&flag, nb * 64, hipDeviceMallocUncached) != hipSuccess) { printf("no uncached memory\n"); break; }
    }
    hipMalloc(&out, nb * PAY * 8); hipMalloc(&err, 4);
    for (int scope = 0; scope < 2; scope++) {
      for (int nsync : {1, 201}) {
        float best = 1e9f; unsigned herr = 0;
        for (int rep = 0; rep < 4; rep++) {
          hipMemset(flag, 0, nb * 64); hipMemset(err, 0, 4); hipDeviceSynchronize();
          hipEventRecord(e0, 0);
          if (scope == 0) hipLaunchKernelGGL(k_nbr<__HIP_MEMORY_SCOPE_AGENT>, dim3(nb), dim3(nt), 0, 0, pay, flag, nsync, err, out);
          else hipLaunchKernelGGL(k_nbr<__HIP_MEMORY_SCOPE_SYSTEM>, dim3(nb), dim3(nt), 0, 0, pay, flag, nsync, err, out);
          hipEventRecord(e1, 0); hipEventSynchronize(e1);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (ms < best) best = ms;
          unsigned e; hipMemcpy(&e, err, 4, hipMemcpyDeviceToHost); herr |= e;
        }
        printf("%s memory, %s scope, nsync=%3d: %.2f us total  err=%u\n", mem ? "uncached" : "normal  ", scope ? "system" : "agent ", nsync, best * 1e3, herr);
      }
    }
    hipFree(pay); hipFree(flag); hipFree(out); hipFree(err);
  }
  return 0;
}
