// What a kernel boundary costs on one in-order stream (MI355X, this ROCm): chains of N dependent launches of (a) an empty kernel, (b) a kernel of
// `wgs` workgroups that each do one dependent global-memory round trip, timed with HIP events around the whole chain and from the host
// (launch of the first kernel to the end of hipStreamSynchronize).  build: hipcc --offload-arch=gfx950 -O3 -o launch_chain launch_chain.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#include <algorithm>
__global__ void k_empty() {}
__global__ void k_trip(const int* __restrict__ in, int* __restrict__ out, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[(in[i] + i) % n] + 1;   // two dependent loads
}
int main() {
  const int n = 1 << 20;
  int *a, *b;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int wgs : {0, 1, 64, 750, 3000}) {
    for (int N : {1, 2, 4, 8}) {
      std::vector<float> ev; std::vector<double> host;
      for (int rep = 0; rep < 220; ++rep) {
        hipStreamSynchronize(s);
        auto t0 = std::chrono::steady_clock::now();
        hipEventRecord(e0, s);
        for (int k = 0; k < N; ++k) {
          if (wgs == 0) hipLaunchKernelGGL(k_empty, dim3(1), dim3(64), 0, s);
          else hipLaunchKernelGGL(k_trip, dim3(wgs), dim3(256), 0, s, (k & 1) ? b : a, (k & 1) ? a : b, wgs * 256);
        }
        hipEventRecord(e1, s);
        hipStreamSynchronize(s);
        auto t1 = std::chrono::steady_clock::now();
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (rep >= 20) { ev.push_back(ms * 1e3f); host.push_back(std::chrono::duration<double>(t1 - t0).count() * 1e6); }
      }
      std::sort(ev.begin(), ev.end()); std::sort(host.begin(), host.end());
      printf("workgroups %5d  chain of %d: events %.1f us (%.1f per kernel)   host launch-to-sync %.1f us\n", wgs, N, ev[ev.size() / 2], ev[ev.size() / 2] / N, host[host.size() / 2]);
    }
  }
  return 0;
}
