// What a kernel pays for WRITING (MI355X): a chain of dependent launches of a kernel whose 750 workgroups each do two dependent loads, and in which a few waves
// additionally write -- plain store, device-scope atomic (with and without using the result), non-temporal store -- either at once or after a chain of
// dependent loads (a late writer = the scoring kernel's surviving waves).  Per-kernel time from HIP events around chains of 8.
// build: hipcc --offload-arch=gfx950 -O3 -o write_tail write_tail.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>
__global__ void k(const int* __restrict__ in, int* __restrict__ out, unsigned* __restrict__ cnt, int n, int mode, int writers, int late) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  int v = in[(in[i % n] + i) % n];
  if ((threadIdx.x & 63) == 0 && (int)blockIdx.x < writers) {
    for (int k2 = 0; k2 < late; ++k2) v = in[(v + i + k2) % n];     // dependent loads in front of the write
    if (mode == 1) out[i] = v + 1;
    else if (mode == 2) atomicAdd(cnt + 64 * (blockIdx.x & 7), 1u);
    else if (mode == 3) { const unsigned b = atomicAdd(cnt + 64 * (blockIdx.x & 7), 1u); out[(i + b) % n] = v; }
    else if (mode == 4) __builtin_nontemporal_store(v + 1, out + i);
  }
  if (v == 0x7fffffff) out[0] = v;
}
int main() {
  const int n = 1 << 20;
  int *a, *b; unsigned* c;
  hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&c, 4096 * 4);
  hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4); hipMemset(c, 0, 4096 * 4);
  hipStream_t s; hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const char* names[] = {"no write", "plain store", "atomic, result unused", "atomic + dependent store", "non-temporal store"};
  for (int late : {0, 8})
    for (int writers : {1, 30})
      for (int mode = 0; mode < 5; ++mode) {
        std::vector<float> ev;
        for (int rep = 0; rep < 120; ++rep) {
          hipStreamSynchronize(s);
          hipEventRecord(e0, s);
          for (int k2 = 0; k2 < 8; ++k2) hipLaunchKernelGGL(k, dim3(750), dim3(256), 0, s, a, b, c, 750 * 256, mode, writers, late);
          hipEventRecord(e1, s);
          hipStreamSynchronize(s);
          float ms; hipEventElapsedTime(&ms, e0, e1);
          if (rep >= 20) ev.push_back(ms * 1e3f / 8);
        }
        std::sort(ev.begin(), ev.end());
        printf("late loads %d  writers %2d  %-26s %.2f us per kernel\n", late, writers, names[mode], ev[ev.size() / 2]);
      }
  return 0;
}
