// What does a device-scope atomicAdd with return cost when every wave of a full grid hits the SAME address, and how does it scale
// with the number of counters and their spacing?  (The scoring kernel appends candidates through one counter: on rendered banks
// 1 000 - 30 000 candidates per frame pass the coarse level and the kernel's time follows their number, ~5 ns per append.)
//   hipcc -O3 --offload-arch=gfx950 atomic_append.hip -o atomic_append && ./atomic_append
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

// every wave performs `per_wave` appends (lane 0 only: what the compiler's wave aggregation leaves), each on counter
// (wave id + i) % n_counters, counters `stride_words` apart; the returned index is used (stored) like a list slot would be
__global__ __launch_bounds__(256) void k_append(uint32_t* counters, int n_counters, int stride_words, int per_wave, uint32_t* sink) {
  const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
  if ((threadIdx.x & 63) != 0) return;
  uint32_t acc = 0;
  for (int i = 0; i < per_wave; ++i) {
    const int c = (wave + i) % n_counters;
    acc += atomicAdd(counters + (size_t)c * stride_words, 1u);
  }
  if (acc == 0xffffffffu) sink[0] = acc;
}

// the same number of appends, but a workgroup's four waves first meet in LDS and ONE lane reserves for all of them
__global__ __launch_bounds__(256) void k_append_block(uint32_t* counters, int n_counters, int stride_words, int per_wave, uint32_t* sink) {
  __shared__ uint32_t s_n;
  if (threadIdx.x == 0) s_n = 0;
  __syncthreads();
  if ((threadIdx.x & 63) == 0) atomicAdd(&s_n, (uint32_t)per_wave);
  __syncthreads();
  if (threadIdx.x == 0) {
    const uint32_t base = atomicAdd(counters + (size_t)(blockIdx.x % n_counters) * stride_words, s_n);
    if (base == 0xffffffffu) sink[0] = base;
  }
}

int main() {
  const int blocks = 64 * 750;   // 64 frames x 3000 templates / 4 waves per workgroup
  uint32_t *d_c, *d_sink;
  const size_t bytes = 64ull * 65536;
  CK(hipMalloc(&d_c, bytes));
  CK(hipMalloc(&d_sink, 4));
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  printf("%d workgroups x 4 waves; appends per wave, counters, spacing -> ms per launch, ns per append\n", blocks);
  const int strides[] = {1, 32, 64, 1024, 16384};
  for (int per_wave : {1, 4})
    for (int nc : {1, 4, 16, 64})
      for (int sw : strides) {
        if (nc == 1 && sw != 1) continue;
        if ((size_t)nc * sw * 4 > bytes) continue;
        CK(hipMemset(d_c, 0, bytes));
        for (int rep = 0; rep < 2; ++rep) {
          CK(hipEventRecord(e0));
          hipLaunchKernelGGL(k_append, dim3(blocks), dim3(256), 0, 0, d_c, nc, sw, per_wave, d_sink);
          CK(hipEventRecord(e1));
          CK(hipEventSynchronize(e1));
        }
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        printf("  per wave %d  counters %2d  spacing %6d B : %7.3f ms  %6.2f ns per append\n", per_wave, nc, sw * 4, ms, ms * 1e6 / ((double)blocks * 4 * per_wave));
      }
  for (int nc : {1, 16}) {
    CK(hipMemset(d_c, 0, bytes));
    for (int rep = 0; rep < 2; ++rep) {
      CK(hipEventRecord(e0));
      hipLaunchKernelGGL(k_append_block, dim3(blocks), dim3(256), 0, 0, d_c, nc, 64, 4, d_sink);
      CK(hipEventRecord(e1));
      CK(hipEventSynchronize(e1));
    }
    float ms = 0;
    CK(hipEventElapsedTime(&ms, e0, e1));
    printf("  one reservation per workgroup (4 waves x 4), counters %2d: %7.3f ms  %6.2f ns per workgroup\n", nc, ms, ms * 1e6 / (double)blocks);
  }
  // the empty grid, for scale
  CK(hipEventRecord(e0));
  hipLaunchKernelGGL(k_append, dim3(blocks), dim3(256), 0, 0, d_c, 1, 1, 0, d_sink);
  CK(hipEventRecord(e1));
  CK(hipEventSynchronize(e1));
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  printf("  no appends: %7.3f ms\n", ms);
  return 0;
}
