// Can the host store straight into device memory (fine-grained allocation, PCIe BAR) and how fast?  A candidate for the single-frame
// upload path (no staging copy, no DMA start latency).  Measured on MI355X at the end of round 2: see profiles/r02_bar_store.txt
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstring>
#include <chrono>
#include <vector>
#include <thread>
#include <atomic>
#include <immintrin.h>
__attribute__((target("avx2"))) static void nt_copy(char* dst, const char* src, size_t n) {   // non-temporal 32-byte stores (dst 32-byte aligned here)
  size_t i = 0;
  for (; i + 32 <= n; i += 32) _mm256_stream_si256((__m256i*)(dst + i), _mm256_loadu_si256((const __m256i*)(src + i)));
  for (; i < n; ++i) dst[i] = src[i];
  _mm_sfence();
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
__global__ void sum(const unsigned* p, size_t n, unsigned long long* out) {
  unsigned long long s = 0;
  for (size_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
  atomicAdd(out, s);
}
int main() {
  const size_t bytes = 1536000;
  void* d = nullptr;
  CK(hipExtMallocWithFlags(&d, bytes, hipDeviceMallocFinegrained));
  hipPointerAttribute_t a;
  CK(hipPointerGetAttributes(&a, d));
  printf("fine-grained device alloc ok: type %d hostPointer %p devicePointer %p\n", (int)a.type, a.hostPointer, a.devicePointer);
  std::vector<unsigned> src(bytes / 4);
  for (size_t i = 0; i < src.size(); ++i) src[i] = (unsigned)i * 2654435761u;
  unsigned long long expect = 0; for (unsigned v : src) expect += v;
  unsigned long long* out; CK(hipMalloc(&out, 8));
  fflush(stdout);
  // persistent worker threads (spinning on a generation counter), like the library's copy pool: thread creation is not in the time
  for (int threads : {1, 2, 3, 4, 6, 8}) {
    std::atomic<int> gen{0}, done{0};
    std::atomic<bool> stop{false};
    std::vector<std::thread> th;
    for (int t = 1; t < threads; ++t)
      th.emplace_back([&, t]() {
        int seen = 0;
        for (;;) {
          while (gen.load(std::memory_order_acquire) == seen) { if (stop.load()) return; }
          seen = gen.load();
          size_t lo = bytes / threads * t, hi = t + 1 == threads ? bytes : bytes / threads * (t + 1);
          nt_copy((char*)d + lo, (const char*)src.data() + lo, hi - lo);
          done.fetch_add(1, std::memory_order_release);
        }
      });
    double best = 1e9, sum_us = 0;
    const int reps = 50;
    for (int rep = 0; rep < reps; ++rep) {
      auto t0 = std::chrono::steady_clock::now();
      done.store(0);
      gen.fetch_add(1, std::memory_order_release);
      nt_copy((char*)d, (const char*)src.data(), bytes / threads);
      while (done.load(std::memory_order_acquire) != threads - 1) {}
      double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us < best) best = us;
      sum_us += us;
    }
    stop.store(true);
    for (auto& x : th) x.join();
    CK(hipMemset(out, 0, 8));
    hipLaunchKernelGGL(sum, dim3(64), dim3(256), 0, 0, (const unsigned*)d, bytes / 4, out);
    unsigned long long got = 0; CK(hipMemcpy(&got, out, 8, hipMemcpyDeviceToHost));
    printf("host stores into device memory, %d threads: best %.1f us (%.1f GB/s), mean %.1f us, device sees it: %s\n", threads, best, bytes / best / 1e3, sum_us / reps, got == expect ? "yes" : "NO");
    fflush(stdout);
  }
  return 0;
}
