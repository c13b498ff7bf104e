// Host->device frame transfer options for the host-frame boundary of lmx_match (the reference hands match() host cv::Mat's:
// src/linemod_ensenso_detect_3_mult_detect_service.cpp:324-344).  One batch = 64 frames x (640x480x3 + 640x480x2) B = 98.3 MB.
//   1. hipMemcpyAsync from one pinned buffer (SDMA), 2. the same in 128 per-image calls, 3. a kernel pulling from mapped
//   pinned memory (uint4 per thread), 4. hipMemcpyAsync straight from pageable memory, 5. host memcpy pageable -> pinned with
//   1..16 threads (the staging step in front of 1-3).
//   build: hipcc --offload-arch=gfx950 -O3 -pthread -o h2d_rate h2d_rate.hip
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

__global__ __launch_bounds__(256) void k_pull(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}
// 4 independent loads in flight per thread
__global__ __launch_bounds__(256) void k_pull4(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  const size_t stride = (size_t)gridDim.x * 256u;
  size_t i = (size_t)blockIdx.x * 256u + threadIdx.x;
  for (; i + 3 * stride < n16; i += 4 * stride) {
    const uint4 a = src[i], b = src[i + stride], c = src[i + 2 * stride], d = src[i + 3 * stride];
    dst[i] = a; dst[i + stride] = b; dst[i + 2 * stride] = c; dst[i + 3 * stride] = d;
  }
  for (; i < n16; i += stride) dst[i] = src[i];
}

// the staging copy liblmx uses (csrc/lmx_hostcopy.cpp): AVX2 non-temporal stores into the pinned buffer
#include <immintrin.h>
__attribute__((target("avx2"))) static void copy_nt(uint8_t* dst, const uint8_t* src, size_t n) {
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256((const __m256i*)(src + i)), b = _mm256_loadu_si256((const __m256i*)(src + i + 32));
    const __m256i c = _mm256_loadu_si256((const __m256i*)(src + i + 64)), d = _mm256_loadu_si256((const __m256i*)(src + i + 96));
    _mm256_stream_si256((__m256i*)(dst + i), a); _mm256_stream_si256((__m256i*)(dst + i + 32), b);
    _mm256_stream_si256((__m256i*)(dst + i + 64), c); _mm256_stream_si256((__m256i*)(dst + i + 96), d);
  }
  if (i < n) memcpy(dst + i, src + i, n - i);
  _mm_sfence();
}

static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

int main() {
  const size_t frame = 640 * 480 * 5, n_frames = 64, bytes = frame * n_frames;
  CHECK(hipSetDevice(0));
  hipStream_t s;
  CHECK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  uint8_t *h_pin = nullptr, *d = nullptr, *h_pin_dev = nullptr;
  CHECK(hipHostMalloc((void**)&h_pin, bytes, hipHostMallocMapped));
  CHECK(hipHostGetDevicePointer((void**)&h_pin_dev, h_pin, 0));
  CHECK(hipMalloc((void**)&d, bytes));
  std::vector<uint8_t> pageable(bytes);
  for (size_t i = 0; i < bytes; ++i) pageable[i] = (uint8_t)(i * 2654435761u >> 24);
  memcpy(h_pin, pageable.data(), bytes);
  const int reps = 10;
  auto report = [&](const char* what, double sec) { printf("%-64s %8.3f ms per batch  %7.2f GB/s  %8.0f frames/s\n", what, sec * 1e3, bytes / sec / 1e9, n_frames / sec); fflush(stdout); };
  {
    CHECK(hipMemcpyAsync(d, h_pin, bytes, hipMemcpyHostToDevice, s)); CHECK(hipStreamSynchronize(s));
    double t0 = now();
    for (int r = 0; r < reps; ++r) CHECK(hipMemcpyAsync(d, h_pin, bytes, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    report("hipMemcpyAsync, one call, pinned source", (now() - t0) / reps);
  }
  {
    double t0 = now();
    for (int r = 0; r < reps; ++r)
      for (size_t f = 0; f < n_frames; ++f) {
        CHECK(hipMemcpyAsync(d + f * frame, h_pin + f * frame, 640 * 480 * 3, hipMemcpyHostToDevice, s));
        CHECK(hipMemcpyAsync(d + f * frame + 640 * 480 * 3, h_pin + f * frame + 640 * 480 * 3, 640 * 480 * 2, hipMemcpyHostToDevice, s));
      }
    double t_issue = now() - t0;
    CHECK(hipStreamSynchronize(s));
    report("hipMemcpyAsync, 128 calls per batch, pinned source", (now() - t0) / reps);
    printf("    (host time to issue the 128 calls: %.3f ms per batch)\n", t_issue / reps * 1e3);
  }
  {
    double t0 = now();
    for (int r = 0; r < reps; ++r)
      for (size_t f = 0; f < n_frames; ++f)
        CHECK(hipMemcpy2DAsync(d + f * frame, 640 * 3, h_pin + f * frame, 640 * 3 + 0, 640 * 3, 480, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    printf("hipMemcpy2DAsync, 64 colour images (row-wise 2D copy)            %8.3f ms per 64 calls\n", (now() - t0) / reps * 1e3);
  }
  for (int blocks : {64, 256, 1024, 4096}) {
    for (int variant = 0; variant < 2; ++variant) {
      auto launch = [&]() {
        if (variant == 0) hipLaunchKernelGGL(k_pull, dim3(blocks), dim3(256), 0, s, (uint4*)d, (const uint4*)h_pin_dev, bytes / 16);
        else hipLaunchKernelGGL(k_pull4, dim3(blocks), dim3(256), 0, s, (uint4*)d, (const uint4*)h_pin_dev, bytes / 16);
      };
      launch(); CHECK(hipStreamSynchronize(s));
      double t0 = now();
      for (int r = 0; r < reps; ++r) launch();
      CHECK(hipStreamSynchronize(s));
      char name[128];
      snprintf(name, sizeof(name), "kernel pull from mapped pinned memory, %d blocks, %s", blocks, variant ? "4 loads in flight" : "1 load in flight");
      report(name, (now() - t0) / reps);
    }
  }
  {
    double t0 = now();
    for (int r = 0; r < 3; ++r) CHECK(hipMemcpyAsync(d, pageable.data(), bytes, hipMemcpyHostToDevice, s));
    CHECK(hipStreamSynchronize(s));
    report("hipMemcpyAsync straight from pageable memory", (now() - t0) / 3);
  }
  for (int nt : {1, 2, 4, 8, 12, 16}) {
    double t0 = now();
    for (int r = 0; r < reps; ++r) {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
          for (size_t f = t; f < n_frames; f += nt) memcpy(h_pin + f * frame, pageable.data() + f * frame, frame);
        });
      for (auto& x : th) x.join();
    }
    char name[128];
    snprintf(name, sizeof(name), "host memcpy pageable -> pinned, %d thread(s) (spawned per batch)", nt);
    report(name, (now() - t0) / reps);
  }
  for (int nt : {1, 2, 4, 8, 16}) {
    double t0 = now();
    for (int r = 0; r < reps; ++r) {
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t]() {
          for (size_t f = t; f < n_frames; f += nt) copy_nt(h_pin + f * frame, pageable.data() + f * frame, frame);
        });
      for (auto& x : th) x.join();
    }
    char name[128];
    snprintf(name, sizeof(name), "host copy pageable -> pinned, non-temporal stores, %d thread(s)", nt);
    report(name, (now() - t0) / reps);
  }
  // staging (8 threads) overlapped with the DMA of the previous batch: two pinned buffers
  {
    uint8_t* h_pin2 = nullptr;
    CHECK(hipHostMalloc((void**)&h_pin2, bytes, hipHostMallocMapped));
    uint8_t* bufs[2] = {h_pin, h_pin2};
    hipEvent_t done[2];
    CHECK(hipEventCreateWithFlags(&done[0], hipEventDisableTiming)); CHECK(hipEventCreateWithFlags(&done[1], hipEventDisableTiming));
    const int nt = 8;
    double t0 = now();
    for (int r = 0; r < 2 * reps; ++r) {
      uint8_t* b = bufs[r & 1];
      if (r >= 2) CHECK(hipEventSynchronize(done[r & 1]));
      std::vector<std::thread> th;
      for (int t = 0; t < nt; ++t)
        th.emplace_back([&, t]() { for (size_t f = t; f < n_frames; f += nt) memcpy(b + f * frame, pageable.data() + f * frame, frame); });
      for (auto& x : th) x.join();
      CHECK(hipMemcpyAsync(d, b, bytes, hipMemcpyHostToDevice, s));
      CHECK(hipEventRecord(done[r & 1], s));
    }
    CHECK(hipStreamSynchronize(s));
    report("pipelined: 8-thread staging of batch i+1 || DMA of batch i", (now() - t0) / (2 * reps));
  }
  return 0;
}
