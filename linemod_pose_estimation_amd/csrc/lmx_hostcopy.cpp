// Host-side copy into the pinned staging buffers of lmx_ctx_upload.  The staging buffer is written once by the CPU and read
// once by the DMA engine, so the stores bypass the caches (AVX2 non-temporal stores: no read-for-ownership of the destination
// lines, no eviction of the caller's working set); the source is read normally.  Falls back to memcpy on CPUs without AVX2 and
// for short copies.  Plain C++ (no HIP): compiled for the host only.
#include <immintrin.h>

#include <cstdint>
#include <cstring>

namespace lmx {

__attribute__((target("avx2"))) static void copy_nt_avx2(uint8_t* dst, const uint8_t* src, size_t n) {
  const size_t head = (32 - (reinterpret_cast<uintptr_t>(dst) & 31)) & 31;
  if (head) { std::memcpy(dst, src, head); dst += head; src += head; n -= head; }
  size_t i = 0;
  for (; i + 128 <= n; i += 128) {
    const __m256i a = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i));
    const __m256i b = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 32));
    const __m256i c = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 64));
    const __m256i d = _mm256_loadu_si256(reinterpret_cast<const __m256i*>(src + i + 96));
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i), a);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 32), b);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 64), c);
    _mm256_stream_si256(reinterpret_cast<__m256i*>(dst + i + 96), d);
  }
  if (i < n) std::memcpy(dst + i, src + i, n - i);
  _mm_sfence();
}

// Streamed input (lmx_internal.hpp, StreamWait): publish the progress word behind the rows.  The rows above went out as non-temporal stores through
// the same write-combining mapping; the fence in front keeps the flag behind them (PCIe posted writes then stay in order), the fence behind pushes
// the flag out of the write-combining buffer now instead of whenever it fills.
void stream_store_flag(uint32_t* flag, uint32_t value) {
  _mm_sfence();
  _mm_stream_si32(reinterpret_cast<int*>(flag), (int)value);
  _mm_sfence();
}

void stream_copy(void* dst, const void* src, size_t n) {
  static const bool avx2 = __builtin_cpu_supports("avx2");
  if (avx2 && n >= 1024) copy_nt_avx2(static_cast<uint8_t*>(dst), static_cast<const uint8_t*>(src), n);
  else std::memcpy(dst, src, n);
}

}  // namespace lmx
