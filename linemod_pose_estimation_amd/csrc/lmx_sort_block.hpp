// Workgroup-parallel form of csrc/lmx_sort_emul.hpp (the restatement of libstdc++'s std::sort): the SAME result -- the same
// permutation, ties included -- computed by 256 threads over data in LDS instead of by one lane.
//
// Why that is possible although std::sort is a sequential, unstable algorithm whose order of ties is a property of its exact
// sequence of moves (see lmx_sort_emul.hpp):
//   1. The quicksort phase (__introsort_loop) is a tree of partitions.  The two halves of a partition are disjoint and never touch
//      each other again, so all ranges of one tree level are processed at once; both children of a range get depth_limit - 1, hence a
//      level has ONE depth limit.
//   2. __unguarded_partition(first + 1, last, pivot = first) with a fixed pivot is a pure function of the range:  let L = the positions
//      (ascending) holding an element that is NOT < pivot, R = the positions (descending) holding one that is NOT > pivot.  The
//      sequential loop swaps L[0] <-> R[0], L[1] <-> R[1], ... as long as L[k] < R[k] (both lists are monotone, so the swaps are the
//      prefix k < s), and returns  cut = min(L[s], R[s-1])  (the first pointer stops at the next original stopper, or at the stopper the
//      last swap put at R[s-1]).  Flags, one prefix sum, a scatter, s independent swaps: data parallel.
//   3. __move_median_to_first: three comparisons per range, one thread per range.
//   4. The final insertion sort never moves an element across a partition cut (everything left of a cut is <= its pivot <= everything
//      right of it, and the insertion stops at the first element that is not greater), and within a leaf (<= 16 elements between two
//      cuts) insertion sort is a STABLE sort -- whose result is unique: position = #smaller + #equal-and-earlier in the leaf.
//   5. Depth limit reached (2 * floor(log2 n) levels; median-of-three killers): the remaining ranges are heap-sorted one per thread
//      with the sequential routine of lmx_sort_emul.hpp; rare and small.
// The elements are (key, tag) pairs: `key` is a 64-bit value whose unsigned order IS the comparator (equal keys = equivalent elements),
// `tag` a 16-bit payload (the element's original index) that moves with it.
// Pinned by tests/test_sort_emulation.py::test_device_block_sort_equals_std_sort through lmx_debug_device_sort_perm: the device
// permutation equals libstdc++'s std::sort for random, tie-heavy and adversarial inputs.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "lmx_sort_emul.hpp"

namespace lmx {
namespace sortblk {

constexpr int kThreads = 256;
constexpr int kLeaf = 16;          // libstdc++'s _S_threshold
constexpr int kMaxRanges = 128;    // ranges of > 16 elements among <= 2048

struct Scratch {                   // LDS, besides key[] / tag[] themselves
  unsigned short* seg;             // [n] range of the level an element belongs to, 0xffff = none
  unsigned short* lpos;            // [n]
  unsigned short* rpos;            // [n]
  uint32_t* cut_bits;              // [(n + 31) / 32] bit i = a partition cut (or the array start) lies in front of position i
  // per range, double-buffered by level
  unsigned short r_first[2][kMaxRanges], r_last[2][kMaxRanges];
  uint32_t r_base[kMaxRanges], r_end[kMaxRanges], r_cnt[kMaxRanges];
  unsigned short r_cut[kMaxRanges], r_left[kMaxRanges], r_right[kMaxRanges];
  uint32_t wave_sum[kThreads / 64];
  int n_next;
};

// inclusive sum over the block of `v[0..PER)` per thread (thread t owns elements t*PER .. t*PER+PER-1): v becomes the inclusive scan
template <int PER>
__device__ __forceinline__ void block_scan_inclusive(uint32_t (&v)[PER], uint32_t* wave_sum, int tid) {
#pragma unroll
  for (int j = 1; j < PER; ++j) v[j] += v[j - 1];
  uint32_t tot = v[PER - 1], x = tot;
  const int lane = tid & 63, wave = tid >> 6;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = (uint32_t)__shfl_up((int)x, d, 64);
    if (lane >= d) x += y;
  }
  if (lane == 63) wave_sum[wave] = x;
  __syncthreads();
  uint32_t before = x - tot;
  for (int w = 0; w < wave; ++w) before += wave_sum[w];
#pragma unroll
  for (int j = 0; j < PER; ++j) v[j] += before;
  __syncthreads();
}

// key[0..n), tag[0..n) in LDS -> sorted exactly as std::sort would leave them.  All kThreads threads of the workgroup call it.
// `spill` (global memory, >= 8 * n bytes per workgroup) is used by the heap-sort fallback only.
template <int NMAX>
__device__ void sort(unsigned long long* key, unsigned short* tag, int n, Scratch& S, unsigned long long* spill) {
  constexpr int PER = NMAX / kThreads;
  static_assert(NMAX % kThreads == 0, "whole elements per thread");
  const int tid = threadIdx.x;
  if (n <= 1) return;
  for (int i = tid; i < n; i += kThreads) S.seg[i] = n > kLeaf ? 0 : 0xffff;
  for (int i = tid; i < (n + 31) / 32; i += kThreads) S.cut_bits[i] = i == 0 ? 1u : 0u;
  if (tid == 0) { S.r_first[0][0] = 0; S.r_last[0][0] = (unsigned short)n; S.n_next = 0; }
  __syncthreads();
  int n_ranges = n > kLeaf ? 1 : 0, buf = 0, lg = 0;
  for (int v = n; v > 1; v >>= 1) ++lg;
  int depth = 2 * lg;
  while (n_ranges > 0) {
    if (depth == 0) {
      // heap sort of every remaining range (partial_sort over the whole range), one thread per range; keys are not moved while the heap
      // works on positions, then keys and tags are permuted through the spill area / the lpos slots of the range
      if (tid < n_ranges) {
        const int first = S.r_first[buf][tid], last = S.r_last[buf][tid];
        for (int k = first; k < last; ++k) S.rpos[k] = (unsigned short)k;
        sortemu::heap_sort(S.rpos, first, last, [&](unsigned short a, unsigned short b) { return key[a] < key[b]; });
        for (int k = first; k < last; ++k) { spill[k] = key[S.rpos[k]]; S.lpos[k] = tag[S.rpos[k]]; }
        for (int k = first; k < last; ++k) { key[k] = spill[k]; tag[k] = S.lpos[k]; }
      }
      __syncthreads();
      break;
    }
    --depth;
    // median of three -> position first (one thread per range)
    if (tid < n_ranges) {
      const int first = S.r_first[buf][tid], last = S.r_last[buf][tid];
      const int x = first + 1, y = first + (last - first) / 2, z = last - 1;
      const unsigned long long kx = key[x], ky = key[y], kz = key[z];
      int pick;
      if (kx < ky) pick = ky < kz ? y : (kx < kz ? z : x);
      else pick = kx < kz ? x : (ky < kz ? z : y);
      const unsigned long long tk = key[first]; key[first] = key[pick]; key[pick] = tk;
      const unsigned short tt = tag[first]; tag[first] = tag[pick]; tag[pick] = tt;
      S.r_cnt[tid] = 0;
    }
    __syncthreads();
    // stoppers of the two pointers: low half = "not < pivot" (first pointer), high half = "not > pivot" (last pointer)
    uint32_t f[PER];
    unsigned short rg[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid * PER + j;
      f[j] = 0; rg[j] = 0xffff;
      if (i < n) {
        rg[j] = S.seg[i];
        if (rg[j] != 0xffff) {
          const int first = S.r_first[buf][rg[j]];
          if (i != first) {
            const unsigned long long pk = key[first], k = key[i];
            f[j] = (k >= pk ? 1u : 0u) | (k <= pk ? 0x10000u : 0u);
          }
        }
      }
    }
    uint32_t sc[PER];
#pragma unroll
    for (int j = 0; j < PER; ++j) sc[j] = f[j];
    block_scan_inclusive<PER>(sc, S.wave_sum, tid);
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid * PER + j;
      if (rg[j] == 0xffff) continue;
      if (i == S.r_first[buf][rg[j]]) S.r_base[rg[j]] = sc[j];          // flags are 0 at the pivot: inclusive = exclusive
      if (i == S.r_last[buf][rg[j]] - 1) S.r_end[rg[j]] = sc[j];
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid * PER + j;
      if (rg[j] == 0xffff || f[j] == 0) continue;
      const int slot0 = S.r_first[buf][rg[j]] + 1;
      const uint32_t base = S.r_base[rg[j]], end = S.r_end[rg[j]];
      if (f[j] & 1u) S.lpos[slot0 + (int)((sc[j] & 0xffffu) - 1u - (base & 0xffffu))] = (unsigned short)i;   // rank from the left
      if (f[j] >> 16) S.rpos[slot0 + (int)((end >> 16) - (sc[j] >> 16))] = (unsigned short)i;                 // rank from the right
    }
    __syncthreads();
    // the swaps: slot k of a range pairs L[k] with R[k]
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid * PER + j;
      if (rg[j] == 0xffff) continue;
      const int first = S.r_first[buf][rg[j]];
      if (i == first) continue;
      const uint32_t base = S.r_base[rg[j]], end = S.r_end[rg[j]];
      const int k = i - (first + 1), nL = (int)((end & 0xffffu) - (base & 0xffffu)), nR = (int)((end >> 16) - (base >> 16));
      if (k < nL && k < nR) {
        const int a = S.lpos[i], b = S.rpos[i];
        if (a < b) {
          const unsigned long long tk = key[a]; key[a] = key[b]; key[b] = tk;
          const unsigned short tt = tag[a]; tag[a] = tag[b]; tag[b] = tt;
          atomicAdd(&S.r_cnt[rg[j]], 1u);
        }
      }
    }
    __syncthreads();
    // cut and the next level's ranges (one thread per range)
    if (tid < n_ranges) {
      const int first = S.r_first[buf][tid], last = S.r_last[buf][tid];
      const uint32_t base = S.r_base[tid], end = S.r_end[tid];
      const int nL = (int)((end & 0xffffu) - (base & 0xffffu)), s = (int)S.r_cnt[tid];
      int cut;
      if (s < nL && (s == 0 || S.lpos[first + 1 + s] < S.rpos[first + s])) cut = S.lpos[first + 1 + s];
      else cut = s > 0 ? S.rpos[first + s] : last;   // s == 0 && nL == 0 cannot happen (the median guarantees a stopper)
      S.r_cut[tid] = (unsigned short)cut;
      if (cut < n) atomicOr(&S.cut_bits[cut >> 5], 1u << (cut & 31));
      unsigned short lid = 0xffff, rid = 0xffff;
      if (cut - first > kLeaf) { lid = (unsigned short)atomicAdd(&S.n_next, 1); S.r_first[buf ^ 1][lid] = (unsigned short)first; S.r_last[buf ^ 1][lid] = (unsigned short)cut; }
      if (last - cut > kLeaf) { rid = (unsigned short)atomicAdd(&S.n_next, 1); S.r_first[buf ^ 1][rid] = (unsigned short)cut; S.r_last[buf ^ 1][rid] = (unsigned short)last; }
      S.r_left[tid] = lid; S.r_right[tid] = rid;
    }
    __syncthreads();
#pragma unroll
    for (int j = 0; j < PER; ++j) {
      const int i = tid * PER + j;
      if (rg[j] != 0xffff) S.seg[i] = i < S.r_cut[rg[j]] ? S.r_left[rg[j]] : S.r_right[rg[j]];
    }
    n_ranges = S.n_next;
    buf ^= 1;
    __syncthreads();
    if (tid == 0) S.n_next = 0;
    __syncthreads();
  }
  // final insertion sort = a stable sort of every leaf: rank by counting inside the leaf
  unsigned long long kk[PER];
  unsigned short tt[PER];
  int dst[PER];
#pragma unroll
  for (int j = 0; j < PER; ++j) {
    const int i = tid * PER + j;
    dst[j] = -1;
    if (i >= n) continue;
    int w = i >> 5;
    uint32_t m = S.cut_bits[w] & (0xffffffffu >> (31 - (i & 31)));
    while (m == 0) m = S.cut_bits[--w];                       // bit 0 of word 0 is always set
    const int lo = (w << 5) + 31 - __clz((int)m);
    int hi = n;
    {
      int w2 = i >> 5;
      uint32_t m2 = (i & 31) == 31 ? 0u : (S.cut_bits[w2] & (0xffffffffu << ((i & 31) + 1)));
      const int nw = (n + 31) / 32;
      while (m2 == 0 && ++w2 < nw) m2 = S.cut_bits[w2];
      if (m2 != 0) hi = min(n, (w2 << 5) + __ffs((int)m2) - 1);
    }
    const unsigned long long k = key[i];
    int r = lo;
    for (int q = lo; q < hi; ++q) {
      const unsigned long long o = key[q];
      r += (o < k || (o == k && q < i)) ? 1 : 0;
    }
    kk[j] = k; tt[j] = tag[i]; dst[j] = r;
  }
  __syncthreads();
#pragma unroll
  for (int j = 0; j < PER; ++j)
    if (dst[j] >= 0) { key[dst[j]] = kk[j]; tag[dst[j]] = tt[j]; }
  __syncthreads();
}

}  // namespace sortblk
}  // namespace lmx
