// A deterministic restatement of the sorting algorithm behind libstdc++'s std::sort (introsort: median-of-three quicksort with a
// depth limit of 2*floor(log2 n), heap sort when the limit is hit, segments of <= 16 elements finished by one insertion sort),
// written from the published description of that algorithm so that it can run where libstdc++ cannot: inside a HIP kernel.
//
// Why it exists.  Upstream's observable output depends on std::sort's order of TIES twice: Detector::match sorts the matches by
// (similarity desc, template_id asc) and then removes ADJACENT duplicates with std::unique, and the reference's
// nonMaximaSuppressionUsingIOU sorts the clusters by score with std::sort before its greedy suppression
// (/root/reference/src/rgbdDetector.cpp:462-530).  std::sort is not stable, so "which equal element ends up where" is a property of
// the algorithm; to reproduce the reference's results on the device, the device must run the same sequence of comparisons and
// moves.  The functions below operate on an array of indices `a[0..n)` with a strict-weak-order predicate less(i, j) on the
// elements the indices stand for; moving indices is equivalent to moving the elements.
//
// Pinned by tests/test_sort_emulation.py: for hundreds of thousands of random and adversarial tie-heavy inputs the permutation
// equals the one libstdc++'s std::sort produces (through lmx_debug_introsort_perm vs the oracle's lmo_std_sort_perm).
#pragma once

#include <stdint.h>

#ifdef __HIPCC__
#define LMX_HD __host__ __device__ __forceinline__
#else
#define LMX_HD inline
#endif

namespace lmx {
namespace sortemu {

template <typename Idx, typename Less>
LMX_HD void unguarded_linear_insert(Idx* a, int last, Less less) {
  const Idx val = a[last];
  int next = last - 1;
  while (less(val, a[next])) {
    a[last] = a[next];
    last = next;
    --next;
  }
  a[last] = val;
}

template <typename Idx, typename Less>
LMX_HD void insertion_sort(Idx* a, int first, int last, Less less) {
  if (first == last) return;
  for (int i = first + 1; i != last; ++i) {
    if (less(a[i], a[first])) {
      const Idx val = a[i];
      for (int k = i; k > first; --k) a[k] = a[k - 1];   // move_backward(first, i, i + 1)
      a[first] = val;
    } else {
      unguarded_linear_insert(a, i, less);
    }
  }
}

template <typename Idx, typename Less>
LMX_HD void push_heap(Idx* a, int first, int hole, int top, Idx value, Less less) {
  int parent = (hole - 1) / 2;
  while (hole > top && less(a[first + parent], value)) {
    a[first + hole] = a[first + parent];
    hole = parent;
    parent = (hole - 1) / 2;
  }
  a[first + hole] = value;
}

template <typename Idx, typename Less>
LMX_HD void adjust_heap(Idx* a, int first, int hole, int len, Idx value, Less less) {
  const int top = hole;
  int second = hole;
  while (second < (len - 1) / 2) {
    second = 2 * (second + 1);
    if (less(a[first + second], a[first + (second - 1)])) --second;
    a[first + hole] = a[first + second];
    hole = second;
  }
  if ((len & 1) == 0 && second == (len - 2) / 2) {
    second = 2 * (second + 1);
    a[first + hole] = a[first + (second - 1)];
    hole = second - 1;
  }
  push_heap(a, first, hole, top, value, less);
}

// heap sort of [first, last): what the depth limit falls back to (partial_sort over the whole range)
template <typename Idx, typename Less>
LMX_HD void heap_sort(Idx* a, int first, int last, Less less) {
  const int len = last - first;
  if (len >= 2) {
    int parent = (len - 2) / 2;
    for (;;) {
      const Idx value = a[first + parent];
      adjust_heap(a, first, parent, len, value, less);
      if (parent == 0) break;
      --parent;
    }
  }
  while (last - first > 1) {
    --last;
    const Idx value = a[last];
    a[last] = a[first];
    adjust_heap(a, first, 0, last - first, value, less);
  }
}

template <typename Idx, typename Less>
LMX_HD void move_median_to_first(Idx* a, int result, int x, int y, int z, Less less) {
  auto swap = [&](int p, int q) { const Idx t = a[p]; a[p] = a[q]; a[q] = t; };
  if (less(a[x], a[y])) {
    if (less(a[y], a[z])) swap(result, y);
    else if (less(a[x], a[z])) swap(result, z);
    else swap(result, x);
  } else if (less(a[x], a[z])) swap(result, x);
  else if (less(a[y], a[z])) swap(result, z);
  else swap(result, y);
}

template <typename Idx, typename Less>
LMX_HD int unguarded_partition(Idx* a, int first, int last, int pivot, Less less) {
  for (;;) {
    while (less(a[first], a[pivot])) ++first;
    --last;
    while (less(a[pivot], a[last])) --last;
    if (!(first < last)) return first;
    const Idx t = a[first]; a[first] = a[last]; a[last] = t;
    ++first;
  }
}

// std::sort(a, a + n, less): the recursion of the quicksort phase (always on the upper part, the lower part continues the
// loop) is kept on an explicit stack of at most 2*log2(n) + 1 frames
template <typename Idx, typename Less>
LMX_HD void sort(Idx* a, int n, Less less) {
  if (n <= 0) return;
  constexpr int kThreshold = 16;
  int lg = 0;
  for (int v = n; v > 1; v >>= 1) ++lg;
  struct Frame { int first, last, depth; };
  Frame stack[72];
  int sp = 0;
  stack[sp++] = Frame{0, n, 2 * lg};
  while (sp > 0) {
    Frame f = stack[--sp];
    int first = f.first, last = f.last, depth = f.depth;
    while (last - first > kThreshold) {
      if (depth == 0) { heap_sort(a, first, last, less); break; }
      --depth;
      const int mid = first + (last - first) / 2;
      move_median_to_first(a, first, first + 1, mid, last - 1, less);
      const int cut = unguarded_partition(a, first + 1, last, first, less);
      // libstdc++ recurses into [cut, last) FIRST and then continues with [first, cut): the two ranges are disjoint, so the order
      // in which they are processed does not change the result; the upper range goes onto the stack
      if (sp < 72) stack[sp++] = Frame{cut, last, depth};
      last = cut;
    }
  }
  // final insertion sort
  if (n > kThreshold) {
    insertion_sort(a, 0, kThreshold, less);
    for (int i = kThreshold; i != n; ++i) unguarded_linear_insert(a, i, less);
  } else {
    insertion_sort(a, 0, n, less);
  }
}

}  // namespace sortemu
}  // namespace lmx
