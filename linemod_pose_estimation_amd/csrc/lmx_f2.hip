// SURVEY.md 8f row 2 on the device: the immediate consumer of match() --
//   Detector::match's own std::sort + std::unique (upstream A.10), then the reference's
//   rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU
//   (/root/reference/src/rgbdDetector.cpp:36-70, 72-85, 118-144, 462-574; chained at
//   src/linemod_ensenso_detect_3_mult_detect_service.cpp:376-447)
// -- as ONE kernel that consumes the raw-match slot written by k_refine, one workgroup per frame, everything in LDS.
//
// Both std::sort calls of that chain are unstable sorts whose order of ties is observable (std::unique removes ADJACENT duplicates;
// the greedy NMS walks the clusters in sorted order), so the kernel reproduces libstdc++'s algorithm: the big sort (stage C, up to 2048
// records) with lmx_sort_block.hpp, the workgroup-parallel form that yields the same permutation (round 3: the one-lane emulation took
// 1.4 ms at 500 records per frame, twice the host path; profiles/r03_f2_timing.txt), the small one (stage G, a handful of clusters) with
// the sequential restatement lmx_sort_emul.hpp on one lane:
//   A  collect the frame's records from the slot (all frames of a batch share one list)            parallel, LDS append
//   B  restore upstream insertion order: bitonic sort by order_key (keys are unique)                parallel
//   C  std::sort by Match::operator< (similarity desc, template_id asc)                             parallel emulated introsort
//   D  std::unique (x, y, similarity, class equal; == is an equivalence, so "equal to the previous" decides)   parallel + block scan
//   E  rcd_voting: key = {y / step, x / step, depth ring}; std::map order = bitonic sort by (key, position in the match list)
//   F  clusters = runs of equal key: size filter, mean similarity (double), mean rect (integer division)       lane 0
//   G  std::sort by score desc (emulated) + greedy IoU suppression at 0.4 with the reference's int/float arithmetic   lane 0
// Frames with more than F2_MAX records are flagged and finished by the host path (lmx_cluster_matches), so the result is the
// reference's for every input.
#include <hip/hip_runtime.h>

#include "lmx_internal.hpp"
#include "lmx_sort_block.hpp"
#include "lmx_sort_emul.hpp"

namespace lmx {

namespace {

__device__ __forceinline__ void bitonic_sort_u64(unsigned long long* key, int n_pow2, int tid, int nthreads) {
  for (int k = 2; k <= n_pow2; k <<= 1)
    for (int j = k >> 1; j > 0; j >>= 1) {
      for (int i = tid; i < n_pow2; i += nthreads) {
        const int ixj = i ^ j;
        if (ixj > i) {
          const unsigned long long a = key[i], b = key[ixj];
          const bool up = (i & k) == 0;
          if ((a > b) == up) { key[i] = b; key[ixj] = a; }
        }
      }
      __syncthreads();
    }
}

// the reference's computeIoU (rgbdDetector.cpp:532-574): boxes as {x, y, w, h}; int products converted to float, float division
// (wrapping int arithmetic spelled out, as in lmx_cluster.cpp's box_overlap_ratio: rects of clusters left of / above the origin are huge)
__device__ __forceinline__ int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
__device__ __forceinline__ int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
__device__ __forceinline__ int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
__device__ float box_iou(const int* p, const int* q) {
  const int p_x0 = p[0], p_x1 = wsub(wadd(p[0], p[2]), 1), p_y0 = p[1], p_y1 = wsub(wadd(p[1], p[3]), 1);
  const int q_x0 = q[0], q_x1 = wsub(wadd(q[0], q[2]), 1), q_y0 = q[1], q_y1 = wsub(wadd(q[1], q[3]), 1);
  const int lo_x = max(p_x0, q_x0), hi_x = min(p_x1, q_x1), lo_y = max(p_y0, q_y0), hi_y = min(p_y1, q_y1);
  const bool overlap_x = (lo_x >= p_x0 && lo_x <= p_x1) || (lo_x >= q_x0 && lo_x <= q_x1);
  const bool overlap_y = (lo_y >= p_y0 && lo_y <= p_y1) || (lo_y >= q_y0 && lo_y <= q_y1);
  const float shared = (overlap_x && overlap_y) ? (float)wmul(wadd(wsub(hi_x, lo_x), 1), wadd(wsub(hi_y, lo_y), 1)) : 0.0f;
  const float total = (float)wadd(wmul(p[2], p[3]), wmul(q[2], q[3])) - shared;
  return shared / total;
}

}  // namespace

// `int v; v /= n;` with n a size_t, as the reference writes it: v is converted to size_t first (value mod 2^64), the quotient back to int
__device__ __forceinline__ int div_by_size(int v, int n) {
  return (int)(unsigned)((unsigned long long)(long long)v / (unsigned long long)n);
}

// Match::operator< as one unsigned comparison: similarity descending (IEEE bits made order-preserving, then inverted), template_id
// ascending (signed -> offset binary).  Equal keys <=> neither element is less than the other.
__device__ __forceinline__ unsigned long long match_sort_key(float similarity, int template_id) {
  uint32_t b = __float_as_uint(similarity);
  b = (b & 0x80000000u) ? ~b : (b | 0x80000000u);
  return ((unsigned long long)(~b) << 32) | (unsigned long long)((uint32_t)template_id ^ 0x80000000u);
}

// test hook (lmx_debug_device_sort_perm): the permutation sortblk::sort produces for n <= F2_MAX (similarity, template_id) pairs
__global__ __launch_bounds__(256) void k_debug_block_sort(const float* sim, const int* tid_in, int n, int* perm, unsigned long long* spill) {
  __shared__ unsigned long long s_key[F2_MAX];
  __shared__ unsigned short s_tag[F2_MAX], s_seg[F2_MAX], s_lpos[F2_MAX], s_rpos[F2_MAX];
  __shared__ uint32_t s_cut[F2_MAX / 32];
  __shared__ sortblk::Scratch S;
  if (threadIdx.x == 0) { S.seg = s_seg; S.lpos = s_lpos; S.rpos = s_rpos; S.cut_bits = s_cut; }
  for (int i = threadIdx.x; i < n; i += 256) { s_key[i] = match_sort_key(sim[i], tid_in[i]); s_tag[i] = (unsigned short)i; }
  __syncthreads();
  sortblk::sort<F2_MAX>(s_key, s_tag, n, S, spill);
  for (int i = threadIdx.x; i < n; i += 256) perm[i] = s_tag[i];
}

__global__ __launch_bounds__(256) void k_f2_finalize_cluster(F2Params p) {
  constexpr int NMAX = F2_MAX;
  constexpr int PER = (NMAX + 255) / 256;   // items per thread
  __shared__ unsigned long long s_key[NMAX];
  __shared__ float s_sim[NMAX];
  __shared__ int s_tid[NMAX];
  __shared__ short s_x[NMAX], s_y[NMAX];     // image coordinates (< 32768: lmx_bank_add_class validates feature ranges, frames are smaller)
  __shared__ unsigned short s_cls[NMAX];
  __shared__ unsigned short s_perm[NMAX], s_keep[NMAX];   // 52 KB of LDS up to here
  __shared__ unsigned short s_seg[NMAX], s_rpos[NMAX];    // stage C (sortblk::sort; s_keep doubles as its lpos): 64 KB with its range tables
  __shared__ uint32_t s_cut[NMAX / 32];
  __shared__ sortblk::Scratch S;
  __shared__ int s_n, s_nfinal;
  const int tid = threadIdx.x, frame = blockIdx.x;
  uint32_t* counts = p.out_counts + (size_t)frame * 4;
  lmx_match_t* out_m = p.out_matches + (size_t)frame * NMAX;
  if (tid == 0) { s_n = 0; s_nfinal = 0; }
  __syncthreads();
  // A: this frame's records (order arbitrary), identified by their index in the slot's list
  const uint32_t n_total = min(p.hdr[1], p.cap);
  for (uint32_t i = tid; i < n_total; i += 256)
    if (p.recs[i].frame == frame) {
      const int pos = atomicAdd(&s_n, 1);
      if (pos < NMAX) { s_key[pos] = p.recs[i].order_key; s_tid[pos] = (int)i; }   // s_tid borrowed: record index
    }
  __syncthreads();
  const int n = s_n;
  if (n > NMAX) {   // too many for the LDS path: the host finishes this frame
    if (tid == 0) { counts[0] = (uint32_t)n; counts[1] = 0; counts[2] = 0; counts[3] = 1; }
    return;
  }
  // B: insertion order.  order_key is unique per record (class slot | template_id | coarse raster index), so the low bits of
  // a 64-bit key cannot carry the record index; sort (key) and recover the record by a second lookup table: pack the record's
  // LDS position instead -- positions < 2048 need 11 bits, order_key uses 48 + ... bits, so sort pairs through two passes:
  // keys are unique, hence sorting keys alone and locating each record by binary search of its key is exact.
  int n2 = 1;
  while (n2 < n) n2 <<= 1;
  unsigned long long my_key[PER];   // this thread's unsorted keys (slots tid, tid + 256, ...)
  {
    int cnt = 0;
    for (int i = tid; i < n2; i += 256, ++cnt) my_key[cnt] = i < n ? s_key[i] : ~0ull;
    __syncthreads();
    cnt = 0;
    for (int i = tid; i < n2; i += 256, ++cnt) s_key[i] = my_key[cnt];
  }
  __syncthreads();
  bitonic_sort_u64(s_key, n2, tid, 256);
  // rank of every record = position of its key in the sorted keys
  {
    int cnt = 0;
    for (int i = tid; i < n; i += 256, ++cnt) {
      const unsigned long long k = my_key[cnt];
      int lo = 0, hi = n - 1;
      while (lo < hi) { const int mid = (lo + hi) >> 1; if (s_key[mid] < k) lo = mid + 1; else hi = mid; }
      s_perm[lo] = (unsigned short)i;   // insertion position lo holds LDS slot i
    }
  }
  __syncthreads();
  // load the records in insertion order
  {
    int rec_of_pos[PER];
    int cnt = 0;
    for (int i = tid; i < n; i += 256) rec_of_pos[cnt++] = s_tid[s_perm[i]];
    __syncthreads();
    cnt = 0;
    for (int i = tid; i < n; i += 256) {
      const lmx_raw_match_t r = p.recs[rec_of_pos[cnt++]];
      s_sim[i] = r.similarity; s_tid[i] = r.template_id; s_x[i] = (short)r.x; s_y[i] = (short)r.y; s_cls[i] = (unsigned short)r.class_index;
    }
  }
  __syncthreads();
  // C: std::sort with Match::operator<, order of ties as libstdc++ leaves it (s_key is free again: the insertion order is in place)
  if (tid == 0) { S.seg = s_seg; S.lpos = s_keep; S.rpos = s_rpos; S.cut_bits = s_cut; }
  for (int i = tid; i < n; i += 256) { s_key[i] = match_sort_key(s_sim[i], s_tid[i]); s_perm[i] = (unsigned short)i; }
  __syncthreads();
  sortblk::sort<NMAX>(s_key, s_perm, n, S, reinterpret_cast<unsigned long long*>(p.scratch + (size_t)frame * NMAX * 32));
  // D: std::unique.  Thread t owns positions t * PER .. t * PER + PER - 1; the kept elements' output positions come from a block scan
  {
    uint32_t kf[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int j = tid * PER + q;
      bool keep = j < n;
      if (keep && j > 0) {
        const int a = s_perm[j - 1], b = s_perm[j];
        keep = !(s_x[a] == s_x[b] && s_y[a] == s_y[b] && s_sim[a] == s_sim[b] && s_cls[a] == s_cls[b]);
      }
      kf[q] = keep ? 1u : 0u;
    }
    uint32_t sc[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) sc[q] = kf[q];
    sortblk::block_scan_inclusive<PER>(sc, S.wave_sum, tid);
#pragma unroll
    for (int q = 0; q < PER; ++q) {
      const int j = tid * PER + q;
      if (j < n) s_keep[j] = (unsigned short)(kf[q] ? sc[q] - 1u : 0xffffu);
      if (j == n - 1) s_nfinal = (int)sc[q];
    }
  }
  __syncthreads();
  const int nf = s_nfinal;
  // final list -> global, and compact the LDS arrays into final order (through registers: in-place permutation)
  {
    float r_sim[PER]; int r_tid[PER], r_x[PER], r_y[PER], r_cls[PER], r_dst[PER];
    int cnt = 0;
    for (int j = tid; j < n; j += 256, ++cnt) {
      const int src = s_perm[j];
      r_dst[cnt] = s_keep[j] == 0xffff ? -1 : (int)s_keep[j];
      r_sim[cnt] = s_sim[src]; r_tid[cnt] = s_tid[src]; r_x[cnt] = s_x[src]; r_y[cnt] = s_y[src]; r_cls[cnt] = s_cls[src];
    }
    __syncthreads();
    cnt = 0;
    for (int j = tid; j < n; j += 256, ++cnt) {
      const int d = r_dst[cnt];
      if (d < 0) continue;
      s_sim[d] = r_sim[cnt]; s_tid[d] = r_tid[cnt]; s_x[d] = (short)r_x[cnt]; s_y[d] = (short)r_y[cnt]; s_cls[d] = (unsigned short)r_cls[cnt];
      lmx_match_t m;
      m.x = r_x[cnt]; m.y = r_y[cnt]; m.similarity = r_sim[cnt]; m.template_id = r_tid[cnt]; m.class_index = r_cls[cnt];
      out_m[d] = m;
    }
  }
  __syncthreads();
  if (!p.do_clusters) {
    if (tid == 0) { counts[0] = (uint32_t)nf; counts[1] = 0; counts[2] = 0; counts[3] = 0; }
    return;
  }
  // E: rcd_voting keys.  {y / step, x / step, ring} compared lexicographically as signed ints (std::map<vector<int>, ...>):
  // offset-binary fields of 17 + 17 + 19 bits, the match's position in the final list in the low 11 bits (keeps the order in
  // which matches were voted into a bin)
  __shared__ int s_bad;
  if (tid == 0) s_bad = 0;
  __syncthreads();
  int nf2 = 1;
  while (nf2 < nf) nf2 <<= 1;
  for (int j = tid; j < nf2; j += 256) {
    unsigned long long key = ~0ull;
    if (j < nf) {
      const int t = s_tid[j];
      if (t < 0 || (uint32_t)t >= p.n_templates) { atomicOr(&s_bad, 1); }
      else {
        const int iy = s_y[j] / p.step, ix = s_x[j] / p.step;
        const float depth = (float)p.dists[t];
        const float ring_step = (float)p.radius_step;
        const int ring = (int)((depth - p.radius_min) / ring_step);   // float - double -> double, / float -> double, like the reference
        const long long fy = (long long)iy + (1 << 16), fx = (long long)ix + (1 << 16), fr = (long long)ring + (1 << 18);
        if (fy < 0 || fy >= (1 << 17) || fx < 0 || fx >= (1 << 17) || fr < 0 || fr >= (1 << 19)) atomicOr(&s_bad, 2);
        else key = ((unsigned long long)fy << 47) | ((unsigned long long)fx << 30) | ((unsigned long long)fr << 11) | (unsigned long long)j;
      }
    }
    s_key[j] = key;
  }
  __syncthreads();
  if (s_bad) {   // side-car too short / indices outside the packed range: host path
    if (tid == 0) { counts[0] = (uint32_t)nf; counts[1] = 0; counts[2] = 0; counts[3] = 2; }
    return;
  }
  bitonic_sort_u64(s_key, nf2, tid, 256);
  // F + G on one lane: cluster runs, filter, scores, rects, std::sort by score, greedy NMS
  if (tid == 0) {
    lmx_cluster_t* out_c = p.out_clusters + (size_t)frame * NMAX;
    int32_t* out_mem = p.out_members + (size_t)frame * NMAX;
    // clusters in std::map order = runs of equal key in the sorted array; per-cluster attributes go to a global scratch area
    // (score, range, rect: 8 + 8 + 16 bytes per cluster), order and suppression flags to the LDS arrays that are free now
    int nc = 0;
    double* c_score = reinterpret_cast<double*>(p.scratch + (size_t)frame * NMAX * 32);
    unsigned long long* c_range = reinterpret_cast<unsigned long long*>(p.scratch + (size_t)frame * NMAX * 32 + (size_t)NMAX * 8);   // begin << 32 | count
    int* c_rect = reinterpret_cast<int*>(p.scratch + (size_t)frame * NMAX * 32 + (size_t)NMAX * 16);
    for (int b = 0; b < nf;) {
      int e = b + 1;
      while (e < nf && (s_key[e] >> 11) == (s_key[b] >> 11)) ++e;
      const int cnt = e - b;
      if (cnt > p.size_thresh) {   // cluster_filter: drop clusters with size <= thresh
        double sum = 0.0;
        int X = 0, Y = 0, Wd = 0, Ht = 0;
        for (int k = b; k < e; ++k) {
          const int j = (int)(s_key[k] & 2047u);
          sum += (double)s_sim[j];
          const int32_t* r = p.rects + (size_t)s_tid[j] * 4;
          X += s_x[j]; Y += s_y[j]; Wd += r[2]; Ht += r[3];
        }
        c_range[nc] = ((unsigned long long)b << 32) | (unsigned)cnt;
        c_score[nc] = sum / cnt;
        // upstream: `int X; ... X /= matches.size();` -- the int is converted to size_t for the division (src/rgbdDetector.cpp:
        // nonMaximaSuppressionUsingIOU), so a NEGATIVE sum (matches left of / above the image origin) divides as 2^64 + X
        c_rect[4 * nc + 0] = div_by_size(X, cnt); c_rect[4 * nc + 1] = div_by_size(Y, cnt);
        c_rect[4 * nc + 2] = div_by_size(Wd, cnt); c_rect[4 * nc + 3] = div_by_size(Ht, cnt);
        ++nc;
      }
      b = e;
    }
    for (int i = 0; i < nc; ++i) { s_perm[i] = (unsigned short)i; s_keep[i] = 0; }
    sortemu::sort(s_perm, nc, [&](unsigned short a, unsigned short b) { return c_score[a] > c_score[b]; });
    for (int a = 0; a < nc; ++a) {
      if (s_keep[s_perm[a]]) continue;
      for (int b = a + 1; b < nc; ++b)
        if (!s_keep[s_perm[b]]) {
          const double v = (double)box_iou(&c_rect[4 * s_perm[a]], &c_rect[4 * s_perm[b]]);
          if (v > 0.4) s_keep[s_perm[b]] = 1;
        }
    }
    int n_out = 0, n_mem = 0;
    for (int a = 0; a < nc; ++a) {
      const int c = s_perm[a];
      if (s_keep[c]) continue;
      const int b = (int)(c_range[c] >> 32), cnt = (int)(c_range[c] & 0xffffffffu);
      lmx_cluster_t o;
      const unsigned long long key = s_key[b];
      o.index[0] = (int)((key >> 47) & 0x1ffff) - (1 << 16);
      o.index[1] = (int)((key >> 30) & 0x1ffff) - (1 << 16);
      o.index[2] = (int)((key >> 11) & 0x7ffff) - (1 << 18);
      for (int k = 0; k < 4; ++k) o.rect[k] = c_rect[4 * c + k];
      o.score = c_score[c];
      o.member_begin = n_mem; o.member_count = cnt;
      for (int k = b; k < b + cnt; ++k) out_mem[n_mem++] = (int32_t)(s_key[k] & 2047u);
      out_c[n_out++] = o;
    }
    counts[0] = (uint32_t)nf; counts[1] = (uint32_t)n_out; counts[2] = (uint32_t)n_mem; counts[3] = 0;
  }
}

void launch_debug_block_sort(hipStream_t s, const float* sim, const int* tid, int n, int* perm, unsigned long long* spill) {
  hipLaunchKernelGGL(k_debug_block_sort, dim3(1), dim3(256), 0, s, sim, tid, n, perm, spill);
}

void launch_f2(hipStream_t s, const F2Params& p) {
  hipLaunchKernelGGL(k_f2_finalize_cluster, dim3((unsigned)p.n_frames), dim3(256), 0, s, p);
}

}  // namespace lmx
