// HIP kernels (gfx950 / CDNA4, wave64) of the LINEMOD matching path.
//
// Stage map (upstream cv::linemod, call site /root/reference/src/rgbdDetector.cpp:33):
//   k_color_quantize      a4+a5(+a6) quantizedOrientations + hysteresisGradient fused over an LDS tile; the same tile also
//                         produces cv::pyrDown of the colour source for the next pyramid level.  The orientation label is an
//                         exact integer rule (orientation_label16), no float left in this kernel
//   k_depth_quantize      a7(+a8) quantizedNormals + medianBlur(5) fused (counting median on cumulative u64 counters: labels take
//                         only 9 values); also writes level 1's labels (nearest-neighbour /2)
//   k_nn_down2            a8   DepthNormalPyramid::pyrDown for levels >= 2
//   k_spread_linearize_t  a10+a11+a12 spread(T) + computeResponseMaps + linearize x8, one pass, LDS strip, T in {4, 5, 8};
//                         finer levels: linearised spread bytes only; coarsest level: nibble-packed response memories
//   k_spread_linearize, k_pack_nibbles   the same for any T / width (byte memories, then two responses per byte)
//   k_score_coarse_sb     a13+a14+a15 similarity + addSimilarities + threshold scan, one wave per (frame, template), for banks
//                         with <= 63 coarsest-level features per template (every bank the reference trains): feature table as
//                         16-dword scalar blocks; k_score_coarse_u8 = its predecessor, k_score_coarse = the generic version
//   k_refine              a16  similarityLocal + argmax + threshold, one workgroup per candidate; for one or two frames per call
//                         its last workgroup also publishes the records to the pinned slot (no read-back launch)
//   k_small_depth_color, k_small_spread   the small-batch chain's fused launches (depth L0 + colour L1; spread of both levels)
//   k_publish_records, k_publish_blocks, k_pull_blocks, k_copy_bytes   read-back / gather-block copies by kernel (pinned host
//                         memory mapped into the device, or a peer's buffer)
//   k_pre_color, k_pre_depth   SURVEY 8f row 4: the node-side steps in front of match()
// None of this is GEMM-shaped: integer / LUT / byte-add work, no MFMA.  The one float stage (normal normalisation) keeps
// upstream's written order; the library is built with -ffp-contract=off and hipcc's default correctly rounded fp32
// divide/sqrt (depth_lut_index_lean spells the same two expansions out without their out-of-range steps).

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <type_traits>

#include "lmx_internal.hpp"
#include "lmx_color_quantize.hpp"

// Wave priorities (s_setprio 0..3) of the memory-bound kernels against the issue-bound quantisers of the other lanes; experiment switches,
// see DESIGN.md section 8 (round 4) for what was measured
#ifndef LMX_PRIO_SCORE
#define LMX_PRIO_SCORE 0
#endif
#ifndef LMX_PRIO_REFINE
#define LMX_PRIO_REFINE 0
#endif
#ifndef LMX_SC_EXIT
#define LMX_SC_EXIT 0
#endif
#ifndef LMX_PRIO_QUANT
#define LMX_PRIO_QUANT 0
#endif
#ifndef LMX_PRIO_SPREAD
#define LMX_PRIO_SPREAD 0
#endif
namespace lmx {

namespace {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

__device__ __forceinline__ int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) p = p < 0 ? -p : 2 * (len - 1) - p;
  return p;
}

// The first workgroup of a batch's first kernel resets the output slot: its 16-dword header and the candidate list's stripe counters
// in front of it (lmx_internal.hpp, kCandStripes).
__device__ __forceinline__ void clear_slot_counters(uint32_t* header, int tid) {
  if (tid < 16) header[tid] = 0u;
  else if (tid >= 64 && tid < 64 + kCandStripes + 1) (header - (size_t)(kCandStripes + 1) * kStripeWords)[(size_t)(tid - 64) * kStripeWords] = 0u;
}

__device__ __forceinline__ uint32_t load_u32_unaligned(const uint8_t* p) {
  uint32_t v;
  __builtin_memcpy(&v, p, 4);  // gfx950 runs in unaligned-access mode: one global_load_dword
  return v;
}

// Tile -> workgroup mapping of the per-pixel kernels.  Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share an XCD
// and its private L2), so with the plain (tile_x, tile_y, frame) grid neighbouring tiles of a frame run on different XCDs and
// every XCD's L2 fetches its own copy of the halo rows (PMC, round 1: k_depth_quantize moved 2.6x, k_color_quantize 1.4x the
// bytes it needs through HBM).  With >= 8 frames in the batch the grid is 1-D and XCD k processes frames k, k + 8, ...: all
// tiles of a frame share one L2, a halo is fetched from HBM once.  n_frames_x = 0 selects the plain 3-D grid (small batches).
// Placement only changes speed and traffic, never results.
__device__ __forceinline__ bool tile_of_block(const uint3 bid, int n_frames_x, int tiles_x, int tiles_y, int& tx, int& ty, int& frame) {
  if (n_frames_x > 0) {
    const int xcd = bid.x & 7, idx = bid.x >> 3;
    const int per = tiles_x * tiles_y;
    const int fslot = idx / per, t = idx - fslot * per;
    frame = xcd + 8 * fslot;
    ty = t / tiles_x;
    tx = t - ty * tiles_x;
    return frame < n_frames_x;
  }
  tx = bid.x; ty = bid.y; frame = bid.z;
  return true;
}

// SIMILARITY_LUT: chunk 2k = orientation k vs low nibble, 2k+1 = vs high nibble.
struct SimLut { uint8_t v[256]; };
constexpr SimLut kSimLut = {{
    0, 4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4,  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 3, 4, 4, 3, 3, 4, 4, 2, 3, 4, 4, 3, 3, 4, 4,  0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1,
    0, 2, 3, 3, 4, 4, 4, 4, 3, 3, 3, 3, 4, 4, 4, 4,  0, 2, 1, 2, 0, 2, 1, 2, 0, 2, 1, 2, 0, 2, 1, 2,
    0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4,  0, 3, 2, 3, 1, 3, 2, 3, 0, 3, 2, 3, 1, 3, 2, 3,
    0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3,  0, 4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4,
    0, 1, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,  0, 3, 4, 4, 3, 3, 4, 4, 2, 3, 4, 4, 3, 3, 4, 4,
    0, 2, 1, 2, 0, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2,  0, 2, 3, 3, 4, 4, 4, 4, 3, 3, 3, 3, 4, 4, 4, 4,
    0, 3, 2, 3, 1, 3, 2, 3, 0, 3, 2, 3, 1, 3, 2, 3,  0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4}};
// The eight responses of a spread byte as one u64 (byte o = max(LUT_lo[o][v & 15], LUT_hi[o][v >> 4])), built at compile time: the
// spread kernels used to derive their 256-entry LDS copy from SIMILARITY_LUT with sixteen table loads per thread and workgroup.
struct RespTab { unsigned long long v[256]; };
constexpr RespTab make_resp_tab(const SimLut& lut) {
  RespTab t{};
  for (int v = 0; v < 256; ++v) {
    unsigned long long r = 0;
    for (int o = 0; o < 8; ++o) {
      const uint8_t lo = lut.v[32 * o + (v & 15)], hi = lut.v[32 * o + 16 + (v >> 4)];
      r |= (unsigned long long)(lo > hi ? lo : hi) << (8 * o);
    }
    t.v[v] = r;
  }
  return t;
}
__constant__ RespTab c_resp_tab = make_resp_tab(kSimLut);

// =========================================================================================================
// a4 + a5 (+ a6)  quantizedOrientations + hysteresisGradient fused over an LDS tile, + cv::pyrDown of the source for the next level.
// The body lives in lmx_color_quantize.hpp (it also compiles for the CPU, where tests/test_color_kernel_host.py runs it against the oracle);
// here: the workgroup -> tile mapping, the slot-header clear of a chain's first kernel, the barrier.  The body takes its workgroup index as an
// argument so that the small-batch chain can run it inside a fused launch (k_small_depth_color below); k_color_quantize passes its own.
// =========================================================================================================
constexpr int CQ_TW = cq::TW, CQ_TH = 16;   // tile of the small-batch chain and of small images
constexpr int CQ_TH_TALL = 32;               // batches: the taller tile recomputes less halo (74 x 42 inputs per 64 x 32 outputs against 74 x 26 per 64 x 16)

__device__ __forceinline__ int orientation_label16(int dx, int dy) { return cq::orientation_label16(dx, dy); }

// Streamed input (lmx_internal.hpp, StreamWait): block-uniform wait until `need_rows` rows of the batch have landed.  Thread 0 polls -- acquire at
// system scope, so that the loads behind the barrier cannot be served from lines cached before the host wrote them -- with a pause between two
// polls; false when the wall clock ran out (the caller's workgroup then leaves its tile alone).
__device__ __forceinline__ bool stream_wait_rows(const StreamWait& w, uint32_t first_row, uint32_t need_rows) {
  __shared__ int s_stream_ok;
  if (threadIdx.x == 0) {
    int ok = 1;
    const unsigned long long t0 = wall_clock64();
    for (;;) {
      // relaxed while polling (an acquire here would invalidate the caches on every poll), ONE acquire fence once the rows are there
      const uint32_t v = __hip_atomic_load(w.flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
      const uint32_t top = (v >> 20) == w.seq ? (v & 0xfffffu) : 0u;   // rows [0, top) are there
      if (top >= need_rows) break;
      if (w.flag_hi != nullptr) {
        const uint32_t h = __hip_atomic_load(w.flag_hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((h >> 20) == w.seq && ((h & 0xfffffu) <= first_row || (h & 0xfffffu) <= top)) break;   // rows [h, end) are there too
      }
      if (wall_clock64() - t0 > (unsigned long long)w.timeout_ticks) {
        ok = 0;
        atomicOr(w.fail, 1u);
        break;
      }
      __builtin_amdgcn_s_sleep(4);
    }
    __atomic_thread_fence(__ATOMIC_ACQUIRE);   // system scope: the loads behind the barrier must not be served from lines cached before the host's stores
    s_stream_ok = ok;
  }
  __syncthreads();
  return s_stream_ok != 0;
}

struct CqRun {   // one stage of the tile for the calling thread, then the barrier that separates it from the next stage
  template <typename F>
  __device__ __forceinline__ void operator()(F&& stage) const {
    stage((int)threadIdx.x);
    __syncthreads();
  }
};

template <int TH, bool TRAIN>
__device__ __forceinline__ void color_quantize_body(const uint3 bid, const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                    uint8_t* __restrict__ pyr_dst, float* __restrict__ mag_dst, int H, int W, float thr_sq,
                                                    uint32_t* __restrict__ clear16, int n_frames_x, const StreamWait wait = StreamWait()) {
  static_assert(TH == 16 || TH == 32, "tile heights the launchers use");
  __shared__ __align__(16) uint8_t s_raw[cq::Geo<TH>::LDS_BYTES];
  // first kernel of a batch's chain: clears the output slot's 64-byte header (candidate / match counters) in passing, which
  // saves the chain a separate memset kernel (4 us + a launch gap, 10 % of a single-frame step)
  if (clear16 != nullptr && (bid.x | bid.y | bid.z) == 0) clear_slot_counters(clear16, (int)threadIdx.x);
  int tile_x, tile_y, frame;
  if (!tile_of_block(bid, n_frames_x, (W + CQ_TW - 1) / CQ_TW, (H + TH - 1) / TH, tile_x, tile_y, frame)) return;
  // streamed input: the tile reads source rows y0 - 5 .. y0 + TH + 4 of its frame
  if (wait.flag != nullptr && !stream_wait_rows(wait, (uint32_t)(frame * H + max(0, tile_y * TH - 5)), (uint32_t)(frame * H + min(H, tile_y * TH + TH + 5)))) return;
  const size_t px = (size_t)H * W;
  cq::color_quantize_tile<TH, TRAIN>(tile_x, tile_y, src + (size_t)frame * px * 3, dst + (size_t)frame * px,
                                     pyr_dst ? pyr_dst + (size_t)frame * (H >> 1) * (W >> 1) * 3 : nullptr, TRAIN ? mag_dst + (size_t)frame * px : nullptr, H, W, thr_sq,
                                     s_raw, CqRun{});
}
template <int TH, bool TRAIN>
__global__ __launch_bounds__(256) void k_color_quantize(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                        uint8_t* __restrict__ pyr_dst, float* __restrict__ mag_dst, int H, int W, float thr_sq,
                                                        uint32_t* __restrict__ clear16, int n_frames_x, StreamWait wait) {
  if (LMX_PRIO_QUANT) __builtin_amdgcn_s_setprio(LMX_PRIO_QUANT);
  color_quantize_body<TH, TRAIN>(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), src, dst, pyr_dst, mag_dst, H, W, thr_sq, clear16, n_frames_x, wait);
}

// =========================================================================================================
// a7  quantizedNormals (before medianBlur).  NORMAL_LUT[v3][v2][v1] is data on the bank (include/lmx.h,
// lmx_bank_set_normal_lut): the kernel reads its 8000 entries, already converted to median bins (0 for "no label", k + 1 for
// label 1 << k, ascending label value order), from global memory -- 8 KB, resident in the vector L1 next to the depth tile.
// Index = (v3 * 20 + v2) * 20 + v1 like C lays the array out; v1, v2 = (int)(n * 10 + 10) and v3 = (int)(nz * 20 + 20) lie in
// [0, 20] (|n| <= 1 up to rounding, nz <= 0), a flat index >= 8000 (upstream: out-of-bounds read) gives bin 0.
// =========================================================================================================
// Products of the LSQ: with IntT = int every operand is below 2^23 in magnitude and every product below 2^31 (bounds in the
// comment of depth_bin_at), so the full-rate 24-bit multiplier gives the exact value; long long keeps the generic multiply.
__device__ __forceinline__ int lsq_mul(int a, int b) { return __mul24(a, b); }
__device__ __forceinline__ long long lsq_mul(long long a, long long b) { return a * b; }

// Median bin of the pixel at p1 (which lies inside the r = 5 frame of the image; row stride W) before the median, 0 for far
// pixels and for shadows.  Upstream accumulates in `long`.  With |delta| < difference_threshold <= 200 every intermediate fits
// int32 (|A| <= 150, |b| <= 30*thr = 6000, |ddx| <= 1.8e6 < 2^23, |1150*ddx| <= 2.07e9 < 2^31, det <= 22500, det*d <= 22500*65535
// < 2^31), so IntT = int gives the same values at a fraction of the cost of emulated 64-bit multiplies; larger thresholds use
// long long.
// The int32 form of depth_bin_at, written for gfx950's issue costs (profiles/r04_valu_issue_microbench.txt): the same integers and the same
// floats as the generic form below, fewer and cheaper instructions.
//  * validity of a tap without compare / select: t = delta + thr - 1 must lie in [0, 2 thr - 2]; nm = (t | (2 thr - 2 - t)) >> 31 is -1 for
//    an invalid tap and 0 for a valid one (thr <= 0: the range is empty, nothing is valid, as in the generic form); the masked delta is
//    delta & ~nm (one v_bitop3), the number of valid taps of a group of n is n + sum(nm).
//  * the factors 25, 5 and 1150 of the least-squares solution are pulled out of the products: with a0, a3, a1 the tap counts and B0, B1
//    the delta sums,  det = 625 (a0 a3 - a1^2),  ddx = 125 (a3 B0 - a1 B1),  ddy = 125 (a0 B1 - a1 B0)  ->  nx = 143750 X, ny = 143750 Y,
//    nz = -625 (D d): ten 24-bit multiplies instead of fourteen (three of them 32-bit).  |X|, |Y| <= 8 * 6 * 200 < 2^23, D d <= 36 * 65535 <
//    2^23, and the results are the generic form's 1150 ddx, 1150 ddy, -det d (< 2^31, bounds above).
//  * LMX_DQ_LEAN_NORM: sqrtf and 1.0f / s are LLVM's correctly rounded expansions with the steps that only serve operands outside this
//    kernel's range removed: the sum of squares is an integer-valued float in [1, 2^65) or 0 (returned before), s in [1, 2^33): no
//    denormal pre-scaling of the square root's argument, v_div_scale returns its operands unscaled (VCC clear) and v_div_fixup passes the
//    quotient through.  What is left is the same v_sqrt_f32 + two-sided ulp correction and the same v_rcp_f32 + three Newton steps.
#ifndef LMX_DQ_LEAN
#define LMX_DQ_LEAN 1
#endif
#ifndef LMX_DQ_LEAN_NORM
#define LMX_DQ_LEAN_NORM 1
#endif
#ifndef LMX_DQ_PIPE
#define LMX_DQ_PIPE 1
#endif
// 24-bit multiplies as instructions: __mul24 is a pattern the compiler may (and here does) turn back into the quarter-rate v_mul_lo_u32 or a
// 64-bit v_mad_u64_u32 once it has proved the operands small
__device__ __forceinline__ int mul24_vv(int a, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b)); return r; }
__device__ __forceinline__ int mul24_sv(int k, int b) { int r; asm("v_mul_i32_i24 %0, %1, %2" : "=v"(r) : "s"(k), "v"(b)); return r; }
__device__ __forceinline__ int mad24_vsv(int a, int k, int c) { int r; asm("v_mad_i32_i24 %0, %1, %2, %3" : "=v"(r) : "v"(a), "s"(k), "v"(c)); return r; }
struct DepthTaps { int d; int t[8]; };   // the pixel and its eight neighbours at distance 5: (-5,-5) (-5,0) (-5,5) (0,-5) (0,5) (5,-5) (5,0) (5,5) as (dy, dx)
__device__ __forceinline__ DepthTaps depth_load_taps(const uint16_t* __restrict__ p1, int W) {
  const int r = 5;
  const uint16_t* p0 = p1 - (size_t)r * W;   // three row pointers, column offsets are immediates
  const uint16_t* p2 = p1 + (size_t)r * W;
  DepthTaps tp;
  tp.d = p1[0];
  tp.t[0] = p0[-r]; tp.t[1] = p0[0]; tp.t[2] = p0[r]; tp.t[3] = p1[-r]; tp.t[4] = p1[r]; tp.t[5] = p2[-r]; tp.t[6] = p2[0]; tp.t[7] = p2[r];
  return tp;
}
// Index of the pixel's NORMAL_LUT entry, or -1 where its bin is 0 without a look-up (far pixel, no valid neighbour pair, index past the table).
// Branch-free: every lane runs the whole arithmetic, the cases are folded into the result at the end.
__device__ __forceinline__ int depth_lut_index_lean(const DepthTaps& tp, int distance_threshold, int difference_threshold) {
  const int d = tp.d;
  const int c0 = d - (difference_threshold - 1), K = 2 * difference_threshold - 2;
  int nm[8], md[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const int t = tp.t[k] - c0;
    nm[k] = cq::sign_mask(t | (K - t));
    md[k] = (int)__builtin_amdgcn_bitop3_b32((uint32_t)(tp.t[k] - d), (uint32_t)nm[k], (uint32_t)nm[k], 0x30);   // a & ~b
  }
  const int a0 = 6 + (nm[0] + nm[2] + nm[3] + nm[4] + nm[5] + nm[7]);
  const int a3 = 6 + (nm[0] + nm[1] + nm[2] + nm[5] + nm[6] + nm[7]);
  const int a1 = (nm[0] + nm[7]) - (nm[2] + nm[5]);
  const int B0 = (md[2] + md[4] + md[7]) - (md[0] + md[3] + md[5]);
  const int B1 = (md[5] + md[6] + md[7]) - (md[0] + md[1] + md[2]);
  const int D = mul24_vv(a0, a3) - mul24_vv(a1, a1);
  const int X = mul24_vv(a3, B0) - mul24_vv(a1, B1);
  const int Y = mul24_vv(a0, B1) - mul24_vv(a1, B0);
  float nx = (float)mul24_sv(143750, X);
  float ny = (float)mul24_sv(143750, Y);
  float nz = (float)mul24_sv(-625, mul24_vv(D, d));
  const float ss = nx * nx + ny * ny + nz * nz;   // 0 (-> bin 0: sqrtf(ss) > 0 <=> ss > 0) or an integer-valued float >= 1
#if LMX_DQ_LEAN_NORM
  float s = __builtin_amdgcn_sqrtf(ss);
  {
    const float sd = __int_as_float(__float_as_int(s) - 1), su = __int_as_float(__float_as_int(s) + 1);
    const float rd = __builtin_fmaf(-sd, s, ss), ru = __builtin_fmaf(-su, s, ss);
    s = rd <= 0.0f ? sd : s;
    s = ru > 0.0f ? su : s;
  }
  float inv;
  {
    float rc = __builtin_amdgcn_rcpf(s);
    rc = __builtin_fmaf(__builtin_fmaf(-s, rc, 1.0f), rc, rc);
    float q = rc;                                                  // 1.0f * rc
    q = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), rc, q);
    inv = __builtin_fmaf(__builtin_fmaf(-s, q, 1.0f), rc, q);
  }
#else
  const float s = sqrtf(ss);
  const float inv = 1.0f / s;
#endif
  nx *= inv; ny *= inv; nz *= inv;
  const int v1 = (int)(nx * 10 + 10);
  const int v2 = (int)(ny * 10 + 10);
  const int v3 = (int)(nz * 20 + 20);
  const int idx = mad24_vsv(mad24_vsv(v3, 20, v2), 20, v1);   // v1, v2, v3 in [0, 20] whenever ss > 0 (garbage otherwise: folded away below)
  const bool look = d < distance_threshold && ss > 0 && (unsigned)idx < (unsigned)LMX_NORMAL_LUT_SIZE;
  return look ? idx : -1;
}
__device__ __forceinline__ int depth_bin_at_lean(const uint16_t* __restrict__ p1, int W, int distance_threshold, int difference_threshold,
                                                 const uint8_t* __restrict__ lut_bins) {
  if (!((int)p1[0] < distance_threshold)) return 0;   // far pixel: no neighbour loads (the pipelined interior loop below loads them regardless)
  const int idx = depth_lut_index_lean(depth_load_taps(p1, W), distance_threshold, difference_threshold);
  return idx >= 0 ? lut_bins[idx] : 0;
}

template <typename IntT>
__device__ __forceinline__ int depth_bin_at(const uint16_t* __restrict__ p1, int W, int distance_threshold, int difference_threshold,
                                            const uint8_t* __restrict__ lut_bins) {
  if constexpr (LMX_DQ_LEAN && std::is_same<IntT, int>::value) return depth_bin_at_lean(p1, W, distance_threshold, difference_threshold, lut_bins);
  const int r = 5;
  // three row pointers, column offsets are immediates: 3 address computations for the 9 loads
  const uint16_t* p0 = p1 - (size_t)r * W;
  const uint16_t* p2 = p1 + (size_t)r * W;
  const IntT d = p1[0];
  if (!(d < distance_threshold)) return 0;
  // accumBilateral over the 8 offsets (i, j) in {-5,0,5}^2: f = |delta| < threshold.  Because i, j are +-5 or 0 the sums
  // factor exactly (integers, any evaluation order gives the same value as upstream's running sums):
  //   A0 = 25 * #valid(i != 0)   A3 = 25 * #valid(j != 0)   A1 = 25 * (valid same-sign corners - valid opposite-sign corners)
  //   b0 = 5 * (sum of valid deltas at i = +5  -  at i = -5)     b1 likewise over j
  IntT dl[8];
  dl[0] = (IntT)p0[-r] - d; dl[1] = (IntT)p0[0] - d; dl[2] = (IntT)p0[r] - d;
  dl[3] = (IntT)p1[-r] - d;                          dl[4] = (IntT)p1[r] - d;
  dl[5] = (IntT)p2[-r] - d; dl[6] = (IntT)p2[0] - d; dl[7] = (IntT)p2[r] - d;
  IntT f[8], md[8];
  const IntT thr = difference_threshold;
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    // |delta| < thr  <=>  0 <= delta + thr - 1 < 2 * thr - 1  (one add and one unsigned compare; thr >= 1, else nothing is valid)
    f[k] = (thr > 0 && (unsigned long long)(dl[k] + thr - 1) < (unsigned long long)(2 * thr - 1)) ? 1 : 0;
    md[k] = f[k] ? dl[k] : 0;
  }
  const IntT A0 = 25 * (f[0] + f[2] + f[3] + f[4] + f[5] + f[7]);
  const IntT A3 = 25 * (f[0] + f[1] + f[2] + f[5] + f[6] + f[7]);
  const IntT A1 = 25 * (f[0] + f[7] - f[2] - f[5]);
  const IntT b0 = 5 * ((md[2] + md[4] + md[7]) - (md[0] + md[3] + md[5]));
  const IntT b1 = 5 * ((md[5] + md[6] + md[7]) - (md[0] + md[1] + md[2]));
  const IntT det = lsq_mul(A0, A3) - lsq_mul(A1, A1);
  const IntT ddx = lsq_mul(A3, b0) - lsq_mul(A1, b1);
  const IntT ddy = lsq_mul(A0, b1) - lsq_mul(A1, b0);
  float nx = (float)lsq_mul((IntT)1150, ddx);
  float ny = (float)lsq_mul((IntT)1150, ddy);
  float nz = (float)(-lsq_mul(det, d));
  float s = sqrtf(nx * nx + ny * ny + nz * nz);
  if (!(s > 0)) return 0;
  float inv = 1.0f / s;
  nx *= inv; ny *= inv; nz *= inv;
  const int v1 = (int)(nx * 10 + 10);
  const int v2 = (int)(ny * 10 + 10);
  const int v3 = (int)(nz * 20 + 20);
  const unsigned idx = (unsigned)((v3 * 20 + v2) * 20 + v1);  // a (never occurring) negative index wraps past the table too
  return idx < (unsigned)LMX_NORMAL_LUT_SIZE ? lut_bins[idx] : 0;
}

// the same for any pixel of the image: 0 outside the r = 5 frame (upstream leaves a frame of r (+1 at the far side) unset)
template <typename IntT>
__device__ __forceinline__ int depth_raw_bin(const uint16_t* __restrict__ src, int H, int W, int y, int x, int distance_threshold,
                                             int difference_threshold, const uint8_t* __restrict__ lut_bins) {
  const int r = 5;
  if (!(y >= r && y < H - r - 1 && x >= r && x < W - r - 1)) return 0;
  return depth_bin_at<IntT>(src + (size_t)y * W + x, W, distance_threshold, difference_threshold, lut_bins);
}

// a7 fused: quantizedNormals + medianBlur(5, BORDER_REPLICATE).  Tile = 64 x DQ_TH outputs; the labels before the median
// are computed for the halo-2 region at CLAMPED image coordinates (that is what the replicate border of the median reads)
// and kept in LDS as CUMULATIVE counters: labels take 9 values, a u64 holds nine 6-bit fields, and a pixel of bin b is a 1 in the
// fields b .. 8 (0x0001041041041041 << 6 b; what the shift pushes past field 8 is never looked at).  A 5x5 window is the sum of
// 25 of them: field k = number of pixels with bin <= k (<= 25, stays inside the field) -- the prefix sums a counting median
// needs, without computing them per output pixel (round 4; until then one-hot counters and four shift-adds per pixel).  A thread
// slides the window down its column segment (row sums of 5 pixels, + entering row - leaving row).  Median = first bin whose
// count reaches 13: + 19 sets bit 5 of exactly those fields, field 8 always among them, and the label is 256 >> their number.
constexpr int DQ_TH = 32;  // tile height (multiple of 4): taller tiles recompute fewer halo labels (68x36 per 64x32 outputs)

template <typename IntT>
__device__ __forceinline__ void depth_quantize_body(const uint3 bid, const uint16_t* __restrict__ src, uint8_t* __restrict__ dst, uint8_t* __restrict__ dst_half,
                                                    int H, int W, int distance_threshold, int difference_threshold,
                                                    const uint8_t* __restrict__ lut_bins, uint32_t* __restrict__ clear16, int n_frames_x,
                                                    const StreamWait wait = StreamWait()) {
  constexpr int RW = 64 + 4, RH = DQ_TH + 4, RS = 68;
  constexpr int RPS = DQ_TH / 4;  // output rows per thread (4 row segments of one column)
  constexpr unsigned long long ONES = 0x0001041041041041ull;  // bit 0 of each of the nine 6-bit fields
  constexpr unsigned long long CUM = ONES;                    // a pixel of bin b counts in the fields b .. 8: CUM << 6 b (bits past field 8 are never looked at)
  __shared__ unsigned long long s_oh[RH][RS];
  const int tid = threadIdx.x;
  if (clear16 != nullptr && (bid.x | bid.y | bid.z) == 0) clear_slot_counters(clear16, tid);  // see k_color_quantize
  int tile_x, tile_y, frame;
  if (!tile_of_block(bid, n_frames_x, (W + 63) / 64, (H + DQ_TH - 1) / DQ_TH, tile_x, tile_y, frame)) return;
  // streamed input: labels of the halo-2 region read depth rows y0 - 2 - 5 .. y0 + DQ_TH + 1 + 5 of the tile's frame
  if (wait.flag != nullptr && !stream_wait_rows(wait, (uint32_t)(frame * H + max(0, tile_y * DQ_TH - 7)), (uint32_t)(frame * H + min(H, tile_y * DQ_TH + DQ_TH + 7)))) return;
  const int x0 = tile_x * 64, y0 = tile_y * DQ_TH;
  src += (size_t)frame * H * W;
  dst += (size_t)frame * H * W;
  if (dst_half) dst_half += (size_t)frame * (H >> 1) * (W >> 1);  // a8 fused: the next level's image is dst(2y, 2x)
  if (y0 - 2 >= 5 && y0 + DQ_TH + 1 < H - 6 && x0 - 2 >= 5 && x0 + 65 < W - 6) {
    // interior tile: no clamping, every pixel inside the r = 5 frame; the item index advances by 256 = 3 * 68 + 52, kept as
    // (row, column, pointer) so that no division is left in the loop
    static_assert(RW == 68 && RS == RW, "256 = 3 * RW + 52");
    int ly = tid / RW, lx = tid - ly * RW;
    const uint16_t* p = src + (size_t)(y0 - 2 + ly) * W + (x0 - 2 + lx);
    unsigned long long* q = &s_oh[ly][lx];
    constexpr int NI = (RH * RW + 255) / 256;             // items per thread; only the last one can fall outside the region
    static_assert((NI - 1) * 256 < RH * RW, "every thread's first NI - 1 items are inside");
    if constexpr (LMX_DQ_PIPE && LMX_DQ_LEAN && std::is_same<IntT, int>::value) {
      // Software pipeline: the nine loads of item i + 1 are issued before item i is computed, and item i's table look-up is consumed one
      // iteration later -- one memory round trip per item on the critical path instead of three (pixel, neighbours, table).  What a wave of
      // the one-frame call spends its time on (a tile then has its CU to itself: nothing else hides the latency), and cheaper in the
      // batched launch as well.
      DepthTaps cur = depth_load_taps(p, W);
      unsigned long long* q_prev = q;
      int bin_prev = 0;
#pragma unroll
      for (int it = 0; it < NI; ++it) {
        const int wrap = cq::sign_mask(RW - 52 - 1 - lx);   // -1 iff lx + 52 >= RW (mask arithmetic: no compare + three selects per item)
        lx += 52 + (wrap & -RW);
        ly += 3 - wrap;
        p += 3 * W + 52 + (wrap & (W - RW));
        DepthTaps nxt = {};
        if (it + 2 < NI || (it + 2 == NI && ly < RH)) nxt = depth_load_taps(p, W);
        const int idx = depth_lut_index_lean(cur, distance_threshold, difference_threshold);
        if (it > 0) *q_prev = CUM << (6 * bin_prev);
        bin_prev = lut_bins[idx >= 0 ? idx : LMX_NORMAL_LUT_SIZE];   // unconditional: the entry behind the table is 0 (kNormalBinsDeviceBytes)
        q_prev = q;
        q += 3 * RS + 52;                                   // RS == RW: the same step with and without a wrap
        cur = nxt;
      }
      // the last item: ly has moved one item past it; it was inside iff the row before the last advance was (ly - 3 + wrap < RH is not kept: recompute)
      if ((NI - 1) * 256 + tid < RH * RW) *q_prev = CUM << (6 * bin_prev);
    } else {
#pragma unroll 2
      for (int it = 0; it < NI; ++it) {
        if (ly < RH) *q = CUM << (6 * depth_bin_at<IntT>(p, W, distance_threshold, difference_threshold, lut_bins));
        const int wrap = cq::sign_mask(RW - 52 - 1 - lx);   // -1 iff lx + 52 >= RW (mask arithmetic: no compare + three selects per item)
        lx += 52 + (wrap & -RW);
        ly += 3 - wrap;
        p += 3 * W + 52 + (wrap & (W - RW));
        q += 3 * RS + 52;                                   // RS == RW: the same step with and without a wrap
      }
    }
  } else {
    for (int i = tid; i < RH * RW; i += 256) {
      int ly = i / RW, lx = i - ly * RW;
      int gy = clampi(y0 - 2 + ly, 0, H - 1), gx = clampi(x0 - 2 + lx, 0, W - 1);
      s_oh[ly][lx] = CUM << (6 * depth_raw_bin<IntT>(src, H, W, gy, gx, distance_threshold, difference_threshold, lut_bins));
    }
  }
  __syncthreads();
  const int lx = tid & 63, seg = tid >> 6;
  const int gx = x0 + lx;
  auto row_sum = [&](int k) {
    const unsigned long long* row = &s_oh[seg * RPS + k][lx];
    return row[0] + row[1] + row[2] + row[3] + row[4];
  };
  unsigned long long ring[5];
#pragma unroll
  for (int k = 0; k < 4; ++k) ring[k] = row_sum(k);
  unsigned long long cnt = ring[0] + ring[1] + ring[2] + ring[3];
  static_assert(RPS % 2 == 0 && DQ_TH % 2 == 0, "row parity below");
  const int gy0 = y0 + seg * RPS;
  uint8_t* drow = dst + (size_t)gy0 * W + gx;   // one address computation per thread, rows by constant strides
  const bool half_ok = dst_half != nullptr && !(gx & 1) && (gx >> 1) < (W >> 1);
  uint8_t* hrow = half_ok ? dst_half + (size_t)(gy0 >> 1) * (W >> 1) + (gx >> 1) : nullptr;
#pragma unroll
  for (int j = 0; j < RPS; ++j) {
    ring[(j + 4) % 5] = row_sum(j + 4);
    cnt += ring[(j + 4) % 5];
    // fields whose cumulative count reaches 13 (+ 19 carries into bit 5 of the field): field 8 always does (25 pixels), so the count is 1..9;
    // the median bin is 9 - count, its label 1 << (bin - 1) or 0 for bin 0 = the low byte of 256 >> count
    const int reach = __popcll((cnt + 19ull * ONES) & (ONES << 5));
    const int gy = gy0 + j;
    if (gy < H && gx < W) {
      const uint8_t lab = (uint8_t)(256u >> reach);
      *drow = lab;
      // gy0 is even (RPS and DQ_TH are): even j <=> even row
      if (!(j & 1) && half_ok && (gy >> 1) < (H >> 1)) *hrow = lab;
    }
    drow += W;
    if (j & 1) hrow += (W >> 1);
    cnt -= ring[j % 5];
  }
}
template <typename IntT>
__global__ __launch_bounds__(256) void k_depth_quantize(const uint16_t* __restrict__ src, uint8_t* __restrict__ dst, uint8_t* __restrict__ dst_half,
                                                        int H, int W, int distance_threshold, int difference_threshold,
                                                        const uint8_t* __restrict__ lut_bins, uint32_t* __restrict__ clear16, int n_frames_x) {
  if (LMX_PRIO_QUANT) __builtin_amdgcn_s_setprio(LMX_PRIO_QUANT);
  depth_quantize_body<IntT>(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), src, dst, dst_half, H, W, distance_threshold, difference_threshold, lut_bins, clear16,
                            n_frames_x);
}

// Small batches (one or two frames per call: the reference's own pattern): the depth quantiser of level 0 and the colour quantiser of
// level 1 do not depend on each other (both follow the colour quantiser of level 0) and neither fills the GPU (150 + 75 workgroups for a
// 640x480 frame), so they share ONE launch; a kernel boundary costs this chain ~4 us of its ~60.  Workgroups [0, n_depth) run the depth
// body, the rest the colour body; static LDS of both bodies adds up (~40 KB), which does not matter at this size.
struct SmallQuantArgs {
  const uint16_t* depth; uint8_t* dq; uint8_t* dq_half; int H, W, distance_threshold, difference_threshold; const uint8_t* lut_bins;
  const uint8_t* bgr1; uint8_t* cq1; uint8_t* pyr2; int H1, W1; float thr_sq;
  int n_depth, dtx, dty, ctx, cty;
  StreamWait wait;   // for the depth workgroups
};
template <typename IntT>
__global__ __launch_bounds__(256) void k_small_depth_color(SmallQuantArgs a) {
  if ((int)blockIdx.x < a.n_depth) {
    const int b = (int)blockIdx.x, per = a.dtx * a.dty, f = b / per, t = b - f * per;
    depth_quantize_body<IntT>(make_uint3((unsigned)(t % a.dtx), (unsigned)(t / a.dtx), (unsigned)f), a.depth, a.dq, a.dq_half, a.H, a.W, a.distance_threshold,
                              a.difference_threshold, a.lut_bins, nullptr, 0, a.wait);
  } else {
    const int b = (int)blockIdx.x - a.n_depth, per = a.ctx * a.cty, f = b / per, t = b - f * per;
    color_quantize_body<CQ_TH, false>(make_uint3((unsigned)(t % a.ctx), (unsigned)(t / a.ctx), (unsigned)f), a.bgr1, a.cq1, a.pyr2, nullptr, a.H1, a.W1, a.thr_sq, nullptr, 0);
  }
}


__global__ __launch_bounds__(256) void k_nn_down2(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int Hd, int Wd) {
  const int frame = blockIdx.z;
  src += (size_t)frame * Hd * Wd * 4;
  dst += (size_t)frame * Hd * Wd;
  const int x = blockIdx.x * 64 + (threadIdx.x & 63);
  const int y = blockIdx.y * 4 + (threadIdx.x >> 6);
  if (x >= Wd || y >= Hd) return;
  dst[(size_t)y * Wd + x] = src[(size_t)(2 * y) * (Wd * 2) + 2 * x];
}

// =========================================================================================================
// a10 + a11 + a12  spread(T) -> computeResponseMaps -> linearize, one pass.
// One workgroup = one strip of T image rows (one row of cells) over the full width.
//   s_src : rows y0 .. y0+2T-2, columns 0 .. W+T-2 (zero outside the image: spread only ORs in-bounds pixels)
//   s_h   : horizontal OR over T columns;  s_sp : vertical OR over T rows  (OR is separable)
//   s_tab : for every spread byte v, the 8 responses max(LUT_lo[o][v&15], LUT_hi[o][v>>4]) packed in a u64
// Output: thread handles 4 consecutive cells of one (grid_y, grid_x) row and stores one dword per orientation,
// so a wave writes runs of Wc contiguous bytes into each linear memory.
// =========================================================================================================
__global__ __launch_bounds__(256) void k_spread_linearize(const uint8_t* __restrict__ quant, uint8_t* __restrict__ lm, uint8_t* __restrict__ ls,
                                                          LevelGeom g) {
  extern __shared__ __align__(16) uint8_t smem[];
  const int T = g.T, W = g.W, H = g.H, Wc = g.Wc;
  const int rows_in = 2 * T - 1;
  const int Wp = (W + T - 1 + 3) & ~3;
  unsigned long long* s_tab = reinterpret_cast<unsigned long long*>(smem);  // 256 x 8 B
  uint8_t* s_src = smem + 2048;                                              // rows_in x Wp
  uint8_t* s_h = s_src + rows_in * Wp;                                       // rows_in x W
  uint8_t* s_sp = s_h + rows_in * W;                                         // T x W

  const int tid = threadIdx.x;
  const int cy = blockIdx.x;  // cell row
  const int frame = blockIdx.z;
  quant += (size_t)frame * W * H;
  if (lm) lm += (size_t)frame * g.mod_stride;
  if (ls) ls += (size_t)frame * g.ls_stride;
  const int y0 = cy * T;

  {
    s_tab[tid] = c_resp_tab.v[tid];
  }
  for (int i = tid; i < rows_in * Wp; i += 256) {
    int ly = i / Wp, x = i - ly * Wp;
    int y = y0 + ly;
    s_src[i] = (y < H && x < W) ? quant[(size_t)y * W + x] : (uint8_t)0;
  }
  __syncthreads();
  for (int i = tid; i < rows_in * W; i += 256) {
    int ly = i / W, x = i - ly * W;
    const uint8_t* p = s_src + ly * Wp + x;
    uint8_t v = 0;
    for (int c = 0; c < T; ++c) v |= p[c];
    s_h[i] = v;
  }
  __syncthreads();
  for (int i = tid; i < T * W; i += 256) {
    int ly = i / W, x = i - ly * W;
    uint8_t v = 0;
    for (int r = 0; r < T; ++r) v |= s_h[(ly + r) * W + x];
    s_sp[i] = v;
  }
  __syncthreads();

  const uint32_t cells = g.cells;
  if (ls != nullptr) {  // finer level: only the spread byte per cell, in linearize() order
    const int n = T * T * Wc;
    for (int i = tid; i < n; i += 256) {
      int grid = i / Wc, j = i - grid * Wc;
      int gy = grid / T, gx = grid - gy * T;
      const uint8_t v = s_sp[gy * W + gx + j * T];
      if (g.ls_bands) {   // banded form (LevelGeom): every cell lives in its own band and in the right half of the band before it
        const uint32_t r1 = (uint32_t)(grid * g.Hc + cy) + 1u;
        ls[(uint32_t)(j >> 4) * g.ls_band_stride + r1 * 32u + (uint32_t)(j & 15)] = v;
        ls[j >= 16 ? (uint32_t)((j >> 4) - 1) * g.ls_band_stride + r1 * 32u + 16u + (uint32_t)(j & 15)
                   : (g.ls_bands - 1u) * g.ls_band_stride + (r1 - 1u) * 32u + 16u + (uint32_t)j] = v;
      } else {
        ls[(size_t)grid * cells + (size_t)cy * Wc + j] = v;
      }
    }
    return;
  }
  if ((Wc & 3) == 0) {
    const int groups_per_row = Wc >> 2;
    const int n_groups = T * T * groups_per_row;
    for (int i = tid; i < n_groups; i += 256) {
      int grid = i / groups_per_row, j4 = i - grid * groups_per_row;
      int gy = grid / T, gx = grid - gy * T;
      const uint8_t* sp = s_sp + gy * W + gx + (4 * j4) * T;
      unsigned long long r0 = s_tab[sp[0]], r1 = s_tab[sp[T]], r2 = s_tab[sp[2 * T]], r3 = s_tab[sp[3 * T]];
      uint8_t* out = lm + (size_t)grid * cells + (size_t)cy * Wc + 4 * j4;
#pragma unroll
      for (int o = 0; o < 8; ++o) {
        uint32_t d = (uint32_t)((r0 >> (8 * o)) & 0xff) | ((uint32_t)((r1 >> (8 * o)) & 0xff) << 8) |
                     ((uint32_t)((r2 >> (8 * o)) & 0xff) << 16) | ((uint32_t)((r3 >> (8 * o)) & 0xff) << 24);
        *reinterpret_cast<uint32_t*>(out + (size_t)o * g.ori_stride) = d;
      }
    }
  } else {
    const int n = T * T * Wc;
    for (int i = tid; i < n; i += 256) {
      int grid = i / Wc, j = i - grid * Wc;
      int gy = grid / T, gx = grid - gy * T;
      unsigned long long r = s_tab[s_sp[gy * W + gx + j * T]];
      uint8_t* out = lm + (size_t)grid * cells + (size_t)cy * Wc + j;
#pragma unroll
      for (int o = 0; o < 8; ++o) out[(size_t)o * g.ori_stride] = (uint8_t)(r >> (8 * o));
    }
  }
}

// (row, column) of a flat index that advances by the workgroup size through a [rows][d] array, d uniform but only known at
// run time: `i / d` per element costs ~25 VALU instructions (the generic unsigned division), this costs three.
// The start uses a float reciprocal, exact here: (tid + 0.5) / d is at least 0.5 / d away from an integer and tid, d < 2^16.
struct RowCol {
  int row, col, q, r, d;
  __device__ __forceinline__ RowCol(int tid, int d_) : d(d_) {
    const float inv = 1.0f / (float)d_;
    row = (int)(((float)tid + 0.5f) * inv);
    col = tid - row * d_;
    q = __builtin_amdgcn_readfirstlane((int)(256.5f * inv));  // 256 / d, uniform
    r = 256 - q * d_;
  }
  __device__ __forceinline__ void next() {
    row += q; col += r;
    if (col >= d) { col -= d; ++row; }
  }
};

// Fast variant for compile-time T with W % 4 == 0 and Wc % 4 == 0: the OR passes run on dwords (4 pixels per op).
//   vertical OR first (pure dword ORs down a column of 2T-1 rows), then the horizontal OR with v_alignbyte_b32
//   (bytes x+c .. x+c+3 for c < T come from at most two neighbouring dwords), then the same table/transposition
//   output stage as the generic kernel.
//   lmn != null (coarsest level, Wc % 8 == 0): the eight response maps are written nibble-packed straight away (8 cells ->
//   one dword per orientation); no byte-wide linear memories and no separate packing pass exist in that case.
template <int T>
__device__ __forceinline__ void spread_linearize_t_body(const uint3 bid, const SpreadBatch& batch, const LevelGeom& g, int n_frames_x) {
  // bid.y = modality (all modalities of a level share one launch)
  const uint8_t* __restrict__ quant = batch.quant[bid.y];
  uint8_t* __restrict__ lm = batch.lm[bid.y];
  uint8_t* __restrict__ ls = batch.ls[bid.y];
  uint8_t* __restrict__ lmn = batch.lmn[bid.y];
  extern __shared__ __align__(16) uint8_t smem[];
  constexpr int RI = 2 * T - 1;
  constexpr int ND = (T + 2) / 4 + 2;  // dwords a horizontal window of T shifts can touch
  const int W = g.W, H = g.H, Wc = g.Wc;
  const int W4 = W >> 2;
  const int Wd = W4 + ND;              // dwords per staged row, zero beyond the image
  unsigned long long* s_tab = reinterpret_cast<unsigned long long*>(smem);
  uint32_t* s_src = reinterpret_cast<uint32_t*>(smem + 2048);  // RI x Wd
  uint32_t* s_v = s_src + RI * Wd;                              // T x Wd
  uint32_t* s_sp32 = s_v + T * Wd;                              // T x W4
  const uint8_t* s_sp = reinterpret_cast<const uint8_t*>(s_sp32);

  const int tid = threadIdx.x;
  // All strips of a frame go to ONE XCD (workgroup b runs on XCD b % 8): a strip contributes only Wc/2 (nibbles) or Wc bytes
  // to each of the T*T*8 output rows, so the rows' cache lines are completed by different strips; inside one L2 they merge
  // before they are written back, spread over eight L2s every strip ships partial lines (measured: the stores were 2/3 of
  // the coarsest level's launch).  n_frames_x = 0: plain (cy, frame) grid for small batches.
  int cy, frame;
  if (n_frames_x > 0) {
    const int xcd = bid.x & 7, idx = bid.x >> 3;
    frame = xcd + 8 * (idx / g.Hc);
    cy = idx - (idx / g.Hc) * g.Hc;
    if (frame >= n_frames_x) return;
  } else {
    cy = bid.x;
    frame = bid.z;
  }
  quant += (size_t)frame * W * H;
  if (lm) lm += (size_t)frame * g.mod_stride;
  if (ls) ls += (size_t)frame * g.ls_stride;
  if (lmn) lmn += (size_t)frame * g.nib_mod_stride;
  const int y0 = cy * T;
  if (ls == nullptr) {
    s_tab[tid] = c_resp_tab.v[tid];
  }
  {
    RowCol rc(tid, Wd);
    for (int i = tid; i < RI * Wd; i += 256, rc.next()) {
      const int y = y0 + rc.row;
      s_src[i] = (y < H && rc.col < W4) ? reinterpret_cast<const uint32_t*>(quant + (size_t)y * W)[rc.col] : 0u;
    }
  }
  __syncthreads();
  for (int j = tid; j < Wd; j += 256) {
    uint32_t d[RI];
#pragma unroll
    for (int r = 0; r < RI; ++r) d[r] = s_src[r * Wd + j];
#pragma unroll
    for (int ly = 0; ly < T; ++ly) {
      uint32_t v = d[ly];
#pragma unroll
      for (int r = 1; r < T; ++r) v |= d[ly + r];
      s_v[ly * Wd + j] = v;
    }
  }
  __syncthreads();
  RowCol rh(tid, W4);
  for (int i = tid; i < T * W4; i += 256, rh.next()) {
    const uint32_t* p = s_v + rh.row * Wd + rh.col;
    uint32_t d[ND];
#pragma unroll
    for (int q = 0; q < ND; ++q) d[q] = p[q];
    uint32_t out = d[0];
#pragma unroll
    for (int c = 1; c < T; ++c) {
      const int q = c >> 2, sh = c & 3;
      out |= sh ? __builtin_amdgcn_alignbyte(d[q + 1], d[q], sh) : d[q];
    }
    s_sp32[i] = out;
  }
  __syncthreads();
  const uint32_t cells = g.cells;
  const int groups_per_row = Wc >> 2;
  const int n_groups = T * T * groups_per_row;
  if (ls != nullptr) {  // finer level: one dword = the spread bytes of 4 consecutive cells
    RowCol ro(tid, groups_per_row);
    uint8_t* ls_row = ls + (size_t)cy * Wc;
    for (int i = tid; i < n_groups; i += 256, ro.next()) {
      const int grid = ro.row, j4 = ro.col;
      const int gy = grid / T, gx = grid - gy * T;
      const uint8_t* sp = s_sp + gy * W + gx + (4 * j4) * T;
      const uint32_t d = (uint32_t)sp[0] | ((uint32_t)sp[T] << 8) | ((uint32_t)sp[2 * T] << 16) | ((uint32_t)sp[3 * T] << 24);
      if (g.ls_bands) {   // banded form (LevelGeom): the group's own band, and the right half of the band before it
        const uint32_t j = 4u * (uint32_t)j4, r1 = (uint32_t)(grid * g.Hc + cy) + 1u;
        *reinterpret_cast<uint32_t*>(ls + (j >> 4) * g.ls_band_stride + r1 * 32u + (j & 15u)) = d;
        *reinterpret_cast<uint32_t*>(ls + (j >= 16u ? ((j >> 4) - 1u) * g.ls_band_stride + r1 * 32u + 16u + (j & 15u)
                                                   : (g.ls_bands - 1u) * g.ls_band_stride + (r1 - 1u) * 32u + 16u + j)) = d;
      } else {
        *reinterpret_cast<uint32_t*>(ls_row + (uint32_t)grid * cells + 4 * j4) = d;
      }
    }
    return;
  }
  if (lmn != nullptr) {  // coarsest level, nibble-packed: 8 consecutive cells -> one dword per orientation
    // Consecutive lanes take consecutive 8-cell groups of one (gy, gx) row, so a wave's stores form runs of Wc/2 bytes per row
    // (the reverse order, neighbouring gx in neighbouring lanes, avoids the LDS bank conflicts of the byte reads below but
    // scatters every store over 64 rows and measured 25 % slower).  The 8 table entries (8 orientations x 8 bits each,
    // values 0..4) are paired into nibbles with one shift-or per half and transposed with v_perm_b32.
    const int groups8 = Wc >> 3;
    RowCol ro(tid, groups8);
    for (int i = tid; i < T * T * groups8; i += 256, ro.next()) {
      const int grid = ro.row, j8 = ro.col;
      const int gy = grid / T, gx = grid - gy * T;
      const uint8_t* sp = s_sp + gy * W + gx + (8 * j8) * T;
      uint32_t lo[4], hi[4];  // pair p: orientation bytes, cell 2p in the low nibble, cell 2p+1 in the high nibble
#pragma unroll
      for (int pr = 0; pr < 4; ++pr) {
        const unsigned long long ra = s_tab[sp[(2 * pr) * T]], rb = s_tab[sp[(2 * pr + 1) * T]];
        lo[pr] = (uint32_t)ra | ((uint32_t)rb << 4);
        hi[pr] = (uint32_t)(ra >> 32) | ((uint32_t)(rb >> 32) << 4);
      }
      uint8_t* out = lmn + (((size_t)grid * cells + (size_t)cy * Wc) >> 1) + 4 * j8;
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const uint32_t* x = h ? hi : lo;  // 4 x 4 byte transpose: out dword o = (x0.o, x1.o, x2.o, x3.o)
        const uint32_t a = __builtin_amdgcn_perm(x[1], x[0], 0x05010400u), b = __builtin_amdgcn_perm(x[1], x[0], 0x07030602u);
        const uint32_t c2 = __builtin_amdgcn_perm(x[3], x[2], 0x05010400u), d2 = __builtin_amdgcn_perm(x[3], x[2], 0x07030602u);
        *reinterpret_cast<uint32_t*>(out + (size_t)(4 * h + 0) * g.nib_ori_stride) = __builtin_amdgcn_perm(c2, a, 0x05040100u);
        *reinterpret_cast<uint32_t*>(out + (size_t)(4 * h + 1) * g.nib_ori_stride) = __builtin_amdgcn_perm(c2, a, 0x07060302u);
        *reinterpret_cast<uint32_t*>(out + (size_t)(4 * h + 2) * g.nib_ori_stride) = __builtin_amdgcn_perm(d2, b, 0x05040100u);
        *reinterpret_cast<uint32_t*>(out + (size_t)(4 * h + 3) * g.nib_ori_stride) = __builtin_amdgcn_perm(d2, b, 0x07060302u);
      }
    }
    return;
  }
  for (int i = tid; i < n_groups; i += 256) {
    int grid = i / groups_per_row, j4 = i - grid * groups_per_row;
    int gy = grid / T, gx = grid - gy * T;
    const uint8_t* sp = s_sp + gy * W + gx + (4 * j4) * T;
    unsigned long long r0 = s_tab[sp[0]], r1 = s_tab[sp[T]], r2 = s_tab[sp[2 * T]], r3 = s_tab[sp[3 * T]];
    uint8_t* out = lm + (size_t)grid * cells + (size_t)cy * Wc + 4 * j4;
#pragma unroll
    for (int o = 0; o < 8; ++o) {
      uint32_t dd = (uint32_t)((r0 >> (8 * o)) & 0xff) | ((uint32_t)((r1 >> (8 * o)) & 0xff) << 8) |
                    ((uint32_t)((r2 >> (8 * o)) & 0xff) << 16) | ((uint32_t)((r3 >> (8 * o)) & 0xff) << 24);
      *reinterpret_cast<uint32_t*>(out + (size_t)o * g.ori_stride) = dd;
    }
  }
}
template <int T>
__global__ __launch_bounds__(256) void k_spread_linearize_t(SpreadBatch batch, LevelGeom g, int n_frames_x) {
  if (LMX_PRIO_SPREAD) __builtin_amdgcn_s_setprio(LMX_PRIO_SPREAD);
  spread_linearize_t_body<T>(make_uint3(blockIdx.x, blockIdx.y, blockIdx.z), batch, g, n_frames_x);
}
// Small batches: both pyramid levels of a two-level bank in ONE launch (see k_small_depth_color): workgroups [0, n0) spread level 0,
// the rest the coarsest level.  Dynamic LDS = the larger of the two bodies' needs.
template <int T0, int T1>
__global__ __launch_bounds__(256) void k_small_spread(SpreadBatch b0, LevelGeom g0, SpreadBatch b1, LevelGeom g1, int n0, int n_mod) {
  if ((int)blockIdx.x < n0) {
    const int b = (int)blockIdx.x, per = g0.Hc * n_mod, f = b / per, t = b - f * per;
    spread_linearize_t_body<T0>(make_uint3((unsigned)(t % g0.Hc), (unsigned)(t / g0.Hc), (unsigned)f), b0, g0, 0);
  } else {
    const int b = (int)blockIdx.x - n0, per = g1.Hc * n_mod, f = b / per, t = b - f * per;
    spread_linearize_t_body<T1>(make_uint3((unsigned)(t % g1.Hc), (unsigned)(t / g1.Hc), (unsigned)f), b1, g1, 0);
  }
}

// =========================================================================================================
// Nibble packing of the coarsest level's linear memories (responses are 0..4): halves the bytes k_score_coarse has to
// pull through the vector cache, which is what bounds it.  byte i = elem(2i) | elem(2i + 1) << 4, elem = 0 past the
// orientation's matrix.  One thread packs 8 consecutive elements (two aligned dwords -> one dword).
// =========================================================================================================
__global__ __launch_bounds__(256) void k_pack_nibbles(const uint8_t* __restrict__ lm, uint8_t* __restrict__ lmn, LevelGeom g) {
  const int frame = blockIdx.z, ori = blockIdx.y;
  const uint32_t n_elem = (uint32_t)g.T * g.T * g.cells;
  const uint32_t t = blockIdx.x * 256 + threadIdx.x;  // group of 8 elements
  const uint32_t e0 = t * 8;
  if (e0 >= n_elem + 8) return;
  const uint8_t* src = lm + (size_t)frame * g.mod_stride + (size_t)ori * g.ori_stride;
  uint8_t* dst = lmn + (size_t)frame * g.nib_mod_stride + (size_t)ori * g.nib_ori_stride;
  // the zero pad behind the matrix makes reads past n_elem return 0, which is what the flat-array semantics prescribe
  const uint2 w = *reinterpret_cast<const uint2*>(src + e0);
  const uint32_t even = ((w.x & 0xffu)) | ((w.x >> 8) & 0xff00u) | ((w.y & 0xffu) << 16) | ((w.y << 8) & 0xff000000u);   // elems 0,2,4,6
  const uint32_t odd = ((w.x >> 8) & 0xffu) | ((w.x >> 16) & 0xff00u) | ((w.y << 8) & 0xff0000u) | (w.y & 0xff000000u);  // elems 1,3,5,7
  *reinterpret_cast<uint32_t*>(dst + t * 4) = even | (odd << 4);
}

// =========================================================================================================
// a13 + a14 + a15  similarity + addSimilarities + coarse threshold scan.
// One wave per (frame, template).  A lane's dword of nibble-packed memory holds 8 consecutive placements, so one
// wave-load covers 512 placements.  Three features are summed per nibble (3 * 4 = 12 <= 15, no carry), then widened
// into two packed-byte accumulators (even / odd placements): byte sums are <= 63 * 4 = 252, so 32-bit adds of packed
// bytes never carry either; widening to u16 happens once per modality (addSimilarities).
// Feature offsets are wave-uniform: lane f loads table entry f (one coalesced 256-B load) and v_readlane broadcasts
// it into an SGPR, so the data loads of several feature groups are in flight before the first add.
// =========================================================================================================
constexpr int SC_WAVES_PER_BLOCK = 4;
constexpr int SC_GROUP = 3;           // features summed in the nibble domain
constexpr int SC_GU = 4;              // groups whose loads are issued back to back
constexpr int SC_CHUNK_LANES = 63;    // lanes of a chunk that own placements (lane 63 only supplies lane 62's neighbour dword)
constexpr int SC_CHUNK_POS = SC_CHUNK_LANES * 8;  // 504 placements per chunk

struct ScoreParams {
  const TemplateInfo* info;
  const TemplateLevelInfo* linfo;  // [G][L]
  const uint32_t* coarse_off;      // [G][M][kFeatStride] nibble-packed offsets
  const uint32_t* uni_off;         // [G][kFeatStride] unified modality-interleaved table (k_score_coarse_u8), or null
  const uint32_t* blk_off;         // [G][SB_MAX_BLOCKS][SB_BLOCK] scalar-block table (k_score_coarse_sb), or null
  const ScoreInfo* sinfo;          // [G]
  const uint8_t* feat_count_coarse;  // [G][M] features per (template, modality) at the coarsest level
  const int32_t* class_slot;       // [n_classes] -> slot or -1
  const uint8_t* lm[kMaxModalities];
  uint32_t mod_stride;             // nib_mod_stride
  int32_t G, L, M;
  int32_t nf_max;
  int32_t n_frames, blocks_per_frame, xcd_frames;
  float threshold;
  Candidate* cands;      // striped list: kCandStripes stripes of cap / kCandStripes entries, then the spill region (cap entries)
  uint32_t* stripes;     // kCandStripes + 1 counters, kStripeWords apart
  uint32_t cap;
  uint32_t n_stripes;    // stripes in use (power of two <= kCandStripes); the spill counter is always counter kCandStripes
};

// The passing placements of one chunk (bit q of pass_mask: the lane's placement j0 + q, raw sum raw8[q]) join the candidate list with
// ONE reservation for the whole wave (see lmx_internal.hpp: a reservation is an atomic that serialises per 128-byte line, so they are
// few and spread over kCandStripes lines).  `seq` numbers the wave's chunks and rotates the stripe, so that a template which passes
// at thousands of placements does not fill one stripe.  Order inside the list never mattered: k_refine's records carry their own
// order key.
__device__ __forceinline__ void append_candidates(const ScoreParams& p, int g, int frame, int lane, int j0, const uint32_t (&raw8)[8], uint32_t pass_mask,
                                                  uint32_t seq) {
  const uint32_t cnt = (uint32_t)__popc(pass_mask);
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t t = (uint32_t)__shfl_up((int)incl, d, 64);
    if (lane >= d) incl += t;
  }
  const uint32_t total = (uint32_t)__builtin_amdgcn_readlane((int)incl, 63);
  if (total == 0) return;
  const uint32_t sc = p.cap / p.n_stripes;
  const uint32_t stripe = ((uint32_t)g + 5u * (uint32_t)frame + seq) & (p.n_stripes - 1u);
  uint32_t base = 0;
  if (lane == 0) base = atomicAdd(p.stripes + (size_t)stripe * kStripeWords, total);
  base = (uint32_t)__builtin_amdgcn_readfirstlane((int)base);
  const uint32_t fit = base < sc ? min(total, sc - base) : 0u;   // the first `fit` of the wave's candidates stay in the stripe
  uint32_t sbase = 0;
  if (fit < total) {
    if (lane == 0) sbase = atomicAdd(p.stripes + (size_t)kCandStripes * kStripeWords, total - fit);
    sbase = (uint32_t)__builtin_amdgcn_readfirstlane((int)sbase);
  }
  uint32_t my = incl - cnt;
#pragma unroll
  for (int q = 0; q < 8; ++q)
    if ((pass_mask >> q) & 1u) {
      size_t slot;
      bool ok = true;
      if (my < fit) slot = (size_t)stripe * sc + base + my;
      else {
        const uint32_t o = sbase + (my - fit);
        ok = o < p.cap;                                           // only when more than cap candidates exist: the batch fails anyway
        slot = (size_t)p.n_stripes * sc + o;
      }
      if (ok) {
        Candidate c;
        c.g = (uint32_t)g; c.pos = (uint32_t)(j0 + q); c.raw = raw8[q]; c.frame = (uint32_t)frame;
        p.cands[slot] = c;
      }
      my += 1;
    }
}

// GU groups of SC_GROUP features: all GU * SC_GROUP * NCH dword loads are issued before the first add.
// A feature's placement run starts at an arbitrary element (nibble) e0, and misaligned dword loads cost the vector cache
// ~35 % here, so the loads are made 4-byte aligned: lane l of chunk k loads aligned dword (63k + l) of the run, fetches its
// right neighbour's dword with a DPP wave shift and funnel-shifts its own 8 nibbles out with v_alignbit_b32 (the shift,
// 4 * (e0 & 7) bits, is wave-uniform).  Lane 63 only feeds lane 62, hence 63 dwords = 504 placements per chunk.
__device__ __forceinline__ uint32_t shifted_dword(uint32_t d, uint32_t shift_bits) {
  // bound_ctrl: lane 63 (no source lane) reads 0, so no "old" value has to be materialised
  const uint32_t nxt = (uint32_t)__builtin_amdgcn_mov_dpp((int)d, 0x130 /* wave_shl:1 */, 0xf, 0xf, true);
  return __builtin_amdgcn_alignbit(nxt, d, shift_bits);  // ({nxt, d} >> shift_bits)[31:0], shift = 4 * (e0 & 7)
}

template <int NCH, int GU>
__device__ __forceinline__ void score_groups(const uint8_t* lm_lane, uint32_t my_off, int grp, uint32_t (&acc_lo)[NCH], uint32_t (&acc_hi)[NCH]) {
  uint32_t v[GU][SC_GROUP][NCH];
  uint32_t sh[GU][SC_GROUP];
#pragma unroll
  for (int a = 0; a < GU; ++a)
#pragma unroll
    for (int u = 0; u < SC_GROUP; ++u) {
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (grp + a) * SC_GROUP + u);
      sh[a][u] = (off & 7u) * 4u;
      const uint32_t* src = reinterpret_cast<const uint32_t*>(lm_lane + ((off >> 3) << 2));
#pragma unroll
      for (int k = 0; k < NCH; ++k) v[a][u][k] = src[k * SC_CHUNK_LANES];
    }
#pragma unroll
  for (int a = 0; a < GU; ++a)
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      const uint32_t nib = shifted_dword(v[a][0][k], sh[a][0]) + shifted_dword(v[a][1][k], sh[a][1]) + shifted_dword(v[a][2][k], sh[a][2]);
      acc_lo[k] += nib & 0x0f0f0f0fu;
      acc_hi[k] += (nib >> 4) & 0x0f0f0f0fu;
    }
}

// Exact pruning.  After some features, a placement whose partial sum S satisfies S + 4 * remaining <= raw_threshold can
// never pass the strict '>' test (a feature adds at most 4), so it needs S >= need := raw_threshold + 1 - 4 * remaining.
// If no placement of the wave's pass reaches `need`, the remaining loads are skipped: the emitted candidates are
// unchanged for every input, only the work is data dependent (at the reference's thresholds of 92-94 nearly every
// (template, frame) pair dies after the first 12 features).
template <int NCH>
__device__ __forceinline__ bool score_alive(const uint32_t (&tot)[NCH][4], const uint32_t (&acc_lo)[NCH], const uint32_t (&acc_hi)[NCH], int need) {
  if (need <= 0) return true;
  const uint32_t bias = (uint32_t)(0x8000 - need) * 0x00010001u;  // u16 halves: (S + bias) has bit 15 set  <=>  S >= need
  uint32_t hit = 0;
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    hit |= tot[k][0] + (acc_lo[k] & 0x00ff00ffu) + bias;
    hit |= tot[k][1] + ((acc_lo[k] >> 8) & 0x00ff00ffu) + bias;
    hit |= tot[k][2] + (acc_hi[k] & 0x00ff00ffu) + bias;
    hit |= tot[k][3] + ((acc_hi[k] >> 8) & 0x00ff00ffu) + bias;
  }
  return __any(((hit & 0x80008000u) != 0) && ((threadIdx.x & 63) < SC_CHUNK_LANES)) != 0;
}

// PRUNE = false (LMX_SCORE_NO_PRUNE, a measurement switch): the early exits are compiled out and every feature of every placement is read
// and added, similarity()'s full work -- the same candidates by construction, since the exits only ever skip work that cannot
// change the outcome.  Gives the kernel's data-INDEPENDENT cost next to the data-dependent one of the shipped configuration.
template <int NCH, bool PRUNE>
__device__ __forceinline__ void score_pass(const ScoreParams& p, int g, int frame, int lane, int pbase, int positions, int raw_threshold,
                                           int nf_total) {
  uint32_t tot[NCH][4];
#pragma unroll
  for (int k = 0; k < NCH; ++k)
#pragma unroll
    for (int q = 0; q < 4; ++q) tot[k][q] = 0;
  const int n_groups = (p.nf_max + SC_GROUP - 1) / SC_GROUP;  // table rows are padded to 64 entries with zero-run offsets
  int consumed = 0;                                            // features of the modalities already folded into tot
  for (int m = 0; m < p.M; ++m) {
    const uint8_t* lm = p.lm[m] + (size_t)frame * p.mod_stride + (pbase >> 1) + lane * 4;
    const uint32_t my_off = p.coarse_off[((size_t)g * p.M + m) * kFeatStride + lane];
    const int nf_m = p.feat_count_coarse[(size_t)g * p.M + m];
    uint32_t acc_lo[NCH], acc_hi[NCH];
#pragma unroll
    for (int k = 0; k < NCH; ++k) acc_lo[k] = acc_hi[k] = 0;
    int grp = 0;
    for (; grp + SC_GU <= n_groups; grp += SC_GU) {
      score_groups<NCH, SC_GU>(lm, my_off, grp, acc_lo, acc_hi);
      const int processed = consumed + min(nf_m, (grp + SC_GU) * SC_GROUP);
      if (PRUNE && !score_alive<NCH>(tot, acc_lo, acc_hi, raw_threshold + 1 - 4 * (nf_total - processed))) return;
    }
    for (; grp < n_groups; ++grp) score_groups<NCH, 1>(lm, my_off, grp, acc_lo, acc_hi);
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      tot[k][0] += acc_lo[k] & 0x00ff00ffu;         // placements 0, 4 of the lane's 8
      tot[k][1] += (acc_lo[k] >> 8) & 0x00ff00ffu;  // placements 2, 6
      tot[k][2] += acc_hi[k] & 0x00ff00ffu;         // placements 1, 5
      tot[k][3] += (acc_hi[k] >> 8) & 0x00ff00ffu;  // placements 3, 7
      acc_lo[k] = acc_hi[k] = 0;
    }
    consumed += nf_m;
    if (PRUNE && m + 1 < p.M && !score_alive<NCH>(tot, acc_lo, acc_hi, raw_threshold + 1 - 4 * (nf_total - consumed))) return;
  }
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    const int j0 = pbase + (k * SC_CHUNK_LANES + lane) * 8;
    const uint32_t raw8[8] = {tot[k][0] & 0xffffu, tot[k][2] & 0xffffu, tot[k][1] & 0xffffu, tot[k][3] & 0xffffu,
                              tot[k][0] >> 16,     tot[k][2] >> 16,     tot[k][1] >> 16,     tot[k][3] >> 16};
    uint32_t pass = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q)
      if (lane < SC_CHUNK_LANES && j0 + q < positions && (int)raw8[q] > raw_threshold) pass |= 1u << q;
    append_candidates(p, g, frame, lane, j0, raw8, pass, (uint32_t)(pbase / SC_CHUNK_POS + k));
  }
}

template <bool PRUNE>
__global__ __launch_bounds__(64 * SC_WAVES_PER_BLOCK) void k_score_coarse(ScoreParams p) {
  const int lane = threadIdx.x & 63;
  // Workgroups are dealt round-robin over the 8 XCDs (blockIdx % 8 share an XCD and its private 4 MiB L2).  With >= 8
  // frames in the batch each XCD scores its own frames (k, k+8, ...): a frame's nibble-packed memories (1.3 MB) are
  // then fetched into ONE L2 instead of all eight.  Placement only changes speed, never results.
  int frame, tblock;
  if (p.xcd_frames) {
    const int k = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    frame = k + 8 * (sidx / p.blocks_per_frame);
    tblock = sidx % p.blocks_per_frame;
    if (frame >= p.n_frames) return;
  } else {
    frame = blockIdx.x / p.blocks_per_frame;
    tblock = blockIdx.x % p.blocks_per_frame;
  }
  const int g = __builtin_amdgcn_readfirstlane(tblock * SC_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (g >= p.G) return;
  if (p.class_slot[p.info[g].class_index] < 0) return;
  const TemplateLevelInfo li = p.linfo[(size_t)g * p.L + (p.L - 1)];
  const int positions = li.positions;
  const int nf = li.nf_total;
  if (positions <= 0 || nf <= 0) return;
  const int raw_threshold = (int)(2 * nf + (p.threshold / 100.f) * (2 * nf) + 0.5f);
  int pbase = 0;
  for (; pbase + 2 * SC_CHUNK_POS < positions; pbase += 3 * SC_CHUNK_POS) score_pass<3, PRUNE>(p, g, frame, lane, pbase, positions, raw_threshold, nf);
  const int rest = positions - pbase;  // <= 2 chunks here (or <= 0 when the last full pass covered everything)
  if (rest > SC_CHUNK_POS) score_pass<2, PRUNE>(p, g, frame, lane, pbase, positions, raw_threshold, nf);
  else if (rest > 0) score_pass<1, PRUNE>(p, g, frame, lane, pbase, positions, raw_threshold, nf);
}

// ---------------------------------------------------------------------------------------------------------
// k_score_coarse_u8: the same scores for banks whose templates have at most 63 coarsest-level features in total (every
// bank the reference trains: 63 features at level 0 -> 31 per modality at level 1), where a placement's sum (<= 252) fits
// one byte for ALL modalities together.  PMC counters showed k_score_coarse issuing VALU instructions ~87 % of its time
// (7.7 per load), so the lever is executing fewer of them:
//   * one modality-interleaved feature table per template (groups of 3, round robin): whichever modality discriminates in a
//     scene starts pruning after the first groups instead of after all features of the other modality;
//   * the bound test every SC8_GU groups, directly on the packed bytes (no widening to u16);
//   * pruning per 504-placement chunk: a dead chunk's loads and adds are skipped (wave-uniform branches).  Dead lanes of a
//     live chunk keep accumulating and are simply never revived.  (Also tried: redirecting the loads of dead lanes to one
//     address so that the wave touches fewer cache lines -- 8 % slower, the kernel is not bound by lines fetched.)
// Results are identical to k_score_coarse for every input (the bound is exact in any feature order); only the work differs.
// ---------------------------------------------------------------------------------------------------------
constexpr int SC8_GU = 4;  // groups (of 3 features) between two bound tests (3..4 equal within noise, 5+ slower; scalar loads of the table
                           // instead of v_readlane: 7 % slower)

// bit 7 of each byte of the result: byte of acc >= need (1 <= need <= 255, wave-uniform)
__device__ __forceinline__ uint32_t bytes_ge(uint32_t acc, int need) {
  if (need <= 128) {
    const uint32_t t = (acc | 0x80808080u) - (uint32_t)need * 0x01010101u;  // no borrow: every byte is >= 128 >= need
    return (t | acc) & 0x80808080u;
  }
  const uint32_t t = (acc & 0x7f7f7f7fu) + (uint32_t)(256 - need) * 0x01010101u;  // no carry: <= 127 + 127
  return t & acc & 0x80808080u;
}

// One round = GU groups of 3 features.  FAST: the three entries of a group share their nibble shift (the host sorts the
// table that way, see build_device_bank), so the three dwords are added first -- nibble sums <= 12, no carries -- and ONE DPP +
// funnel shift serves the group: 4 instead of 8 instructions in front of the nibble split.
template <int NCH, int GU, bool FAST>
__device__ __forceinline__ void score_round_u8(__amdgpu_buffer_rsrc_t rsrc, uint32_t my_off, int grp, const uint32_t (&lane_off)[NCH],
                                               const bool (&chunk_on)[NCH], uint32_t (&acc_lo)[NCH], uint32_t (&acc_hi)[NCH]) {
  uint32_t v[NCH][GU][SC_GROUP];
  uint32_t sh[GU][SC_GROUP], boff[GU][SC_GROUP];  // wave-uniform (SGPRs)
#pragma unroll
  for (int a = 0; a < GU; ++a)
#pragma unroll
    for (int u = 0; u < SC_GROUP; ++u) {
      const uint32_t off = (uint32_t)__builtin_amdgcn_readlane((int)my_off, (grp + a) * SC_GROUP + u);
      sh[a][u] = off >> 27;            // entry = byte offset | funnel-shift bits << 27 (build_device_bank)
      boff[a][u] = off & 0x07ffffffu;
    }
  // one wave-uniform branch per chunk and phase; all loads of the round are issued before the first add
  // buffer loads: descriptor (the frame's memories, wave-uniform) + lane offset in a VGPR + the feature's offset as the
  // instruction's scalar offset operand -- no address arithmetic at all per load, neither vector nor scalar
#pragma unroll
  for (int k = 0; k < NCH; ++k)
    if (chunk_on[k]) {
#pragma unroll
      for (int a = 0; a < GU; ++a)
#pragma unroll
        for (int u = 0; u < SC_GROUP; ++u) v[k][a][u] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off[k], (int)boff[a][u], 0);
    }
#pragma unroll
  for (int k = 0; k < NCH; ++k)
    if (chunk_on[k]) {
#pragma unroll
      for (int a = 0; a < GU; ++a) {
        const uint32_t nib = FAST ? shifted_dword(v[k][a][0] + v[k][a][1] + v[k][a][2], sh[a][0])
                                  : shifted_dword(v[k][a][0], sh[a][0]) + shifted_dword(v[k][a][1], sh[a][1]) + shifted_dword(v[k][a][2], sh[a][2]);
        acc_lo[k] += nib & 0x0f0f0f0fu;
        acc_hi[k] += (nib >> 4) & 0x0f0f0f0fu;
      }
    }
}

template <int NCH, bool PRUNE>
__device__ __forceinline__ void score_pass_u8(const ScoreParams& p, const uint8_t* lm_frame, uint32_t my_off, int g, int frame, int lane,
                                              int pbase, int positions, int raw_threshold, int nf_total) {
  // raw buffer descriptor over this frame's nibble memories from the pass's first placement on (uniform: built from scalars)
  const unsigned long long wave_base = (unsigned long long)(lm_frame + (pbase >> 1));
  const uint32_t base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_base);
  const uint32_t base_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wave_base >> 32));
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)base_hi << 32) | base_lo), 0, 0x7fffffff,
                                                                  0x00020000 /* 32-bit raw data format */);
  uint32_t acc_lo[NCH], acc_hi[NCH], lane_off[NCH];
  bool alive[NCH], chunk_on[NCH];
  auto refresh = [&]() {
    bool any = false;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      chunk_on[k] = __any(alive[k]) != 0;
      any = any || chunk_on[k];
    }
    return any;
  };
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    acc_lo[k] = acc_hi[k] = 0;
    alive[k] = lane < SC_CHUNK_LANES && pbase + (k * SC_CHUNK_LANES + lane) * 8 < positions;
    lane_off[k] = (uint32_t)(k * SC_CHUNK_LANES + lane) * 4u;
  }
  if (!refresh()) return;
  const uint32_t gcount = (uint32_t)__builtin_amdgcn_readlane((int)my_off, kFeatStride - 1);
  const int n_fast = (int)(gcount & 0xffu), n_groups = (int)((gcount >> 8) & 0xffu);  // padding entries read a zero run
  auto prune = [&](int processed) {
    const int need = raw_threshold + 1 - 4 * (nf_total - processed);
    if (!PRUNE || need <= 0) return true;
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      if (chunk_on[k]) alive[k] = alive[k] && ((bytes_ge(acc_lo[k], need) | bytes_ge(acc_hi[k], need)) != 0);
    return refresh();
  };
  int grp = 0;
  for (; grp + SC8_GU <= n_fast; grp += SC8_GU) {
    score_round_u8<NCH, SC8_GU, true>(rsrc, my_off, grp, lane_off, chunk_on, acc_lo, acc_hi);
    if (!prune(min(nf_total, (grp + SC8_GU) * SC_GROUP))) return;
  }
  for (; grp < n_fast; ++grp) score_round_u8<NCH, 1, true>(rsrc, my_off, grp, lane_off, chunk_on, acc_lo, acc_hi);
  if (!prune(min(nf_total, grp * SC_GROUP))) return;
  for (; grp + SC8_GU <= n_groups; grp += SC8_GU) {
    score_round_u8<NCH, SC8_GU, false>(rsrc, my_off, grp, lane_off, chunk_on, acc_lo, acc_hi);
    if (!prune(min(nf_total, (grp + SC8_GU) * SC_GROUP))) return;
  }
  for (; grp < n_groups; ++grp) score_round_u8<NCH, 1, false>(rsrc, my_off, grp, lane_off, chunk_on, acc_lo, acc_hi);
  if (!prune(nf_total)) return;
  // after the last test (need = raw_threshold + 1) a lane is alive iff one of its placements passes
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    if (!chunk_on[k]) continue;   // uniform: the append below is a wave-wide operation (dead lanes pass nothing)
    const int j0 = pbase + (k * SC_CHUNK_LANES + lane) * 8;
    uint32_t raw8[8], pass = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      raw8[q] = (((q & 1) ? acc_hi[k] : acc_lo[k]) >> (8 * (q >> 1))) & 0xffu;  // nibble q of the lane's dword
      if (alive[k] && j0 + q < positions && (int)raw8[q] > raw_threshold) pass |= 1u << q;
    }
    append_candidates(p, g, frame, lane, j0, raw8, pass, (uint32_t)(pbase / SC_CHUNK_POS + k));
  }
}

template <bool PRUNE>
__global__ __launch_bounds__(64 * SC_WAVES_PER_BLOCK) void k_score_coarse_u8(ScoreParams p) {
  const int lane = threadIdx.x & 63;
  int frame, tblock;  // XCD-aware frame placement, as in k_score_coarse
  if (p.xcd_frames) {
    const int k = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    frame = k + 8 * (sidx / p.blocks_per_frame);
    tblock = sidx % p.blocks_per_frame;
    if (frame >= p.n_frames) return;
  } else {
    frame = blockIdx.x / p.blocks_per_frame;
    tblock = blockIdx.x % p.blocks_per_frame;
  }
  const int g = __builtin_amdgcn_readfirstlane(tblock * SC_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (g >= p.G) return;
  // A wave lives ~5 us, so the chain of dependent loads in front of the first data load matters: ONE scalar 16-byte load has
  // everything about the template, the table row is fetched at the same time, and only the class filter is a second hop.
  const ScoreInfo si = p.sinfo[g];
  const uint32_t my_off = p.uni_off[(size_t)g * kFeatStride + lane];
  if (p.class_slot[si.class_index] < 0) return;
  const int positions = si.positions;
  const int nf = si.nf_total;
  if (positions <= 0 || nf <= 0) return;
  const int raw_threshold = (int)(2 * nf + (p.threshold / 100.f) * (2 * nf) + 0.5f);
  const uint8_t* lm_frame = p.lm[0] + (size_t)frame * p.mod_stride;
  // at most two chunks (1008 placements) per pass: the load buffer of a round is 12 dwords per chunk, and a third chunk
  // would cost a wave of occupancy
  int pbase = 0;
  for (; pbase + SC_CHUNK_POS < positions; pbase += 2 * SC_CHUNK_POS) score_pass_u8<2, PRUNE>(p, lm_frame, my_off, g, frame, lane, pbase, positions, raw_threshold, nf);
  if (pbase < positions) score_pass_u8<1, PRUNE>(p, lm_frame, my_off, g, frame, lane, pbase, positions, raw_threshold, nf);
}

// ---------------------------------------------------------------------------------------------------------
// k_score_coarse_sb ("scalar blocks"): k_score_coarse_u8 with the feature table on the SCALAR side.  PMC + the issue-rate
// microbenchmark of round 2 (profiles/r02a_pmc_summary.txt, profiles/r02_valu_issue_microbench.txt) showed the u8 kernel short of
// both issue ports at once: per wave 402 VALU (v_readlane, DPP, v_alignbit, v_add3 issue at ~4.2 cycles, not 2) AND 332 SALU
// (s_add for the lane index, s_and / s_lshr to split every table entry; one scalar instruction per ~4.4 cycles per SIMD).
// Here the table row of a template is a sequence of 16-dword blocks in the constant address space:
//     dword 0..14  byte offsets of 5 groups x 3 features (relative to the frame's memories, modality displacement included)
//     dword 15     bits 0..24: the five groups' funnel shifts (5 bits each), bits 25..31: real features consumed up to here
// One s_load_dwordx16 brings a whole round (15 features) into SGPRs that feed the buffer loads' scalar-offset operand directly:
// no v_readlane, no per-entry scalar arithmetic.  Every group is "fast" (its three dwords share the shift and are added before
// the ONE DPP + funnel shift): the host pads the < 3 leftovers of each shift class with zero-run entries (build_device_bank).
// Same exact pruning, same candidates for every input.
// ---------------------------------------------------------------------------------------------------------
typedef const uint32_t __attribute__((address_space(4))) lmx_cu32_const;   // constant address space: uniform loads become s_load

template <int NCH, bool PRUNE>
__device__ __forceinline__ void score_pass_sb(const ScoreParams& p, const uint8_t* lm_frame, lmx_cu32_const* row, int n_blocks, int g, int frame, int lane,
                                              int pbase, int positions, int raw_threshold, int nf_total) {
  const unsigned long long wave_base = (unsigned long long)(lm_frame + (pbase >> 1));
  const uint32_t base_lo = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)wave_base);
  const uint32_t base_hi = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(wave_base >> 32));
  __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc(reinterpret_cast<void*>(((unsigned long long)base_hi << 32) | base_lo), 0, 0x7fffffff,
                                                                  0x00020000 /* 32-bit raw data format */);
  uint32_t acc_lo[NCH], acc_hi[NCH], lane_off[NCH];
  bool alive[NCH], chunk_on[NCH];
  auto refresh = [&]() {
    bool any = false;
#pragma unroll
    for (int k = 0; k < NCH; ++k) {
      chunk_on[k] = __any(alive[k]) != 0;
      any = any || chunk_on[k];
    }
    return any;
  };
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    acc_lo[k] = acc_hi[k] = 0;
    alive[k] = lane < SC_CHUNK_LANES && pbase + (k * SC_CHUNK_LANES + lane) * 8 < positions;
    lane_off[k] = (uint32_t)(k * SC_CHUNK_LANES + lane) * 4u;
  }
  if (!refresh()) return;
  for (int b = 0; b < n_blocks; ++b) {
    lmx_cu32_const* blk = row + b * SB_BLOCK;
    uint32_t off[SB_BLOCK - 1];
#pragma unroll
    for (int i = 0; i < SB_BLOCK - 1; ++i) off[i] = blk[i];
    const uint32_t meta = blk[SB_BLOCK - 1];
    uint32_t v[NCH][SB_BLOCK - 1];
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      if (chunk_on[k]) {
#pragma unroll
        for (int i = 0; i < SB_BLOCK - 1; ++i) v[k][i] = (uint32_t)__builtin_amdgcn_raw_buffer_load_b32(rsrc, (int)lane_off[k], (int)off[i], 0);
      }
#pragma unroll
    for (int k = 0; k < NCH; ++k)
      if (chunk_on[k]) {
        uint32_t nib[SB_GROUPS];
#pragma unroll
        for (int q = 0; q < SB_GROUPS; ++q) nib[q] = shifted_dword(v[k][3 * q] + v[k][3 * q + 1] + v[k][3 * q + 2], (meta >> (5 * q)) & 31u);
        // nibbles <= 12 each: five of them (<= 60 per byte) are summed before they join the accumulators (<= 252)
        acc_lo[k] += (nib[0] & 0x0f0f0f0fu) + (nib[1] & 0x0f0f0f0fu) + (nib[2] & 0x0f0f0f0fu) + (nib[3] & 0x0f0f0f0fu) + (nib[4] & 0x0f0f0f0fu);
        acc_hi[k] += ((nib[0] >> 4) & 0x0f0f0f0fu) + ((nib[1] >> 4) & 0x0f0f0f0fu) + ((nib[2] >> 4) & 0x0f0f0f0fu) + ((nib[3] >> 4) & 0x0f0f0f0fu) +
                     ((nib[4] >> 4) & 0x0f0f0f0fu);
      }
    const int need = raw_threshold + 1 - 4 * (nf_total - (int)(meta >> 25));
    if (PRUNE && need > 0) {
#pragma unroll
      for (int k = 0; k < NCH; ++k)
        if (chunk_on[k]) alive[k] = alive[k] && ((bytes_ge(acc_lo[k], need) | bytes_ge(acc_hi[k], need)) != 0);
      if (!refresh()) return;
    }
  }
  // the last block's test ran with need = raw_threshold + 1 (every feature consumed): a lane is alive iff one of its placements passes
#pragma unroll
  for (int k = 0; k < NCH; ++k) {
    if (!chunk_on[k]) continue;   // uniform: the append below is a wave-wide operation (dead lanes pass nothing)
    const int j0 = pbase + (k * SC_CHUNK_LANES + lane) * 8;
    uint32_t raw8[8], pass = 0;
#pragma unroll
    for (int q = 0; q < 8; ++q) {
      raw8[q] = (((q & 1) ? acc_hi[k] : acc_lo[k]) >> (8 * (q >> 1))) & 0xffu;  // nibble q of the lane's dword
      if (alive[k] && j0 + q < positions && (int)raw8[q] > raw_threshold) pass |= 1u << q;
    }
    append_candidates(p, g, frame, lane, j0, raw8, pass, (uint32_t)(pbase / SC_CHUNK_POS + k));
  }
}

template <bool PRUNE>
__global__ __launch_bounds__(64 * SC_WAVES_PER_BLOCK) void k_score_coarse_sb(ScoreParams p) {
  if (LMX_PRIO_SCORE) __builtin_amdgcn_s_setprio(LMX_PRIO_SCORE);
  if (LMX_SC_EXIT == 1) return;   // latency experiments (scripts/build_variants.py scexit): where a one-frame launch spends its time
  const int lane = threadIdx.x & 63;
  int frame, tblock;  // XCD-aware frame placement, as in k_score_coarse
  if (p.xcd_frames) {
    const int k = blockIdx.x & 7, sidx = blockIdx.x >> 3;
    frame = k + 8 * (sidx / p.blocks_per_frame);
    tblock = sidx % p.blocks_per_frame;
    if (frame >= p.n_frames) return;
  } else {
    frame = blockIdx.x / p.blocks_per_frame;
    tblock = blockIdx.x % p.blocks_per_frame;
  }
  const int g = __builtin_amdgcn_readfirstlane(tblock * SC_WAVES_PER_BLOCK + (threadIdx.x >> 6));
  if (g >= p.G) return;
  const ScoreInfo si = p.sinfo[g];
  if (p.class_slot[si.class_index] < 0) return;
  const int positions = si.positions;
  const int nf = si.nf_total;
  if (positions <= 0 || nf <= 0) return;
  const int raw_threshold = (int)(2 * nf + (p.threshold / 100.f) * (2 * nf) + 0.5f);
  const uint8_t* lm_frame = p.lm[0] + (size_t)frame * p.mod_stride;
  lmx_cu32_const* row = (lmx_cu32_const*)(uintptr_t)(p.blk_off + (size_t)g * (SB_BLOCK * SB_MAX_BLOCKS));
  const int n_blocks = LMX_SC_EXIT == 3 ? 1 : (int)((si.groups >> 16) & 0xffu);
  if (LMX_SC_EXIT == 2) { const uint32_t r0 = row[0]; if ((uint32_t)positions + r0 == 0xfffffff1u) p.stripes[0] = r0; return; }   // template info, class filter and the first table dword loaded
  int pbase = 0;
  if (LMX_SC_EXIT == 3 || LMX_SC_EXIT == 4) {   // one pass of two chunks only (3: its first block only)
    score_pass_sb<2, PRUNE>(p, lm_frame, row, n_blocks, g, frame, lane, pbase, positions, raw_threshold, nf);
    return;
  }
  for (; pbase + SC_CHUNK_POS < positions; pbase += 2 * SC_CHUNK_POS) score_pass_sb<2, PRUNE>(p, lm_frame, row, n_blocks, g, frame, lane, pbase, positions, raw_threshold, nf);
  if (pbase < positions) score_pass_sb<1, PRUNE>(p, lm_frame, row, n_blocks, g, frame, lane, pbase, positions, raw_threshold, nf);
}

// =========================================================================================================
// a16  pyramid refinement.  One WORKGROUP (four waves) per candidate, each wave a quarter of the features: the 16x16 patch of
// similarityLocal is 256 cells = 4 per lane (lane -> row lane>>2, columns 4*(lane&3)..+3), again as packed u8 sums that meet in
// LDS; argmax with upstream's first-maximum rule by a wave max-reduction over (score << 8 | 255 - cell).
// =========================================================================================================
#ifndef LMX_RF_UNROLL
#define LMX_RF_UNROLL 8
#endif
constexpr int RF_UNROLL = LMX_RF_UNROLL;  // gathers in flight per wave and batch; a wave's share of a modality is 16 (4, 8 and 16 timed the same)

// Response of orientation o to a spread byte v, without a table: with M_k[o] = the set of source bits whose response is >= k
// (nested: M_4 in M_3 in M_2 in M_1, read off SIMILARITY_LUT, asymmetric high nibble included),
//   max(LUT_lo[o][v & 15], LUT_hi[o][v >> 4]) = [v & M_1] + [v & M_2] + [v & M_3] + [v & M_4]     (verified for all 8 x 256 cases).
// Byte k-1 of c_resp_masks[o] is M_k[o].  On four packed spread bytes each indicator is the classic SWAR "byte is non-zero".
__constant__ uint32_t c_resp_masks[8] = {0x0103070fu, 0x02070f1fu, 0x040e1f3fu, 0x081c3e7fu, 0x10387cfeu, 0x2070f8fdu, 0x40e0f1fbu, 0x80c1e3f7u};

// M_4[o] is the single bit o, so the fourth indicator is (v >> o) & 1.  The masks come as VGPRs (from the LDS table s_masks below)
// and the constants as literals: on gfx950 a VALU instruction with an SGPR source issues at half rate, v_bitop3_b32 on three VGPRs
// at full rate (scripts/microbench/valu_issue), and k_refine is bound by exactly this arithmetic on busy scenes.
__device__ __forceinline__ uint32_t byte_nonzero_bit7(uint32_t d, uint32_t m, uint32_t c7) {   // bit 7 of every byte: (d & m) has a bit there
  const uint32_t a = __builtin_amdgcn_bitop3_b32(d, m, c7, 0x80) + 0x7f7f7f7fu;   // a & b & c; c7 = 0x7f7f7f7f in a VGPR (VOP3 takes no literal)
  return __builtin_amdgcn_bitop3_b32(a, d, m, 0xf8);                               // a | (b & c)
}
__device__ __forceinline__ uint32_t response4(uint32_t d, uint4 mk, uint32_t c7) {   // mk = {M_1, M_2, M_3 replicated into every byte, o}
  const uint32_t f1 = (byte_nonzero_bit7(d, mk.x, c7) >> 7) & 0x01010101u, f2 = (byte_nonzero_bit7(d, mk.y, c7) >> 7) & 0x01010101u;
  const uint32_t f3 = (byte_nonzero_bit7(d, mk.z, c7) >> 7) & 0x01010101u, f4 = (d >> mk.w) & 0x01010101u;
  return (f1 + f2) + (f3 + f4);  // four responses 0..4, one per byte
}

struct RefineParams {
  const TemplateInfo* info;
  const TemplateLevelInfo* linfo;
  const FeatEntry* feat;       // [L][G][M][kFeatStride]
  const uint8_t* feat_count;   // [L][G][M]
  const int32_t* class_slot;
  LevelGeom geom[kMaxLevels];
  const uint8_t* ls[kMaxLevels][kMaxModalities];  // linearised spread images of the finer levels
  int32_t G, L, M;
  float threshold;
  const Candidate* cands;
  const uint32_t* stripes;   // the striped candidate list's counters (lmx_internal.hpp)
  uint32_t* header;          // the slot's 16-dword header: word 0 receives the number of candidates
  uint32_t cap;
  uint32_t n_stripes;
  lmx_raw_match_t* matches;
  uint32_t* match_count;
  // read-back folded into this kernel (null = off): the LAST workgroup to finish copies the slot's 64-byte header and the records it
  // counts (<= pub_max) from pub_src to pub_dst (the pinned host mirror, device view); pub_counter counts finished workgroups
  uint4* pub_dst;
  const uint4* pub_src;
  uint32_t* pub_counter;
  uint32_t pub_max;
  uint32_t pub_seq;      // goes to word 7 of the published header once everything else of the slot is visible to the host
};

// One workgroup (4 waves) per candidate: the gathers of a candidate are a dependent chain of batches (table entry ->
// 8 loads in flight -> adds), and with one wave per candidate the kernel's duration was that chain (16 batches per level),
// not its total work.  Wave w takes features [16w, 16w+16) of every modality, the four partial patch sums meet in LDS, and
// every wave then evaluates the same arg-max, which keeps the control flow uniform without a broadcast.
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(8, 8))) void k_refine(RefineParams p) {
  if (LMX_PRIO_REFINE) __builtin_amdgcn_s_setprio(LMX_PRIO_REFINE);
  __shared__ uint32_t s_part[2][4][2][64];  // [parity of the level step][wave][lo, hi][lane]
  __shared__ uint32_t s_mid[2][4][2][64];   // the same for the early-exit test between two modalities
  __shared__ uint4 s_masks[8];              // per orientation: M_1, M_2, M_3 replicated into every byte, and o
  // The candidate list is striped (lmx_internal.hpp): s_start[r] = number of entries in the regions before r (stripes 0 .. K-1, then
  // the spill region); the sum of the stripe counters is the number of candidates the scoring kernel found, which workgroup 0 leaves
  // in header word 0 for publish / export / the host (more than cap = overflow, as with a single list).  All counters, the spill
  // counter included, are requested at once: one memory round trip in front of the first candidate, as with a single counter.
  __shared__ uint32_t s_start[kCandStripes + 2];
  if (threadIdx.x < 64) {
    static_assert(kCandStripes == 64, "one wave scans the stripe counters");
    const uint32_t sc = p.cap / p.n_stripes;
    const uint32_t c_raw = threadIdx.x < p.n_stripes ? p.stripes[(size_t)threadIdx.x * kStripeWords] : 0u;
    const uint32_t spill = p.stripes[(size_t)kCandStripes * kStripeWords];
    if (threadIdx.x < 8) {
      const uint32_t mm = c_resp_masks[threadIdx.x];
      s_masks[threadIdx.x] = make_uint4((mm & 0xffu) * 0x01010101u, ((mm >> 8) & 0xffu) * 0x01010101u, ((mm >> 16) & 0xffu) * 0x01010101u, threadIdx.x);
    }
    uint32_t incl = min(c_raw, sc), tot = c_raw;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t a = (uint32_t)__shfl_up((int)incl, d, 64), b = (uint32_t)__shfl_up((int)tot, d, 64);
      if ((int)threadIdx.x >= d) { incl += a; tot += b; }
    }
    s_start[threadIdx.x + 1] = incl;
    if (threadIdx.x == 0) s_start[0] = 0;
    if (threadIdx.x == 63) {
      s_start[kCandStripes + 1] = incl + min(spill, p.cap);
      if (blockIdx.x == 0) p.header[0] = tot;
    }
  }
  __syncthreads();
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
  const int bp_base = (int)(threadIdx.x >> 6) * 64;   // ds_bpermute byte address of lane 16 * wave (kept in a VGPR)
  uint32_t c7;
  asm volatile("v_mov_b32 %0, 0x7f7f7f7f" : "=v"(c7));   // a VGPR on purpose, see response4
  const uint32_t n = s_start[kCandStripes + 1];
  int region = 0;
  constexpr int RF_REC_BATCH = 16;
  __shared__ lmx_raw_match_t s_rec[RF_REC_BATCH];   // thread 0's records that have no slot yet
  int n_rec_buffered = 0;                           // (meaningful in thread 0 only)
  auto flush_records = [&]() {                      // thread 0: one reservation for everything buffered
    if (n_rec_buffered == 0) return;
    const uint32_t base = atomicAdd(p.match_count, (uint32_t)n_rec_buffered);
    for (int i = 0; i < n_rec_buffered; ++i)
      if (base + (uint32_t)i < p.cap) p.matches[base + (uint32_t)i] = s_rec[i];   // beyond cap: counted, not stored (overflow is reported)
    n_rec_buffered = 0;
  };
  for (uint32_t ci = blockIdx.x; ci < n; ci += gridDim.x) {
    while (ci >= s_start[region + 1]) ++region;   // uniform; ci only grows
    // stripe r starts at r * SC, the spill region (region kCandStripes; regions n_stripes .. kCandStripes - 1 are empty) at n_stripes * SC
    const size_t entry = (size_t)min((uint32_t)region, p.n_stripes) * (p.cap / p.n_stripes) + (ci - s_start[region]);
    // ci is uniform, and so is everything derived from the candidate: say so (readfirstlane), or the compiler treats the patch
    // origin as per-lane data and wraps every gather in an exec-masked branch with a scalar reload inside (round 2: that
    // skeleton alone was 100 of the kernel's 230 us on busy scenes).
    const Candidate c = p.cands[entry];
    const int g = __builtin_amdgcn_readfirstlane((int)c.g);
    const int frame = __builtin_amdgcn_readfirstlane((int)c.frame);
    const uint32_t c_pos = (uint32_t)__builtin_amdgcn_readfirstlane((int)c.pos);
    const int c_raw = __builtin_amdgcn_readfirstlane((int)c.raw);
    const int Lc = p.L - 1;
    const LevelGeom& gc = p.geom[Lc];
    const int offc = gc.T / 2 + (gc.T % 2 - 1);
    int x = (int)(c_pos % (uint32_t)gc.Wc) * gc.T + offc;
    int y = (int)(c_pos / (uint32_t)gc.Wc) * gc.T + offc;
    // Everything that depends only on g is requested together, right behind the candidate itself: the level infos here, the
    // feature-table rows of every modality at the top of the level loop.  The kernel is a chain of dependent memory round trips per
    // candidate (measured in round 2: neither fewer cache lines per patch nor more gathers in flight changed its time), so the
    // rows' feature counts come out of the rows themselves (entry 63 is always padding; its y field carries the count) instead of
    // a second table, and a wave's 16 gathers of a modality are in flight at once.
    const TemplateLevelInfo* lg = p.linfo + (size_t)g * p.L;
    const int nfc = lg[Lc].nf_total;
    float sim = (c_raw * 100.f) / (4 * nfc) + 0.5f;
    bool alive = true;
    int step = 0;
    for (int l = Lc - 1; l >= 0 && alive; --l, ++step) {
      const LevelGeom& gl = p.geom[l];
      const TemplateLevelInfo li = lg[l];
      const FeatEntry* rows = p.feat + (((size_t)l * p.G + g) * p.M) * kFeatStride + lane;
      FeatEntry next_row = rows[0];   // the row of modality m + 1 is requested before the gathers of modality m start
      const int T = gl.T, border = 8 * T, off = T / 2 + (T % 2 - 1);
      const int max_x = gl.W - li.width - border, max_y = gl.H - li.height - border;
      x = x * 2 + 1; y = y * 2 + 1;
      x = max(x, border); y = max(y, border);
      x = min(x, max_x); y = min(y, max_y);
      // C++ integer division truncates toward zero (x may be negative when the template exceeds the image)
      const int ocx = x / T - 8, ocy = y / T - 8;
      const int offset_x = ocx * T, offset_y = ocy * T;
      const int row = lane >> 2, col4 = (lane & 3) * 4;
      const int delta = ocy * gl.Wc + ocx;
      const int lsW = gl.W, lsH = gl.H;
      const uint32_t zero_off = (uint32_t)gl.ls_zero_off;
      const uint32_t lane_off = (uint32_t)(row * (gl.ls_bands ? 32 : gl.Wc) + col4);
      uint32_t tot_lo = 0, tot_hi = 0;
      int seen = 0;   // features summed so far (all modalities up to m)
      for (int m = 0; m < p.M; ++m) {
        // lane f holds feature f's table entry; entries past the count point at the zero pad
        const FeatEntry my = next_row;
        if (m + 1 < p.M) next_row = rows[(size_t)(m + 1) * kFeatStride];
        const int nf = __builtin_amdgcn_readlane((int)(uint16_t)my.y, kFeatStride - 1) & 0xff;
        const uint8_t* ls = p.ls[l][m] + (size_t)frame * gl.ls_stride;   // uniform; a lane's cell is a 32-bit offset on top
        // Lane f prepares feature f once (VALU, all 64 at a time): where its patch starts (upstream skips features that leave the
        // image after the shift; those and the padded entries read the zero pad) and which row of s_masks its orientation uses.
        // The gather loop broadcasts both through the LDS crossbar (ds_bpermute: no VALU slot, and the values arrive in VGPRs).
        const int fx = (int)my.x + offset_x, fy = (int)my.y + offset_y;
        const bool valid = (lane < nf) & (fx >= 0) & (fy >= 0) & (fx < lsW) & (fy < lsH);
        uint32_t my_a = (my.off & 0x1fffffffu) + (uint32_t)delta;       // flat image: upstream's element index, shifted
        if (gl.ls_bands) {                                               // banded image: the band of the patch's first column
          const uint32_t C = (my.off & 0xfffu) + (uint32_t)ocx, R1 = ((my.off >> 12) & 0x1ffffu) + (uint32_t)(ocy + 1);
          my_a = (C >> 4) * gl.ls_band_stride + R1 * 32u + (C & 15u);
        }
        my_a = (valid ? my_a : zero_off) | (my.off & 0xe0000000u);      // the orientation rides in the top three bits
        uint32_t acc = 0;  // this wave's 16 features: sums <= 64 per byte
        if (16 * wave < nf) {
#pragma unroll
          for (int f0 = 0; f0 < 16; f0 += RF_UNROLL) {
            uint32_t v[RF_UNROLL], row_of[RF_UNROLL];
#pragma unroll
            for (int u = 0; u < RF_UNROLL; ++u) {
              const uint32_t a = (uint32_t)__builtin_amdgcn_ds_bpermute(bp_base + 4 * (f0 + u), (int)my_a);
              row_of[u] = (a >> 25) & 0x70u;   // byte offset of the orientation's row in s_masks
              v[u] = load_u32_unaligned(ls + (size_t)((a & 0x1fffffffu) + lane_off));
            }
#pragma unroll
            for (int u = 0; u < RF_UNROLL; ++u)
              acc += response4(v[u], *reinterpret_cast<const uint4*>(reinterpret_cast<const uint8_t*>(s_masks) + row_of[u]), c7);
          }
        }
        tot_lo += acc & 0x00ff00ffu;
        tot_hi += (acc >> 8) & 0x00ff00ffu;
        seen += nf;
        // Exact early exit between two modalities (round 3).  A feature adds at most 4, so the best cell of the patch can end at
        // max(partial) + 4 * (features still to come); when that bound already fails the threshold -- the same float expression as the
        // final test, monotone in its argument -- the candidate is dropped here, as the final test would drop it.  Nine of ten
        // candidates that pass the coarse level at thresholds of 80-88 die at this level: they now cost one modality's gathers.
        if (m + 1 < p.M) {
          uint32_t (*mid)[2][64] = s_mid[m & 1];
          mid[wave][0][lane] = tot_lo;
          mid[wave][1][lane] = tot_hi;
          __syncthreads();
          const uint32_t lo = mid[0][0][lane] + mid[1][0][lane] + mid[2][0][lane] + mid[3][0][lane];
          const uint32_t hi = mid[0][1][lane] + mid[1][1][lane] + mid[2][1][lane] + mid[3][1][lane];
          uint32_t mx = max(max(lo & 0xffffu, lo >> 16), max(hi & 0xffffu, hi >> 16));
#pragma unroll
          for (int sft = 32; sft >= 1; sft >>= 1) mx = max(mx, (uint32_t)__shfl_xor((int)mx, sft, 64));
          const int bound = __builtin_amdgcn_readfirstlane((int)mx) + 4 * (li.nf_total - seen);
          if ((bound * 100.f) / (4 * li.nf_total) < p.threshold) { alive = false; break; }   // every wave computes the same bound
        }
      }
      if (!alive) break;
      // partial sums of the four waves (u16 fields, <= 4 * 63 * M in total)
      uint32_t (*part)[2][64] = s_part[step & 1];
      part[wave][0][lane] = tot_lo;
      part[wave][1][lane] = tot_hi;
      __syncthreads();
      tot_lo = part[0][0][lane] + part[1][0][lane] + part[2][0][lane] + part[3][0][lane];
      tot_hi = part[0][1][lane] + part[1][1][lane] + part[2][1][lane] + part[3][1][lane];
      const uint32_t s4[4] = {tot_lo & 0xffffu, tot_hi & 0xffffu, tot_lo >> 16, tot_hi >> 16};
      uint32_t key = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        uint32_t cell = (uint32_t)(row * 16 + col4 + b);
        uint32_t k = (s4[b] << 8) | (255u - cell);
        key = k > key ? k : key;
      }
#pragma unroll
      for (int sft = 32; sft >= 1; sft >>= 1) {
        uint32_t o = (uint32_t)__shfl_xor((int)key, sft, 64);
        key = o > key ? o : key;
      }
      key = (uint32_t)__builtin_amdgcn_readfirstlane((int)key);   // every lane holds the maximum
      const int best = (int)(key >> 8);
      int best_r = -1, best_c = -1;
      if (best > 0) {
        int cell = 255 - (int)(key & 255u);
        best_r = cell >> 4; best_c = cell & 15;
      }
      x = (x / T - 8 + best_c) * T + off;
      y = (y / T - 8 + best_r) * T + off;
      sim = (best * 100.f) / (4 * li.nf_total);
      if (sim < p.threshold) alive = false;
    }
    if (alive && threadIdx.x == 0) {
      // The record waits in LDS: a workgroup reserves room for up to RF_REC_BATCH of its records with ONE atomic (flush_records).  The
      // record counter is one address, an atomic on it costs 11 ns however many waves queue (profiles/r03_atomic_append_microbench.txt),
      // and where most candidates survive -- low thresholds -- those atomics, not the refinement, were the kernel's time.
      const TemplateInfo ti = p.info[g];
      lmx_raw_match_t mm;
      mm.x = x; mm.y = y; mm.similarity = sim; mm.template_id = ti.template_id; mm.class_index = ti.class_index;
      mm.frame = frame;
      mm.order_key = ((uint64_t)(uint32_t)p.class_slot[ti.class_index] << 48) | ((uint64_t)(uint32_t)ti.template_id << 24) |
                     (uint64_t)c_pos;
      s_rec[n_rec_buffered++] = mm;
      if (n_rec_buffered == RF_REC_BATCH) flush_records();
    }
    // s_part is double-buffered by level step; a new candidate starts again at step 0 while slower waves may still be
    // reading this candidate's last buffer only if that buffer is the one about to be written: separate them
    __syncthreads();
  }
  if (threadIdx.x == 0) flush_records();
  // Read-back without a kernel of its own (k_publish_records used to follow: ~4 us of launch per batch, 6 % of a single-frame
  // call).  Every workgroup takes a ticket when it is done; the one that draws the last ticket sees all records (release fence
  // before the ticket, acquire fence after it) and copies header + counted records through the mapping of the pinned slot.
  // Only workgroups that had a candidate (and workgroup 0, so that an empty list is published too) take part: a ticket is an atomic
  // on ONE address, and two thousand of them in a row cost more than the launch they replace.
  const uint32_t takers = max(1u, min(n, gridDim.x));
  if (p.pub_dst != nullptr && blockIdx.x < takers) {
    __shared__ uint32_t s_last;
    if (threadIdx.x == 0) {
      __threadfence();
      s_last = atomicAdd(p.pub_counter, 1u) == takers - 1u ? 1u : 0u;
    }
    __syncthreads();
    if (s_last) {
      __threadfence();
      const uint32_t n_rec = min(__hip_atomic_load(p.match_count, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), p.pub_max);
      const uint32_t n16 = 4u + 2u * n_rec;   // 64-byte header + 32-byte records, in uint4 units
      for (uint32_t i = threadIdx.x; i < n16; i += 256u) {
        uint4 v = p.pub_src[i];
        if (i == 0) v.z = p.cap;               // header word 2: capacity of the candidate list (see k_publish_records)
        if (i == 1) v.w = 0u;                  // header word 7 follows below, behind everything else
        p.pub_dst[i] = v;
      }
      // the host polls word 7 of the pinned header for this batch's sequence number instead of waiting on the slot's event (a few microseconds of
      // wake-up per one-frame call): every thread's copies are released to system scope first
      __threadfence_system();
      __syncthreads();
      if (threadIdx.x == 0) {
        __hip_atomic_store(reinterpret_cast<uint32_t*>(p.pub_dst) + 7, p.pub_seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        *p.pub_counter = 0u;   // ready for the slot's next batch
      }
    }
  }
}

}  // namespace

// =========================================================================================================
// node-side steps in front of match(), on the device.
// k_pre_color: (MONO8 -> BGR) + GaussianBlur 3x3 (sigma 0 -> [1 2 1]/4 per axis, exact: (sum + 8) >> 4, BORDER_REFLECT_101
// on the FULL frame) + crop, written straight into the level-0 colour buffer.  One thread per output byte.
// k_pre_depth: float metres -> u16 millimetres like convertTo(CV_16UC1, 1000.0): v = z * 1000.f, cvRound (half to even),
// saturate; NaN / Inf / |v| >= 2^31 take x86's "integer indefinite" and saturate to 0 (upstream behaviour on x86-64).
// =========================================================================================================
__global__ __launch_bounds__(256) void k_pre_color(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst, int SH, int SW, int SC, int H,
                                                   int W, int crop_x, int crop_y, int blur3) {
  const int frame = blockIdx.z;
  src += (size_t)frame * SH * SW * SC;
  dst += (size_t)frame * H * W * 3;
  const int j = blockIdx.x * 256 + threadIdx.x;
  const int y = blockIdx.y;
  if (j >= W * 3) return;
  const int x = j / 3, c = j - x * 3;
  const int sc = SC == 1 ? 0 : c;
  const int sy = crop_y + y, sx = crop_x + x;
  int v;
  if (blur3) {
    int acc = 0;
#pragma unroll
    for (int dy = -1; dy <= 1; ++dy) {
      const uint8_t* row = src + (size_t)reflect101(sy + dy, SH) * SW * SC;
      const int r = row[reflect101(sx - 1, SW) * SC + sc] + 2 * row[sx * SC + sc] + row[reflect101(sx + 1, SW) * SC + sc];
      acc += dy == 0 ? 2 * r : r;
    }
    v = (acc + 8) >> 4;
  } else {
    v = src[((size_t)sy * SW + sx) * SC + sc];
  }
  dst[((size_t)y * W + x) * 3 + c] = (uint8_t)v;
}

__global__ __launch_bounds__(256) void k_pre_depth(const void* __restrict__ src, uint16_t* __restrict__ dst, int SH, int SW, int H, int W,
                                                   int crop_x, int crop_y, int is_float) {
  const int frame = blockIdx.z;
  const int x = blockIdx.x * 256 + threadIdx.x, y = blockIdx.y;
  if (x >= W) return;
  const size_t si = (size_t)frame * SH * SW + (size_t)(crop_y + y) * SW + (crop_x + x);
  uint16_t out;
  if (is_float) {
    const float v = reinterpret_cast<const float*>(src)[si] * 1000.f;
    if (!(v > -2147483648.f && v < 2147483648.f)) out = 0;  // NaN, Inf, out of int range: cvtss2si yields INT_MIN -> saturates to 0
    else {
      const float r = rintf(v);
      out = r < 0.f ? 0 : (r > 65535.f ? 65535 : (uint16_t)r);
    }
  } else {
    out = reinterpret_cast<const uint16_t*>(src)[si];
  }
  dst[(size_t)frame * H * W + (size_t)y * W + x] = out;
}

// Debug/test entry: the 16-bin label (0..16, before '& 7') the production code assigns to a gradient (dx, dy) with
// |dx|, |dy| <= 1020, for exhaustive comparison with the CPU oracle's float stage (fastAtan2 + round-half-even).
__global__ void k_debug_orientation_label(const short* __restrict__ dx, const short* __restrict__ dy, uint8_t* __restrict__ out, size_t n) {
  const size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  // the production path runs cq::orientation_label8 (no compares); the hook reports the 16-bin label only where the two rules agree mod 8
  const int q16 = orientation_label16(dx[i], dy[i]);
  out[i] = (uint32_t)(q16 & 7) == cq::orientation_label8(dx[i], dy[i]) ? (uint8_t)q16 : (uint8_t)0xff;
}

// Read-back by kernel instead of by DMA.  hipMemcpyAsync(DeviceToHost) was seen to block the submitting thread for 5-11 ms
// now and then when copies from two streams are in flight (ROCm 7.2, MI355X), which a pipelined caller pays in full; a copy
// kernel writing through the host mapping of the pinned buffer is queued like any other kernel.  `dst` may be pinned host
// memory (device-visible) or device memory.
//   k_publish_records: a slot's [64-byte header][records] block, only as many records as the header says (<= max_records)
//   k_copy_bytes     : plain copy, 16 bytes per thread and step (both pointers 16-byte aligned, bytes % 16 == 0)
//                      header word 2 of the copy = cand_cap, the capacity of the candidate list behind the counts (0 = not stated): a
//                      reader of the block can tell a dropped candidate (word 0 > word 2) from a complete result
__global__ __launch_bounds__(256) void k_publish_records(uint4* __restrict__ dst, const uint4* __restrict__ src, uint32_t max_records, uint32_t cand_cap) {
  const uint32_t n = min(reinterpret_cast<const uint32_t*>(src)[1], max_records);
  const uint32_t n16 = 4u + 2u * n;  // 64-byte header + 32-byte records, in uint4 units
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) {
    uint4 v = src[i];
    if (i == 0) v.z = cand_cap;
    dst[i] = v;
  }
}
__global__ __launch_bounds__(256) void k_copy_bytes(uint4* __restrict__ dst, const uint4* __restrict__ src, size_t n16) {
  for (size_t i = (size_t)blockIdx.x * 256u + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256u) dst[i] = src[i];
}

// the same for `n_blocks` gather blocks (one per rank) `block_bytes` apart in both buffers: blockIdx.y = block
__global__ __launch_bounds__(256) void k_publish_blocks(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, size_t block_bytes, uint32_t max_records) {
  const uint4* s4 = reinterpret_cast<const uint4*>(src + (size_t)blockIdx.y * block_bytes);
  uint4* d4 = reinterpret_cast<uint4*>(dst + (size_t)blockIdx.y * block_bytes);
  const uint32_t n = min(reinterpret_cast<const uint32_t*>(s4)[1], max_records);
  const uint32_t n16 = 4u + 2u * n;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) d4[i] = s4[i];
}
// all-gather by pulling (peer-copy collective of lmx_group.cpp): block y comes from its own source buffer (another member's send
// block: the same device or a peer-accessible one), header + counted records only
__global__ __launch_bounds__(256) void k_pull_blocks(uint8_t* __restrict__ dst, const PullSources srcs, size_t block_bytes, uint32_t max_records) {
  const uint4* s4 = reinterpret_cast<const uint4*>(srcs.src[blockIdx.y]);
  uint4* d4 = reinterpret_cast<uint4*>(dst + (size_t)blockIdx.y * block_bytes);
  const uint32_t n = min(reinterpret_cast<const uint32_t*>(s4)[1], max_records);
  const uint32_t n16 = 4u + 2u * n;
  for (uint32_t i = blockIdx.x * 256u + threadIdx.x; i < n16; i += gridDim.x * 256u) d4[i] = s4[i];
}
void launch_pull_blocks(hipStream_t s, void* dst, const PullSources& srcs, int n_blocks, size_t block_bytes, uint32_t max_records) {
  hipLaunchKernelGGL(k_pull_blocks, dim3(4, n_blocks), dim3(256), 0, s, reinterpret_cast<uint8_t*>(dst), srcs, block_bytes, max_records);
}
void launch_publish_blocks(hipStream_t s, void* dst, const void* src, int n_blocks, size_t block_bytes, uint32_t max_records) {
  hipLaunchKernelGGL(k_publish_blocks, dim3(4, n_blocks), dim3(256), 0, s, reinterpret_cast<uint8_t*>(dst), reinterpret_cast<const uint8_t*>(src), block_bytes,
                     max_records);
}

// Host -> device transfer of caller-owned PINNED images by one kernel per modality: blockIdx.y = frame, the per-frame source
// pointers / row strides sit in a small table in pinned memory as well.  Measured on MI355X (scripts/microbench/h2d_rate.hip): a
// kernel pulling from mapped pinned memory moves 55-57 GB/s, the same as one big hipMemcpyAsync, while 128 per-image
// hipMemcpyAsync calls reach 32 GB/s.  Rows of 16-byte multiples from 16-byte aligned sources go as uint4; anything else bytewise.
__global__ __launch_bounds__(128) void k_pull_frames(const PullEntry* __restrict__ tab, uint8_t* __restrict__ dst, size_t frame_bytes, int rows,
                                                     uint32_t row_bytes) {
  const PullEntry e = tab[blockIdx.y];
  uint8_t* d = dst + (size_t)blockIdx.y * frame_bytes;
  const uint8_t* src = reinterpret_cast<const uint8_t*>(e.src);
  if (((e.src | e.row_stride | row_bytes) & 15u) == 0) {
    const uint32_t n16 = row_bytes >> 4;
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const uint4* sp = reinterpret_cast<const uint4*>(src + (size_t)r * e.row_stride);
      uint4* dp = reinterpret_cast<uint4*>(d + (size_t)r * row_bytes);
      for (uint32_t c = threadIdx.x; c < n16; c += 128) dp[c] = sp[c];
    }
  } else {
    for (int r = blockIdx.x; r < rows; r += gridDim.x) {
      const uint8_t* sp = src + (size_t)r * e.row_stride;
      uint8_t* dp = d + (size_t)r * row_bytes;
      for (uint32_t c = threadIdx.x; c < row_bytes; c += 128) dp[c] = sp[c];
    }
  }
}
void launch_pull_frames(hipStream_t s, const PullEntry* tab, uint8_t* dst, size_t frame_bytes, int rows, uint32_t row_bytes, int n_frames) {
  const int bx = std::max(1, std::min(rows, 2048 / std::max(1, n_frames)));
  hipLaunchKernelGGL(k_pull_frames, dim3(bx, n_frames), dim3(128), 0, s, tab, dst, frame_bytes, rows, row_bytes);
}

void launch_publish_records(hipStream_t s, void* dst, const void* src, uint32_t max_records, uint32_t cand_cap) {
  hipLaunchKernelGGL(k_publish_records, dim3(16), dim3(256), 0, s, reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(src), max_records, cand_cap);
}
void launch_copy_bytes(hipStream_t s, void* dst, const void* src, size_t bytes) {
  const size_t n16 = bytes / 16;
  const unsigned blocks = (unsigned)std::min<size_t>(256, (n16 + 255) / 256);
  if (n16) hipLaunchKernelGGL(k_copy_bytes, dim3(blocks), dim3(256), 0, s, reinterpret_cast<uint4*>(dst), reinterpret_cast<const uint4*>(src), n16);
}

void launch_pre_color(hipStream_t s, const uint8_t* src, uint8_t* dst, int SH, int SW, int SC, int H, int W, int crop_x, int crop_y, int blur3,
                      int n_frames) {
  hipLaunchKernelGGL(k_pre_color, dim3((W * 3 + 255) / 256, H, n_frames), dim3(256), 0, s, src, dst, SH, SW, SC, H, W, crop_x, crop_y, blur3);
}
void launch_pre_depth(hipStream_t s, const void* src, uint16_t* dst, int SH, int SW, int H, int W, int crop_x, int crop_y, int is_float,
                      int n_frames) {
  hipLaunchKernelGGL(k_pre_depth, dim3((W + 255) / 256, H, n_frames), dim3(256), 0, s, src, dst, SH, SW, H, W, crop_x, crop_y, is_float);
}

void launch_debug_orientation_label(hipStream_t s, const short* dx, const short* dy, uint8_t* out, size_t n) {
  hipLaunchKernelGGL(k_debug_orientation_label, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, dx, dy, out, n);
}

// ---- launchers --------------------------------------------------------------------------------------------
// Extra (unused) dynamic LDS per workgroup of the issue-bound per-pixel kernels: caps how many of their workgroups a CU holds, which
// leaves wave slots for the memory-bound scoring kernel of ANOTHER device lane to run beside them (device lanes).
// LMX_LDS_PAD_COLOR / _DEPTH / _SPREAD (bytes) override the defaults for experiments.
static size_t lds_pad(const char* env, size_t dflt) {
  const char* e = std::getenv(env);
  return e ? (size_t)std::strtoul(e, nullptr, 10) : dflt;
}

void launch_color_quantize(hipStream_t s, const uint8_t* bgr, uint8_t* quant, uint8_t* pyr_next, int H, int W, int n_frames, float weak_threshold,
                           float* mag_out, uint32_t* clear16, const StreamWait* wait) {
  const StreamWait sw = wait ? *wait : StreamWait();
  const bool xcd = n_frames >= 8;   // XCD-aware tile placement, see tile_of_block
  // batches take the tall tile (less halo per output); one or two frames per call keep 16 rows: twice the workgroups for a launch that
  // does not fill the GPU anyway.  LMX_COLOR_TILE=16|32 pins it (A/B switch, read once).
  static const int forced = []() { const char* e = std::getenv("LMX_COLOR_TILE"); return e ? std::atoi(e) : 0; }();
  const bool tall = forced ? forced == CQ_TH_TALL : (xcd && H >= 2 * CQ_TH_TALL);
  const int th = tall ? CQ_TH_TALL : CQ_TH;
  const int tx = (W + CQ_TW - 1) / CQ_TW, ty = (H + th - 1) / th;
  dim3 grid = xcd ? dim3((unsigned)(tx * ty * 8 * ((n_frames + 7) / 8))) : dim3(tx, ty, n_frames);
  static const size_t pad = lds_pad("LMX_LDS_PAD_COLOR", 0);
  const float thr_sq = weak_threshold * weak_threshold;
  const int nfx = xcd ? n_frames : 0;
  // the trainer's instantiation also writes the squared magnitudes (extractTemplate ranks candidates by them)
  if (tall && mag_out) hipLaunchKernelGGL((k_color_quantize<CQ_TH_TALL, true>), grid, dim3(256), pad, s, bgr, quant, pyr_next, mag_out, H, W, thr_sq, clear16, nfx, sw);
  else if (tall) hipLaunchKernelGGL((k_color_quantize<CQ_TH_TALL, false>), grid, dim3(256), pad, s, bgr, quant, pyr_next, mag_out, H, W, thr_sq, clear16, nfx, sw);
  else if (mag_out) hipLaunchKernelGGL((k_color_quantize<CQ_TH, true>), grid, dim3(256), pad, s, bgr, quant, pyr_next, mag_out, H, W, thr_sq, clear16, nfx, sw);
  else hipLaunchKernelGGL((k_color_quantize<CQ_TH, false>), grid, dim3(256), pad, s, bgr, quant, pyr_next, mag_out, H, W, thr_sq, clear16, nfx, sw);
}

// quant_half (or null): also writes the next pyramid level's label image, upstream's nearest-neighbour pyrDown dst(y, x) = src(2y, 2x)
void launch_depth_quantize(hipStream_t s, const uint16_t* depth, uint8_t* quant, uint8_t* quant_half, int H, int W, int n_frames,
                           int distance_threshold, int difference_threshold, const uint8_t* lut_bins, uint32_t* clear16) {
  const int tx = (W + 63) / 64, ty = (H + DQ_TH - 1) / DQ_TH;
  const bool xcd = n_frames >= 8;   // XCD-aware tile placement, see tile_of_block
  dim3 grid = xcd ? dim3((unsigned)(tx * ty * 8 * ((n_frames + 7) / 8))) : dim3(tx, ty, n_frames);
  const int nfx = xcd ? n_frames : 0;
  static const size_t pad = lds_pad("LMX_LDS_PAD_DEPTH", 0);
  if (difference_threshold <= 200)
    hipLaunchKernelGGL(k_depth_quantize<int>, grid, dim3(256), pad, s, depth, quant, quant_half, H, W, distance_threshold, difference_threshold, lut_bins, clear16, nfx);
  else
    hipLaunchKernelGGL(k_depth_quantize<long long>, grid, dim3(256), pad, s, depth, quant, quant_half, H, W, distance_threshold, difference_threshold,
                       lut_bins, clear16, nfx);
}

// Detector::match(..., masks) (SURVEY A.9 / upstream QuantizedPyramid::quantize: `quantized.copyTo(dst, mask)`): the labels of a level survive
// where the level's mask is non-zero.  Upstream halves the mask per pyramid level with resize(INTER_NEAREST), i.e. mask_l(y, x) =
// mask_0(y << l, x << l); the level-0 mask is all this kernel needs.  One thread per 4 label bytes.
__global__ __launch_bounds__(256) void k_apply_mask(uint8_t* __restrict__ quant, const uint8_t* __restrict__ mask0, int Hl, int Wl, int W0, int H0, int level) {
  const int frame = blockIdx.z, y = blockIdx.y, x4 = (blockIdx.x * 256 + threadIdx.x) * 4;
  if (x4 >= Wl) return;
  uint8_t* q = quant + (size_t)frame * Hl * Wl + (size_t)y * Wl + x4;
  const uint8_t* m = mask0 + (size_t)frame * H0 * W0 + (size_t)(y << level) * W0;
  const int n = min(4, Wl - x4);
  for (int k = 0; k < n; ++k)
    if (m[(x4 + k) << level] == 0) q[k] = 0;
}
void launch_apply_mask(hipStream_t s, uint8_t* quant, const uint8_t* mask0, int Hl, int Wl, int W0, int H0, int level, int n_frames) {
  hipLaunchKernelGGL(k_apply_mask, dim3((unsigned)((Wl + 1023) / 1024), (unsigned)Hl, (unsigned)n_frames), dim3(256), 0, s, quant, mask0, Hl, Wl, W0, H0, level);
}

void launch_nn_down2(hipStream_t s, const uint8_t* src, uint8_t* dst, int Hd, int Wd, int n_frames) {
  dim3 grid((Wd + 63) / 64, (Hd + 3) / 4, n_frames);
  hipLaunchKernelGGL(k_nn_down2, grid, dim3(256), 0, s, src, dst, Hd, Wd);
}

template <int T>
static void launch_spread_linearize_t(hipStream_t s, const SpreadBatch& b, int n_mod, const LevelGeom& g, int n_frames) {
  constexpr int ND = (T + 2) / 4 + 2;
  const int Wd = g.W / 4 + ND;
  static const size_t pad = lds_pad("LMX_LDS_PAD_SPREAD", 0);
  size_t smem = 2048 + (size_t)(2 * T - 1 + T) * Wd * 4 + (size_t)T * g.W + pad;
  if (n_frames >= 8)
    hipLaunchKernelGGL(k_spread_linearize_t<T>, dim3((unsigned)(g.Hc * 8 * ((n_frames + 7) / 8)), n_mod, 1), dim3(256), smem, s, b, g, n_frames);
  else
    hipLaunchKernelGGL(k_spread_linearize_t<T>, dim3(g.Hc, n_mod, n_frames), dim3(256), smem, s, b, g, 0);
}

bool spread_writes_nibbles(const LevelGeom& g) {  // the fused coarsest-level path of k_spread_linearize_t
  return (g.W & 3) == 0 && (g.Wc & 7) == 0 && (g.T == 4 || g.T == 5 || g.T == 8);
}

// Coarsest level: `lmn` is the nibble-packed destination; when spread_writes_nibbles(g) it is written directly and `lm` is not
// touched, otherwise `lm` gets the byte-wide memories and the caller runs k_pack_nibbles.  Finer levels: only `ls`.
static bool spread_fast_path(const LevelGeom& g) { return (g.W & 3) == 0 && (g.Wc & 3) == 0 && (g.T == 4 || g.T == 5 || g.T == 8); }

bool launch_spread_linearize_all(hipStream_t s, const SpreadBatch& b_in, int n_mod, const LevelGeom& g, int n_frames) {
  if (!spread_fast_path(g) || n_mod < 1) return false;
  SpreadBatch b = b_in;
  for (int m = 0; m < n_mod; ++m)
    if (b.ls[m] == nullptr && b.lmn[m] != nullptr && !spread_writes_nibbles(g)) b.lmn[m] = nullptr;
  if (g.T == 4) launch_spread_linearize_t<4>(s, b, n_mod, g, n_frames);
  else if (g.T == 5) launch_spread_linearize_t<5>(s, b, n_mod, g, n_frames);
  else launch_spread_linearize_t<8>(s, b, n_mod, g, n_frames);
  return true;
}

void launch_spread_linearize(hipStream_t s, const uint8_t* quant, uint8_t* lm, uint8_t* ls, uint8_t* lmn, const LevelGeom& g, int n_frames) {
  SpreadBatch b{};
  b.quant[0] = quant; b.lm[0] = lm; b.ls[0] = ls; b.lmn[0] = lmn;
  if (launch_spread_linearize_all(s, b, 1, g, n_frames)) return;
  const int rows_in = 2 * g.T - 1;
  const int Wp = (g.W + g.T - 1 + 3) & ~3;
  size_t smem = 2048 + (size_t)rows_in * Wp + (size_t)rows_in * g.W + (size_t)g.T * g.W;
  dim3 grid(g.Hc, 1, n_frames);
  hipLaunchKernelGGL(k_spread_linearize, grid, dim3(256), smem, s, quant, lm, ls, g);
}

void launch_pack_nibbles(hipStream_t s, const uint8_t* lm, uint8_t* lmn, const LevelGeom& g, int n_frames) {
  const uint32_t n_elem = (uint32_t)g.T * g.T * g.cells;
  const uint32_t groups = (n_elem + 8 + 7) / 8;
  dim3 grid((groups + 255) / 256, 8, n_frames);
  hipLaunchKernelGGL(k_pack_nibbles, grid, dim3(256), 0, s, lm, lmn, g);
}

// 0 = generic k_score_coarse, 1 = k_score_coarse_u8, 2 = k_score_coarse_sb (default when the bank qualifies).  The context chooses once,
// when it is created (DeviceBankView::score_variant; environment LMX_SCORE_KERNEL = generic | u8 | sb, LMX_SCORE_GENERIC=1 = "generic").
int score_kernel_variant(const DeviceBankView& bank) { return bank.uni_ok ? bank.score_variant : 0; }

void launch_score_coarse(hipStream_t s, const DeviceBankView& bank, const LevelGeom& g, const uint8_t* const* lm_mod, int n_frames,
                         float threshold, const int32_t* class_slot, Candidate* cands, uint32_t* header, uint32_t cap, int n_stripes) {
  ScoreParams p;
  p.info = bank.info; p.linfo = bank.linfo; p.coarse_off = bank.coarse_off; p.class_slot = class_slot;
  p.feat_count_coarse = bank.feat_count + (size_t)(bank.L - 1) * bank.G * bank.M;
  for (int m = 0; m < kMaxModalities; ++m) p.lm[m] = m < bank.M ? lm_mod[m] : nullptr;
  p.mod_stride = g.nib_mod_stride;
  p.G = bank.G; p.L = bank.L; p.M = bank.M; p.nf_max = bank.nf_max_coarse;
  p.threshold = threshold;
  p.cands = cands; p.stripes = stripes_of_header(header); p.cap = cap; p.n_stripes = (uint32_t)n_stripes;
  if (bank.G <= 0) return;
  p.n_frames = n_frames;
  p.blocks_per_frame = (bank.G + SC_WAVES_PER_BLOCK - 1) / SC_WAVES_PER_BLOCK;
  p.xcd_frames = n_frames >= 8 ? 1 : 0;
  const int frame_slots = p.xcd_frames ? 8 * ((n_frames + 7) / 8) : n_frames;
  p.uni_off = bank.uni_ok ? bank.coarse_uni : nullptr;
  p.blk_off = bank.uni_ok ? bank.coarse_blk : nullptr;
  p.sinfo = bank.sinfo;
  const int variant = score_kernel_variant(bank);
  const dim3 grid((unsigned)(p.blocks_per_frame * frame_slots)), block(64 * SC_WAVES_PER_BLOCK);
  if (bank.score_no_prune) {   // LMX_SCORE_NO_PRUNE: similarity()'s full work, same candidates (see score_pass)
    if (variant == 2) hipLaunchKernelGGL(k_score_coarse_sb<false>, grid, block, 0, s, p);
    else if (variant == 1) hipLaunchKernelGGL(k_score_coarse_u8<false>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_score_coarse<false>, grid, block, 0, s, p);
  } else {
    if (variant == 2) hipLaunchKernelGGL(k_score_coarse_sb<true>, grid, block, 0, s, p);
    else if (variant == 1) hipLaunchKernelGGL(k_score_coarse_u8<true>, grid, block, 0, s, p);
    else hipLaunchKernelGGL(k_score_coarse<true>, grid, block, 0, s, p);
  }
}

bool launch_refine(hipStream_t s, const DeviceBankView& bank, const KernelParams& kp, int n_frames, float threshold,
                   const int32_t* class_slot, const Candidate* cands, uint32_t* header, uint32_t cap, int n_stripes,
                   lmx_raw_match_t* matches, uint32_t* match_count, void* pub_dst, const void* pub_src, uint32_t* pub_counter, uint32_t pub_max, uint32_t pub_seq) {
  RefineParams p;
  p.info = bank.info; p.linfo = bank.linfo; p.feat = bank.feat; p.feat_count = bank.feat_count; p.class_slot = class_slot;
  for (int l = 0; l < kMaxLevels; ++l) {
    p.geom[l] = kp.geom[l];
    for (int m = 0; m < kMaxModalities; ++m) p.ls[l][m] = kp.fb.ls[l][m];
  }
  p.G = bank.G; p.L = bank.L; p.M = bank.M; p.threshold = threshold;
  p.cands = cands; p.stripes = stripes_of_header(header); p.header = header; p.cap = cap; p.n_stripes = (uint32_t)n_stripes; p.matches = matches; p.match_count = match_count;
  p.pub_dst = reinterpret_cast<uint4*>(pub_dst); p.pub_src = reinterpret_cast<const uint4*>(pub_src); p.pub_counter = pub_counter; p.pub_max = pub_max; p.pub_seq = pub_seq;
  if (bank.G <= 0) return false;   // nothing launched: the caller publishes with k_publish_records
  (void)n_frames;  // candidates of all frames share one list
  // small batches: few candidates, and every workgroup costs the last one a ticket
  hipLaunchKernelGGL(k_refine, dim3(n_frames <= 2 ? 512 : 2048), dim3(256), 0, s, p);
  return true;
}

// ---- fused launches of the small-batch chain (one or two frames per call) -------------------------------------------------------
bool launch_small_depth_color(hipStream_t s, const uint16_t* depth, uint8_t* dq, uint8_t* dq_half, int H, int W, int distance_threshold, int difference_threshold,
                              const uint8_t* lut_bins, const uint8_t* bgr1, uint8_t* cq1, uint8_t* pyr2, int H1, int W1, float weak_threshold, int n_frames,
                              const StreamWait* wait) {
  SmallQuantArgs a;
  a.wait = wait ? *wait : StreamWait();
  a.depth = depth; a.dq = dq; a.dq_half = dq_half; a.H = H; a.W = W; a.distance_threshold = distance_threshold; a.difference_threshold = difference_threshold;
  a.lut_bins = lut_bins; a.bgr1 = bgr1; a.cq1 = cq1; a.pyr2 = pyr2; a.H1 = H1; a.W1 = W1; a.thr_sq = weak_threshold * weak_threshold;
  a.dtx = (W + 63) / 64; a.dty = (H + DQ_TH - 1) / DQ_TH; a.ctx = (W1 + CQ_TW - 1) / CQ_TW; a.cty = (H1 + CQ_TH - 1) / CQ_TH;
  a.n_depth = a.dtx * a.dty * n_frames;
  const unsigned grid = (unsigned)(a.n_depth + a.ctx * a.cty * n_frames);
  if (difference_threshold <= 200) hipLaunchKernelGGL(k_small_depth_color<int>, dim3(grid), dim3(256), 0, s, a);
  else hipLaunchKernelGGL(k_small_depth_color<long long>, dim3(grid), dim3(256), 0, s, a);
  return true;
}

template <int T0, int T1>
static void launch_small_spread_t(hipStream_t s, const SpreadBatch& b0, const LevelGeom& g0, const SpreadBatch& b1, const LevelGeom& g1, int n_mod, int n_frames) {
  auto need = [](int T, const LevelGeom& g) { return (size_t)2048 + (size_t)(2 * T - 1 + T) * (g.W / 4 + (T + 2) / 4 + 2) * 4 + (size_t)T * g.W; };
  const size_t smem = std::max(need(T0, g0), need(T1, g1));
  const int n0 = g0.Hc * n_mod * n_frames, n1 = g1.Hc * n_mod * n_frames;
  hipLaunchKernelGGL((k_small_spread<T0, T1>), dim3((unsigned)(n0 + n1)), dim3(256), smem, s, b0, g0, b1, g1, n0, n_mod);
}

// Levels 0 and 1 of a two-level bank in one launch; false when the pair of T has no fused kernel (the caller launches per level).
bool launch_small_spread(hipStream_t s, const SpreadBatch& b0_in, const LevelGeom& g0, const SpreadBatch& b1_in, const LevelGeom& g1, int n_mod, int n_frames) {
  if (!spread_fast_path(g0) || !spread_fast_path(g1) || !spread_writes_nibbles(g1) || n_mod < 1) return false;
  SpreadBatch b0 = b0_in, b1 = b1_in;
  for (int m = 0; m < n_mod; ++m) { b0.lm[m] = nullptr; b0.lmn[m] = nullptr; b1.ls[m] = nullptr; }
  if (g0.T == 5 && g1.T == 8) launch_small_spread_t<5, 8>(s, b0, g0, b1, g1, n_mod, n_frames);
  else if (g0.T == 4 && g1.T == 8) launch_small_spread_t<4, 8>(s, b0, g0, b1, g1, n_mod, n_frames);
  else return false;
  return true;
}

}  // namespace lmx
