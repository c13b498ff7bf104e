// k_color_quantize's body: SURVEY a4 + a5 (+ a6) -- quantizedOrientations + hysteresisGradient fused over an LDS tile, and cv::pyrDown of the
// colour source for the next pyramid level from the same tile (upstream cv::linemod ColorGradientPyramid, call site
// /root/reference/src/rgbdDetector.cpp:33).  All integer; the orientation label is an exact integer rule (orientation_label).
//
// Round 4 rewrite ("instruction diet").  The round-3 kernel issued 1811 VALU instructions per wave = 226 per output pixel
// (static count per stage x trip counts reproduces the PMC figure: scripts/isa_mix.py); this one is built from what the issue-rate
// microbenchmark says instructions cost on gfx950 (profiles/r04_valu_issue_microbench.txt):
//   * both separable filters run HORIZONTAL FIRST ON BYTES with v_dot4_u32_u8 -- a window that starts at byte offset o of a dword needs no
//     funnel shift, only the weight vector shifted by o -- and then VERTICAL on u16 ROW PAIRS with v_dot2_u32_u16 (rows 2p, 2p+1 of a column
//     share a dword): 2.5 + 4 multiply-class instructions per pixel and channel for the 7 x 7 blur (was ~13 instructions: byte splits, packed
//     16-bit multiplies and adds, then eight dot2 per output pair), 2 + 3 per output for pyrDown's 5 x 5 (whose two passes cost 322 instructions
//     per wave before, 82 per output pixel in the horizontal one: index arithmetic, not filtering);
//   * v_cndmask_b32 is cheap only directly behind the v_cmp that made its mask (6.3 cycles the pair); a second select on the same condition
//     costs 11-19 cycles (22.8 with a stale VCC), v_max / v_min / v_ffbl / shifts by a VGPR amount issue at half rate.  Selections are
//     therefore arithmetic: mask = (a - b) >> 31 in a VGPR, v_bitop3_b32 (full rate) picks; |x| = (x ^ s) - s; conditional negation likewise;
//   * the 3 x 3 vote reads ready-made one-hot nibble counters (1 << 4 * label, a dword per pixel) instead of rebuilding them from label bytes.
// Stages (a barrier between each): A load | Ph, Pv pyrDown | Bh, Bv blur | D Sobel + label | E vote.
//
// The same source compiles for the CPU with LMX_CQ_HOST defined (tests/cpp/cq_host.cpp): the 256 threads of a workgroup are emulated stage by
// stage -- stages only communicate through LDS across barriers -- and the result is compared with the oracle on this container's CPU
// (tests/test_color_kernel_host.py), borders, ragged tiles and both tile heights included.
#pragma once

#include <stdint.h>
#include <stddef.h>

#include <type_traits>

#if defined(LMX_CQ_HOST)
#include <cmath>
#include <cstring>
#define CQ_FN static inline
#else
#define CQ_FN __device__ __forceinline__
#endif

#ifndef LMX_CQ_SKIP
#define LMX_CQ_SKIP 0   // timing experiments only (scripts/build_variants.py color): bit k compiles one stage's work out -- A 1, P 2, Bh 4, Bv 8, D 16, E 32; results are wrong
#endif

namespace lmx {
namespace cq {

// ---- the few machine instructions the body is written against ----------------------------------------------------------------------
#if defined(LMX_CQ_HOST)
CQ_FN uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) {   // v_perm_b32: selector byte 0..3 = lo's bytes, 4..7 = hi's bytes, 0x0c = 0x00
  const uint64_t pool = ((uint64_t)hi << 32) | lo;
  uint32_t out = 0;
  for (int i = 0; i < 4; ++i) {
    const uint32_t s = (sel >> (8 * i)) & 0xffu;
    const uint32_t b = s < 8 ? (uint32_t)((pool >> (8 * s)) & 0xffu) : (s == 0x0c ? 0u : 0xffu);
    out |= b << (8 * i);
  }
  return out;
}
CQ_FN uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) {
  for (int i = 0; i < 4; ++i) c += ((a >> (8 * i)) & 0xffu) * ((b >> (8 * i)) & 0xffu);
  return c;
}
CQ_FN uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) { return c + (a & 0xffffu) * (b & 0xffffu) + (a >> 16) * (b >> 16); }
CQ_FN int mul24(int a, int b) { return a * b; }
CQ_FN uint32_t umul24(uint32_t a, uint32_t b) { return a * b; }
CQ_FN uint32_t select_mask(uint32_t if_set, uint32_t if_clear, uint32_t mask) { return (if_set & mask) | (if_clear & ~mask); }
CQ_FN int uniform(int v) { return v; }
CQ_FN int sign_mask(int v) { return v < 0 ? -1 : 0; }
CQ_FN int sign_bit(int v) { return v < 0 ? 1 : 0; }
CQ_FN int ffs32(uint32_t v) { return v ? __builtin_ctz(v) + 1 : 0; }
CQ_FN uint32_t load_u32(const uint8_t* p) { uint32_t v; std::memcpy(&v, p, 4); return v; }
CQ_FN void store_u32(uint8_t* p, uint32_t v) { std::memcpy(p, &v, 4); }
CQ_FN void store_u16(uint8_t* p, uint32_t v) { const uint16_t h = (uint16_t)v; std::memcpy(p, &h, 2); }
struct u32x4 { uint32_t x, y, z, w; };
struct u32x2 { uint32_t x, y; };
#else
CQ_FN uint32_t perm(uint32_t hi, uint32_t lo, uint32_t sel) { return __builtin_amdgcn_perm(hi, lo, sel); }
CQ_FN uint32_t udot4(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot4(a, b, c, false); }
typedef unsigned short cq_us2 __attribute__((ext_vector_type(2)));
CQ_FN uint32_t udot2(uint32_t a, uint32_t b, uint32_t c) { return __builtin_amdgcn_udot2(__builtin_bit_cast(cq_us2, a), __builtin_bit_cast(cq_us2, b), c, false); }
CQ_FN int mul24(int a, int b) { return __mul24(a, b); }
CQ_FN uint32_t umul24(uint32_t a, uint32_t b) { return __umul24(a, b); }
// (if_set & mask) | (if_clear & ~mask) in one full-rate instruction; truth table index = a << 2 | b << 1 | c with a = if_set, b = if_clear, c = mask
CQ_FN uint32_t select_mask(uint32_t if_set, uint32_t if_clear, uint32_t mask) { return __builtin_amdgcn_bitop3_b32(if_set, if_clear, mask, 0xe4); }
CQ_FN int uniform(int v) { return __builtin_amdgcn_readfirstlane(v); }
// x >> 31 (arithmetic: 0 / -1) and (unsigned)x >> 31 (0 / 1), opaque to the optimiser on purpose: written as plain shifts LLVM turns
// "(a - b) >> 31" back into v_cmp + v_cndmask and "(x ^ s) - s" into v_sub + v_max_i32 -- the half-rate / VCC-dependent forms this
// kernel avoids (see the header of this file)
CQ_FN int sign_mask(int v) { int m; asm("v_ashrrev_i32 %0, 31, %1" : "=v"(m) : "v"(v)); return m; }
CQ_FN int sign_bit(int v) { int m; asm("v_lshrrev_b32 %0, 31, %1" : "=v"(m) : "v"(v)); return m; }
CQ_FN int ffs32(uint32_t v) { return __ffs((int)v); }
CQ_FN uint32_t load_u32(const uint8_t* p) { uint32_t v; __builtin_memcpy(&v, p, 4); return v; }   // gfx950 runs in unaligned-access mode: one load
CQ_FN void store_u32(uint8_t* p, uint32_t v) { __builtin_memcpy(p, &v, 4); }
CQ_FN void store_u16(uint8_t* p, uint32_t v) { const uint16_t h = (uint16_t)v; __builtin_memcpy(p, &h, 2); }
typedef uint4 u32x4;
typedef uint2 u32x2;
#endif

CQ_FN int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
constexpr uint32_t b4(uint32_t b0, uint32_t b1, uint32_t b2, uint32_t b3) { return b0 | (b1 << 8) | (b2 << 16) | (b3 << 24); }   // weight bytes of a dot4
constexpr uint32_t h2(uint32_t lo, uint32_t hi) { return lo | (hi << 16); }                                                        // weight halves of a dot2

// The label (0..7) of a Sobel gradient.  Upstream computes saturate_cast<uchar>(cvRound(fastAtan2(dy, dx) * (16/360))) & 7 in float (phase +
// convertTo in hysteresisGradient).  Sobel outputs of 8-bit images are integers in [-1020, 1020], and over that whole domain the float
// pipeline (polynomial, 90-/180-/360- folds, round-half-even) is a pure function of the octant and of two thresholds on
// min(|dx|,|dy|) / max(|dx|,|dy|): the 16-bin label changes between 182/915 and 73/367 and between 661/989 and 264/395 in every octant, so any
// rational inside those gaps reproduces it EXACTLY: 255/1282 and 925/1384 (their mediants).  Checked for all 2041^2 gradients against the
// oracle's float restatement on the CPU (tests/test_oracle_kat.py, tests/test_color_kernel_host.py) and on the device
// (tests/test_gpu_parity.py::test_orientation_quantiser_exhaustive).  Products stay below 2^23 -> 24-bit multiplies.
//   s  = [mn * 1282 > mx * 255] + [mn * 1384 > mx * 925]          (0..2: how far the vector is from its nearer axis)
//   q0 = |dx| >= |dy| ? s : 4 - s                                   (first quadrant, 0..4)
//   16-bin label = dx < 0 ? 8 - q0 : q0, then dy < 0 ? 16 - that : that;  mod 8 both folds are a negation: label & 7 = (signs differ ? -q0 : q0) & 7
// No compare / select instruction is left: sign masks by arithmetic shifts, indicator bits from the sign of the cross products.
CQ_FN int orientation_label16(int dx, int dy) {   // the 16-bin form (0..16), kept for the exhaustive test hook
  const int ax = dx < 0 ? -dx : dx, ay = dy < 0 ? -dy : dy;
  const int mn = ax < ay ? ax : ay, mx = ax < ay ? ay : ax;
  const int s = (int)(mul24(mn, 1282) > mul24(mx, 255)) + (int)(mul24(mn, 1384) > mul24(mx, 925));
  int q = ax >= ay ? s : 4 - s;
  q = dx < 0 ? 8 - q : q;
  return dy < 0 ? 16 - q : q;
}
CQ_FN uint32_t orientation_label8(int dx, int dy) {
  const int sx = sign_mask(dx), sy = sign_mask(dy);    // 0 / -1
  const int ax = (dx ^ sx) - sx, ay = (dy ^ sy) - sy;
  const int d = ax - ay, lt = sign_mask(d);            // lt = -1 iff |dx| < |dy|
  const int mn = ay + (d & lt), mx = ax - (d & lt);
  const int t1 = mul24(mx, 255) - mul24(mn, 1282), t2 = mul24(mx, 925) - mul24(mn, 1384);   // negative iff the indicator holds
  const int s = sign_bit(t1) + sign_bit(t2);
  const int q0 = (s ^ lt) + (lt & 5);                  // lt ? 4 - s : s
  const int sg = sx ^ sy;                              // -1 iff exactly one component is negative
  return (uint32_t)((q0 ^ sg) - sg) & 7u;
}

constexpr int TW = 64;   // tile width in output pixels
template <int TH>
struct Geo {   // sizes of one 64 x TH tile's working set; LDS layout
  static constexpr int IW = TW + 10, IH = TH + 10, IS = 76;   // clamped source tile, halo 5: 74 x IH bytes per channel plane, row stride 76 (19 dwords)
  static constexpr int SH = TH + 4, SW = TW + 4;              // smoothed region, halo 2 (row stride 68 = 17 dwords)
  static constexpr int QH = TH + 2, QS = 68;                  // label region, halo 1: 66 columns used, row stride 68
  static constexpr int HP = IH / 2;                           // row pairs of the blur's horizontal pass (rows 2p, 2p + 1)
  static constexpr int PP = TH / 2 + 2;                       // row pairs of pyrDown's horizontal pass: pair q = rows (2q + 1, 2q + 2), q = 1 .. TH/2 + 2, stored at q - 1
  static constexpr size_t SZ_IN = (size_t)3 * IH * IS, SZ_SM = (size_t)3 * SH * SW, SZ_H = (size_t)3 * HP * SW * 4, SZ_PH = (size_t)3 * PP * 32 * 4;
  static constexpr size_t SZ_OH = (size_t)QH * QS * 4, SZ_FL = (size_t)QH * QS;
  // region 1: s_in (A -> Ph, Bh), then s_sm (Bv -> D).   region 2: s_ph (Ph -> Pv), then s_h (Bh -> Bv), then s_oh + s_fl (D -> E).
  static constexpr size_t OFF_R2 = (SZ_IN + 15) & ~(size_t)15;
  static constexpr size_t OFF_FL = OFF_R2 + SZ_OH;
  static constexpr size_t LDS_BYTES = OFF_R2 + SZ_H;          // 26 720 bytes at TH = 32 (six workgroups per CU), 16 536 at TH = 16
  static_assert(TH % 4 == 0 && IH % 2 == 0, "row pairs");
  static_assert(SZ_SM <= SZ_IN && SZ_PH <= SZ_H && SZ_OH + SZ_FL <= SZ_H, "tenants fit their regions");
};

// One tile.  `tid` runs over the 256 threads of the workgroup inside every stage; `run(stage)` calls stage(tid) for the calling thread and
// ends with a barrier on the device, and loops tid over 0..255 on the host.
template <int TH, bool TRAIN, typename Run>
CQ_FN void color_quantize_tile(int tile_x, int tile_y, const uint8_t* __restrict__ src /* frame */, uint8_t* __restrict__ dst /* frame */,
                               uint8_t* __restrict__ pyr_dst /* frame or null */, float* __restrict__ mag_dst /* frame, TRAIN only */, int H, int W, float thr_sq,
                               uint8_t* __restrict__ s_raw, Run run) {
  typedef Geo<TH> G;
  constexpr int IH = G::IH, IS = G::IS, SH = G::SH, SW = G::SW, QH = G::QH, QS = G::QS, HP = G::HP, PP = G::PP;
  uint32_t* const s_in32 = reinterpret_cast<uint32_t*>(s_raw);                    // [3 * IH][19]
  uint8_t* const s_in = s_raw;                                                    // [3 * IH][IS]
  uint32_t* const s_r2 = reinterpret_cast<uint32_t*>(s_raw + G::OFF_R2);         // s_ph [3 * PP][32] | s_h [3 * HP][SW] | s_oh [QH][QS]
  uint8_t* const s_sm = s_raw;                                                    // [3 * SH][SW]
  uint8_t* const s_fl = s_raw + G::OFF_FL;                                        // [QH][QS]
  const int x0 = tile_x * TW, y0 = tile_y * TH;

  // ---- A: the clamped source tile (BORDER_REPLICATE of the blur), de-interleaved into three byte planes -------------------------------
  run([&](int tid) {
    if (LMX_CQ_SKIP & 1) return;
    if (x0 >= 5 && x0 + IS - 5 <= W) {
      // interior columns: 4 pixels = 3 dwords per task (the row segment starts at byte 3 * (x0 - 5): not dword aligned), de-interleaved with
      // v_perm_b32 into one dword per plane
      // a thread's tasks (IH * 19 / 256: three, for 30 threads of the tall tile four) are unrolled so that all their loads are in flight before
      // the first permute: the stage is the latency of one round trip to L2 / HBM, not of three or four in a row
      constexpr int NT = (IH * (IS / 4) + 255) / 256;
      uint32_t d[NT][3];
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int i = tid + 256 * k;
        if (i < IH * (IS / 4)) {
          const int ly = i / (IS / 4), t = i - ly * (IS / 4);
          const int gy = clampi(y0 - 5 + ly, 0, H - 1);
          const uint8_t* p = src + (umul24((uint32_t)gy, (uint32_t)W) + (uint32_t)((x0 - 5) + 4 * t)) * 3u;   // one frame is < 4 GiB
          d[k][0] = load_u32(p); d[k][1] = load_u32(p + 4); d[k][2] = load_u32(p + 8);
        }
      }
#pragma unroll
      for (int k = 0; k < NT; ++k) {
        const int i = tid + 256 * k;
        if (i < IH * (IS / 4)) {
          const int ly = i / (IS / 4), t = i - ly * (IS / 4);
          const uint32_t d0 = d[k][0], d1 = d[k][1], d2 = d[k][2];
          // d0 = b0 g0 r0 b1 | d1 = g1 r1 b2 g2 | d2 = r2 b3 g3 r3   (byte 0 first)
          s_in32[(0 * IH + ly) * (IS / 4) + t] = perm(d2, perm(d1, d0, 0x00060300u), 0x05020100u);
          s_in32[(1 * IH + ly) * (IS / 4) + t] = perm(d2, perm(d1, d0, 0x00070401u), 0x06020100u);
          s_in32[(2 * IH + ly) * (IS / 4) + t] = perm(d2, perm(d1, d0, 0x00000502u), 0x07040100u);
        }
      }
    } else {
      for (int i = tid; i < IH * IS; i += 256) {   // all 76 columns: the filters read whole dwords
        const int ly = i / IS, lx = i - ly * IS;
        const int gy = clampi(y0 - 5 + ly, 0, H - 1), gx = clampi(x0 - 5 + lx, 0, W - 1);
        const uint8_t* p = src + ((size_t)gy * W + gx) * 3;
        s_in[(0 * IH + ly) * IS + lx] = p[0];
        s_in[(1 * IH + ly) * IS + lx] = p[1];
        s_in[(2 * IH + ly) * IS + lx] = p[2];
      }
    }
  });

  // ---- P: cv::pyrDown of the source tile for the next level: 5 x 5 [1 4 6 4 1]^2, (s + 128) >> 8, BORDER_REFLECT_101 -----------------------
  // Output (Y, X) of the half-size image reads source rows 2Y-2 .. 2Y+2 and columns 2X-2 .. 2X+2 = tile rows 2Yl+3 .. 2Yl+7, tile columns
  // 2Xl+3 .. 2Xl+7 (Yl = Y - y0/2, Xl = X - x0/2).  Reflection (image sides >= 4) only ever changes the first and the last output row /
  // column of the IMAGE: there the five weights fold onto three or four pixels -- (., ., 6, 8, 2) at 0, (1, 4, 7, 4, .) at the far side of
  // an even length -- and the clamped halo of s_in is never read.  Ph: horizontal pass on bytes, two outputs from three dwords; the five
  // bytes of output 2j start at byte 3 of dword j, those of output 2j + 1 at byte 1 of dword j + 1, so the weights sit in shifted byte lanes
  // instead of the data being funnel-shifted.  Results (<= 4080) are stored as row pairs for the vertical dot2.
  if (pyr_dst != nullptr && !(LMX_CQ_SKIP & 2)) {
    const int Hd = H >> 1, Wd = W >> 1, X0 = x0 >> 1, Y0 = y0 >> 1;
    const bool edge_x = X0 == 0 || X0 + 32 >= Wd;        // block-uniform: the tile holds output column 0 or Wd - 1
    const bool w_even = (W & 1) == 0, h_even = (H & 1) == 0;
    run([&](int tid) {
      for (int i = tid; i < 3 * PP * 16; i += 256) {
        const int R = i >> 4, j = i & 15;                           // R = c * PP + (q - 1)
        const int c = (R >= PP) + (R >= 2 * PP);
        const uint32_t* rowA = s_in32 + (2 * R + 3 + 6 * c) * (IS / 4) + j;   // tile row c * IH + 2q + 1  (IH - 2 PP = 6)
        const uint32_t* rowB = rowA + IS / 4;
        uint32_t wa0 = b4(0, 0, 0, 1), wa1 = b4(4, 6, 4, 1), wb0 = b4(0, 1, 4, 6), wb1 = b4(4, 1, 0, 0);
        if (edge_x) {
          const int X = X0 + 2 * j;
          if (X == 0) { wa0 = 0; wa1 = b4(0, 6, 8, 2); }
          if (X == Wd - 1 && w_even) wa1 = b4(4, 7, 4, 0);
          if (X + 1 == Wd - 1 && w_even) { wb0 = b4(0, 1, 4, 7); wb1 = b4(4, 0, 0, 0); }
        }
        const uint32_t a0 = rowA[0], a1 = rowA[1], a2 = rowA[2], c0 = rowB[0], c1 = rowB[1], c2 = rowB[2];
        const uint32_t ea = udot4(a0, wa0, udot4(a1, wa1, 0)), oa = udot4(a1, wb0, udot4(a2, wb1, 0));
        const uint32_t eb = udot4(c0, wa0, udot4(c1, wa1, 0)), ob = udot4(c1, wb0, udot4(c2, wb1, 0));
        u32x2 out;
        out.x = ea | (eb << 16);
        out.y = oa | (ob << 16);
        *reinterpret_cast<u32x2*>(s_r2 + R * 32 + 2 * j) = out;
      }
    });
    // Pv: vertical pass on the row pairs, two neighbouring output pixels (6 bytes of the BGR-interleaved next-level image) per thread
    run([&](int tid) {
      const int Yl = tid >> 4, xp = tid & 15;
      const int Y = Y0 + Yl, X = X0 + 2 * xp;
      if (Yl < TH / 2 && Y < Hd && X < Wd) {
        uint32_t w0 = h2(1, 4), w1 = h2(6, 4), w2 = h2(1, 0);
        if (Y == 0) { w0 = 0; w1 = h2(6, 8); w2 = h2(2, 0); }
        if (Y == Hd - 1 && h_even) { w1 = h2(7, 4); w2 = 0; }
        uint32_t px[2][3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          const uint32_t* p = s_r2 + (c * PP + Yl) * 32 + 2 * xp;     // pairs q = Yl + 1, + 2, + 3
          const u32x2 p0 = *reinterpret_cast<const u32x2*>(p), p1 = *reinterpret_cast<const u32x2*>(p + 32), p2 = *reinterpret_cast<const u32x2*>(p + 64);
          px[0][c] = udot2(p0.x, w0, udot2(p1.x, w1, udot2(p2.x, w2, 128u)));   // <= 255 * 256 + 128: the result is byte 1
          px[1][c] = udot2(p0.y, w0, udot2(p1.y, w1, udot2(p2.y, w2, 128u)));
        }
        uint8_t* out = pyr_dst + (umul24((uint32_t)Y, (uint32_t)Wd) + (uint32_t)X) * 3u;
        const uint32_t bg0 = perm(px[0][1], px[0][0], 0x0c0c0501u);      // b0 g0 . .
        if (X + 1 < Wd) {
          store_u32(out, perm(perm(px[1][0], px[0][2], 0x05010c0cu), bg0, 0x07060100u));   // b0 g0 r0 b1
          store_u16(out + 4, perm(px[1][2], px[1][1], 0x0c0c0501u));                        // g1 r1
        } else {
          out[0] = (uint8_t)(px[0][0] >> 8); out[1] = (uint8_t)(px[0][1] >> 8); out[2] = (uint8_t)(px[0][2] >> 8);
        }
      }
    });
  }

  // ---- Bh: GaussianBlur 7 x 7, horizontal pass on bytes.  h(row, x) = sum_j k_j * s_in[row][x + j], k = {8, 28, 56, 72, 56, 28, 8}, for the 68
  // smoothed columns; a thread takes two rows (a row pair) x four columns: six dwords, twenty dot4 (window offset o inside the first dword ->
  // the weight vector shifted by o bytes, a third dword for o >= 2), results (<= 65 280) packed per column as (row 2p, row 2p + 1).
  run([&](int tid) {
    if (LMX_CQ_SKIP & 4) return;
    int R = tid / 17, xq = tid - R * 17;                    // R = c * HP + p: source rows 2R, 2R + 1 of the flat [3 * IH] plane stack
    for (int i = tid; i < 3 * HP * 17; i += 256) {
      const uint32_t* rowA = s_in32 + (2 * R) * (IS / 4) + xq;
      const uint32_t* rowB = rowA + IS / 4;
      const uint32_t a0 = rowA[0], a1 = rowA[1], a2 = rowA[2], c0 = rowB[0], c1 = rowB[1], c2 = rowB[2];
      u32x4 out;
      out.x = udot4(a0, b4(8, 28, 56, 72), udot4(a1, b4(56, 28, 8, 0), 0)) | (udot4(c0, b4(8, 28, 56, 72), udot4(c1, b4(56, 28, 8, 0), 0)) << 16);
      out.y = udot4(a0, b4(0, 8, 28, 56), udot4(a1, b4(72, 56, 28, 8), 0)) | (udot4(c0, b4(0, 8, 28, 56), udot4(c1, b4(72, 56, 28, 8), 0)) << 16);
      out.z = udot4(a0, b4(0, 0, 8, 28), udot4(a1, b4(56, 72, 56, 28), udot4(a2, b4(8, 0, 0, 0), 0))) |
              (udot4(c0, b4(0, 0, 8, 28), udot4(c1, b4(56, 72, 56, 28), udot4(c2, b4(8, 0, 0, 0), 0))) << 16);
      out.w = udot4(a0, b4(0, 0, 0, 8), udot4(a1, b4(28, 56, 72, 56), udot4(a2, b4(28, 8, 0, 0), 0))) |
              (udot4(c0, b4(0, 0, 0, 8), udot4(c1, b4(28, 56, 72, 56), udot4(c2, b4(28, 8, 0, 0), 0))) << 16);
      *reinterpret_cast<u32x4*>(s_r2 + R * SW + 4 * xq) = out;
      R += 15; xq += 1;                                     // i += 256 = 15 * 17 + 1
      if (xq >= 17) { xq -= 17; R += 1; }
    }
  });

  // ---- Bv: vertical pass on the row pairs, (sum + 2^15) >> 16.  Smoothed rows 2a and 2a + 1 of four columns from four pairs each:
  // row 2a = rows 2a .. 2a+6 = pairs a .. a+3 with weights (k0 k1)(k2 k3)(k4 k5)(k6 .), row 2a + 1 = the same pairs with (. k0)(k1 k2)(k3 k4)(k5 k6).
  run([&](int tid) {
    if (LMX_CQ_SKIP & 8) return;
    int Rv = tid / 17, xq = tid - Rv * 17;                  // Rv = c * (SH / 2) + a
    for (int i = tid; i < 3 * (SH / 2) * 17; i += 256) {
      const int c = (Rv >= SH / 2) + (Rv >= SH);
      const uint32_t* p = s_r2 + (Rv + 3 * c) * SW + 4 * xq;     // pair c * HP + a  (HP - SH / 2 = 3)
      const u32x4 p0 = *reinterpret_cast<const u32x4*>(p), p1 = *reinterpret_cast<const u32x4*>(p + SW), p2 = *reinterpret_cast<const u32x4*>(p + 2 * SW),
                  p3 = *reinterpret_cast<const u32x4*>(p + 3 * SW);
#define CQ_EVEN(f) udot2(p0.f, h2(8, 28), udot2(p1.f, h2(56, 72), udot2(p2.f, h2(56, 28), udot2(p3.f, h2(8, 0), 32768u))))
#define CQ_ODD(f) udot2(p0.f, h2(0, 8), udot2(p1.f, h2(28, 56), udot2(p2.f, h2(72, 56), udot2(p3.f, h2(28, 8), 32768u))))
      // every sum is < 2^24: the smoothed value is byte 2
      const uint32_t e01 = perm(CQ_EVEN(y), CQ_EVEN(x), 0x0c0c0602u), e23 = perm(CQ_EVEN(w), CQ_EVEN(z), 0x06020c0cu);
      const uint32_t o01 = perm(CQ_ODD(y), CQ_ODD(x), 0x0c0c0602u), o23 = perm(CQ_ODD(w), CQ_ODD(z), 0x06020c0cu);
#undef CQ_EVEN
#undef CQ_ODD
      uint32_t* out = reinterpret_cast<uint32_t*>(s_sm) + (2 * Rv) * (SW / 4) + xq;   // smoothed row c * SH + 2a = 2 Rv
      out[0] = e01 | e23;
      out[SW / 4] = o01 | o23;
      Rv += 15; xq += 1;
      if (xq >= 17) { xq -= 17; Rv += 1; }
    }
  });

  // ---- D: Sobel 3 x 3 on the smoothed planes (BORDER_REPLICATE: indices are CLAMPED image coordinates), strongest channel, label, "magnitude^2 >
  // weak^2" flag.  Wave w owns label rows [start, start + count) of label columns 0 .. 63 (a rolling 3-row window down the column); the two halo
  // columns 64, 65 are 2 * QH more pixels, done afterwards by QH lanes of waves 2 and 3 (the waves with fewer rows).  Tiles whose halo-1 label
  // region lies strictly inside the image (block-uniform) skip every clamp, range and border test.  Per label pixel the stage leaves
  //   s_oh = 1 << 4 * label   (one vote for the 3 x 3 histogram of stage E, packed 4-bit counters)     s_fl = 0xff if the flag holds, else 0
  // Image-border pixels carry label 0 (upstream zeroes the first / last row and column before the vote).
  run([&](int tid) {
    if (LMX_CQ_SKIP & 16) return;
    const int thr_i = (int)fminf(floorf(thr_sq), 1.0e9f);   // integer m: (float)m > thr_sq  <=>  m > floor(thr_sq)
    const int w = uniform(tid >> 6), lxq = tid & 63;        // wave-uniform: row indices, clamps and tests go to the scalar unit
    constexpr int DROWS = QH / 4;                           // QH = 4 * DROWS + 2: waves 0 and 1 take one row more
    static_assert(QH % 4 == 2, "label rows over four waves");
    const int start = w < 2 ? (DROWS + 1) * w : DROWS * w + 2, count = w < 2 ? DROWS + 1 : DROWS;
    uint32_t* const s_oh = s_r2;
    auto stage_d = [&](auto interior_tag) {
      constexpr bool INTERIOR = decltype(interior_tag)::value;
      auto emit = [&](int bdx, int bdy, int bm, int ly, int lx, int gy, int gx) {
        uint32_t q = orientation_label8(bdx, bdy);
        if (!INTERIOR) {
          const bool inside = gy >= 0 && gy < H && gx >= 0 && gx < W;
          const bool border = (gy <= 0) | (gy >= H - 1) | (gx <= 0) | (gx >= W - 1);
          if (border) q = 0;
          if (TRAIN && inside && ly >= 1 && ly <= TH && lx >= 1 && lx <= TW) mag_dst[(size_t)gy * W + gx] = (float)bm;
        } else if (TRAIN && ly >= 1 && ly <= TH && lx >= 1 && lx <= TW) {
          mag_dst[(size_t)gy * W + gx] = (float)bm;
        }
        s_oh[ly * QS + lx] = 1u << (4u * q);
        s_fl[ly * QS + lx] = (uint8_t)sign_mask(thr_i - bm);   // 0xff iff bm > thr_i
      };
      // strongest channel; upstream picks the first channel whose magnitude is >= both others: a later one only wins with a strictly greater one
      auto strongest = [](const int (&dx)[3], const int (&dy)[3], int& bdx, int& bdy, int& bm) {
        bm = mul24(dx[0], dx[0]) + mul24(dy[0], dy[0]); bdx = dx[0]; bdy = dy[0];
#pragma unroll
        for (int c = 1; c < 3; ++c) {
          const int m = mul24(dx[c], dx[c]) + mul24(dy[c], dy[c]);
          const uint32_t gt = (uint32_t)sign_mask(bm - m);   // all ones iff m > bm (both < 2^22)
          bm = (int)select_mask((uint32_t)m, (uint32_t)bm, gt);
          bdx = (int)select_mask((uint32_t)dx[c], (uint32_t)bdx, gt);
          bdy = (int)select_mask((uint32_t)dy[c], (uint32_t)bdy, gt);
        }
      };
      {
        const int gx = x0 - 1 + lxq;
        const int cxm = INTERIOR ? lxq : clampi(gx - 1, 0, W - 1) - (x0 - 2);
        const int cxc = INTERIOR ? lxq + 1 : clampi(gx, 0, W - 1) - (x0 - 2);
        const int cxp = INTERIOR ? lxq + 2 : clampi(gx + 1, 0, W - 1) - (x0 - 2);
        int Rw[3][3], Dw[3][3];  // [row slot][channel]: a + 2b + c and c - a of the row's three columns
#pragma unroll
        for (int k = 0; k < DROWS + 3; ++k) {
          if (k < count + 2) {  // wave-uniform
            const int rr = INTERIOR ? start + k : clampi(y0 - 2 + start + k, 0, H - 1) - (y0 - 2);
#pragma unroll
            for (int c = 0; c < 3; ++c) {
              const uint8_t* row = s_sm + (c * SH + rr) * SW;
              // (Tried: ONE unaligned dword read per row and channel instead of three byte reads, a + 2b + c and c - a as dot4s.  Unaligned LDS
              // reads compile to a single ds_read_b32 on gfx950 but run far slower than three ds_read_u8: the kernel went from 0.145 to 0.215 ms
              // per step in a same-box A/B, profiles/r04_color_quantize_ab.txt.  Byte reads stay.)
              const int a = (LMX_CQ_SKIP & 128) ? row[cxc] : row[cxm], b = row[cxc], cc = (LMX_CQ_SKIP & 128) ? row[cxc] : row[cxp];   // (bit 128: timing experiment, one read instead of three)
              Rw[k % 3][c] = a + 2 * b + cc;
              Dw[k % 3][c] = cc - a;
            }
            if (k >= 2) {
              int dx[3], dy[3];
#pragma unroll
              for (int c = 0; c < 3; ++c) {
                dx[c] = Dw[(k - 2) % 3][c] + 2 * Dw[(k - 1) % 3][c] + Dw[k % 3][c];
                dy[c] = Rw[k % 3][c] - Rw[(k - 2) % 3][c];
              }
              int bdx, bdy, bm;
              strongest(dx, dy, bdx, bdy, bm);
              const int ly = start + (k - 2);
              emit(bdx, bdy, bm, ly, lxq, y0 - 1 + ly, gx);
            }
          }
        }
      }
      if (w >= 2 && lxq < QH) {
        const int ly = lxq, lxe = TW + (w - 2);
        const int gx = x0 - 1 + lxe, gy = y0 - 1 + ly;
        const int cx[3] = {INTERIOR ? lxe : clampi(gx - 1, 0, W - 1) - (x0 - 2), INTERIOR ? lxe + 1 : clampi(gx, 0, W - 1) - (x0 - 2),
                           INTERIOR ? lxe + 2 : clampi(gx + 1, 0, W - 1) - (x0 - 2)};
        const int ry[3] = {INTERIOR ? ly : clampi(gy - 1, 0, H - 1) - (y0 - 2), INTERIOR ? ly + 1 : clampi(gy, 0, H - 1) - (y0 - 2),
                           INTERIOR ? ly + 2 : clampi(gy + 1, 0, H - 1) - (y0 - 2)};
        int dx[3], dy[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
          int Rr[3], Dr[3];
#pragma unroll
          for (int k = 0; k < 3; ++k) {
            const uint8_t* row = s_sm + (c * SH + ry[k]) * SW;
            const int a = row[cx[0]], b = row[cx[1]], cc = row[cx[2]];
            Rr[k] = a + 2 * b + cc;
            Dr[k] = cc - a;
          }
          dx[c] = Dr[0] + 2 * Dr[1] + Dr[2];
          dy[c] = Rr[2] - Rr[0];
        }
        int bdx, bdy, bm;
        strongest(dx, dy, bdx, bdy, bm);
        emit(bdx, bdy, bm, ly, lxe, gy, gx);
      }
    };
    if (x0 >= 2 && x0 + TW + 2 <= W && y0 >= 2 && y0 + TH + 2 <= H) stage_d(std::true_type{});
    else stage_d(std::false_type{});
  });

  // ---- E: hysteresisGradient's 3 x 3 vote.  A pixel whose flag holds takes the label that at least 5 of its 9 neighbours (itself included) carry, as
  // a one-hot byte; everything else, and the image's first / last row and column, is 0.  Row sums of three one-hot dwords, three of those per
  // pixel: packed 4-bit counts (<= 9); "some bin >= 5" is (cnt + 0x33333333) & 0x88888888, at most one bin can.  No winner: ffs gives 0, the
  // shift count wraps to 31 and the stored low byte is 0 -- no compare.
  run([&](int tid) {
    if (LMX_CQ_SKIP & 32) return;
    const uint32_t* const s_oh = s_r2;
    const int seg = tid >> 6, lx = tid & 63;
    const int gx = x0 + lx;
    constexpr int ER = TH / 4;   // output rows per wave
    uint32_t rc[ER + 2];
#pragma unroll
    for (int k = 0; k < ER + 2; ++k) {
      const uint32_t* row = s_oh + (seg * ER + k) * QS + lx;
      rc[k] = row[0] + row[1] + row[2];
    }
    const bool interior = x0 >= 1 && x0 + TW + 1 <= W && y0 >= 1 && y0 + TH + 1 <= H;   // block-uniform: every output pixel is an inner pixel of the image
    uint8_t* out = dst + umul24((uint32_t)(y0 + seg * ER), (uint32_t)W) + (uint32_t)gx;
#pragma unroll
    for (int j = 0; j < ER; ++j) {
      const int gy = y0 + seg * ER + j;
      const uint32_t cnt = rc[j] + rc[j + 1] + rc[j + 2];
      const uint32_t mj = (cnt + 0x33333333u) & 0x88888888u;      // nibble >= 8  <=>  >= 5 of the 9 votes
      const uint32_t hot = 1u << (((uint32_t)(ffs32(mj) - 1) >> 2) & 31u);
      uint32_t v = hot & s_fl[(seg * ER + j + 1) * QS + lx + 1];
      if (interior) {
        out[0] = (uint8_t)v;
      } else if (gy < H && gx < W) {
        if (!(gy >= 1 && gy < H - 1 && gx >= 1 && gx < W - 1)) v = 0;
        out[0] = (uint8_t)v;
      }
      out += W;
    }
  });
}

}  // namespace cq
}  // namespace lmx
