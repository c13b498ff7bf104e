// oracle/linemod_oracle.cpp -- CPU restatement of the LINEMOD matching path.
//
// *** TEST INFRASTRUCTURE, NOT PRODUCT CODE. ***
// Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load this library.
// The shipped path (linemod_pose_estimation_amd/csrc -> liblmx.so) never links, loads or calls it.
//
// What it restates.  The reference's hot path is one call,
//     linemod_detector->match(sources, threshold, matches, std::vector<String>(), noArray());
// (reference src/rgbdDetector.cpp:31-34, declared include/linemod_pose_estimation/rgbdDetector.h:150,
// reached from src/linemod_ensenso_detect_3_mult_detect_service.cpp:344 and src/linemod_carmine_detect.cpp:348).
// All arithmetic behind that call lives in OpenCV's cv::linemod (third-party, NOT vendored in
// /root/reference and not pinned there: `find_package(OpenCV REQUIRED)`, reference CMakeLists.txt:22; era
// evidence points at OpenCV 2.4.x objdetect/linemod == opencv_contrib rgbd/linemod, algorithm unchanged
// since).  This file restates that published algorithm stage by stage from SURVEY.md Appendix A
// (A.1 - A.11); each function names the appendix item and the reference call site that fixes its
// parameters (T = {5,8}: reference src/renderer.cpp:182-185; default modality constructors:
// src/renderer.cpp:180-181; class filter empty, no masks: src/rgbdDetector.cpp:33).
//
// PARITY UNPINNED.  The reference has no tests, fixtures or golden vectors for this path
//  and neither OpenCV nor the reference can be built or run in this image, so
// the restatement cannot be checked against upstream outputs.  It is pinned only by known-answer tests
// derived from the published algorithm (tests/test_oracle_kat.py) and by a second, independent numpy
// restatement of the integer stages (tests/np_restatement.py).  One piece is restatement-DEFINED rather
// than restated: upstream's 8000-entry `normal_lut.i` cannot be reproduced here, so DepthNormal labels
// come from the documented generator `lmo_normal_lut` below.
//
// Why C++ and not plain C: the observable result of match() depends on libstdc++'s std::sort (unstable,
// implementation-defined order of ties), std::unique and std::remove_if applied to the matches in
// upstream insertion order (A.10).  Calling the same library routines is the faithful restatement.
//
// Build: see oracle/Makefile (g++ -O3 -ffp-contract=off, no -ffast-math; float ops must keep
// their written order because fastAtan2 feeds a round-half-even quantiser).

#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <map>
#include <string>
#include <vector>

namespace {

typedef unsigned char uchar;
typedef unsigned short ushort;

inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// BORDER_REFLECT_101: gfedcb|abcdefgh|gfedcba
inline int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * (len - 1) - p;
  }
  return p;
}

// ------------------------------------------------------------------------------------------------
// A.6  SIMILARITY_LUT (16 chunks of 16: chunk 2k = orientation k vs low nibble, 2k+1 = vs high nibble)
// ------------------------------------------------------------------------------------------------
const uchar SIMILARITY_LUT[256] = {
    0, 4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4,  0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0,
    0, 3, 4, 4, 3, 3, 4, 4, 2, 3, 4, 4, 3, 3, 4, 4,  0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1, 0, 1,
    0, 2, 3, 3, 4, 4, 4, 4, 3, 3, 3, 3, 4, 4, 4, 4,  0, 2, 1, 2, 0, 2, 1, 2, 0, 2, 1, 2, 0, 2, 1, 2,
    0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4,  0, 3, 2, 3, 1, 3, 2, 3, 0, 3, 2, 3, 1, 3, 2, 3,
    0, 0, 1, 1, 2, 2, 2, 2, 3, 3, 3, 3, 3, 3, 3, 3,  0, 4, 3, 4, 2, 4, 3, 4, 1, 4, 3, 4, 2, 4, 3, 4,
    0, 1, 0, 1, 1, 1, 1, 1, 2, 2, 2, 2, 2, 2, 2, 2,  0, 3, 4, 4, 3, 3, 4, 4, 2, 3, 4, 4, 3, 3, 4, 4,
    0, 2, 1, 2, 0, 2, 1, 2, 1, 2, 1, 2, 1, 2, 1, 2,  0, 2, 3, 3, 4, 4, 4, 4, 3, 3, 3, 3, 4, 4, 4, 4,
    0, 3, 2, 3, 1, 3, 2, 3, 0, 3, 2, 3, 1, 3, 2, 3,  0, 1, 2, 2, 3, 3, 3, 3, 4, 4, 4, 4, 4, 4, 4, 4};

// ------------------------------------------------------------------------------------------------
// A.2 step 1: GaussianBlur(src, 7x7, sigma 0, BORDER_REPLICATE) on 8U, C channels.
// Fixed kernel {8,28,56,72,56,28,8}/256 per axis; exact integer: (sum_y sum_x ky kx p + 2^15) >> 16.
// ------------------------------------------------------------------------------------------------
const int GK7[7] = {8, 28, 56, 72, 56, 28, 8};

void gaussian7(const uchar* src, int H, int W, int C, size_t stride, uchar* dst /* H*W*C packed */) {
  // horizontal pass on an edge-replicated copy of each row (BORDER_REPLICATE), then vertical pass over row pointers
  // clamped at the top/bottom.  Same integers as the per-tap-clamped form; written this way so the inner loops vectorise.
  const int WC = W * C;
  std::vector<unsigned short> rowbuf((size_t)H * WC);  // 8.8 fixed point, <= 255*256 = 65280
  std::vector<uchar> pad((size_t)(W + 6) * C);
  for (int y = 0; y < H; ++y) {
    const uchar* s = src + (size_t)y * stride;
    for (int c = 0; c < C; ++c) {
      for (int k = 0; k < 3; ++k) { pad[k * C + c] = s[c]; pad[(W + 3 + k) * C + c] = s[(W - 1) * C + c]; }
    }
    std::memcpy(&pad[3 * C], s, (size_t)WC);
    unsigned short* r = &rowbuf[(size_t)y * WC];
    const uchar* p = pad.data();
    for (int x = 0; x < WC; ++x)
      r[x] = (unsigned short)(8 * (p[x] + p[x + 6 * C]) + 28 * (p[x + C] + p[x + 5 * C]) + 56 * (p[x + 2 * C] + p[x + 4 * C]) + 72 * p[x + 3 * C]);
  }
  for (int y = 0; y < H; ++y) {
    const unsigned short* r[7];
    for (int k = 0; k < 7; ++k) r[k] = &rowbuf[(size_t)clampi(y + k - 3, 0, H - 1) * WC];
    uchar* d = dst + (size_t)y * WC;
    for (int x = 0; x < WC; ++x) {
      int acc = 8 * (r[0][x] + r[6][x]) + 28 * (r[1][x] + r[5][x]) + 56 * (r[2][x] + r[4][x]) + 72 * r[3][x];
      int v = (acc + (1 << 15)) >> 16;
      d[x] = (uchar)(v > 255 ? 255 : v);
    }
  }
}

// A.2 step 2: Sobel 3x3 -> s16, BORDER_REPLICATE, scale 1.  dx = [1 2 1]^T x [-1 0 1], dy = [-1 0 1]^T x [1 2 1].
void sobel3(const uchar* sm, int H, int W, int C, short* dx, short* dy) {
  // edge-replicated copy (BORDER_REPLICATE), then branch-free 3x3 stencils that vectorise
  const int WC = W * C, PW = (W + 2) * C;
  std::vector<uchar> pad((size_t)(H + 2) * PW);
  for (int y = -1; y <= H; ++y) {
    const uchar* s = sm + (size_t)clampi(y, 0, H - 1) * WC;
    uchar* p = &pad[(size_t)(y + 1) * PW];
    for (int c = 0; c < C; ++c) { p[c] = s[c]; p[(W + 1) * C + c] = s[(W - 1) * C + c]; }
    std::memcpy(p + C, s, (size_t)WC);
  }
  for (int y = 0; y < H; ++y) {
    const uchar* __restrict r0 = &pad[(size_t)y * PW];
    const uchar* __restrict r1 = r0 + PW;
    const uchar* __restrict r2 = r1 + PW;
    short* __restrict ox = dx + (size_t)y * WC;
    short* __restrict oy = dy + (size_t)y * WC;
    for (int x = 0; x < WC; ++x) {
      // columns x-1, x, x+1 of the padded rows sit at offsets x, x+C, x+2C
      int gx = (r0[x + 2 * C] + 2 * r1[x + 2 * C] + r2[x + 2 * C]) - (r0[x] + 2 * r1[x] + r2[x]);
      int gy = (r2[x] + 2 * r2[x + C] + r2[x + 2 * C]) - (r0[x] + 2 * r0[x + C] + r0[x + 2 * C]);
      ox[x] = (short)gx;
      oy[x] = (short)gy;
    }
  }
}

// A.2 step 4: cv::fastAtan2 in degrees (polynomial form; operation order is part of the spec).
inline float fast_atan2_deg(float y, float x) {
  const float p1 = 0.9997878412794807f * (float)(180 / 3.14159265358979323846);
  const float p3 = -0.3258083974640975f * (float)(180 / 3.14159265358979323846);
  const float p5 = 0.1555786518463281f * (float)(180 / 3.14159265358979323846);
  const float p7 = -0.04432655554792128f * (float)(180 / 3.14159265358979323846);
  float ax = std::fabs(x), ay = std::fabs(y);
  float a, c, c2;
  if (ax >= ay) {
    c = ay / (ax + (float)DBL_EPSILON);
    c2 = c * c;
    a = (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  } else {
    c = ax / (ay + (float)DBL_EPSILON);
    c2 = c * c;
    a = 90.f - (((p7 * c2 + p5) * c2 + p3) * c2 + p1) * c;
  }
  if (x < 0) a = 180.f - a;
  if (y < 0) a = 360.f - a;
  return a;
}

// saturate_cast<uchar>(float): round half to even, clamp to [0,255]
inline uchar sat_u8_rint(float v) {
  int i = (int)lrintf(v);  // default rounding mode = nearest even
  return (uchar)(i < 0 ? 0 : (i > 255 ? 255 : i));
}

// A.2: quantizedOrientations(src 8UC3) -> magnitude (f32, squared), quantized angle (u8 one-hot)
void quantized_orientations(const uchar* src, int H, int W, size_t stride, float weak_threshold, uchar* quant,
                            float* magnitude, uchar* q_unfiltered_out /* optional */) {
  const int C = 3;
  std::vector<uchar> smoothed((size_t)H * W * C);
  gaussian7(src, H, W, C, stride, smoothed.data());
  std::vector<short> dx((size_t)H * W * C), dy((size_t)H * W * C);
  sobel3(smoothed.data(), H, W, C, dx.data(), dy.data());

  std::vector<float> angle((size_t)H * W);
  for (size_t i = 0; i < (size_t)H * W; ++i) {
    int x0 = dx[i * 3 + 0], y0 = dy[i * 3 + 0];
    int x1 = dx[i * 3 + 1], y1 = dy[i * 3 + 1];
    int x2 = dx[i * 3 + 2], y2 = dy[i * 3 + 2];
    int m0 = x0 * x0 + y0 * y0, m1 = x1 * x1 + y1 * y1, m2 = x2 * x2 + y2 * y2;
    float sx, sy;
    if (m0 >= m1 && m0 >= m2) { sx = (float)x0; sy = (float)y0; magnitude[i] = (float)m0; }
    else if (m1 >= m0 && m1 >= m2) { sx = (float)x1; sy = (float)y1; magnitude[i] = (float)m1; }
    else { sx = (float)x2; sy = (float)y2; magnitude[i] = (float)m2; }
    angle[i] = fast_atan2_deg(sy, sx);
  }

  // hysteresisGradient (A.2 step 5)
  const float threshold = weak_threshold * weak_threshold;
  std::vector<uchar> qu((size_t)H * W);
  const float scale = (float)(16.0 / 360.0);
  for (size_t i = 0; i < (size_t)H * W; ++i) qu[i] = sat_u8_rint(angle[i] * scale);
  for (int x = 0; x < W; ++x) { qu[x] = 0; qu[(size_t)(H - 1) * W + x] = 0; }
  for (int y = 0; y < H; ++y) { qu[(size_t)y * W] = 0; qu[(size_t)y * W + W - 1] = 0; }
  for (int y = 1; y < H - 1; ++y)
    for (int x = 1; x < W - 1; ++x) qu[(size_t)y * W + x] &= 7;
  if (q_unfiltered_out) std::memcpy(q_unfiltered_out, qu.data(), (size_t)H * W);

  std::memset(quant, 0, (size_t)H * W);
  for (int y = 1; y < H - 1; ++y)
    for (int x = 1; x < W - 1; ++x) {
      if (magnitude[(size_t)y * W + x] > threshold) {
        int hist[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy2 = -1; dy2 <= 1; ++dy2)
          for (int dx2 = -1; dx2 <= 1; ++dx2) hist[qu[(size_t)(y + dy2) * W + (x + dx2)] & 7]++;
        int max_votes = 0, index = -1;
        for (int i = 0; i < 8; ++i)
          if (max_votes < hist[i]) { index = i; max_votes = hist[i]; }
        if (max_votes >= 5) quant[(size_t)y * W + x] = (uchar)(1 << index);
      }
    }
}

// A.3: cv::pyrDown, 5-tap [1 4 6 4 1] separable, (s+128)>>8, BORDER_REFLECT_101, dst = (W/2, H/2)
void pyrdown_u8(const uchar* src, int H, int W, int C, size_t stride, uchar* dst) {
  int Hd = H / 2, Wd = W / 2;
  std::vector<unsigned short> rows((size_t)H * Wd * C);  // <= 255*16
  std::vector<int> cx((size_t)Wd * 5);
  for (int x = 0; x < Wd; ++x)
    for (int k = 0; k < 5; ++k) cx[(size_t)x * 5 + k] = reflect101(2 * x + k - 2, W) * C;
  for (int y = 0; y < H; ++y) {
    const uchar* s = src + (size_t)y * stride;
    unsigned short* r = &rows[(size_t)y * Wd * C];
    for (int x = 0; x < Wd; ++x) {
      const int* ix = &cx[(size_t)x * 5];
      for (int c = 0; c < C; ++c)
        r[x * C + c] = (unsigned short)(s[ix[0] + c] + 4 * s[ix[1] + c] + 6 * s[ix[2] + c] + 4 * s[ix[3] + c] + s[ix[4] + c]);
    }
  }
  const int WC = Wd * C;
  for (int y = 0; y < Hd; ++y) {
    const unsigned short* r[5];
    for (int k = 0; k < 5; ++k) r[k] = &rows[(size_t)reflect101(2 * y + k - 2, H) * WC];
    uchar* d = dst + (size_t)y * WC;
    for (int x = 0; x < WC; ++x) d[x] = (uchar)((r[0][x] + 4 * r[1][x] + 6 * r[2][x] + 4 * r[3][x] + r[4][x] + 128) >> 8);
  }
}

// ------------------------------------------------------------------------------------------------
// A.4 DepthNormal.  NORMAL_LUT is DATA: upstream indexes `static const uchar NORMAL_LUT[20][20][20]` (its normal_lut.i, a
// table of one-hot labels) as NORMAL_LUT[v3][v2][v1], i.e. byte v3*400 + v2*20 + v1 of 8000.  The table is a parameter of the
// detector here (lmo_detector_set_normal_lut), exactly like in liblmx (lmx_bank_set_normal_lut), so a holder of upstream's
// normal_lut.i gets upstream's labels.  v1, v2 can reach 20 (nx or ny == 1.0f) and v3 reaches 20 when nz rounds to 0: C's flat
// array arithmetic then lands in the next row / plane, which is reproduced (flat index); flat indices >= 8000 are an
// out-of-bounds read upstream (UB) and are DEFINED to yield "no label" (0) here and on the device.
// The DEFAULT table is restatement-defined, because normal_lut.i is not in this container: the eight bins are the
// azimuth sectors of the image-plane projection (nx, ny) of the unit normal, sector k centred on k*45 degrees (the "cone of 8
// vectors" of the LINEMOD paper), nz does not enter.  Cell (v2, v1) -> centre cx = 2*v1 - 19, cy = 2*v2 - 19 (odd integers,
// never 0); a = |cx|, b = |cy|: if 2ab < a^2 - b^2 the sector is 0 (cx>0) or 4; else if 2ab < b^2 - a^2 it is 2 (cy>0) or 6;
// else the diagonal 1 / 3 / 5 / 7 by the signs of (cx, cy).  (2ab = |a^2-b^2| has no nonzero integer solution: no ties.)
// ------------------------------------------------------------------------------------------------
inline uchar normal_label_bit(int v2, int v1) {
  int cx = 2 * v1 - 19, cy = 2 * v2 - 19;
  int a = std::abs(cx), b = std::abs(cy);
  int k;
  if (2 * a * b < a * a - b * b) k = cx > 0 ? 0 : 4;
  else if (2 * a * b < b * b - a * a) k = cy > 0 ? 2 : 6;
  else if (cx > 0) k = cy > 0 ? 1 : 7;
  else k = cy > 0 ? 3 : 5;
  return (uchar)(1 << k);
}

const int NORMAL_LUT_SIZE = 20 * 20 * 20;
void default_normal_lut(uchar* out /* [20][20][20] */) {
  for (int v3 = 0; v3 < 20; ++v3)
    for (int v2 = 0; v2 < 20; ++v2)
      for (int v1 = 0; v1 < 20; ++v1) out[(v3 * 20 + v2) * 20 + v1] = normal_label_bit(v2, v1);
}
struct DefaultNormalLut {
  uchar v[NORMAL_LUT_SIZE];
  DefaultNormalLut() { default_normal_lut(v); }
};
const uchar* default_normal_lut_ptr() { static const DefaultNormalLut t; return t.v; }

inline void accum_bilateral(long delta, long i, long j, long* A, long* b, int threshold) {
  long f = std::labs(delta) < threshold ? 1 : 0;
  const long fi = f * i, fj = f * j;
  A[0] += fi * i; A[1] += fi * j; A[3] += fj * j;
  b[0] += fi * delta; b[1] += fj * delta;
}

// median of 5x5 window, BORDER_REPLICATE (cv::medianBlur ksize 5, 8U).  The 13th smallest of 25 is the smallest v with
// count(x <= v) >= 13; v is built bit by bit from the top (8 rounds of 25 compares), a formulation with no data-dependent
// branches so it vectorises across pixels (upstream uses a SIMD sorting network; the value is the same).
typedef unsigned char v32u8 __attribute__((vector_size(32)));

void median5(const uchar* src, int H, int W, uchar* dst) {
  const int Wp = W + 4;
  std::vector<uchar> pad((size_t)(H + 4) * Wp);
  for (int y = -2; y < H + 2; ++y) {
    const uchar* s = src + (size_t)clampi(y, 0, H - 1) * W;
    uchar* p = &pad[(size_t)(y + 2) * Wp];
    p[0] = p[1] = s[0];
    std::memcpy(p + 2, s, (size_t)W);
    p[W + 2] = p[W + 3] = s[W - 1];
  }
  std::vector<uchar> v((size_t)W), cnt((size_t)W), trial((size_t)W);
  for (int y = 0; y < H; ++y) {
    std::fill(v.begin(), v.end(), 0);
    for (int bit = 7; bit >= 0; --bit) {
      const uchar low = (uchar)((1u << bit) - 1);
      uchar* cn = cnt.data();
      uchar* vv = v.data();
      int x = 0;
      for (; x + 32 <= W; x += 32) {  // 32 pixels per step with GCC vector extensions (AVX2 under -march=x86-64-v3)
        v32u8 vcur, t, c = {};
        std::memcpy(&vcur, vv + x, 32);
        t = vcur | low;
        for (int dy = 0; dy < 5; ++dy) {
          const uchar* row = &pad[(size_t)(y + dy) * Wp + x];
          for (int dx = 0; dx < 5; ++dx) {
            v32u8 r;
            std::memcpy(&r, row + dx, 32);
            c -= (v32u8)(r <= t);  // a true lane is 0xFF = -1
          }
        }
        const v32u8 keep = (v32u8)(c >= 13);
        vcur = (vcur & keep) | ((vcur | (uchar)(1u << bit)) & ~keep);
        std::memcpy(vv + x, &vcur, 32);
      }
      for (; x < W; ++x) {
        const uchar t = (uchar)(vv[x] | low);
        int c = 0;
        for (int dy = 0; dy < 5; ++dy)
          for (int dx = 0; dx < 5; ++dx) c += pad[(size_t)(y + dy) * Wp + x + dx] <= t;
        cn[x] = (uchar)c;
        vv[x] = (uchar)(c >= 13 ? vv[x] : (vv[x] | (1u << bit)));
      }
    }
    std::memcpy(dst + (size_t)y * W, v.data(), (size_t)W);
  }
}

void quantized_normals(const ushort* src, int H, int W, size_t stride_elems, int distance_threshold,
                       int difference_threshold, uchar* dst, uchar* pre_median_out /* optional */, const uchar* normal_lut = NULL) {
  if (!normal_lut) normal_lut = default_normal_lut_ptr();
  std::vector<uchar> raw((size_t)H * W, 0);
  const int r = 5;
  const int ox[8] = {-r, 0, r, -r, r, -r, 0, r};
  const int oy[8] = {-r, -r, -r, 0, 0, r, r, r};
  for (int y = r; y < H - r - 1; ++y)
    for (int x = r; x < W - r - 1; ++x) {
      long d = src[(size_t)y * stride_elems + x];
      uchar out = 0;
      if (d < distance_threshold) {
        long A[4] = {0, 0, 0, 0}, b[2] = {0, 0};
        for (int k = 0; k < 8; ++k)
          accum_bilateral((long)src[(size_t)(y + oy[k]) * stride_elems + (x + ox[k])] - d, ox[k], oy[k], A, b,
                          difference_threshold);
        long det = A[0] * A[3] - A[1] * A[1];
        long ddx = A[3] * b[0] - A[1] * b[1];
        long ddy = -A[1] * b[0] + A[0] * b[1];
        float nx = static_cast<float>(1150 * ddx);
        float ny = static_cast<float>(1150 * ddy);
        float nz = static_cast<float>(-det * d);
        float s = sqrtf(nx * nx + ny * ny + nz * nz);
        if (s > 0) {
          float inv = 1.0f / s;
          nx *= inv; ny *= inv; nz *= inv;
          int v1 = static_cast<int>(nx * 10 + 10);
          int v2 = static_cast<int>(ny * 10 + 10);
          int v3 = static_cast<int>(nz * 20 + 20);
          const int idx = (v3 * 20 + v2) * 20 + v1;  // NORMAL_LUT[v3][v2][v1] as C lays it out
          out = (idx >= 0 && idx < NORMAL_LUT_SIZE) ? normal_lut[idx] : 0;
        }
      }
      raw[(size_t)y * W + x] = out;
    }
  if (pre_median_out) std::memcpy(pre_median_out, raw.data(), (size_t)H * W);
  median5(raw.data(), H, W, dst);
}

// A.5 spread
void spread(const uchar* src, int H, int W, int T, uchar* dst) {
  std::memset(dst, 0, (size_t)H * W);
  for (int r = 0; r < T; ++r)
    for (int c = 0; c < T; ++c)
      for (int y = 0; y < H - r; ++y) {
        const uchar* s = src + (size_t)(y + r) * W + c;
        uchar* d = dst + (size_t)y * W;
        for (int x = 0; x < W - c; ++x) d[x] |= s[x];
      }
}

// A.6 response maps
void response_maps(const uchar* src, int H, int W, uchar* maps /* [8][H*W] */) {
  size_t n = (size_t)H * W;
  for (int ori = 0; ori < 8; ++ori) {
    const uchar* lo = SIMILARITY_LUT + 32 * ori;
    const uchar* hi = lo + 16;
    uchar* m = maps + ori * n;
    for (size_t i = 0; i < n; ++i) m[i] = std::max(lo[src[i] & 15], hi[src[i] >> 4]);
  }
}

// A.7 linearize
void linearize(const uchar* map, int H, int W, int T, uchar* lin /* [T*T][(W/T)*(H/T)] */) {
  uchar* m = lin;
  for (int r0 = 0; r0 < T; ++r0)
    for (int c0 = 0; c0 < T; ++c0)
      for (int r = r0; r < H; r += T)
        for (int c = c0; c < W; c += T) *m++ = map[(size_t)r * W + c];
}

struct Feature { int x, y, label; };
struct Template { int width, height, pyramid_level; std::vector<Feature> features; };
typedef std::vector<Template> TemplatePyramid;

// One modality's linear memories at one level: 8 orientations x [T*T][W'*H'] each its own matrix
// (upstream: vector<Mat>).  Reads past the end of an orientation's matrix are upstream UB; here they
// are DEFINED to return 0 (documented in DESIGN.md).
struct LinearMemories {
  int T, Wc, Hc;
  std::vector<uchar> mem[8];
  inline uchar at(int label, long idx) const { return (idx >= 0 && idx < (long)mem[label].size()) ? mem[label][idx] : 0; }
};

inline long lm_base(const Feature& f, int T, int Wc, long row_len) {
  int grid_index = (f.y % T) * T + (f.x % T);
  long lm_index = (long)(f.y / T) * Wc + (f.x / T);
  return grid_index * row_len + lm_index;
}

// A.8 similarity
void similarity(const LinearMemories& lm, const Template& templ, std::vector<uchar>& dst, int size_w, int size_h, int T) {
  int W = size_w / T, H = size_h / T;
  int wf = (templ.width - 1) / T + 1, hf = (templ.height - 1) / T + 1;
  int span_x = W - wf, span_y = H - hf;
  int template_positions = span_y * W + span_x + 1;
  dst.assign((size_t)W * H, 0);
  if (template_positions > W * H) template_positions = W * H;  // cannot happen for width,height >= 1
  long row_len = (long)W * H;
  for (size_t i = 0; i < templ.features.size(); ++i) {
    Feature f = templ.features[i];
    if (f.x < 0 || f.x >= size_w || f.y < 0 || f.y >= size_h) continue;
    long base = lm_base(f, T, W, row_len);
    if (base + template_positions <= (long)lm.mem[f.label].size()) {
      // in-bounds: the contiguous byte add upstream vectorises with _mm_add_epi8
      const uchar* __restrict p = lm.mem[f.label].data() + base;
      uchar* __restrict d = dst.data();
      for (int j = 0; j < template_positions; ++j) d[j] = (uchar)(d[j] + p[j]);
    } else {
      for (int j = 0; j < template_positions; ++j) dst[j] = (uchar)(dst[j] + lm.at(f.label, base + j));
    }
  }
}

// A.9 similarityLocal
void similarity_local(const LinearMemories& lm, const Template& templ, uchar* dst /* 256 */, int size_w, int size_h,
                      int T, int cx, int cy) {
  int W = size_w / T;
  long row_len = (long)W * (size_h / T);
  std::memset(dst, 0, 256);
  int offset_x = (cx / T - 8) * T, offset_y = (cy / T - 8) * T;
  for (size_t i = 0; i < templ.features.size(); ++i) {
    Feature f = templ.features[i];
    f.x += offset_x; f.y += offset_y;
    if (f.x < 0 || f.y < 0 || f.x >= size_w || f.y >= size_h) continue;
    long base = lm_base(f, T, W, row_len);
    if (base + 15L * W + 16 <= (long)lm.mem[f.label].size()) {
      const uchar* p = lm.mem[f.label].data() + base;
      for (int row = 0; row < 16; ++row, p += W)
        for (int col = 0; col < 16; ++col) dst[row * 16 + col] = (uchar)(dst[row * 16 + col] + p[col]);
    } else {
      for (int row = 0; row < 16; ++row)
        for (int col = 0; col < 16; ++col)
          dst[row * 16 + col] = (uchar)(dst[row * 16 + col] + lm.at(f.label, base + (long)row * W + col));
    }
  }
}

struct Match {
  int x, y; float similarity; int class_index; int template_id;
  const std::string* class_id;
  int slot, coarse_pos;  // bookkeeping only (insertion order of the candidate): not part of upstream's Match
  bool operator<(const Match& rhs) const {
    if (similarity != rhs.similarity) return similarity > rhs.similarity;
    return template_id < rhs.template_id;
  }
  bool operator==(const Match& rhs) const {
    return x == rhs.x && y == rhs.y && similarity == rhs.similarity && *class_id == *rhs.class_id;
  }
};
struct MatchPredicate {
  float threshold;
  bool operator()(const Match& m) const { return m.similarity < threshold; }
};

enum { MOD_COLOR_GRADIENT = 0, MOD_DEPTH_NORMAL = 1 };

struct Modality {
  int type;
  float weak_threshold, strong_threshold; int num_features;        // ColorGradient (A.1 defaults 10, 55, 63)
  int distance_threshold, difference_threshold, extract_threshold;  // DepthNormal (2000, 50, 2)
};

struct Source { const void* data; int rows, cols; size_t stride_bytes; };

struct Detector {
  std::vector<int> T_at_level;
  std::vector<Modality> modalities;
  std::map<std::string, std::vector<TemplatePyramid> > class_templates;
  // intermediates of the last match() (for stage-level parity tests)
  std::vector<std::vector<uchar> > last_quantized;            // [l*M+m] H_l*W_l
  std::vector<LinearMemories> last_lm;                        // [l*M+m]
  std::vector<std::pair<int, int> > last_sizes;               // (w,h) per level
  std::vector<Match> last_matches;
  std::vector<Match> last_raw;
  std::vector<std::string> class_names;                       // index -> id (map order)
  long stat_candidates = 0;                                   // coarse candidates of last match
  std::vector<uchar> normal_lut;                              // empty = the default table; else [20][20][20]
  const uchar* lut() const { return normal_lut.empty() ? NULL : normal_lut.data(); }
};

// A.9 matchClass
void match_class(Detector& det, const std::vector<LinearMemories>& lms /* [l*M+m] */,
                 const std::vector<std::pair<int, int> >& sizes, float threshold, std::vector<Match>& matches,
                 const std::string& class_id, int class_index, const std::vector<TemplatePyramid>& tps, int slot) {
  const int M = (int)det.modalities.size();
  const int L = (int)det.T_at_level.size();
  std::vector<std::vector<uchar> > sims(M);
  for (size_t template_id = 0; template_id < tps.size(); ++template_id) {
    const TemplatePyramid& tp = tps[template_id];
    int lowest_start = (int)tp.size() - M;
    int lowest_T = det.T_at_level.back();
    int sw = sizes.back().first, sh = sizes.back().second;
    int W = sw / lowest_T, H = sh / lowest_T;
    int num_features = 0;
    for (int i = 0; i < M; ++i) {
      const Template& templ = tp[lowest_start + i];
      num_features += (int)templ.features.size();
      similarity(lms[(L - 1) * M + i], templ, sims[i], sw, sh, lowest_T);
    }
    std::vector<ushort> total((size_t)W * H);
    for (size_t j = 0; j < total.size(); ++j) {
      int s = 0;
      for (int i = 0; i < M; ++i) s += sims[i][j];
      total[j] = (ushort)s;
    }
    int raw_threshold = static_cast<int>(2 * num_features + (threshold / 100.f) * (2 * num_features) + 0.5f);
    std::vector<Match> candidates;
    for (int r = 0; r < H; ++r)
      for (int c = 0; c < W; ++c) {
        int raw_score = total[(size_t)r * W + c];
        if (raw_score > raw_threshold) {
          int offset = lowest_T / 2 + (lowest_T % 2 - 1);
          Match m;
          m.x = c * lowest_T + offset; m.y = r * lowest_T + offset;
          m.similarity = (raw_score * 100.f) / (4 * num_features) + 0.5f;
          m.class_id = &class_id; m.class_index = class_index; m.template_id = (int)template_id;
          m.slot = slot; m.coarse_pos = r * W + c;
          candidates.push_back(m);
        }
      }
    det.stat_candidates += (long)candidates.size();

    for (int l = L - 2; l >= 0; --l) {
      int T = det.T_at_level[l];
      int start = l * M;
      int sw2 = sizes[l].first, sh2 = sizes[l].second;
      int border = 8 * T;
      int offset = T / 2 + (T % 2 - 1);
      int max_x = sw2 - tp[start].width - border;
      int max_y = sh2 - tp[start].height - border;
      uchar loc[256];
      for (size_t mi = 0; mi < candidates.size(); ++mi) {
        Match& match2 = candidates[mi];
        int x = match2.x * 2 + 1, y = match2.y * 2 + 1;
        x = std::max(x, border); y = std::max(y, border);
        x = std::min(x, max_x); y = std::min(y, max_y);
        int numFeatures = 0;
        ushort total2[256];
        std::memset(total2, 0, sizeof(total2));
        for (int i = 0; i < M; ++i) {
          const Template& templ = tp[start + i];
          numFeatures += (int)templ.features.size();
          similarity_local(lms[l * M + i], templ, loc, sw2, sh2, T, x, y);
          for (int j = 0; j < 256; ++j) total2[j] = (ushort)(total2[j] + loc[j]);
        }
        int best_score = 0, best_r = -1, best_c = -1;
        for (int r = 0; r < 16; ++r)
          for (int c = 0; c < 16; ++c) {
            int score = total2[r * 16 + c];
            if (score > best_score) { best_score = score; best_r = r; best_c = c; }
          }
        match2.x = (x / T - 8 + best_c) * T + offset;
        match2.y = (y / T - 8 + best_r) * T + offset;
        match2.similarity = (best_score * 100.f) / (4 * numFeatures);
      }
      MatchPredicate pred; pred.threshold = threshold;
      candidates.erase(std::remove_if(candidates.begin(), candidates.end(), pred), candidates.end());
    }
    matches.insert(matches.end(), candidates.begin(), candidates.end());
  }
}

// Build quantized images + linear memories for all levels/modalities (A.10 first half)
// `masks` (or NULL): Detector::match's last argument, one packed level-0 mask per modality (empty = no mask).  Upstream keeps the mask in
// the QuantizedPyramid: quantize() copies the level's labels THROUGH it (`angle.copyTo(dst, mask)` / `normal.copyTo(dst, mask)`), pyrDown()
// halves it with resize(INTER_NEAREST) -- while DepthNormal's next level is resized from the UNMASKED normal image.
int build_pyramid(Detector& det, const Source* sources, int n_sources, const std::vector<std::vector<uchar> >* masks = NULL) {
  const int M = (int)det.modalities.size();
  const int L = (int)det.T_at_level.size();
  if (n_sources != M) return -1;
  int H0 = sources[0].rows, W0 = sources[0].cols;
  for (int i = 1; i < M; ++i)
    if (sources[i].rows != H0 || sources[i].cols != W0) return -2;
  det.last_quantized.assign((size_t)L * M, std::vector<uchar>());
  det.last_lm.assign((size_t)L * M, LinearMemories());
  det.last_sizes.clear();

  // per-modality pyramid state
  std::vector<std::vector<uchar> > color_src(M);   // current BGR image for ColorGradient
  std::vector<std::vector<uchar> > cur_quant(M);   // current quantized image
  std::vector<std::vector<uchar> > cur_mask(M);    // current level's mask per modality
  if (masks) cur_mask = *masks;
  int H = H0, W = W0;
  for (int l = 0; l < L; ++l) {
    int T = det.T_at_level[l];
    if (l > 0) { H /= 2; W /= 2; }
    if (H % T != 0 || W % T != 0) return -3;      // linearize CV_Assert
    if (((long)H * W) % 16 != 0) return -4;       // computeResponseMaps CV_Assert
    for (int m = 0; m < M; ++m) {
      const Modality& mod = det.modalities[m];
      std::vector<uchar> q((size_t)H * W);
      if (mod.type == MOD_COLOR_GRADIENT) {
        std::vector<float> mag((size_t)H * W);
        if (l == 0) {
          quantized_orientations((const uchar*)sources[m].data, H, W, sources[m].stride_bytes, mod.weak_threshold,
                                 q.data(), mag.data(), NULL);
          // keep a packed copy of the source for pyrDown
          color_src[m].resize((size_t)H * W * 3);
          for (int y = 0; y < H; ++y)
            std::memcpy(&color_src[m][(size_t)y * W * 3], (const uchar*)sources[m].data + (size_t)y * sources[m].stride_bytes,
                        (size_t)W * 3);
        } else {
          std::vector<uchar> next((size_t)H * W * 3);
          pyrdown_u8(color_src[m].data(), H * 2, W * 2, 3, (size_t)W * 2 * 3, next.data());
          color_src[m].swap(next);
          quantized_orientations(color_src[m].data(), H, W, (size_t)W * 3, mod.weak_threshold, q.data(), mag.data(), NULL);
        }
      } else {
        if (l == 0) {
          quantized_normals((const ushort*)sources[m].data, H, W, sources[m].stride_bytes / 2, mod.distance_threshold,
                            mod.difference_threshold, q.data(), NULL, det.lut());
        } else {
          const std::vector<uchar>& prev = cur_quant[m];
          for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) q[(size_t)y * W + x] = prev[(size_t)(2 * y) * (W * 2) + 2 * x];
        }
      }
      cur_quant[m] = q;
      if (!cur_mask[m].empty()) {
        if (l > 0) {   // resize(mask, next_mask, size, 0, 0, INTER_NEAREST)
          std::vector<uchar> next((size_t)H * W);
          for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) next[(size_t)y * W + x] = cur_mask[m][(size_t)(2 * y) * (W * 2) + 2 * x];
          cur_mask[m].swap(next);
        }
        for (size_t i = 0; i < q.size(); ++i)
          if (!cur_mask[m][i]) q[i] = 0;   // quantize(): dst = zeros; labels.copyTo(dst, mask)
      }
      std::vector<uchar> spr((size_t)H * W), maps((size_t)8 * H * W);
      spread(q.data(), H, W, T, spr.data());
      response_maps(spr.data(), H, W, maps.data());
      LinearMemories& lm = det.last_lm[(size_t)l * M + m];
      lm.T = T; lm.Wc = W / T; lm.Hc = H / T;
      for (int o = 0; o < 8; ++o) {
        lm.mem[o].resize((size_t)H * W);
        linearize(maps.data() + (size_t)o * H * W, H, W, T, lm.mem[o].data());
      }
      det.last_quantized[(size_t)l * M + m].swap(q);
    }
    det.last_sizes.push_back(std::make_pair(W, H));
  }
  return 0;
}

}  // namespace

// =================================================================================================
// C API (ctypes-friendly)
// =================================================================================================
extern "C" {

struct lmo_match_t { int32_t x, y; float similarity; int32_t template_id; int32_t class_index; };

const unsigned char* lmo_similarity_lut() { return SIMILARITY_LUT; }

void lmo_normal_lut(unsigned char* out /* [20][20][20] */) { default_normal_lut(out); }

void lmo_gaussian7(const unsigned char* src, int H, int W, int C, size_t stride, unsigned char* dst) { gaussian7(src, H, W, C, stride, dst); }
void lmo_sobel3(const unsigned char* sm, int H, int W, int C, short* dx, short* dy) { sobel3(sm, H, W, C, dx, dy); }
float lmo_fast_atan2(float y, float x) { return fast_atan2_deg(y, x); }
void lmo_quantized_orientations(const unsigned char* src, int H, int W, size_t stride, float weak_threshold,
                                unsigned char* quant, float* magnitude, unsigned char* q_unfiltered) {
  quantized_orientations(src, H, W, stride, weak_threshold, quant, magnitude, q_unfiltered);
}
void lmo_pyrdown(const unsigned char* src, int H, int W, int C, size_t stride, unsigned char* dst) { pyrdown_u8(src, H, W, C, stride, dst); }
void lmo_quantized_normals_lut(const unsigned short* src, int H, int W, size_t stride_elems, int distance_threshold,
                               int difference_threshold, unsigned char* dst, unsigned char* pre_median, const unsigned char* lut /* [8000] or NULL */) {
  quantized_normals(src, H, W, stride_elems, distance_threshold, difference_threshold, dst, pre_median, lut);
}
void lmo_quantized_normals(const unsigned short* src, int H, int W, size_t stride_elems, int distance_threshold,
                           int difference_threshold, unsigned char* dst, unsigned char* pre_median) {
  quantized_normals(src, H, W, stride_elems, distance_threshold, difference_threshold, dst, pre_median);
}
void lmo_median5(const unsigned char* src, int H, int W, unsigned char* dst) { median5(src, H, W, dst); }
void lmo_spread(const unsigned char* src, int H, int W, int T, unsigned char* dst) { spread(src, H, W, T, dst); }
void lmo_response_maps(const unsigned char* src, int H, int W, unsigned char* maps) { response_maps(src, H, W, maps); }
void lmo_linearize(const unsigned char* map, int H, int W, int T, unsigned char* lin) { linearize(map, H, W, T, lin); }

// ---- node-side steps in front of match() -------------------------------------------------------
// cv::GaussianBlur(img, img, Size(3,3), 0, 0): sigma 0, ksize 3 -> fixed kernel {0.25, 0.5, 0.25} per axis; on 8U the
// fixed-point pipeline gives exactly (sum_{dy,dx} k[dy] k[dx] p + 8) >> 4 with k = {1,2,1}; default border REFLECT_101.
// Reference: src/linemod_ensenso_detect_3_mult_detect_service.cpp:325 (blur on the full frame), :324,326 (crop), :293-297 (MONO8).
void lmo_pre_color(const unsigned char* src, int SH, int SW, int SC, size_t stride, int crop_x, int crop_y, int H, int W, int blur3,
                   unsigned char* dst /* H*W*3 */) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x)
      for (int c = 0; c < 3; ++c) {
        const int sc = SC == 1 ? 0 : c, sy = crop_y + y, sx = crop_x + x;
        int v;
        if (blur3) {
          int acc = 0;
          for (int dy = -1; dy <= 1; ++dy)
            for (int dx = -1; dx <= 1; ++dx) {
              const int w = (dy == 0 ? 2 : 1) * (dx == 0 ? 2 : 1);
              acc += w * src[(size_t)reflect101(sy + dy, SH) * stride + (size_t)reflect101(sx + dx, SW) * SC + sc];
            }
          v = (acc + 8) >> 4;
        } else {
          v = src[(size_t)sy * stride + (size_t)sx * SC + sc];
        }
        dst[((size_t)y * W + x) * 3 + c] = (uchar)v;
      }
}
// mat_depth_m.convertTo(mat_depth, CV_16UC1, 1000.0) (service.cpp:853; carmine:829-839): float * 1000.f, cvRound (half to even),
// saturate_cast<ushort>; NaN / Inf / out-of-int-range take x86's integer-indefinite value INT_MIN and saturate to 0.
void lmo_pre_depth(const float* src, size_t stride_elems, int crop_x, int crop_y, int H, int W, unsigned short* dst) {
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      const float v = src[(size_t)(crop_y + y) * stride_elems + (crop_x + x)] * 1000.f;
      unsigned short out;
      if (!(v > -2147483648.f && v < 2147483648.f)) out = 0;
      else {
        const long r = lrintf(v);
        out = (unsigned short)(r < 0 ? 0 : (r > 65535 ? 65535 : r));
      }
      dst[(size_t)y * W + x] = out;
    }
}

// 16-bin label (0..16, before '& 7') of n gradients: saturate_cast<uchar>(fastAtan2(dy, dx) * (16/360)) as in hysteresisGradient
void lmo_orientation_labels(const short* dx, const short* dy, size_t n, unsigned char* out) {
  const float scale = (float)(16.0 / 360.0);
  for (size_t i = 0; i < n; ++i) out[i] = sat_u8_rint(fast_atan2_deg((float)dy[i], (float)dx[i]) * scale);
}
int lmo_raw_threshold(int num_features, float threshold) {
  return static_cast<int>(2 * num_features + (threshold / 100.f) * (2 * num_features) + 0.5f);
}

// ---- detector -----------------------------------------------------------------------------------
// modality_desc: per modality 7 floats {type, weak, strong, num_features, distance_thr, difference_thr, extract_thr}
void* lmo_detector_create(int pyramid_levels, const int* T, int n_modalities, const float* modality_desc) {
  Detector* d = new Detector();
  d->T_at_level.assign(T, T + pyramid_levels);
  for (int m = 0; m < n_modalities; ++m) {
    const float* p = modality_desc + 7 * m;
    Modality mod;
    mod.type = (int)p[0]; mod.weak_threshold = p[1]; mod.strong_threshold = p[2]; mod.num_features = (int)p[3];
    mod.distance_threshold = (int)p[4]; mod.difference_threshold = (int)p[5]; mod.extract_threshold = (int)p[6];
    d->modalities.push_back(mod);
  }
  return d;
}
void lmo_detector_destroy(void* h) { delete (Detector*)h; }
// NORMAL_LUT[20][20][20] of the DepthNormal modality (upstream normal_lut.i); NULL restores the default table
void lmo_detector_set_normal_lut(void* h, const unsigned char* lut) {
  Detector* d = (Detector*)h;
  if (lut) d->normal_lut.assign(lut, lut + NORMAL_LUT_SIZE);
  else d->normal_lut.clear();
}

// templates: int32 [n_pyramids * L*M][5] = {width, height, pyramid_level, feat_begin, feat_count}
// features:  int32 [total][3] = {x, y, label}
int lmo_detector_add_class(void* h, const char* class_id, int n_pyramids, const int32_t* templates, const int32_t* features) {
  Detector* d = (Detector*)h;
  const int per = (int)(d->T_at_level.size() * d->modalities.size());
  std::vector<TemplatePyramid>& v = d->class_templates[class_id];
  for (int p = 0; p < n_pyramids; ++p) {
    TemplatePyramid tp(per);
    for (int k = 0; k < per; ++k) {
      const int32_t* t = templates + ((size_t)p * per + k) * 5;
      tp[k].width = t[0]; tp[k].height = t[1]; tp[k].pyramid_level = t[2];
      if (t[4] > 63) return -1;  // CV_Assert(features.size() <= 63) in similarity()
      for (int f = 0; f < t[4]; ++f) {
        const int32_t* ft = features + ((size_t)t[3] + f) * 3;
        Feature ff; ff.x = ft[0]; ff.y = ft[1]; ff.label = ft[2];
        tp[k].features.push_back(ff);
      }
    }
    v.push_back(tp);
  }
  return (int)v.size();
}

// A.10 Detector::match.  sources: n_sources x {data, rows, cols, stride_bytes}.  class filter: NULL/0 = all.
// Returns number of matches (after sort+unique), or <0 on assertion failure.
long lmo_detector_match_masked(void* h, const void* const* src_data, const int* src_rows, const int* src_cols,
                               const size_t* src_stride, int n_sources, float threshold, const char* const* class_ids,
                               int n_class_ids, const unsigned char* const* mask_data, const size_t* mask_stride);
long lmo_detector_match(void* h, const void* const* src_data, const int* src_rows, const int* src_cols,
                        const size_t* src_stride, int n_sources, float threshold, const char* const* class_ids,
                        int n_class_ids) {
  return lmo_detector_match_masked(h, src_data, src_rows, src_cols, src_stride, n_sources, threshold, class_ids, n_class_ids, NULL, NULL);
}
// Detector::match with its `masks` argument: mask_data[i] = 8UC1 mask of source i (same size), or NULL for "no mask" (an empty Mat)
long lmo_detector_match_masked(void* h, const void* const* src_data, const int* src_rows, const int* src_cols,
                               const size_t* src_stride, int n_sources, float threshold, const char* const* class_ids,
                               int n_class_ids, const unsigned char* const* mask_data, const size_t* mask_stride) {
  Detector& det = *(Detector*)h;
  det.last_matches.clear();
  det.stat_candidates = 0;
  std::vector<Source> sources(n_sources);
  for (int i = 0; i < n_sources; ++i) {
    sources[i].data = src_data[i]; sources[i].rows = src_rows[i]; sources[i].cols = src_cols[i];
    sources[i].stride_bytes = src_stride[i];
  }
  std::vector<std::vector<uchar> > masks(n_sources);
  if (mask_data)
    for (int i = 0; i < n_sources; ++i)
      if (mask_data[i]) {
        masks[i].resize((size_t)src_rows[i] * src_cols[i]);
        for (int y = 0; y < src_rows[i]; ++y) std::memcpy(&masks[i][(size_t)y * src_cols[i]], mask_data[i] + (size_t)y * mask_stride[i], (size_t)src_cols[i]);
      }
  int rc = build_pyramid(det, sources.data(), n_sources, mask_data ? &masks : NULL);
  if (rc != 0) return rc;
  det.class_names.clear();
  for (std::map<std::string, std::vector<TemplatePyramid> >::const_iterator it = det.class_templates.begin();
       it != det.class_templates.end(); ++it)
    det.class_names.push_back(it->first);

  std::vector<Match>& matches = det.last_matches;
  if (n_class_ids == 0) {
    int ci = 0;
    for (std::map<std::string, std::vector<TemplatePyramid> >::const_iterator it = det.class_templates.begin();
         it != det.class_templates.end(); ++it, ++ci)
      match_class(det, det.last_lm, det.last_sizes, threshold, matches, it->first, ci, it->second, ci);
  } else {
    for (int i = 0; i < n_class_ids; ++i) {
      std::map<std::string, std::vector<TemplatePyramid> >::const_iterator it = det.class_templates.find(class_ids[i]);
      if (it != det.class_templates.end()) {
        int ci = (int)std::distance(det.class_templates.cbegin(), it);
        match_class(det, det.last_lm, det.last_sizes, threshold, matches, it->first, ci, it->second, i);
      }
    }
  }
  det.last_raw = matches;  // insertion order, before sort/unique (what GPU shards exchange)
  std::sort(matches.begin(), matches.end());
  matches.erase(std::unique(matches.begin(), matches.end()), matches.end());
  return (long)matches.size();
}

long lmo_detector_get_matches(void* h, lmo_match_t* out, long cap) {
  Detector& det = *(Detector*)h;
  long n = std::min<long>(cap, (long)det.last_matches.size());
  for (long i = 0; i < n; ++i) {
    const Match& m = det.last_matches[i];
    out[i].x = m.x; out[i].y = m.y; out[i].similarity = m.similarity; out[i].template_id = m.template_id;
    out[i].class_index = m.class_index;
  }
  return n;
}
// pre-sort matches in insertion order, in the layout of lmx_raw_match_t (32 bytes)
struct lmo_raw_t { int32_t x, y; float similarity; int32_t template_id; int32_t class_index; int32_t frame; uint64_t order_key; };
long lmo_detector_get_raw(void* h, lmo_raw_t* out, long cap) {
  Detector& det = *(Detector*)h;
  long n = std::min<long>(cap, (long)det.last_raw.size());
  for (long i = 0; i < n; ++i) {
    const Match& m = det.last_raw[i];
    out[i].x = m.x; out[i].y = m.y; out[i].similarity = m.similarity; out[i].template_id = m.template_id;
    out[i].class_index = m.class_index; out[i].frame = 0;
    out[i].order_key = ((uint64_t)(uint32_t)m.slot << 48) | ((uint64_t)(uint32_t)m.template_id << 24) | (uint64_t)(uint32_t)m.coarse_pos;
  }
  return (long)det.last_raw.size();
}
long lmo_detector_last_candidates(void* h) { return ((Detector*)h)->stat_candidates; }

int lmo_detector_num_classes(void* h) { return (int)((Detector*)h)->class_templates.size(); }
const char* lmo_detector_class_name(void* h, int idx) {
  Detector& det = *(Detector*)h;
  int i = 0;
  for (std::map<std::string, std::vector<TemplatePyramid> >::const_iterator it = det.class_templates.begin();
       it != det.class_templates.end(); ++it, ++i)
    if (i == idx) return it->first.c_str();
  return NULL;
}

// intermediates of the last match: quantized image [l*M+m] (H_l*W_l) and linear memories [8][T*T][W'H']
int lmo_detector_get_quantized(void* h, int level, int modality, unsigned char* out) {
  Detector& det = *(Detector*)h;
  size_t idx = (size_t)level * det.modalities.size() + modality;
  if (idx >= det.last_quantized.size()) return -1;
  std::memcpy(out, det.last_quantized[idx].data(), det.last_quantized[idx].size());
  return 0;
}
int lmo_detector_get_linear_memory(void* h, int level, int modality, unsigned char* out) {
  Detector& det = *(Detector*)h;
  size_t idx = (size_t)level * det.modalities.size() + modality;
  if (idx >= det.last_lm.size()) return -1;
  const LinearMemories& lm = det.last_lm[idx];
  size_t n = lm.mem[0].size();
  for (int o = 0; o < 8; ++o) std::memcpy(out + o * n, lm.mem[o].data(), n);
  return 0;
}

// Stand-alone similarity / similarityLocal on caller-provided linear memories (for kernel-level tests).
// lm: [8][T*T][Wc*Hc]; feats int32[n][3]
void lmo_similarity(const unsigned char* lm, int size_w, int size_h, int T, int templ_w, int templ_h,
                    const int32_t* feats, int n_feats, unsigned char* dst) {
  LinearMemories L; L.T = T; L.Wc = size_w / T; L.Hc = size_h / T;
  size_t n = (size_t)size_w * size_h;
  for (int o = 0; o < 8; ++o) L.mem[o].assign(lm + o * n, lm + (o + 1) * n);
  Template t; t.width = templ_w; t.height = templ_h; t.pyramid_level = 0;
  for (int i = 0; i < n_feats; ++i) { Feature f; f.x = feats[3 * i]; f.y = feats[3 * i + 1]; f.label = feats[3 * i + 2]; t.features.push_back(f); }
  std::vector<uchar> d;
  similarity(L, t, d, size_w, size_h, T);
  std::memcpy(dst, d.data(), d.size());
}
void lmo_similarity_local(const unsigned char* lm, int size_w, int size_h, int T, const int32_t* feats, int n_feats,
                          int cx, int cy, unsigned char* dst /* 256 */) {
  LinearMemories L; L.T = T; L.Wc = size_w / T; L.Hc = size_h / T;
  size_t n = (size_t)size_w * size_h;
  for (int o = 0; o < 8; ++o) L.mem[o].assign(lm + o * n, lm + (o + 1) * n);
  Template t; t.width = 0; t.height = 0; t.pyramid_level = 0;
  for (int i = 0; i < n_feats; ++i) { Feature f; f.x = feats[3 * i]; f.y = feats[3 * i + 1]; f.label = feats[3 * i + 2]; t.features.push_back(f); }
  similarity_local(L, t, dst, size_w, size_h, T, cx, cy);
}

}  // extern "C"

// =================================================================================================
// SURVEY.md 8f row 2 -- the reference's OWN post-processing of the match list, restated function by function from
// /root/reference/src/rgbdDetector.cpp: rcd_voting (:36-70), cluster_filter(map) (:72-85), cluster_scoring (:118-130),
// similarity_score_calc (:132-144), nonMaximaSuppressionUsingIOU (:462-530), computeIoU (:532-574); call order from
// src/linemod_ensenso_detect_3_mult_detect_service.cpp:376-447.  Only std types are involved there (Match, Rect as 4 ints),
// so this follows the reference text closely.  One deviation, shared with the product and stated in DESIGN.md:
// cluster_filter erases from the map while iterating over it (undefined behaviour); the evident intent -- drop every
// cluster with size <= thresh -- is implemented.
// =================================================================================================
namespace {
struct PMatch { int x, y; float similarity; int template_id; int src_index; };
struct PRect { int x, y, width, height; };
struct ClusterData {
  ClusterData(const std::vector<int>& index_, double score_) : index(index_), score(score_), is_checked(false), rect{0, 0, 0, 0} {}
  std::vector<PMatch> matches;
  std::vector<int> index;
  double score;
  bool is_checked;
  PRect rect;
};
typedef std::map<std::vector<int>, std::vector<PMatch> > MapMatch;

void rcd_voting(const double* Obj_origin_dists, const double& renderer_radius_min, const int& vote_row_col_step,
                const double& renderer_radius_step_, const std::vector<PMatch>& matches, MapMatch& map_match) {
  int voting_width_step = vote_row_col_step;
  int voting_height_step = vote_row_col_step;
  float voting_depth_step = renderer_radius_step_;
  for (size_t k = 0; k < matches.size(); ++k) {
    const PMatch& match = matches[k];
    int height_index = match.y / voting_height_step;
    int width_index = match.x / voting_width_step;
    float depth = Obj_origin_dists[match.template_id];
    int depth_index = (int)((depth - renderer_radius_min) / voting_depth_step);
    std::vector<int> index(3);
    index[0] = height_index; index[1] = width_index; index[2] = depth_index;
    if (map_match.find(index) == map_match.end()) {
      std::vector<PMatch> temp;
      temp.push_back(match);
      map_match.insert(std::pair<std::vector<int>, std::vector<PMatch> >(index, temp));
    } else {
      map_match[index].push_back(match);
    }
  }
}

void cluster_filter(MapMatch& map_match, int thresh) {
  for (MapMatch::iterator it = map_match.begin(); it != map_match.end();) {
    if ((int)it->second.size() <= thresh) map_match.erase(it++);  // reference: erase(it) then ++it (UB); intent kept
    else ++it;
  }
}

double similarity_score_calc(std::vector<PMatch> match_cluster) {
  double sum_score = 0.0;
  int num = 0;
  for (std::vector<PMatch>::iterator it_match = match_cluster.begin(); it_match != match_cluster.end(); ++it_match) {
    sum_score += it_match->similarity;
    num++;
  }
  sum_score /= num;
  return sum_score;
}

void cluster_scoring(MapMatch& map_match, std::vector<ClusterData>& cluster_data) {
  for (MapMatch::iterator it_map = map_match.begin(); it_map != map_match.end(); ++it_map) {
    double score = similarity_score_calc(it_map->second);
    cluster_data.push_back(ClusterData(it_map->first, score));
  }
}

bool sortScoreCluster(const ClusterData& cluster1, const ClusterData& cluster2) { return (cluster1.score > cluster2.score); }

// The reference computes in plain int; for rects near 2^32 / n (see the size_t division above) its sums and products overflow and, on its
// platform, wrap.  The wrap is written out here (W* helpers) so that this checker does not depend on what a compiler makes of signed overflow.
static inline int Wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
static inline int Wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
static inline int Wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
float computeIoU(PRect rect1, PRect rect2) {
  int rect1_minX, rect1_minY, rect1_maxX, rect1_maxY;
  int rect2_minX, rect2_minY, rect2_maxX, rect2_maxY;
  rect1_minX = rect1.x; rect1_maxX = Wsub(Wadd(rect1.x, rect1.width), 1); rect1_minY = rect1.y; rect1_maxY = Wsub(Wadd(rect1.y, rect1.height), 1);
  rect2_minX = rect2.x; rect2_maxX = Wsub(Wadd(rect2.x, rect2.width), 1); rect2_minY = rect2.y; rect2_maxY = Wsub(Wadd(rect2.y, rect2.height), 1);
  int minX = std::max(rect1_minX, rect2_minX);
  int maxX = std::min(rect1_maxX, rect2_maxX);
  int minY = std::max(rect1_minY, rect2_minY);
  int maxY = std::min(rect1_maxY, rect2_maxY);
  bool is_x_inter = false, is_y_inter = false;
  if ((minX >= rect1_minX && minX <= rect1_maxX) || (minX >= rect2_minX && minX <= rect2_maxX)) is_x_inter = true;
  if ((minY >= rect1_minY && minY <= rect1_maxY) || (minY >= rect2_minY && minY <= rect2_maxY)) is_y_inter = true;
  float inter_area;
  if (is_x_inter && is_y_inter) inter_area = Wmul(Wadd(Wsub(maxX, minX), 1), Wadd(Wsub(maxY, minY), 1));
  else inter_area = 0.0;
  float union_area = Wadd(Wmul(rect1.width, rect1.height), Wmul(rect2.width, rect2.height)) - inter_area;
  float IoU = inter_area / union_area;
  return IoU;
}

void nonMaximaSuppressionUsingIOU(std::vector<ClusterData>& cluster_data, const PRect* Rects_, MapMatch& map_match) {
  std::vector<ClusterData>::iterator it1 = cluster_data.begin();
  for (; it1 != cluster_data.end(); ++it1) {
    MapMatch::iterator it2 = map_match.find(it1->index);
    it1->matches = it2->second;
    int X = 0, Y = 0, WIDTH = 0, HEIGHT = 0;
    for (std::vector<PMatch>::iterator it3 = it1->matches.begin(); it3 != it1->matches.end(); ++it3) {
      PRect tmp = Rects_[it3->template_id];
      X += it3->x; Y += it3->y;
      WIDTH += tmp.width; HEIGHT += tmp.height;
    }
    X /= it1->matches.size(); Y /= it1->matches.size(); WIDTH /= it1->matches.size(); HEIGHT /= it1->matches.size();
    it1->rect = PRect{X, Y, WIDTH, HEIGHT};
  }
  std::sort(cluster_data.begin(), cluster_data.end(), sortScoreCluster);
  for (it1 = cluster_data.begin(); it1 != cluster_data.end(); ++it1) {
    if (!it1->is_checked) {
      std::vector<ClusterData>::iterator it2 = it1;
      it2++;
      for (; it2 != cluster_data.end(); ++it2) {
        if (!it2->is_checked) {
          double IoU = computeIoU(it1->rect, it2->rect);
          if (IoU > 0.4) it2->is_checked = true;
        }
      }
    }
  }
  std::vector<ClusterData> nms_cluster_data;
  for (it1 = cluster_data.begin(); it1 != cluster_data.end(); ++it1)
    if (!it1->is_checked) nms_cluster_data.push_back(*it1);
  cluster_data.clear();
  cluster_data = nms_cluster_data;
}
}  // namespace

extern "C" {
struct lmo_cluster_t { int32_t index[3]; int32_t rect[4]; double score; int32_t member_begin, member_count; };
// matches: lmo_match_t[n]; rects: int32[n_templates][4]; returns the number of clusters (members: indices into matches)
long lmo_cluster_matches(const lmo_match_t* matches, long n, const double* obj_origin_dists, const int32_t* rects, int vote_row_col_step,
                         double renderer_radius_min, double renderer_radius_step, int thresh, lmo_cluster_t* clusters, int32_t* members) {
  std::vector<PMatch> ms;
  for (long i = 0; i < n; ++i) ms.push_back(PMatch{matches[i].x, matches[i].y, matches[i].similarity, matches[i].template_id, (int)i});
  MapMatch map_match;
  rcd_voting(obj_origin_dists, renderer_radius_min, vote_row_col_step, renderer_radius_step, ms, map_match);
  cluster_filter(map_match, thresh);
  std::vector<ClusterData> cluster_data;
  cluster_scoring(map_match, cluster_data);
  if (cluster_data.size() != 0) nonMaximaSuppressionUsingIOU(cluster_data, reinterpret_cast<const PRect*>(rects), map_match);
  long nm = 0;
  for (size_t c = 0; c < cluster_data.size(); ++c) {
    const ClusterData& cd = cluster_data[c];
    for (int k = 0; k < 3; ++k) clusters[c].index[k] = cd.index[k];
    clusters[c].rect[0] = cd.rect.x; clusters[c].rect[1] = cd.rect.y; clusters[c].rect[2] = cd.rect.width; clusters[c].rect[3] = cd.rect.height;
    clusters[c].score = cd.score;
    clusters[c].member_begin = (int32_t)nm; clusters[c].member_count = (int32_t)cd.matches.size();
    for (size_t k = 0; k < cd.matches.size(); ++k) members[nm++] = cd.matches[k].src_index;
  }
  return (long)cluster_data.size();
}
}  // extern "C"

// =================================================================================================
// SURVEY.md Appendix A.11 / 8f row 3 -- trainer side: Detector::addTemplate, restated in upstream's own structure
// (Modality::process -> QuantizedPyramid::{pyrDown, extractTemplate}, selectScatteredFeatures, cropTemplates).
// Reference call sites: /root/reference/src/renderer.cpp:308, src/renderer_only_image.cpp:266.
// =================================================================================================
namespace {
struct Candidate {
  Candidate(int x, int y, int label, float score_) : score(score_) { f.x = x; f.y = y; f.label = label; }
  bool operator<(const Candidate& rhs) const { return score > rhs.score; }  // sort candidates with high score to the front
  Feature f;
  float score;
};

inline int getLabel(int quantized) {
  switch (quantized) {
    case 1: return 0; case 2: return 1; case 4: return 2; case 8: return 3;
    case 16: return 4; case 32: return 5; case 64: return 6; case 128: return 7;
    default: return -1;  // upstream: CV_Error
  }
}

// cv::erode(src, dst, Mat(), Point(-1,-1), iterations, BORDER_REPLICATE): 3x3 rectangular minimum, repeated
void erode_replicate(const std::vector<uchar>& src, int H, int W, int iterations, std::vector<uchar>& dst) {
  std::vector<uchar> cur = src, nxt(src.size());
  for (int it = 0; it < iterations; ++it) {
    for (int y = 0; y < H; ++y)
      for (int x = 0; x < W; ++x) {
        uchar v = 255;
        for (int dy = -1; dy <= 1; ++dy)
          for (int dx = -1; dx <= 1; ++dx) v = std::min(v, cur[(size_t)clampi(y + dy, 0, H - 1) * W + clampi(x + dx, 0, W - 1)]);
        nxt[(size_t)y * W + x] = v;
      }
    cur.swap(nxt);
  }
  dst = cur;
}

// cv::distanceTransform(src, dst, DIST_C, 3): chessboard distance to the nearest zero pixel of src.  Brute force over a growing
// square ring would be quadratic; the classic two-pass chamfer with unit weights is exact for the chessboard metric.
void distance_transform_c(const std::vector<uchar>& src, int H, int W, std::vector<float>& dst) {
  const int INF = 1 << 28;  // stands for upstream's initial distance: no zero pixel reachable
  std::vector<int> d((size_t)H * W);
  for (size_t i = 0; i < d.size(); ++i) d[i] = src[i] ? INF : 0;
  for (int y = 0; y < H; ++y)
    for (int x = 0; x < W; ++x) {
      int& v = d[(size_t)y * W + x];
      if (y > 0) {
        if (x > 0) v = std::min(v, d[(size_t)(y - 1) * W + x - 1] + 1);
        v = std::min(v, d[(size_t)(y - 1) * W + x] + 1);
        if (x + 1 < W) v = std::min(v, d[(size_t)(y - 1) * W + x + 1] + 1);
      }
      if (x > 0) v = std::min(v, d[(size_t)y * W + x - 1] + 1);
    }
  for (int y = H - 1; y >= 0; --y)
    for (int x = W - 1; x >= 0; --x) {
      int& v = d[(size_t)y * W + x];
      if (y + 1 < H) {
        if (x + 1 < W) v = std::min(v, d[(size_t)(y + 1) * W + x + 1] + 1);
        v = std::min(v, d[(size_t)(y + 1) * W + x] + 1);
        if (x > 0) v = std::min(v, d[(size_t)(y + 1) * W + x - 1] + 1);
      }
      if (x + 1 < W) v = std::min(v, d[(size_t)y * W + x + 1] + 1);
    }
  dst.resize(d.size());
  for (size_t i = 0; i < d.size(); ++i) dst[i] = (float)std::min(d[i], INF);
}

void selectScatteredFeatures(const std::vector<Candidate>& candidates, std::vector<Feature>& features, size_t num_features, float distance) {
  features.clear();
  float distance_sq = distance * distance;
  int i = 0;
  while (features.size() < num_features) {
    Candidate c = candidates[i];
    bool keep = true;
    for (int j = 0; (j < (int)features.size()) && keep; ++j) {
      Feature f = features[j];
      keep = (c.f.x - f.x) * (c.f.x - f.x) + (c.f.y - f.y) * (c.f.y - f.y) >= distance_sq;
    }
    if (keep) features.push_back(c.f);
    if (++i == (int)candidates.size()) {
      i = 0;
      distance -= 1.0f;
      distance_sq = distance * distance;
    }
  }
}

struct ColorGradientPyramid {
  std::vector<uchar> src, mask, angle;  // src: packed BGR
  std::vector<float> magnitude;
  int rows, cols, pyramid_level;
  float weak_threshold; size_t num_features; float strong_threshold;
  void update() {
    angle.resize((size_t)rows * cols); magnitude.resize((size_t)rows * cols);
    quantized_orientations(src.data(), rows, cols, (size_t)cols * 3, weak_threshold, angle.data(), magnitude.data(), NULL);
  }
  void pyrDown() {
    num_features /= 2;
    ++pyramid_level;
    std::vector<uchar> next_src((size_t)(rows / 2) * (cols / 2) * 3);
    pyrdown_u8(src.data(), rows, cols, 3, (size_t)cols * 3, next_src.data());
    if (!mask.empty()) {  // resize(mask, next_mask, size, 0, 0, INTER_NEAREST)
      std::vector<uchar> next_mask((size_t)(rows / 2) * (cols / 2));
      for (int y = 0; y < rows / 2; ++y)
        for (int x = 0; x < cols / 2; ++x) next_mask[(size_t)y * (cols / 2) + x] = mask[(size_t)(2 * y) * cols + 2 * x];
      mask.swap(next_mask);
    }
    src.swap(next_src);
    rows /= 2; cols /= 2;
    update();
  }
  bool extractTemplate(Template& templ) const {
    std::vector<uchar> local_mask;
    if (!mask.empty()) {
      erode_replicate(mask, rows, cols, 1, local_mask);
      for (size_t i = 0; i < mask.size(); ++i) local_mask[i] = (uchar)std::max(0, (int)mask[i] - (int)local_mask[i]);  // subtract(mask, local_mask, local_mask)
    }
    std::vector<Candidate> candidates;
    bool no_mask = local_mask.empty();
    float threshold_sq = strong_threshold * strong_threshold;
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        if (no_mask || local_mask[(size_t)r * cols + c]) {
          uchar quantized = angle[(size_t)r * cols + c];
          if (quantized > 0) {
            float score = magnitude[(size_t)r * cols + c];
            if (score > threshold_sq) candidates.push_back(Candidate(c, r, getLabel(quantized), score));
          }
        }
      }
    if (candidates.size() < num_features) return false;
    std::stable_sort(candidates.begin(), candidates.end());
    float distance = static_cast<float>(candidates.size() / num_features + 1);
    selectScatteredFeatures(candidates, templ.features, num_features, distance);
    templ.width = -1; templ.height = -1; templ.pyramid_level = pyramid_level;
    return true;
  }
};

struct DepthNormalPyramid {
  std::vector<uchar> mask, normal;
  int rows, cols, pyramid_level;
  size_t num_features; int extract_threshold;
  void pyrDown() {
    num_features /= 2;
    extract_threshold /= 2;
    ++pyramid_level;
    std::vector<uchar> next_normal((size_t)(rows / 2) * (cols / 2));
    for (int y = 0; y < rows / 2; ++y)
      for (int x = 0; x < cols / 2; ++x) next_normal[(size_t)y * (cols / 2) + x] = normal[(size_t)(2 * y) * cols + 2 * x];
    if (!mask.empty()) {
      std::vector<uchar> next_mask((size_t)(rows / 2) * (cols / 2));
      for (int y = 0; y < rows / 2; ++y)
        for (int x = 0; x < cols / 2; ++x) next_mask[(size_t)y * (cols / 2) + x] = mask[(size_t)(2 * y) * cols + 2 * x];
      mask.swap(next_mask);
    }
    normal.swap(next_normal);
    rows /= 2; cols /= 2;
  }
  bool extractTemplate(Template& templ) const {
    std::vector<uchar> local_mask;
    if (!mask.empty()) erode_replicate(mask, rows, cols, 2, local_mask);
    std::vector<float> distances[8];
    std::vector<uchar> temp((size_t)rows * cols);
    for (int i = 0; i < 8; ++i) {
      for (size_t k = 0; k < temp.size(); ++k) temp[k] = (local_mask.empty() || local_mask[k]) ? (uchar)(1 << i) : 0;  // temp.setTo(1 << i, local_mask)
      for (size_t k = 0; k < temp.size(); ++k) temp[k] &= normal[k];                                                  // bitwise_and(temp, normal, temp)
      distance_transform_c(temp, rows, cols, distances[i]);
    }
    int label_counts[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    std::vector<Candidate> candidates;
    bool no_mask = local_mask.empty();
    for (int r = 0; r < rows; ++r)
      for (int c = 0; c < cols; ++c) {
        if (no_mask || local_mask[(size_t)r * cols + c]) {
          uchar quantized = normal[(size_t)r * cols + c];
          if (quantized != 0 && quantized != 255) {
            int label = getLabel(quantized);
            float score = distances[label][(size_t)r * cols + c];
            if (score >= extract_threshold) {
              candidates.push_back(Candidate(c, r, label, score));
              ++label_counts[label];
            }
          }
        }
      }
    if (candidates.size() < num_features) return false;
    for (size_t i = 0; i < candidates.size(); ++i) {
      Candidate& c = candidates[i];
      c.score /= (float)label_counts[c.f.label];
    }
    std::stable_sort(candidates.begin(), candidates.end());
    float area = no_mask ? (float)normal.size() : (float)std::count_if(local_mask.begin(), local_mask.end(), [](uchar v) { return v != 0; });
    float distance = sqrtf(area) / sqrtf((float)num_features) + 1.5f;
    selectScatteredFeatures(candidates, templ.features, num_features, distance);
    templ.width = -1; templ.height = -1; templ.pyramid_level = pyramid_level;
    return true;
  }
};

void cropTemplates(std::vector<Template>& templates, int* bb) {
  int min_x = INT_MAX, min_y = INT_MAX, max_x = INT_MIN, max_y = INT_MIN;
  for (size_t i = 0; i < templates.size(); ++i) {
    Template& templ = templates[i];
    for (size_t j = 0; j < templ.features.size(); ++j) {
      int x = templ.features[j].x << templ.pyramid_level;
      int y = templ.features[j].y << templ.pyramid_level;
      min_x = std::min(min_x, x); min_y = std::min(min_y, y);
      max_x = std::max(max_x, x); max_y = std::max(max_y, y);
    }
  }
  if (min_x % 2 == 1) --min_x;
  if (min_y % 2 == 1) --min_y;
  for (size_t i = 0; i < templates.size(); ++i) {
    Template& templ = templates[i];
    templ.width = (max_x - min_x) >> templ.pyramid_level;
    templ.height = (max_y - min_y) >> templ.pyramid_level;
    int offset_x = min_x >> templ.pyramid_level;
    int offset_y = min_y >> templ.pyramid_level;
    for (size_t j = 0; j < templ.features.size(); ++j) {
      templ.features[j].x -= offset_x;
      templ.features[j].y -= offset_y;
    }
  }
  bb[0] = min_x; bb[1] = min_y; bb[2] = max_x - min_x; bb[3] = max_y - min_y;
}
}  // namespace

extern "C" {
// Detector::addTemplate(sources, class_id, object_mask, &bounding_box): returns template_id, or -1 if some level yields too few features
int lmo_detector_add_template(void* h, const void* const* src_data, const int* src_rows, const int* src_cols, const size_t* src_stride,
                              int n_sources, const char* class_id, const unsigned char* mask, size_t mask_stride, int* bounding_box) {
  Detector& det = *(Detector*)h;
  const int num_modalities = (int)det.modalities.size();
  const int pyramid_levels = (int)det.T_at_level.size();
  if (n_sources != num_modalities) return -2;
  const int rows = src_rows[0], cols = src_cols[0];
  std::vector<uchar> object_mask;
  if (mask) {
    object_mask.resize((size_t)rows * cols);
    for (int y = 0; y < rows; ++y) std::memcpy(&object_mask[(size_t)y * cols], mask + (size_t)y * mask_stride, (size_t)cols);
  }
  TemplatePyramid tp(num_modalities * pyramid_levels);
  for (int i = 0; i < num_modalities; ++i) {
    const Modality& mod = det.modalities[i];
    if (mod.type == MOD_COLOR_GRADIENT) {
      ColorGradientPyramid qp;
      qp.rows = rows; qp.cols = cols; qp.pyramid_level = 0; qp.mask = object_mask;
      qp.weak_threshold = mod.weak_threshold; qp.num_features = (size_t)mod.num_features; qp.strong_threshold = mod.strong_threshold;
      qp.src.resize((size_t)rows * cols * 3);
      for (int y = 0; y < rows; ++y) std::memcpy(&qp.src[(size_t)y * cols * 3], (const uchar*)src_data[i] + (size_t)y * src_stride[i], (size_t)cols * 3);
      qp.update();
      for (int l = 0; l < pyramid_levels; ++l) {
        if (l > 0) qp.pyrDown();
        if (!qp.extractTemplate(tp[l * num_modalities + i])) return -1;
      }
    } else {
      DepthNormalPyramid qp;
      qp.rows = rows; qp.cols = cols; qp.pyramid_level = 0; qp.mask = object_mask;
      qp.num_features = (size_t)mod.num_features; qp.extract_threshold = mod.extract_threshold;
      qp.normal.resize((size_t)rows * cols);
      quantized_normals((const ushort*)src_data[i], rows, cols, src_stride[i] / 2, mod.distance_threshold, mod.difference_threshold, qp.normal.data(), NULL, det.lut());
      for (int l = 0; l < pyramid_levels; ++l) {
        if (l > 0) qp.pyrDown();
        if (!qp.extractTemplate(tp[l * num_modalities + i])) return -1;
      }
    }
  }
  cropTemplates(tp, bounding_box);
  std::vector<TemplatePyramid>& template_pyramids = det.class_templates[class_id];
  int template_id = (int)template_pyramids.size();
  template_pyramids.push_back(tp);
  return template_id;
}

// read back template k (= l*M+m) of a pyramid: returns feature count; meta = {width, height, pyramid_level}
int lmo_detector_get_template(void* h, const char* class_id, int template_id, int k, int* meta, int32_t* feats /* [63][3] */) {
  Detector& det = *(Detector*)h;
  std::map<std::string, std::vector<TemplatePyramid> >::const_iterator it = det.class_templates.find(class_id);
  if (it == det.class_templates.end() || template_id < 0 || template_id >= (int)it->second.size()) return -1;
  const Template& t = it->second[template_id][k];
  meta[0] = t.width; meta[1] = t.height; meta[2] = t.pyramid_level;
  for (size_t i = 0; i < t.features.size(); ++i) { feats[3 * i] = t.features[i].x; feats[3 * i + 1] = t.features[i].y; feats[3 * i + 2] = t.features[i].label; }
  return (int)t.features.size();
}
}  // extern "C"

// libstdc++'s std::sort itself, on (similarity, template_id) records with Match::operator< and on scores with the reference's
// sortScoreCluster: the ground truth for the product's device-side restatement of the algorithm (csrc/lmx_sort_emul.hpp).
extern "C" {
void lmo_std_sort_perm(const float* sim, const int32_t* tid, long n, int32_t* perm) {
  struct R { float s; int32_t t; int32_t i; };
  std::vector<R> v((size_t)n);
  for (long i = 0; i < n; ++i) v[(size_t)i] = R{sim[i], tid[i], (int32_t)i};
  std::sort(v.begin(), v.end(), [](const R& a, const R& b) { return a.s != b.s ? a.s > b.s : a.t < b.t; });
  for (long i = 0; i < n; ++i) perm[i] = v[(size_t)i].i;
}
void lmo_std_sort_perm_score(const double* score, long n, int32_t* perm) {
  struct R { double s; int32_t i; };
  std::vector<R> v((size_t)n);
  for (long i = 0; i < n; ++i) v[(size_t)i] = R{score[i], (int32_t)i};
  std::sort(v.begin(), v.end(), [](const R& a, const R& b) { return a.s > b.s; });
  for (long i = 0; i < n; ++i) perm[i] = v[(size_t)i].i;
}
}  // extern "C"
