// liblmx.so, SURVEY 8f row 2 on the host: rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU as the
// reference's nodes chain them (/root/reference/src/rgbdDetector.cpp:36-144, 462-574; called ..._service.cpp:376-447).  The device form
// of the same chain is csrc/lmx_f2.hip.

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <mutex>
#include <shared_mutex>
#include <thread>
#include <cctype>
#include <cmath>
#include <cstdlib>
#include <cstdarg>
#include <cstdio>
#include <cstring>
#include <functional>
#include <map>
#include <memory>
#include <string>
#include <vector>

#include <sys/stat.h>

#include "lmx_ctx.hpp"

using namespace lmx;

namespace lmx {

// The depth ring of a vote is `(int)((dist - renderer_radius_min) / renderer_radius_step)` in float, as the reference computes it
// (src/rgbdDetector.cpp:48-56).  A step that is not positive and finite, a distance that is not finite or a quotient an int cannot hold
// make that conversion undefined in the reference and different between x86 and the GPU here: refused up front for the whole side-car.
lmx_status check_vote_rings(const double* dists, size_t n, const lmx_cluster_params* pp) {
  const float step = (float)pp->renderer_radius_step;
  if (!(step > 0.0f) || !std::isfinite(step) || !std::isfinite((float)pp->renderer_radius_min)) {
    set_error("renderer_radius_step must be positive and finite, renderer_radius_min finite");
    return LMX_ERR_INVALID_ARG;
  }
  for (size_t i = 0; i < n; ++i) {
    const float q = ((float)dists[i] - pp->renderer_radius_min) / step;
    if (!(q > -1.0e9f && q < 1.0e9f)) { set_error("template %zu: origin distance %g gives no usable depth ring", i, dists[i]); return LMX_ERR_INVALID_ARG; }
  }
  return LMX_OK;
}

}  // namespace lmx

// ---- rcd_voting -> cluster_filter -> cluster_scoring -> nonMaximaSuppressionUsingIOU (host) ----------------
namespace {
struct HostCluster {
  std::vector<int> index;
  double score = 0;
  bool suppressed = false;
  int rect[4] = {0, 0, 0, 0};
  std::vector<int32_t> members;  // indices into the caller's match array, in the order they were voted in
};
bool by_score_desc(const HostCluster& a, const HostCluster& b) { return a.score > b.score; }  // the comparator of rgbdDetector.h:127-130

// Overlap of two boxes {x, y, w, h} the way the reference's NMS measures it (rgbdDetector.cpp:532-574): inclusive pixel extents,
// the intersection area as an int product converted to float, the union in float, float division.  Same arithmetic as the
// device version in lmx_f2.hip (box_iou): the int/float mix is part of the observable result.
// The reference does this in plain `int`; with the rects its size_t division produces for clusters left of / above the origin
// (coordinates near 2^32 / n) those sums and products overflow, which on its platform wraps.  Here the wrap is spelled out
// (unsigned arithmetic, then back to int): the same values without undefined behaviour (found by UBSan on the host build).
inline int wadd(int a, int b) { return (int)((unsigned)a + (unsigned)b); }
inline int wsub(int a, int b) { return (int)((unsigned)a - (unsigned)b); }
inline int wmul(int a, int b) { return (int)((unsigned)a * (unsigned)b); }
float box_overlap_ratio(const int* p, const int* q) {
  const int p_x0 = p[0], p_x1 = wsub(wadd(p[0], p[2]), 1), p_y0 = p[1], p_y1 = wsub(wadd(p[1], p[3]), 1);
  const int q_x0 = q[0], q_x1 = wsub(wadd(q[0], q[2]), 1), q_y0 = q[1], q_y1 = wsub(wadd(q[1], q[3]), 1);
  const int lo_x = std::max(p_x0, q_x0), hi_x = std::min(p_x1, q_x1), lo_y = std::max(p_y0, q_y0), hi_y = std::min(p_y1, q_y1);
  const bool overlap_x = (lo_x >= p_x0 && lo_x <= p_x1) || (lo_x >= q_x0 && lo_x <= q_x1);
  const bool overlap_y = (lo_y >= p_y0 && lo_y <= p_y1) || (lo_y >= q_y0 && lo_y <= q_y1);
  const float shared = (overlap_x && overlap_y) ? (float)wmul(wadd(wsub(hi_x, lo_x), 1), wadd(wsub(hi_y, lo_y), 1)) : 0.0f;
  const float total = (float)wadd(wmul(p[2], p[3]), wmul(q[2], q[3])) - shared;
  return shared / total;
}
}  // namespace

extern "C" lmx_status lmx_cluster_matches(const lmx_match_t* matches, size_t n_matches, const double* obj_origin_dists, const int32_t* rects,
                                          size_t n_templates, const lmx_cluster_params* pp, lmx_cluster_t* clusters, size_t cap_clusters,
                                          size_t* n_clusters, int32_t* members, size_t cap_members) {
  return lmx::guarded("lmx_cluster_matches", [&]() -> lmx_status {
  if ((n_matches && !matches) || !obj_origin_dists || !rects || !pp || !n_clusters || (cap_clusters && !clusters) || (cap_members && !members)) {
    lmx::set_error("lmx_cluster_matches: null argument");
    return LMX_ERR_INVALID_ARG;
  }
  if (pp->vote_row_col_step <= 0) { lmx::set_error("vote_row_col_step must be positive"); return LMX_ERR_INVALID_ARG; }
  if (pp->cluster_size_thresh < 0) { lmx::set_error("cluster_size_thresh must not be negative"); return LMX_ERR_INVALID_ARG; }   // see lmx_ctx_set_cluster_sidecar
  if (lmx_status vs = check_vote_rings(obj_origin_dists, n_templates, pp)) return vs;
  // rcd_voting: bins keyed by {y/step, x/step, depth ring}; std::map keeps them in lexicographic order like upstream
  std::map<std::vector<int>, std::vector<int32_t>> map_match;
  const float voting_depth_step = (float)pp->renderer_radius_step;
  for (size_t i = 0; i < n_matches; ++i) {
    const lmx_match_t& m = matches[i];
    if (m.template_id < 0 || (size_t)m.template_id >= n_templates) { lmx::set_error("match %zu: template_id %d outside the side-car arrays", i, m.template_id); return LMX_ERR_INVALID_ARG; }
    const float depth = (float)obj_origin_dists[m.template_id];
    std::vector<int> index(3);
    index[0] = m.y / pp->vote_row_col_step;
    index[1] = m.x / pp->vote_row_col_step;
    index[2] = (int)((depth - pp->renderer_radius_min) / voting_depth_step);
    map_match[index].push_back((int32_t)i);
  }
  // cluster_filter(map, thresh) -- intended semantics (see header) -- and cluster_scoring (similarity_score_calc)
  std::vector<HostCluster> cd;
  for (auto it = map_match.begin(); it != map_match.end(); ++it) {
    if ((long)it->second.size() <= (long)pp->cluster_size_thresh) continue;
    HostCluster c;
    c.index = it->first;
    double sum_score = 0.0;
    int num = 0;
    for (int32_t mi : it->second) { sum_score += matches[mi].similarity; num++; }
    c.score = sum_score / num;
    c.members = it->second;
    cd.push_back(c);
  }
  if (!cd.empty()) {
    // nonMaximaSuppressionUsingIOU: mean rect, sort by score (std::sort, like upstream), greedy suppression at IoU > 0.4
    for (HostCluster& c : cd) {
      int sum_x = 0, sum_y = 0, sum_w = 0, sum_h = 0;   // integer sums, like the reference
      for (int32_t mi : c.members) {
        const int32_t* r = rects + (size_t)matches[mi].template_id * 4;
        sum_x = wadd(sum_x, matches[mi].x); sum_y = wadd(sum_y, matches[mi].y); sum_w = wadd(sum_w, r[2]); sum_h = wadd(sum_h, r[3]);   // int, wrapping
      }
      // `X /= it1->matches.size();` in the reference divides by a size_t: the int sum is converted to size_t first, so a negative
      // sum (matches left of / above the origin) divides as 2^64 + X; the quotient goes back to int
      const size_t n = c.members.size();
      auto div_by_size = [n](int v) { return (int)(unsigned)((unsigned long long)(long long)v / (unsigned long long)n); };
      c.rect[0] = div_by_size(sum_x); c.rect[1] = div_by_size(sum_y); c.rect[2] = div_by_size(sum_w); c.rect[3] = div_by_size(sum_h);
    }
    std::sort(cd.begin(), cd.end(), by_score_desc);
    for (size_t a = 0; a < cd.size(); ++a) {
      if (cd[a].suppressed) continue;
      for (size_t b = a + 1; b < cd.size(); ++b)
        if (!cd[b].suppressed) {
          const double ratio = box_overlap_ratio(cd[a].rect, cd[b].rect);
          if (ratio > 0.4) cd[b].suppressed = true;
        }
    }
  }
  size_t nc = 0, nm = 0;
  lmx_status st = LMX_OK;
  for (const HostCluster& c : cd) {
    if (c.suppressed) continue;
    if (nc < cap_clusters && nm + c.members.size() <= cap_members) {
      lmx_cluster_t& o = clusters[nc];
      o.index[0] = c.index[0]; o.index[1] = c.index[1]; o.index[2] = c.index[2];
      for (int k = 0; k < 4; ++k) o.rect[k] = c.rect[k];
      o.score = c.score;
      o.member_begin = (int32_t)nm; o.member_count = (int32_t)c.members.size();
      std::memcpy(members + nm, c.members.data(), c.members.size() * sizeof(int32_t));
    } else {
      st = LMX_ERR_OVERFLOW;
    }
    nc += 1; nm += c.members.size();
  }
  *n_clusters = nc;
  if (st != LMX_OK) lmx::set_error("%zu clusters / %zu members exceed the output capacity", nc, nm);
  return st;
  });
}
