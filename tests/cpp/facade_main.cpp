// Exercises the C++ facade (include/lmx_linemod.hpp) exactly the way the reference's node code would after the
// INTEGRATION.md edit: readLinemod(yml) once, then linemod_detection-style match() on a strided BGR (+ depth) view.
// usage: facade_main <templates.yml> <W> <H> <row_stride_pixels> <threshold> <bgr.raw> [depth.raw]
// prints one line per match: x y similarity class_id template_id
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>
#include <vector>

#include "lmx_linemod.hpp"

static std::vector<char> slurp(const char* p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv) {
  if (argc < 7) { std::fprintf(stderr, "usage\n"); return 2; }
  try {
    std::shared_ptr<lmx::linemod::Detector> detector = lmx::linemod::readLinemod(argv[1]);
    const int W = std::atoi(argv[2]), H = std::atoi(argv[3]), stride_px = std::atoi(argv[4]);
    const float threshold = (float)std::atof(argv[5]);
    std::vector<char> bgr = slurp(argv[6]), depth;
    std::vector<lmx::linemod::Image> sources;
    sources.push_back(lmx::linemod::Image{bgr.data(), H, W, 3, 1, (size_t)stride_px * 3});
    if (argc > 7) {
      depth = slurp(argv[7]);
      sources.push_back(lmx::linemod::Image{depth.data(), H, W, 1, 2, (size_t)stride_px * 2});
    }
    std::vector<lmx::linemod::Match> matches;
    // == rgbdDetector::linemod_detection: match(sources, threshold, matches, std::vector<String>(), noArray())
    detector->match(sources, threshold, matches, std::vector<std::string>());
    std::printf("classes %zu templates %d levels %d\n", detector->classIds().size(), detector->numTemplates(), detector->pyramidLevels());
    for (const lmx::linemod::Match& m : matches)
      std::printf("%d %d %.9g %s %d\n", m.x, m.y, m.similarity, m.class_id.c_str(), m.template_id);
    // misuse -> exception, like CV_Assert(sources.size() == modalities.size())
    try {
      std::vector<lmx::linemod::Image> none;
      detector->match(none, threshold, matches);
      std::printf("no exception\n");
    } catch (const lmx::linemod::Exception& e) {
      std::printf("exception status %d\n", (int)e.status);
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
