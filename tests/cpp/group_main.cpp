// Multi-GPU matching from C++ only (include/lmx.h: lmx_group_*): what rgbdDetector::linemod_detection would call on a node with
// several MI355X -- load the bank, create a device group (single process, ncclCommInitAll), match a batch of frames.
// usage: group_main <templates.yml> <n_devices> <gather_capacity> <W> <H> <threshold> <n_frames> <frames.raw: per frame bgr then depth(u16)>
// prints "frame f: x y similarity class_index template_id" per match and the final gather capacity
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "lmx.h"

int main(int argc, char** argv) {
  if (argc < 9) { std::fprintf(stderr, "usage\n"); return 2; }
  lmx_bank* bank = nullptr;
  if (lmx_bank_load_yaml(argv[1], &bank) != LMX_OK) { std::fprintf(stderr, "error: %s\n", lmx_last_error()); return 1; }
  const int n_dev = std::atoi(argv[2]), W = std::atoi(argv[4]), H = std::atoi(argv[5]), n_frames = std::atoi(argv[7]);
  const float threshold = (float)std::atof(argv[6]);
  const int M = lmx_bank_num_modalities(bank);
  std::ifstream f(argv[8], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const size_t per_frame = (size_t)W * H * 3 + (M > 1 ? (size_t)W * H * 2 : 0);
  if (raw.size() < per_frame * n_frames) { std::fprintf(stderr, "error: frames file too short\n"); return 1; }
  std::vector<lmx_image> sources;
  for (int i = 0; i < n_frames; ++i) {
    const char* base = raw.data() + (size_t)i * per_frame;
    sources.push_back(lmx_image{base, H, W, 3, 1, (size_t)W * 3});
    if (M > 1) sources.push_back(lmx_image{base + (size_t)W * H * 3, H, W, 1, 2, (size_t)W * 2});
  }
  lmx_group_desc gd = {};
  gd.n_devices = n_dev; gd.width = W; gd.height = H; gd.max_batch = n_frames; gd.gather_capacity = std::atoi(argv[3]);
  lmx_group* group = nullptr;
  if (lmx_group_create(bank, &gd, &group) != LMX_OK) { std::fprintf(stderr, "error: %s\n", lmx_last_error()); return 1; }
  const size_t cap = 4096;
  std::vector<lmx_match_t> out(cap * n_frames);
  std::vector<size_t> n_out(n_frames);
  for (int rep = 0; rep < 2; ++rep)   // the second batch runs with the capacity the first one settled on
    if (lmx_group_match_batch(group, n_frames, sources.data(), M, threshold, nullptr, 0, out.data(), cap, n_out.data()) != LMX_OK) {
      std::fprintf(stderr, "error: %s\n", lmx_last_error());
      return 1;
    }
  std::printf("group of %d, gather capacity %d\n", lmx_group_size(group), lmx_group_gather_capacity(group));
  for (int i = 0; i < n_frames; ++i)
    for (size_t k = 0; k < n_out[i]; ++k) {
      const lmx_match_t& m = out[(size_t)i * cap + k];
      std::printf("frame %d: %d %d %.9g %d %d\n", i, m.x, m.y, m.similarity, m.class_index, m.template_id);
    }
  lmx_group_destroy(group);
  lmx_bank_destroy(bank);
  return 0;
}
