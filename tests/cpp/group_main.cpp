// Multi-GPU matching from C++ only (include/lmx.h: lmx_group_*): what rgbdDetector::linemod_detection would call on a node with
// several MI355X -- load the bank, create a device group (single process), match batches of frames.
// usage: group_main <templates.yml> <n_members> <gather_capacity> <W> <H> <threshold> <n_frames> <frames.raw: per frame bgr then depth(u16)>
//                   [rccl|peer] [distinct|same] [batch|pipeline] [max_candidates] [frame_groups]
//   rccl / peer     : the collective (peer = device-to-device block copies; needed when members share a device)
//   distinct / same : members on devices 0..n-1, or all of them on device 0
//   batch           : lmx_group_match_batch twice (the second call runs with the capacity the first one settled on)
//   frame_groups    : G of the G x R member grid (default 1: template sharding only); member k takes frames of group k / R
//   pipeline        : upload / submit / finish with as many batches in flight as the group allows; batch b matches the frames rotated
//                     by b, n_frames - (b % n_frames) of them, at threshold + 8 * (b % 2)
// prints "batch b frame f: x y similarity class_index template_id" per match and a summary line
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <iterator>
#include <vector>

#include "lmx.h"

int main(int argc, char** argv) {
  if (argc < 9) { std::fprintf(stderr, "usage\n"); return 2; }
  lmx_bank* bank = nullptr;
  if (lmx_bank_load_yaml(argv[1], &bank) != LMX_OK) { std::fprintf(stderr, "error: %s\n", lmx_last_error()); return 1; }
  const int n_dev = std::atoi(argv[2]), W = std::atoi(argv[4]), H = std::atoi(argv[5]), n_frames = std::atoi(argv[7]);
  const float threshold = (float)std::atof(argv[6]);
  const bool peer = argc > 9 && std::strcmp(argv[9], "peer") == 0;
  const bool same = argc > 10 && std::strcmp(argv[10], "same") == 0;
  const bool pipeline = argc > 11 && std::strcmp(argv[11], "pipeline") == 0;
  const int max_candidates = argc > 12 ? std::atoi(argv[12]) : 0;
  const int frame_groups = argc > 13 ? std::atoi(argv[13]) : 0;
  const int M = lmx_bank_num_modalities(bank);
  std::ifstream f(argv[8], std::ios::binary);
  std::vector<char> raw((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
  const size_t per_frame = (size_t)W * H * 3 + (M > 1 ? (size_t)W * H * 2 : 0);
  if (raw.size() < per_frame * n_frames) { std::fprintf(stderr, "error: frames file too short\n"); return 1; }
  auto sources_rotated = [&](int rot, int n) {
    std::vector<lmx_image> s;
    for (int i = 0; i < n; ++i) {
      const char* base = raw.data() + (size_t)((i + rot) % n_frames) * per_frame;
      s.push_back(lmx_image{base, H, W, 3, 1, (size_t)W * 3});
      if (M > 1) s.push_back(lmx_image{base + (size_t)W * H * 3, H, W, 1, 2, (size_t)W * 2});
    }
    return s;
  };
  std::vector<int32_t> devices((size_t)n_dev, 0);
  for (int i = 0; i < n_dev; ++i) devices[(size_t)i] = same ? 0 : i;
  lmx_group_desc gd = {};
  gd.n_devices = n_dev; gd.devices = devices.data(); gd.width = W; gd.height = H; gd.max_batch = n_frames; gd.gather_capacity = std::atoi(argv[3]);
  gd.max_candidates = max_candidates;
  gd.frame_groups = frame_groups;
  gd.collective = peer ? LMX_GROUP_COLLECTIVE_PEER_COPY : LMX_GROUP_COLLECTIVE_RCCL;
  if (pipeline) gd.flags = LMX_CTX_OVERLAP;
  lmx_group* group = nullptr;
  if (lmx_group_create(bank, &gd, &group) != LMX_OK) { std::fprintf(stderr, "error: %s\n", lmx_last_error()); return 1; }
  const size_t cap = 4096;
  std::vector<lmx_match_t> out(cap * n_frames);
  std::vector<size_t> n_out(n_frames);
  auto print = [&](int b, int n) {
    for (int i = 0; i < n; ++i)
      for (size_t k = 0; k < n_out[i]; ++k) {
        const lmx_match_t& m = out[(size_t)i * cap + k];
        std::printf("batch %d frame %d: %d %d %.9g %d %d\n", b, i, m.x, m.y, m.similarity, m.class_index, m.template_id);
      }
  };
  if (!pipeline) {
    std::vector<lmx_image> sources = sources_rotated(0, n_frames);
    for (int rep = 0; rep < 2; ++rep) {
      if (lmx_group_match_batch(group, n_frames, sources.data(), M, threshold, nullptr, 0, out.data(), cap, n_out.data()) != LMX_OK) {
        std::fprintf(stderr, "error: %s\n", lmx_last_error());
        return 1;
      }
      print(rep, n_frames);
    }
  } else {
    const int depth = lmx_group_depth(group), n_batches = 3 * depth + 1;
    std::vector<int> sizes;
    int done = 0;
    for (int b = 0; b < n_batches; ++b) {
      if ((int)sizes.size() - done == depth) {
        if (lmx_group_finish(group, sizes[(size_t)done], out.data(), cap, n_out.data()) != LMX_OK) { std::fprintf(stderr, "error: %s\n", lmx_last_error()); return 1; }
        print(done, sizes[(size_t)done]);
        ++done;
      }
      const int n = n_frames - (b % n_frames);
      std::vector<lmx_image> sources = sources_rotated(b, n);   // the descriptors may go away after upload: the frames were copied
      if (lmx_group_upload(group, n, sources.data(), M) != LMX_OK || lmx_group_submit(group, n, threshold + 8.f * (b % 2), nullptr, 0) != LMX_OK) {
        std::fprintf(stderr, "error: %s\n", lmx_last_error());
        return 1;
      }
      sizes.push_back(n);
    }
    while (done < (int)sizes.size()) {
      if (lmx_group_finish(group, sizes[(size_t)done], out.data(), cap, n_out.data()) != LMX_OK) { std::fprintf(stderr, "error: %s\n", lmx_last_error()); return 1; }
      print(done, sizes[(size_t)done]);
      ++done;
    }
  }
  std::printf("group of %d, collective %s, depth %d, frame groups %d, gather capacity %d\n", lmx_group_size(group), lmx_group_collective_name(group),
              lmx_group_depth(group), lmx_group_frame_groups(group), lmx_group_gather_capacity(group));
  lmx_group_destroy(group);
  lmx_bank_destroy(bank);
  return 0;
}
