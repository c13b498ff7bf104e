// A caller written the way the reference's nodes and trainers are written (cv::linemod through cv::Ptr / cv::FileStorage /
// cv::Mat), compiled against include/lmx_cv_linemod.hpp to show that such code needs no edits: detector construction and
// persistence follow the shape of readLinemod / writeLinemod (reference src/renderer.cpp:42-70, src/rgbdDetector.cpp:1668-1680),
// matching follows rgbdDetector::linemod_detection (src/rgbdDetector.cpp:31-34) on a cropped ROI view of a wider frame
// (src/linemod_ensenso_detect_3_mult_detect_service.cpp:324-344), template access follows the drawing loop (..._service.cpp:350-361).
// The only line a maintainer adds is the facade include (here directly; in the reference at the end of rgbdDetector.h).
//
// usage: cv_facade_main match  <templates.yml> <W> <H> <frame_cols> <crop_x> <threshold> <bgr.raw> [depth.raw]
//        cv_facade_main train  <out.yml> <W> <H> <n_views> <views.raw: per view bgr, depth(u16), mask(u8)> <n_modalities>
//        cv_facade_main rewrite <in.yml> <out.yml>
//        cv_facade_main count  <templates.yml> <W> <H> <threshold> <bgr.raw> <depth.raw>     (number of matches + checksum)
//        cv_facade_main requests <templates.yml> <W> <H> <threshold> <n> <bgr.raw> [depth.raw]   (per-request times of readLinemod + match)
//        cv_facade_main publish <templates.yml>  (the node's own message type `linemod_pose_estimation::linemod`, whose generated header comes AFTER the facade)
//        cv_facade_main cow <templates.yml>      (Detector::load shares the cached bank; a modified detector gets a private copy)
//        cv_facade_main threads <templates.yml> <W> <H> <threshold> <bgrA.raw> <depthA.raw> <bgrB.raw> <depthB.raw>
#include <opencv2/opencv.hpp>          // the stand-in under tests/cpp/cv_standin (a real build has OpenCV here)
#include "lmx_cv_linemod.hpp"          // <- the one added include; from here on cv::linemod is the MI355X implementation
// The service node's include order (src/linemod_ensenso_detect_3_mult_detect_service.cpp:1, :15-16): rgbdDetector.h -- which now ends with the
// facade and its `#define linemod lmx_linemod` -- comes first, the package's generated service / message headers after it.  Stand-ins with
// gencpp's shape (tests/cpp/cv_standin/linemod_pose_estimation/): `typedef ... linemod;` in there is renamed by the macro, consistently for
// this translation unit; run_publish below uses the type the way a node would.
#include "linemod_pose_estimation/linemod_pose.h"
#include "linemod_pose_estimation/linemod.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <thread>

using namespace cv;
using namespace std;

static cv::Ptr<cv::linemod::Detector> readLinemod(const std::string& filename) {
  cv::Ptr<cv::linemod::Detector> detector(new cv::linemod::Detector);
  cv::FileStorage fs(filename, cv::FileStorage::READ);
  if (!fs.isOpened()) throw std::runtime_error("cannot open " + filename);
  detector->read(fs.root());
  cv::FileNode fn = fs["classes"];
  for (cv::FileNodeIterator i = fn.begin(), iend = fn.end(); i != iend; ++i) detector->readClass(*i);
  return detector;
}

static void writeLinemod(const cv::Ptr<cv::linemod::Detector>& detector, const std::string& filename) {
  cv::FileStorage fs(filename, cv::FileStorage::WRITE);
  detector->write(fs);
  std::vector<cv::String> ids = detector->classIds();
  fs << "classes" << "[";
  for (int i = 0; i < (int)ids.size(); ++i) {
    fs << "{";
    detector->writeClass(ids[i], fs);
    fs << "}";
  }
  fs << "]";
}

// same shape as rgbdDetector::linemod_detection
static void linemod_detection(Ptr<linemod::Detector> linemod_detector, const vector<Mat>& sources, const float& threshold,
                              std::vector<linemod::Match>& matches) {
  linemod_detector->match(sources, threshold, matches, std::vector<String>(), noArray());
}

static std::vector<char> slurp(const char* p) {
  std::ifstream f(p, std::ios::binary);
  return std::vector<char>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static int run_match(int argc, char** argv) {
  if (argc < 9) return 2;
  const int W = atoi(argv[3]), H = atoi(argv[4]), frame_cols = atoi(argv[5]), crop_x = atoi(argv[6]);
  const float threshold = (float)atof(argv[7]);
  std::vector<char> bgr = slurp(argv[8]), depth;
  // every request builds its detector again, like the service node's callback (..._service.cpp:1784-1786): the second one
  // finds the bank resident on the device
  for (int request = 0; request < 2; ++request) {
    Ptr<linemod::Detector> detector = readLinemod(argv[2]);
    if (detector->classIds().empty()) { printf("empty detector\n"); return 1; }
    Mat frame(H, frame_cols, CV_8UC3, bgr.data());
    std::vector<Mat> sources;
    sources.push_back(frame(Rect(crop_x, 0, W, H)));          // ROI view: row stride of the full frame
    if (argc > 9) {
      if (depth.empty()) depth = slurp(argv[9]);
      Mat dframe(H, frame_cols, CV_16UC1, depth.data());
      sources.push_back(dframe(Rect(crop_x, 0, W, H)));
    }
    std::vector<linemod::Match> matches;
    linemod_detection(detector, sources, threshold, matches);
    printf("request %d context_cached %d\n", request, (int)detector->contextWasCached());
    if (request == 0) continue;
    printf("classes %zu templates %d levels %d T0 %d modalities %zu\n", detector->classIds().size(), detector->numTemplates(), detector->pyramidLevels(),
           detector->getT(0), detector->getModalities().size());
    int num_modalities = (int)detector->getModalities().size();
    for (std::vector<linemod::Match>::iterator it = matches.begin(); it != matches.end(); it++) {
      std::vector<cv::linemod::Template> templates = detector->getTemplates(it->class_id, it->template_id);
      int n_features = 0;
      for (int m = 0; m < num_modalities; ++m)
        for (int i = 0; i < (int)templates[m].features.size(); ++i) {
          cv::linemod::Feature f = templates[m].features[i];
          n_features += (f.x >= 0 && f.y >= 0 && f.label < 8);
        }
      printf("%d %d %.9g %s %d %d\n", it->x, it->y, it->similarity, it->class_id.c_str(), it->template_id, n_features);
    }
    // the quantized images upstream can hand back
    std::vector<Mat> quantized;
    std::vector<linemod::Match> again;
    detector->match(sources, threshold, again, std::vector<String>(), quantized);
    unsigned long sum = 0;
    for (size_t q = 0; q < quantized.size(); ++q)
      for (int y = 0; y < quantized[q].rows; ++y)
        for (int x = 0; x < quantized[q].cols; ++x) sum += quantized[q].at<unsigned char>(y, x);
    printf("quantized %zu %dx%d %lu same %d\n", quantized.size(), quantized.empty() ? 0 : quantized[0].cols, quantized.empty() ? 0 : quantized[0].rows, sum,
           (int)(again.size() == matches.size()));
    // misuse: upstream CV_Assert(sources.size() == modalities.size())
    try {
      std::vector<Mat> none;
      detector->match(none, threshold, matches);
      printf("no exception\n");
    } catch (const std::exception& e) {
      printf("exception\n");
    }
  }
  return 0;
}

static int run_train(int argc, char** argv) {
  if (argc < 8) return 2;
  const int W = atoi(argv[3]), H = atoi(argv[4]), n_views = atoi(argv[5]), n_mod = atoi(argv[7]);
  std::vector<char> raw = slurp(argv[6]);
  std::vector<cv::Ptr<cv::linemod::Modality> > modalities;
  modalities.push_back(cv::Ptr<cv::linemod::ColorGradient>(new cv::linemod::ColorGradient));
  if (n_mod > 1) modalities.push_back(cv::Ptr<cv::linemod::DepthNormal>(new cv::linemod::DepthNormal));
  std::vector<int> ensenso_T;
  ensenso_T.push_back(5);
  ensenso_T.push_back(8);
  cv::Ptr<cv::linemod::Detector> detector_(new cv::linemod::Detector(modalities, ensenso_T));
  const size_t per_view = (size_t)W * H * 3 + (size_t)W * H * 2 + (size_t)W * H;
  for (int v = 0; v < n_views; ++v) {
    char* base = raw.data() + (size_t)v * per_view;
    cv::Mat image(H, W, CV_8UC3, base), depth(H, W, CV_16UC1, base + (size_t)W * H * 3), mask(H, W, CV_8UC1, base + (size_t)W * H * 5);
    std::vector<cv::Mat> sources;
    sources.push_back(image);
    if (n_mod > 1) sources.push_back(depth);
    cv::Rect bb;
    int template_in = detector_->addTemplate(sources, "obj", mask, &bb);
    printf("view %d template %d bb %d %d %d %d\n", v, template_in, bb.x, bb.y, bb.width, bb.height);
  }
  writeLinemod(detector_, argv[2]);
  printf("written %d templates\n", detector_->numTemplates());
  return 0;
}

// A threshold far below the reference's: more coarse candidates than the device lists hold by default; upstream has no such limit,
// the facade grows the lists and repeats the call.
static int run_count(int argc, char** argv) {
  if (argc < 8) return 2;
  const int W = atoi(argv[3]), H = atoi(argv[4]);
  const float threshold = (float)atof(argv[5]);
  std::vector<char> bgr = slurp(argv[6]), depth = slurp(argv[7]);
  Ptr<linemod::Detector> detector = readLinemod(argv[2]);
  std::vector<Mat> sources;
  sources.push_back(Mat(H, W, CV_8UC3, bgr.data()));
  sources.push_back(Mat(H, W, CV_16UC1, depth.data()));
  for (int call = 0; call < 2; ++call) {
    std::vector<linemod::Match> matches;
    linemod_detection(detector, sources, threshold, matches);
    unsigned long long sum = 0;
    for (size_t i = 0; i < matches.size(); ++i) sum = sum * 1000003ull + (unsigned long long)(matches[i].x * 7 + matches[i].y * 13 + matches[i].template_id * 31 + (int)(matches[i].similarity * 1000.f));
    printf("matches %zu checksum %llu\n", matches.size(), sum);
  }
  return 0;
}

// Two threads, each with its OWN detector read from the same yml (as two callbacks of a node would have): the facade hands both
// the same cached device context, and the library serialises their calls.  Every result must equal the one-thread result.
static int run_threads(int argc, char** argv) {
  if (argc < 10) return 2;
  const int W = atoi(argv[3]), H = atoi(argv[4]);
  const float threshold = (float)atof(argv[5]);
  std::vector<char> raw[4] = {slurp(argv[6]), slurp(argv[7]), slurp(argv[8]), slurp(argv[9])};
  std::vector<std::vector<linemod::Match> > expect(2);
  {
    Ptr<linemod::Detector> detector = readLinemod(argv[2]);
    for (int t = 0; t < 2; ++t) {
      std::vector<Mat> sources;
      sources.push_back(Mat(H, W, CV_8UC3, raw[2 * t].data()));
      sources.push_back(Mat(H, W, CV_16UC1, raw[2 * t + 1].data()));
      linemod_detection(detector, sources, threshold, expect[t]);
    }
  }
  int bad[2] = {0, 0}, cached[2] = {0, 0};
  std::vector<std::thread> workers;
  for (int t = 0; t < 2; ++t)
    workers.push_back(std::thread([&, t]() {
      Ptr<linemod::Detector> detector = readLinemod(argv[2]);
      std::vector<Mat> sources;
      sources.push_back(Mat(H, W, CV_8UC3, raw[2 * t].data()));
      sources.push_back(Mat(H, W, CV_16UC1, raw[2 * t + 1].data()));
      for (int it = 0; it < 40; ++it) {
        std::vector<linemod::Match> matches;
        linemod_detection(detector, sources, threshold, matches);
        bool same = matches.size() == expect[t].size();
        for (size_t i = 0; same && i < matches.size(); ++i)
          same = matches[i] == expect[t][i] && matches[i].template_id == expect[t][i].template_id;
        bad[t] += same ? 0 : 1;
      }
      cached[t] = (int)detector->contextWasCached();
    }));
  for (size_t i = 0; i < workers.size(); ++i) workers[i].join();
  printf("threads expect %zu %zu mismatches %d %d shared_context %d\n", expect[0].size(), expect[1].size(), bad[0], bad[1], cached[0] + cached[1]);
  return bad[0] + bad[1] ? 1 : 0;
}

// The service node's callback, timed: every request reads the templates file into a NEW detector (readLinemod) and matches one frame
// (src/linemod_ensenso_detect_3_mult_detect_service.cpp:1784-1786, 324-344).  Prints the median milliseconds of the two halves.
static int run_requests(int argc, char** argv) {
  if (argc < 8) return 2;
  const int W = atoi(argv[3]), H = atoi(argv[4]);
  const float threshold = (float)atof(argv[5]);
  const int n = atoi(argv[6]);
  std::vector<char> bgr = slurp(argv[7]), depth;
  if (argc > 8) depth = slurp(argv[8]);
  std::vector<double> t_read, t_match;
  size_t n_matches = 0;
  const bool cached = std::string(argv[1]) == "requests_cached";   // the one-line variant of readLinemod: cv::linemod::Detector::load
  for (int request = 0; request < n + 3; ++request) {
    const auto t0 = std::chrono::steady_clock::now();
    Ptr<linemod::Detector> detector = cached ? linemod::Detector::load(argv[2]) : readLinemod(argv[2]);
    const auto t1 = std::chrono::steady_clock::now();
    std::vector<Mat> sources;
    sources.push_back(Mat(H, W, CV_8UC3, bgr.data()));
    if (!depth.empty()) sources.push_back(Mat(H, W, CV_16UC1, depth.data()));
    std::vector<linemod::Match> matches;
    linemod_detection(detector, sources, threshold, matches);
    const auto t2 = std::chrono::steady_clock::now();
    n_matches = matches.size();
    if (request >= 3) {   // the first requests parse the file and build the device context
      t_read.push_back(std::chrono::duration<double, std::milli>(t1 - t0).count());
      t_match.push_back(std::chrono::duration<double, std::milli>(t2 - t1).count());
    }
  }
  std::sort(t_read.begin(), t_read.end());
  std::sort(t_match.begin(), t_match.end());
  printf("requests %d matches %zu readLinemod_ms %.3f match_ms %.3f\n", n, n_matches, t_read[t_read.size() / 2], t_match[t_match.size() / 2]);
  // both ways of building the detector give the same detector: classes, templates, matches (and a loaded one can still be modified)
  Ptr<linemod::Detector> a = readLinemod(argv[2]), b = linemod::Detector::load(argv[2]);
  std::vector<Mat> sources;
  sources.push_back(Mat(H, W, CV_8UC3, bgr.data()));
  if (!depth.empty()) sources.push_back(Mat(H, W, CV_16UC1, depth.data()));
  std::vector<linemod::Match> ma, mb;
  linemod_detection(a, sources, threshold, ma);
  linemod_detection(b, sources, threshold, mb);
  bool same = ma.size() == mb.size() && a->classIds() == b->classIds() && a->numTemplates() == b->numTemplates() && a->pyramidLevels() == b->pyramidLevels() &&
              a->getT(0) == b->getT(0) && a->getModalities().size() == b->getModalities().size();
  for (size_t i = 0; same && i < ma.size(); ++i)
    same = ma[i].x == mb[i].x && ma[i].y == mb[i].y && ma[i].similarity == mb[i].similarity && ma[i].class_id == mb[i].class_id && ma[i].template_id == mb[i].template_id;
  if (same && b->numTemplates() > 0) {
    const String cid = b->classIds()[0];
    same = a->getTemplates(cid, 0).size() == b->getTemplates(cid, 0).size() && a->getTemplates(cid, 0)[0].features.size() == b->getTemplates(cid, 0)[0].features.size();
  }
  printf("load_equals_readLinemod %d context_cached %d\n", (int)same, (int)b->contextWasCached());
  return same ? 0 : 1;
}

// A detector from Detector::load shares the file's cached bank; modifying it (here: readClass of the first class under another id) must
// give it a private copy and leave another detector loaded from the same file untouched.  No GPU involved.
static int run_cow(int argc, char** argv) {
  if (argc < 3) return 2;
  Ptr<linemod::Detector> a = linemod::Detector::load(argv[2]), b = linemod::Detector::load(argv[2]);
  const int before = b->numTemplates();
  cv::FileStorage fs(argv[2], cv::FileStorage::READ);
  cv::FileNode fn = fs["classes"];
  b->readClass(*fn.begin(), "copy_of_first");
  printf("a %d classes %zu | b %d classes %zu | before %d levels %d T0 %d modalities %zu\n", a->numTemplates(), a->classIds().size(), b->numTemplates(),
         b->classIds().size(), before, a->pyramidLevels(), a->getT(0), a->getModalities().size());
  return (a->numTemplates() == before && b->numTemplates() > before && b->classIds().size() == a->classIds().size() + 1) ? 0 : 1;
}

// The node's message type next to cv::linemod in one translation unit: `linemod_pose_estimation::linemod` (a message) and `cv::linemod::Match`
// (the facade's namespace) are both spelled with the token the macro renames.  Fills one message per class from a detector and "publishes" it.
static bool linemod_pose_callback(linemod_pose_estimation::linemod_pose::Request& req, linemod_pose_estimation::linemod_pose::Response& res) {
  res.pose.translation.x = 0.001 * req.object_id;
  return true;
}
static int run_publish(int argc, char** argv) {
  if (argc < 3) return 2;
  Ptr<linemod::Detector> det = linemod::Detector::load(argv[2]);
  std::vector<linemod_pose_estimation::linemod> outbox;
  const std::vector<cv::String> ids = det->classIds();
  for (size_t i = 0; i < ids.size(); ++i) {
    linemod_pose_estimation::linemod msg;
    msg.id = (int32_t)i;
    msg.tranform.translation.x = det->numTemplates(ids[i]);
    msg.tranform.rotation.w = 1.0;
    outbox.push_back(msg);
  }
  linemod_pose_estimation::linemodPtr last(new linemod_pose_estimation::linemod(outbox.back()));
  linemod_pose_estimation::linemodConstPtr view = last;
  linemod_pose_estimation::linemod_pose srv;
  srv.request.object_id = view->id;
  if (!linemod_pose_callback(srv.request, srv.response)) return 1;
  cv::linemod::Match m;     // the facade's type, same token
  m.template_id = view->id;
  printf("%s %zu messages, last id %d templates %.0f, match template_id %d, pose.x %.3f\n", ros::message_traits::DataType<linemod_pose_estimation::linemod>::value(),
         outbox.size(), view->id, view->tranform.translation.x, m.template_id, srv.response.pose.translation.x);
  return 0;
}

int main(int argc, char** argv) {
  if (argc < 2) { fprintf(stderr, "usage\n"); return 2; }
  try {
    const std::string mode = argv[1];
    if (mode == "match") return run_match(argc, argv);
    if (mode == "train") return run_train(argc, argv);
    if (mode == "threads") return run_threads(argc, argv);
    if (mode == "count") return run_count(argc, argv);
    if (mode == "requests" || mode == "requests_cached") return run_requests(argc, argv);
    if (mode == "cow") return run_cow(argc, argv);
    if (mode == "publish") return run_publish(argc, argv);
    if (mode == "rewrite" && argc >= 4) {
      writeLinemod(readLinemod(argv[2]), argv[3]);   // FileNode -> Detector -> FileStorage, no device needed
      return 0;
    }
    fprintf(stderr, "usage\n");
    return 2;
  } catch (const std::exception& e) {
    fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
}
