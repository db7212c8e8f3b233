// odometer_pipeline_demo.cpp -- IcpOdometer::laserCloudCallback as the reference writes it
// (/root/reference/src/icpslam/icp_odometer.cpp:96-101 voxelFilterCloud, :147-220 the callback), with the two type names
// INTEGRATION.md swaps and nothing else changed: a FRESH icpgpu::VoxelGrid and a FRESH
// icpgpu::GeneralizedIterativeClosestPoint per scan, host clouds in, `*prev_cloud_ = *curr_cloud_` on success.  The callbacks
// run one at a time on <threads> worker threads in rotation, like a subscriber of queue size 1 under ros::AsyncSpinner(4)
// (icpslam_node.cpp:9): consecutive scans are handled by DIFFERENT threads.
// This is the boundary exactly as integrated; bench.py times it (gicp.shim_pipeline_scans_per_sec) beside the resident
// pipeline, tests/test_cpp_shim.py compares every scan's transform with the C-ABI pipeline's bit for bit.
//
// usage: odometer_pipeline_demo <scanA.bin> <nA> <scanB.bin> <nB> <n_scans> <leaf> <max_iters> <threads> [warmup]
//   scan k is A for even k, B for odd k (raw, unfiltered, 16-byte points)
// stdout: one line per registered scan:  k converged iterations fitness n_filtered T[16]
//         last line:  TIMING scans <n> seconds <s> scans_per_sec <r> warmup <w> threads <t>
#include <chrono>
#include <condition_variable>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "icpgpu_registration.hpp"

namespace mock_pcl {  // memory layout and metadata members of pcl::PointCloud<pcl::PointXYZ> (PCL is not in this image)
struct alignas(16) PointXYZ {
  float x, y, z, pad;
};
struct PointCloud {
  std::vector<PointXYZ> points;
  unsigned width = 0, height = 1;
  bool is_dense = true;
  std::size_t size() const { return points.size(); }
  using Ptr = std::shared_ptr<PointCloud>;
};
}  // namespace mock_pcl
using Cloud = mock_pcl::PointCloud;

static Cloud::Ptr load(const char* path, std::size_t n) {
  auto c = std::make_shared<Cloud>();
  c->points.resize(n);
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  if (n && std::fread(c->points.data(), sizeof(mock_pcl::PointXYZ), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
  std::fclose(f);
  return c;
}

// ---- the reference's class, reduced to the members the callback touches ----------------------------------------------------
struct IcpOdometer {
  const double ICP_FITNESS_THRESH = 0.1, ICP_MAX_CORR_DIST = 1.0, ICP_EPSILON = 1e-06;  // icp_odometer.h:62-64
  double ICP_MAX_ITERS = 10;                                                               // icp_odometer.h:65
  float voxel_leaf_size_ = 0.2f;                                                           // config/icpslam.yaml:14
  Cloud::Ptr prev_cloud_{new Cloud()}, curr_cloud_{new Cloud()};
  std::vector<std::string> lines;
  // ICPGPU_DEMO_TIMING=1: wall time per stage of the callback, summed (printed as a STAGES line behind TIMING)
  double stage_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  int stage_n = 0;
  bool stage_timing = std::getenv("ICPGPU_DEMO_TIMING") != nullptr;

  void voxelFilterCloud(Cloud::Ptr* input, Cloud::Ptr* output) {  // icp_odometer.cpp:96-101
    icpgpu::VoxelGrid<Cloud> voxel_filter;                        // was: pcl::VoxelGrid<pcl::PointXYZ>
    voxel_filter.setInputCloud(*input);
    voxel_filter.setLeafSize(voxel_leaf_size_, voxel_leaf_size_, voxel_leaf_size_);
    voxel_filter.filter(**output);
  }

  void laserCloudCallback(int k, Cloud::Ptr cloud_msg) {  // icp_odometer.cpp:147-220 (ROS / tf parts left out)
    Cloud::Ptr input_cloud = cloud_msg;  // pcl::fromROSMsg(*cloud_msg, *input_cloud): the message conversion is the caller's, not timed here
    auto t_mark = std::chrono::steady_clock::now();
    auto mark = [&](int stage) {
      if (!stage_timing || k < 4) return;  // (the first callbacks create the contexts and size every buffer)
      const auto now = std::chrono::steady_clock::now();
      stage_us[stage] += std::chrono::duration<double, std::micro>(now - t_mark).count();
      t_mark = now;
    };
    voxelFilterCloud(&input_cloud, &curr_cloud_);
    mark(0);
    if (prev_cloud_->points.size() == 0) {
      *prev_cloud_ = *curr_cloud_;
      return;
    }
    icpgpu::GeneralizedIterativeClosestPoint<Cloud> icp;         // was: pcl::GeneralizedIterativeClosestPoint<PointXYZ, PointXYZ>
    icp.setMaximumIterations(ICP_MAX_ITERS);
    icp.setTransformationEpsilon(ICP_EPSILON);
    icp.setMaxCorrespondenceDistance(ICP_MAX_CORR_DIST);
    icp.setRANSACIterations(0);
    mark(1);
    icp.setInputSource(curr_cloud_);
    icp.setInputTarget(prev_cloud_);
    mark(2);
    Cloud::Ptr curr_cloud_in_prev_frame(new Cloud());
    icp.align(*curr_cloud_in_prev_frame);
    mark(3);
    const auto T = icp.getFinalTransformation();
    const bool conv = icp.hasConverged();
    const double fit = conv ? icp.getFitnessScore() : -1.0;
    mark(4);
    char buf[640];
    int o = std::snprintf(buf, sizeof buf, "%d %d %d %.17g %zu", k, conv ? 1 : 0, icp.getResult().iterations, fit, curr_cloud_->points.size());
    for (int i = 0; i < 16; ++i) o += std::snprintf(buf + o, sizeof buf - o, " %.9g", T.data()[i]);
    lines.emplace_back(buf);
    if (conv && fit < 20) *prev_cloud_ = *curr_cloud_;            // :201-210 (updateICPOdometry always succeeds here)
    mark(5);
    stage_n += k >= 4 ? 1 : 0;
  }
};

int main(int argc, char** argv) {
  if (argc < 9) {
    std::fprintf(stderr, "usage: %s scanA.bin nA scanB.bin nB n_scans leaf max_iters threads [warmup]\n", argv[0]);
    return 2;
  }
  Cloud::Ptr scans[2] = {load(argv[1], std::strtoull(argv[2], nullptr, 10)), load(argv[3], std::strtoull(argv[4], nullptr, 10))};
  const int n_scans = std::atoi(argv[5]);
  const int n_threads = std::max(1, std::atoi(argv[8]));
  const int warmup = argc > 9 ? std::atoi(argv[9]) : 3;
  IcpOdometer odo;
  odo.voxel_leaf_size_ = (float)std::atof(argv[6]);
  odo.ICP_MAX_ITERS = std::atof(argv[7]);

  std::mutex m;
  std::condition_variable cv;
  int next = 0;  // the scan whose callback runs next
  std::string error;
  std::chrono::steady_clock::time_point t0, t1;
  auto worker = [&](int id) {
    for (;;) {
      std::unique_lock<std::mutex> lk(m);
      cv.wait(lk, [&] { return next >= n_scans || next % n_threads == id; });
      if (next >= n_scans) return;
      const int k = next;
      if (k == warmup) t0 = std::chrono::steady_clock::now();
      try {
        odo.laserCloudCallback(k, scans[k % 2]);  // (under the lock: one callback at a time, as the subscriber guarantees)
      } catch (const std::exception& e) {
        error = e.what();
        next = n_scans;
        cv.notify_all();
        return;
      }
      if (k == n_scans - 1) t1 = std::chrono::steady_clock::now();
      ++next;
      cv.notify_all();
    }
  };
  std::vector<std::thread> pool;
  for (int t = 0; t < n_threads; ++t) pool.emplace_back(worker, t);
  for (auto& t : pool) t.join();
  if (!error.empty()) {
    std::fprintf(stderr, "%s\n", error.c_str());
    return 3;
  }
  for (const auto& l : odo.lines) std::printf("%s\n", l.c_str());
  const double secs = std::chrono::duration<double>(t1 - t0).count();
  const int timed = n_scans - warmup;
  std::printf("TIMING scans %d seconds %.6f scans_per_sec %.3f warmup %d threads %d\n", timed, secs, timed > 0 && secs > 0 ? timed / secs : 0.0,
              warmup, n_threads);
  if (odo.stage_timing) icpgpu::release_cached_contexts();  // (the development library prints its own stage timers when a context ends)
  if (odo.stage_timing && odo.stage_n)
    std::printf("STAGES us per callback (%d callbacks, the first four left out): VoxelGrid::filter %.1f | object + setters %.1f | setInputSource + setInputTarget (calls deferred to align) %.1f | "
                "align %.1f | getFinalTransformation + hasConverged + getFitnessScore %.1f | result line + *prev = *curr %.1f\n",
                odo.stage_n, odo.stage_us[0] / odo.stage_n, odo.stage_us[1] / odo.stage_n, odo.stage_us[2] / odo.stage_n, odo.stage_us[3] / odo.stage_n,
                odo.stage_us[4] / odo.stage_n, odo.stage_us[5] / odo.stage_n);
#if defined(ICPGPU_SHIM_TIMING)
  {  // (all callbacks, the warm-up ones included)
    const double* u = icpgpu::detail::shim_us_array();
    std::printf("SHIM us per align (%d aligns): uploads / recognitions %.1f | icpgpu_align_view %.1f | output.points.assign %.1f\n", n_scans - 1,
                u[1] / (n_scans - 1), u[2] / (n_scans - 1), u[3] / (n_scans - 1));
  }
#endif
  return 0;
}
