// shim_demo.cpp -- a caller written like the reference's call site (icp_odometer.cpp:186-201) against the C++ shim.
// usage: shim_demo <src.bin> <n_src> <tgt.bin> <n_tgt> <max_iters>     (clouds: raw float32 x,y,z,pad records)
// prints: converged iterations fitness T[16] (column-major) checksum-of-aligned-cloud
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "icpgpu_registration.hpp"

namespace mock_pcl {  // stand-in with the memory layout of pcl::PointXYZ / pcl::PointCloud (PCL is not in this image)
struct alignas(16) PointXYZ {
  float x, y, z, pad;
};
struct PointCloud {
  std::vector<PointXYZ> points;
  std::size_t size() const { return points.size(); }
  using Ptr = std::shared_ptr<PointCloud>;
};
}  // namespace mock_pcl

static mock_pcl::PointCloud::Ptr load(const char* path, std::size_t n) {
  auto c = std::make_shared<mock_pcl::PointCloud>();
  c->points.resize(n);
  FILE* f = std::fopen(path, "rb");
  if (!f) { std::perror(path); std::exit(2); }
  if (n && std::fread(c->points.data(), sizeof(mock_pcl::PointXYZ), n, f) != n) { std::fprintf(stderr, "short read\n"); std::exit(2); }
  std::fclose(f);
  return c;
}

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const double ICP_MAX_CORR_DIST = 1.0, ICP_EPSILON = 1e-06, ICP_MAX_ITERS = std::atof(argv[5]);
  auto curr_cloud_ = load(argv[1], std::strtoull(argv[2], nullptr, 10));
  auto prev_cloud_ = load(argv[3], std::strtoull(argv[4], nullptr, 10));
  try {
    icpgpu::IterativeClosestPoint<mock_pcl::PointCloud> icp;
    icp.setMaximumIterations(ICP_MAX_ITERS);
    icp.setTransformationEpsilon(ICP_EPSILON);
    icp.setMaxCorrespondenceDistance(ICP_MAX_CORR_DIST);
    icp.setRANSACIterations(0);
    icp.setInputSource(curr_cloud_);
    icp.setInputTarget(prev_cloud_);
    mock_pcl::PointCloud::Ptr curr_cloud_in_prev_frame(new mock_pcl::PointCloud());
    icp.align(*curr_cloud_in_prev_frame);
    const auto T = icp.getFinalTransformation();
    const bool ok = icp.hasConverged() && icp.getFitnessScore() < 20;
    double checksum = 0.0;
    for (const auto& p : curr_cloud_in_prev_frame->points) checksum += (double)p.x + 2.0 * p.y + 3.0 * p.z + p.pad;
    std::printf("%d %d %.17g", ok ? 1 : 0, icp.getResult().iterations, icp.getFitnessScore());
    for (int i = 0; i < 16; ++i) std::printf(" %.9g", T.data()[i]);
    std::printf(" %.17g\n", checksum);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  return 0;
}
