// gicp_shim_demo.cpp -- the reference's call site as it is literally written (pcl::GeneralizedIterativeClosestPoint,
// icp_odometer.cpp:186-201) against icpgpu::GeneralizedIterativeClosestPoint, with a cloud type that carries PCL's
// width / height / is_dense, plus the two-objects-on-one-thread case (odometer and mapper share the cached context):
// object A aligns, object B aligns another pair, A.getFitnessScore() must still be A's.
// usage: gicp_shim_demo <src.bin> <n_src> <tgt.bin> <n_tgt> <max_iters>
// line 1: converged iterations fitness T[16] width height is_dense        (GICP, object A)
// line 2: fitness of A asked AFTER object B (point-to-point, swapped clouds) used the context; fitness of B
#include <cstdio>
#include <string>
#include <cstdlib>
#include <memory>
#include <vector>

#include "icpgpu_registration.hpp"

namespace mock_pcl {  // memory layout and metadata members of pcl::PointCloud<pcl::PointXYZ> (PCL is not in this image)
struct alignas(16) PointXYZ {
  float x, y, z, pad;
};
struct PointCloud {
  std::vector<PointXYZ> points;
  unsigned width = 7, height = 3;  // stale values, as a reused cloud would carry
  bool is_dense = false;
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
    icpgpu::GeneralizedIterativeClosestPoint<mock_pcl::PointCloud> icp;
    icp.setMaximumIterations(ICP_MAX_ITERS);
    icp.setTransformationEpsilon(ICP_EPSILON);
    icp.setMaxCorrespondenceDistance(ICP_MAX_CORR_DIST);
    icp.setRANSACIterations(0);
    if (argc > 6 && std::string(argv[6]) == "quadratic") icp.setQuadraticInnerSolver(true);  // (not a PCL method: icpgpu.h, icpgpu_gicp_inner)
    icp.setInputSource(curr_cloud_);
    icp.setInputTarget(prev_cloud_);
    mock_pcl::PointCloud::Ptr out(new mock_pcl::PointCloud());
    icp.align(*out);
    const auto T = icp.getFinalTransformation();
    const double fit_a = icp.getFitnessScore();
    std::printf("%d %d %.17g", icp.hasConverged() ? 1 : 0, icp.getResult().iterations, fit_a);
    for (int i = 0; i < 16; ++i) std::printf(" %.9g", T.data()[i]);
    std::printf(" %u %u %d\n", out->width, out->height, out->is_dense ? 1 : 0);

    icpgpu::IterativeClosestPoint<mock_pcl::PointCloud> other;  // same thread, same device: the same cached context
    other.setMaximumIterations(3);
    other.setInputSource(prev_cloud_);
    other.setInputTarget(curr_cloud_);
    mock_pcl::PointCloud scratch;
    other.align(scratch);
    std::printf("%.17g %.17g\n", icp.getFitnessScore(), other.getFitnessScore());
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  return 0;
}
