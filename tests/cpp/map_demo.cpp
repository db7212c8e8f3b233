// map_demo.cpp -- the mapper's sequence (octree_mapper.cpp:133-172) written against the C++ shim.
// usage: map_demo <scan0.bin> <n0> <scan1.bin> <n1> <pose1: 16 floats column-major> <pose1_inv: 16 floats>
// prints: map_size_after_seed n_nn converged iterations T[16] map_size_after_growth checksum(nn cloud)
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
  if (argc < 5 + 32) return 2;
  auto scan0 = load(argv[1], std::strtoull(argv[2], nullptr, 10));
  auto cloud = load(argv[3], std::strtoull(argv[4], nullptr, 10));
  float pose[16], pose_inv[16];
  for (int i = 0; i < 16; ++i) {
    pose[i] = (float)std::atof(argv[5 + i]);
    pose_inv[i] = (float)std::atof(argv[21 + i]);
  }
  try {
    icpgpu::OctreeMap<mock_pcl::PointCloud> map(/*octree_resolution_=*/0.5);
    map.addPointsToMap(*scan0, icpgpu::Matrix4::Identity());                       // first scan: the map is empty (:137-141)
    const std::size_t seeded = map.size();
    mock_pcl::PointCloud::Ptr nn_cloud(new mock_pcl::PointCloud());
    map.approxNearestNeighbors(*cloud, icpgpu::make_matrix4(pose), icpgpu::make_matrix4(pose_inv), *nn_cloud);  // :145-146
    icpgpu::IterativeClosestPoint<mock_pcl::PointCloud> icp;                      // estimateTransformICP, :104-117
    icp.setMaximumIterations(30);
    icp.setTransformationEpsilon(1e-6);
    icp.setMaxCorrespondenceDistance(1.0);
    icp.setRANSACIterations(0);
    icp.setInputSource(cloud);
    icp.setInputTargetFromMap();
    mock_pcl::PointCloud aligned;
    icp.align(aligned);
    const auto T = icp.getFinalTransformation();
    double checksum = 0.0;
    for (const auto& p : nn_cloud->points) checksum += (double)p.x + 2.0 * p.y + 3.0 * p.z + p.pad;
    std::printf("%zu %zu %d %d", seeded, nn_cloud->size(), icp.hasConverged() ? 1 : 0, icp.getResult().iterations);
    for (int i = 0; i < 16; ++i) std::printf(" %.9g", T.data()[i]);
    map.addPointsToMap(*cloud, icpgpu::make_matrix4(pose));                        // grown with the (here: raw) pose, :152
    std::printf(" %zu %.17g\n", map.size(), checksum);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "%s\n", e.what());
    return 3;
  }
  return 0;
}
