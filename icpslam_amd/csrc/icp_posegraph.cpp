// icp_posegraph.cpp -- the data contract either side of the hot path for BASELINE config 5:
// pose chaining of the per-scan ICP transforms, keyframe selection, and the pose graph handed to the (unchanged,
// external, CPU) g2o backend -- written as g2o's text format so `pose_graph_utils`/g2o can load it as is.
//
// Restates, in plain double arithmetic (no Eigen in this image):
//   Pose6DOF::fromEigenMatrix   /root/reference/src/utils/pose6DOF.cpp:185-190   (matrix -> position + unit quaternion)
//   Pose6DOF::compose           /root/reference/src/utils/pose6DOF.cpp:98-105    (pos = p1 + R1 p2, rot = q1 q2, normalised)
//   Pose6DOF::inverse           /root/reference/src/utils/pose6DOF.cpp:117-122
//   Pose6DOF::distanceEuclidean /root/reference/src/utils/pose6DOF.cpp:94-96     (|subtract(p1,p2).pos| = |p2 - p1|)
//   IcpOdometer::updateICPOdometry  /root/reference/src/icpslam/icp_odometer.cpp:109-113 (new_pose = prev_pose (+) T)
//   IcpSlam::mainLoop keyframe rule /root/reference/src/icpslam/icpslam.cpp:143-152 (first pose, or moved > KFS_DIST_THRESH = 0.3 m)
//   IcpSlam::addNewKeyframe         /root/reference/src/icpslam/icpslam.cpp:70-89  (vertex = pose in odom, T_map_to_odom = I before
//                                    any optimisation; edge new -> prev with measurement new^-1 (+) prev, information = diag)
//   information diagonal            /root/reference/config/icpslam.yaml:21 (icp_information_matrix)
#include <cmath>
#include <cstdio>
#include <cstring>
#include <new>
#include <vector>

#include "../../include/icpgpu.h"

namespace {

struct Quat {
  double x, y, z, w;
};

Quat quat_normalized(Quat q) {
  const double n = std::sqrt(q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w);
  if (n > 0) {
    q.x /= n;
    q.y /= n;
    q.z /= n;
    q.w /= n;
  } else {
    q = {0, 0, 0, 1};
  }
  return q;
}

Quat quat_mul(const Quat& a, const Quat& b) {
  return {a.w * b.x + a.x * b.w + a.y * b.z - a.z * b.y, a.w * b.y - a.x * b.z + a.y * b.w + a.z * b.x,
          a.w * b.z + a.x * b.y - a.y * b.x + a.z * b.w, a.w * b.w - a.x * b.x - a.y * b.y - a.z * b.z};
}

Quat quat_conj(const Quat& q) { return {-q.x, -q.y, -q.z, q.w}; }

void quat_rotate(const Quat& q, const double v[3], double out[3]) {
  // v' = v + 2 w (u x v) + 2 u x (u x v), u = (x, y, z)
  const double u[3] = {q.x, q.y, q.z};
  const double c1[3] = {u[1] * v[2] - u[2] * v[1], u[2] * v[0] - u[0] * v[2], u[0] * v[1] - u[1] * v[0]};
  const double c2[3] = {u[1] * c1[2] - u[2] * c1[1], u[2] * c1[0] - u[0] * c1[2], u[0] * c1[1] - u[1] * c1[0]};
  for (int a = 0; a < 3; ++a) out[a] = v[a] + 2.0 * (q.w * c1[a] + c2[a]);
}

// rotation matrix (row-major r[3][3]) -> quaternion, the branch structure of Eigen::Quaternion(Matrix3)
Quat quat_from_matrix(const double r[3][3]) {
  Quat q;
  const double t = r[0][0] + r[1][1] + r[2][2];
  if (t > 0.0) {
    double s = std::sqrt(t + 1.0);
    q.w = 0.5 * s;
    s = 0.5 / s;
    q.x = (r[2][1] - r[1][2]) * s;
    q.y = (r[0][2] - r[2][0]) * s;
    q.z = (r[1][0] - r[0][1]) * s;
  } else {
    int i = 0;
    if (r[1][1] > r[0][0]) i = 1;
    if (r[2][2] > r[i][i]) i = 2;
    const int j = (i + 1) % 3, k = (j + 1) % 3;
    double s = std::sqrt(r[i][i] - r[j][j] - r[k][k] + 1.0);
    double v[3];
    v[i] = 0.5 * s;
    s = 0.5 / s;
    q.w = (r[k][j] - r[j][k]) * s;
    v[j] = (r[j][i] + r[i][j]) * s;
    v[k] = (r[k][i] + r[i][k]) * s;
    q.x = v[0];
    q.y = v[1];
    q.z = v[2];
  }
  return quat_normalized(q);
}

Quat Q(const icpgpu_pose& p) { return {p.quat[0], p.quat[1], p.quat[2], p.quat[3]}; }
void setQ(icpgpu_pose& p, const Quat& q) {
  p.quat[0] = q.x;
  p.quat[1] = q.y;
  p.quat[2] = q.z;
  p.quat[3] = q.w;
}

icpgpu_pose identity_pose() {
  icpgpu_pose p;
  p.pos[0] = p.pos[1] = p.pos[2] = 0.0;
  p.quat[0] = p.quat[1] = p.quat[2] = 0.0;
  p.quat[3] = 1.0;
  return p;
}

icpgpu_pose compose(const icpgpu_pose& a, const icpgpu_pose& b) {
  icpgpu_pose o;
  double rb[3];
  quat_rotate(Q(a), b.pos, rb);
  for (int k = 0; k < 3; ++k) o.pos[k] = a.pos[k] + rb[k];
  setQ(o, quat_normalized(quat_mul(Q(a), Q(b))));
  return o;
}

icpgpu_pose inverse(const icpgpu_pose& a) {
  icpgpu_pose o;
  const Quat qi = quat_conj(Q(a));  // unit quaternion: inverse == conjugate
  double r[3];
  quat_rotate(qi, a.pos, r);
  for (int k = 0; k < 3; ++k) o.pos[k] = -r[k];
  setQ(o, qi);
  return o;
}

}  // namespace

struct icpgpu_posegraph {
  double kf_dist = 0.3;
  double info[6] = {0.06, 0.06, 10.0, 0.001, 0.001, 2.0};
  icpgpu_pose latest = identity_pose();
  icpgpu_pose last_kf_pose = identity_pose();
  std::vector<icpgpu_pose> poses;      // one per accepted transform (icp_odom_poses_)
  std::vector<icpgpu_pose> keyframes;  // pose_in_odom of every keyframe
  std::vector<long> keyframe_scan;     // index (into the pushed transforms) that created the keyframe
  long pushed = 0;
};

extern "C" {

int icpgpu_pose_from_matrix(const float* T, icpgpu_pose* out) {
  if (!T || !out) return ICPGPU_ERR_INVALID_ARG;
  double r[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) r[i][j] = (double)T[j * 4 + i];  // column-major input, cast like `.cast<double>()`
  out->pos[0] = (double)T[12];
  out->pos[1] = (double)T[13];
  out->pos[2] = (double)T[14];
  setQ(*out, quat_from_matrix(r));
  return ICPGPU_OK;
}

int icpgpu_pose_to_matrix(const icpgpu_pose* p, float* T) {
  if (!p || !T) return ICPGPU_ERR_INVALID_ARG;
  // tf::Matrix3x3::setRotation (what Pose6DOF::toTFTransform feeds pcl_ros::transformPointCloud), then the cast to float
  const double x = p->quat[0], y = p->quat[1], z = p->quat[2], w = p->quat[3];
  const double d = x * x + y * y + z * z + w * w;
  if (!(d > 0.0)) return ICPGPU_ERR_INVALID_ARG;
  const double s = 2.0 / d;
  const double xs = x * s, ys = y * s, zs = z * s;
  const double wx = w * xs, wy = w * ys, wz = w * zs, xx = x * xs, xy = x * ys, xz = x * zs, yy = y * ys, yz = y * zs, zz = z * zs;
  const double r[3][3] = {{1.0 - (yy + zz), xy - wz, xz + wy}, {xy + wz, 1.0 - (xx + zz), yz - wx}, {xz - wy, yz + wx, 1.0 - (xx + yy)}};
  for (int i = 0; i < 3; ++i) {
    for (int j = 0; j < 3; ++j) T[j * 4 + i] = (float)r[i][j];
    T[12 + i] = (float)p->pos[i];
    T[i * 4 + 3] = 0.0f;
  }
  T[15] = 1.0f;
  return ICPGPU_OK;
}

int icpgpu_pose_compose(const icpgpu_pose* a, const icpgpu_pose* b, icpgpu_pose* out) {
  if (!a || !b || !out) return ICPGPU_ERR_INVALID_ARG;
  *out = compose(*a, *b);
  return ICPGPU_OK;
}

int icpgpu_pose_inverse(const icpgpu_pose* a, icpgpu_pose* out) {
  if (!a || !out) return ICPGPU_ERR_INVALID_ARG;
  *out = inverse(*a);
  return ICPGPU_OK;
}

int icpgpu_posegraph_create(icpgpu_posegraph** out, double keyframe_distance, const double* information_diag6) {
  if (!out) return ICPGPU_ERR_INVALID_ARG;
  icpgpu_posegraph* g = new (std::nothrow) icpgpu_posegraph();
  if (!g) return ICPGPU_ERR_OOM;
  if (keyframe_distance >= 0.0) g->kf_dist = keyframe_distance;
  if (information_diag6) std::memcpy(g->info, information_diag6, sizeof(g->info));
  *out = g;
  return ICPGPU_OK;
}

int icpgpu_posegraph_destroy(icpgpu_posegraph* g) {
  delete g;
  return ICPGPU_OK;
}

int icpgpu_posegraph_set_initial_pose(icpgpu_posegraph* g, const icpgpu_pose* p) {
  if (!g || !p || !g->poses.empty()) return ICPGPU_ERR_INVALID_ARG;
  g->latest = *p;
  setQ(g->latest, quat_normalized(Q(*p)));
  return ICPGPU_OK;
}

int icpgpu_posegraph_push(icpgpu_posegraph* g, const float* T, int accepted, long* keyframe_id) {
  if (!g || !T) return ICPGPU_ERR_INVALID_ARG;
  if (keyframe_id) *keyframe_id = -1;
  const long scan = g->pushed++;
  if (!accepted) return ICPGPU_OK;  // icp_odometer.cpp:201-210: a rejected registration changes nothing
  icpgpu_pose t;
  icpgpu_pose_from_matrix(T, &t);
  g->latest = compose(g->latest, t);  // icp_odometer.cpp:112-113
  g->poses.push_back(g->latest);
  double d2 = 0.0;
  for (int k = 0; k < 3; ++k) d2 += (g->latest.pos[k] - g->last_kf_pose.pos[k]) * (g->latest.pos[k] - g->last_kf_pose.pos[k]);
  if (g->keyframes.empty() || std::sqrt(d2) > g->kf_dist) {  // icpslam.cpp:143
    g->keyframes.push_back(g->latest);
    g->keyframe_scan.push_back(scan);
    g->last_kf_pose = g->latest;
    if (keyframe_id) *keyframe_id = (long)g->keyframes.size() - 1;
  }
  return ICPGPU_OK;
}

long icpgpu_posegraph_num_poses(const icpgpu_posegraph* g) { return g ? (long)g->poses.size() : -1; }
long icpgpu_posegraph_num_keyframes(const icpgpu_posegraph* g) { return g ? (long)g->keyframes.size() : -1; }

int icpgpu_posegraph_get_pose(const icpgpu_posegraph* g, long i, icpgpu_pose* out) {
  if (!g || !out || i < 0 || i >= (long)g->poses.size()) return ICPGPU_ERR_INVALID_ARG;
  *out = g->poses[(size_t)i];
  return ICPGPU_OK;
}

int icpgpu_posegraph_get_keyframe(const icpgpu_posegraph* g, long i, icpgpu_pose* out, long* scan_index) {
  if (!g || !out || i < 0 || i >= (long)g->keyframes.size()) return ICPGPU_ERR_INVALID_ARG;
  *out = g->keyframes[(size_t)i];
  if (scan_index) *scan_index = g->keyframe_scan[(size_t)i];
  return ICPGPU_OK;
}

int icpgpu_posegraph_get_edge(const icpgpu_posegraph* g, long new_kf, icpgpu_pose* out) {
  if (!g || !out || new_kf < 1 || new_kf >= (long)g->keyframes.size()) return ICPGPU_ERR_INVALID_ARG;
  *out = compose(inverse(g->keyframes[(size_t)new_kf]), g->keyframes[(size_t)new_kf - 1]);  // icpslam.cpp:82
  return ICPGPU_OK;
}

int icpgpu_posegraph_write_g2o(const icpgpu_posegraph* g, const char* path) {
  if (!g || !path) return ICPGPU_ERR_INVALID_ARG;
  FILE* f = std::fopen(path, "w");
  if (!f) return ICPGPU_ERR_INVALID_ARG;
  for (size_t i = 0; i < g->keyframes.size(); ++i) {
    const icpgpu_pose& p = g->keyframes[i];
    std::fprintf(f, "VERTEX_SE3:QUAT %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g\n", i, p.pos[0], p.pos[1], p.pos[2], p.quat[0],
                 p.quat[1], p.quat[2], p.quat[3]);
  }
  for (size_t i = 1; i < g->keyframes.size(); ++i) {
    const icpgpu_pose e = compose(inverse(g->keyframes[i]), g->keyframes[i - 1]);
    std::fprintf(f, "EDGE_SE3:QUAT %zu %zu %.17g %.17g %.17g %.17g %.17g %.17g %.17g", i, i - 1, e.pos[0], e.pos[1], e.pos[2],
                 e.quat[0], e.quat[1], e.quat[2], e.quat[3]);
    for (int r = 0; r < 6; ++r)  // upper triangle of the 6x6 information matrix, row-major
      for (int c = r; c < 6; ++c) std::fprintf(f, " %.17g", r == c ? g->info[r] : 0.0);
    std::fprintf(f, "\n");
  }
  std::fclose(f);
  return ICPGPU_OK;
}

}  // extern "C"
