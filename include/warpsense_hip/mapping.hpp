// mapping.hpp — ROS-free, Eigen-free twins of the reference's host callers of the device API:
//
//   cuda::TSDFMapping        src/warpsense/tsdf_mapping.cpp:30-95   (protected ROS-free ctor :30-41, update_tsdf :62-95,
//                                                                     convert_pose_to_gpu :77-85)
//   cuda::TSDFRegistration   src/warpsense/tsdf_registration.cpp:22-96
//
// The reference versions take Eigen::Matrix4f; rmagine::Matrix4x4f is bit-compatible with it (column-major
// 4x4 float, include/warpsense/math/matrix4x4.h:13-21), which is what the reference itself relies on when it
// reinterpret_casts between the two (tsdf_registration.cpp:32).  Map ownership stays with the caller.
#pragma once

#include <cmath>
#include <memory>
#include <mutex>
#include <shared_mutex>

#include "warpsense_hip/compat.hpp"

namespace cuda
{
struct HotPathParams // the knobs of params/params.yaml the hot path reads (include/params/map_params.h:52-98)
{
  int map_resolution = 64;        // mm per voxel
  int tau = 1000;                 // map/max_distance * 1000
  int max_weight = 10 * 64;       // map/max_weight * WEIGHT_RESOLUTION
  int max_iterations = 200;       // registration/max_iterations
  float it_weight_gradient = 0.1f;
  float epsilon = 0.03f;
};

class TSDFMapping
{
public:
  TSDFMapping(const HotPathParams &params, DeviceMap &local_map)
      : params_(params), cuda_map_(local_map),
        tsdf_(std::make_unique<TSDFCuda>(cuda_map_, params.tau, params.max_weight, params.map_resolution))
  {
  }
  virtual ~TSDFMapping() = default;

  // tsdf_mapping.cpp:77-85 with include/util/util.h:8-18,52-56
  void convert_pose_to_gpu(const rmagine::Matrix4x4f &pose, rmagine::Pointi &pos_rm, rmagine::Pointi &up_rm) const
  {
    const int MR = 32768;
    int R[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j) R[i][j] = (int)(pose.at(i, j) * MR);
    // (rotation_mat * (0,0,MR,1)).head(3) / MR with a zero translation column
    up_rm = rmagine::Pointi(R[0][2] * MR / MR, R[1][2] * MR / MR, R[2][2] * MR / MR);
    pos_rm = rmagine::Pointi((int)std::floor(pose.at(0, 3) / (float)params_.map_resolution),
                             (int)std::floor(pose.at(1, 3) / (float)params_.map_resolution),
                             (int)std::floor(pose.at(2, 3) / (float)params_.map_resolution));
  }

  void update_tsdf(const std::vector<rmagine::Pointi> &scan_points, const rmagine::Matrix4x4f &pose)
  {
    rmagine::Pointi pos_rm, up_rm;
    convert_pose_to_gpu(pose, pos_rm, up_rm);
    std::unique_lock lock(mutex_);
    tsdf_->update_tsdf(scan_points, pos_rm, up_rm);
  }
  void update_tsdf(const std::vector<rmagine::Pointi> &scan_points, const rmagine::Pointi &pos_rm, const rmagine::Pointi &up_rm)
  {
    std::unique_lock lock(mutex_);
    tsdf_->update_tsdf(scan_points, pos_rm, up_rm);
  }
  void update_tsdf(DeviceMap &result, const std::vector<rmagine::Pointi> &scan_points, const rmagine::Matrix4x4f &pose)
  {
    rmagine::Pointi pos_rm, up_rm;
    convert_pose_to_gpu(pose, pos_rm, up_rm);
    std::unique_lock lock(mutex_);
    tsdf_->update_tsdf(result, scan_points, pos_rm, up_rm);
  }
  TSDFCuda &tsdf() { return *tsdf_; }

protected:
  HotPathParams params_;
  DeviceMap &cuda_map_;
  std::unique_ptr<TSDFCuda> tsdf_;
  std::shared_mutex mutex_;
};

class TSDFRegistration : public TSDFMapping
{
public:
  TSDFRegistration(const HotPathParams &params, DeviceMap &local_map)
      : TSDFMapping(params, local_map), reg_(std::make_unique<RegistrationCuda>(cuda_map_))
  {
  }

  // tsdf_registration.cpp:28-96; the Gauss-Newton loop runs on the device
  rmagine::Matrix4x4f register_cloud(std::vector<rmagine::Pointi> &cloud, const rmagine::Matrix4x4f &pretransform)
  {
    reg_->prepare_registration(cloud);
    std::shared_lock lock(mutex_);
    return reg_->register_cloud(tsdf_->device_map(), pretransform, params_.max_iterations, params_.it_weight_gradient,
                                params_.epsilon, params_.map_resolution, &last_iterations_);
  }
  // The reference's OWN loop shape (tsdf_registration.cpp:55-92), for a caller that is relinked and not changed: one
  // perform_registration per iteration -- launch, 44 sums back to the host -- then the 6x6 solve (the reference: Eigen's
  // hf.inverse() * gf; here the elimination of oracle/ws_oracle.c:wso_solve6, so that the poses are the oracle's bit for bit),
  // xi_to_transform (registration/util.h:5-39) and the pose product on the host.  Same result as register_cloud above; timed
  // next to it by examples/dropin_bench.cpp (`relink only` against `with the one-line change`).
  rmagine::Matrix4x4f register_cloud_reference_loop(std::vector<rmagine::Pointi> &cloud, const rmagine::Matrix4x4f &pretransform)
  {
    rmagine::Matrix4x4f total = pretransform;
    const int center[3] = {(int)total.at(0, 3), (int)total.at(1, 3), (int)total.at(2, 3)};
    float alpha = 0.f, prev[4] = {0.f, 0.f, 0.f, 0.f};
    bool finished = false;
    rmagine::Matrix6x6l h;
    rmagine::Point6l g;
    int e = 0, c = 0;
    reg_->prepare_registration(cloud);
    std::shared_lock lock(mutex_);
    int i = 0;
    for (; i < params_.max_iterations && !finished; ++i)
    {
      reg_->perform_registration(tsdf_->device_map(), &total, h, g, e, c, params_.map_resolution);
      if (c == 0) // guard (the reference divides by zero at :80)
      {
        ++i;
        break;
      }
      double A[6][6], b[6], xi[6];
      const double w = (double)(alpha * (float)c); // alpha * gpu_c is a float product (:66)
      for (int r = 0; r < 6; ++r)
      {
        b[r] = (double)g.at(r);
        for (int q = 0; q < 6; ++q) A[r][q] = (double)h.at(r, q) + (r == q ? w : 0.0);
      }
      if (!detail_solve6(A, b, xi))
      {
        ++i;
        break;
      }
      for (double &v : xi) v = -v;
      rmagine::Matrix4x4f tr = detail_xi_to_transform(xi, center);
      alpha += params_.it_weight_gradient;
      rmagine::Matrix4x4f next;
      for (int col = 0; col < 4; ++col)
        for (int row = 0; row < 4; ++row)
        {
          float s = 0.f;
          for (int k = 0; k < 4; ++k) s += tr.at(row, k) * total.at(k, col);
          next.at(row, col) = s;
        }
      total = next;
      const float err = (float)e / c;
      if (std::fabs(err - prev[2]) < params_.epsilon && std::fabs(err - prev[0]) < params_.epsilon) finished = true;
      prev[0] = prev[1];
      prev[1] = prev[2];
      prev[2] = prev[3];
      prev[3] = err;
    }
    last_iterations_ = i;
    return total;
  }
  int last_iterations() const { return last_iterations_; }
  RegistrationCuda &registration() { return *reg_; }

  // Gauss-Jordan with partial pivoting, multipliers from the pivots' reciprocals: the operations of wso_solve6 in their order
  static bool detail_solve6(double (&A)[6][6], double (&b)[6], double (&x)[6])
  {
    double inv[6];
    for (int k = 0; k < 6; ++k)
    {
      int piv = k;
      double best = std::fabs(A[k][k]);
      for (int i = k + 1; i < 6; ++i)
        if (std::fabs(A[i][k]) > best)
        {
          best = std::fabs(A[i][k]);
          piv = i;
        }
      if (best == 0.0) return false;
      if (piv != k)
      {
        for (int j = 0; j < 6; ++j) std::swap(A[k][j], A[piv][j]);
        std::swap(b[k], b[piv]);
      }
      inv[k] = 1.0 / A[k][k];
      for (int i = 0; i < 6; ++i)
      {
        if (i == k) continue;
        const double f = A[i][k] * inv[k];
        for (int j = k + 1; j < 6; ++j) A[i][j] -= f * A[k][j];
        b[i] -= f * b[k];
      }
    }
    for (int i = 0; i < 6; ++i) x[i] = b[i] * inv[i];
    return true;
  }
  // registration/util.h:5-39 (the float / double mix of the reference's Eigen expression, as oracle/ws_oracle.c spells it out)
  static rmagine::Matrix4x4f detail_xi_to_transform(const double (&xi)[6], const int (&center)[3])
  {
    const double theta = std::sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
    float L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
    if (theta != 0.0)
    {
      const double lx = xi[0] / theta, ly = xi[1] / theta, lz = xi[2] / theta;
      L[0][1] = (float)-lz; L[0][2] = (float)ly;
      L[1][0] = (float)lz;  L[1][2] = (float)-lx;
      L[2][0] = (float)-ly; L[2][1] = (float)lx;
    }
    const float s = (float)std::sin(theta), omc = (float)(1 - std::cos(theta));
    float R[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
      {
        float ll = 0.f;
        for (int k = 0; k < 3; ++k) ll += (omc * L[i][k]) * L[k][j];
        R[i][j] = ((i == j ? 1.f : 0.f) + s * L[i][j]) + ll;
      }
    const float oc[3] = {-(float)center[0], -(float)center[1], -(float)center[2]};
    rmagine::Matrix4x4f T;
    for (int col = 0; col < 4; ++col)
      for (int row = 0; row < 4; ++row) T.at(row, col) = 0.f;
    T.at(3, 3) = 1.f;
    for (int i = 0; i < 3; ++i)
    {
      for (int j = 0; j < 3; ++j) T.at(i, j) = R[i][j];
      const float shift = ((R[i][0] * oc[0] + R[i][1] * oc[1]) + R[i][2] * oc[2]) + 0.f * 1.f;
      T.at(i, 3) = (shift + (float)center[i]) + (float)xi[3 + i];
    }
    return T;
  }

private:
  std::unique_ptr<RegistrationCuda> reg_;
  int last_iterations_ = 0;
};

} // namespace cuda
