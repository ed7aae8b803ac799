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
  int last_iterations() const { return last_iterations_; }
  RegistrationCuda &registration() { return *reg_; }

private:
  std::unique_ptr<RegistrationCuda> reg_;
  int last_iterations_ = 0;
};

} // namespace cuda
