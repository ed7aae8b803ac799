// Compile-only check of the drop-in boundary: the statements below are the reference's own uses of cuda::DeviceMap
// and the device classes (file:line given per statement) against include/warpsense_hip/compat.hpp,
// reached through the shipped forwarding headers include/warpsense/cuda/{cleanup,device_map,device_map_wrapper,registration,update_tsdf}.h.
// HDF5LocalMap is replaced by a stand-in with the same accessors (the real one needs HighFive): the point is that the
// CALL SITES compile unchanged, not that the map does.
#include <array>
#include <memory>
#include <vector>

// through the names the reference's sources use (tsdf_mapping.cpp:1, tsdf_mapping.h:7-9, pcd2tsdf.cpp:20): the forwarding
// headers of include/warpsense/cuda/, found ahead of the reference's own by -I order
#include "warpsense/cuda/cleanup.h"
#include "warpsense/cuda/device_map.h"
#include "warpsense/cuda/device_map_wrapper.h"
#include "warpsense/cuda/registration.h"
#include "warpsense/cuda/update_tsdf.h"

#if defined(CALLSITES_EXPECT_REFERENCE_TYPES) && !defined(WARPSENSE_HIP_USE_REFERENCE_TYPES)
#error "with the reference's include directory on the path the forwarding headers must pick the reference's own math types"
#endif

struct Vec3iLike // what Eigen::Vector3i offers to device_map.h:42-48
{
  std::array<int, 3> v{};
  int *data() { return v.data(); }
  const int *data() const { return v.data(); }
};
class HDF5LocalMap
{
public:
  using Ptr = std::shared_ptr<HDF5LocalMap>;
  HDF5LocalMap(int sx, int sy, int sz) : data_((size_t)sx * sy * sz)
  {
    size_.v = {sx, sy, sz};
    offset_.v = {sx / 2, sy / 2, sz / 2};
  }
  Vec3iLike &get_size() { return size_; }
  Vec3iLike &get_offset() { return offset_; }
  Vec3iLike &get_pos() { return pos_; }
  TSDFEntry *get_data() { return data_.data(); }

private:
  Vec3iLike size_, offset_, pos_;
  std::vector<TSDFEntry> data_;
};

int callsites(bool run)
{
  std::shared_ptr<HDF5LocalMap> hdf5_local_map_ = std::make_shared<HDF5LocalMap>(5, 5, 5);
  std::shared_ptr<HDF5LocalMap> cuda_local_map = hdf5_local_map_;
  using namespace cuda;
  if (!run) return 0;
  {
    DeviceMap existing_cuda_map(hdf5_local_map_); // src/warpsense/tsdf_mapping.cpp:114
    TSDFCuda tsdf(existing_cuda_map, 600, 10 * 64, 64); // tsdf_mapping.cpp:30-41 (ctor arguments of the member)
    tsdf.avg_map().to_host(existing_cuda_map);     // tsdf_mapping.cpp:116
    tsdf.avg_map().to_device(existing_cuda_map);   // tsdf_mapping.cpp:122
    tsdf.new_map().update_params(existing_cuda_map); // tsdf_mapping.cpp:123
  }
  {
    DeviceMap existing_cuda_map(hdf5_local_map_); // tsdf_mapping.cpp:141
    (void)existing_cuda_map.get_size();
  }
  {
    cuda::DeviceMap avg_cuda_map(hdf5_local_map_); // src/warpsense/app.cpp:218
    (void)avg_cuda_map.in_bounds(0, 0, 0);
  }
  {
    cuda::DeviceMap existing_cuda_map(hdf5_local_map_); // src/featsense/mapping.cpp:187
    (void)existing_cuda_map.get_index(rmagine::Vector3i(0, 0, 0));
  }
  {
    cuda::DeviceMap cuda_map(cuda_local_map); // test/pcd2tsdf.cpp:117
    cuda::TSDFCuda tsdf(cuda_map, 600, 10 * 64, 64); // test/pcd2tsdf.cpp:118
    std::vector<rmagine::Pointi> points_rm(1);
    rmagine::Pointi pos(0, 0, 0), up(0, 0, 32768);
    tsdf.update_tsdf(cuda_map, points_rm, pos, up); // test/pcd2tsdf.cpp:119-ff (download overload)
    cuda::RegistrationCuda reg(cuda_map);             // test/pcd_registration.cpp:297
    reg.prepare_registration(points_rm);              // test/pcd_registration.cpp:298
    rmagine::Matrix4x4f T;
    T.setIdentity();
    rmagine::Matrix6x6l h;
    rmagine::Point6l g;
    int e = 0, c = 0;
    reg.perform_registration(tsdf.device_map(), &T, h, g, e, c, 64); // tsdf_registration.cpp:57
    // include/warpsense/cuda/device_map.h:116-128, public members
    (void)cuda_map.in_bounds_with_buffer_neg(rmagine::Vector3i(1, 1, 1), 1);
    (void)cuda_map.in_bounds_with_buffer_pos(rmagine::Vector3i(3, 0, 0), 1);
    cuda::pause();
    // beyond the reference: the multi-GPU loop of this library through the same class
    unsigned char handle[WS_IPC_HANDLE_BYTES];
    reg.peer_mailbox(handle);
    reg.peer_connect(0, 1, handle);
    rmagine::Matrix4x4f out;
    int its = 0;
    (void)reg.register_cloud_peers(tsdf.device_map(), 0, points_rm.size(), T, 200, 0.1f, 0.03f, 64, out, &its);
  }
  return 0;
}
