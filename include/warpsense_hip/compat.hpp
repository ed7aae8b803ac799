// compat.hpp — the reference's C++ device API re-implemented as thin inline classes over the C ABI
// (include/warpsense_hip.h).  Same namespaces, class names, member names and signatures as
//
//   include/warpsense/cuda/device_map.h:32-164          cuda::DeviceMap (+ overflow :14-30)
//   include/warpsense/cuda/device_map_wrapper.h:10-36   cuda::DeviceMapMemWrapper
//   include/warpsense/cuda/update_tsdf.h:9-34           cuda::TSDFCuda
//   include/warpsense/cuda/registration.h:10-45         cuda::RegistrationCuda
//   include/warpsense/cuda/cleanup.h:6-8                cuda::pause / cuda::cleanup
//   include/map/tsdf.h:16-140                           TSDFEntry
//   include/warpsense/math/*.h                          rmagine::Vector3 / Vector6 / Matrix4x4 / Matrix6x6 (POD layouts)
//
// so that src/warpsense/{tsdf_mapping,tsdf_registration}.cpp compile against it unchanged once the
// reference's four device headers forward to this file.  Error convention as in the
// reference: a failing device call prints the reason and exit(1)s (include/warpsense/cuda/common.cuh:10-21).
//
// When this header is used INSIDE the reference tree, define WARPSENSE_HIP_USE_REFERENCE_TYPES before
// including it: the rmagine:: math types and TSDFEntry then come from the reference's own headers.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <memory>
#include <vector>

#include "warpsense_hip.h"

#ifdef WARPSENSE_HIP_USE_REFERENCE_TYPES
#include "map/tsdf.h"
#include "warpsense/math/math.h"
#else
// ---- minimal POD twins of the reference's math types (layout-compatible, only what the hot path needs)
namespace rmagine
{
template <typename T>
struct Vector3
{
  T x{0}, y{0}, z{0};
  Vector3() = default;
  Vector3(T x_, T y_, T z_) : x(x_), y(y_), z(z_) {}
  T prod() const { return x * y * z; }
  bool operator==(const Vector3 &o) const { return x == o.x && y == o.y && z == o.z; }
};
using Vector3i = Vector3<int>;
using Pointi = Vector3<int>;
using Pointl = Vector3<long>;

template <typename T>
struct Vector6
{
  T data[6];
  T &at(unsigned i) { return data[i]; }
  const T &at(unsigned i) const { return data[i]; }
  T &operator[](unsigned i) { return data[i]; }
  const T &operator[](unsigned i) const { return data[i]; }
  void setZeros()
  {
    for (auto &v : data) v = 0;
  }
};
using Point6l = Vector6<long>;

// column-major like the reference: at(i, j) == data[j][i] (matrix4x4.h:175-185, matrix6x6.h:112-115)
template <typename S>
struct Matrix4x4
{
  S data[4][4];
  S &at(unsigned i, unsigned j) { return data[j][i]; }
  const S &at(unsigned i, unsigned j) const { return data[j][i]; }
  S &operator()(unsigned i, unsigned j) { return data[j][i]; }
  const S &operator()(unsigned i, unsigned j) const { return data[j][i]; }
  void setIdentity()
  {
    for (unsigned i = 0; i < 4; ++i)
      for (unsigned j = 0; j < 4; ++j) data[j][i] = (i == j) ? S(1) : S(0);
  }
};
using Matrix4x4f = Matrix4x4<float>;

template <typename S>
struct Matrix6x6
{
  S data[6][6];
  S &at(unsigned i, unsigned j) { return data[j][i]; }
  const S &at(unsigned i, unsigned j) const { return data[j][i]; }
  void setZeros()
  {
    for (auto &c : data)
      for (auto &v : c) v = 0;
  }
};
using Matrix6x6l = Matrix6x6<long>;
} // namespace rmagine

// include/map/tsdf.h:16-46 — value in the low half, weight in the high half of one uint32
class TSDFEntry
{
public:
  using RawType = uint32_t;
  using ValueType = int16_t;
  using WeightType = int16_t;
  TSDFEntry() = default;
  TSDFEntry(ValueType value, WeightType weight) { data_.tsdf.value = value; data_.tsdf.weight = weight; }
  explicit TSDFEntry(RawType raw) { data_.raw = raw; }
  RawType raw() const { return data_.raw; }
  void raw(RawType v) { data_.raw = v; }
  ValueType value() const { return data_.tsdf.value; }
  void value(ValueType v) { data_.tsdf.value = v; }
  WeightType weight() const { return data_.tsdf.weight; }
  void weight(WeightType w) { data_.tsdf.weight = w; }
  bool operator==(const TSDFEntry &o) const { return raw() == o.raw(); }

private:
  union
  {
    RawType raw;
    struct
    {
      ValueType value;
      WeightType weight;
    } tsdf;
  } data_;
};
static_assert(sizeof(TSDFEntry) == 4, "TSDFEntry must stay one 32-bit word");
#endif // WARPSENSE_HIP_USE_REFERENCE_TYPES

static_assert(sizeof(long) == 8, "the reference's `long` accumulators are 64 bit (test/test.cu:48-61)");
static_assert(sizeof(rmagine::Pointi) == 12 && sizeof(rmagine::Point6l) == 48 && sizeof(rmagine::Matrix4x4f) == 64 &&
                  sizeof(rmagine::Matrix6x6l) == 288,
              "POD layouts handed through the C ABI");

namespace cuda
{
namespace detail
{
inline void check(int rc, const char *what, const char *file, int line)
{
  if (rc != WS_OK)
  {
    fprintf(stderr, "Error: %s:%d, code: %d, reason: %s (%s)\n", file, line, rc, ws_last_error(), what);
    exit(1);
  }
}
inline ws_context *context()
{
  static ws_context *ctx = [] {
    ws_context *c = nullptr;
    check(ws_ctx_create(-1, &c), "ws_ctx_create", __FILE__, __LINE__);
    return c;
  }();
  return ctx;
}
} // namespace detail
#define WS_CHECK(call) ::cuda::detail::check((call), #call, __FILE__, __LINE__)

template <typename T>
inline T overflow(T val, T max)
{
  if (val >= 2 * max) return val - 2 * max;
  if (val >= max) return val - max;
  return val;
}

// Non-owning view of a ring-buffer map in HOST memory (the owner is the local map, device_map.h:66).
struct DeviceMap
{
  DeviceMap(int *size, int *offset, TSDFEntry *data, int *pos)
      : size_(reinterpret_cast<rmagine::Pointi *>(size)), offset_(reinterpret_cast<rmagine::Pointi *>(offset)), data_(data),
        pos_(reinterpret_cast<rmagine::Pointi *>(pos))
  {
  }
  DeviceMap(rmagine::Pointi *size, rmagine::Pointi *offset, TSDFEntry *data, rmagine::Pointi *pos)
      : size_(size), offset_(offset), data_(data), pos_(pos)
  {
  }
  // device_map.h:42-48: view of a local map held by a shared_ptr (HDF5LocalMap::Ptr in the reference).  A template so
  // that this header does not pull in the HDF5 map: any map type with get_size()/get_offset()/get_pos() returning
  // something with .data() -> int[3] (Eigen::Vector3i there) and get_data() -> TSDFEntry* binds.
  template <class LocalMapT>
  DeviceMap(const std::shared_ptr<LocalMapT> &map)
      : size_(reinterpret_cast<rmagine::Pointi *>(map->get_size().data())), offset_(reinterpret_cast<rmagine::Pointi *>(map->get_offset().data())),
        data_(map->get_data()), pos_(reinterpret_cast<rmagine::Pointi *>(map->get_pos().data()))
  {
  }
  DeviceMap() = default;
  DeviceMap(const DeviceMap &) = delete;
  DeviceMap(DeviceMap &&) = delete;
  DeviceMap operator=(const DeviceMap &) = delete;
  DeviceMap operator=(DeviceMap &&) = delete;

  const rmagine::Pointi *get_size() const { return size_; }
  const rmagine::Pointi *get_offset() const { return offset_; }
  const rmagine::Pointi *get_pos() const { return pos_; }

  int get_index(const rmagine::Vector3i &p) const
  {
    int x_offset = overflow(p.x - pos_->x + offset_->x + size_->x, size_->x) * size_->y * size_->z;
    int y_offset = overflow(p.y - pos_->y + offset_->y + size_->y, size_->y) * size_->z;
    int z_offset = overflow(p.z - pos_->z + offset_->z + size_->z, size_->z);
    return x_offset + y_offset + z_offset;
  }
  bool in_bounds(int x, int y, int z) const
  {
    return std::abs(x - pos_->x) <= size_->x / 2 && std::abs(y - pos_->y) <= size_->y / 2 && std::abs(z - pos_->z) <= size_->z / 2;
  }
  bool in_bounds(rmagine::Vector3i p) const { return in_bounds(p.x, p.y, p.z); }
  // device_map.h:116-128: `buffer` is a size_t there, so both sides of the comparison are unsigned
  // (a buffer larger than size/2 wraps and accepts everything, exactly like the reference)
  bool in_bounds_with_buffer_neg(rmagine::Vector3i p, size_t buffer) const
  {
    const size_t ax = (size_t)std::abs(p.x - pos_->x), ay = (size_t)std::abs(p.y - pos_->y), az = (size_t)std::abs(p.z - pos_->z);
    return ax <= ((size_t)(size_->x / 2) - buffer) && ay <= ((size_t)(size_->y / 2) - buffer) && az <= ((size_t)(size_->z / 2) - buffer);
  }
  bool in_bounds_with_buffer_pos(rmagine::Vector3i p, size_t buffer) const
  {
    const size_t ax = (size_t)std::abs(p.x - pos_->x), ay = (size_t)std::abs(p.y - pos_->y), az = (size_t)std::abs(p.z - pos_->z);
    return ax <= ((size_t)(size_->x / 2) + buffer) && ay <= ((size_t)(size_->y / 2) + buffer) && az <= ((size_t)(size_->z / 2) + buffer);
  }
  TSDFEntry &value_unchecked(int x, int y, int z) { return data_[get_index(rmagine::Vector3i(x, y, z))]; }
  const TSDFEntry &value_unchecked(int x, int y, int z) const { return data_[get_index(rmagine::Vector3i(x, y, z))]; }
  TSDFEntry &value_unchecked(const rmagine::Vector3i &p) { return data_[get_index(p)]; }
  const TSDFEntry &value_unchecked(const rmagine::Vector3i &p) const { return data_[get_index(p)]; }

  rmagine::Pointi *size_ = nullptr;
  rmagine::Pointi *offset_ = nullptr;
  TSDFEntry *data_ = nullptr;
  rmagine::Pointi *pos_ = nullptr;
};

// Device copy of one map; owned by a TSDFCuda (device_map_wrapper.h:10-36).
struct DeviceMapMemWrapper
{
  DeviceMapMemWrapper() = default;
  DeviceMapMemWrapper(const DeviceMapMemWrapper &) = delete;
  DeviceMapMemWrapper operator=(const DeviceMapMemWrapper &) = delete;

  void to_device(const DeviceMap &m)
  {
    WS_CHECK(ws_map_upload(map_, which_, &m.size_->x, &m.pos_->x, &m.offset_->x, reinterpret_cast<const uint32_t *>(m.data_)));
  }
  void update_params(const DeviceMap &m) { WS_CHECK(ws_map_set_params(map_, which_, &m.size_->x, &m.pos_->x, &m.offset_->x)); }
  void to_host(const DeviceMap &m) const
  {
    WS_CHECK(ws_map_download(map_, which_, &m.size_->x, &m.pos_->x, &m.offset_->x, reinterpret_cast<uint32_t *>(m.data_)));
  }
  // what kernels of the reference receive; here an opaque handle RegistrationCuda understands
  DeviceMap *dev() const { return reinterpret_cast<DeviceMap *>(map_); }
  // additions: dense world-voxel boxes [lo, hi] (inclusive, x major / z fastest) out of / into the
  // device ring buffer, so a map shift or an export moves only the slabs it needs (reference: whole-map to_host/to_device)
  void extract_box(const rmagine::Pointi &lo, const rmagine::Pointi &hi, std::vector<TSDFEntry> &out) const
  {
    out.resize((size_t)(hi.x - lo.x + 1) * (size_t)(hi.y - lo.y + 1) * (size_t)(hi.z - lo.z + 1));
    WS_CHECK(ws_map_extract_box(map_, which_, &lo.x, &hi.x, reinterpret_cast<uint32_t *>(out.data())));
  }
  void insert_box(const rmagine::Pointi &lo, const rmagine::Pointi &hi, const std::vector<TSDFEntry> &in)
  {
    WS_CHECK(ws_map_insert_box(map_, which_, &lo.x, &hi.x, reinterpret_cast<const uint32_t *>(in.data())));
  }

  ws_map *map_ = nullptr;
  int which_ = WS_MAP_AVG;
};

class TSDFCuda
{
public:
  explicit TSDFCuda(const DeviceMap &existing_map, int tau, int max_weight, int map_resolution)
  {
    WS_CHECK(ws_map_create(detail::context(), &existing_map.size_->x, &existing_map.pos_->x, &existing_map.offset_->x,
                           reinterpret_cast<const uint32_t *>(existing_map.data_), tau, max_weight, map_resolution, &map_));
    avg_map_.map_ = new_map_.map_ = map_;
    avg_map_.which_ = WS_MAP_AVG;
    new_map_.which_ = WS_MAP_NEW;
  }
  ~TSDFCuda() { ws_map_destroy(map_); }
  TSDFCuda(const TSDFCuda &) = delete;
  TSDFCuda &operator=(const TSDFCuda &) = delete;

  void update_tsdf(const std::vector<rmagine::Pointi> &scan_points, const rmagine::Pointi &scanner_pos, const rmagine::Pointi &up)
  {
    int rc = ws_tsdf_update(map_, scan_points.empty() ? nullptr : &scan_points[0].x, scan_points.size(), &scanner_pos.x, &up.x);
    if (rc == WS_ERR_TOO_MANY_POINTS)
    {
      // update_tsdf.cu:146-150: message and return
      fprintf(stderr, "HIP Error: %s:%d - %s\n", __FILE__, __LINE__, ws_last_error());
      return;
    }
    WS_CHECK(rc);
  }
  void update_tsdf(DeviceMap &result, const std::vector<rmagine::Pointi> &scan_points, const rmagine::Pointi &scanner_pos,
                   const rmagine::Pointi &up)
  {
    update_tsdf(scan_points, scanner_pos, up);
    avg_map_.to_host(result);
  }
  void update_tsdf(DeviceMap &result, DeviceMap &latest_map, const std::vector<rmagine::Pointi> &scan_points,
                   const rmagine::Pointi &scanner_pos, const rmagine::Pointi &up)
  {
    update_tsdf(scan_points, scanner_pos, up);
    avg_map_.to_host(result);
    new_map_.to_host(latest_map);
  }
  // addition: scan already resident on the device (e.g. ScanPreprocessor::points_dev()).  Asynchronous like a stream copy: the buffer must
  // stay unchanged until the kernels of this call have read it (anything enqueued later on the context's stream is safe)
  void update_tsdf_dev(const int32_t *xyz_dev, size_t n, const rmagine::Pointi &scanner_pos, const rmagine::Pointi &up)
  {
    int rc = ws_tsdf_update_dev(map_, xyz_dev, n, &scanner_pos.x, &up.x);
    if (rc == WS_ERR_TOO_MANY_POINTS)
    {
      fprintf(stderr, "HIP Error: %s:%d - %s\n", __FILE__, __LINE__, ws_last_error());
      return;
    }
    WS_CHECK(rc);
  }
  DeviceMap *device_map() { return avg_map_.dev(); }
  const DeviceMap *device_map() const { return avg_map_.dev(); }
  const DeviceMapMemWrapper &avg_map() const { return avg_map_; }
  DeviceMapMemWrapper &avg_map() { return avg_map_; }
  const DeviceMapMemWrapper &new_map() const { return new_map_; }
  DeviceMapMemWrapper &new_map() { return new_map_; }
  ws_map *handle() { return map_; }

private:
  ws_map *map_ = nullptr;
  DeviceMapMemWrapper avg_map_;
  DeviceMapMemWrapper new_map_;
};

class RegistrationCuda
{
public:
  explicit RegistrationCuda(const DeviceMap & /*map*/) { WS_CHECK(ws_reg_create(detail::context(), 128 * 1024, &reg_)); }
  ~RegistrationCuda() { ws_reg_destroy(reg_); }
  RegistrationCuda(const RegistrationCuda &) = delete;
  RegistrationCuda &operator=(const RegistrationCuda &) = delete;

  void prepare_registration(const std::vector<rmagine::Pointi> &points)
  {
    WS_CHECK(ws_reg_prepare(reg_, points.empty() ? nullptr : &points[0].x, points.size()));
  }
  void prepare_registration_dev(const int32_t *xyz_dev, size_t n) { WS_CHECK(ws_reg_prepare_dev(reg_, xyz_dev, n)); }
  void perform_registration(const DeviceMap *map_dev, const rmagine::Matrix4x4f *pretransform, rmagine::Matrix6x6l &h,
                            rmagine::Point6l &g, int &e, int &c, int map_resolution)
  {
    int32_t ee = 0, cc = 0;
    WS_CHECK(ws_reg_iterate(reg_, reinterpret_cast<const ws_map *>(map_dev), &pretransform->data[0][0], map_resolution, flags_,
                            reinterpret_cast<int64_t *>(&h.data[0][0]), reinterpret_cast<int64_t *>(&g.data[0]), &ee, &cc));
    e = ee;
    c = cc;
  }
  // the whole loop of TSDFRegistration::register_cloud (tsdf_registration.cpp:28-96) on the device
  rmagine::Matrix4x4f register_cloud(const DeviceMap *map_dev, const rmagine::Matrix4x4f &pretransform, int max_iterations,
                                     float it_weight_gradient, float epsilon, int map_resolution, int *iterations = nullptr)
  {
    rmagine::Matrix4x4f out;
    int32_t it = 0;
    WS_CHECK(ws_register_cloud(reg_, reinterpret_cast<const ws_map *>(map_dev), &pretransform.data[0][0], max_iterations,
                               it_weight_gradient, epsilon, map_resolution, flags_, &out.data[0][0], &it));
    if (iterations) *iterations = it;
    return out;
  }
  void set_flags(uint32_t flags) { flags_ = flags; }

  // ---- point-sharded registration over several GPUs (one process per GPU, the map replicated, every rank has prepared the
  // whole cloud): the ranks' 44 sums meet in mailboxes in each other's HBM, ws_register_cloud_peers (warpsense_hip.h).
  // mailbox(): this rank's mailbox as a 64-byte IPC handle, to be all-gathered by the caller's transport;
  // connect(): map the peers' mailboxes; register_cloud_peers(): all ranks together, false if a peer did not deliver
  // (then: barrier, reset_peers(), barrier, and the RCCL route -- ws_reg_iterate_shard_dev -- for that cloud).
  void peer_mailbox(unsigned char handle[WS_IPC_HANDLE_BYTES]) { WS_CHECK(ws_reg_peer_mailbox(reg_, handle)); }
  void peer_connect(int rank, int world, const unsigned char *handles /* world x WS_IPC_HANDLE_BYTES */, int blocks = 0)
  {
    WS_CHECK(ws_reg_peer_connect(reg_, rank, world, handles, blocks));
  }
  void reset_peers() { WS_CHECK(ws_reg_peer_reset(reg_)); }
  bool register_cloud_peers(const DeviceMap *map_dev, size_t first, size_t count, const rmagine::Matrix4x4f &pretransform, int max_iterations,
                            float it_weight_gradient, float epsilon, int map_resolution, rmagine::Matrix4x4f &out, int *iterations = nullptr)
  {
    int32_t it = 0;
    const int rc = ws_register_cloud_peers(reg_, reinterpret_cast<const ws_map *>(map_dev), first, count, &pretransform.data[0][0], max_iterations,
                                           it_weight_gradient, epsilon, map_resolution, flags_, &out.data[0][0], &it);
    if (rc == WS_ERR_TIMEOUT) return false;
    WS_CHECK(rc);
    if (iterations) *iterations = it;
    return true;
  }

private:
  ws_reg *reg_ = nullptr;
  uint32_t flags_ = WS_REG_ALL_POINTS;
};

// App::preprocess on the device (src/warpsense/app.cpp:119-148): float metres -> distinct int mm voxel-centre points
// transformed by the pose, in first-occurrence order; the result stays on the device until the next call.
class ScanPreprocessor
{
public:
  explicit ScanPreprocessor(size_t max_points = 128 * 1024) { WS_CHECK(ws_scan_create(detail::context(), max_points, &scan_)); }
  ~ScanPreprocessor() { ws_scan_destroy(scan_); }
  ScanPreprocessor(const ScanPreprocessor &) = delete;
  ScanPreprocessor &operator=(const ScanPreprocessor &) = delete;

  size_t preprocess(const float *cloud_xyz, size_t n, size_t stride_floats, const rmagine::Matrix4x4f &pose, int map_resolution)
  {
    WS_CHECK(ws_scan_preprocess(scan_, cloud_xyz, n, stride_floats, &pose.data[0][0], map_resolution, &n_out_));
    return n_out_;
  }
  const int32_t *points_dev() const { return ws_scan_points_dev(scan_); }
  size_t size() const { return n_out_; }
  std::vector<rmagine::Pointi> download() const
  {
    std::vector<rmagine::Pointi> pts(n_out_);
    size_t n = 0;
    WS_CHECK(ws_scan_download(scan_, n_out_ ? &pts[0].x : nullptr, n_out_, &n));
    return pts;
  }

private:
  ws_scan *scan_ = nullptr;
  size_t n_out_ = 0;
};

inline void pause() { WS_CHECK(ws_sync(detail::context())); }
inline void cleanup() { WS_CHECK(ws_device_reset()); }

} // namespace cuda
