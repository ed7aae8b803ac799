// app.hpp — ROS-free C++ twins of the code AROUND the hot path, on top of compat.hpp / mapping.hpp:
//
//   GlobalMap      HDF5GlobalMap            src/map/hdf5_global_map.cpp:4-221   64^3-voxel chunks, 64-chunk LRU, .h5 file
//   LocalMap       HDF5LocalMap             src/map/hdf5_local_map.cpp:5-217    host ring buffer, shift(), write_back()
//   MappingNode    cuda::TSDFMapping::map_shift  tsdf_mapping.cpp:97-136        + device-side shift_map / write_back
//   App            warpsense::App           src/warpsense/app.cpp:30-224        cloud_callback sequencing without ROS
//
// The .h5 file needs libwarpsense_h5.so (include/warpsense_h5.h); define WARPSENSE_HIP_WITH_H5 before including this
// header to compile the file backing in, otherwise GlobalMap keeps every chunk in memory.
#pragma once

#include <algorithm>
#include <array>
#include <chrono>
#include <cmath>
#include <list>
#include <map>
#include <mutex>
#include <set>
#include <stdexcept>
#include <string>
#include <thread>

#include "warpsense_hip/mapping.hpp"
#ifdef WARPSENSE_HIP_WITH_H5
#include "warpsense_h5.h"
#endif

namespace warpsense
{
namespace rm = rmagine;

inline int floor_div(int a, int b) { return a >= 0 ? a / b : -((-a + b - 1) / b); }

// ---------------------------------------------------------------------------------------------------- GlobalMap
class GlobalMap
{
public:
  static constexpr int CHUNK_SIZE = 64;  // hdf5_global_map.h:75
  static constexpr int NUM_CHUNKS = 64;  // hdf5_global_map.h:78
  using Key = std::array<int, 3>;

  GlobalMap(int16_t default_value, int16_t default_weight, const std::string &filename = std::string())
      : default_entry_(default_value, default_weight), filename_(filename)
  {
#ifdef WARPSENSE_HIP_WITH_H5
    if (!filename_.empty() && ws_h5_create(filename_.c_str(), &file_) != 0) throw std::runtime_error(ws_h5_last_error());
#else
    if (!filename_.empty()) throw std::runtime_error("GlobalMap: built without WARPSENSE_HIP_WITH_H5");
#endif
  }
  ~GlobalMap()
  {
#ifdef WARPSENSE_HIP_WITH_H5
    if (file_)
    {
      write_back();
      ws_h5_close(file_);
    }
#endif
  }
  GlobalMap(const GlobalMap &) = delete;
  GlobalMap &operator=(const GlobalMap &) = delete;

  const std::string &filename() const { return filename_; }
  bool has_file() const { return file_backed(); }
  TSDFEntry get_default_tsdf_entry() const { return default_entry_; }

  // activate_chunk — hdf5_global_map.cpp:59-135: cached, else read from the file, else default-filled; the least
  // recently used chunk is written to the file when a 65th is needed (memory-only maps keep everything)
  std::vector<TSDFEntry::RawType> &activate_chunk(const Key &c)
  {
    auto it = chunks_.find(c);
    if (it != chunks_.end())
    {
      if (file_backed()) touch(c);
      return it->second;
    }
    std::vector<TSDFEntry::RawType> data((size_t)CHUNK_SIZE * CHUNK_SIZE * CHUNK_SIZE);
    bool found = false;
#ifdef WARPSENSE_HIP_WITH_H5
    if (file_)
    {
      int32_t ex = 0;
      if (ws_h5_read_chunk(file_, c[0], c[1], c[2], data.data(), &ex) != 0) throw std::runtime_error(ws_h5_last_error());
      found = ex != 0;
    }
#endif
    if (!found) std::fill(data.begin(), data.end(), default_entry_.raw());
    if (file_backed() && (int)chunks_.size() >= NUM_CHUNKS)
    {
      const Key old = lru_.back();
      write_chunk(old, chunks_[old]);
      chunks_.erase(old);
      lru_.pop_back();
    }
    auto &ref = chunks_[c];
    ref = std::move(data);
    if (file_backed()) lru_.push_front(c);
    return ref;
  }
  static int index_from_pos(int x, int y, int z, const Key &c) // :53-57
  {
    return (x - c[0] * CHUNK_SIZE) * CHUNK_SIZE * CHUNK_SIZE + (y - c[1] * CHUNK_SIZE) * CHUNK_SIZE + (z - c[2] * CHUNK_SIZE);
  }
  TSDFEntry get_value(int x, int y, int z)
  {
    const Key c{floor_div(x, CHUNK_SIZE), floor_div(y, CHUNK_SIZE), floor_div(z, CHUNK_SIZE)};
    return TSDFEntry(activate_chunk(c)[index_from_pos(x, y, z, c)]);
  }
  void set_value(int x, int y, int z, const TSDFEntry &v)
  {
    const Key c{floor_div(x, CHUNK_SIZE), floor_div(y, CHUNK_SIZE), floor_div(z, CHUNK_SIZE)};
    activate_chunk(c)[index_from_pos(x, y, z, c)] = v.raw();
  }
  // true if the map holds data for this chunk (in memory or in the file): a chunk never seen is all default
  bool has_chunk(const Key &c)
  {
    std::lock_guard<std::recursive_mutex> g(mutex_);
    return chunks_.count(c) != 0 || in_file_.count(c) != 0;
  }
  // dense world-voxel box [lo, hi] (inclusive, x major / z fastest) <-> chunks.  Thread safe per chunk: the map-shift
  // worker files slabs while the scan thread writes poses (the HDF5 library is used by one thread at a time).
  void save_box(const rm::Pointi &lo, const rm::Pointi &hi, const std::vector<TSDFEntry> &box) { move_box(lo, hi, const_cast<TSDFEntry *>(box.data()), true); }
  void save_box(const rm::Pointi &lo, const rm::Pointi &hi, const TSDFEntry *box) { move_box(lo, hi, const_cast<TSDFEntry *>(box), true); }
  void load_box(const rm::Pointi &lo, const rm::Pointi &hi, std::vector<TSDFEntry> &box)
  {
    box.resize((size_t)(hi.x - lo.x + 1) * (size_t)(hi.y - lo.y + 1) * (size_t)(hi.z - lo.z + 1));
    move_box(lo, hi, box.data(), false);
  }
  void write_back() // :160-176
  {
#ifdef WARPSENSE_HIP_WITH_H5
    std::lock_guard<std::recursive_mutex> g(mutex_);
    if (!file_) return;
    for (auto &kv : chunks_) write_chunk(kv.first, kv.second);
    ws_h5_flush(file_);
#endif
  }
  // write_pose — :178-199 (rotation as Eigen would turn it into a quaternion, every value rounded to 3 decimals)
  std::array<float, 7> write_pose(const rm::Matrix4x4f &pose, float scale)
  {
    std::array<float, 7> v = pose_values(pose, scale);
#ifdef WARPSENSE_HIP_WITH_H5
    std::lock_guard<std::recursive_mutex> g(mutex_);
    if (file_ && ws_h5_write_pose(file_, v.data()) != 0) throw std::runtime_error(ws_h5_last_error());
#endif
    return v;
  }
  void write_meta(int tau, const rm::Pointi &size, float max_distance, int resolution, int max_weight) // :207-221
  {
#ifdef WARPSENSE_HIP_WITH_H5
    if (file_ && ws_h5_write_meta(file_, tau, &size.x, max_distance, resolution, max_weight) != 0) throw std::runtime_error(ws_h5_last_error());
#else
    (void)tau; (void)size; (void)max_distance; (void)resolution; (void)max_weight;
#endif
  }
  static std::array<float, 7> pose_values(const rm::Matrix4x4f &pose, float scale)
  {
    std::array<float, 7> v{};
    for (int k = 0; k < 3; ++k) v[k] = (float)(std::round(((double)pose.at(k, 3) / scale) * 1000.0f) / 1000.0f);
    float q[4]; // x y z w, Eigen::Quaternionf(Matrix3f)
    const auto m = [&](int i, int j) { return pose.at(i, j); };
    float t = m(0, 0) + m(1, 1) + m(2, 2);
    if (t > 0.f)
    {
      t = std::sqrt(t + 1.0f);
      q[3] = 0.5f * t;
      t = 0.5f / t;
      q[0] = (m(2, 1) - m(1, 2)) * t;
      q[1] = (m(0, 2) - m(2, 0)) * t;
      q[2] = (m(1, 0) - m(0, 1)) * t;
    }
    else
    {
      int i = 0;
      if (m(1, 1) > m(0, 0)) i = 1;
      if (m(2, 2) > m(i, i)) i = 2;
      const int j = (i + 1) % 3, k = (j + 1) % 3;
      t = std::sqrt(m(i, i) - m(j, j) - m(k, k) + 1.0f);
      q[i] = 0.5f * t;
      t = 0.5f / t;
      q[3] = (m(k, j) - m(j, k)) * t;
      q[j] = (m(j, i) + m(i, j)) * t;
      q[k] = (m(k, i) + m(i, k)) * t;
    }
    for (int n = 0; n < 4; ++n) v[3 + n] = std::round(q[n] * 1000.0f) / 1000.0f;
    return v;
  }
  size_t active_chunks() const { return chunks_.size(); }

private:
  bool file_backed() const
  {
#ifdef WARPSENSE_HIP_WITH_H5
    return file_ != nullptr;
#else
    return false;
#endif
  }
  void touch(const Key &c)
  {
    lru_.remove(c);
    lru_.push_front(c);
  }
  void write_chunk(const Key &c, const std::vector<TSDFEntry::RawType> &data)
  {
#ifdef WARPSENSE_HIP_WITH_H5
    if (ws_h5_write_chunk(file_, c[0], c[1], c[2], data.data()) != 0) throw std::runtime_error(ws_h5_last_error());
    in_file_.insert(c);
#else
    (void)c; (void)data;
#endif
  }
  void move_box(const rm::Pointi &lo, const rm::Pointi &hi, TSDFEntry *box, bool save)
  {
    const int cs = CHUNK_SIZE;
    const size_t ey = (size_t)(hi.y - lo.y + 1), ez = (size_t)(hi.z - lo.z + 1);
    for (int cx = floor_div(lo.x, cs); cx <= floor_div(hi.x, cs); ++cx)
      for (int cy = floor_div(lo.y, cs); cy <= floor_div(hi.y, cs); ++cy)
        for (int cz = floor_div(lo.z, cs); cz <= floor_div(hi.z, cs); ++cz)
        {
          const Key c{cx, cy, cz};
          std::lock_guard<std::recursive_mutex> g(mutex_);
          auto &chunk = activate_chunk(c);
          const int ax = std::max(lo.x, cx * cs), bx = std::min(hi.x, cx * cs + cs - 1);
          const int ay = std::max(lo.y, cy * cs), by = std::min(hi.y, cy * cs + cs - 1);
          const int az = std::max(lo.z, cz * cs), bz = std::min(hi.z, cz * cs + cs - 1);
          for (int x = ax; x <= bx; ++x)
            for (int y = ay; y <= by; ++y)
            {
              TSDFEntry::RawType *cp = &chunk[index_from_pos(x, y, az, c)];
              TSDFEntry *bp = &box[((size_t)(x - lo.x) * ey + (size_t)(y - lo.y)) * ez + (size_t)(az - lo.z)];
              for (int z = az; z <= bz; ++z, ++cp, ++bp)
              {
                if (save)
                  *cp = bp->raw();
                else
                  bp->raw(*cp);
              }
            }
        }
  }

  TSDFEntry default_entry_;
  std::string filename_;
  std::map<Key, std::vector<TSDFEntry::RawType>> chunks_;
  std::list<Key> lru_;
  std::set<Key> in_file_; // chunks written to the file by this object
  std::recursive_mutex mutex_;
#ifdef WARPSENSE_HIP_WITH_H5
  ws_h5 *file_ = nullptr;
#else
  void *file_ = nullptr;
#endif
};

// ---------------------------------------------------------------------------------------------------- LocalMap
// The in-memory state of HDF5LocalMap (hdf5_local_map.cpp:5-20): odd sizes, offset = size / 2, default-filled.
class LocalMap
{
public:
  LocalMap(int sx, int sy, int sz, GlobalMap &map)
      : size_(sx % 2 ? sx : sx + 1, sy % 2 ? sy : sy + 1, sz % 2 ? sz : sz + 1), pos_(0, 0, 0), offset_(size_.x / 2, size_.y / 2, size_.z / 2),
        map_(map), data_((size_t)size_.x * size_.y * size_.z, map.get_default_tsdf_entry())
  {
  }
  rm::Pointi &get_size() { return size_; }
  rm::Pointi &get_pos() { return pos_; }
  rm::Pointi &get_offset() { return offset_; }
  std::vector<TSDFEntry> &data() { return data_; }
  GlobalMap &global_map() { return map_; }
  bool in_bounds(int x, int y, int z) const
  {
    return std::abs(x - pos_.x) <= size_.x / 2 && std::abs(y - pos_.y) <= size_.y / 2 && std::abs(z - pos_.z) <= size_.z / 2;
  }
  TSDFEntry &value(int x, int y, int z)
  {
    if (!in_bounds(x, y, z)) throw std::out_of_range("Index out of bounds"); // hdf5_local_map.h:172-181
    const size_t xi = (size_t)((x - pos_.x + offset_.x + size_.x) % size_.x), yi = (size_t)((y - pos_.y + offset_.y + size_.y) % size_.y),
                 zi = (size_t)((z - pos_.z + offset_.z + size_.z) % size_.z);
    return data_[(xi * size_.y + yi) * size_.z + zi];
  }

private:
  rm::Pointi size_, pos_, offset_;
  GlobalMap &map_;
  std::vector<TSDFEntry> data_;
};

// ---------------------------------------------------------------------------------------------------- MappingNode
// cuda::TSDFRegistration plus the map-shift / export side of cuda::TSDFMapping, with the window moved ON THE DEVICE.
class MappingNode
{
public:
  MappingNode(const cuda::HotPathParams &params, LocalMap &local_map)
      : params_(params), local_map_(local_map), view_(&local_map.get_size(), &local_map.get_offset(), local_map.data().data(), &local_map.get_pos()),
        gpu_(params, view_)
  {
  }
  cuda::TSDFRegistration &gpu() { return gpu_; }

  // TSDFMapping::map_shift body (tsdf_mapping.cpp:109-126) == HDF5LocalMap::shift (hdf5_local_map.cpp:53-118) per axis:
  // save the slab that leaves, move pos/offset, load the slab that enters -- each slab packed / unpacked by the GPU.
  void shift_map(const rm::Pointi &new_pos)
  {
    wait_shift();
    auto &avg = gpu_.tsdf().avg_map();
    auto &fresh = gpu_.tsdf().new_map();
    rm::Pointi &size = local_map_.get_size(), &pos = local_map_.get_pos(), &off = local_map_.get_offset();
    int *sz = &size.x, *ps = &pos.x, *of = &off.x;
    const int np[3] = {new_pos.x, new_pos.y, new_pos.z};
    std::vector<TSDFEntry> slab;
    for (int axis = 0; axis < 3; ++axis)
    {
      const int d = np[axis] - ps[axis];
      if (d == 0) continue;
      if (std::abs(d) > sz[axis]) throw std::out_of_range("shift_map: further than one window");
      int lo[3], hi[3];
      for (int k = 0; k < 3; ++k)
      {
        lo[k] = ps[k] - sz[k] / 2;
        hi[k] = ps[k] + sz[k] / 2;
      }
      if (d > 0)
        hi[axis] = lo[axis] + d - 1;
      else
        lo[axis] = hi[axis] + d + 1;
      avg.extract_box(rm::Pointi(lo[0], lo[1], lo[2]), rm::Pointi(hi[0], hi[1], hi[2]), slab);
      local_map_.global_map().save_box(rm::Pointi(lo[0], lo[1], lo[2]), rm::Pointi(hi[0], hi[1], hi[2]), slab);
      ps[axis] += d;
      of[axis] = (of[axis] + d + sz[axis]) % sz[axis];
      avg.update_params(view_);
      fresh.update_params(view_); // new_map is (tau, 0) everywhere: only its window moves
      for (int k = 0; k < 3; ++k)
      {
        lo[k] = ps[k] - sz[k] / 2;
        hi[k] = ps[k] + sz[k] / 2;
      }
      if (d > 0)
        lo[axis] = hi[axis] - (d - 1);
      else
        hi[axis] = lo[axis] - d - 1;
      local_map_.global_map().load_box(rm::Pointi(lo[0], lo[1], lo[2]), rm::Pointi(hi[0], hi[1], hi[2]), slab);
      avg.insert_box(rm::Pointi(lo[0], lo[1], lo[2]), rm::Pointi(hi[0], hi[1], hi[2]), slab);
    }
  }

  // The same shift the way the reference runs it — off the scan path (its own thread, tsdf_mapping.cpp:97-136): the
  // window moves by device kernels inside this call (ws_shift_begin: pack the leaving slabs, move pos/offset, fill the
  // entering slabs with the default entry; nothing waits), chunks the global map already holds for the entering slabs are
  // uploaded (revisits only), and a worker thread files the leaving slabs into the global map once their copy to pinned
  // memory has landed.  Same maps as shift_map().
  // staging for asynchronous shifts of up to `shift_voxels` per axis (plus slack), allocated now instead of inside the first
  // shift: a pinned allocation of that size takes tens of milliseconds (56 ms of the first shifting scan of the 1025^3 stream)
  void reserve_shift(int shift_voxels)
  {
    const rm::Pointi &size = local_map_.get_size();
    const uint64_t d = (uint64_t)(shift_voxels > 0 ? shift_voxels : 0) + 8u;
    WS_CHECK(ws_shift_reserve(gpu_.tsdf().handle(), d * ((uint64_t)size.x * size.y + (uint64_t)size.y * size.z + (uint64_t)size.x * size.z)));
  }
  void shift_map_async(const rm::Pointi &new_pos)
  {
    wait_shift();
    auto &avg = gpu_.tsdf().avg_map();
    rm::Pointi &size = local_map_.get_size(), &pos = local_map_.get_pos(), &off = local_map_.get_offset();
    int *sz = &size.x, *ps = &pos.x, *of = &off.x;
    const int np[3] = {new_pos.x, new_pos.y, new_pos.z};
    ws_shift *ticket = nullptr;
    WS_CHECK(ws_shift_begin(gpu_.tsdf().handle(), np, local_map_.global_map().get_default_tsdf_entry().raw(), &ticket));
    // Between ws_shift_begin and the start of the worker, loading the entering boxes can throw: the window has moved on the
    // device by then, so the leaving slabs are still filed (synchronously, by this guard) and the ticket is closed -- an
    // open ticket would make every later ws_shift_begin fail.
    GlobalMap *gm = &local_map_.global_map();
    std::string *err = &shift_error_;
    struct Owner
    {
      ws_shift *t;
      int n;
      GlobalMap *gm;
      std::string *err;
      bool armed;
      ~Owner()
      {
        if (armed) file_leaving_slabs(t, n, gm, err);
      }
    } owner{ticket, ws_shift_count(ticket), gm, err, true};
    for (int axis = 0; axis < 3; ++axis)
    {
      const int d = np[axis] - ps[axis];
      ps[axis] += d;
      of[axis] = ((of[axis] + d) % sz[axis] + sz[axis]) % sz[axis];
    }
    const int n = ws_shift_count(ticket), cs = GlobalMap::CHUNK_SIZE;
    std::vector<TSDFEntry> part;
    for (int i = 0; i < n; ++i)
    {
      int lo[3], hi[3];
      WS_CHECK(ws_shift_entering(ticket, i, lo, hi));
      for (int k = 0; k < 3; ++k) // only what is still inside the final window
      {
        lo[k] = std::max(lo[k], ps[k] - sz[k] / 2);
        hi[k] = std::min(hi[k], ps[k] + sz[k] / 2);
      }
      if (lo[0] > hi[0] || lo[1] > hi[1] || lo[2] > hi[2]) continue;
      for (int cx = floor_div(lo[0], cs); cx <= floor_div(hi[0], cs); ++cx)
        for (int cy = floor_div(lo[1], cs); cy <= floor_div(hi[1], cs); ++cy)
          for (int cz = floor_div(lo[2], cs); cz <= floor_div(hi[2], cs); ++cz)
          {
            if (!local_map_.global_map().has_chunk(GlobalMap::Key{cx, cy, cz})) continue;
            const rm::Pointi a(std::max(lo[0], cx * cs), std::max(lo[1], cy * cs), std::max(lo[2], cz * cs)),
                b(std::min(hi[0], cx * cs + cs - 1), std::min(hi[1], cy * cs + cs - 1), std::min(hi[2], cz * cs + cs - 1));
            local_map_.global_map().load_box(a, b, part);
            avg.insert_box(a, b, part);
          }
    }
    shift_error_.clear();
    owner.armed = false; // from here on the worker owns the ticket
    shift_worker_ = std::thread(file_leaving_slabs, ticket, n, gm, err);
  }
  // The leaving slabs of an asynchronous shift -> the global map; closes the ticket whatever happens (an open ticket makes
  // the next ws_shift_begin fail), and a failure -- a HIP error while waiting for the slabs, an exception out of the global
  // map -- is kept for wait_shift() instead of being swallowed or ending the process through std::terminate.
  static void file_leaving_slabs(ws_shift *ticket, int n, GlobalMap *gm, std::string *err)
  {
      struct Closer
      {
        ws_shift *t;
        ~Closer() { ws_shift_end(t); }
      } closer{ticket};
      try
      {
      if (ws_shift_wait(ticket) != WS_OK)
      {
        *err = std::string("ws_shift_wait: ") + ws_last_error();
        return;
      }
      std::vector<TSDFEntry> own;
      for (int i = 0; i < n; ++i)
      {
        int lo[3], hi[3];
        const uint32_t *data = nullptr;
        if (ws_shift_slab(ticket, i, lo, hi, &data) != WS_OK) continue;
        TSDFEntry *box = reinterpret_cast<TSDFEntry *>(const_cast<uint32_t *>(data));
        const size_t ey = (size_t)(hi[1] - lo[1] + 1), ez = (size_t)(hi[2] - lo[2] + 1);
        // a corner that entered with an earlier axis of this shift and leaves with this one was packed as default fill:
        // put the global map's own data there, so that saving the slab changes nothing for it
        for (int j = 0; j < i; ++j)
        {
          int elo[3], ehi[3];
          if (ws_shift_entering(ticket, j, elo, ehi) != WS_OK) continue;
          int a[3], b[3];
          bool any = true;
          for (int k = 0; k < 3; ++k)
          {
            a[k] = std::max(lo[k], elo[k]);
            b[k] = std::min(hi[k], ehi[k]);
            any = any && a[k] <= b[k];
          }
          if (!any) continue;
          gm->load_box(rm::Pointi(a[0], a[1], a[2]), rm::Pointi(b[0], b[1], b[2]), own);
          const size_t oy = (size_t)(b[1] - a[1] + 1), oz = (size_t)(b[2] - a[2] + 1);
          for (int x = a[0]; x <= b[0]; ++x)
            for (int y = a[1]; y <= b[1]; ++y)
              for (int z = a[2]; z <= b[2]; ++z)
                box[((size_t)(x - lo[0]) * ey + (size_t)(y - lo[1])) * ez + (size_t)(z - lo[2])] =
                    own[((size_t)(x - a[0]) * oy + (size_t)(y - a[1])) * oz + (size_t)(z - a[2])];
        }
        gm->save_box(rm::Pointi(lo[0], lo[1], lo[2]), rm::Pointi(hi[0], hi[1], hi[2]), box);
      }
      }
      catch (const std::exception &e)
      {
        *err = std::string("filing the leaving slabs: ") + e.what();
      }
      catch (...)
      {
        *err = "filing the leaving slabs: unknown exception";
      }
  }
  // joins the worker of the last asynchronous shift; throws if its slabs did not reach the global map
  void wait_shift()
  {
    if (shift_worker_.joinable()) shift_worker_.join();
    if (!shift_error_.empty())
    {
      const std::string msg = "asynchronous map shift failed: " + shift_error_;
      shift_error_.clear();
      throw std::runtime_error(msg);
    }
  }
  ~MappingNode()
  {
    if (shift_worker_.joinable()) shift_worker_.join(); // (a destructor does not throw; the error was the caller's to collect)
  }

  // HDF5LocalMap::write_back + HDF5GlobalMap::write_back (hdf5_local_map.cpp:210-217, app.cpp:215-221) from the device map
  void write_back()
  {
    wait_shift();
    auto &avg = gpu_.tsdf().avg_map();
    const rm::Pointi &size = local_map_.get_size(), &pos = local_map_.get_pos();
    const int cs = GlobalMap::CHUNK_SIZE;
    const rm::Pointi lo(pos.x - size.x / 2, pos.y - size.y / 2, pos.z - size.z / 2), hi(pos.x + size.x / 2, pos.y + size.y / 2, pos.z + size.z / 2);
    std::vector<TSDFEntry> slab;
    for (int cx = floor_div(lo.x, cs); cx <= floor_div(hi.x, cs); ++cx)
    {
      const rm::Pointi a(std::max(lo.x, cx * cs), lo.y, lo.z), b(std::min(hi.x, cx * cs + cs - 1), hi.y, hi.z);
      avg.extract_box(a, b, slab);
      local_map_.global_map().save_box(a, b, slab);
    }
    local_map_.global_map().write_back();
  }
  // the reference's route, kept for comparison: whole window to the host array of the local map
  void download() { gpu_.tsdf().avg_map().to_host(view_); }

private:
  cuda::HotPathParams params_;
  LocalMap &local_map_;
  cuda::DeviceMap view_;
  cuda::TSDFRegistration gpu_;
  std::thread shift_worker_;
  std::string shift_error_; // written by the worker, read after join()
};

// ---------------------------------------------------------------------------------------------------- App
struct AppParams
{
  cuda::HotPathParams hot;
  float max_distance = 1.0f; // map/max_distance (m)
  float shift = 10.0f;       // map/shift (m)
  int map_size[3] = {513, 513, 513}; // voxels
  int initial_weight = 0;
  bool async_shift = false; // MappingNode::shift_map_async: the map shift off the scan path
};

// wall-clock microseconds of the stages of one cloud_callback -- the reference's RuntimeEvaluator forms "preprocess", "tsdf",
// "registration", "total" (app.cpp:68-111), plus the map shift's turn
struct StageTimes
{
  double preprocess_us = 0, tsdf_us = 0, registration_us = 0, shift_us = 0, total_us = 0;
};

class App
{
public:
  App(const AppParams &p, const std::string &h5_filename = std::string(), size_t max_points = 128 * 1024)
      : params_(p), global_map_((int16_t)p.hot.tau, (int16_t)p.initial_weight, h5_filename),
        local_map_(p.map_size[0], p.map_size[1], p.map_size[2], global_map_), node_(p.hot, local_map_), pre_(max_points)
  {
    pose_.setIdentity();
    last_tsdf_pose_.setIdentity();
    last_shift_pose_.setIdentity();
    if (p.async_shift) node_.reserve_shift((int)std::ceil(p.shift * 1000.f / (float)p.hot.map_resolution));
    if (global_map_.has_file()) global_map_.write_meta(p.hot.tau, local_map_.get_size(), p.max_distance, p.hot.map_resolution, p.hot.max_weight);
  }

  // App::cloud_callback — app.cpp:65-117; `pretransform` stands for imu_acc_.acc_transform(stamp) (identity: no IMU).
  // The map-shift thread's turn (tsdf_mapping.cpp:104-127) runs synchronously at the end.
  const rm::Matrix4x4f &cloud_callback(const float *cloud_xyz, size_t n, size_t stride_floats, const rm::Matrix4x4f *pretransform = nullptr)
  {
    using clk = std::chrono::steady_clock;
    auto us = [](clk::time_point a, clk::time_point b) { return std::chrono::duration<double, std::micro>(b - a).count(); };
    const clk::time_point t0 = clk::now();
    const size_t n_pts = pre_.preprocess(cloud_xyz, n, stride_floats, pose_, params_.hot.map_resolution); // App::preprocess :119-148
    const clk::time_point t1 = clk::now();
    if (!initialized_ || distance_m(last_tsdf_pose_, pose_) > 0.3f || shifted_)
    {
      initialized_ = true;
      last_tsdf_pose_ = pose_;
      rm::Pointi pos_rm, up_rm;
      node_.gpu().convert_pose_to_gpu(pose_, pos_rm, up_rm);
      node_.gpu().tsdf().update_tsdf_dev(pre_.points_dev(), n_pts, pos_rm, up_rm);
      shifted_ = false;
      ++n_updates_;
    }
    const clk::time_point t2 = clk::now();
    rm::Matrix4x4f pre;
    if (pretransform)
      pre = *pretransform;
    else
      pre.setIdentity();
    node_.gpu().registration().prepare_registration_dev(pre_.points_dev(), n_pts);
    const rm::Matrix4x4f transform =
        node_.gpu().registration().register_cloud(node_.gpu().tsdf().device_map(), pre, params_.hot.max_iterations, params_.hot.it_weight_gradient,
                                                  params_.hot.epsilon, params_.hot.map_resolution, &last_iterations_);
    update_pose_estimate(transform);
    if (global_map_.has_file()) global_map_.write_pose(pose_, 1000.f);
    const clk::time_point t3 = clk::now();
    map_shift();
    const clk::time_point t4 = clk::now();
    last_points_ = n_pts;
    times_.preprocess_us = us(t0, t1);
    times_.tsdf_us = us(t1, t2);
    times_.registration_us = us(t2, t3);
    times_.shift_us = us(t3, t4);
    times_.total_us = us(t0, t4);
    return pose_;
  }
  void update_pose_estimate(const rm::Matrix4x4f &t) // app.cpp:172-176
  {
    float R[3][3];
    for (int i = 0; i < 3; ++i)
      for (int j = 0; j < 3; ++j)
      {
        float acc = 0.f;
        for (int k = 0; k < 3; ++k) acc += t.at(i, k) * pose_.at(k, j);
        R[i][j] = acc;
      }
    for (int i = 0; i < 3; ++i)
    {
      for (int j = 0; j < 3; ++j) pose_.at(i, j) = R[i][j];
      pose_.at(i, 3) += t.at(i, 3);
    }
  }
  void map_shift()
  {
    if (distance_m(last_shift_pose_, pose_) >= params_.shift)
    {
      last_shift_pose_ = pose_;
      const int res = params_.hot.map_resolution;
      const rm::Pointi target((int)std::floor(pose_.at(0, 3) / (float)res), (int)std::floor(pose_.at(1, 3) / (float)res),
                              (int)std::floor(pose_.at(2, 3) / (float)res)); // to_map, util/util.h:52-56
      if (params_.async_shift)
        node_.shift_map_async(target);
      else
        node_.shift_map(target);
      shifted_ = true;
      ++n_shifts_;
    }
  }
  void terminate() // app.cpp:192-224
  {
    if (initialized_) node_.write_back();
  }

  const rm::Matrix4x4f &pose() const { return pose_; }
  int last_iterations() const { return last_iterations_; }
  const StageTimes &last_times() const { return times_; }
  size_t last_points() const { return last_points_; }
  int n_updates() const { return n_updates_; }
  int n_shifts() const { return n_shifts_; }
  MappingNode &node() { return node_; }
  LocalMap &local_map() { return local_map_; }
  GlobalMap &global_map() { return global_map_; }

private:
  static float distance_m(const rm::Matrix4x4f &a, const rm::Matrix4x4f &b)
  {
    float s = 0.f;
    for (int k = 0; k < 3; ++k)
    {
      const float d = a.at(k, 3) / 1000.f - b.at(k, 3) / 1000.f;
      s += d * d;
    }
    return std::sqrt(s);
  }

  AppParams params_;
  GlobalMap global_map_;
  LocalMap local_map_;
  MappingNode node_;
  cuda::ScanPreprocessor pre_;
  rm::Matrix4x4f pose_, last_tsdf_pose_, last_shift_pose_;
  bool initialized_ = false, shifted_ = false;
  int last_iterations_ = 0, n_updates_ = 0, n_shifts_ = 0;
  size_t last_points_ = 0;
  StageTimes times_;
};

} // namespace warpsense
