// ref_driver.cpp — thin extern "C" driver around the REFERENCE'S OWN headers (test infrastructure).
//
// Built only where /root/reference exists (this container), by oracle/Makefile, into oracle/_ref/.
// It compiles the reference headers where they lie (-I/root/reference/include, -D__NVCC__ is the
// switch device_map.h:6-8 itself offers to drop its HDF5 include); nothing from the reference is
// copied and no stand-in header is involved.  The .cu kernels themselves need nvcc and are NOT built.
//
// What this pins in oracle/ws_oracle.c:
//   cuda::DeviceMap::{get_index,in_bounds,in_bounds_with_buffer_pos/neg}  include/warpsense/cuda/device_map.h:93-128
//   rmagine::Vector3<int|long>::{l2norm,cross}, operators                 include/warpsense/math/vector3.h
//   TSDFEntry bit layout                                                  include/map/tsdf.h:16-46
//   Matrix4x4 / Matrix6x6 storage order                                   include/warpsense/math/matrix{4x4,6x6}.h
//   cu_avg_tsdf_krnl's per-voxel body through TSDFEntry's accessors       src/warpsense/cuda/update_tsdf.cu:19-41
//   calc_jacobis_krnl's lookups, gradient rule and cross product          src/warpsense/cuda/registration.cu:217-253
#include <algorithm>
#include <cstdint>
#include <cstring>

#include "map/tsdf.h"
#include "warpsense/consts.h"
#include "warpsense/cuda/device_map.h"
#include "warpsense/math/math.h"

namespace rm = rmagine;

struct RefMap
{
  rm::Pointi size, pos, offset;
  cuda::DeviceMap map;
  RefMap(const int32_t *s, const int32_t *p, const int32_t *o, uint32_t *data)
      : size(s[0], s[1], s[2]), pos(p[0], p[1], p[2]), offset(o[0], o[1], o[2]),
        map(&size, &offset, reinterpret_cast<TSDFEntry *>(data), &pos)
  {
  }
};

extern "C" {

void *ref_map_create(const int32_t *size, const int32_t *pos, const int32_t *offset, uint32_t *data)
{
  return new RefMap(size, pos, offset, data);
}
void ref_map_destroy(void *m) { delete static_cast<RefMap *>(m); }

int32_t ref_get_index(void *m, int32_t x, int32_t y, int32_t z)
{
  return static_cast<RefMap *>(m)->map.get_index(rm::Vector3i(x, y, z));
}
int ref_in_bounds(void *m, int32_t x, int32_t y, int32_t z) { return static_cast<RefMap *>(m)->map.in_bounds(x, y, z); }
int ref_in_bounds_pos(void *m, int32_t x, int32_t y, int32_t z, int32_t b)
{
  return static_cast<RefMap *>(m)->map.in_bounds_with_buffer_pos(rm::Vector3i(x, y, z), b);
}
int ref_in_bounds_neg(void *m, int32_t x, int32_t y, int32_t z, int32_t b)
{
  return static_cast<RefMap *>(m)->map.in_bounds_with_buffer_neg(rm::Vector3i(x, y, z), b);
}
uint32_t ref_value_unchecked_raw(void *m, int32_t x, int32_t y, int32_t z)
{
  return static_cast<RefMap *>(m)->map.value_unchecked(x, y, z).raw();
}
void ref_write_unchecked(void *m, int32_t x, int32_t y, int32_t z, int16_t v, int16_t w)
{
  static_cast<RefMap *>(m)->map.value_unchecked(x, y, z) = TSDFEntry(v, w);
}

uint32_t ref_pack(int16_t v, int16_t w) { return TSDFEntry(v, w).raw(); }
int16_t ref_entry_value(uint32_t raw) { return TSDFEntry(raw).value(); }
int16_t ref_entry_weight(uint32_t raw) { return TSDFEntry(raw).weight(); }
int ref_sizeof_entry() { return (int)sizeof(TSDFEntry); }
int ref_sizeof_long() { return (int)sizeof(long); }

int32_t ref_l2norm_i(int32_t x, int32_t y, int32_t z) { return rm::Pointi(x, y, z).l2norm(); }
int64_t ref_l2norm_l(int64_t x, int64_t y, int64_t z) { return rm::Pointl(x, y, z).l2norm(); }
void ref_cross_i(const int32_t *a, const int32_t *b, int32_t *out)
{
  auto c = rm::Pointi(a[0], a[1], a[2]).cross(rm::Pointi(b[0], b[1], b[2]));
  out[0] = c.x; out[1] = c.y; out[2] = c.z;
}

// The ray set-up arithmetic of update_tsdf.cu:57-63 written with the reference's own vector operators
// (only the expression shapes are restated; all arithmetic runs through vector3.h).
int ref_ray_setup(const int32_t *point, const int32_t *pos_mm, const int32_t *up, int32_t *distance_out, int64_t *interp_out)
{
  rm::Pointi p(point[0], point[1], point[2]), pos(pos_mm[0], pos_mm[1], pos_mm[2]), upv(up[0], up[1], up[2]);
  rm::Pointi dir = p - pos;
  int distance = dir.l2norm();
  *distance_out = distance;
  if (distance == 0) return -1;
  rm::Pointl nd = (dir.cast<long>() * (long)MATRIX_RESOLUTION) / (long)distance;
  rm::Pointl iv = nd.cross(nd.cross(upv.cast<long>()) / (long)MATRIX_RESOLUTION);
  long inorm = iv.l2norm();
  if (inorm == 0) return -2;
  iv = (iv * (long)MATRIX_RESOLUTION) / inorm;
  interp_out[0] = iv.x; interp_out[1] = iv.y; interp_out[2] = iv.z;
  return 0;
}

// The per-voxel body of cu_avg_tsdf_krnl (update_tsdf.cu:19-41) on ONE pair of entries, written with the reference's own
// TSDFEntry accessors: value() / weight() getters, the ValueType / WeightType conversions of the setters.  Only the
// expression shapes are restated (`min` of the kernel = std::min<int> here); the int16 truncation of the average and the
// promotion rules all come from tsdf.h.  Returns the updated existing entry; *fresh_io is reset like the kernel does.
uint32_t ref_integrate_entry(uint32_t existing_raw, uint32_t *fresh_io, int max_weight, int tau)
{
  TSDFEntry existing_entry(existing_raw), new_entry(*fresh_io);
  if (new_entry.weight() > 0 && existing_entry.weight() > 0)
  {
    existing_entry.value((existing_entry.value() * existing_entry.weight() + new_entry.value() * new_entry.weight()) / (existing_entry.weight() + new_entry.weight()));
    existing_entry.weight(std::min<int>(max_weight, existing_entry.weight() + new_entry.weight()));
  }
  else if (new_entry.weight() != 0 && existing_entry.weight() <= 0)
  {
    existing_entry.value(new_entry.value());
    existing_entry.weight(new_entry.weight());
  }
  new_entry.value(tau);
  new_entry.weight(0);
  *fresh_io = new_entry.raw();
  return existing_entry.raw();
}

// The map-lookup half of calc_jacobis_krnl (registration.cu:217-253) for one transformed point: `buf` = the point's voxel,
// `point` = the point minus the centre, both as the kernel has them after cu_transform_point.  Bounds test, the seven
// lookups and the cross product run through cuda::DeviceMap and rmagine::Vector3 of the reference; the gradient rule is
// restated.  Returns the mask; jacobi_out = (cross, gradient) widened to long, *value_out = the voxel's value.
int ref_jacobi(void *m, const int32_t *buf_in, const int32_t *point_in, int64_t *jacobi_out, int16_t *value_out)
{
  const cuda::DeviceMap *map = &static_cast<RefMap *>(m)->map;
  rm::Pointi buf(buf_in[0], buf_in[1], buf_in[2]), point(point_in[0], point_in[1], point_in[2]);
  if (!map->in_bounds_with_buffer_neg(buf, 1)) return 0;
  const auto &current = map->value_unchecked(buf);
  if (current.weight() == 0) return 0;
  const auto &x_next = map->value_unchecked(buf.x + 1, buf.y, buf.z);
  const auto &x_last = map->value_unchecked(buf.x - 1, buf.y, buf.z);
  const auto &y_next = map->value_unchecked(buf.x, buf.y + 1, buf.z);
  const auto &y_last = map->value_unchecked(buf.x, buf.y - 1, buf.z);
  const auto &z_next = map->value_unchecked(buf.x, buf.y, buf.z + 1);
  const auto &z_last = map->value_unchecked(buf.x, buf.y, buf.z - 1);
  rm::Pointi gradient;
  if (x_next.weight() != 0 && x_last.weight() != 0 && !((x_next.value() > 0 && x_last.value() < 0) || (x_next.value() < 0 && x_last.value() > 0)))
    gradient.x = (x_next.value() - x_last.value()) / 2;
  if (y_next.weight() != 0 && y_last.weight() != 0 && !((y_next.value() > 0 && y_last.value() < 0) || (y_next.value() < 0 && y_last.value() > 0)))
    gradient.y = (y_next.value() - y_last.value()) / 2;
  if (z_next.weight() != 0 && z_last.weight() != 0 && !((z_next.value() > 0 && z_last.value() < 0) || (z_next.value() < 0 && z_last.value() > 0)))
    gradient.z = (z_next.value() - z_last.value()) / 2;
  auto cross = point.cross(gradient);
  rm::Point6l jacobi((long)cross.x, (long)cross.y, (long)cross.z, (long)gradient.x, (long)gradient.y, (long)gradient.z);
  std::memcpy(jacobi_out, &jacobi, 6 * sizeof(int64_t));
  *value_out = current.value();
  return 1;
}

// Matrix storage order probes: write at(i,j)=10*i+j, return the flat arrays
void ref_matrix4_layout(float *flat16)
{
  rm::Matrix4x4f m;
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) m.at(i, j) = (float)(10 * i + j);
  std::memcpy(flat16, &m, sizeof(float) * 16);
}
void ref_matrix6_layout(int64_t *flat36)
{
  rm::Matrix6x6l m;
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 6; ++j) m.at(i, j) = 10 * i + j;
  std::memcpy(flat36, &m, sizeof(int64_t) * 36);
}
int ref_consts(int which) { return which == 0 ? MATRIX_RESOLUTION : WEIGHT_RESOLUTION; }

} // extern "C"
