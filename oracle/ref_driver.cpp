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
