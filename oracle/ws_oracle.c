/*
 * ws_oracle.c — CPU ORACLE (test infrastructure, NOT product code).  See ws_oracle.h.
 *
 * Every function cites the reference file:line (under /root/reference) whose
 * behaviour it restates.  Integer arithmetic mirrors the reference's C++ types:
 * int32 operations wrap (the CUDA hardware behaviour of the reference's `int`
 * expressions), `long` is int64, divisions truncate toward zero, float->int
 * conversions truncate.
 *
 * Documented deviations (reference has undefined behaviour there):
 *   - rays with distance == 0 or interpolation_norm == 0 are skipped
 *     (the CUDA kernel divides by zero; the CPU path has these guards,
 *     src/cpu/update_tsdf.cpp:593,602),
 *   - c == 0 in the Gauss-Newton loop stops the loop instead of producing NaN.
 */
#include "ws_oracle.h"

#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ---------- wrapping int32 helpers ---------- */
static inline int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
static inline int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
static inline int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
static inline int64_t wmul64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
static inline int64_t wadd64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
static inline int64_t wsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
static inline int32_t iabs32(int32_t a) { return a < 0 ? wsub(0, a) : a; }

/* Vector3<int>::l2norm(): T(sqrtf(x*x+y*y+z*z)) with the sum in int (math/vector3.h:318-330).
 * Beyond ~46 m the int sum wraps and can be negative: sqrtf gives NaN, and the DEVICE float->int conversion of the
 * reference's CUDA path (cvt.rzi.s32.f32) turns NaN into 0 (x86's cvttss2si would give INT_MIN; the oracle restates
 * the CUDA kernel, and gfx950's v_cvt_i32_f32 agrees with it). */
static inline int32_t l2norm_i(int32_t x, int32_t y, int32_t z)
{
  int32_t sq = wadd(wadd(wmul(x, x), wmul(y, y)), wmul(z, z));
  if (sq < 0) return 0;
  return (int32_t)sqrtf((float)sq);
}
/* Vector3<long>::l2norm(): long(sqrtf(float(long sum))) (math/vector3.h:318-330).  A wrapped, negative sum gives NaN here
 * too, but the 64-bit conversion of the reference's CUDA path (cvt.rzi.s64.f32, i.e. __float2ll_rz) turns NaN into
 * 0x8000000000000000, not 0 -- the same value x86's cvttss2si gives, so the host-built reference header (the golden)
 * agrees with the CUDA kernel for this one.  Stated explicitly, because float -> integer of NaN is undefined in C. */
static inline int64_t l2norm_l(int64_t x, int64_t y, int64_t z)
{
  int64_t sq = wadd64(wadd64(wmul64(x, x), wmul64(y, y)), wmul64(z, z));
  if (sq < 0) return INT64_MIN;
  return (int64_t)sqrtf((float)sq);
}

int32_t wso_l2norm_i(int32_t x, int32_t y, int32_t z) { return l2norm_i(x, y, z); }
int64_t wso_l2norm_l(int64_t x, int64_t y, int64_t z) { return l2norm_l(x, y, z); }
/* Vector3<int>::cross, math/vector3.h:269-277 */
void wso_cross_i(const int32_t a[3], const int32_t b[3], int32_t out[3])
{
  out[0] = wsub(wmul(a[1], b[2]), wmul(a[2], b[1]));
  out[1] = wsub(wmul(a[2], b[0]), wmul(a[0], b[2]));
  out[2] = wsub(wmul(a[0], b[1]), wmul(a[1], b[0]));
}

/* ---------- TSDFEntry (include/map/tsdf.h:16-23,32-46) ---------- */
uint32_t wso_pack(int16_t value, int16_t weight)
{
  return (uint32_t)(uint16_t)value | ((uint32_t)(uint16_t)weight << 16);
}
int16_t wso_value(uint32_t raw) { return (int16_t)(raw & 0xffffu); }
int16_t wso_weight(uint32_t raw) { return (int16_t)(raw >> 16); }

/* ---------- ring buffer (include/warpsense/cuda/device_map.h) ---------- */
/* overflow(), device_map.h:14-30 */
static inline int64_t ovf(int64_t val, int64_t max)
{
  if (val >= 2 * max) return val - 2 * max;
  if (val >= max) return val - max;
  return val;
}

/* DeviceMap::get_index, device_map.h:93-101 (z fastest). 64-bit so 2049^3 maps do not wrap. */
int64_t wso_get_index(const wso_map *m, int32_t x, int32_t y, int32_t z)
{
  int64_t sx = m->size[0], sy = m->size[1], sz = m->size[2];
  int64_t xo = ovf((int64_t)x - m->pos[0] + m->offset[0] + sx, sx) * sy * sz;
  int64_t yo = ovf((int64_t)y - m->pos[1] + m->offset[1] + sy, sy) * sz;
  int64_t zo = ovf((int64_t)z - m->pos[2] + m->offset[2] + sz, sz);
  return xo + yo + zo;
}

/* DeviceMap::in_bounds, device_map.h:109-114 */
int wso_in_bounds(const wso_map *m, int32_t x, int32_t y, int32_t z)
{
  int32_t ax = iabs32(wsub(x, m->pos[0])), ay = iabs32(wsub(y, m->pos[1])), az = iabs32(wsub(z, m->pos[2]));
  return ax <= m->size[0] / 2 && ay <= m->size[1] / 2 && az <= m->size[2] / 2;
}

/* device_map.h:123-128; `buffer` is size_t there, so the comparison is unsigned */
int wso_in_bounds_with_buffer_pos(const wso_map *m, int32_t x, int32_t y, int32_t z, int32_t buffer)
{
  uint64_t b = (uint64_t)(int64_t)buffer;
  uint64_t ax = (uint64_t)(int64_t)iabs32(wsub(x, m->pos[0]));
  uint64_t ay = (uint64_t)(int64_t)iabs32(wsub(y, m->pos[1]));
  uint64_t az = (uint64_t)(int64_t)iabs32(wsub(z, m->pos[2]));
  return ax <= (uint64_t)(int64_t)(m->size[0] / 2) + b && ay <= (uint64_t)(int64_t)(m->size[1] / 2) + b &&
         az <= (uint64_t)(int64_t)(m->size[2] / 2) + b;
}

/* device_map.h:116-121 */
int wso_in_bounds_with_buffer_neg(const wso_map *m, int32_t x, int32_t y, int32_t z, int32_t buffer)
{
  uint64_t b = (uint64_t)(int64_t)buffer;
  uint64_t ax = (uint64_t)(int64_t)iabs32(wsub(x, m->pos[0]));
  uint64_t ay = (uint64_t)(int64_t)iabs32(wsub(y, m->pos[1]));
  uint64_t az = (uint64_t)(int64_t)iabs32(wsub(z, m->pos[2]));
  return ax <= (uint64_t)(int64_t)(m->size[0] / 2) - b && ay <= (uint64_t)(int64_t)(m->size[1] / 2) - b &&
         az <= (uint64_t)(int64_t)(m->size[2] / 2) - b;
}

/* ---------- fixed-point helpers ---------- */
/* cu_to_int_mat, cuda/util.h:24-35 (== to_int_mat, util/util.h:8-11): (int)(float * 32768) */
void wso_to_int_mat(const float T[16], int32_t M[16])
{
  for (int i = 0; i < 16; ++i) M[i] = (int32_t)(T[i] * (float)WSO_MATRIX_RESOLUTION);
}

/* cu_transform_point, cuda/util.h:11-22; matrices are column-major: at(i,j) = data[j][i] (matrix4x4.h:175-185) */
void wso_transform_point(const int32_t p[3], const int32_t M[16], int32_t out[3])
{
  for (int i = 0; i < 3; ++i)
  {
    int32_t v = wadd(wadd(wmul(M[0 * 4 + i], p[0]), wmul(M[1 * 4 + i], p[1])), wmul(M[2 * 4 + i], p[2]));
    v = wadd(v, M[3 * 4 + i]);
    out[i] = v / WSO_MATRIX_RESOLUTION;
  }
}

/* cu_to_map, cuda/util.h:111-114: floor of the FLOAT quotient */
void wso_to_map(const int32_t p[3], int32_t res, int32_t out[3])
{
  out[0] = (int32_t)floorf((float)p[0] / (float)res);
  out[1] = (int32_t)floor((double)((float)p[1] / (float)res));
  out[2] = (int32_t)floor((double)((float)p[2] / (float)res));
}

/* update_tsdf.cu:49-50: tan(45/128 deg)/2 * 32768 -> 100 */
int32_t wso_dz_per_distance(void)
{
  float angle = 45.f / 128.f;
  return (int32_t)(tan(angle / 180 * M_PI) / 2.0 * WSO_MATRIX_RESOLUTION);
}

/* TSDFMapping::convert_pose_to_gpu, tsdf_mapping.cpp:77-85 + to_map util/util.h:52-56 */
void wso_convert_pose(const float pose[16], int32_t res, int32_t pos_vox[3], int32_t up[3])
{
  int32_t M[16], R[16];
  wso_to_int_mat(pose, M);
  memset(R, 0, sizeof R);
  for (int j = 0; j < 3; ++j)
    for (int i = 0; i < 3; ++i) R[j * 4 + i] = M[j * 4 + i];
  R[15] = 1; /* Matrix4i::Identity() with the 3x3 block replaced */
  /* transform_point(Point(0,0,MR), rotation_mat): (mat * [p;1]).head(3) / MR, util/util.h:13-18 */
  int32_t p[3] = {0, 0, WSO_MATRIX_RESOLUTION};
  for (int i = 0; i < 3; ++i)
  {
    int32_t v = wadd(wadd(wmul(R[0 * 4 + i], p[0]), wmul(R[1 * 4 + i], p[1])), wmul(R[2 * 4 + i], p[2]));
    v = wadd(v, R[3 * 4 + i]);
    up[i] = v / WSO_MATRIX_RESOLUTION;
  }
  for (int i = 0; i < 3; ++i) pos_vox[i] = (int32_t)floorf(pose[12 + i] / (float)res);
}

/* ---------- atomic_tsdf_min, executed serially (cuda/util.h:70-102) ---------- */
int wso_tsdf_min(uint32_t *addr, uint32_t new_raw)
{
  uint32_t old = *addr;
  int old_v = wso_value(old), new_v = wso_value(new_raw);
  if (abs(old_v) < abs(new_v) || wso_weight(old) > 0) return 0;
  *addr = new_raw; /* the CAS succeeds at once when nobody else runs */
  return 1;
}

/* update_tsdf.cu:57-63: distance and the unit interpolation vector of one ray, all in `long` like the reference.
 * returns -1 for distance == 0, -2 for interpolation_norm == 0 (the reference divides by zero there) */
int wso_ray_setup(const int32_t point[3], const int32_t pos[3], const int32_t up[3], int32_t *distance_out, int64_t iv[3])
{
  const int64_t MR = WSO_MATRIX_RESOLUTION;
  int32_t dir[3] = {wsub(point[0], pos[0]), wsub(point[1], pos[1]), wsub(point[2], pos[2])};
  int32_t distance = l2norm_i(dir[0], dir[1], dir[2]); /* :58 */
  *distance_out = distance;
  if (distance == 0) return -1;
  int64_t nd[3], c1[3];
  for (int k = 0; k < 3; ++k) nd[k] = wmul64((int64_t)dir[k], MR) / distance;
  int64_t u[3] = {up[0], up[1], up[2]};
  c1[0] = wsub64(wmul64(nd[1], u[2]), wmul64(nd[2], u[1])) / MR;
  c1[1] = wsub64(wmul64(nd[2], u[0]), wmul64(nd[0], u[2])) / MR;
  c1[2] = wsub64(wmul64(nd[0], u[1]), wmul64(nd[1], u[0])) / MR;
  iv[0] = wsub64(wmul64(nd[1], c1[2]), wmul64(nd[2], c1[1]));
  iv[1] = wsub64(wmul64(nd[2], c1[0]), wmul64(nd[0], c1[2]));
  iv[2] = wsub64(wmul64(nd[0], c1[1]), wmul64(nd[1], c1[0]));
  int64_t inorm = l2norm_l(iv[0], iv[1], iv[2]);
  if (inorm == 0) return -2;
  for (int k = 0; k < 3; ++k) iv[k] = wmul64(iv[k], MR) / inorm;
  return 0;
}

/* debugging aid for the tests: record every candidate written to one watched voxel of the next wso_update_min call */
static int64_t g_watch_idx = -1;
static int32_t *g_watch_out = 0;
static size_t g_watch_cap = 0, g_watch_n = 0;
void wso_debug_watch(int64_t idx, int32_t *out /* rows of 6: point, len, step, value, weight, accepted */, size_t cap)
{
  g_watch_idx = idx;
  g_watch_out = out;
  g_watch_cap = cap;
  g_watch_n = 0;
}
size_t wso_debug_watch_count(void) { return g_watch_n; }

/* ---------- cu_min_tsdf_krnl, update_tsdf.cu:45-128, one "thread" after the other ---------- */
void wso_update_min(wso_map *new_map, const int32_t *xyz, size_t n, const int32_t scanner_pos[3],
                    const int32_t up[3], int32_t tau, int32_t res, wso_update_stats *stats)
{
  const int64_t MR = WSO_MATRIX_RESOLUTION;
  const int32_t dz_per_distance = wso_dz_per_distance();
  const int32_t weight_epsilon = tau / 10;
  wso_update_stats st = {0, 0, 0, 0};
  /* cu_to_mm, cuda/util.h:116-123 */
  int32_t pos[3];
  for (int k = 0; k < 3; ++k) pos[k] = wadd(wmul(scanner_pos[k], res), res / 2);

  for (size_t ix = 0; ix < n; ++ix)
  {
    const int32_t *point = xyz + 3 * ix;
    int32_t cell[3];
    wso_to_map(point, res, cell);
    if (!wso_in_bounds_with_buffer_pos(new_map, cell[0], cell[1], cell[2], tau / res / 2)) continue; /* :55 */
    st.rays_in_bounds++;

    int32_t dir[3] = {wsub(point[0], pos[0]), wsub(point[1], pos[1]), wsub(point[2], pos[2])};
    int32_t distance;
    int64_t iv[3];
    if (wso_ray_setup(point, pos, up, &distance, iv) != 0) { st.rays_degenerate++; continue; } /* guards, see header */

    int32_t prev[3] = {0, 0, 0}; /* :65 */
    for (int32_t len = 1; len <= distance + tau; len += res / 2) /* :67 */
    {
      int32_t proj[3], index[3];
      for (int k = 0; k < 3; ++k)
      {
        proj[k] = wadd(pos[k], wmul(dir[k], len) / distance);
        index[k] = proj[k] / res;
      }
      if (index[0] == prev[0] && index[1] == prev[1]) continue; /* :71 */
      prev[0] = index[0]; prev[1] = index[1]; prev[2] = index[2];
      if (!wso_in_bounds(new_map, index[0], index[1], index[2])) continue;

      /* :81-98 */
      int32_t tc[3];
      for (int k = 0; k < 3; ++k) tc[k] = wadd(wmul(index[k], res), res / 2);
      int64_t initial_value = l2norm_i(wsub(point[0], tc[0]), wsub(point[1], tc[1]), wsub(point[2], tc[2]));
      int32_t value = (int32_t)(initial_value < (int64_t)tau ? initial_value : (int64_t)tau);
      if (len > distance) value = -value;
      int32_t weight = WSO_WEIGHT_RESOLUTION;
      if (value < -weight_epsilon) weight = WSO_WEIGHT_RESOLUTION * (tau + value) / (tau - weight_epsilon);
      if (weight == 0) continue;
      int16_t v16 = (int16_t)value, w16 = (int16_t)weight;

      /* :101-105 */
      int32_t delta_z = wmul(dz_per_distance, len) / WSO_MATRIX_RESOLUTION;
      int32_t iter_steps = (delta_z * 2) / res + 1;
      int32_t mid = delta_z / res;
      int32_t lowest[3];
      for (int k = 0; k < 3; ++k) lowest[k] = wsub(proj[k], (int32_t)(wmul64((int64_t)delta_z, iv[k]) / MR));

      for (int32_t step = 0; step < iter_steps; ++step) /* :107-125 */
      {
        int32_t idx[3];
        int64_t sm = (int64_t)wmul(step, res);
        for (int k = 0; k < 3; ++k) idx[k] = wadd(lowest[k], (int32_t)(wmul64(sm, iv[k]) / MR)) / res;
        if (!wso_in_bounds(new_map, idx[0], idx[1], idx[2])) continue;
        int16_t w = w16;
        if (step != mid) w = (int16_t)(w * -1);
        st.write_calls++;
        const int64_t lin = wso_get_index(new_map, idx[0], idx[1], idx[2]);
        const int acc = wso_tsdf_min(new_map->data + lin, wso_pack(v16, w));
        st.accepted += acc;
        if (lin == g_watch_idx && g_watch_out && g_watch_n < g_watch_cap)
        {
          int32_t *o = g_watch_out + 6 * g_watch_n++;
          o[0] = (int32_t)ix; o[1] = len; o[2] = step; o[3] = v16; o[4] = w; o[5] = acc;
        }
      }
    }
  }
  if (stats) *stats = st;
}

/* ---------- cu_avg_tsdf_krnl, update_tsdf.cu:13-43 ---------- */
void wso_update_avg(uint32_t *new_data, uint32_t *avg_data, int64_t n_vox, int32_t max_weight, int32_t tau)
{
  const uint32_t reset = wso_pack((int16_t)tau, 0);
  for (int64_t i = 0; i < n_vox; ++i)
  {
    int32_t nv = wso_value(new_data[i]), nw = wso_weight(new_data[i]);
    int32_t ev = wso_value(avg_data[i]), ew = wso_weight(avg_data[i]);
    if (nw > 0 && ew > 0)
    {
      int16_t v = (int16_t)((ev * ew + nv * nw) / (ew + nw));
      int32_t ws = ew + nw;
      int16_t w = (int16_t)(max_weight < ws ? max_weight : ws);
      avg_data[i] = wso_pack(v, w);
    }
    else if (nw != 0 && ew <= 0)
    {
      avg_data[i] = wso_pack((int16_t)nv, (int16_t)nw);
    }
    new_data[i] = reset;
  }
}

/* TSDFCuda::update_tsdf, update_tsdf.cu:143-166 */
void wso_update_tsdf(wso_map *avg_map, wso_map *new_map, const int32_t *xyz, size_t n,
                     const int32_t scanner_pos[3], const int32_t up[3], int32_t tau, int32_t max_weight,
                     int32_t res, wso_update_stats *stats)
{
  wso_update_min(new_map, xyz, n, scanner_pos, up, tau, res, stats);
  int64_t n_vox = (int64_t)avg_map->size[0] * avg_map->size[1] * avg_map->size[2];
  wso_update_avg(new_map->data, avg_map->data, n_vox, max_weight, tau);
}

/* ---------- calc_jacobis_krnl<int>, registration.cu:194-257 ---------- */
static size_t reg_effective_n(size_t n, uint32_t flags)
{
  if (flags & WSO_REG_COMPAT_REFERENCE_LAUNCH)
  {
    /* <<<128,512>>> launches 65536 threads, no grid-stride loop (registration.cu:353) */
    if (n > 65536) n = 65536;
  }
  return n;
}

void wso_calc_jacobis(const wso_map *map, const float T[16], const int32_t *xyz, size_t n, int32_t res,
                      int64_t *jacobis, int16_t *values, uint8_t *mask, uint32_t flags)
{
  int32_t M[16];
  wso_to_int_mat(T, M);
  int32_t center[3] = {(int32_t)T[12], (int32_t)T[13], (int32_t)T[14]};
  size_t n_eff = reg_effective_n(n, flags);
  for (size_t idx = 0; idx < n; ++idx)
  {
    mask[idx] = 0;
    values[idx] = 0;
    for (int k = 0; k < 6; ++k) jacobis[6 * idx + k] = 0;
    if (idx >= n_eff) continue;

    int32_t point[3], buf[3];
    wso_transform_point(xyz + 3 * idx, M, point);
    for (int k = 0; k < 3; ++k) buf[k] = point[k] / res;
    for (int k = 0; k < 3; ++k) point[k] = wsub(point[k], center[k]);
    if (!wso_in_bounds_with_buffer_neg(map, buf[0], buf[1], buf[2], 1)) continue;
    uint32_t cur = map->data[wso_get_index(map, buf[0], buf[1], buf[2])];
    if (wso_weight(cur) == 0) continue;

    int32_t grad[3] = {0, 0, 0};
    for (int k = 0; k < 3; ++k)
    {
      int32_t a[3] = {buf[0], buf[1], buf[2]}, b[3] = {buf[0], buf[1], buf[2]};
      a[k] += 1;
      b[k] -= 1;
      uint32_t nx = map->data[wso_get_index(map, a[0], a[1], a[2])];
      uint32_t ls = map->data[wso_get_index(map, b[0], b[1], b[2])];
      int32_t nv = wso_value(nx), lv = wso_value(ls);
      if (wso_weight(nx) != 0 && wso_weight(ls) != 0 && !((nv > 0 && lv < 0) || (nv < 0 && lv > 0)))
        grad[k] = (nv - lv) / 2;
    }
    /* point.cross(gradient) in int, math/vector3.h:269-277 */
    int32_t cr[3];
    cr[0] = wsub(wmul(point[1], grad[2]), wmul(point[2], grad[1]));
    cr[1] = wsub(wmul(point[2], grad[0]), wmul(point[0], grad[2]));
    cr[2] = wsub(wmul(point[0], grad[1]), wmul(point[1], grad[0]));
    for (int k = 0; k < 3; ++k)
    {
      jacobis[6 * idx + k] = cr[k];
      jacobis[6 * idx + 3 + k] = grad[k];
    }
    values[idx] = wso_value(cur);
    mask[idx] = 1;
  }
}

/* ---------- h_g_e_reduction_krnl + host reduce, registration.cu:14-192,310-345: an exact integer sum ---------- */
void wso_reduce(const int64_t *jacobis, const int16_t *values, const uint8_t *mask, size_t n, int64_t h[36],
                int64_t g[6], int32_t *e, int32_t *c, uint32_t flags)
{
  size_t n_eff = n;
  if (flags & WSO_REG_COMPAT_REFERENCE_LAUNCH)
  {
    /* 32 blocks each own a slice of floor(N/32) points (registration.cu:355-356); for N < 128 only
     * block 0 reads in-range data, the other blocks' reads are out of range in the reference. */
    if (n >= 128) n_eff = 32 * (n / 32);
  }
  memset(h, 0, 36 * sizeof(int64_t));
  memset(g, 0, 6 * sizeof(int64_t));
  int32_t ee = 0, cc = 0;
  for (size_t idx = 0; idx < n_eff; ++idx)
  {
    if (!mask[idx]) continue;
    const int64_t *J = jacobis + 6 * idx;
    int64_t v = values[idx];
    for (int j = 0; j < 6; ++j)
      for (int i = 0; i < 6; ++i) h[j * 6 + i] = wadd64(h[j * 6 + i], wmul64(J[i], J[j]));
    for (int i = 0; i < 6; ++i) g[i] = wadd64(g[i], wmul64(J[i], v));
    ee = wadd(ee, abs((int)values[idx]));
    cc = wadd(cc, 1);
  }
  *e = ee;
  *c = cc;
}

/* RegistrationCuda::perform_registration, registration.cu:347-368 */
void wso_reg_iterate(const wso_map *map, const float T[16], const int32_t *xyz, size_t n, int32_t res,
                     int64_t h[36], int64_t g[6], int32_t *e, int32_t *c, uint32_t flags)
{
  int64_t *J = (int64_t *)malloc(sizeof(int64_t) * 6 * (n ? n : 1));
  int16_t *v = (int16_t *)malloc(sizeof(int16_t) * (n ? n : 1));
  uint8_t *m = (uint8_t *)malloc(n ? n : 1);
  wso_calc_jacobis(map, T, xyz, n, res, J, v, m, flags);
  wso_reduce(J, v, m, n, h, g, e, c, flags);
  free(J);
  free(v);
  free(m);
}

/* ---------- 6x6 solve in double (Eigen's hf.inverse() * g, tsdf_registration.cpp:69) ----------
 * Gauss-Jordan elimination with partial pivoting; the multipliers are formed with the pivot's reciprocal (one division per
 * pivot, six in all) and the elimination clears the column above the pivot as well, so there is no back substitution:
 * x[i] = b[i] * (1 / pivot i).  The reference inverts with Eigen (PartialPivLU for a 6x6, version unpinned) and multiplies:
 * parity for this one step is UNPINNED either way and absorbed by the pose tolerance (1e-4 m / 1e-4 rad); the choice of
 * elimination order is this repository's.  Until round 5 it was LU + back substitution with eleven divisions: on the GPU the
 * solve is one wave's dependent chain in the middle of every Gauss-Newton iteration, and the six divisions plus 36 operand
 * fetches of the back substitution were 40 % of it.  tests/test_oracle_pins.py holds this solver against numpy's LAPACK solve
 * (the reference's algorithm class) on the benchmark's own normal equations: relative difference below 1e-9. */
int wso_solve6(const double A_in[36], const double b_in[6], double x[6])
{
  double A[6][6], b[6], inv[6];
  for (int i = 0; i < 6; ++i)
  {
    b[i] = b_in[i];
    for (int j = 0; j < 6; ++j) A[i][j] = A_in[i * 6 + j];
  }
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double best = fabs(A[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (fabs(A[i][k]) > best) { best = fabs(A[i][k]); piv = i; }
    if (best == 0.0) return -1;
    if (piv != k)
    {
      for (int j = 0; j < 6; ++j) { double t = A[k][j]; A[k][j] = A[piv][j]; A[piv][j] = t; }
      double t = b[k]; b[k] = b[piv]; b[piv] = t;
    }
    inv[k] = 1.0 / A[k][k];
    for (int i = 0; i < 6; ++i)
    {
      if (i == k) continue;
      double f = A[i][k] * inv[k];
      for (int j = k + 1; j < 6; ++j) A[i][j] -= f * A[k][j];
      b[i] -= f * b[k];
    }
  }
  for (int i = 0; i < 6; ++i) x[i] = b[i] * inv[i];
  return 0;
}

/* ---------- xi_to_transform, include/warpsense/registration/util.h:5-39 ---------- */
void wso_xi_to_transform(const double xi[6], const int32_t center[3], float T[16])
{
  double theta = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
  float L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  if (theta != 0.0)
  {
    double lx = xi[0] / theta, ly = xi[1] / theta, lz = xi[2] / theta;
    L[0][1] = (float)-lz; L[0][2] = (float)ly;
    L[1][0] = (float)lz;  L[1][2] = (float)-lx;
    L[2][0] = (float)-ly; L[2][1] = (float)lx;
  }
  float s = (float)sin(theta), omc = (float)(1 - cos(theta));
  float R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
    {
      float ll = 0.f;
      for (int k = 0; k < 3; ++k) ll += (omc * L[i][k]) * L[k][j];
      R[i][j] = ((i == j ? 1.f : 0.f) + s * L[i][j]) + ll;
    }
  float oc[3] = {-(float)center[0], -(float)center[1], -(float)center[2]};
  for (int i = 0; i < 16; ++i) T[i] = 0.f;
  T[15] = 1.f;
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) T[j * 4 + i] = R[i][j];
    float shift = ((R[i][0] * oc[0] + R[i][1] * oc[1]) + R[i][2] * oc[2]) + 0.f * 1.f;
    T[12 + i] = (shift + (float)center[i]) + (float)xi[3 + i];
  }
}

static void matmul4f(const float A[16], const float B[16], float C[16])
{
  float out[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i)
    {
      float s = 0.f;
      for (int k = 0; k < 4; ++k) s += A[k * 4 + i] * B[j * 4 + k];
      out[j * 4 + i] = s;
    }
  memcpy(C, out, sizeof out);
}

/* ---------- TSDFRegistration::register_cloud, tsdf_registration.cpp:28-96 ---------- */
/* state set-up of :30-52 */
void wso_gn_begin(wso_gn_state *st, const float T_in[16], int32_t max_iterations, float it_weight_gradient, float epsilon)
{
  memset(st, 0, sizeof *st);
  memcpy(st->T, T_in, 16 * sizeof(float));
  for (int k = 0; k < 3; ++k) st->center[k] = (int32_t)T_in[12 + k]; /* :33 */
  st->alpha = 0.f;
  st->it_weight_gradient = it_weight_gradient;
  st->epsilon = epsilon;
  st->max_iterations = max_iterations;
}

/* one pass through the loop body after perform_registration returned h,g,e,c (:63-92); sums = h[36] g[6] e c */
void wso_gn_update(wso_gn_state *st, const int64_t sums[44])
{
  if (st->finished || st->iterations >= st->max_iterations) return;
  const int32_t e = (int32_t)sums[42], c = (int32_t)sums[43];
  st->iterations += 1;
  if (c == 0) { st->finished = 1; return; } /* guard, see header */

  double hf[36], gf[6], xi[6];
  double w = (double)(st->alpha * (float)c); /* alpha * gpu_c is a float product, :66 */
  for (int r = 0; r < 6; ++r)
  {
    gf[r] = (double)sums[36 + r];
    for (int q = 0; q < 6; ++q) hf[r * 6 + q] = (double)sums[q * 6 + r] + (r == q ? w : 0.0);
  }
  if (wso_solve6(hf, gf, xi) != 0) { st->finished = 1; return; }
  for (int r = 0; r < 6; ++r) xi[r] = -xi[r];

  float tr[16];
  wso_xi_to_transform(xi, st->center, tr);
  st->alpha += st->it_weight_gradient;
  matmul4f(tr, st->T, st->T);

  float err = (float)e / c;
  if (fabsf(err - st->prev[2]) < st->epsilon && fabsf(err - st->prev[0]) < st->epsilon) st->finished = 1;
  st->prev[0] = st->prev[1]; st->prev[1] = st->prev[2]; st->prev[2] = st->prev[3];
  st->prev[3] = err;
}

int wso_register_cloud(const wso_map *map, const int32_t *xyz, size_t n, const float T_in[16],
                       int32_t max_iterations, float it_weight_gradient, float epsilon, int32_t res,
                       uint32_t flags, float T_out[16], int64_t *trace, int32_t trace_cap)
{
  wso_gn_state st;
  wso_gn_begin(&st, T_in, max_iterations, it_weight_gradient, epsilon);
  while (!st.finished && st.iterations < st.max_iterations)
  {
    int64_t sums[44];
    int32_t e, c;
    wso_reg_iterate(map, st.T, xyz, n, res, sums, sums + 36, &e, &c, flags);
    sums[42] = e;
    sums[43] = c;
    if (trace && st.iterations < trace_cap) memcpy(trace + 44 * (size_t)st.iterations, sums, sizeof sums);
    wso_gn_update(&st, sums);
  }
  memcpy(T_out, st.T, 16 * sizeof(float));
  return st.iterations;
}

/* ------------------------------------------------------------------------------------------------
 * App::preprocess — src/warpsense/app.cpp:119-148.
 *   :129-132  skip the point if x < 0.3 && y < 0.3 && z < 0.3 (floats compared with the double literal)
 *   :134      Pointf(x * 1000.f, y * 1000.f, z * 1000.f)
 *   :135-140  voxel centre = (int)(std::floor(p / res) * res + res / 2): float / int, float * int, int / int
 *   :142      transform_point(voxel_center, to_int_mat(pose_)) — util/util.h:20-34, int arithmetic, / 32768
 *   :123,142-146  unordered_set<Pointi>: every distinct point once; iteration order is the implementation's,
 *             so the restatement fixes it: order of first occurrence.
 * Non-finite coordinates (undefined behaviour in the reference: float -> int of NaN) are skipped.
 */
static int pre_cmp(const void *a, const void *b)
{
  const int64_t *x = (const int64_t *)a, *y = (const int64_t *)b;
  for (int k = 0; k < 3; ++k)
    if (x[k] != y[k]) return x[k] < y[k] ? -1 : 1;
  return x[3] < y[3] ? -1 : (x[3] > y[3] ? 1 : 0); /* ties: input index, so the first occurrence sorts first */
}

size_t wso_preprocess(const float *xyz, size_t n, size_t stride, const float pose[16], int32_t res, int32_t *out)
{
  int32_t M[16];
  wso_to_int_mat(pose, M);
  /* (x, y, z, index) of every surviving point, sorted to find duplicates, then put back in input order */
  int64_t *rec = (int64_t *)malloc((n ? n : 1) * 4 * sizeof(int64_t));
  unsigned char *keep = (unsigned char *)calloc(n ? n : 1, 1);
  int32_t *pts = (int32_t *)malloc((n ? n : 1) * 3 * sizeof(int32_t));
  size_t m = 0;
  for (size_t i = 0; i < n; ++i)
  {
    const float x = xyz[i * stride + 0], y = xyz[i * stride + 1], z = xyz[i * stride + 2];
    if (!isfinite(x) || !isfinite(y) || !isfinite(z)) continue;
    if (x < 0.3 && y < 0.3 && z < 0.3) continue;
    const float px = x * 1000.f, py = y * 1000.f, pz = z * 1000.f;
    int32_t c[3], q[3];
    c[0] = (int32_t)(floorf(px / (float)res) * (float)res + (float)(res / 2));
    c[1] = (int32_t)(floorf(py / (float)res) * (float)res + (float)(res / 2));
    c[2] = (int32_t)(floorf(pz / (float)res) * (float)res + (float)(res / 2));
    wso_transform_point(c, M, q);
    for (int k = 0; k < 3; ++k) pts[3 * i + k] = q[k];
    rec[4 * m + 0] = q[0];
    rec[4 * m + 1] = q[1];
    rec[4 * m + 2] = q[2];
    rec[4 * m + 3] = (int64_t)i;
    ++m;
  }
  qsort(rec, m, 4 * sizeof(int64_t), pre_cmp);
  for (size_t j = 0; j < m; ++j)
    if (j == 0 || rec[4 * j + 0] != rec[4 * (j - 1) + 0] || rec[4 * j + 1] != rec[4 * (j - 1) + 1] || rec[4 * j + 2] != rec[4 * (j - 1) + 2])
      keep[rec[4 * j + 3]] = 1;
  size_t n_out = 0;
  for (size_t i = 0; i < n; ++i)
    if (keep[i])
    {
      for (int k = 0; k < 3; ++k) out[3 * n_out + k] = pts[3 * i + k];
      ++n_out;
    }
  free(rec);
  free(keep);
  free(pts);
  return n_out;
}
