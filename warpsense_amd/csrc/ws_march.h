// ws_march.h — the per-step arithmetic of the reference's ray-march (update_tsdf.cu:52-125), shared by the
// scatter kernels.  Everything here is exact integer arithmetic identical to oracle/ws_oracle.c:wso_update_min.
#pragma once

#include "ws_device.h"

namespace ws
{
// Per-ray constants of update_tsdf.cu:52-63, computed once by ray_setup_kernel.
struct RaySetup // 48 bytes
{
  int32_t dx, dy, dz;    // direction_vector = point - pos (mm)
  int32_t distance;      // (int)|direction_vector|
  int32_t ivx, ivy, ivz; // interpolation_vector (unit length == MATRIX_RESOLUTION)
  int32_t steps;         // iterations of the ray-march loop; 0 = ray contributes nothing
  uint64_t div_m;        // multiply-shift constants for the division by `distance`
  int32_t div_k;
  int32_t pad;
};
static_assert(sizeof(RaySetup) == 48, "ws_map::rays is sized for 48-byte records");

// exact floor(x / d) for 0 <= x < 2^31 by multiply-shift: M = ceil(2^k / d), k = 31 + ceil(log2 d)
// (error e = M*d - 2^k < d <= 2^(k-31), so x*e < 2^k for every x < 2^31)
struct FastDiv
{
  uint64_t M;
  int32_t k;
  int32_t d;
};
__host__ __device__ inline FastDiv make_fastdiv(int32_t d)
{
  FastDiv f;
  f.d = d;
  int l = 0;
  while ((1ll << l) < d) ++l;
  f.k = 31 + l;
  const uint64_t p = 1ull << f.k; // k <= 62
  f.M = p / (uint64_t)d + ((p % (uint64_t)d) ? 1 : 0);
  return f;
}
// C-style truncating division of any int32 by the prepared positive divisor
__device__ __forceinline__ int32_t div_trunc(int32_t x, uint64_t M, int32_t k, int32_t d)
{
  const uint32_t ax = x < 0 ? (uint32_t)0 - (uint32_t)x : (uint32_t)x;
  uint32_t q;
  if (ax == 0x80000000u)
    q = ax / (uint32_t)d; // |INT_MIN| is outside the multiply-shift range
  else
    q = (uint32_t)(((uint64_t)ax * M) >> k); // ax < 2^31, M <= 2^32
  return x < 0 ? (int32_t)((uint32_t)0 - q) : (int32_t)q;
}

// scan-wide constants of the march
struct MarchFrame
{
  int32_t posx, posy, posz; // cu_to_mm(scanner_pos), cuda/util.h:116-123
  int32_t res, half, tau, weight_epsilon;
  uint64_t rM; // multiply-shift division by res
  int32_t rK;
  MapParams map;
};
__host__ __device__ inline MarchFrame make_march_frame(const int32_t scanner_pos[3], int32_t res, int32_t tau, const MapParams &map)
{
  MarchFrame f;
  f.res = res;
  f.half = res / 2;
  f.tau = tau;
  f.weight_epsilon = tau / 10;
  f.posx = (int32_t)((uint32_t)scanner_pos[0] * (uint32_t)res + (uint32_t)f.half);
  f.posy = (int32_t)((uint32_t)scanner_pos[1] * (uint32_t)res + (uint32_t)f.half);
  f.posz = (int32_t)((uint32_t)scanner_pos[2] * (uint32_t)res + (uint32_t)f.half);
  const FastDiv fd = make_fastdiv(res);
  f.rM = fd.M;
  f.rK = fd.k;
  f.map = map;
  return f;
}

// Walk the steps [k0, k1) of one ray and call emit(k, fan_step, vx, vy, vz, value, positive) for every
// write_tsdf_min the reference would issue (update_tsdf.cu:67-125): voxel in bounds, weight != 0.
// `positive` = on-ray entry (step == mid, positive weight), else the weight is negated.
template <class Emit>
__device__ __forceinline__ void march_steps(const MarchFrame &f, const RaySetup &r, int32_t k0, int32_t k1, Emit &&emit)
{
  const int64_t MR = MATRIX_RESOLUTION;
  const int32_t res = f.res, half = f.half, tau = f.tau;
  const int32_t px = wadd(f.posx, r.dx), py = wadd(f.posy, r.dy), pz = wadd(f.posz, r.dz);
  const int64_t ivx = r.ivx, ivy = r.ivy, ivz = r.ivz;

  // `prev` of update_tsdf.cu:65-76 is always the (x, y) index of the previous step (or (0,0) before the first)
  int32_t prevx = 0, prevy = 0;
  if (k0 > 0)
  {
    const int32_t len = 1 + (k0 - 1) * half;
    prevx = div_trunc(wadd(f.posx, div_trunc(wmul(r.dx, len), r.div_m, r.div_k, r.distance)), f.rM, f.rK, res);
    prevy = div_trunc(wadd(f.posy, div_trunc(wmul(r.dy, len), r.div_m, r.div_k, r.distance)), f.rM, f.rK, res);
  }
  for (int32_t k = k0; k < k1; ++k)
  {
    const int32_t len = 1 + k * half;
    const int32_t projx = wadd(f.posx, div_trunc(wmul(r.dx, len), r.div_m, r.div_k, r.distance));
    const int32_t projy = wadd(f.posy, div_trunc(wmul(r.dy, len), r.div_m, r.div_k, r.distance));
    const int32_t projz = wadd(f.posz, div_trunc(wmul(r.dz, len), r.div_m, r.div_k, r.distance));
    const int32_t ixx = div_trunc(projx, f.rM, f.rK, res), iyy = div_trunc(projy, f.rM, f.rK, res), izz = div_trunc(projz, f.rM, f.rK, res);
    if (ixx == prevx && iyy == prevy) continue;
    prevx = ixx;
    prevy = iyy;
    if (!in_bounds(f.map, ixx, iyy, izz)) continue;

    // update_tsdf.cu:81-98
    const int32_t tcx = wadd(wmul(ixx, res), half), tcy = wadd(wmul(iyy, res), half), tcz = wadd(wmul(izz, res), half);
    int32_t value = l2norm_i(wsub(px, tcx), wsub(py, tcy), wsub(pz, tcz));
    value = value < tau ? value : tau;
    if (len > r.distance) value = -value;
    if (tsdf_weight(value, tau, f.weight_epsilon) == 0) continue;

    // update_tsdf.cu:101-105
    const int32_t delta_z = wmul(DZ_PER_DISTANCE, len) / MATRIX_RESOLUTION;
    const int32_t iter_steps = (delta_z * 2) / res + 1;
    const int32_t mid = delta_z / res;
    const int32_t lowx = wsub(projx, (int32_t)(wmul64(delta_z, ivx) / MR));
    const int32_t lowy = wsub(projy, (int32_t)(wmul64(delta_z, ivy) / MR));
    const int32_t lowz = wsub(projz, (int32_t)(wmul64(delta_z, ivz) / MR));
    for (int32_t step = 0; step < iter_steps; ++step)
    {
      const int64_t sm = (int64_t)wmul(step, res);
      const int32_t vx = div_trunc(wadd(lowx, (int32_t)(wmul64(sm, ivx) / MR)), f.rM, f.rK, res);
      const int32_t vy = div_trunc(wadd(lowy, (int32_t)(wmul64(sm, ivy) / MR)), f.rM, f.rK, res);
      const int32_t vz = div_trunc(wadd(lowz, (int32_t)(wmul64(sm, ivz) / MR)), f.rM, f.rK, res);
      if (!in_bounds(f.map, vx, vy, vz)) continue;
      emit(k, step, vx, vy, vz, value, step == mid);
    }
  }
}

// order key of a candidate: point(20) | ray step(16) | fan step(8)
__device__ __forceinline__ uint64_t order_key(uint32_t ix, int32_t k, int32_t step)
{
  return ((uint64_t)ix << 24) | ((uint64_t)(uint32_t)k << 8) | (uint64_t)(uint32_t)step;
}
__device__ __forceinline__ uint64_t make_kpos(uint64_t t, int32_t value) { return (t << 16) | ((uint32_t)value & 0xffffu); }
__device__ __forceinline__ uint64_t make_kneg(uint64_t t, int32_t value)
{
  const uint64_t a = (uint64_t)((uint32_t)(value < 0 ? -value : value) & 0x7fffu);
  return (a << 45) | ((T_MASK - t) << 1) | (value < 0 ? 1u : 0u);
}

} // namespace ws
