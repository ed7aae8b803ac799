// ws_march.h — the per-step arithmetic of the reference's ray-march (update_tsdf.cu:52-125), shared by the
// scatter kernels.  Everything here is exact integer arithmetic identical to oracle/ws_oracle.c:wso_update_min.
#pragma once

#include "ws_device.h"

namespace ws
{
// Per-ray constants of update_tsdf.cu:52-63, computed once by ray_setup_kernel.
struct RaySetup // 56 bytes
{
  int32_t dx, dy, dz;    // direction_vector = point - pos (mm)
  int32_t distance;      // (int)|direction_vector|
  int32_t ivx, ivy, ivz; // interpolation_vector (unit length == MATRIX_RESOLUTION)
  int32_t steps;         // iterations of the ray-march loop; 0 = ray contributes nothing
  uint32_t div_m;        // multiply-shift constants for the division by `distance`: M = ceil(2^k / distance), k = 31 + ceil(log2 distance),
  uint32_t spare;        // always below 2^32 -- a 32-BIT field: read out of a 64-bit one, `(uint32_t)div_m` reached the kernels as a
  int32_t div_k;         // 64-bit operand with a zero high half, and every v_mul_hi_u32 with it dragged a wasted v_mad_u64_u32 along
  int32_t pad;    // bit 0: the division-free walk (march_steps_fast) is exact for this ray; bits 1..: direction bin
  int32_t kfirst; // first step of the ray TAIL: steps [kfirst, steps) go through the order keys, [0, kfirst) are free space
  uint32_t ub;    // upper bound of the tail's scatter targets: sum over its steps of iter_steps (update_tsdf.cu:102)
};
static_assert(sizeof(RaySetup) == 56, "ws_map::rays is sized with ray_setup_bytes()");

// scan-wide constants of the march
struct MarchFrame
{
  int32_t posx, posy, posz; // cu_to_mm(scanner_pos), cuda/util.h:116-123
  int32_t res, half, tau, weight_epsilon;
  uint64_t rM; // multiply-shift division by res
  int32_t rK;
  uint32_t rM32; // the same constant as 32 bits (it is below 2^32 for every res >= 2) for v_mul_hi_u32
  int32_t rS;    // rK - 32
  int32_t ringK[3]; // offset - pos + size: storage coordinate = ring(v + ringK, size) (device_map.h:93-101)
  // the same two steps for the rays that stay inside the window (RAY_SIMPLE), with fewer instructions (div_res_b / ring_b below):
  // the division on a numerator made non-negative by adding divB = res * divBq (a bound on |coordinate| inside the window)
  uint32_t divB;
  int32_t divBq;
  int32_t resm1;    // res - 1
  int32_t ringB[3]; // offset - pos - divBq
  bool biased_ok;   // divB < 2^30 and divBq < 2^22: otherwise no ray of the scan is RAY_SIMPLE
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
  f.rM32 = (uint32_t)fd.M; // M = ceil(2^(31+l) / res) with 2^(l-1) < res <= 2^l: M < 2^32
  f.rS = fd.k - 32;
  for (int k = 0; k < 3; ++k) f.ringK[k] = (int32_t)((uint32_t)map.offset[k] - (uint32_t)map.pos[k] + (uint32_t)map.size[k]);
  {
    // |coordinate| of any voxel inside the window, in millimetres, is below (|pos| + size / 2 + 1) * res on every axis
    int64_t reach = 0;
    for (int k = 0; k < 3; ++k)
    {
      const int64_t r = (map.pos[k] < 0 ? -(int64_t)map.pos[k] : (int64_t)map.pos[k]) + map.size[k] / 2 + 4;
      reach = r > reach ? r : reach;
    }
    f.biased_ok = reach < (1ll << 22) && reach * res < (1ll << 30);
    f.divBq = f.biased_ok ? (int32_t)reach : 0;
    f.divB = (uint32_t)f.divBq * (uint32_t)res;
    f.resm1 = res - 1;
    for (int k = 0; k < 3; ++k) f.ringB[k] = (int32_t)((uint32_t)map.offset[k] - (uint32_t)map.pos[k] - (uint32_t)f.divBq);
  }
  f.map = map;
  return f;
}

typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));
typedef u32x4 __attribute__((aligned(4))) u32x4_a4; // four consecutive voxels of a column: dword aligned only

// ring-buffer storage coordinates of a world voxel (device_map.h:93-101)
__device__ __forceinline__ void storage_coords(const MapParams &m, int32_t vx, int32_t vy, int32_t vz, int32_t &sx, int32_t &sy, int32_t &sz)
{
  sx = ring(vx - m.pos[0] + m.offset[0] + m.size[0], m.size[0]);
  sy = ring(vy - m.pos[1] + m.offset[1] + m.size[1], m.size[1]);
  sz = ring(vz - m.pos[2] + m.offset[2] + m.size[2], m.size[2]);
}
__device__ __forceinline__ int64_t storage_index(const MapParams &m, int32_t sx, int32_t sy, int32_t sz)
{
  // sizes are below 2^24 and size[0] * size[1] below 2^31 (checked by ws_map_create): one full-rate 24-bit multiply-add
  const int32_t row = (int32_t)(__umul24((uint32_t)sx, (uint32_t)m.size[1]) + (uint32_t)sy);
  return (int64_t)row * (int64_t)m.size[2] + sz;
}
// Walk the steps [k0, k1) of one ray and call emit(k, fan_step, vx, vy, vz, value, positive) for every
// write_tsdf_min the reference would issue (update_tsdf.cu:67-125): voxel in bounds, weight != 0.
// `positive` = on-ray entry (step == mid, positive weight), else the weight is negated.
template <class Emit>
__device__ __forceinline__ void march_steps_direct(const MarchFrame &f, const RaySetup &r, int32_t k0, int32_t k1, Emit &&emit)
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
    if (tsdf_weight_is_zero(value, tau, f.weight_epsilon)) continue;

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

// ---- the same walk without a division or a 64-bit multiply per step ------------------------------------
// 32-bit integer multiplies are quarter rate on CDNA and the direct form needs ~40 of them per step
// (six multiply-shift divisions, three 64-bit products per fan step): ~700 cycles per wave and step,
// ~300 us of pure ALU for one pass over a 131 072-ray scan.  Everything below is exact:
//   * |d|*len = q*dist + r is carried from step to step (len grows by res/2: add the per-ray increment
//     |d|*(res/2) = aq*dist + ar, carry once) — trunc(d*len/dist) = sign(d)*q;
//   * proj moves by less than one voxel per step, so floor(proj/res) and the remainder are carried too;
//     trunc(x/res) = floor unless x < 0 with a non-zero remainder;
//   * fan offsets are small: delta_z*iv and step*res*iv fit 32 bits, /32768 toward zero is a shift.
// RaySetup.pad bit 0 marks rays for which all of this holds without int32 wrap (always, inside the
// reference's own no-overflow domain); other rays use march_steps_direct.
struct AxisWalk
{
  int32_t q, r;    // |d| * len = q * dist + r
  int32_t aq, ar;  // per-step increment of (q, r)
  int32_t neg;     // d < 0
  int32_t pos;     // ray origin on this axis
  int32_t proj;    // pos + sign(d) * q            == update_tsdf.cu:69
  int32_t fi, rem; // floor(proj / res) and proj - fi * res in [0, res)
};
__device__ __forceinline__ int32_t trunc_shift15(int32_t p) { return (p + ((p >> 31) & (MATRIX_RESOLUTION - 1))) >> 15; }

__device__ __forceinline__ void axis_init(AxisWalk &w, const MarchFrame &f, const RaySetup &r, int32_t d, int32_t pos, int32_t k)
{
  const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
  w.neg = d < 0 ? 1 : 0;
  w.pos = pos;
  const uint32_t inc = ad * (uint32_t)f.half;
  w.aq = (int32_t)(((uint64_t)inc * r.div_m) >> r.div_k);
  w.ar = (int32_t)(inc - (uint32_t)w.aq * (uint32_t)r.distance);
  const uint32_t n = ad * (uint32_t)(1 + k * f.half);
  w.q = (int32_t)(((uint64_t)n * r.div_m) >> r.div_k);
  w.r = (int32_t)(n - (uint32_t)w.q * (uint32_t)r.distance);
  w.proj = w.neg ? pos - w.q : pos + w.q;
  const int32_t t = div_trunc(w.proj, f.rM, f.rK, f.res);
  w.fi = t - ((w.proj < 0 && t * f.res != w.proj) ? 1 : 0);
  w.rem = w.proj - w.fi * f.res;
}
__device__ __forceinline__ void axis_step(AxisWalk &w, int32_t dist, int32_t res)
{
  w.r += w.ar;
  int32_t dq = w.aq;
  if (w.r >= dist)
  {
    w.r -= dist;
    dq += 1;
  }
  w.q += dq;
  const int32_t dp = w.neg ? -dq : dq; // |dp| <= res/2 + 1 <= res
  w.proj += dp;
  w.rem += dp;
  if (w.rem >= res)
  {
    w.rem -= res;
    w.fi += 1;
  }
  else if (w.rem < 0)
  {
    w.rem += res;
    w.fi -= 1;
  }
}
// trunc(proj / res)
__device__ __forceinline__ int32_t axis_index(const AxisWalk &w) { return w.fi + ((w.proj < 0 && w.rem != 0) ? 1 : 0); }
// centre of that voxel minus `from`, without a multiply: index * res = proj - rem (+ res when the index was bumped)
__device__ __forceinline__ int32_t axis_centre_delta(const AxisWalk &w, int32_t from, int32_t res, int32_t half)
{
  const int32_t base = w.proj - w.rem + ((w.proj < 0 && w.rem != 0) ? res : 0);
  return from - (base + half);
}
// trunc((proj + e) / res) from the carried floor/remainder of proj; |e| is a few voxels at most
__device__ __forceinline__ int32_t axis_index_offset(const AxisWalk &w, int32_t e, const MarchFrame &f)
{
  const int32_t res = f.res;
  int32_t rem = w.rem + e, fi = w.fi;
  if (rem >= res)
  {
    rem -= res;
    fi += 1;
    if (rem >= res)
    {
      const int32_t m = div_trunc(rem, f.rM, f.rK, res);
      fi += m;
      rem -= m * res;
    }
  }
  else if (rem < 0)
  {
    rem += res;
    fi -= 1;
    if (rem < 0)
    {
      const int32_t m = div_trunc(res - 1 - rem, f.rM, f.rK, res); // ceil(-rem / res)
      fi -= m;
      rem += m * res;
    }
  }
  return fi + ((w.proj + e < 0 && rem != 0) ? 1 : 0);
}

// FREE_SPACE: the caller guarantees that every step of [k0, k1) lies more than tau (+ the centre/fan slack) in front
// of the hit point, so value == tau (weight 64, negated off the ray) without computing it (no squares, no sqrt).
template <bool FREE_SPACE, class Emit>
__device__ __forceinline__ void march_steps_fast(const MarchFrame &f, const RaySetup &r, int32_t k0, int32_t k1, Emit &&emit)
{
  const int32_t res = f.res, half = f.half, tau = f.tau;
  const int32_t px = f.posx + r.dx, py = f.posy + r.dy, pz = f.posz + r.dz;
  AxisWalk wx, wy, wz;
  const int32_t kstart = k0 > 0 ? k0 - 1 : 0;
  axis_init(wx, f, r, r.dx, f.posx, kstart);
  axis_init(wy, f, r, r.dy, f.posy, kstart);
  axis_init(wz, f, r, r.dz, f.posz, kstart);
  int32_t prevx = 0, prevy = 0;
  if (k0 > 0)
  {
    prevx = axis_index(wx);
    prevy = axis_index(wy);
    axis_step(wx, r.distance, res);
    axis_step(wy, r.distance, res);
    axis_step(wz, r.distance, res);
  }
  int32_t len = 1 + k0 * half;
  int32_t last_dz = -1, c0x = 0, c0y = 0, c0z = 0;
  for (int32_t k = k0; k < k1; ++k, len += half)
  {
    if (k > k0)
    {
      axis_step(wx, r.distance, res);
      axis_step(wy, r.distance, res);
      axis_step(wz, r.distance, res);
    }
    const int32_t ixx = axis_index(wx), iyy = axis_index(wy);
    if (ixx == prevx && iyy == prevy) continue;
    prevx = ixx;
    prevy = iyy;
    const int32_t izz = axis_index(wz);
    if (!in_bounds(f.map, ixx, iyy, izz)) continue;

    int32_t value = tau;
    if (!FREE_SPACE)
    {
      value = l2norm_i(axis_centre_delta(wx, px, res, half), axis_centre_delta(wy, py, res, half), axis_centre_delta(wz, pz, res, half));
      value = value < tau ? value : tau;
      if (len > r.distance) value = -value;
      if (tsdf_weight_is_zero(value, tau, f.weight_epsilon)) continue;
    }

    const int32_t delta_z = (DZ_PER_DISTANCE * len) >> 15; // len > 0
    if (delta_z != last_dz)
    {
      // delta_z grows by one every 328 mm of ray: the fan base offset is recomputed a few times per ray
      last_dz = delta_z;
      c0x = trunc_shift15(delta_z * r.ivx);
      c0y = trunc_shift15(delta_z * r.ivy);
      c0z = trunc_shift15(delta_z * r.ivz);
    }
    int32_t iter_steps = 1, mid = 0;
    if (delta_z * 2 >= res)
    {
      iter_steps = (delta_z * 2) / res + 1;
      mid = delta_z / res;
    }
    for (int32_t step = 0; step < iter_steps; ++step)
    {
      int32_t ex = -c0x, ey = -c0y, ez = -c0z;
      if (step)
      {
        const int32_t sm = step * res;
        ex += trunc_shift15(sm * r.ivx);
        ey += trunc_shift15(sm * r.ivy);
        ez += trunc_shift15(sm * r.ivz);
      }
      const int32_t vx = axis_index_offset(wx, ex, f);
      const int32_t vy = axis_index_offset(wy, ey, f);
      const int32_t vz = axis_index_offset(wz, ez, f);
      if (!in_bounds(f.map, vx, vy, vz)) continue;
      emit(k, step, vx, vy, vz, value, step == mid);
    }
  }
}

template <bool FREE_SPACE = false, class Emit>
__device__ __forceinline__ void march_steps(const MarchFrame &f, const RaySetup &r, int32_t k0, int32_t k1, Emit &&emit)
{
  if (r.pad & 1)
    march_steps_fast<FREE_SPACE>(f, r, k0, k1, emit);
  else
    march_steps_direct(f, r, k0, k1, emit);
}

// trunc(x / res) for |x| < 2^31 with one v_mul_hi_u32
__device__ __forceinline__ int32_t div_res(int32_t x, const MarchFrame &f)
{
  const uint32_t ax = (uint32_t)(x < 0 ? -x : x);
  const uint32_t q = __umulhi(ax, f.rM32) >> f.rS;
  return x < 0 ? -(int32_t)q : (int32_t)q;
}
// ring-buffer storage coordinate of a world voxel coordinate INSIDE the window: (v - pos + offset + size) mod size,
// the sum being below 3 * size (device_map.h:14-30); min(x, x - size) in unsigned arithmetic subtracts size iff x >= size
__device__ __forceinline__ int32_t ring_fast(int32_t v, int32_t ringK, int32_t size)
{
  uint32_t x = (uint32_t)(v + ringK);
  x = min(x, x - (uint32_t)size);
  x = min(x, x - (uint32_t)size);
  return (int32_t)x;
}

// The same two for a coordinate INSIDE THE WINDOW (every target of a RAY_SIMPLE ray), fewer instructions -- the marches are bound
// by vector-instruction issue:
//   div_res_b   trunc(y / res) + divBq as ONE unsigned multiply-shift: trunc = floor for y >= 0 and floor((y + res - 1) / res) for
//               y < 0, and y + divB >= 0 -- five instructions (shift, and, add3, mul_hi, shift) instead of the seven of
//               |y| -> multiply-shift -> sign back;
//   ring_b      the biased quotient + ringB = v - pos + offset lies in [-size/2, size/2 + size): it is itself, or itself -+ size,
//               whichever is in [0, size) -- the smallest of the three as unsigned numbers: add, sub, add, min3.
__device__ __forceinline__ uint32_t div_res_b(int32_t y, const MarchFrame &f)
{
  const int32_t fix = (y >> 31) & f.resm1;
  return __umulhi((uint32_t)(y + fix) + f.divB, f.rM32) >> f.rS;
}
__device__ __forceinline__ int32_t ring_b(uint32_t qb, int32_t ringB, int32_t size)
{
  const uint32_t x = qb + (uint32_t)ringB;
  return (int32_t)min(min(x, x - (uint32_t)size), x + (uint32_t)size);
}
// ring_b for a quotient of the MIRRORED coordinate (s y, s = -1 where sm = ~0): the sign comes back inside the addition
__device__ __forceinline__ int32_t ring_m(uint32_t qm, uint32_t sm, uint32_t kc, int32_t size)
{
  const uint32_t x = (qm ^ sm) + kc;
  return (int32_t)min(min(x, x - (uint32_t)size), x + (uint32_t)size);
}
// trunc(m * iv / 32768) for m >= 0 (a fan offset, update_tsdf.cu:108-112): the sign of the product is iv's, so the rounding
// toward zero is a per-ray bias -- 32767 for a negative iv -- in front of an arithmetic shift (mad, shift instead of
// multiply, sign, mask, add, shift)
__device__ __forceinline__ int32_t iv_bias(int32_t iv) { return (iv >> 31) & (MATRIX_RESOLUTION - 1); }
__device__ __forceinline__ int32_t trunc15_biased(int32_t m, int32_t iv, int32_t bias) { return (__mul24(m, iv) + bias) >> 15; }

// ---- compacting walk -----------------------------------------------------------------------------------------
// The walks above do the per-sample arithmetic for all lanes and the per-candidate work for the lanes whose sample
// entered a new voxel column — about half of them, at different samples in different lanes, so the expensive part runs
// half empty.  The kernels therefore split the two: a SAMPLE phase that only advances the rays and pushes what a
// candidate needs into a per-wave queue in LDS, and an EMIT phase that pops 64 entries at a time with every lane busy.
// The sample phase needs no division and no multiplication:
//   * |d| * len = q * dist + r is carried as before (len grows by res/2 per step);
//   * the position along the direction of travel is a = s*pos + q (s = sign of d); `gap` counts the millimetres left
//     before trunc(a / res) changes — the column test of update_tsdf.cu:71 becomes "gap <= 0".  Cells of the truncating
//     division are res wide, except the one around zero ([-res+1, res-1]): entering it from below adds res - 1.
// Only for rays marked RAY_SIMPLE by ray_setup_kernel: no int32 wrap (march_steps_fast's domain) AND the whole ray with
// its fans stays inside the window, so the two in_bounds tests per candidate (update_tsdf.cu:73,113) are decided per ray.
constexpr int32_t RAY_FAST = 1, RAY_SIMPLE = 1 << 30;

struct AxisRun
{
  int32_t r, ar, aq; // |d| * len = q * dist + r; per-step increment of (q, r)
  int32_t q;
  int32_t gap;  // x, y only
  int32_t spos; // s * pos
  int32_t sm;   // 0 for d >= 0, -1 for d < 0: proj = ((spos + q) ^ sm) - sm
};
__device__ __forceinline__ void run_init(AxisRun &w, const MarchFrame &f, const RaySetup &r, int32_t d, int32_t pos, int32_t k, bool want_gap)
{
  const uint32_t ad = (uint32_t)(d < 0 ? -d : d);
  w.sm = d < 0 ? -1 : 0;
  w.spos = d < 0 ? -pos : pos;
  const uint32_t inc = ad * (uint32_t)f.half;
  w.aq = (int32_t)(((uint64_t)inc * r.div_m) >> r.div_k);
  w.ar = (int32_t)(inc - (uint32_t)w.aq * (uint32_t)r.distance);
  const uint32_t n = ad * (uint32_t)(1 + k * f.half);
  w.q = (int32_t)(((uint64_t)n * r.div_m) >> r.div_k);
  w.r = (int32_t)(n - (uint32_t)w.q * (uint32_t)r.distance);
  w.gap = 0x3fffffff;
  if (want_gap && ad != 0)
  {
    const int32_t a = w.spos + w.q;
    const int32_t m = div_trunc(a < 0 ? -a : a, f.rM, f.rK, f.res); // |a| / res
    const int32_t rem = (a < 0 ? -a : a) - m * f.res;
    if (a >= 0)
      w.gap = f.res - rem;
    else
      w.gap = m >= 1 ? rem + 1 : f.res - a;
  }
}
// ---- the same sample step with fewer instructions (the march kernels are bound by VALU issue) ---------------------------
// The remainder is kept BIASED by 2^32 - dist, so that "r + ar >= dist" is the carry out of one 32-bit add, and the carry
// feeds the quotient (add with carry) and the gap (subtract with borrow) directly: 8 vector instructions for an x / y axis,
// 4 for z, no branch.  Lanes whose samples are used up keep stepping (their results are masked by the caller), so a wave
// needs no per-lane exec regions in the sample phase.
struct AxisFast
{
  uint32_t rb, ar, bias; // rb = r + bias, bias = 2^32 - dist
  int32_t aq, q, gap, spos, sm;
};
__device__ __forceinline__ AxisFast fast_from(const AxisRun &w, int32_t dist)
{
  AxisFast f;
  f.bias = (uint32_t)0 - (uint32_t)dist;
  f.rb = (uint32_t)w.r + f.bias;
  f.ar = (uint32_t)w.ar;
  f.aq = w.aq;
  f.q = w.q;
  f.gap = w.gap;
  f.spos = w.spos;
  f.sm = w.sm;
  return f;
}
__device__ __forceinline__ bool fast_step(AxisFast &w, int32_t res)
{
  const uint32_t t = w.rb + w.ar;
  const uint32_t c = t < w.rb ? 1u : 0u; // carry: r + ar >= dist
  w.rb = t + (c ? w.bias : 0u);
  const int32_t dq = w.aq + (int32_t)c;
  w.q += dq;
  w.gap -= dq;
  const bool crossed = w.gap <= 0;
  w.gap += crossed ? res : 0;
  return crossed;
}
__device__ __forceinline__ void fast_step_z(AxisFast &w)
{
  const uint32_t t = w.rb + w.ar;
  const uint32_t c = t < w.rb ? 1u : 0u;
  w.rb = t + (c ? w.bias : 0u);
  w.q += w.aq + (int32_t)c;
}
__device__ __forceinline__ int32_t fast_proj(AxisFast &w, bool crossed, int32_t res)
{
  const int32_t a = w.spos + w.q;
  if (crossed && (uint32_t)(a + res - 1) < (uint32_t)(res - 1)) w.gap += res - 1; // entered the cell around zero from below
  return (a ^ w.sm) - w.sm;
}

} // namespace ws
