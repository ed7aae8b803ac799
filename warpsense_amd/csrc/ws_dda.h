// ws_dda.h — the ray march of update_tsdf.cu:65-76 as a walk over COLUMN CHANGES instead of samples (round 5).
//
// The reference samples a ray every res/2 millimetres and acts on a sample only when its (x, y) voxel column differs from
// the previous sample's (update_tsdf.cu:71).  Rounds 1-4 stepped every sample (20 vector instructions for the three axes,
// branch-free) and pushed the samples that qualify through a per-wave queue in LDS, 64 at a time, to the part that does the
// work: about two samples and one push per candidate.  This header walks from one qualifying sample to the next directly --
// the steps at which an axis enters a new voxel are a Bresenham sequence -- and computes the sample's position from its step
// number by one exact multiply-shift per axis: one loop iteration per candidate, in the lane of its ray, no queue.
//
// Exactness.  On one axis, mirrored so that the position grows: a(k) = s*pos + q(k), q(k) = floor(|d| * len_k / dist),
// len_k = 1 + k * half (update_tsdf.cu:67-69; s = sign of d; the sample's position is s * a).  trunc(a / res) changes when a
// reaches a BOUNDARY: m * res for m >= 1, and -m * res + 1 for m >= 1 (the cell of index 0 is [-res + 1, res - 1], 2 res - 1
// wide; every other cell is res wide).  The first step with a(k) >= b is
//     K(b) = ceil((Q * dist - |d|) / (|d| * half)),  Q = b - s*pos          (Q >= 1 for a boundary ahead of the sample k = 0)
// because q(k) >= Q  <=>  |d| * len_k >= Q * dist  <=>  len_k >= ceil(Q * dist / |d|), and ceil((ceil(y) - 1) / h) = ceil((y - 1) / h).
// With D = |d| * half and N = Q * dist - |d| = K * D - rho (0 <= rho < D), the next boundary (Q + res) adds W = res * dist =
// wq * D + wr to N: K' = K + wq + (rho < wr), rho' = rho - wr (+ D if it borrowed) -- three instructions, no division.  The
// one irregular spacing (from -res + 1 to res: W' = (2 res - 1) * dist) is taken by a slow path at the step it happens.
// Everything stays below 2^31 for the rays the kernels send here (RAY_SIMPLE, ws_march.h: no int32 wrap anywhere on the ray).
//
// Plain C++ (compiled by hipcc for the kernels and by g++ for tools/dda_check.cpp, which holds it against the sample-by-sample
// walk on millions of random rays).
#pragma once

#include <cstdint>

#if defined(__HIPCC__)
#define WS_DDA_FN __host__ __device__ __forceinline__
#else
#define WS_DDA_FN inline
#endif

namespace ws
{
constexpr uint32_t DDA_NEVER = 0xffffffffu;

WS_DDA_FN uint32_t dda_mulhi(uint32_t a, uint32_t b)
{
#if defined(__HIP_DEVICE_COMPILE__)
  return __umulhi(a, b);
#else
  return (uint32_t)(((uint64_t)a * b) >> 32);
#endif
}

// per-ray constants of the position q(k) = floor(|d| * len_k / dist): exact for |d| * len_k < 2^31 with the 32-bit multiplier
// M = ceil(2^(31 + l) / dist), 2^(l-1) < dist <= 2^l (the ray set-up's div_m / div_k: M < 2^32 for dist >= 2)
struct DdaRay
{
  uint32_t M32;
  int32_t sh; // div_k - 32 >= 0
};
WS_DDA_FN uint32_t dda_q(uint32_t ad, int32_t len, const DdaRay &r) { return dda_mulhi(ad * (uint32_t)len, r.M32) >> r.sh; }
// ... from the product n = |d| * len_k = (|d| * half) * k + |d| itself (one multiply-add per axis and step in the walk's loop)
WS_DDA_FN uint32_t dda_qn(uint32_t n, const DdaRay &r) { return dda_mulhi(n, r.M32) >> r.sh; }

struct DdaAxis
{
  uint32_t K;   // step at which the axis enters its next voxel (DDA_NEVER: never)
  uint32_t rho; // K * D - N
  uint32_t wq, wr, D;
  uint32_t Ksp; // the step at which the boundary -res + 1 is crossed (after it the spacing is 2 res - 1 once); DDA_NEVER: not ahead
};

// ceil(N / D) and the remainder K * D - N for 0 <= N < 2^40, 0 < D < 2^31
WS_DDA_FN void dda_ceil_div(int64_t N, uint32_t D, uint32_t &K, uint32_t &rho)
{
  // one double division: both operands are exact, the quotient is off by less than one
  int64_t k = (int64_t)((double)N / (double)D);
  int64_t rem = N - k * (int64_t)D;
  if (rem < 0)
  {
    k -= 1;
    rem += D;
  }
  else if (rem >= (int64_t)D)
  {
    k += 1;
    rem -= D;
  }
  // floor -> ceil
  if (rem != 0)
  {
    k += 1;
    rem = (int64_t)D - rem;
  }
  K = (uint32_t)k;
  rho = (uint32_t)rem;
}

// The walk stands at the sample `k_at` (its mirrored position a = spos + q): the next boundary ahead and the step that reaches it.
//   ad = |d|, spos = s * pos, q = q(k_at), dist, res, half as in the reference
WS_DDA_FN void dda_axis_init(DdaAxis &w, uint32_t ad, int32_t spos, uint32_t q, int32_t dist, int32_t res, int32_t half)
{
  w.K = w.Ksp = DDA_NEVER;
  w.rho = 0;
  w.wq = w.wr = 0;
  w.D = 1;
  if (ad == 0) return;
  w.D = ad * (uint32_t)half;
  const uint32_t W = (uint32_t)res * (uint32_t)dist;
  w.wq = W / w.D;
  w.wr = W - w.wq * w.D;
  const int32_t a = spos + (int32_t)q;
  int32_t b; // the next boundary: smallest boundary > a
  if (a >= 0)
  {
    b = (a / res + 1) * res;
  }
  else
  {
    const int32_t m = (-a) / res; // a in (-(m + 1) res, -m res]
    b = m >= 1 ? -m * res + 1 : res;
    if (m >= 1)
    {
      // the boundary -res + 1 lies ahead: the step that crosses it
      uint32_t rs;
      dda_ceil_div((int64_t)(-res + 1 - spos) * dist - (int64_t)ad, w.D, w.Ksp, rs);
    }
  }
  dda_ceil_div((int64_t)(b - spos) * dist - (int64_t)ad, w.D, w.K, w.rho);
}

// the axis has just entered a new voxel at step w.K: the step of the next boundary
WS_DDA_FN void dda_axis_advance(DdaAxis &w)
{
  const uint32_t t = w.rho - w.wr;
  const uint32_t borrow = w.rho < w.wr ? 1u : 0u;
  w.K = w.K + w.wq + borrow;
  w.rho = t + (borrow ? w.D : 0u);
}
// ... when the boundary just crossed was -res + 1 (step == Ksp): the next one is `res`, 2 res - 1 further
WS_DDA_FN void dda_axis_after_zero_cell(DdaAxis &w, uint32_t ad, int32_t spos, int32_t dist, int32_t res)
{
  dda_ceil_div((int64_t)(res - spos) * dist - (int64_t)ad, w.D, w.K, w.rho);
  w.Ksp = DDA_NEVER;
}

} // namespace ws
