// ws_device.h — device-side fixed-point helpers shared by the kernels (gfx950).
//
// Integer semantics follow the reference exactly: `int` expressions wrap (written through unsigned so
// the compiler cannot assume "no overflow"), divisions truncate toward zero, `long` is int64.
#pragma once

#include "ws_internal.h"

namespace ws
{
__device__ __forceinline__ int32_t wmul(int32_t a, int32_t b) { return (int32_t)((uint32_t)a * (uint32_t)b); }
__device__ __forceinline__ int32_t wadd(int32_t a, int32_t b) { return (int32_t)((uint32_t)a + (uint32_t)b); }
__device__ __forceinline__ int32_t wsub(int32_t a, int32_t b) { return (int32_t)((uint32_t)a - (uint32_t)b); }
__device__ __forceinline__ int64_t wmul64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a * (uint64_t)b); }
__device__ __forceinline__ int64_t wadd64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a + (uint64_t)b); }
__device__ __forceinline__ int64_t wsub64(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
__device__ __forceinline__ int32_t iabs32(int32_t a) { return a < 0 ? wsub(0, a) : a; }

// exact floor(x / d) for 0 <= x < 2^31 by multiply-shift: M = ceil(2^k / d), k = 31 + ceil(log2 d)
// (error e = M*d - 2^k < d <= 2^(k-31), so x*e < 2^k for every x <= 2^31)
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
// C-style truncating 64-bit division through one double division when both operands are below 2^53 (always, for the ray
// set-up's lengths and normalised directions; the plain division otherwise).  Exact: the operands convert exactly, the
// correctly rounded quotient is off by less than |q| 2^-53 < 1 / |den|, i.e. by less than the distance of a non-integer
// num / den to the next integer, and an integer quotient is representable -- so truncating the double gives trunc(num / den).
// ~40 instructions instead of the ~120 of the expanded 64-bit division.
__device__ __forceinline__ int64_t div_trunc_i64(int64_t num, int64_t den)
{
  const uint64_t an = num < 0 ? (uint64_t)0 - (uint64_t)num : (uint64_t)num;
  const uint64_t ad = den < 0 ? (uint64_t)0 - (uint64_t)den : (uint64_t)den;
  if (an < (1ull << 53) && ad < (1ull << 53)) return (int64_t)((double)num / (double)den);
  return num / den;
}

// make_fastdiv on the device without the 64-bit division and the search loop (same M, k)
__device__ __forceinline__ FastDiv make_fastdiv_dev(int32_t d)
{
  FastDiv f;
  f.d = d;
  const int l = d > 1 ? 32 - __clz(d - 1) : 0; // smallest l with 2^l >= d
  f.k = 31 + l;
  const uint64_t p = 1ull << f.k; // k <= 62, p / d < 2^32
  uint64_t q = (uint64_t)((double)p / (double)d); // floor(p / d) or one more
  int64_t r = (int64_t)(p - q * (uint64_t)d);
  if (r < 0)
  {
    q -= 1;
    r += d;
  }
  f.M = q + (r != 0 ? 1 : 0);
  return f;
}

// C-style truncating division of any int32 by the prepared positive divisor
__device__ __forceinline__ int32_t div_trunc(int32_t x, uint64_t M, int32_t k, int32_t d)
{
  // ax <= 2^31 (|INT_MIN| included: e < d <= 2^(k-31) gives ax * e < 2^k for ax = 2^31 as well) and M < 2^32, so the
  // product fits 64 bits and the quotient is exact: no special case, no branch
  const uint32_t ax = x < 0 ? (uint32_t)0 - (uint32_t)x : (uint32_t)x;
  const uint32_t q = (uint32_t)(((uint64_t)ax * M) >> k);
  return x < 0 ? (int32_t)((uint32_t)0 - q) : (int32_t)q;
}
__device__ __forceinline__ int32_t div_trunc(int32_t x, const FastDiv &f) { return div_trunc(x, f.M, f.k, f.d); }

// Vector3<int>::l2norm — include/warpsense/math/vector3.h:318-330: int(sqrtf(float(int sum))).
// sqrtf, NOT __fsqrt_rn: without OCML_BASIC_ROUNDED_OPERATIONS the latter is the 1-ulp hardware approximation, and the
// truncation turns one ulp into a different integer (sqrt(19838114) -> 4453 instead of 4454: one ray in 65 536).
__device__ __forceinline__ int32_t l2norm_i(int32_t x, int32_t y, int32_t z)
{
  int32_t sq = wadd(wadd(wmul(x, x), wmul(y, y)), wmul(z, z));
  if (sq < 0) return 0; // NaN -> 0 like the reference's cvt.rzi.s32.f32 (and v_cvt_i32_f32); explicit: fptosi of NaN is poison to the compiler
  return (int32_t)sqrtf((float)sq);
}
// Vector3<long>::l2norm — same header, T = long.  A wrapped (negative) sum is NaN after sqrtf; the reference's CUDA code
// converts it with cvt.rzi.s64.f32 (= __float2ll_rz), which gives 0x8000000000000000 for NaN (the 32-bit conversion
// above gives 0, like v_cvt_i32_f32).  gfx950 has no f32 -> i64 instruction and the compiler's expansion is not
// specified for NaN, so the case is decided here (oracle/ws_oracle.c:l2norm_l does the same).
__device__ __forceinline__ int64_t l2norm_l(int64_t x, int64_t y, int64_t z)
{
  int64_t sq = wadd64(wadd64(wmul64(x, x), wmul64(y, y)), wmul64(z, z));
  if (sq < 0) return INT64_MIN;
  return (int64_t)sqrtf((float)sq);
}

// TSDFEntry — include/map/tsdf.h:16-23
__device__ __forceinline__ uint32_t pack_entry(int32_t value, int32_t weight)
{
  return ((uint32_t)value & 0xffffu) | ((uint32_t)weight << 16);
}
__device__ __forceinline__ int32_t entry_value(uint32_t raw) { return (int32_t)(int16_t)(raw & 0xffffu); }
__device__ __forceinline__ int32_t entry_weight(uint32_t raw) { return (int32_t)(int16_t)(raw >> 16); }

// overflow() — include/warpsense/cuda/device_map.h:14-30
__device__ __forceinline__ int32_t ring(int32_t val, int32_t max)
{
  if (val >= 2 * max) return val - 2 * max;
  if (val >= max) return val - max;
  return val;
}

// DeviceMap::get_index — device_map.h:93-101, z fastest; 64-bit product so 2049^3 does not wrap
__device__ __forceinline__ int64_t get_index(const MapParams &m, int32_t x, int32_t y, int32_t z)
{
  int32_t xi = ring(x - m.pos[0] + m.offset[0] + m.size[0], m.size[0]);
  int32_t yi = ring(y - m.pos[1] + m.offset[1] + m.size[1], m.size[1]);
  int32_t zi = ring(z - m.pos[2] + m.offset[2] + m.size[2], m.size[2]);
  // size[0] * size[1] < 2^31 (checked by ws_map_create): one 32-bit multiply, then v_mad_i64_i32
  const int32_t row = xi * m.size[1] + yi;
  return (int64_t)row * (int64_t)m.size[2] + zi;
}

// DeviceMap::in_bounds — device_map.h:109-114
__device__ __forceinline__ bool in_bounds(const MapParams &m, int32_t x, int32_t y, int32_t z)
{
  return iabs32(wsub(x, m.pos[0])) <= m.size[0] / 2 && iabs32(wsub(y, m.pos[1])) <= m.size[1] / 2 &&
         iabs32(wsub(z, m.pos[2])) <= m.size[2] / 2;
}
// in_bounds_with_buffer_pos / _neg — device_map.h:116-128 (unsigned compare: `buffer` is size_t there)
__device__ __forceinline__ bool in_bounds_buffer(const MapParams &m, int32_t x, int32_t y, int32_t z, int64_t buffer)
{
  uint64_t ax = (uint64_t)(int64_t)iabs32(wsub(x, m.pos[0]));
  uint64_t ay = (uint64_t)(int64_t)iabs32(wsub(y, m.pos[1]));
  uint64_t az = (uint64_t)(int64_t)iabs32(wsub(z, m.pos[2]));
  return ax <= (uint64_t)((int64_t)(m.size[0] / 2) + buffer) && ay <= (uint64_t)((int64_t)(m.size[1] / 2) + buffer) &&
         az <= (uint64_t)((int64_t)(m.size[2] / 2) + buffer);
}

// weight ramp — update_tsdf.cu:90-94
__device__ __forceinline__ int32_t tsdf_weight(int32_t value, int32_t tau, int32_t weight_epsilon)
{
  int32_t weight = WEIGHT_RESOLUTION;
  if (value < -weight_epsilon) weight = WEIGHT_RESOLUTION * (tau + value) / (tau - weight_epsilon);
  return weight;
}

// the same with the division by the scan constant (tau - weight_epsilon) prepared (make_fastdiv): 64 * (tau + value) >= 0
__device__ __forceinline__ int32_t tsdf_weight(int32_t value, int32_t tau, int32_t weight_epsilon, const FastDiv &wdiv)
{
  int32_t weight = WEIGHT_RESOLUTION;
  if (value < -weight_epsilon) weight = div_trunc(WEIGHT_RESOLUTION * (tau + value), wdiv);
  return weight;
}
// weight == 0 (update_tsdf.cu:96: the candidate is skipped) without dividing: 64 * (tau + value) < tau - weight_epsilon
__device__ __forceinline__ bool tsdf_weight_is_zero(int32_t value, int32_t tau, int32_t weight_epsilon)
{
  return value < -weight_epsilon && WEIGHT_RESOLUTION * (tau + value) < tau - weight_epsilon;
}

// cu_avg_tsdf_krnl body — update_tsdf.cu:19-34; returns the updated existing entry
__device__ __forceinline__ uint32_t integrate_entry(uint32_t existing, uint32_t fresh, int32_t max_weight)
{
  int32_t nv = entry_value(fresh), nw = entry_weight(fresh);
  int32_t ev = entry_value(existing), ew = entry_weight(existing);
  if (nw > 0 && ew > 0)
  {
    // (ev * ew + nv * nw) / (ew + nw), C division -- without the compiler's 43-instruction expansion of a general 32-bit signed
    // division (four of them per thread and tile were a third of the fused resolve's vector instructions).  The quotient is a
    // weighted mean of two int16 values, so |q| <= 32768, the divisor is below 2^16 and |num| below 2^31: one float reciprocal
    // (v_rcp_f32, 1 ulp) gives |num| / den to within 32769 * 2^-22 < 0.01, i.e. a truncated estimate that is off by one at most,
    // and one exact remainder decides.  The result is the exact integer quotient for every input of that domain.
    const int32_t num = ev * ew + nv * nw, den = ew + nw;
    const uint32_t an = (uint32_t)(num < 0 ? -num : num);
    uint32_t q = (uint32_t)((float)an * __builtin_amdgcn_rcpf((float)den));
    const int32_t r = (int32_t)(an - q * (uint32_t)den);
    q = r < 0 ? q - 1u : (r >= den ? q + 1u : q);
    const int32_t v = num < 0 ? -(int32_t)q : (int32_t)q;
    int32_t w = min(max_weight, ew + nw);
    return pack_entry(v, w);
  }
  if (nw != 0 && ew <= 0) return fresh;
  return existing;
}

} // namespace ws
