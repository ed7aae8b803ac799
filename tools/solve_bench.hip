// solve_bench.hip — how long does one wave need for the 6x6 solve of the Gauss-Newton update?  Variants of solve6_wave
// (warpsense_amd/csrc/registration.hip) timed as a dependent chain on one wave.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-fast-math tools/solve_bench.hip -o /tmp/sb && /tmp/sb
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cmath>
#include <vector>

__device__ __forceinline__ double lane_read(double v, int src_lane)
{
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src_lane);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src_lane);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double lane_gather(double v, int src_lane)
{
  const int lo = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2loint(v));
  const int hi = __builtin_amdgcn_ds_bpermute(src_lane << 2, __double2hiint(v));
  return __hiloint2double(hi, lo);
}
template <int K>
__device__ __forceinline__ int row8_share(int v)
{
  const int r = __builtin_amdgcn_update_dpp(v, v, 0x150 + K, 0xf, 0x3, false);
  return __builtin_amdgcn_update_dpp(r, v, 0x150 + 8 + K, 0xf, 0xc, false);
}
template <int K>
__device__ __forceinline__ double row8_share(double v)
{
  return __hiloint2double(row8_share<K>(__double2hiint(v)), row8_share<K>(__double2loint(v)));
}
__device__ __forceinline__ double rcp_refined(double d)
{
  double r = __builtin_amdgcn_rcp(d);
  double e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  e = __builtin_fma(-d, r, 1.0);
  r = __builtin_fma(r, e, r);
  return r;
}
template <bool CHECK>
__device__ __forceinline__ double div_with_rcp(double num, double den, double y, bool checked)
{
  bool vcc, unused;
  const double n_s = __builtin_amdgcn_div_scale(num, den, true, &vcc);
  if (CHECK)
  {
    const double d_s = __builtin_amdgcn_div_scale(num, den, false, &unused);
    if (__any(checked && !(d_s == den))) return num / den;
  }
  const double q0 = n_s * y;
  const double e = __builtin_fma(-den, q0, n_s);
  const double q = __builtin_amdgcn_div_fmas(e, y, q0, vcc);
  return __builtin_amdgcn_div_fixup(q, den, num);
}

// V0: the round-1 solve (row swaps, three gathers per step, plain divisions)
__device__ __forceinline__ int solve_v0(double a, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double pv = lane_read(a, 8 * k + k);
    double best = fabs(pv);
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
    {
      const double v = lane_read(a, 8 * i + k);
      if (fabs(v) > best) { best = fabs(v); pv = v; piv = i; }
    }
    if (best == 0.0) return -1;
    if (k == 5) break;
    const int rr = r == k ? piv : (r == piv ? k : r);
    const double an = lane_gather(a, 8 * rr + c);
    const double rowk = lane_gather(a, 8 * piv + c);
    const double colk = lane_gather(a, 8 * rr + k);
    const double f = colk / pv;
    a = (r > k && c >= k) ? an - f * rowk : an;
  }
  double U[6][6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
    b[i] = lane_read(a, 8 * i + 6);
#pragma unroll
    for (int j = i; j < 6; ++j) U[i][j] = lane_read(a, 8 * i + j);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double t = b[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) t -= U[i][j] * x[j];
    x[i] = t / U[i][i];
  }
  return 0;
}

// V0 with cycle stamps: st[0..5] after the pivot search / after the update of steps 0..4, ... (row swaps, three gathers per step, plain divisions)
__device__ __forceinline__ int solve_v0_prof(double a, double (&x)[6], long long *st)
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double pv = lane_read(a, 8 * k + k);
    double best = fabs(pv);
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
    {
      const double v = lane_read(a, 8 * i + k);
      if (fabs(v) > best) { best = fabs(v); pv = v; piv = i; }
    }
    if (best == 0.0) return -1;
    st[2 * k] = __builtin_readcyclecounter();
    if (k == 5) break;
    const int rr = r == k ? piv : (r == piv ? k : r);
    const double an = lane_gather(a, 8 * rr + c);
    const double rowk = lane_gather(a, 8 * piv + c);
    const double colk = lane_gather(a, 8 * rr + k);
    const double f = colk / pv;
    a = (r > k && c >= k) ? an - f * rowk : an;
    st[2 * k + 1] = __builtin_readcyclecounter();
  }
  double U[6][6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
    b[i] = lane_read(a, 8 * i + 6);
#pragma unroll
    for (int j = i; j < 6; ++j) U[i][j] = lane_read(a, 8 * i + j);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double t = b[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) t -= U[i][j] * x[j];
    x[i] = t / U[i][i];
    st[12 + (5 - i)] = __builtin_readcyclecounter();
  }
  return 0;
}


__global__ __launch_bounds__(64) void prof_kernel(const double *Ab, double *out, long long *stamps)
{
  const int lane = threadIdx.x, r = lane >> 3, c = lane & 7;
  double a = (r < 6 && c < 7) ? Ab[r * 7 + c] : 0.0;
  double x[6] = {0, 0, 0, 0, 0, 0};
  long long st[20];
  for (int i = 0; i < 20; ++i) st[i] = 0;
  for (int rep = 0; rep < 3; ++rep)
  {
    st[18] = __builtin_readcyclecounter();
    st[19] = __builtin_readcyclecounter();
    solve_v0_prof(a + x[0] * 1e-300, x, st);
  }
  if (lane == 0)
  {
    for (int i = 0; i < 20; ++i) stamps[i] = st[i];
    out[0] = x[0];
  }
}

// V1: rows stay, DPP column neighbour, reciprocal of every element refined before the search, MODE bit 0: fallback check
template <bool CHECK, bool PRE_RCP>
__device__ __forceinline__ int solve_v1(double a, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
  int pos[6] = {0, 1, 2, 3, 4, 5};
  uint32_t pivots = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    double y = 0;
    if (PRE_RCP) y = rcp_refined(a);
    int piv = k;
    double pv = lane_read(a, 8 * pos[k] + k);
    double best = fabs(pv);
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
    {
      const double v = lane_read(a, 8 * pos[i] + k);
      if (fabs(v) > best) { best = fabs(v); pv = v; piv = i; }
    }
    if (best == 0.0) return -1;
    int prow = pos[k];
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
      if (i == piv) { prow = pos[i]; pos[i] = pos[k]; }
    pos[k] = prow;
    if (k == 5) break;
    pivots |= 1u << prow;
    const double rowk = lane_gather(a, 8 * prow + c);
    double colk;
    if (k == 0) colk = row8_share<0>(a);
    else if (k == 1) colk = row8_share<1>(a);
    else if (k == 2) colk = row8_share<2>(a);
    else if (k == 3) colk = row8_share<3>(a);
    else colk = row8_share<4>(a);
    const bool below = r < 6 && ((pivots >> r) & 1u) == 0;
    double f;
    if (PRE_RCP) f = div_with_rcp<CHECK>(colk, pv, lane_read(y, 8 * prow + k), below);
    else f = colk / pv;
    if (below && c >= k) a = a - f * rowk;
  }
  double y = 0;
  if (PRE_RCP) y = rcp_refined(a);
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double t = lane_read(a, 8 * pos[i] + 6);
#pragma unroll
    for (int j = i + 1; j < 6; ++j) t -= lane_read(a, 8 * pos[i] + j) * x[j];
    if (PRE_RCP) x[i] = div_with_rcp<CHECK>(t, lane_read(a, 8 * pos[i] + i), lane_read(y, 8 * pos[i] + i), true);
    else x[i] = t / lane_read(a, 8 * pos[i] + i);
  }
  return 0;
}

// V2: V0's elimination (row swaps) + reciprocal of the pivot refined from pv right away (no lane y), back substitution
// with the six diagonal reciprocals refined at once in the lanes
template <bool CHECK>
__device__ __forceinline__ int solve_v2(double a, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double pv = lane_read(a, 8 * k + k);
    double best = fabs(pv);
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
    {
      const double v = lane_read(a, 8 * i + k);
      if (fabs(v) > best) { best = fabs(v); pv = v; piv = i; }
    }
    if (best == 0.0) return -1;
    if (k == 5) break;
    const int rr = r == k ? piv : (r == piv ? k : r);
    const double an = lane_gather(a, 8 * rr + c);
    const double rowk = lane_gather(a, 8 * piv + c);
    const double colk = lane_gather(a, 8 * rr + k);
    const double f = colk / pv;
    a = (r > k && c >= k) ? an - f * rowk : an;
  }
  const double y = rcp_refined(a);
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double t = lane_read(a, 8 * i + 6);
#pragma unroll
    for (int j = i + 1; j < 6; ++j) t -= lane_read(a, 8 * i + j) * x[j];
    x[i] = div_with_rcp<CHECK>(t, lane_read(a, 8 * i + i), lane_read(y, 8 * i + i), true);
  }
  return 0;
}

// V6: V0 with the pivot search kept in the vector unit: the candidates are copied to VGPRs behind an opaque asm, so the
// compare / select chain is v_cmp + v_cndmask instead of v_cmp -> SGPR mask -> s_cselect -> v_mov (a round trip through
// the scalar unit per candidate); the singular-matrix exit is taken once at the end.
__device__ __forceinline__ double in_vgpr(double v)
{
  asm volatile("" : "+v"(v));
  return v;
}
template <bool TOURNAMENT>
__device__ __forceinline__ int solve_v6(double a, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
  int singular = 0;
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double pv;
    if (!TOURNAMENT)
    {
      pv = in_vgpr(lane_read(a, 8 * k + k));
#pragma unroll
      for (int i = k + 1; i < 6; ++i)
      {
        const double v = in_vgpr(lane_read(a, 8 * i + k));
        const bool g = fabs(v) > fabs(pv);
        pv = g ? v : pv;
        piv = g ? i : piv;
      }
    }
    else
    {
      double v[6];
      int ix[6];
#pragma unroll
      for (int i = k; i < 6; ++i)
      {
        v[i] = in_vgpr(lane_read(a, 8 * i + k));
        ix[i] = i;
      }
      // pairwise, the earlier row wins ties: (k,k+1) (k+2,k+3) ... then the winners
#pragma unroll
      for (int stride = 1; stride < 6; stride *= 2)
#pragma unroll
        for (int i = k; i + stride < 6; i += 2 * stride)
        {
          const bool g = fabs(v[i + stride]) > fabs(v[i]);
          v[i] = g ? v[i + stride] : v[i];
          ix[i] = g ? ix[i + stride] : ix[i];
        }
      pv = v[k];
      piv = ix[k];
    }
    singular |= (pv == 0.0) ? 1 : 0;
    if (k == 5) break;
    const int rr = r == k ? piv : (r == piv ? k : r);
    const double an = lane_gather(a, 8 * rr + c);
    const double rowk = lane_gather(a, 8 * piv + c);
    const double colk = lane_gather(a, 8 * rr + k);
    const double f = colk / pv;
    a = (r > k && c >= k) ? an - f * rowk : an;
  }
  if (__builtin_amdgcn_readfirstlane(singular)) return -1;
  double U[6][6], b[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
    b[i] = lane_read(a, 8 * i + 6);
#pragma unroll
    for (int j = i; j < 6; ++j) U[i][j] = lane_read(a, 8 * i + j);
  }
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double t = b[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) t -= U[i][j] * x[j];
    x[i] = t / U[i][i];
  }
  return 0;
}

// num / den with y = rcp_refined(den); sets `bad` when v_div_scale would have rescaled the denominator (the caller then
// repeats the whole solve with plain divisions) -- no branch, no scalar unit on the way
__device__ __forceinline__ double div_deferred(double num, double den, double y, int &bad)
{
  bool vcc, unused;
  const double n_s = __builtin_amdgcn_div_scale(num, den, true, &vcc);
  const double d_s = __builtin_amdgcn_div_scale(num, den, false, &unused);
  bad |= (d_s == den) ? 0 : 1;
  const double q0 = n_s * y;
  const double e = __builtin_fma(-den, q0, n_s);
  const double q = __builtin_amdgcn_div_fmas(e, y, q0, vcc);
  return __builtin_amdgcn_div_fixup(q, den, num);
}

// V8: V6 + reciprocal halves of the divisions moved off the chain (checked at the end, fallback = V6)
template <bool ELIM_RCP, bool BACK_RCP, bool BACK_VGPR>
__device__ __forceinline__ int solve_v8(double a_in, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
  int singular = 0, bad = 0;
  double a = a_in;
#pragma unroll
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double pv = in_vgpr(lane_read(a, 8 * k + k));
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
    {
      const double v = in_vgpr(lane_read(a, 8 * i + k));
      const bool g = fabs(v) > fabs(pv);
      pv = g ? v : pv;
      piv = g ? i : piv;
    }
    singular |= (pv == 0.0) ? 1 : 0;
    if (k == 5) break;
    const int rr = r == k ? piv : (r == piv ? k : r);
    const double an = lane_gather(a, 8 * rr + c);
    const double rowk = lane_gather(a, 8 * piv + c);
    const double colk = lane_gather(a, 8 * rr + k);
    double f;
    if (ELIM_RCP)
    {
      const double y = rcp_refined(pv); // while the gathers are in flight
      int bad_here = 0;
      f = div_deferred(colk, pv, y, bad_here);
      bad |= (r > k && r < 6) ? bad_here : 0;
    }
    else
      f = colk / pv;
    a = (r > k && c >= k) ? an - f * rowk : an;
  }
  if (__builtin_amdgcn_readfirstlane(singular)) return -1;
  const double y = BACK_RCP ? rcp_refined(a) : 0.0;
  double U[6][6], b[6], Y[6];
#pragma unroll
  for (int i = 0; i < 6; ++i)
  {
    b[i] = lane_read(a, 8 * i + 6);
    if (BACK_VGPR) b[i] = in_vgpr(b[i]);
    if (BACK_RCP) Y[i] = lane_read(y, 8 * i + i);
    if (BACK_RCP && BACK_VGPR) Y[i] = in_vgpr(Y[i]);
#pragma unroll
    for (int j = i; j < 6; ++j)
    {
      U[i][j] = lane_read(a, 8 * i + j);
      if (BACK_VGPR) U[i][j] = in_vgpr(U[i][j]);
    }
  }
#pragma unroll
  for (int i = 5; i >= 0; --i)
  {
    double t = b[i];
#pragma unroll
    for (int j = i + 1; j < 6; ++j) t -= U[i][j] * x[j];
    if (BACK_RCP) x[i] = div_deferred(t, U[i][i], Y[i], bad);
    else x[i] = t / U[i][i];
  }
  if (__any(bad != 0)) return solve_v6<false>(a_in, x);
  return 0;
}

template <int V>
__global__ __launch_bounds__(64) void bench_kernel(const double *Ab, double *out, long long *cycles, int iters)
{
  const int lane = threadIdx.x;
  const int r = lane >> 3, c = lane & 7;
  double a0 = (r < 6 && c < 7) ? Ab[r * 7 + c] : 0.0;
  double x[6] = {0, 0, 0, 0, 0, 0};
  double acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
  {
    const double a = a0 + acc * 1e-300; // dependency on the previous solve
    int rc;
    if (V == 0) rc = solve_v0(a, x);
    else if (V == 1) rc = solve_v1<true, true>(a, x);
    else if (V == 2) rc = solve_v1<false, true>(a, x);
    else if (V == 3) rc = solve_v1<false, false>(a, x);
    else if (V == 4) rc = solve_v2<true>(a, x);
    else if (V == 6) rc = solve_v6<false>(a, x);
    else if (V == 8) rc = solve_v8<true, false, false>(a, x);
    else if (V == 9) rc = solve_v8<false, true, false>(a, x);
    else if (V == 10) rc = solve_v8<true, true, false>(a, x);
    else if (V == 11) rc = solve_v8<false, false, true>(a, x);
    else if (V == 12) rc = solve_v8<true, true, true>(a, x);
    else if (V == 7) rc = solve_v6<true>(a, x);
    else rc = solve_v2<false>(a, x);
    acc += x[0] + x[5] + rc;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0)
  {
    cycles[0] = t1 - t0;
    for (int i = 0; i < 6; ++i) out[i] = x[i];
    out[6] = acc;
  }
}

#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int V>
int run(const char *name, const double *dAb, double *dout, long long *dcyc)
{
  const int iters = 2000;
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0));
  CK(hipEventCreate(&e1));
  for (int rep = 0; rep < 2; ++rep)
  {
    CK(hipEventRecord(e0, 0));
    hipLaunchKernelGGL(bench_kernel<V>, dim3(1), dim3(64), 0, 0, dAb, dout, dcyc, iters);
    CK(hipEventRecord(e1, 0));
    CK(hipEventSynchronize(e1));
  }
  float ms = 0;
  CK(hipEventElapsedTime(&ms, e0, e1));
  double out[7];
  long long cyc;
  CK(hipMemcpy(out, dout, sizeof(out), hipMemcpyDeviceToHost));
  CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
  printf("%-64s %7.3f us per solve (%lld counter ticks)  x = %.17g %.17g %.17g\n", name, ms * 1000.0 / iters, cyc / iters, out[0], out[3], out[5]);
  return 0;
}

// V0 cut short: only the first STEPS elimination steps, optionally the back substitution (cost per phase by difference)
template <int STEPS, bool BACK>
__device__ __forceinline__ int solve_cut(double a, double (&x)[6])
{
  const int lane = threadIdx.x & 63, r = lane >> 3, c = lane & 7;
#pragma unroll
  for (int k = 0; k < STEPS; ++k)
  {
    int piv = k;
    double pv = lane_read(a, 8 * k + k);
    double best = fabs(pv);
#pragma unroll
    for (int i = k + 1; i < 6; ++i)
    {
      const double v = lane_read(a, 8 * i + k);
      if (fabs(v) > best) { best = fabs(v); pv = v; piv = i; }
    }
    if (best == 0.0) return -1;
    if (k == 5) break;
    const int rr = r == k ? piv : (r == piv ? k : r);
    const double an = lane_gather(a, 8 * rr + c);
    const double rowk = lane_gather(a, 8 * piv + c);
    const double colk = lane_gather(a, 8 * rr + k);
    const double f = colk / pv;
    a = (r > k && c >= k) ? an - f * rowk : an;
  }
  if (BACK)
  {
    double U[6][6], b[6];
#pragma unroll
    for (int i = 0; i < 6; ++i)
    {
      b[i] = lane_read(a, 8 * i + 6);
#pragma unroll
      for (int j = i; j < 6; ++j) U[i][j] = lane_read(a, 8 * i + j);
    }
#pragma unroll
    for (int i = 5; i >= 0; --i)
    {
      double t = b[i];
#pragma unroll
      for (int j = i + 1; j < 6; ++j) t -= U[i][j] * x[j];
      x[i] = t / U[i][i];
    }
  }
  else
  {
    x[0] = lane_read(a, 9);
    x[5] = lane_read(a, 45);
  }
  return 0;
}

template <int STEPS, bool BACK>
__global__ __launch_bounds__(64) void cut_kernel(const double *Ab, double *out, long long *cycles, int iters)
{
  const int lane = threadIdx.x;
  const int r = lane >> 3, c = lane & 7;
  double a0 = (r < 6 && c < 7) ? Ab[r * 7 + c] : 0.0;
  double x[6] = {0, 0, 0, 0, 0, 0};
  double acc = 0;
  const long long t0 = __builtin_readcyclecounter();
  for (int it = 0; it < iters; ++it)
  {
    const double a = a0 + acc * 1e-300;
    const int rc = solve_cut<STEPS, BACK>(a, x);
    acc += x[0] + x[5] + rc;
  }
  const long long t1 = __builtin_readcyclecounter();
  if (lane == 0)
  {
    cycles[0] = t1 - t0;
    out[6] = acc;
  }
}
template <int STEPS, bool BACK>
int run_cut(const double *dAb, double *dout, long long *dcyc)
{
  const int iters = 2000;
  for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((cut_kernel<STEPS, BACK>), dim3(1), dim3(64), 0, 0, dAb, dout, dcyc, iters);
  long long cyc;
  CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
  printf("steps %d back %d: %lld cycles\n", STEPS, (int)BACK, cyc / iters);
  return 0;
}

int main()
{
  // a Gauss-Newton-like system: SPD-ish matrix with a wide range of magnitudes, pivoting needed in places
  double Ab[42];
  uint64_t s = 88172645463325252ull;
  auto rnd = [&]() { s ^= s << 13; s ^= s >> 7; s ^= s << 17; return (double)(s >> 11) / 9007199254740992.0 - 0.5; };
  double M[6][6];
  for (int i = 0; i < 6; ++i) for (int j = 0; j < 6; ++j) M[i][j] = rnd() * (i < 3 ? 1e6 : 1e2);
  for (int i = 0; i < 6; ++i)
    for (int j = 0; j < 7; ++j)
    {
      double v = 0;
      if (j < 6) for (int k = 0; k < 6; ++k) v += M[k][i] * M[k][j];
      else v = rnd() * 1e7;
      Ab[i * 7 + j] = v;
    }
  double *dAb, *dout;
  long long *dcyc;
  CK(hipMalloc((void **)&dAb, sizeof(Ab)));
  CK(hipMalloc((void **)&dout, 64));
  CK(hipMalloc((void **)&dcyc, 8));
  CK(hipMemcpy(dAb, Ab, sizeof(Ab), hipMemcpyHostToDevice));
  {
    long long *dst, st[20];
    CK(hipMalloc((void **)&dst, sizeof(st)));
    hipLaunchKernelGGL(prof_kernel, dim3(1), dim3(64), 0, 0, dAb, dout, dst);
    CK(hipMemcpy(st, dst, sizeof(st), hipMemcpyDeviceToHost));
    printf("stamp overhead %lld\n", st[19] - st[18]);
    long long prev = st[19];
    const char *names[18] = {"search0", "update0", "search1", "update1", "search2", "update2", "search3", "update3", "search4", "update4", "search5", "-", "x5", "x4", "x3", "x2", "x1", "x0"};
    for (int i = 0; i < 18; ++i)
    {
      if (i == 11) continue;
      printf("%s %lld  ", names[i], st[i] - prev);
      prev = st[i];
    }
    printf("\n");
  }
  if (run_cut<0, false>(dAb, dout, dcyc) || run_cut<1, false>(dAb, dout, dcyc) || run_cut<2, false>(dAb, dout, dcyc) || run_cut<3, false>(dAb, dout, dcyc) ||
      run_cut<4, false>(dAb, dout, dcyc) || run_cut<5, false>(dAb, dout, dcyc) || run_cut<6, false>(dAb, dout, dcyc) || run_cut<0, true>(dAb, dout, dcyc) ||
      run_cut<6, true>(dAb, dout, dcyc))
    return 1;
  if (run<0>("V0 row swaps, 3 gathers, plain divisions", dAb, dout, dcyc)) return 1;
  if (run<1>("V1 rows stay, DPP, lane reciprocals, fallback check", dAb, dout, dcyc)) return 1;
  if (run<2>("V1 without the fallback check", dAb, dout, dcyc)) return 1;
  if (run<3>("V1 rows stay, DPP, plain divisions", dAb, dout, dcyc)) return 1;
  if (run<6>("V6 = V0 with the pivot search in the vector unit", dAb, dout, dcyc)) return 1;
  if (run<7>("V6 with a tournament instead of a chain", dAb, dout, dcyc)) return 1;
  if (run<8>("V8 elimination reciprocal hoisted, deferred check", dAb, dout, dcyc)) return 1;
  if (run<9>("V8 back substitution reciprocals at once, deferred check", dAb, dout, dcyc)) return 1;
  if (run<10>("V8 both", dAb, dout, dcyc)) return 1;
  if (run<11>("V8 back substitution operands in VGPRs only", dAb, dout, dcyc)) return 1;
  if (run<12>("V8 all three", dAb, dout, dcyc)) return 1;
  if (run<4>("V2 = V0 elimination + diagonal reciprocals at once, check", dAb, dout, dcyc)) return 1;
  if (run<5>("V2 without the fallback check", dAb, dout, dcyc)) return 1;
  return 0;
}
