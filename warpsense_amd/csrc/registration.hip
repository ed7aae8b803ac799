// registration.hip — Point-to-TSDF registration for MI355X (gfx950).
//
// Replaces calc_jacobis_krnl + h_g_e_reduction_krnl + the host reduce() of the reference
// (src/warpsense/cuda/registration.cu:14-257,310-368) and moves the Gauss-Newton update of
// cuda::TSDFRegistration::register_cloud (src/warpsense/tsdf_registration.cpp:55-92) onto the device.
//
//   reg_accumulate_kernel  one fused pass: fixed-point transform, voxel + 6-neighbour gather, gradient,
//                          Jacobian, and the per-lane accumulation of the 21 unique terms of J J^T, the 6
//                          of J v, |v| and the count in int64 registers; wave64 shuffle tree, LDS across
//                          the 4 waves, one 29-word partial per workgroup.  No Jacobian/value/mask round
//                          trip through HBM (the reference writes and re-reads 51 B per point).
//   reg_finish_kernel      one workgroup: sums the partials (exact integer sums -> order independent,
//                          bit-identical to the reference's tree), mirrors h to 6x6, and optionally runs
//                          the 6x6 solve, xi -> SE(3) and the convergence test in double/float like the
//                          host code of the reference.
#include "ws_device.h"

namespace ws
{
constexpr int REG_BLOCKS = 256;  // one workgroup per CU
constexpr int REG_THREADS = 512; // 8 waves
constexpr int REG_TERMS = 29;    // 21 h + 6 g + e + c

struct AccArgs
{
  const int32_t *points;
  uint32_t first;
  uint32_t end; // exclusive
  const uint32_t *map_data;
  MapParams map;
  int32_t res;
  const float *T;         // 16 floats, column-major
  const GnState *state;   // may be null (ws_reg_iterate)
  int64_t *partials;      // [REG_TERMS][REG_BLOCKS]
};

__device__ __forceinline__ int64_t shfl_down_i64(int64_t v, int delta)
{
  int lo = __shfl_down((int)(uint32_t)((uint64_t)v & 0xffffffffull), delta, 64);
  int hi = __shfl_down((int)(uint32_t)((uint64_t)v >> 32), delta, 64);
  return (int64_t)(((uint64_t)(uint32_t)hi << 32) | (uint32_t)lo);
}

__global__ __launch_bounds__(REG_THREADS) void reg_accumulate_kernel(AccArgs a)
{
  __shared__ int64_t lds[REG_THREADS / 64][REG_TERMS];
  if (a.state != nullptr)
  {
    if (a.state->finished || a.state->iterations >= a.state->max_iterations) return;
  }

  // cu_to_int_mat (cuda/util.h:24-35): (int)(float * 32768)
  int32_t M[12];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int i = 0; i < 3; ++i) M[j * 3 + i] = (int32_t)(a.T[j * 4 + i] * (float)MATRIX_RESOLUTION);
  // registration.cu:208: center = (int) translation of the CURRENT transform
  const int32_t cx = (int32_t)a.T[12], cy = (int32_t)a.T[13], cz = (int32_t)a.T[14];
  const int32_t res = a.res;

  int64_t acc[REG_TERMS];
#pragma unroll
  for (int k = 0; k < REG_TERMS; ++k) acc[k] = 0;

  for (uint32_t idx = a.first + blockIdx.x * REG_THREADS + threadIdx.x; idx < a.end; idx += REG_BLOCKS * REG_THREADS)
  {
    const int32_t px = a.points[3 * (size_t)idx + 0], py = a.points[3 * (size_t)idx + 1], pz = a.points[3 * (size_t)idx + 2];
    // cu_transform_point (cuda/util.h:11-22), int32 wrap like the reference
    int32_t qx = wadd(wadd(wadd(wmul(M[0], px), wmul(M[3], py)), wmul(M[6], pz)), M[9]) / MATRIX_RESOLUTION;
    int32_t qy = wadd(wadd(wadd(wmul(M[1], px), wmul(M[4], py)), wmul(M[7], pz)), M[10]) / MATRIX_RESOLUTION;
    int32_t qz = wadd(wadd(wadd(wmul(M[2], px), wmul(M[5], py)), wmul(M[8], pz)), M[11]) / MATRIX_RESOLUTION;
    const int32_t bx = qx / res, by = qy / res, bz = qz / res;
    qx = wsub(qx, cx);
    qy = wsub(qy, cy);
    qz = wsub(qz, cz);
    if (!in_bounds_buffer(a.map, bx, by, bz, -1)) continue; // in_bounds_with_buffer_neg(buf, 1), registration.cu:217

    // all 7 gathers are issued before the first use (the 6 neighbours are in bounds by the test above)
    const uint32_t cur = a.map_data[get_index(a.map, bx, by, bz)];
    const uint32_t xn = a.map_data[get_index(a.map, bx + 1, by, bz)];
    const uint32_t xl = a.map_data[get_index(a.map, bx - 1, by, bz)];
    const uint32_t yn = a.map_data[get_index(a.map, bx, by + 1, bz)];
    const uint32_t yl = a.map_data[get_index(a.map, bx, by - 1, bz)];
    const uint32_t zn = a.map_data[get_index(a.map, bx, by, bz + 1)];
    const uint32_t zl = a.map_data[get_index(a.map, bx, by, bz - 1)];
    if (entry_weight(cur) == 0) continue;

    // registration.cu:233-246
    int32_t gx = 0, gy = 0, gz = 0;
    {
      const int32_t nv = entry_value(xn), lv = entry_value(xl);
      if (entry_weight(xn) != 0 && entry_weight(xl) != 0 && !((nv > 0 && lv < 0) || (nv < 0 && lv > 0))) gx = (nv - lv) / 2;
    }
    {
      const int32_t nv = entry_value(yn), lv = entry_value(yl);
      if (entry_weight(yn) != 0 && entry_weight(yl) != 0 && !((nv > 0 && lv < 0) || (nv < 0 && lv > 0))) gy = (nv - lv) / 2;
    }
    {
      const int32_t nv = entry_value(zn), lv = entry_value(zl);
      if (entry_weight(zn) != 0 && entry_weight(zl) != 0 && !((nv > 0 && lv < 0) || (nv < 0 && lv > 0))) gz = (nv - lv) / 2;
    }
    // point.cross(gradient) in int (math/vector3.h:269-277), widened to long
    int64_t J[6];
    J[0] = wsub(wmul(qy, gz), wmul(qz, gy));
    J[1] = wsub(wmul(qz, gx), wmul(qx, gz));
    J[2] = wsub(wmul(qx, gy), wmul(qy, gx));
    J[3] = gx;
    J[4] = gy;
    J[5] = gz;
    const int64_t v = entry_value(cur);

    // 21 unique terms of J J^T (registration.cu:55-97), row-major upper triangle
    int t = 0;
#pragma unroll
    for (int i = 0; i < 6; ++i)
#pragma unroll
      for (int j = i; j < 6; ++j) acc[t] = wadd64(acc[t], wmul64(J[i], J[j])), ++t;
#pragma unroll
    for (int i = 0; i < 6; ++i) acc[21 + i] = wadd64(acc[21 + i], wmul64(J[i], v));
    acc[27] += (v < 0 ? -v : v);
    acc[28] += 1;
  }

  // wave64 shuffle tree, then LDS across the waves
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < REG_TERMS; ++k)
  {
    int64_t v = acc[k];
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) v = wadd64(v, shfl_down_i64(v, d));
    if (lane == 0) lds[wave][k] = v;
  }
  __syncthreads();
  if (threadIdx.x < REG_TERMS)
  {
    int64_t s = 0;
#pragma unroll
    for (int w = 0; w < REG_THREADS / 64; ++w) s = wadd64(s, lds[w][threadIdx.x]);
    a.partials[(size_t)threadIdx.x * REG_BLOCKS + blockIdx.x] = s;
  }
}

// ---- 6x6 solve + pose update, single lane (tsdf_registration.cpp:63-92, registration/util.h:5-39) ----
__device__ int solve6(double A[6][6], double b[6], double x[6])
{
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double best = fabs(A[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (fabs(A[i][k]) > best)
      {
        best = fabs(A[i][k]);
        piv = i;
      }
    if (best == 0.0) return -1;
    if (piv != k)
    {
      for (int j = 0; j < 6; ++j)
      {
        double t = A[k][j];
        A[k][j] = A[piv][j];
        A[piv][j] = t;
      }
      double t = b[k];
      b[k] = b[piv];
      b[piv] = t;
    }
    for (int i = k + 1; i < 6; ++i)
    {
      double f = A[i][k] / A[k][k];
      for (int j = k; j < 6; ++j) A[i][j] -= f * A[k][j];
      b[i] -= f * b[k];
    }
  }
  for (int i = 5; i >= 0; --i)
  {
    double s = b[i];
    for (int j = i + 1; j < 6; ++j) s -= A[i][j] * x[j];
    x[i] = s / A[i][i];
  }
  return 0;
}

__device__ void gn_update(GnState *st, const int64_t sums[44])
{
  if (st->finished || st->iterations >= st->max_iterations) return;
  const int32_t e = (int32_t)sums[42], c = (int32_t)sums[43];
  st->iterations += 1;
  for (int k = 0; k < 44; ++k) st->sums[k] = sums[k];
  if (c == 0)
  {
    st->finished = 1; // guard: the reference would divide by zero (tsdf_registration.cpp:80)
    return;
  }
  double hf[6][6], gf[6], xi[6];
  const double w = (double)(st->alpha * (float)c);
  for (int r = 0; r < 6; ++r)
  {
    gf[r] = (double)sums[36 + r];
    for (int q = 0; q < 6; ++q) hf[r][q] = (double)sums[q * 6 + r] + (r == q ? w : 0.0);
  }
  if (solve6(hf, gf, xi) != 0)
  {
    st->finished = 1;
    return;
  }
  for (int r = 0; r < 6; ++r) xi[r] = -xi[r];

  // xi_to_transform
  const double theta = sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
  float L[3][3] = {{0, 0, 0}, {0, 0, 0}, {0, 0, 0}};
  if (theta != 0.0)
  {
    const double lx = xi[0] / theta, ly = xi[1] / theta, lz = xi[2] / theta;
    L[0][1] = (float)-lz; L[0][2] = (float)ly;
    L[1][0] = (float)lz;  L[1][2] = (float)-lx;
    L[2][0] = (float)-ly; L[2][1] = (float)lx;
  }
  const float s = (float)sin(theta), omc = (float)(1 - cos(theta));
  float R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
    {
      float ll = 0.f;
      for (int k = 0; k < 3; ++k) ll = __fadd_rn(ll, __fmul_rn(__fmul_rn(omc, L[i][k]), L[k][j]));
      R[i][j] = __fadd_rn(__fadd_rn((i == j ? 1.f : 0.f), __fmul_rn(s, L[i][j])), ll);
    }
  float tr[16];
  for (int i = 0; i < 16; ++i) tr[i] = 0.f;
  tr[15] = 1.f;
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) tr[j * 4 + i] = R[i][j];
    const float oc0 = -(float)st->center[0], oc1 = -(float)st->center[1], oc2 = -(float)st->center[2];
    float shift = __fadd_rn(__fadd_rn(__fmul_rn(R[i][0], oc0), __fmul_rn(R[i][1], oc1)), __fmul_rn(R[i][2], oc2));
    tr[12 + i] = __fadd_rn(__fadd_rn(shift, (float)st->center[i]), (float)xi[3 + i]);
  }
  st->alpha = __fadd_rn(st->alpha, st->it_weight_gradient);
  float out[16];
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i)
    {
      float acc = 0.f;
      for (int k = 0; k < 4; ++k) acc = __fadd_rn(acc, __fmul_rn(tr[k * 4 + i], st->T[j * 4 + k]));
      out[j * 4 + i] = acc;
    }
  for (int i = 0; i < 16; ++i) st->T[i] = out[i];

  const float err = __fdiv_rn((float)e, (float)c);
  if (fabsf(err - st->prev[2]) < st->epsilon && fabsf(err - st->prev[0]) < st->epsilon) st->finished = 1;
  st->prev[0] = st->prev[1];
  st->prev[1] = st->prev[2];
  st->prev[2] = st->prev[3];
  st->prev[3] = err;
}

struct FinishArgs
{
  const int64_t *partials; // [REG_TERMS][REG_BLOCKS]
  int64_t *sums_out;       // 44 or null
  GnState *state;          // null -> no early exit / no solve
  int solve;
};

// maps (i <= j) of the row-major upper triangle to its running index
__device__ __forceinline__ int tri_index(int i, int j)
{
  // i <= j ; rows have 6,5,4,3,2,1 entries
  return i * 6 - (i * (i - 1)) / 2 + (j - i);
}

__global__ __launch_bounds__(256) void reg_finish_kernel(FinishArgs a)
{
  __shared__ int64_t terms[REG_TERMS];
  __shared__ int64_t sums[44];
  if (a.state != nullptr)
  {
    if (a.state->finished || a.state->iterations >= a.state->max_iterations) return;
  }
  // 8 lanes per term: 29 * 8 = 232 active lanes
  const int term = threadIdx.x >> 3, sub = threadIdx.x & 7;
  int64_t s = 0;
  if (term < REG_TERMS)
  {
    for (int b = sub; b < REG_BLOCKS; b += 8) s = wadd64(s, a.partials[(size_t)term * REG_BLOCKS + b]);
  }
  s = wadd64(s, shfl_down_i64(s, 4));
  s = wadd64(s, shfl_down_i64(s, 2));
  s = wadd64(s, shfl_down_i64(s, 1));
  if (term < REG_TERMS && sub == 0) terms[term] = s;
  __syncthreads();
  if (threadIdx.x < 36)
  {
    // Matrix6x6l is column-major: h.at(i,j) = data[j][i] (math/matrix6x6.h:112-115)
    const int j = threadIdx.x / 6, i = threadIdx.x % 6;
    sums[threadIdx.x] = terms[i <= j ? tri_index(i, j) : tri_index(j, i)];
  }
  else if (threadIdx.x < 44)
  {
    int64_t v = terms[21 + (threadIdx.x - 36)];
    if (threadIdx.x >= 42) v = (int64_t)(int32_t)v; // e and c are `int` in the reference (registration.cu:16-21)
    sums[threadIdx.x] = v;
  }
  __syncthreads();
  if (a.sums_out != nullptr && threadIdx.x < 44) a.sums_out[threadIdx.x] = sums[threadIdx.x];
  if (a.solve && a.state != nullptr && threadIdx.x == 0) gn_update(a.state, sums);
}

// the solve alone, fed with externally (all-)reduced sums
__global__ void reg_solve_kernel(GnState *state, const int64_t *sums_dev)
{
  if (threadIdx.x == 0 && blockIdx.x == 0)
  {
    int64_t sums[44];
    for (int k = 0; k < 44; ++k) sums[k] = sums_dev[k];
    sums[42] = (int64_t)(int32_t)sums[42];
    sums[43] = (int64_t)(int32_t)sums[43];
    gn_update(state, sums);
  }
}

int launch_reg_accumulate(ws_reg *r, const ws_map *m, const float *T_dev_or_null, int32_t res, uint32_t flags, size_t first,
                          size_t count, int64_t *sums_dev, bool fused_solve)
{
  ws_context *ctx = r->ctx;
  size_t end = first + count;
  if (end > r->n) end = r->n;
  if (flags & WS_REG_COMPAT_REFERENCE_LAUNCH)
  {
    // <<<128,512>>> covers points 0..65535 only; the reduction drops the last N % 32 points for N >= 128
    size_t lim = r->n;
    if (lim > 65536) lim = 65536;
    if (r->n >= 128)
    {
      size_t red = 32 * (r->n / 32);
      if (red < lim) lim = red;
    }
    if (end > lim) end = lim;
  }
  if (first > end) first = end;

  AccArgs a;
  a.points = r->points;
  a.first = (uint32_t)first;
  a.end = (uint32_t)end;
  a.map_data = m->data[WS_MAP_AVG];
  a.map = m->par[WS_MAP_AVG];
  a.res = res;
  a.T = T_dev_or_null ? T_dev_or_null : r->state->T;
  a.state = T_dev_or_null ? nullptr : r->state;
  a.partials = r->partials;

  FinishArgs f;
  f.partials = r->partials;
  f.sums_out = sums_dev;
  f.state = T_dev_or_null ? nullptr : r->state;
  f.solve = fused_solve ? 1 : 0;

  prof_begin(ctx, WS_K_REG);
  hipLaunchKernelGGL(reg_accumulate_kernel, dim3(REG_BLOCKS), dim3(REG_THREADS), 0, ctx->stream, a);
  hipLaunchKernelGGL(reg_finish_kernel, dim3(1), dim3(256), 0, ctx->stream, f);
  prof_end(ctx, WS_K_REG);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_reg_solve(ws_reg *r, const int64_t *sums_dev)
{
  hipLaunchKernelGGL(reg_solve_kernel, dim3(1), dim3(64), 0, r->ctx->stream, r->state, sums_dev);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

size_t reg_partials_bytes() { return sizeof(int64_t) * REG_TERMS * REG_BLOCKS; }

} // namespace ws
