// cpu_baseline.cpp — PORT of the reference's CPU path (test/bench infrastructure, NOT product code).
//
// A faithful restatement of what `fastsense` / warpsense_cpu.launch executes on the host
// (SURVEY.md §3.2, §8a row a14), keeping the reference's data structures so the timing is
// comparable: per-thread std::unordered_map scatter with the reference hash
// (include/warpsense/types.h:24-31), merge pass, OpenMP.  The reference sources themselves cannot be
// compiled here (Eigen, PCL, HighFive, ROS absent), hence bench.py reports cpu_baseline.kind = "port".
//
//   wscpu_update_tsdf(threads=1)   <- src/cpu/update_tsdf.cpp:397-564  (what fastsense.cpp:172 calls)
//   wscpu_update_tsdf(threads=0)   <- src/cpu/update_tsdf.cpp:566-724  (omp_get_max_threads())
//   wscpu_register_cloud           <- src/cpu/registration.cpp:14-177
//
// Differences to the CUDA semantics that the reference CPU path itself has (kept on purpose):
// pos = scanner_pos*res without +res/2 (:410,:578), Eigen norm() = (int)sqrt((double)sq),
// hash-map rule |new|<|old| || old.w<0 (:508-512), zero-distance / zero-interpolation guards.
#include <omp.h>

#include <cmath>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <unordered_map>
#include <vector>

namespace
{
constexpr int MR = 32768;
constexpr int WR = 64;

struct P3
{
  int x, y, z;
  bool operator==(const P3 &o) const { return x == o.x && y == o.y && z == o.z; }
};
struct P3Hash // include/warpsense/types.h:24-31
{
  std::size_t operator()(const P3 &p) const noexcept
  {
    long long v = ((long long)p.x << 32) ^ ((long long)p.y << 16) ^ (long long)p.z;
    return std::hash<long long>()(v);
  }
};
struct Entry
{
  int16_t value, weight;
};

struct LocalMap // HDF5LocalMap's in-memory ring buffer, include/map/hdf5_local_map.h:124-287
{
  int size[3], pos[3], offset[3];
  uint32_t *data;
  bool in_bounds(int x, int y, int z) const
  {
    return std::abs(x - pos[0]) <= size[0] / 2 && std::abs(y - pos[1]) <= size[1] / 2 && std::abs(z - pos[2]) <= size[2] / 2;
  }
  long index(int x, int y, int z) const
  {
    auto wrap = [](long v, long m) { return ((v % m) + m) % m; };
    long xi = wrap((long)x - pos[0] + offset[0] + size[0], size[0]);
    long yi = wrap((long)y - pos[1] + offset[1] + size[1], size[1]);
    long zi = wrap((long)z - pos[2] + offset[2] + size[2], size[2]);
    return (xi * size[1] + yi) * size[2] + zi;
  }
  Entry get(int x, int y, int z) const
  {
    uint32_t r = data[index(x, y, z)];
    return {(int16_t)(r & 0xffff), (int16_t)(r >> 16)};
  }
  void set(int x, int y, int z, Entry e) { data[index(x, y, z)] = (uint32_t)(uint16_t)e.value | ((uint32_t)(uint16_t)e.weight << 16); }
};

inline int norm_i(long x, long y, long z) { return (int)std::sqrt((double)(int)(x * x + y * y + z * z)); } // Eigen Vector3i::norm()
inline long norm_l(long x, long y, long z) { return (long)std::sqrt((double)(x * x + y * y + z * z)); }     // Matrix<long,3,1>::norm()

int solve6(double A[6][6], double b[6], double x[6])
{
  for (int k = 0; k < 6; ++k)
  {
    int piv = k;
    double best = std::fabs(A[k][k]);
    for (int i = k + 1; i < 6; ++i)
      if (std::fabs(A[i][k]) > best) { best = std::fabs(A[i][k]); piv = i; }
    if (best == 0.0) return -1;
    if (piv != k)
    {
      for (int j = 0; j < 6; ++j) std::swap(A[k][j], A[piv][j]);
      std::swap(b[k], b[piv]);
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

void xi_to_transform(const double xi[6], const int center[3], float T[16]) // registration/util.h:5-39
{
  double theta = std::sqrt(xi[0] * xi[0] + xi[1] * xi[1] + xi[2] * xi[2]);
  float L[3][3] = {};
  if (theta != 0.0)
  {
    double lx = xi[0] / theta, ly = xi[1] / theta, lz = xi[2] / theta;
    L[0][1] = (float)-lz; L[0][2] = (float)ly;
    L[1][0] = (float)lz;  L[1][2] = (float)-lx;
    L[2][0] = (float)-ly; L[2][1] = (float)lx;
  }
  float s = (float)std::sin(theta), omc = (float)(1 - std::cos(theta));
  float R[3][3];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j)
    {
      float ll = 0.f;
      for (int k = 0; k < 3; ++k) ll += (omc * L[i][k]) * L[k][j];
      R[i][j] = ((i == j ? 1.f : 0.f) + s * L[i][j]) + ll;
    }
  for (int i = 0; i < 16; ++i) T[i] = 0.f;
  T[15] = 1.f;
  for (int i = 0; i < 3; ++i)
  {
    for (int j = 0; j < 3; ++j) T[j * 4 + i] = R[i][j];
    float shift = (R[i][0] * -(float)center[0] + R[i][1] * -(float)center[1]) + R[i][2] * -(float)center[2];
    T[12 + i] = (shift + (float)center[i]) + (float)xi[3 + i];
  }
}
} // namespace

extern "C" {

// src/cpu/update_tsdf.cpp:397-564 (threads==1) and :566-724 (threads<=0 -> omp_get_max_threads()).
// scanner_pos is in VOXEL units (the callee multiplies by res, :410/:578).
int wscpu_update_tsdf(const int32_t *size, const int32_t *pos_, const int32_t *offset, uint32_t *data,
                      const int32_t *xyz, size_t n, const int32_t *scanner_pos, const int32_t *up, int tau,
                      int max_weight, int res, int threads)
{
  LocalMap buffer;
  for (int k = 0; k < 3; ++k) { buffer.size[k] = size[k]; buffer.pos[k] = pos_[k]; buffer.offset[k] = offset[k]; }
  buffer.data = data;

  float angle = 45.f / 128.f;
  int dz_per_distance = std::tan(angle / 180 * M_PI) / 2.0 * MR;
  int weight_epsilon = tau / 10;
  int thread_count = threads > 0 ? threads : omp_get_max_threads();
  const bool single_overload = (threads == 1); // the std::vector<Point> overload also checks in_bounds(point/res), :425-429

  std::vector<std::unordered_map<P3, Entry, P3Hash>> values(thread_count);
  const int pos[3] = {scanner_pos[0] * res, scanner_pos[1] * res, scanner_pos[2] * res};

#pragma omp parallel num_threads(thread_count)
  {
    int current_thread = omp_get_thread_num();
    auto &local_values = values[current_thread];

#pragma omp for schedule(static)
    for (long pi = 0; pi < (long)n; ++pi)
    {
      const int *point = xyz + 3 * pi;
      int dir[3] = {point[0] - pos[0], point[1] - pos[1], point[2] - pos[2]};
      int distance = norm_i(dir[0], dir[1], dir[2]);
      if (distance == 0) continue;
      if (single_overload && !buffer.in_bounds(point[0] / res, point[1] / res, point[2] / res)) continue;

      long nd[3], c1[3], iv[3];
      for (int k = 0; k < 3; ++k) nd[k] = ((long)dir[k] * MR) / distance;
      c1[0] = (nd[1] * up[2] - nd[2] * up[1]) / MR;
      c1[1] = (nd[2] * up[0] - nd[0] * up[2]) / MR;
      c1[2] = (nd[0] * up[1] - nd[1] * up[0]) / MR;
      iv[0] = nd[1] * c1[2] - nd[2] * c1[1];
      iv[1] = nd[2] * c1[0] - nd[0] * c1[2];
      iv[2] = nd[0] * c1[1] - nd[1] * c1[0];
      long inorm = norm_l(iv[0], iv[1], iv[2]);
      if (inorm == 0) continue;
      for (int k = 0; k < 3; ++k) iv[k] = (iv[k] * MR) / inorm;

      int prev[3] = {0, 0, 0};
      for (int len = 1; len <= distance + tau; len += res / 2)
      {
        int proj[3], index[3];
        for (int k = 0; k < 3; ++k) { proj[k] = pos[k] + dir[k] * len / distance; index[k] = proj[k] / res; }
        if (index[0] == prev[0] && index[1] == prev[1]) continue;
        prev[0] = index[0]; prev[1] = index[1]; prev[2] = index[2];
        if (!buffer.in_bounds(index[0], index[1], index[2])) continue;

        int tc[3] = {index[0] * res + res / 2, index[1] * res + res / 2, index[2] * res + res / 2};
        long long value = norm_i(point[0] - tc[0], point[1] - tc[1], point[2] - tc[2]);
        value = std::min(value, (long long)tau);
        if (len > distance) value = -value;
        int weight = WR;
        if (value < -weight_epsilon) weight = WR * (tau + value) / (tau - weight_epsilon);
        if (weight == 0) continue;
        Entry object{(int16_t)value, (int16_t)weight};
        int delta_z = dz_per_distance * len / MR;
        int iter_steps = (delta_z * 2) / res + 1;
        int mid = delta_z / res;
        int lowest[3];
        for (int k = 0; k < 3; ++k) lowest[k] = proj[k] - (int)(((long)delta_z * iv[k]) / MR);

        for (int step = 0; step < iter_steps; ++step)
        {
          P3 idx;
          idx.x = (lowest[0] + (int)(((long)(step * res) * iv[0]) / MR)) / res;
          idx.y = (lowest[1] + (int)(((long)(step * res) * iv[1]) / MR)) / res;
          idx.z = (lowest[2] + (int)(((long)(step * res) * iv[2]) / MR)) / res;
          if (!buffer.in_bounds(idx.x, idx.y, idx.z)) continue;
          Entry tmp = object;
          if (step != mid) tmp.weight = (int16_t)(tmp.weight * -1);
          auto existing = local_values.try_emplace(idx, tmp);
          if (!existing.second && (std::llabs(value) < std::abs((int)existing.first->second.value) || existing.first->second.weight < 0))
            existing.first->second = tmp;
        }
      }
    }
    // "#pragma omp for" ends with an implicit barrier; the reference adds an explicit one (:518/:677)
#pragma omp barrier
    for (auto &map_entry : local_values)
    {
      bool skip = false;
      for (int i = 0; i < thread_count; i++)
      {
        if (i == current_thread) continue;
        auto iter = values[i].find(map_entry.first);
        if (iter != values[i].end() && std::fabs((float)iter->second.value) < std::fabs((float)map_entry.second.value)) { skip = true; break; }
      }
      if (skip) continue;
      const P3 &index = map_entry.first;
      int value = map_entry.second.value, weight = map_entry.second.weight;
      Entry entry = buffer.get(index.x, index.y, index.z);
      if (weight > 0 && entry.weight > 0)
      {
        int16_t v = (int16_t)((entry.value * entry.weight + value * weight) / (entry.weight + weight));
        int16_t w = (int16_t)std::min(max_weight, entry.weight + weight);
        buffer.set(index.x, index.y, index.z, {v, w});
      }
      else if (weight != 0 && entry.weight <= 0)
      {
        buffer.set(index.x, index.y, index.z, {(int16_t)value, (int16_t)weight});
      }
    }
  }
  return thread_count;
}

// src/cpu/registration.cpp:14-177. xyz is transformed in place at the end (:168-174). Returns iterations.
int wscpu_register_cloud(const int32_t *size, const int32_t *pos_, const int32_t *offset, uint32_t *data, int32_t *xyz,
                         size_t n, const float *T_in, int max_iterations, float it_weight_gradient, float epsilon,
                         int res, float *T_out, int threads)
{
  LocalMap map;
  for (int k = 0; k < 3; ++k) { map.size[k] = size[k]; map.pos[k] = pos_[k]; map.offset[k] = offset[k]; }
  map.data = data;
  if (threads > 0) omp_set_num_threads(threads);

  float total[16];
  std::memcpy(total, T_in, sizeof total);
  float alpha = 0;
  float previous_errors[4] = {0, 0, 0, 0};
  int error = 0, count = 0, iterations = 0;
  bool finished = false;
  long h[6][6] = {}, g[6] = {};

#pragma omp parallel
  {
    long local_h[6][6], local_g[6];
    int local_error, local_count;
    int M[16];
    for (int i = 0; i < max_iterations && !finished; i++)
    {
      std::memset(local_h, 0, sizeof local_h);
      std::memset(local_g, 0, sizeof local_g);
      local_error = 0;
      local_count = 0;
      int center[3] = {(int)total[12], (int)total[13], (int)total[14]};
      for (int k = 0; k < 16; ++k) M[k] = (int)(total[k] * MR);

#pragma omp for schedule(static) nowait
      for (long j = 0; j < (long)n; j++)
      {
        const int *p = xyz + 3 * j;
        int point[3], buf[3];
        for (int r = 0; r < 3; ++r) point[r] = (M[0 + r] * p[0] + M[4 + r] * p[1] + M[8 + r] * p[2] + M[12 + r]) / MR;
        for (int r = 0; r < 3; ++r) buf[r] = point[r] / res;
        for (int r = 0; r < 3; ++r) point[r] -= center[r];
        // map.value() throws std::out_of_range for every access outside the map -> point skipped (:67,:129)
        if (!map.in_bounds(buf[0], buf[1], buf[2])) continue;
        Entry current = map.get(buf[0], buf[1], buf[2]);
        if (current.weight == 0) continue;
        bool ok = true;
        int grad[3] = {0, 0, 0};
        Entry nx[3], ls[3];
        for (int r = 0; r < 3 && ok; ++r)
        {
          int a[3] = {buf[0], buf[1], buf[2]}, b[3] = {buf[0], buf[1], buf[2]};
          a[r] += 1; b[r] -= 1;
          if (!map.in_bounds(a[0], a[1], a[2]) || !map.in_bounds(b[0], b[1], b[2])) { ok = false; break; }
          nx[r] = map.get(a[0], a[1], a[2]);
          ls[r] = map.get(b[0], b[1], b[2]);
        }
        if (!ok) continue;
        for (int r = 0; r < 3; ++r)
          if (nx[r].weight != 0 && ls[r].weight != 0 && !((nx[r].value > 0 && ls[r].value < 0) || (nx[r].value < 0 && ls[r].value > 0)))
            grad[r] = (nx[r].value - ls[r].value) / 2;
        long J[6];
        J[0] = point[1] * grad[2] - point[2] * grad[1];
        J[1] = point[2] * grad[0] - point[0] * grad[2];
        J[2] = point[0] * grad[1] - point[1] * grad[0];
        J[3] = grad[0]; J[4] = grad[1]; J[5] = grad[2];
        for (int r = 0; r < 6; ++r)
        {
          for (int q = 0; q < 6; ++q) local_h[r][q] += J[r] * J[q];
          local_g[r] += J[r] * current.value;
        }
        local_error += std::abs((int)current.value);
        local_count++;
      }
#pragma omp critical
      {
        for (int r = 0; r < 6; ++r)
        {
          for (int q = 0; q < 6; ++q) h[r][q] += local_h[r][q];
          g[r] += local_g[r];
        }
        error += local_error;
        count += local_count;
      }
#pragma omp barrier
#pragma omp single
      {
        iterations = i + 1;
        if (count == 0)
        {
          finished = true; // guard: the reference would produce NaN here
        }
        else
        {
          double hf[6][6], gf[6], xi[6];
          double w = (double)(alpha * (float)count);
          for (int r = 0; r < 6; ++r)
          {
            gf[r] = (double)g[r];
            for (int q = 0; q < 6; ++q) hf[r][q] = (double)h[r][q] + (r == q ? w : 0.0);
          }
          if (solve6(hf, gf, xi) != 0)
          {
            finished = true;
          }
          else
          {
            for (int r = 0; r < 6; ++r) xi[r] = -xi[r];
            float tr[16], out[16];
            xi_to_transform(xi, center, tr);
            alpha += it_weight_gradient;
            for (int jj = 0; jj < 4; ++jj)
              for (int ii = 0; ii < 4; ++ii)
              {
                float s = 0.f;
                for (int k = 0; k < 4; ++k) s += tr[k * 4 + ii] * total[jj * 4 + k];
                out[jj * 4 + ii] = s;
              }
            std::memcpy(total, out, sizeof out);
            float err = (float)error / count;
            if (std::fabs(err - previous_errors[2]) < epsilon && std::fabs(err - previous_errors[0]) < epsilon) finished = true;
            for (int e = 1; e < 4; e++) previous_errors[e - 1] = previous_errors[e];
            previous_errors[3] = err;
          }
        }
        std::memset(h, 0, sizeof h);
        std::memset(g, 0, sizeof g);
        error = 0;
        count = 0;
      } // implicit barrier of "single"
    }
  }

  int M[16];
  for (int k = 0; k < 16; ++k) M[k] = (int)(total[k] * MR);
#pragma omp parallel for schedule(static)
  for (long i = 0; i < (long)n; i++)
  {
    int *p = xyz + 3 * i;
    int q[3];
    for (int r = 0; r < 3; ++r) q[r] = (M[0 + r] * p[0] + M[4 + r] * p[1] + M[8 + r] * p[2] + M[12 + r]) / MR;
    p[0] = q[0]; p[1] = q[1]; p[2] = q[2];
  }
  std::memcpy(T_out, total, sizeof total);
  return iterations;
}

int wscpu_max_threads() { return omp_get_max_threads(); }

} // extern "C"
