// tsdf_integrate.hip — the parts of the TSDF update that stream the maps instead of walking rays (gfx950):
//   integrate_dense_kernel    cu_avg_tsdf_krnl (src/warpsense/cuda/update_tsdf.cu:13-43) over EVERY voxel: the HBM-roofline stream
//   integrate_sparse_kernel   the same over the touched tiles only (the separate route; the default folds it into tile_resolve_kernel)
//   box_copy / box_fill       slabs of the ring buffer <-> a dense box (device side of the map shift)
//   tsdf_stats_kernel         statistics of the last update from the per-workgroup slots
// Split from tsdf_update.hip in round 6 (that file holds the scatter: set-up, marches, resolve, and the host logic around them).
#include "ws_march.h"

namespace ws
{

// ---------------------------------------------------------------------------------------------------------
// integrate
// ---------------------------------------------------------------------------------------------------------
struct IntegrateArgs
{
  uint32_t *new_data;
  uint32_t *avg_data;
  const TileEntry *tile_list;
  MapParams map;
  int32_t nty, ntz;
  int64_t n_vox;
  int32_t max_weight;
  int32_t tau;
  TsdfCounters *counters;
};

// measured on MI355X, 513^3 (2.16 GB moved), us per launch: 3072 x 8: 393-399 (the round-2 setting), 6144 x 4: 372-380,
// 16384 x 4: 365-366 (5.9 TB/s), 12288 x 2: 380-388, 65536 x 2: 372
#ifndef DENSE_GRID
#define DENSE_GRID 16384
#endif
#ifndef DENSE_UNROLL
#define DENSE_UNROLL 4
#endif
constexpr int SPARSE_GRID = 4096;

// cu_avg_tsdf_krnl (update_tsdf.cu:13-43) over the touched tiles only: one workgroup per tile, the voxel mapping of
// tile_resolve_kernel (64-byte runs along z).
__global__ __launch_bounds__(256) void integrate_sparse_kernel(IntegrateArgs a)
{
  const uint32_t n_list = a.counters->n_listed + a.counters->n_appended; // tiles with records + the others the resolve found
  const uint32_t reset = pack_entry(a.tau, 0);
  // thread t owns the voxels 4t .. 4t+3 of the tile: column t >> (ZB - 2), four consecutive z
  const int col = threadIdx.x >> (TILE_ZB - 2), lx = col >> TILE_YB, ly = col & ((1 << TILE_YB) - 1), z0 = (threadIdx.x & ((1 << (TILE_ZB - 2)) - 1)) * 4;
  for (uint32_t e = blockIdx.x; e < n_list; e += gridDim.x)
  {
    const TileEntry te = a.tile_list[e];
    const int32_t tx = te.tx, ty = te.ty, tz = te.tz;
    const int32_t sx = (tx << TILE_XB) + lx, sy = (ty << TILE_YB) + ly, sz = (tz << TILE_ZB) + z0;
    if (sx >= a.map.size[0] || sy >= a.map.size[1]) continue;
    int nz = a.map.size[2] - sz;
    nz = nz > 4 ? 4 : nz;
    const int64_t idx0 = storage_index(a.map, sx, sy, sz);
    // both arrays in one round trip (the arrays carry 16 bytes of slack behind the last voxel)
    const u32x4 f4 = *reinterpret_cast<const u32x4_a4 *>(a.new_data + idx0);
    const u32x4 e4 = *reinterpret_cast<const u32x4_a4 *>(a.avg_data + idx0);
    const uint32_t fresh[4] = {f4.x, f4.y, f4.z, f4.w}, existing[4] = {e4.x, e4.y, e4.z, e4.w};
#pragma unroll
    for (int j = 0; j < 4; ++j)
    {
      if (j >= nz || fresh[j] == reset) continue;
      const uint32_t updated = integrate_entry(existing[j], fresh[j], a.max_weight);
      if (updated != existing[j]) a.avg_data[idx0 + j] = updated;
      a.new_data[idx0 + j] = reset;
    }
  }
}

// statistics of the last update, on demand (ws_tsdf_stats): the hot kernels keep per-workgroup slots, no shared counters
__global__ __launch_bounds__(256) void tsdf_stats_kernel(TsdfCounters *c, const uint32_t *tail_stats, uint32_t n_tail, const uint32_t *resolve_stats,
                                                         uint32_t n_resolve)
{
  __shared__ uint32_t part[12];
  uint32_t rec = 0, con = 0, grp = 0, til = 0;
  for (uint32_t i = threadIdx.x; i < n_tail; i += 256)
  {
    rec += tail_stats[i];
    grp += tail_stats[WS_TAIL_STATS + i];
  }
  for (uint32_t i = threadIdx.x; i < n_resolve; i += 256)
  {
    con += resolve_stats[2 * i + 0];
    til += resolve_stats[2 * i + 1];
  }
  for (int d = 32; d > 0; d >>= 1)
  {
    rec += __shfl_down(rec, d, 64);
    con += __shfl_down(con, d, 64);
    grp += __shfl_down(grp, d, 64);
    til += __shfl_down(til, d, 64);
  }
  __shared__ uint32_t part4[4];
  if ((threadIdx.x & 63) == 0)
  {
    part[(threadIdx.x >> 6) * 3 + 0] = rec;
    part[(threadIdx.x >> 6) * 3 + 1] = con;
    part[(threadIdx.x >> 6) * 3 + 2] = grp;
    part4[threadIdx.x >> 6] = til;
  }
  __syncthreads();
  if (threadIdx.x == 0)
  {
    c->last_records = part[0] + part[3] + part[6] + part[9] + c->last_free_keyed; // tail records + free-space candidates that joined them
    c->last_contested = part[1] + part[4] + part[7] + part[10];
    c->last_runs = part[2] + part[5] + part[8] + part[11];
    c->last_listed = part4[0] + part4[1] + part4[2] + part4[3];
  }
}
int launch_tsdf_stats(ws_map *m)
{
  hipLaunchKernelGGL(tsdf_stats_kernel, dim3(1), dim3(256), 0, m->ctx->stream, m->counters, (const uint32_t *)m->block_stats, m->tail_blocks,
                     (const uint32_t *)(m->block_stats + 2 * WS_TAIL_STATS), m->resolve_blocks);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

// cu_avg_tsdf_krnl over EVERY voxel: the HBM-roofline stream, 16 B per voxel
// (read new + existing, write existing + reset new), 4 voxels per lane as 128-bit accesses.
__global__ __launch_bounds__(256) void integrate_dense_kernel(IntegrateArgs a)
{
  const int64_t n4 = a.n_vox >> 2;
  const uint32_t reset = pack_entry(a.tau, 0);
  const u32x4 reset4 = {reset, reset, reset, reset};
  u32x4 *new4 = reinterpret_cast<u32x4 *>(a.new_data);
  u32x4 *avg4 = reinterpret_cast<u32x4 *>(a.avg_data);
  constexpr int U = DENSE_UNROLL; // 128-bit accesses in flight per lane and array
  const int64_t stride = (int64_t)gridDim.x * 256 * U;
  for (int64_t base = (int64_t)blockIdx.x * 256 * U + threadIdx.x; base < n4; base += stride)
  {
    u32x4 f[U], e[U];
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t i = base + (int64_t)u * 256;
      if (i < n4)
      {
        f[u] = __builtin_nontemporal_load(&new4[i]);
        e[u] = __builtin_nontemporal_load(&avg4[i]);
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u)
    {
      const int64_t i = base + (int64_t)u * 256;
      if (i < n4)
      {
        u32x4 r = e[u];
        // untouched voxels (new == (tau, 0)) leave avg as it is: only touched ones pay for the weighted average
        if (f[u].x != reset) r.x = integrate_entry(r.x, f[u].x, a.max_weight);
        if (f[u].y != reset) r.y = integrate_entry(r.y, f[u].y, a.max_weight);
        if (f[u].z != reset) r.z = integrate_entry(r.z, f[u].z, a.max_weight);
        if (f[u].w != reset) r.w = integrate_entry(r.w, f[u].w, a.max_weight);
        __builtin_nontemporal_store(r, &avg4[i]);
        __builtin_nontemporal_store(reset4, &new4[i]);
      }
    }
  }
  // tail (n_vox is odd for the reference's odd-sized maps)
  if (blockIdx.x == 0 && threadIdx.x < (a.n_vox & 3))
  {
    const int64_t i = (n4 << 2) + threadIdx.x;
    a.avg_data[i] = integrate_entry(a.avg_data[i], a.new_data[i], a.max_weight);
    a.new_data[i] = reset;
  }
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *dst, uint32_t v, int64_t n)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}

// ---- slabs of the ring buffer <-> a dense box (device side of the map shift) ----
// box-local order: x major, z fastest, like the maps; lo/ext in world voxel coordinates, box inside the window
template <bool PACK>
__global__ __launch_bounds__(256) void box_copy_kernel(uint32_t *map_data, MapParams mp, int32_t lox, int32_t loy, int32_t loz, int32_t ex,
                                                       int32_t ey, int32_t ez, uint32_t *box)
{
  const int64_t n = (int64_t)ex * ey * ez;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
  {
    const int32_t z = (int32_t)(i % ez);
    const int32_t y = (int32_t)((i / ez) % ey);
    const int32_t x = (int32_t)(i / ((int64_t)ez * ey));
    const int64_t idx = get_index(mp, lox + x, loy + y, loz + z);
    if (PACK)
      box[i] = map_data[idx];
    else
      map_data[idx] = box[i];
  }
}

int launch_box_copy(ws_map *m, const MapParams &par, int which, const int32_t lo[3], const int32_t ext[3], uint32_t *box_dev, bool pack, hipStream_t s)
{
  const int64_t n = (int64_t)ext[0] * ext[1] * ext[2];
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  if (pack)
    hipLaunchKernelGGL((box_copy_kernel<true>), dim3((unsigned)blocks), dim3(256), 0, s, m->data[which], par, lo[0], lo[1], lo[2],
                       ext[0], ext[1], ext[2], box_dev);
  else
    hipLaunchKernelGGL((box_copy_kernel<false>), dim3((unsigned)blocks), dim3(256), 0, s, m->data[which], par, lo[0], lo[1], lo[2],
                       ext[0], ext[1], ext[2], box_dev);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

__global__ __launch_bounds__(256) void box_fill_kernel(uint32_t *map_data, MapParams mp, int32_t lox, int32_t loy, int32_t loz, int32_t ex, int32_t ey,
                                                       int32_t ez, uint32_t value)
{
  const int64_t n = (int64_t)ex * ey * ez;
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride)
  {
    const int32_t z = (int32_t)(i % ez);
    const int32_t y = (int32_t)((i / ez) % ey);
    const int32_t x = (int32_t)(i / ((int64_t)ez * ey));
    map_data[get_index(mp, lox + x, loy + y, loz + z)] = value;
  }
}

int launch_box_fill(ws_map *m, const MapParams &par, int which, const int32_t lo[3], const int32_t ext[3], uint32_t value, hipStream_t s)
{
  const int64_t n = (int64_t)ext[0] * ext[1] * ext[2];
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  if (blocks < 1) blocks = 1;
  hipLaunchKernelGGL(box_fill_kernel, dim3((unsigned)blocks), dim3(256), 0, s, m->data[which], par, lo[0], lo[1], lo[2], ext[0], ext[1],
                     ext[2], value);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int fill_u32(ws_context *ctx, uint32_t *dst, uint32_t value, int64_t n)
{
  if (n <= 0) return WS_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_u32_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dst, value, n);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_tsdf_integrate(ws_map *m)
{
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  IntegrateArgs ia;
  ia.new_data = m->data[WS_MAP_NEW];
  ia.avg_data = m->data[WS_MAP_AVG];
  ia.tile_list = m->tile_list;
  ia.map = m->par[WS_MAP_NEW];
  ia.nty = m->nty;
  ia.ntz = m->ntz;
  ia.n_vox = m->n_vox;
  ia.max_weight = m->max_weight;
  ia.tau = m->tau;
  ia.counters = m->counters;
  const dim3 block(256);
  if (!m->fused_done)
  {
    // a non-default new_map must be streamed completely: untouched voxels carry entries too
    const bool dense = (m->integrate_mode == WS_INTEGRATE_DENSE) || !m->new_is_default;
    prof_begin(ctx, WS_K_INTEGRATE);
    if (dense)
    {
      int64_t blocks = ((m->n_vox >> 2) + 256 * DENSE_UNROLL - 1) / (256 * DENSE_UNROLL);
      if (blocks > DENSE_GRID) blocks = DENSE_GRID;
      if (blocks < 1) blocks = 1;
      hipLaunchKernelGGL(integrate_dense_kernel, dim3((unsigned)blocks), block, 0, s, ia);
    }
    else
    {
      hipLaunchKernelGGL(integrate_sparse_kernel, dim3(SPARSE_GRID), block, 0, s, ia);
    }
    prof_end(ctx, WS_K_INTEGRATE);
  }
  WS_HIP(hipGetLastError());
  m->fused_done = false;
  m->new_is_default = true;
  return WS_OK;
}


} // namespace ws
