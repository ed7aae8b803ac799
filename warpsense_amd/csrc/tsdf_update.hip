// tsdf_update.hip — TSDF volume update for MI355X (gfx950).
//
// Replaces cu_min_tsdf_krnl + cu_avg_tsdf_krnl (src/warpsense/cuda/update_tsdf.cu:13-128) of the reference.
//
// The reference scatters with a racy CAS ("first positive-weight entry freezes the voxel",
// include/warpsense/cuda/util.h:70-102), so its result depends on thread arrival order.  This
// implementation computes the result of ONE fixed legal schedule — the serial one, ascending
// (point, ray step, fan step) — deterministically:
//
//   march<EMIT>     every candidate carries an order key t; per voxel two 64-bit atomicMin words keep
//                   kpos = earliest positive-weight candidate, kneg = smallest-|value| (latest on ties)
//                   negative-weight candidate; a byte per 64-voxel tile marks touched tiles.
//   resolve         voxels whose earliest positive candidate cannot have been blocked by any negative
//                   one (|v_pos| <= min |v_neg|) — or that only saw negatives — are final: the entry
//                   is written to new_map.  The rest are "contested" and get a list head.
//   march<COLLECT>  (only if contested voxels exist) appends every candidate of a contested voxel to a
//                   linked list in an arena.
//   resolve_lists   folds each list in ascending key order with the reference's accept rule -> new_map.
//   integrate       weighted average of new_map into avg_map and reset of new_map, either over the
//                   touched tiles only (sparse) or over every voxel (dense, the reference's kernel).
//
// new_map after resolve* is bit-identical to what the reference kernel leaves there when its threads
// run one after the other (oracle/ws_oracle.c: wso_update_min).
#include "ws_device.h"

namespace ws
{

struct MarchArgs
{
  const int32_t *xyz;
  uint32_t n;
  int32_t scanner_pos[3];
  int32_t up[3];
  MapParams map; // new_map's parameters (the reference indexes new_map in the scatter, update_tsdf.cu:55-125)
  int32_t tau;
  int32_t res;
  uint64_t *kpos;
  uint64_t *kneg;
  uint8_t *dirty;
  const uint32_t *new_data; // only read when HAS_S0
  TsdfCounters *counters;
  uint32_t *heads;
  ContestedRecord *arena;
  uint32_t arena_cap;
};

enum
{
  MARCH_EMIT = 0,
  MARCH_COLLECT = 1
};

// One lane walks one ray (update_tsdf.cu:45-128).
template <int MODE, bool HAS_S0>
__global__ __launch_bounds__(256) void march_kernel(MarchArgs a)
{
  const uint32_t ix = blockIdx.x * 256u + threadIdx.x;
  if (ix >= a.n) return;
  if (MODE == MARCH_COLLECT)
  {
    if (__hip_atomic_load(&a.counters->contested, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0) return;
  }

  const int32_t res = a.res, tau = a.tau;
  const int32_t weight_epsilon = tau / 10;
  const int32_t half = res / 2;
  const int32_t px = a.xyz[3 * (size_t)ix + 0], py = a.xyz[3 * (size_t)ix + 1], pz = a.xyz[3 * (size_t)ix + 2];

  // cu_to_map (cuda/util.h:111-114) + in_bounds_with_buffer_pos (update_tsdf.cu:55)
  {
    const float fr = (float)res;
    const int32_t cx = (int32_t)floorf(__fdiv_rn((float)px, fr));
    const int32_t cy = (int32_t)floorf(__fdiv_rn((float)py, fr));
    const int32_t cz = (int32_t)floorf(__fdiv_rn((float)pz, fr));
    if (!in_bounds_buffer(a.map, cx, cy, cz, (int64_t)(tau / res / 2))) return;
  }

  // cu_to_mm (cuda/util.h:116-123)
  const int32_t posx = wadd(wmul(a.scanner_pos[0], res), half);
  const int32_t posy = wadd(wmul(a.scanner_pos[1], res), half);
  const int32_t posz = wadd(wmul(a.scanner_pos[2], res), half);
  const int32_t dx = wsub(px, posx), dy = wsub(py, posy), dz = wsub(pz, posz);
  const int32_t distance = l2norm_i(dx, dy, dz);
  if (distance == 0) return; // guard (the reference divides by zero here; src/cpu/update_tsdf.cpp:593 has the guard)

  // update_tsdf.cu:59-63, in int64 like the reference's `long`
  const int64_t MR = MATRIX_RESOLUTION;
  const int64_t ndx = wmul64(dx, MR) / distance, ndy = wmul64(dy, MR) / distance, ndz = wmul64(dz, MR) / distance;
  const int64_t ux = a.up[0], uy = a.up[1], uz = a.up[2];
  const int64_t c1x = wsub64(wmul64(ndy, uz), wmul64(ndz, uy)) / MR;
  const int64_t c1y = wsub64(wmul64(ndz, ux), wmul64(ndx, uz)) / MR;
  const int64_t c1z = wsub64(wmul64(ndx, uy), wmul64(ndy, ux)) / MR;
  int64_t ivx = wsub64(wmul64(ndy, c1z), wmul64(ndz, c1y));
  int64_t ivy = wsub64(wmul64(ndz, c1x), wmul64(ndx, c1z));
  int64_t ivz = wsub64(wmul64(ndx, c1y), wmul64(ndy, c1x));
  const int64_t inorm = l2norm_l(ivx, ivy, ivz);
  if (inorm == 0) return; // guard (src/cpu/update_tsdf.cpp:602)
  ivx = wmul64(ivx, MR) / inorm;
  ivy = wmul64(ivy, MR) / inorm;
  ivz = wmul64(ivz, MR) / inorm;

  int32_t prevx = 0, prevy = 0; // update_tsdf.cu:65 (z of prev is never compared)
  uint32_t iter = 0;
  const int32_t len_end = distance + tau;
  for (int32_t len = 1; len <= len_end; len += half, ++iter)
  {
    const int32_t projx = wadd(posx, wmul(dx, len) / distance);
    const int32_t projy = wadd(posy, wmul(dy, len) / distance);
    const int32_t projz = wadd(posz, wmul(dz, len) / distance);
    const int32_t ixx = projx / res, iyy = projy / res, izz = projz / res;
    if (ixx == prevx && iyy == prevy) continue;
    prevx = ixx;
    prevy = iyy;
    if (!in_bounds(a.map, ixx, iyy, izz)) continue;

    // update_tsdf.cu:81-98
    const int32_t tcx = wadd(wmul(ixx, res), half), tcy = wadd(wmul(iyy, res), half), tcz = wadd(wmul(izz, res), half);
    int32_t value = l2norm_i(wsub(px, tcx), wsub(py, tcy), wsub(pz, tcz));
    value = value < tau ? value : tau;
    if (len > distance) value = -value;
    const int32_t weight = tsdf_weight(value, tau, weight_epsilon);
    if (weight == 0) continue;
    const uint32_t absval = (uint32_t)(value < 0 ? -value : value) & 0x7fffu;

    // update_tsdf.cu:101-105
    const int32_t delta_z = wmul(DZ_PER_DISTANCE, len) / MATRIX_RESOLUTION;
    const int32_t iter_steps = (delta_z * 2) / res + 1;
    const int32_t mid = delta_z / res;
    const int32_t lowx = wsub(projx, (int32_t)(wmul64(delta_z, ivx) / MR));
    const int32_t lowy = wsub(projy, (int32_t)(wmul64(delta_z, ivy) / MR));
    const int32_t lowz = wsub(projz, (int32_t)(wmul64(delta_z, ivz) / MR));
    if (iter > 0xffffu || iter_steps > 256)
    {
      atomicOr(&a.counters->error, 2u);
      return;
    }

    for (int32_t step = 0; step < iter_steps; ++step)
    {
      const int64_t sm = (int64_t)wmul(step, res);
      const int32_t vx = wadd(lowx, (int32_t)(wmul64(sm, ivx) / MR)) / res;
      const int32_t vy = wadd(lowy, (int32_t)(wmul64(sm, ivy) / MR)) / res;
      const int32_t vz = wadd(lowz, (int32_t)(wmul64(sm, ivz) / MR)) / res;
      if (!in_bounds(a.map, vx, vy, vz)) continue;
      const int64_t idx = get_index(a.map, vx, vy, vz);
      const bool positive = (step == mid);
      const uint64_t t = ((uint64_t)ix << 24) | ((uint64_t)iter << 8) | (uint64_t)step;

      if (HAS_S0)
      {
        // candidates the initial new_map entry would reject can never be accepted later either
        // (the stored |value| only shrinks and a positive weight freezes the voxel): drop them here
        const uint32_t s0 = a.new_data[idx];
        const int32_t a0 = entry_value(s0) < 0 ? -entry_value(s0) : entry_value(s0);
        if (entry_weight(s0) > 0 || (int32_t)absval > a0) continue;
      }

      if (MODE == MARCH_EMIT)
      {
        const int64_t tile = idx >> TILE_SHIFT;
        if (a.dirty[tile] == 0) a.dirty[tile] = 1;
        if (positive)
        {
          const uint64_t key = (t << 16) | ((uint32_t)value & 0xffffu);
          if (key < a.kpos[idx]) atomicMin((unsigned long long *)&a.kpos[idx], (unsigned long long)key);
        }
        else
        {
          const uint64_t key = ((uint64_t)absval << 45) | ((T_MASK - t) << 1) | (value < 0 ? 1u : 0u);
          if (key < a.kneg[idx]) atomicMin((unsigned long long *)&a.kneg[idx], (unsigned long long)key);
        }
      }
      else
      {
        const uint64_t k = a.kpos[idx];
        if ((k >> 60) == 0xCull)
        {
          const uint32_t slot = (uint32_t)(k & 0xffffffffull);
          const uint32_t rec = atomicAdd(&a.counters->records, 1u);
          if (rec < a.arena_cap)
          {
            ContestedRecord r;
            r.key = (t << 17) | (positive ? 0ull : (1ull << 16)) | ((uint32_t)value & 0xffffu);
            r.next = atomicExch(&a.heads[slot], rec);
            r.pad = 0;
            a.arena[rec] = r;
          }
          else
          {
            atomicOr(&a.counters->error, 1u);
          }
        }
      }
    }
  }
}

struct ResolveArgs
{
  uint64_t *kpos;
  uint64_t *kneg;
  uint8_t *dirty;
  uint32_t *new_data;
  int64_t n_vox;
  int64_t n_tiles;
  int32_t tau;
  TsdfCounters *counters;
  uint32_t *contested_lo;
  uint32_t *contested_hi;
  uint32_t *heads;
  uint32_t contested_cap;
  const ContestedRecord *arena;
  uint32_t arena_cap;
};

// One wave scans 64 tile flags, then walks its dirty tiles with one lane per voxel.
__global__ __launch_bounds__(256) void resolve_kernel(ResolveArgs a)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t tile0 = wave * 64;
  if (tile0 >= a.n_tiles) return;
  const int32_t weight_epsilon = a.tau / 10;
  const int64_t my_tile = tile0 + lane;
  const bool flag = my_tile < a.n_tiles && a.dirty[my_tile] != 0;
  unsigned long long mask = __ballot(flag);
  while (mask)
  {
    const int b = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const int64_t idx = ((tile0 + b) << TILE_SHIFT) + lane;
    if (idx >= a.n_vox) continue;
    const uint64_t kp = a.kpos[idx], kn = a.kneg[idx];
    if (kp == KEY_INF && kn == KEY_INF) continue; // untouched voxel: new_map keeps its entry
    bool decided = true;
    int32_t value = 0;
    bool positive = false;
    if (kp != KEY_INF)
    {
      value = (int32_t)(int16_t)(kp & 0xffffu);
      positive = true;
      if (kn != KEY_INF)
      {
        const int32_t ap = value < 0 ? -value : value;
        const int32_t an = (int32_t)(kn >> 45);
        // a negative candidate with a smaller |value| MAY have come first and blocked it: ordered fallback
        if (ap > an) decided = false;
      }
    }
    else
    {
      const int32_t an = (int32_t)(kn >> 45);
      value = (kn & 1ull) ? -an : an;
    }
    if (decided)
    {
      int32_t w = tsdf_weight(value, a.tau, weight_epsilon);
      a.new_data[idx] = pack_entry(value, positive ? w : -w);
      a.kpos[idx] = KEY_INF;
      if (kn != KEY_INF) a.kneg[idx] = KEY_INF;
    }
    else
    {
      const uint32_t slot = atomicAdd(&a.counters->contested, 1u);
      if (slot < a.contested_cap)
      {
        a.contested_lo[slot] = (uint32_t)((uint64_t)idx & 0xffffffffull);
        a.contested_hi[slot] = (uint32_t)((uint64_t)idx >> 32);
        a.heads[slot] = 0xffffffffu;
        a.kpos[idx] = KEY_CONTESTED_TAG | slot;
        a.kneg[idx] = KEY_INF;
      }
      else
      {
        atomicOr(&a.counters->error, 1u);
        a.kpos[idx] = KEY_INF;
        a.kneg[idx] = KEY_INF;
      }
    }
  }
}

// One lane per contested voxel: fold its candidates in ascending key order with the accept rule of
// atomic_tsdf_min (cuda/util.h:70-102): accept iff stored weight <= 0 and |new| <= |stored|.
template <bool HAS_S0>
__global__ __launch_bounds__(256) void resolve_lists_kernel(ResolveArgs a)
{
  const uint32_t slot = blockIdx.x * 256u + threadIdx.x;
  uint32_t n = __hip_atomic_load(&a.counters->contested, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  if (n > a.contested_cap) n = a.contested_cap;
  if (slot >= n) return;
  const int64_t idx = (int64_t)(((uint64_t)a.contested_hi[slot] << 32) | a.contested_lo[slot]);
  const int32_t weight_epsilon = a.tau / 10;
  uint32_t state = HAS_S0 ? a.new_data[idx] : pack_entry(a.tau, 0);
  int32_t sv = entry_value(state), sw = entry_weight(state);
  int32_t sa = sv < 0 ? -sv : sv;
  const uint32_t head = a.heads[slot];
  uint64_t last = 0;
  bool first = true;
  while (sw <= 0)
  {
    // next record in key order (lists are short: a handful of rays reach a far voxel)
    uint64_t best = KEY_INF;
    for (uint32_t r = head; r != 0xffffffffu && r < a.arena_cap; r = a.arena[r].next)
    {
      const uint64_t k = a.arena[r].key;
      if ((first || k > last) && k < best) best = k;
    }
    if (best == KEY_INF) break;
    first = false;
    last = best;
    const int32_t v = (int32_t)(int16_t)(best & 0xffffu);
    const int32_t av = v < 0 ? -v : v;
    if (av <= sa)
    {
      const int32_t w = tsdf_weight(v, a.tau, weight_epsilon);
      sv = v;
      sa = av;
      sw = (best & (1ull << 16)) ? -w : w;
    }
  }
  a.new_data[idx] = pack_entry(sv, sw);
  a.kpos[idx] = KEY_INF;
}

struct IntegrateArgs
{
  uint32_t *new_data;
  uint32_t *avg_data;
  uint8_t *dirty;
  int64_t n_vox;
  int64_t n_tiles;
  int32_t max_weight;
  int32_t tau;
  TsdfCounters *counters;
};

// cu_avg_tsdf_krnl (update_tsdf.cu:13-43) over the touched tiles only.
__global__ __launch_bounds__(256) void integrate_sparse_kernel(IntegrateArgs a)
{
  const int lane = threadIdx.x & 63;
  const int64_t wave = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
  const int64_t tile0 = wave * 64;
  if (tile0 >= a.n_tiles) return;
  const int64_t my_tile = tile0 + lane;
  const bool flag = my_tile < a.n_tiles && a.dirty[my_tile] != 0;
  unsigned long long mask = __ballot(flag);
  if (mask == 0) return;
  if (flag) a.dirty[my_tile] = 0;
  if (lane == 0) atomicAdd(&a.counters->dirty_tiles, (uint32_t)__popcll(mask));
  const uint32_t reset = pack_entry(a.tau, 0);
  while (mask)
  {
    const int b = __ffsll((long long)mask) - 1;
    mask &= mask - 1;
    const int64_t idx = ((tile0 + b) << TILE_SHIFT) + lane;
    if (idx >= a.n_vox) continue;
    const uint32_t fresh = a.new_data[idx];
    if (fresh == reset) continue;
    const uint32_t existing = a.avg_data[idx];
    const uint32_t updated = integrate_entry(existing, fresh, a.max_weight);
    if (updated != existing) a.avg_data[idx] = updated;
    a.new_data[idx] = reset;
  }
}

// cu_avg_tsdf_krnl over EVERY voxel: the HBM-roofline stream, 16 B per voxel
// (read new + existing, write existing + reset new), 4 voxels per lane as 128-bit accesses.
__global__ __launch_bounds__(256) void integrate_dense_kernel(IntegrateArgs a)
{
  const int64_t n4 = a.n_vox >> 2;
  const uint32_t reset = pack_entry(a.tau, 0);
  const uint4 reset4 = make_uint4(reset, reset, reset, reset);
  uint4 *new4 = reinterpret_cast<uint4 *>(a.new_data);
  uint4 *avg4 = reinterpret_cast<uint4 *>(a.avg_data);
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride)
  {
    const uint4 f = new4[i];
    uint4 e = avg4[i];
    e.x = integrate_entry(e.x, f.x, a.max_weight);
    e.y = integrate_entry(e.y, f.y, a.max_weight);
    e.z = integrate_entry(e.z, f.z, a.max_weight);
    e.w = integrate_entry(e.w, f.w, a.max_weight);
    avg4[i] = e;
    new4[i] = reset4;
  }
  // tail (n_vox is odd for the reference's odd-sized maps)
  if (blockIdx.x == 0 && threadIdx.x < (a.n_vox & 3))
  {
    const int64_t i = (n4 << 2) + threadIdx.x;
    a.avg_data[i] = integrate_entry(a.avg_data[i], a.new_data[i], a.max_weight);
    a.new_data[i] = reset;
  }
}

__global__ __launch_bounds__(256) void clear_dirty_kernel(uint8_t *dirty, int64_t n_tiles)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n_tiles; i += stride) dirty[i] = 0;
}

__global__ __launch_bounds__(256) void fill_u32_kernel(uint32_t *dst, uint32_t v, int64_t n)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
}
__global__ __launch_bounds__(256) void fill_u64_kernel(uint64_t *dst, uint64_t v, int64_t n)
{
  const int64_t stride = (int64_t)gridDim.x * 256;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) dst[i] = v;
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
int fill_u64(ws_context *ctx, uint64_t *dst, uint64_t value, int64_t n)
{
  if (n <= 0) return WS_OK;
  int64_t blocks = (n + 255) / 256;
  if (blocks > 4096) blocks = 4096;
  hipLaunchKernelGGL(fill_u64_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, dst, value, n);
  WS_HIP(hipGetLastError());
  return WS_OK;
}

int launch_tsdf_scatter(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3])
{
  ws_context *ctx = m->ctx;
  hipStream_t s = ctx->stream;
  WS_HIP(hipMemsetAsync(m->counters, 0, sizeof(TsdfCounters), s));
  if (n == 0) return WS_OK;

  MarchArgs ma;
  ma.xyz = xyz_dev;
  ma.n = (uint32_t)n;
  for (int k = 0; k < 3; ++k)
  {
    ma.scanner_pos[k] = scanner_pos[k];
    ma.up[k] = up[k];
  }
  ma.map = m->par[WS_MAP_NEW];
  ma.tau = m->tau;
  ma.res = m->res;
  ma.kpos = m->kpos;
  ma.kneg = m->kneg;
  ma.dirty = m->dirty;
  ma.new_data = m->data[WS_MAP_NEW];
  ma.counters = m->counters;
  ma.heads = m->heads;
  ma.arena = m->arena;
  ma.arena_cap = m->arena_cap;

  ResolveArgs ra;
  ra.kpos = m->kpos;
  ra.kneg = m->kneg;
  ra.dirty = m->dirty;
  ra.new_data = m->data[WS_MAP_NEW];
  ra.n_vox = m->n_vox;
  ra.n_tiles = m->n_tiles;
  ra.tau = m->tau;
  ra.counters = m->counters;
  ra.contested_lo = m->contested_vox_lo;
  ra.contested_hi = m->contested_vox_hi;
  ra.heads = m->heads;
  ra.contested_cap = m->contested_cap;
  ra.arena = m->arena;
  ra.arena_cap = m->arena_cap;

  const dim3 block(256);
  const dim3 grid_rays((unsigned)((n + 255) / 256));
  const dim3 grid_tiles((unsigned)((m->n_tiles + 64 * 4 - 1) / (64 * 4)));
  const dim3 grid_lists((m->contested_cap + 255) / 256);
  const bool s0 = !m->new_is_default;

  prof_begin(ctx, WS_K_MARCH_EMIT);
  if (s0)
    hipLaunchKernelGGL((march_kernel<MARCH_EMIT, true>), grid_rays, block, 0, s, ma);
  else
    hipLaunchKernelGGL((march_kernel<MARCH_EMIT, false>), grid_rays, block, 0, s, ma);
  prof_end(ctx, WS_K_MARCH_EMIT);

  prof_begin(ctx, WS_K_RESOLVE);
  hipLaunchKernelGGL(resolve_kernel, grid_tiles, block, 0, s, ra);
  prof_end(ctx, WS_K_RESOLVE);

  prof_begin(ctx, WS_K_MARCH_COLLECT);
  if (s0)
    hipLaunchKernelGGL((march_kernel<MARCH_COLLECT, true>), grid_rays, block, 0, s, ma);
  else
    hipLaunchKernelGGL((march_kernel<MARCH_COLLECT, false>), grid_rays, block, 0, s, ma);
  prof_end(ctx, WS_K_MARCH_COLLECT);

  prof_begin(ctx, WS_K_RESOLVE_LISTS);
  if (s0)
    hipLaunchKernelGGL((resolve_lists_kernel<true>), grid_lists, block, 0, s, ra);
  else
    hipLaunchKernelGGL((resolve_lists_kernel<false>), grid_lists, block, 0, s, ra);
  prof_end(ctx, WS_K_RESOLVE_LISTS);
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
  ia.dirty = m->dirty;
  ia.n_vox = m->n_vox;
  ia.n_tiles = m->n_tiles;
  ia.max_weight = m->max_weight;
  ia.tau = m->tau;
  ia.counters = m->counters;
  const dim3 block(256);
  // a non-default new_map must be streamed completely: untouched voxels carry entries too
  const bool dense = (m->integrate_mode == WS_INTEGRATE_DENSE) || !m->new_is_default;
  prof_begin(ctx, WS_K_INTEGRATE);
  if (dense)
  {
    int64_t blocks = ((m->n_vox >> 2) + 255) / 256;
    if (blocks > 256 * 16) blocks = 256 * 16;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(integrate_dense_kernel, dim3((unsigned)blocks), block, 0, s, ia);
    hipLaunchKernelGGL(clear_dirty_kernel, dim3((unsigned)((m->n_tiles + 255) / 256 > 2048 ? 2048 : (m->n_tiles + 255) / 256)),
                       block, 0, s, m->dirty, m->n_tiles);
  }
  else
  {
    const dim3 grid_tiles((unsigned)((m->n_tiles + 64 * 4 - 1) / (64 * 4)));
    hipLaunchKernelGGL(integrate_sparse_kernel, grid_tiles, block, 0, s, ia);
  }
  prof_end(ctx, WS_K_INTEGRATE);
  WS_HIP(hipGetLastError());
  m->new_is_default = true;
  return WS_OK;
}

} // namespace ws
