// api.hip — the extern "C" surface of libwarpsense_hip.so (include/warpsense_hip.h): handle
// management, host<->HBM transfers, stream ordering and hipEvent profiling.  No kernels here.
#include <cmath>
#include <chrono>
#include <atomic>
#include <algorithm>
#include <cstring>
#include <new>

#include "ws_internal.h"

namespace ws
{
static thread_local std::string g_last_error;

void set_error(const std::string &msg) { g_last_error = msg; }

int hip_fail(hipError_t e, const char *what, const char *file, int line)
{
  char buf[512];
  snprintf(buf, sizeof buf, "HIP error %d (%s) in %s at %s:%d", (int)e, hipGetErrorString(e), what, file, line);
  set_error(buf);
  return WS_ERR_HIP;
}

static int invalid(const char *msg)
{
  set_error(msg);
  return WS_ERR_INVALID;
}

static hipEvent_t take_event(ws_context *ctx)
{
  if (!ctx->pool.empty())
  {
    hipEvent_t e = ctx->pool.back();
    ctx->pool.pop_back();
    return e;
  }
  hipEvent_t e = nullptr;
  (void)hipEventCreate(&e);
  return e;
}

void prof_begin(ws_context *ctx, int cls)
{
  if (!(ctx->prof_mask & (1u << cls))) return;
  ws_context::Span sp;
  sp.a = take_event(ctx);
  sp.b = take_event(ctx);
  sp.cls = cls;
  (void)hipEventRecord(sp.a, ctx->stream);
  ctx->spans.push_back(sp);
}

void prof_end(ws_context *ctx, int cls)
{
  if (!(ctx->prof_mask & (1u << cls))) return;
  for (size_t i = ctx->spans.size(); i-- > 0;)
  {
    if (ctx->spans[i].cls == cls)
    {
      (void)hipEventRecord(ctx->spans[i].b, ctx->stream);
      return;
    }
  }
}

static void prof_resolve(ws_context *ctx)
{
  for (auto &sp : ctx->spans)
  {
    float ms = 0.f;
    if (hipEventSynchronize(sp.b) == hipSuccess && hipEventElapsedTime(&ms, sp.a, sp.b) == hipSuccess)
    {
      ctx->prof_ms[sp.cls] += ms;
      ctx->prof_n[sp.cls] += 1;
    }
    ctx->pool.push_back(sp.a);
    ctx->pool.push_back(sp.b);
  }
  ctx->spans.clear();
}

int check_all_equal_host(const uint32_t *data, int64_t n, uint32_t value)
{
  for (int64_t i = 0; i < n; ++i)
    if (data[i] != value) return 0;
  return 1;
}

static void copy3(int32_t dst[3], const int32_t src[3])
{
  dst[0] = src[0];
  dst[1] = src[1];
  dst[2] = src[2];
}

size_t reg_partials_bytes();
} // namespace ws
static int ctx_take_errors(ws_context *ctx);
// A resident server of ws_reg_iterate (reg_server_kernel) holds the context's stream until it has been idle for 50 us: whoever
// enqueues other work there asks it to leave first (one store into host-mapped memory; the work is ordered behind the kernel
// anyway).  The next ws_reg_iterate waits until that server is really gone and starts a new one -- behind the other work.
static void servers_leave(ws_context *ctx)
{
  if (!ctx) return;
  std::lock_guard<std::mutex> lock(ctx->lists_mu);
  for (ws_reg *r : ctx->regs)
  {
    const uint32_t id = r->srv_launch.load(std::memory_order_acquire);
    if (id == 0 || ws::reg_server_mail_exited(r->srv_mail) == id) continue;
    ws::reg_server_mail_stop(r->srv_mail, id);
    r->srv_stopping.store(true, std::memory_order_release);
  }
}
// every entry point that takes a map looks at the verdict of the scan in flight first (settle_tsdf, tsdf_update.hip)
#define WS_SETTLE(map_ptr)                                                   \
  do                                                                         \
  {                                                                          \
    servers_leave((map_ptr)->ctx);                                           \
    const int rc_settle__ = ws::settle_tsdf(const_cast<ws_map *>(map_ptr));  \
    if (rc_settle__ != WS_OK) return rc_settle__;                            \
  } while (0)

namespace ws
{
constexpr size_t AZ_ALLOC = 1024 * 64 + 8; // direction-bin histogram / offsets (tsdf_update.hip: AZ_BINS + 2 entries)
} // namespace ws

using namespace ws;

extern "C" {

const char *ws_last_error(void) { return g_last_error.c_str(); }
int ws_version(void) { return 1; }

int ws_ctx_create(int device_id, ws_context **out)
{
  if (!out) return invalid("ws_ctx_create: out is NULL");
  // the constant the kernels use for dz_per_distance must be what the reference computes (update_tsdf.cu:49-50)
  {
    float angle = 45.f / 128.f;
    int dz = (int)(std::tan(angle / 180 * M_PI) / 2.0 * MATRIX_RESOLUTION);
    if (dz != DZ_PER_DISTANCE) return invalid("ws_ctx_create: dz_per_distance constant mismatch");
  }
  if (device_id >= 0) WS_HIP(hipSetDevice(device_id));
  int dev = 0;
  WS_HIP(hipGetDevice(&dev));
  ws_context *ctx = new (std::nothrow) ws_context();
  if (!ctx) return invalid("ws_ctx_create: out of host memory");
  ctx->device = dev;
  hipError_t e = hipStreamCreateWithFlags(&ctx->own_stream, hipStreamNonBlocking);
  if (e != hipSuccess)
  {
    delete ctx;
    return hip_fail(e, "hipStreamCreateWithFlags", __FILE__, __LINE__);
  }
  ctx->stream = ctx->own_stream;
  *out = ctx;
  return WS_OK;
}

int ws_ctx_destroy(ws_context *ctx)
{
  if (!ctx) return WS_OK;
  (void)hipStreamSynchronize(ctx->stream);
  prof_resolve(ctx);
  for (auto e : ctx->pool) (void)hipEventDestroy(e);
  if (ctx->own_stream) (void)hipStreamDestroy(ctx->own_stream);
  delete ctx;
  return WS_OK;
}

int ws_ctx_set_stream(ws_context *ctx, void *hip_stream)
{
  if (!ctx) return invalid("ws_ctx_set_stream: ctx is NULL");
  servers_leave(ctx);
  hipStream_t next = hip_stream ? (hipStream_t)hip_stream : ctx->own_stream;
  // work already enqueued on the old stream finishes before anything goes to the new one -- unless one of them is
  // being captured into a graph (a synchronisation would invalidate the capture; the graph orders its own nodes)
  hipStreamCaptureStatus a = hipStreamCaptureStatusNone, b = hipStreamCaptureStatusNone;
  (void)hipStreamIsCapturing(ctx->stream, &a);
  (void)hipStreamIsCapturing(next, &b);
  if (a == hipStreamCaptureStatusNone && b == hipStreamCaptureStatusNone) WS_HIP(hipStreamSynchronize(ctx->stream));
  ctx->stream = next;
  return WS_OK;
}

int ws_sync(ws_context *ctx)
{
  if (!ctx) return invalid("ws_sync: ctx is NULL");
  servers_leave(ctx);
  std::vector<ws_map *> maps;
  {
    std::lock_guard<std::mutex> lock(ctx->lists_mu);
    maps = ctx->maps;
  }
  for (ws_map *m : maps)
  {
    const int rc = settle_tsdf(m); // (an aborted scan is repeated before the stream is drained)
    if (rc != WS_OK) return rc;
  }
  WS_HIP(hipStreamSynchronize(ctx->stream));
  return ctx_take_errors(ctx);
}

int ws_device_reset(void)
{
  WS_HIP(hipDeviceReset());
  return WS_OK;
}

// ------------------------------------------------------------------ maps
static void map_free_records(ws_map *m)
{
  if (m->rec) (void)hipFree(m->rec);
  if (m->big_keys) (void)hipFree(m->big_keys);
  m->rec = nullptr;
  m->big_keys = nullptr;
  m->sub_cap = 0;
  m->big_slots = 0;
}

// candidate records of the ray tails: the pool of sub-chunks (32 x 8 bytes, one tile each) and the (tile, entry number) ->
// entry hash for tiles of more than TILE_DIRECT sub-chunks (two slots per sub-chunk of the pool; keys, then uint32 values)
static int map_alloc_records(ws_map *m, uint64_t subs)
{
  if (subs < 32768) subs = 32768;
  if (subs > SUB_ID_LIMIT) subs = SUB_ID_LIMIT;
  map_free_records(m);
  uint64_t slots = 1u << 16;
  while (slots < 2 * subs && slots < (1ull << 31)) slots <<= 1;
  WS_HIP(hipMalloc((void **)&m->rec, (size_t)subs * SUB_RECS * sizeof(unsigned long long)));
  WS_HIP(hipMalloc((void **)&m->big_keys, (size_t)slots * (sizeof(unsigned long long) + sizeof(uint32_t))));
  m->sub_cap = (uint32_t)subs;
  m->big_slots = (uint32_t)slots;
  m->prepped = false; // the new hash is filled by the stand-alone preparation pass
  return WS_OK;
}

static int map_free(ws_map *m)
{
  if (!m) return WS_OK;
  (void)hipStreamSynchronize(m->ctx->stream);
  if (m->shift_stream) (void)hipStreamSynchronize(m->shift_stream);
  {
    std::lock_guard<std::mutex> lock(m->ctx->lists_mu);
    for (size_t i = 0; i < m->ctx->maps.size(); ++i)
      if (m->ctx->maps[i] == m)
      {
        m->ctx->maps.erase(m->ctx->maps.begin() + (long)i);
        break;
      }
  }
  map_free_records(m);
  void *ptrs[] = {m->data[0], m->data[1], m->vstate, m->az_hist, m->az_off, m->ray_bin, m->ray_order, m->fan_steps, m->rays, m->scan_dev, m->counters, m->tile_nsub,
                  m->tile_ent, m->tile_dirty, m->tile_list, m->block_stats, m->box_stage};
  for (void *p : ptrs)
    if (p) (void)hipFree(p);
  if (m->counters_host) (void)hipHostFree(m->counters_host);
  if (m->status_host) (void)hipHostFree(m->status_host);
  if (m->shift_open) delete m->shift_open;
  if (m->shift_stage_dev) (void)hipFree(m->shift_stage_dev);
  if (m->shift_stage_host) (void)hipHostFree(m->shift_stage_host);
  if (m->shift_event) (void)hipEventDestroy(m->shift_event);
  if (m->shift_stream) (void)hipStreamDestroy(m->shift_stream);
  delete m;
  return WS_OK;
}

// Device-side error bits (a ray outside the record's step / fan range, an aborted scan that was not repeated, internal
// checks) are OR-ed into a host-mapped word by the kernels.  Every entry point that has just synchronised the stream
// looks at it: an inexact map is reported ONCE, by the first such call after the update that produced it.
static int map_take_error(ws_map *m)
{
  if (!m || !m->status_host) return WS_OK;
  volatile uint32_t *st = m->status_host;
  const uint32_t bits = st[0];
  if (!bits) return WS_OK;
  __atomic_fetch_and(m->status_host, ~bits, __ATOMIC_RELAXED);
  m->last_error_bits |= bits;
  if (bits & 8u)
  {
    set_error("TSDF update: internal consistency check failed on the device (record tables); the map is not exact");
    return WS_ERR_INTERNAL;
  }
  if (bits & 4u)
  {
    set_error("TSDF update: a free-space step produced a candidate that is not free space; the map is not exact");
    return WS_ERR_INTERNAL;
  }
  if (bits & 1u)
  {
    set_error("TSDF update: record capacity exceeded, a TSDF update since the last check is not exact");
    return WS_ERR_CAPACITY;
  }
  set_error("TSDF update: a ray needs more ray steps or fan steps than the key of a scan of this many points holds (outside the range of the record fields: "
            "65 536 steps / 255 fan steps for small scans, 32 768 / 63 for 131 072 points, 8192 / 31 for a million) and was dropped");
  return WS_ERR_RANGE;
}

static int ctx_take_errors(ws_context *ctx)
{
  int rc = WS_OK;
  std::lock_guard<std::mutex> lock(ctx->lists_mu);
  for (ws_map *m : ctx->maps)
  {
    const int r = map_take_error(m);
    if (rc == WS_OK) rc = r;
  }
  return rc;
}

int ws_map_create(ws_context *ctx, const int32_t size[3], const int32_t pos[3], const int32_t offset[3],
                  const uint32_t *host_data, int32_t tau, int32_t max_weight, int32_t res, ws_map **out)
{
  if (!ctx || !size || !pos || !offset || !out) return invalid("ws_map_create: NULL argument");
  if (size[0] < 3 || size[1] < 3 || size[2] < 3) return invalid("ws_map_create: map sizes must be >= 3");
  if ((int64_t)size[0] * size[1] >= (1ll << 31) || size[2] > (1 << 20)) return invalid("ws_map_create: map too large (x*y must stay below 2^31 voxels)");
  if (size[0] >= (1 << 24) || size[1] >= (1 << 24) ||
      (int64_t)((size[0] + (1 << TILE_XB) - 1) >> TILE_XB) * ((size[1] + (1 << TILE_YB) - 1) >> TILE_YB) >= (1ll << 24) ||
      ((size[2] + (1 << TILE_ZB) - 1) >> TILE_ZB) >= (1 << 24))
    return invalid("ws_map_create: map too large (more than 2^24 tile columns)");
  if (res < 2) return invalid("ws_map_create: map_resolution must be >= 2 mm (the ray step is resolution/2)");
  if (tau <= 0 || tau > 32767) return invalid("ws_map_create: tau must fit the int16 TSDF value");
  ws_map *m = new (std::nothrow) ws_map();
  if (!m) return invalid("ws_map_create: out of host memory");
  m->ctx = ctx;
  for (int w = 0; w < 2; ++w)
  {
    copy3(m->par[w].size, size);
    copy3(m->par[w].pos, pos);
    copy3(m->par[w].offset, offset);
  }
  m->n_vox = (int64_t)size[0] * size[1] * size[2];
  m->ntx = (size[0] + (1 << TILE_XB) - 1) >> TILE_XB;
  m->nty = (size[1] + (1 << TILE_YB) - 1) >> TILE_YB;
  m->ntz = (size[2] + (1 << TILE_ZB) - 1) >> TILE_ZB;
  m->n_tiles = (int64_t)m->ntx * m->nty * m->ntz;
  if (m->n_tiles >= 0x7fffffffll)
  {
    delete m;
    return invalid("ws_map_create: more than 2^31 tiles");
  }
  m->tau = tau;
  m->max_weight = max_weight;
  m->res = res;

  hipStream_t s = ctx->stream;
  int rc = WS_OK;
#define TRY(call)                                                   \
  do                                                                \
  {                                                                 \
    hipError_t e__ = (call);                                        \
    if (e__ != hipSuccess)                                          \
    {                                                               \
      rc = hip_fail(e__, #call, __FILE__, __LINE__);                \
      map_free(m);                                                  \
      return rc;                                                    \
    }                                                               \
  } while (0)
  const size_t map_bytes = (size_t)m->n_vox * sizeof(uint32_t);
  // 16 bytes of slack: the tile kernels read the four voxels of a column as one access, also at the very end
  TRY(hipMalloc((void **)&m->data[0], map_bytes + 16));
  TRY(hipMalloc((void **)&m->data[1], map_bytes + 16));
  TRY(hipMalloc((void **)&m->vstate, 2 * vstate_plane_bytes(m->n_tiles)));
  TRY(hipMemsetAsync(m->vstate, 0, 2 * vstate_plane_bytes(m->n_tiles), s));
  TRY(hipMalloc((void **)&m->rays, MAX_SCAN_POINTS * ray_setup_bytes()));
  TRY(hipMalloc((void **)&m->az_hist, AZ_ALLOC * sizeof(uint32_t)));
  TRY(hipMalloc((void **)&m->az_off, AZ_ALLOC * sizeof(uint32_t)));
  TRY(hipMalloc((void **)&m->ray_bin, MAX_SCAN_POINTS * 2 * sizeof(uint32_t)));
  TRY(hipMalloc((void **)&m->ray_order, MAX_SCAN_POINTS * sizeof(uint32_t)));
  TRY(hipMalloc((void **)&m->fan_steps, WS_FAN_TABLE * sizeof(int32_t)));
  ws::fill_fan_steps(m->fan_steps_host, m->res, m->ntz, m->nty);
  TRY(hipMemcpyAsync(m->fan_steps, m->fan_steps_host, WS_FAN_TABLE * sizeof(int32_t), hipMemcpyHostToDevice, s));
  TRY(hipMemsetAsync(m->az_hist, 0, AZ_ALLOC * sizeof(uint32_t), s));
  TRY(hipMemsetAsync(m->az_off, 0, AZ_ALLOC * sizeof(uint32_t), s));
  TRY(hipMalloc((void **)&m->scan_dev, MAX_SCAN_POINTS * 3 * sizeof(int32_t)));
  TRY(hipMalloc((void **)&m->counters, sizeof(TsdfCounters)));
  TRY(hipMemsetAsync(m->counters, 0, sizeof(TsdfCounters), s));
  // per-tile bookkeeping of the scatter: entry count, entry table (64 x 4 bytes), two bytes, one 16-byte list entry per 1024 voxels
  TRY(hipMalloc((void **)&m->tile_nsub, (size_t)m->n_tiles * sizeof(uint32_t)));
  TRY(hipMalloc((void **)&m->tile_ent, (size_t)m->n_tiles * TILE_DIRECT * sizeof(uint32_t) + 256));
  TRY(hipMalloc((void **)&m->tile_dirty, 2 * tile_flag_plane_bytes(m->n_tiles)));
  TRY(hipMalloc((void **)&m->tile_list, (size_t)m->n_tiles * sizeof(TileEntry)));
  TRY(hipMemsetAsync(m->tile_nsub, 0, (size_t)m->n_tiles * sizeof(uint32_t), s));
  TRY(hipMemsetAsync(m->tile_dirty, 0, 2 * tile_flag_plane_bytes(m->n_tiles), s));
  TRY(hipMalloc((void **)&m->block_stats, (size_t)WS_BLOCK_STATS * sizeof(uint32_t)));
  TRY(hipMemsetAsync(m->block_stats, 0, (size_t)WS_BLOCK_STATS * sizeof(uint32_t), s));
  TRY(hipHostMalloc((void **)&m->counters_host, sizeof(TsdfCounters), hipHostMallocDefault));
  TRY(hipHostMalloc((void **)&m->status_host, 64, hipHostMallocMapped));
  std::memset(m->status_host, 0, 64);
  TRY(hipHostGetDevicePointer((void **)&m->status_dev, m->status_host, 0));
  // the pool of the ray tails' records: what a 131 072-point scan can need on a map of the reference's size (its record bound is
  // ~75 M at 50 mm) -- scaled down for small maps (1.1 GB for a 64^3 map): 32 records per voxel, scans of 16 384
  // points.  Only a first guess: a scan that needs more is aborted and repeated with a larger pool (settle_tsdf),
  // ws_tsdf_set_capacity() reserves up front.
  {
    const bool small_map = m->n_vox < (16ll << 20);
    const unsigned long long need0 = small_map ? std::max<unsigned long long>(1ull << 20, 32ull * (unsigned long long)m->n_vox) : (80ull << 20);
    rc = map_alloc_records(m, subs_for_scan(m, std::min<unsigned long long>(need0, 80ull << 20), small_map ? 16384 : 131072));
  }
  if (rc != WS_OK)
  {
    map_free(m);
    return rc;
  }
  const uint32_t def = ((uint32_t)tau & 0xffffu);
  if (host_data)
  {
    // DeviceMapMemWrapper(existing_map) x2: avg_map_ and new_map_ both start as the host map (update_tsdf.cu:135-136)
    TRY(hipMemcpyAsync(m->data[0], host_data, map_bytes, hipMemcpyHostToDevice, s));
    TRY(hipMemcpyAsync(m->data[1], m->data[0], map_bytes, hipMemcpyDeviceToDevice, s));
    TRY(hipStreamSynchronize(s));
    m->new_is_default = check_all_equal_host(host_data, m->n_vox, def) != 0;
  }
  else
  {
    rc = fill_u32(ctx, m->data[0], def, m->n_vox);
    if (rc == WS_OK) rc = fill_u32(ctx, m->data[1], def, m->n_vox);
    if (rc != WS_OK)
    {
      map_free(m);
      return rc;
    }
    m->new_is_default = true;
  }
#undef TRY
  {
    std::lock_guard<std::mutex> lock(ctx->lists_mu);
    ctx->maps.push_back(m);
  }
  *out = m;
  return WS_OK;
}

int ws_map_destroy(ws_map *map) { return map_free(map); }

int ws_map_set_params(ws_map *m, int which, const int32_t size[3], const int32_t pos[3], const int32_t offset[3])
{
  if (!m || !size || !pos || !offset || (which != WS_MAP_AVG && which != WS_MAP_NEW)) return invalid("ws_map_set_params: bad argument");
  WS_SETTLE(m);
  if ((int64_t)size[0] * size[1] * size[2] != m->n_vox) return invalid("ws_map_set_params: voxel count differs from the allocation");
  // parameters are kernel arguments: order against work already enqueued is automatic
  copy3(m->par[which].size, size);
  copy3(m->par[which].pos, pos);
  copy3(m->par[which].offset, offset);
  return WS_OK;
}

int ws_map_upload(ws_map *m, int which, const int32_t size[3], const int32_t pos[3], const int32_t offset[3],
                  const uint32_t *host_data)
{
  if (!host_data) return invalid("ws_map_upload: host_data is NULL");
  int rc = ws_map_set_params(m, which, size, pos, offset);
  if (rc != WS_OK) return rc;
  WS_HIP(hipMemcpyAsync(m->data[which], host_data, (size_t)m->n_vox * sizeof(uint32_t), hipMemcpyHostToDevice, m->ctx->stream));
  WS_HIP(hipStreamSynchronize(m->ctx->stream)); // cudaMemcpy of the reference is synchronous; the host buffer may be reused
  if (which == WS_MAP_NEW) m->new_is_default = check_all_equal_host(host_data, m->n_vox, ((uint32_t)m->tau & 0xffffu)) != 0;
  return WS_OK;
}

int ws_map_download(ws_map *m, int which, int32_t size[3], int32_t pos[3], int32_t offset[3], uint32_t *host_data)
{
  if (!m || (which != WS_MAP_AVG && which != WS_MAP_NEW)) return invalid("ws_map_download: bad argument");
  WS_SETTLE(m);
  if (size) copy3(size, m->par[which].size);
  if (pos) copy3(pos, m->par[which].pos);
  if (offset) copy3(offset, m->par[which].offset);
  if (host_data)
  {
    WS_HIP(hipMemcpyAsync(host_data, m->data[which], (size_t)m->n_vox * sizeof(uint32_t), hipMemcpyDeviceToHost, m->ctx->stream));
  }
  WS_HIP(hipStreamSynchronize(m->ctx->stream));
  return map_take_error(m);
}

// ---- box transfers: the device side of the map shift (only the slabs that leave / enter move)
static int box_check(ws_map *m, int which, const int32_t lo[3], const int32_t hi[3], int32_t ext[3], size_t *n)
{
  if (!m || !lo || !hi || (which != WS_MAP_AVG && which != WS_MAP_NEW)) return invalid("box transfer: bad argument");
  const MapParams &p = m->par[which];
  size_t cnt = 1;
  for (int k = 0; k < 3; ++k)
  {
    if (hi[k] < lo[k]) return invalid("box transfer: hi < lo");
    if (std::abs(lo[k] - p.pos[k]) > p.size[k] / 2 || std::abs(hi[k] - p.pos[k]) > p.size[k] / 2)
      return invalid("box transfer: box outside the local map window");
    ext[k] = hi[k] - lo[k] + 1;
    cnt *= (size_t)ext[k];
  }
  *n = cnt;
  if (cnt > m->box_stage_cap)
  {
    WS_HIP(hipStreamSynchronize(m->ctx->stream));
    if (m->box_stage) WS_HIP(hipFree(m->box_stage));
    m->box_stage = nullptr;
    m->box_stage_cap = 0;
    WS_HIP(hipMalloc((void **)&m->box_stage, cnt * sizeof(uint32_t)));
    m->box_stage_cap = cnt;
  }
  return WS_OK;
}

int ws_map_extract_box(ws_map *m, int which, const int32_t lo[3], const int32_t hi[3], uint32_t *host_out)
{
  if (!host_out) return invalid("ws_map_extract_box: host_out is NULL");
  WS_SETTLE(m);
  int32_t ext[3];
  size_t n = 0;
  int rc = box_check(m, which, lo, hi, ext, &n);
  if (rc != WS_OK) return rc;
  rc = launch_box_copy(m, m->par[which], which, lo, ext, m->box_stage, true, m->ctx->stream);
  if (rc != WS_OK) return rc;
  WS_HIP(hipMemcpyAsync(host_out, m->box_stage, n * sizeof(uint32_t), hipMemcpyDeviceToHost, m->ctx->stream));
  WS_HIP(hipStreamSynchronize(m->ctx->stream));
  return WS_OK;
}

int ws_map_insert_box(ws_map *m, int which, const int32_t lo[3], const int32_t hi[3], const uint32_t *host_in)
{
  if (!host_in) return invalid("ws_map_insert_box: host_in is NULL");
  WS_SETTLE(m);
  int32_t ext[3];
  size_t n = 0;
  int rc = box_check(m, which, lo, hi, ext, &n);
  if (rc != WS_OK) return rc;
  WS_HIP(hipMemcpyAsync(m->box_stage, host_in, n * sizeof(uint32_t), hipMemcpyHostToDevice, m->ctx->stream));
  rc = launch_box_copy(m, m->par[which], which, lo, ext, m->box_stage, false, m->ctx->stream);
  if (rc != WS_OK) return rc;
  WS_HIP(hipStreamSynchronize(m->ctx->stream)); // the host buffer may be reused by the caller
  if (which == WS_MAP_NEW) m->new_is_default = false;
  return WS_OK;
}

// ---- map shift off the scan path
static int shift_reserve(ws_map *m, size_t total)
{
  if (total <= m->shift_stage_cap) return WS_OK;
  hipError_t e = hipStreamSynchronize(m->ctx->stream);
  if (e == hipSuccess && m->shift_stage_dev) e = hipFree(m->shift_stage_dev);
  if (e == hipSuccess && m->shift_stage_host) e = hipHostFree(m->shift_stage_host);
  m->shift_stage_dev = nullptr;
  m->shift_stage_host = nullptr;
  m->shift_stage_cap = 0;
  if (e == hipSuccess) e = hipMalloc((void **)&m->shift_stage_dev, total * sizeof(uint32_t));
  if (e == hipSuccess) e = hipHostMalloc((void **)&m->shift_stage_host, total * sizeof(uint32_t), hipHostMallocDefault);
  if (e == hipSuccess && !m->shift_stream) e = hipStreamCreateWithFlags(&m->shift_stream, hipStreamNonBlocking);
  if (e == hipSuccess && !m->shift_event) e = hipEventCreateWithFlags(&m->shift_event, hipEventDisableTiming);
  if (e != hipSuccess) return hip_fail(e, "map shift staging", __FILE__, __LINE__);
  m->shift_stage_cap = total;
  return WS_OK;
}

int ws_shift_reserve(ws_map *m, uint64_t voxels)
{
  if (!m) return invalid("ws_shift_reserve: map is NULL");
  if (m->shift_open) return invalid("ws_shift_reserve: a shift is in flight");
  return shift_reserve(m, (size_t)voxels);
}

int ws_shift_begin(ws_map *m, const int32_t new_pos[3], uint32_t fill_entry, ws_shift **out)
{
  if (!m || !new_pos || !out) return invalid("ws_shift_begin: NULL argument");
  WS_SETTLE(m);
  if (m->shift_open) return invalid("ws_shift_begin: the previous shift of this map has not been ended (ws_shift_end)");
  // only new_map's WINDOW moves here, which is right iff it holds (tau, 0) everywhere -- not between ws_tsdf_scatter_dev and
  // ws_tsdf_integrate, nor for a map created from non-default host data that has not been integrated yet
  if (!m->new_is_default) return invalid("ws_shift_begin: new_map holds entries that have not been integrated (ws_tsdf_integrate first)");
  const MapParams &p0 = m->par[WS_MAP_AVG];
  ws_shift *sh = new (std::nothrow) ws_shift();
  if (!sh) return invalid("ws_shift_begin: out of host memory");
  sh->map = m;
  // plan: per axis, like HDF5LocalMap::shift (hdf5_local_map.cpp:53-118), with the window as it is when that axis moves
  int32_t pos[3] = {p0.pos[0], p0.pos[1], p0.pos[2]};
  size_t total = 0;
  for (int axis = 0; axis < 3; ++axis)
  {
    const int64_t d = (int64_t)new_pos[axis] - pos[axis];
    if (d == 0) continue;
    if (std::llabs((long long)d) > p0.size[axis])
    {
      delete sh;
      return invalid("ws_shift_begin: shift larger than the window");
    }
    const int i = sh->n++;
    int32_t start[3], end[3];
    for (int k = 0; k < 3; ++k)
    {
      start[k] = pos[k] - p0.size[k] / 2;
      end[k] = pos[k] + p0.size[k] / 2;
    }
    if (d > 0)
      end[axis] = start[axis] + (int32_t)d - 1;
    else
      start[axis] = end[axis] + (int32_t)d + 1;
    copy3(sh->leave_lo[i], start);
    copy3(sh->leave_hi[i], end);
    pos[axis] += (int32_t)d;
    for (int k = 0; k < 3; ++k)
    {
      start[k] = pos[k] - p0.size[k] / 2;
      end[k] = pos[k] + p0.size[k] / 2;
    }
    if (d > 0)
      start[axis] = end[axis] - ((int32_t)d - 1);
    else
      end[axis] = start[axis] - (int32_t)d - 1;
    copy3(sh->enter_lo[i], start);
    copy3(sh->enter_hi[i], end);
    sh->offset[i] = total;
    total += (size_t)(sh->leave_hi[i][0] - sh->leave_lo[i][0] + 1) * (size_t)(sh->leave_hi[i][1] - sh->leave_lo[i][1] + 1) *
             (size_t)(sh->leave_hi[i][2] - sh->leave_lo[i][2] + 1);
  }
  sh->total = total;
  hipStream_t s = m->ctx->stream;
  {
    // (a shift larger than anything reserved: the staging buffers grow, which waits for the stream once)
    const int rc0 = shift_reserve(m, total ? total : 1);
    if (rc0 != WS_OK)
    {
      delete sh;
      return rc0;
    }
  }
  // execute the plan on the map's stream.  The window moves in a COPY of the parameters (the box kernels take them by
  // value) and is committed only when every launch and copy has been enqueued: a HIP error half-way leaves the host's
  // view of both maps as it was.
  int rc = WS_OK;
  MapParams par[2] = {m->par[0], m->par[1]};
  for (int i = 0, axis = 0; i < sh->n && rc == WS_OK; ++i, ++axis)
  {
    while (new_pos[axis] == par[WS_MAP_AVG].pos[axis]) ++axis; // the axis slab i belongs to
    int32_t ext[3];
    for (int k = 0; k < 3; ++k) ext[k] = sh->leave_hi[i][k] - sh->leave_lo[i][k] + 1;
    rc = launch_box_copy(m, par[WS_MAP_AVG], WS_MAP_AVG, sh->leave_lo[i], ext, m->shift_stage_dev + sh->offset[i], true, s);
    const int32_t d = new_pos[axis] - par[WS_MAP_AVG].pos[axis];
    for (int w = 0; w < 2; ++w)
    {
      MapParams &p = par[w];
      p.pos[axis] += d;
      p.offset[axis] = (int32_t)((((int64_t)p.offset[axis] + d) % p.size[axis] + p.size[axis]) % p.size[axis]);
    }
    for (int k = 0; k < 3; ++k) ext[k] = sh->enter_hi[i][k] - sh->enter_lo[i][k] + 1;
    if (rc == WS_OK) rc = launch_box_fill(m, par[WS_MAP_AVG], WS_MAP_AVG, sh->enter_lo[i], ext, fill_entry, s);
    // new_map is (tau, 0) everywhere between updates (checked above): only its window moves (DeviceMapMemWrapper::update_params)
  }
  if (rc == WS_OK && total)
  {
    hipError_t e = hipEventRecord(m->shift_event, s);
    if (e == hipSuccess) e = hipStreamWaitEvent(m->shift_stream, m->shift_event, 0);
    if (e == hipSuccess) e = hipMemcpyAsync(m->shift_stage_host, m->shift_stage_dev, total * sizeof(uint32_t), hipMemcpyDeviceToHost, m->shift_stream);
    if (e != hipSuccess) rc = hip_fail(e, "ws_shift_begin copy", __FILE__, __LINE__);
  }
  if (rc != WS_OK)
  {
    delete sh;
    return rc;
  }
  m->par[0] = par[0];
  m->par[1] = par[1];
  m->shift_open = sh;
  *out = sh;
  return WS_OK;
}

int ws_shift_count(const ws_shift *sh) { return sh ? sh->n : 0; }

int ws_shift_entering(const ws_shift *sh, int i, int32_t lo[3], int32_t hi[3])
{
  if (!sh || i < 0 || i >= sh->n || !lo || !hi) return invalid("ws_shift_entering: bad argument");
  copy3(lo, sh->enter_lo[i]);
  copy3(hi, sh->enter_hi[i]);
  return WS_OK;
}

int ws_shift_wait(ws_shift *sh)
{
  if (!sh) return invalid("ws_shift_wait: shift is NULL");
  if (sh->map->shift_stream) WS_HIP(hipStreamSynchronize(sh->map->shift_stream));
  return WS_OK;
}

int ws_shift_slab(const ws_shift *sh, int i, int32_t lo[3], int32_t hi[3], const uint32_t **host_data)
{
  if (!sh || i < 0 || i >= sh->n || !lo || !hi || !host_data) return invalid("ws_shift_slab: bad argument");
  copy3(lo, sh->leave_lo[i]);
  copy3(hi, sh->leave_hi[i]);
  *host_data = sh->map->shift_stage_host + sh->offset[i];
  return WS_OK;
}

int ws_shift_end(ws_shift *sh)
{
  if (!sh) return WS_OK;
  if (sh->map->shift_stream) (void)hipStreamSynchronize(sh->map->shift_stream);
  if (sh->map->shift_open == sh) sh->map->shift_open = nullptr;
  delete sh;
  return WS_OK;
}

int ws_map_get_params(const ws_map *m, int which, int32_t size[3], int32_t pos[3], int32_t offset[3])
{
  if (!m || (which != WS_MAP_AVG && which != WS_MAP_NEW)) return invalid("ws_map_get_params: bad argument");
  if (size) copy3(size, m->par[which].size);
  if (pos) copy3(pos, m->par[which].pos);
  if (offset) copy3(offset, m->par[which].offset);
  return WS_OK;
}

void *ws_map_device_data(ws_map *m, int which)
{
  if (!m || (which != WS_MAP_AVG && which != WS_MAP_NEW)) return nullptr;
  return m->data[which];
}

int64_t ws_map_n_voxels(const ws_map *m) { return m ? m->n_vox : 0; }

// ------------------------------------------------------------------ TSDF update
int ws_tsdf_set_integrate(ws_map *m, int mode)
{
  if (m) WS_SETTLE(m);
  if (!m || (mode != WS_INTEGRATE_SPARSE && mode != WS_INTEGRATE_DENSE && mode != WS_INTEGRATE_SPARSE_SEPARATE))
    return invalid("ws_tsdf_set_integrate: bad argument");
  m->integrate_mode = mode;
  return WS_OK;
}

int ws_tsdf_set_capacity(ws_map *m, uint64_t records)
{
  if (!m) return invalid("ws_tsdf_set_capacity: map is NULL");
  WS_SETTLE(m);
  WS_HIP(hipStreamSynchronize(m->ctx->stream));
  return map_alloc_records(m, (records + SUB_RECS - 1) / SUB_RECS);
}

int ws_debug_tsdf_chunk_policy(ws_map *m, uint64_t budget_bytes, uint32_t est_shift)
{
  if (!m) return invalid("ws_debug_tsdf_chunk_policy: map is NULL");
  WS_SETTLE(m);
  m->chunk_budget_bytes = budget_bytes;
  m->est_shift = est_shift;
  return WS_OK;
}

static int too_many_points(size_t n)
{
  // update_tsdf.cu:146-150: message, no work
  char buf[160];
  snprintf(buf, sizeof buf, "TSDF update with %zu points, larger than the maximum of %zu", n, MAX_SCAN_POINTS);
  set_error(buf);
  return WS_ERR_TOO_MANY_POINTS;
}

int ws_tsdf_scatter_dev(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3])
{
  if (!m || (!xyz_dev && n) || !scanner_pos || !up) return invalid("ws_tsdf_scatter_dev: NULL argument");
  if (n > MAX_SCAN_POINTS) return too_many_points(n);
  servers_leave(m->ctx);
  int rc = launch_tsdf_scatter(m, xyz_dev, n, scanner_pos, up, false);
  if (rc == WS_OK && n) m->new_is_default = false; // new_map now carries the scan until it is integrated
  return rc;
}

int ws_tsdf_integrate(ws_map *m)
{
  if (!m) return invalid("ws_tsdf_integrate: map is NULL");
  servers_leave(m->ctx);
  const int rc0 = settle_tsdf(m);
  if (rc0 != WS_OK) return rc0;
  return launch_tsdf_integrate(m);
}

int ws_tsdf_update_dev(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3])
{
  if (!m || (!xyz_dev && n) || !scanner_pos || !up) return invalid("ws_tsdf_update_dev: NULL argument");
  if (n > MAX_SCAN_POINTS) return too_many_points(n);
  servers_leave(m->ctx);
  int rc;
  // with the default (sparse) integrate the tile resolve folds cu_avg_tsdf_krnl into its write-back (new_map stays
  // (tau, 0)); a non-default new_map is resolved on top of its entries and integrated by the dense pass
  prof_begin(m->ctx, WS_K_UPDATE);
  rc = launch_tsdf_scatter(m, xyz_dev, n, scanner_pos, up, m->integrate_mode == WS_INTEGRATE_SPARSE);
  if (rc == WS_OK) rc = launch_tsdf_integrate(m);
  if (m->pending.active.load(std::memory_order_relaxed)) m->pending.integrate_after = true; // (what a repeat of this scan has to be followed by; the caller holds the map exclusively here)
  prof_end(m->ctx, WS_K_UPDATE);
  return rc;
}

int ws_tsdf_update(ws_map *m, const int32_t *xyz_host, size_t n, const int32_t scanner_pos[3], const int32_t up[3])
{
  if (!m || (!xyz_host && n) || !scanner_pos || !up) return invalid("ws_tsdf_update: NULL argument");
  if (n > MAX_SCAN_POINTS) return too_many_points(n);
  if (n)
  {
    // (scan_dev is what a repeat of the PREVIOUS scan would read -- its verdict first, then the new points)
    WS_SETTLE(m);
    // pageable source: hipMemcpyAsync stages the data before it returns, like the reference's cudaMemcpy (update_tsdf.cu:152)
    WS_HIP(hipMemcpyAsync(m->scan_dev, xyz_host, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, m->ctx->stream));
  }
  return ws_tsdf_update_dev(m, m->scan_dev, n, scanner_pos, up);
}

int ws_tsdf_stats(ws_map *m, ws_tsdf_stats_t *out)
{
  if (!m || !out) return invalid("ws_tsdf_stats: NULL argument");
  {
    const int rcs = settle_tsdf(m);
    if (rcs != WS_OK) return rcs;
    const int rc0 = launch_tsdf_stats(m);
    if (rc0 != WS_OK) return rc0;
  }
  WS_HIP(hipMemcpyAsync(m->counters_host, m->counters, sizeof(TsdfCounters), hipMemcpyDeviceToHost, m->ctx->stream));
  WS_HIP(hipStreamSynchronize(m->ctx->stream));
  const TsdfCounters *c = m->counters_host;
  out->contested_voxels = c->last_contested;
  out->records = c->last_records;
  out->tiles = (int64_t)c->last_listed + c->last_unlisted;
  out->runs = c->last_runs;
  out->free_space_hits = c->last_free_keyed;
  out->record_slots = (int64_t)c->last_need;
  out->record_capacity = (int64_t)m->sub_cap * SUB_RECS;
  const int rc = map_take_error(m);
  out->error_flags = (int32_t)m->last_error_bits;
  out->hash_entries = (int32_t)m->status_host[10];
  m->last_error_bits = 0;
  return rc;
}

// ------------------------------------------------------------------ registration
int ws_reg_destroy(ws_reg *r)
{
  if (!r) return WS_OK;
  if (r->ctx)
  {
    servers_leave(r->ctx);
    std::lock_guard<std::mutex> lock(r->ctx->lists_mu);
    auto &v = r->ctx->regs;
    for (size_t i = 0; i < v.size(); ++i)
      if (v[i] == r)
      {
        v.erase(v.begin() + (long)i);
        break;
      }
  }
  (void)hipStreamSynchronize(r->ctx->stream);
  if (r->srv_mail) (void)hipHostFree(r->srv_mail);
  if (r->srv_ctl) (void)hipFree(r->srv_ctl);
  if (r->points) (void)hipFree(r->points);
  if (r->partials) (void)hipFree(r->partials);
  if (r->state) (void)hipFree(r->state);
  if (r->sums_dev) (void)hipFree(r->sums_dev);
  if (r->state_host) (void)hipHostFree(r->state_host);
  if (r->host_flag) (void)hipHostFree(r->host_flag);
  if (r->result_host) (void)hipHostFree(r->result_host);
  if (r->iter_host) (void)hipHostFree(r->iter_host);
  if (r->grid_bar) (void)hipFree(r->grid_bar);
  if (r->shard_arrived) (void)hipFree(r->shard_arrived);
  (void)ws_reg_peer_disconnect(r);
  if (r->mailbox) (void)hipFree(r->mailbox);
  if (r->peer_block_dev) (void)hipFree(r->peer_block_dev);
  delete r;
  return WS_OK;
}

static int reg_reserve(ws_reg *r, size_t n)
{
  if (n <= r->cap) return WS_OK;
  WS_HIP(hipStreamSynchronize(r->ctx->stream));
  if (r->points) WS_HIP(hipFree(r->points));
  r->points = nullptr;
  r->cap = 0;
  WS_HIP(hipMalloc((void **)&r->points, n * 3 * sizeof(int32_t)));
  r->cap = n;
  return WS_OK;
}

int ws_reg_create(ws_context *ctx, size_t max_points, ws_reg **out)
{
  if (!ctx || !out) return invalid("ws_reg_create: NULL argument");
  ws_reg *r = new (std::nothrow) ws_reg();
  if (!r) return invalid("ws_reg_create: out of host memory");
  r->ctx = ctx;
  if (max_points == 0) max_points = 128 * 1024; // registration.cu:261
  int rc = reg_reserve(r, max_points);
  hipError_t e = hipSuccess;
  if (rc == WS_OK) e = hipMalloc((void **)&r->partials, reg_partials_bytes());
  if (rc == WS_OK && e == hipSuccess) e = hipMalloc((void **)&r->state, 2 * sizeof(GnState));
  if (rc == WS_OK && e == hipSuccess) e = hipMalloc((void **)&r->sums_dev, 44 * sizeof(int64_t));
  if (rc == WS_OK && e == hipSuccess) e = hipHostMalloc((void **)&r->state_host, sizeof(GnState), hipHostMallocDefault);
  if (rc == WS_OK && e == hipSuccess) e = hipHostMalloc((void **)&r->host_flag, 64, hipHostMallocMapped);
  if (rc == WS_OK && e == hipSuccess) e = hipHostGetDevicePointer((void **)&r->host_flag_dev, r->host_flag, 0);
  if (rc == WS_OK && e == hipSuccess) e = hipHostMalloc((void **)&r->result_host, sizeof(GnState), hipHostMallocMapped);
  if (rc == WS_OK && e == hipSuccess) e = hipHostGetDevicePointer((void **)&r->result_host_dev, r->result_host, 0);
  if (rc == WS_OK && e == hipSuccess) e = hipHostMalloc((void **)&r->iter_host, 64 * sizeof(int64_t), hipHostMallocMapped);
  if (rc == WS_OK && e == hipSuccess) e = hipHostGetDevicePointer((void **)&r->iter_host_dev, r->iter_host, 0);
  if (rc == WS_OK && e == hipSuccess) std::memset(r->iter_host, 0, 64 * sizeof(int64_t));
  if (rc == WS_OK && e == hipSuccess) e = hipMemsetAsync(r->state, 0, 2 * sizeof(GnState), ctx->stream);
  if (rc == WS_OK && e == hipSuccess) e = hipMalloc((void **)&r->grid_bar, reg_barrier_bytes());
  if (rc == WS_OK && e == hipSuccess) e = hipMalloc((void **)&r->shard_arrived, 256);
  if (rc == WS_OK && e == hipSuccess) e = hipMemset(r->shard_arrived, 0, 256);
  if (rc == WS_OK && e == hipSuccess) r->loop_supported = reg_loop_supported(ctx->device);
  if (rc == WS_OK && e == hipSuccess) e = hipHostMalloc((void **)&r->srv_mail, reg_server_mail_bytes(), hipHostMallocMapped);
  if (rc == WS_OK && e == hipSuccess) e = hipHostGetDevicePointer((void **)&r->srv_mail_dev, r->srv_mail, 0);
  if (rc == WS_OK && e == hipSuccess) std::memset(r->srv_mail, 0, reg_server_mail_bytes());
  if (rc == WS_OK && e == hipSuccess) e = hipMalloc((void **)&r->srv_ctl, reg_server_ctl_bytes());
  if (rc == WS_OK && e == hipSuccess) e = hipMemset(r->srv_ctl, 0, reg_server_ctl_bytes());
  if (rc == WS_OK && e == hipSuccess)
  {
    if (const char *env = std::getenv("WS_REG_SERVER")) r->srv_enabled = std::atoi(env) != 0;
    if (const char *env = std::getenv("WS_REG_SERVER_IDLE_US")) r->srv_idle_us = (uint32_t)std::max(1, std::atoi(env));
  }
  if (rc != WS_OK || e != hipSuccess)
  {
    if (e != hipSuccess) rc = hip_fail(e, "ws_reg_create allocation", __FILE__, __LINE__);
    ws_reg_destroy(r);
    return rc;
  }
  {
    std::lock_guard<std::mutex> lock(ctx->lists_mu);
    ctx->regs.push_back(r);
  }
  *out = r;
  return WS_OK;
}

int ws_reg_prepare(ws_reg *r, const int32_t *xyz_host, size_t n)
{
  if (!r || (!xyz_host && n)) return invalid("ws_reg_prepare: NULL argument");
  servers_leave(r->ctx); // (a resident server of ws_reg_iterate keeps the cloud in registers: the copy below is ordered behind it)
  int rc = reg_reserve(r, n);
  if (rc != WS_OK) return rc;
  r->n = n;
  if (n) WS_HIP(hipMemcpyAsync(r->points, xyz_host, n * 3 * sizeof(int32_t), hipMemcpyHostToDevice, r->ctx->stream));
  return WS_OK;
}

int ws_reg_prepare_dev(ws_reg *r, const int32_t *xyz_dev, size_t n)
{
  if (!r || (!xyz_dev && n)) return invalid("ws_reg_prepare_dev: NULL argument");
  servers_leave(r->ctx);
  int rc = reg_reserve(r, n);
  if (rc != WS_OK) return rc;
  r->n = n;
  if (n) WS_HIP(hipMemcpyAsync(r->points, xyz_dev, n * 3 * sizeof(int32_t), hipMemcpyDeviceToDevice, r->ctx->stream));
  return WS_OK;
}

const int32_t *ws_reg_points_dev(const ws_reg *r, size_t *n)
{
  if (n) *n = r ? r->n : 0;
  return r ? r->points : nullptr;
}

// ws_reg_iterate through the resident server (reg_server_kernel): see there.  Returns WS_OK with the 44 sums, or an error.
static int reg_iterate_served(ws_reg *r, const ws_map *m, const float T[16], int32_t res, uint32_t flags, int64_t sums[44])
{
  auto exited = [&]() { return reg_server_mail_exited(r->srv_mail); };
  auto wait_gone = [&](uint32_t id) -> int {
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while (exited() != id)
      if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
      {
        WS_HIP(hipStreamSynchronize(r->ctx->stream));
        if (exited() != id)
        {
          set_error("ws_reg_iterate: the resident server did not leave");
          return WS_ERR_INTERNAL;
        }
      }
    return WS_OK;
  };
  uint32_t id = r->srv_launch.load(std::memory_order_acquire);
  bool alive = id != 0 && exited() != id;
  const MapParams &par = m->par[WS_MAP_AVG];
  const bool same = r->srv_sig.map == m && r->srv_sig.points == r->points && r->srv_sig.map_data == m->data[WS_MAP_AVG] && r->srv_sig.n == r->n &&
                    r->srv_sig.res == res && r->srv_sig.flags == flags && std::memcmp(&r->srv_sig.par, &par, sizeof par) == 0;
  if (alive && (!same || r->srv_stopping.load(std::memory_order_acquire)))
  {
    // somebody has enqueued other work behind that server (or the call is for another map / cloud): it must be gone before a
    // request may be written -- it would answer from the state it was launched with
    reg_server_mail_stop(r->srv_mail, id);
    const int rc = wait_gone(id);
    if (rc != WS_OK) return rc;
    alive = false;
  }
  uint32_t seq = r->srv_seq + 1;
  if (seq >= 0x7fffffffu) seq = 1;
  r->srv_seq = seq;
  reg_server_mail_write(r->srv_mail, T, seq);
  auto launch = [&]() -> int {
    id = ++r->srv_ids ? r->srv_ids : ++r->srv_ids;
    r->srv_sig.map = m;
    r->srv_sig.points = r->points;
    r->srv_sig.map_data = m->data[WS_MAP_AVG];
    r->srv_sig.n = r->n;
    r->srv_sig.res = res;
    r->srv_sig.flags = flags;
    r->srv_sig.par = par;
    r->srv_stopping.store(false, std::memory_order_release);
    r->srv_launch.store(id, std::memory_order_release);
    r->srv_launches += 1;
    return launch_reg_server(r, m, res, flags, id, r->srv_served, r->srv_idle_us);
  };
  if (!alive)
  {
    const int rc = launch();
    if (rc != WS_OK) return rc;
  }
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (!reg_server_mail_answer(r->srv_mail, seq, sums))
  {
    if (exited() == id)
    {
      // the server left (idle for too long, or asked to by another thread's call) without having seen this request
      if (reg_server_mail_answer(r->srv_mail, seq, sums)) break;
      const int rc = launch();
      if (rc != WS_OK) return rc;
    }
    if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
    {
      // Not an error yet: the server may be waiting in the stream behind somebody else's work (another thread's upload of a large
      // map takes longer than this).  Drain the stream the ordinary way -- a server that starts finds the request, answers it and
      // leaves when nothing else comes -- and only then look again; a kernel that never ends is the runtime's to report.
      WS_HIP(hipStreamSynchronize(r->ctx->stream));
      if (reg_server_mail_answer(r->srv_mail, seq, sums)) break;
      if (exited() == id)
      {
        // (it left on another thread's request without having seen this one: the next turn of the loop starts a new one)
        const int rc = launch();
        if (rc != WS_OK) return rc;
        WS_HIP(hipStreamSynchronize(r->ctx->stream));
        if (reg_server_mail_answer(r->srv_mail, seq, sums)) break;
      }
      set_error("ws_reg_iterate: the resident server did not answer");
      return WS_ERR_INTERNAL;
    }
  }
  r->srv_served = seq;
  return WS_OK;
}

int ws_reg_iterate(ws_reg *r, const ws_map *m, const float T[16], int32_t res, uint32_t flags, int64_t h[36], int64_t g[6],
                   int32_t *e, int32_t *c)
{
  if (!r || !m || !T || !h || !g || !e || !c) return invalid("ws_reg_iterate: NULL argument");
  if (res < 1) return invalid("ws_reg_iterate: map_resolution must be positive");
  int64_t sums[44];
  if (r->srv_enabled && r->loop_supported)
  {
    // (no WS_SETTLE here: that would ask the server to leave.  A scan whose verdict is open was enqueued by a call that has
    // already done so, and is settled now; a living server implies a settled map)
    const int rcs = ws::settle_tsdf(const_cast<ws_map *>(m));
    if (rcs != WS_OK) return rcs;
    const int rc = reg_iterate_served(r, m, T, res, flags, sums);
    if (rc != WS_OK) return rc;
  }
  else
  {
    WS_SETTLE(m);
    // One launch, nothing copied by the runtime: the pose travels in the kernel arguments (registration.cu:351 copies it), the
    // sums come back through host-mapped memory with the call's sequence number behind them (registration.cu:356-365 copies
    // four results and adds 32 partials up on the host).  The caller cannot go on without them, so the wait is a spin on that
    // word -- bounded: after 20 ms the stream is synchronised the ordinary way, and a kernel that never ends is the runtime's to report.
    const uint32_t seq = ++r->iter_seq ? r->iter_seq : ++r->iter_seq; // (never 0: the block starts zeroed)
    int rc = launch_reg_host_iter(r, m, T, res, flags, seq);
    if (rc != WS_OK) return rc;
    const volatile int64_t *done = r->iter_host + 44;
    const auto t0 = std::chrono::steady_clock::now();
    uint32_t spins = 0;
    while ((uint32_t)*done != seq)
    {
      if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
      {
        WS_HIP(hipStreamSynchronize(r->ctx->stream));
        if ((uint32_t)*done != seq)
        {
          set_error("ws_reg_iterate: the launch ended without its result");
          return WS_ERR_INTERNAL;
        }
        break;
      }
    }
    std::atomic_thread_fence(std::memory_order_acquire);
    std::memcpy(sums, r->iter_host, sizeof sums);
  }
  std::memcpy(h, sums, 36 * sizeof(int64_t));
  std::memcpy(g, sums + 36, 6 * sizeof(int64_t));
  *e = (int32_t)sums[42];
  *c = (int32_t)sums[43];
  return map_take_error(const_cast<ws_map *>(m));
}

// test / tuning entry: the resident server of ws_reg_iterate on or off, its idle time; returns the servers launched so far
int ws_debug_reg_mail_selftest(void) { return reg_server_mail_selftest(); }

int ws_debug_reg_server(ws_reg *r, int32_t enable, int32_t idle_us, int32_t *launches)
{
  if (!r) return invalid("ws_debug_reg_server: reg is NULL");
  servers_leave(r->ctx);
  if (enable >= 0) r->srv_enabled = enable ? 1 : 0;
  if (idle_us > 0) r->srv_idle_us = (uint32_t)idle_us;
  if (launches) *launches = (int32_t)r->srv_launches;
  return WS_OK;
}

int ws_reg_begin(ws_reg *r, const float T_in[16], int32_t max_iterations, float it_weight_gradient, float epsilon)
{
  if (!r || !T_in) return invalid("ws_reg_begin: NULL argument");
  GnState *h = r->state_host;
  // the pinned staging block may still be read by an earlier async copy
  WS_HIP(hipStreamSynchronize(r->ctx->stream));
  std::memset(h, 0, sizeof(GnState));
  std::memcpy(h->core.T, T_in, 16 * sizeof(float));
  // Point center = total_transform.block<3,1>(0,3).cast<int>() — tsdf_registration.cpp:33
  for (int k = 0; k < 3; ++k) h->core.center[k] = (int32_t)T_in[12 + k];
  h->core.alpha = 0.f;
  h->core.it_weight_gradient = it_weight_gradient;
  h->core.epsilon = epsilon;
  h->core.max_iterations = max_iterations;
  *(volatile int32_t *)r->host_flag = 0;
  WS_HIP(hipMemcpyAsync(&r->state[0], h, sizeof(GnState), hipMemcpyHostToDevice, r->ctx->stream));
  WS_HIP(hipMemcpyAsync(&r->state[1], h, sizeof(GnState), hipMemcpyHostToDevice, r->ctx->stream));
  r->latest = 0;
  return WS_OK;
}

int ws_reg_accumulate_dev(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, size_t first, size_t count, int64_t *sums_dev)
{
  if (!r || !m || !sums_dev) return invalid("ws_reg_accumulate_dev: NULL argument");
  WS_SETTLE(m);
  return launch_reg_accumulate(r, m, nullptr, res, flags, first, count, sums_dev);
}

int ws_reg_iterate_shard_dev(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, size_t first, size_t count, int64_t *sums_dev,
                             int32_t apply_previous)
{
  if (!r || !m || !sums_dev) return invalid("ws_reg_iterate_shard_dev: NULL argument");
  WS_SETTLE(m);
  return launch_reg_shard(r, m, res, flags, first, count, sums_dev, apply_previous);
}

int ws_reg_solve_dev(ws_reg *r, const int64_t *sums_dev)
{
  if (!r || !sums_dev) return invalid("ws_reg_solve_dev: NULL argument");
  return launch_reg_solve(r, sums_dev);
}

int ws_reg_poll(ws_reg *r, int32_t *finished, int32_t *iterations, float T_out[16])
{
  if (!r) return invalid("ws_reg_poll: reg is NULL");
  WS_HIP(hipMemcpyAsync(r->state_host, &r->state[r->latest], sizeof(GnState), hipMemcpyDeviceToHost, r->ctx->stream));
  WS_HIP(hipStreamSynchronize(r->ctx->stream));
  const GnCore *h = &r->state_host->core;
  if (finished) *finished = (h->finished || h->iterations >= h->max_iterations) ? 1 : 0;
  if (iterations) *iterations = h->iterations;
  if (T_out) std::memcpy(T_out, h->T, 16 * sizeof(float));
  return WS_OK;
}

// the host side of one resident launch: spin on the flag the kernel raises behind its result (host-mapped memory)
static int wait_resident_loop(ws_reg *r)
{
  volatile int32_t *done = r->host_flag;
  const auto t0 = std::chrono::steady_clock::now();
  uint32_t spins = 0;
  while (*done == 0)
  {
    if ((++spins & 0xfffu) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20))
    {
      WS_HIP(hipStreamSynchronize(r->ctx->stream));
      break;
    }
  }
  std::atomic_thread_fence(std::memory_order_acquire);
  return WS_OK;
}

int ws_register_cloud(ws_reg *r, const ws_map *m, const float T_in[16], int32_t max_iterations, float it_weight_gradient,
                      float epsilon, int32_t res, uint32_t flags, float T_out[16], int32_t *iterations)
{
  if (!r || !m || !T_in || !T_out) return invalid("ws_register_cloud: NULL argument");
  WS_SETTLE(m);
  if (res < 1) return invalid("ws_register_cloud: map_resolution must be positive");
  if (r->loop_mode == WS_REG_LOOP_RESIDENT && r->loop_supported)
  {
    // one launch: the 256 workgroups stay resident and meet at a grid barrier between iterations.  The initial state
    // travels in the kernel arguments and the final state comes back through host-mapped memory, so the host neither
    // waits for earlier work on the stream before enqueueing nor copies anything afterwards.
    GnCore init;
    std::memset(&init, 0, sizeof init);
    std::memcpy(init.T, T_in, 16 * sizeof(float));
    for (int k = 0; k < 3; ++k) init.center[k] = (int32_t)T_in[12 + k]; // tsdf_registration.cpp:33
    init.it_weight_gradient = it_weight_gradient;
    init.epsilon = epsilon;
    init.max_iterations = max_iterations;
    *(volatile int32_t *)r->host_flag = 0; // nothing on the stream writes it any more: every earlier registration was waited for
    int rc = launch_reg_loop(r, m, res, flags, init);
    if (rc != WS_OK) return rc;
    r->latest = 0;
    // The kernel raises the flag in host-mapped memory after its result (release at system scope).  Spinning on it costs a
    // microsecond or two; waking up from hipStreamSynchronize costs tens (measured: 84 -> ~35 us between the end of a
    // registration and the first kernel of the next scan).  Bounded: a kernel that never finishes is the runtime's to report.
    rc = wait_resident_loop(r);
    if (rc != WS_OK) return rc;
    const GnCore *h = &r->result_host->core;
    if (!h->error)
    {
      std::memcpy(T_out, h->T, 16 * sizeof(float));
      if (iterations) *iterations = h->iterations;
      return map_take_error(const_cast<ws_map *>(m));
    }
    // The grid barrier timed out: another kernel held compute units the resident grid needs (its workgroups must all
    // be on the chip at once).  Nothing was lost — the loop state is only ever produced from complete sums — so the
    // registration simply runs again with one launch per iteration, which needs no co-residency.
    r->resident_fallbacks += 1;
  }
  int rc = ws_reg_begin(r, T_in, max_iterations, it_weight_gradient, epsilon);
  if (rc != WS_OK) return rc;
  // One launch per iteration: launch k applies update k (from the partial sums launch k-1 left behind) and
  // accumulates for iteration k.  The host just enqueues; the device raises a flag in host-mapped memory on
  // convergence so the host can stop early (launches already enqueued exit at once).
  const volatile int32_t *flag = r->host_flag;
  int launched = 0;
  for (int k = 0; k <= max_iterations; ++k)
  {
    if (*flag) break;
    rc = launch_reg_iteration(r, m, res, flags, k);
    if (rc != WS_OK) return rc;
    launched = k + 1;
  }
  r->latest = launched > 0 ? ((launched - 1) & 1) : 0;
  int fin = 0, iters = 0;
  rc = ws_reg_poll(r, &fin, &iters, T_out);
  if (rc != WS_OK) return rc;
  if (iterations) *iterations = iters;
  return map_take_error(const_cast<ws_map *>(m));
}

// ------------------------------------------------------------------ multi-GPU resident loop
static int peer_own_mailbox(ws_reg *r)
{
  if (r->mailbox) return WS_OK;
  // fine-grained: coherent for system-scope atomics from every GPU that maps it (and for the polls of the owner)
  WS_HIP(hipExtMallocWithFlags(&r->mailbox, 4096, hipDeviceMallocFinegrained));
  WS_HIP(hipMemset(r->mailbox, 0, 4096));
  return WS_OK;
}

int ws_reg_peer_mailbox(ws_reg *r, void *ipc_handle_out)
{
  if (!r) return invalid("ws_reg_peer_mailbox: reg is NULL");
  static_assert(sizeof(hipIpcMemHandle_t) == WS_IPC_HANDLE_BYTES, "WS_IPC_HANDLE_BYTES");
  int rc = peer_own_mailbox(r);
  if (rc != WS_OK) return rc;
  if (ipc_handle_out)
  {
    hipIpcMemHandle_t h;
    WS_HIP(hipIpcGetMemHandle(&h, r->mailbox));
    std::memcpy(ipc_handle_out, &h, sizeof h);
  }
  return WS_OK;
}

int ws_reg_peer_disconnect(ws_reg *r)
{
  if (!r) return WS_OK;
  if (r->peer_world) (void)hipStreamSynchronize(r->ctx->stream);
  for (int i = 0; i < 8; ++i)
  {
    if (r->peer_opened[i] && r->peer_mailbox[i]) (void)hipIpcCloseMemHandle(r->peer_mailbox[i]);
    r->peer_opened[i] = false;
    r->peer_mailbox[i] = nullptr;
  }
  r->peer_world = 0;
  return WS_OK;
}

static int peer_finish_connect(ws_reg *r, int rank, int world, int blocks)
{
  if (blocks <= 0) blocks = reg_default_blocks();
  if (blocks % reg_groups() != 0 || blocks > reg_default_blocks()) return invalid("ws_reg_peer_connect: blocks must be a multiple of 8, at most 256");
  std::vector<unsigned char> image(reg_peer_block_bytes());
  reg_peer_block_fill(image.data(), r->peer_mailbox, rank, world);
  if (!r->peer_block_dev) WS_HIP(hipMalloc(&r->peer_block_dev, reg_peer_block_bytes()));
  WS_HIP(hipMemcpy(r->peer_block_dev, image.data(), image.size(), hipMemcpyHostToDevice)); // also zeroes `then`: the mailboxes are fresh
  WS_HIP(hipMemset(r->mailbox, 0, reg_mailbox_bytes()));
  r->peer_rank = rank;
  r->peer_world = world;
  r->peer_blocks = blocks;
  r->peer_dirty = false;
  return WS_OK;
}

int ws_reg_peer_connect(ws_reg *r, int32_t rank, int32_t world, const void *ipc_handles, int32_t blocks)
{
  if (!r || !ipc_handles) return invalid("ws_reg_peer_connect: NULL argument");
  if (world < 1 || world > 8 || rank < 0 || rank >= world) return invalid("ws_reg_peer_connect: 1 <= world <= 8, 0 <= rank < world");
  int rc = peer_own_mailbox(r);
  if (rc != WS_OK) return rc;
  (void)ws_reg_peer_disconnect(r);
  // the mailboxes of ranks on other GPUs are reached over xGMI: peer access to every visible device (already enabled / not
  // possible are both fine here: the open below decides)
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) == hipSuccess)
    for (int d = 0; d < n_dev; ++d)
      if (d != r->ctx->device)
      {
        int can = 0;
        if (hipDeviceCanAccessPeer(&can, r->ctx->device, d) == hipSuccess && can) (void)hipDeviceEnablePeerAccess(d, 0);
      }
  (void)hipGetLastError();
  for (int i = 0; i < world; ++i)
  {
    if (i == rank)
    {
      r->peer_mailbox[i] = r->mailbox;
      continue;
    }
    hipIpcMemHandle_t h;
    std::memcpy(&h, static_cast<const unsigned char *>(ipc_handles) + (size_t)i * sizeof h, sizeof h);
    void *p = nullptr;
    hipError_t e = hipIpcOpenMemHandle(&p, h, hipIpcMemLazyEnablePeerAccess);
    if (e != hipSuccess)
    {
      (void)ws_reg_peer_disconnect(r);
      return hip_fail(e, "hipIpcOpenMemHandle (mailbox of a peer rank)", __FILE__, __LINE__);
    }
    r->peer_mailbox[i] = p;
    r->peer_opened[i] = true;
  }
  return peer_finish_connect(r, rank, world, blocks);
}

int ws_reg_peer_connect_local(ws_reg *r, int32_t rank, int32_t world, ws_reg *const *regs, int32_t blocks)
{
  if (!r || !regs) return invalid("ws_reg_peer_connect_local: NULL argument");
  if (world < 1 || world > 8 || rank < 0 || rank >= world || regs[rank] != r) return invalid("ws_reg_peer_connect_local: regs[rank] must be reg, world <= 8");
  (void)ws_reg_peer_disconnect(r);
  for (int i = 0; i < world; ++i)
  {
    if (!regs[i]) return invalid("ws_reg_peer_connect_local: NULL rank");
    const int rc = peer_own_mailbox(regs[i]);
    if (rc != WS_OK) return rc;
    r->peer_mailbox[i] = regs[i]->mailbox;
  }
  return peer_finish_connect(r, rank, world, blocks);
}

int ws_reg_peer_reset(ws_reg *r)
{
  if (!r || !r->peer_world) return invalid("ws_reg_peer_reset: not connected");
  WS_HIP(hipStreamSynchronize(r->ctx->stream));
  return peer_finish_connect(r, r->peer_rank, r->peer_world, r->peer_blocks);
}

int ws_register_cloud_peers(ws_reg *r, const ws_map *m, size_t first, size_t count, const float T_in[16], int32_t max_iterations,
                            float it_weight_gradient, float epsilon, int32_t res, uint32_t flags, float T_out[16], int32_t *iterations)
{
  if (!r || !m || !T_in || !T_out) return invalid("ws_register_cloud_peers: NULL argument");
  WS_SETTLE(m);
  if (!r->peer_world) return invalid("ws_register_cloud_peers: ws_reg_peer_connect first");
  // An exchange that was given up leaves partial additions in the mailboxes and no saved snapshot: a rank's stale addition
  // plus its next one would reach count == world and pass for the all-rank total.  Nothing runs until the mailboxes are fresh.
  if (r->peer_dirty) return invalid("ws_register_cloud_peers: the last exchange failed; call ws_reg_peer_reset on every rank (between two barriers) or reconnect first");
  if (res < 1) return invalid("ws_register_cloud_peers: map_resolution must be positive");
  GnCore init;
  std::memset(&init, 0, sizeof init);
  std::memcpy(init.T, T_in, 16 * sizeof(float));
  for (int k = 0; k < 3; ++k) init.center[k] = (int32_t)T_in[12 + k]; // tsdf_registration.cpp:33
  init.it_weight_gradient = it_weight_gradient;
  init.epsilon = epsilon;
  init.max_iterations = max_iterations;
  *(volatile int32_t *)r->host_flag = 0;
  r->peer_dirty = true; // until this exchange has completed on this rank
  int rc = launch_reg_loop(r, m, res, flags, init, true, first, count);
  if (rc != WS_OK) return rc;
  r->latest = 0;
  rc = wait_resident_loop(r);
  if (rc != WS_OK) return rc;
  const GnCore *h = &r->result_host->core;
  if (h->error)
  {
    // a rank did not deliver (its kernel was not on the chip, or the process is gone): every rank times out within one
    // exchange of the first.  The caller re-runs the registration through the RCCL route (warpsense_amd.dist does) after
    // ws_reg_peer_reset on every rank.
    set_error("ws_register_cloud_peers: the exchange with the peer ranks timed out");
    return WS_ERR_TIMEOUT;
  }
  r->peer_dirty = false;
  std::memcpy(T_out, h->T, 16 * sizeof(float));
  if (iterations) *iterations = h->iterations;
  return map_take_error(const_cast<ws_map *>(m));
}

int ws_reg_set_loop(ws_reg *r, int mode)
{
  if (!r || (mode != WS_REG_LOOP_RESIDENT && mode != WS_REG_LOOP_LAUNCHES)) return invalid("ws_reg_set_loop: bad argument");
  r->loop_mode = mode;
  return WS_OK;
}

int ws_debug_block_stats(ws_map *m, uint32_t *out, size_t words)
{
  if (!m || !out) return invalid("ws_debug_block_stats: NULL argument");
  WS_SETTLE(m);
  if (words > WS_BLOCK_STATS) words = WS_BLOCK_STATS;
  WS_HIP(hipMemcpyAsync(out, m->block_stats, words * sizeof(uint32_t), hipMemcpyDeviceToHost, m->ctx->stream));
  WS_HIP(hipStreamSynchronize(m->ctx->stream));
  return WS_OK;
}

int ws_debug_reg_stall(ws_reg *r, int32_t stall_next, int32_t *fallbacks)
{
  if (!r) return invalid("ws_debug_reg_stall: NULL argument");
  r->debug_stall_next = stall_next ? 1 : 0;
  if (fallbacks) *fallbacks = r->resident_fallbacks;
  return WS_OK;
}

int ws_debug_reg_sums(ws_reg *r, int64_t sums_out[44])
{
  if (!r || !sums_out) return invalid("ws_debug_reg_sums: NULL argument");
  WS_HIP(hipMemcpyAsync(sums_out, r->state[r->latest].sums, 44 * sizeof(int64_t), hipMemcpyDeviceToHost, r->ctx->stream));
  WS_HIP(hipStreamSynchronize(r->ctx->stream));
  return WS_OK;
}

int ws_debug_solve6(ws_context *ctx, const double *A, const double *b, size_t n, double *x, int32_t *status)
{
  if (!ctx || !A || !b || !x || !status) return invalid("ws_debug_solve6: NULL argument");
  double *dA = nullptr, *db = nullptr, *dx = nullptr;
  int32_t *ds = nullptr;
  hipError_t e = hipMalloc((void **)&dA, (n ? n : 1) * 36 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&db, (n ? n : 1) * 6 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&dx, (n ? n : 1) * 6 * sizeof(double));
  if (e == hipSuccess) e = hipMalloc((void **)&ds, (n ? n : 1) * sizeof(int32_t));
  int rc = WS_OK;
  if (e == hipSuccess && n)
  {
    hipStream_t s = ctx->stream;
    e = hipMemcpyAsync(dA, A, n * 36 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(db, b, n * 6 * sizeof(double), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) rc = launch_solve6_test(ctx, dA, db, n, dx, ds);
    if (e == hipSuccess && rc == WS_OK) e = hipMemcpyAsync(x, dx, n * 6 * sizeof(double), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess && rc == WS_OK) e = hipMemcpyAsync(status, ds, n * sizeof(int32_t), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
  }
  if (dA) (void)hipFree(dA);
  if (db) (void)hipFree(db);
  if (dx) (void)hipFree(dx);
  if (ds) (void)hipFree(ds);
  if (e != hipSuccess) return hip_fail(e, "ws_debug_solve6", __FILE__, __LINE__);
  return rc;
}

// ------------------------------------------------------------------ scan pre-processing
int ws_scan_destroy(ws_scan *sc)
{
  if (!sc) return WS_OK;
  (void)hipStreamSynchronize(sc->ctx->stream);
  void *dev[] = {sc->in_stage, sc->tmp, sc->slot_of, sc->keys, sc->first, sc->wg_count, sc->wg_off, sc->counters, sc->out};
  for (void *p : dev)
    if (p) (void)hipFree(p);
  if (sc->host_count) (void)hipHostFree(sc->host_count);
  delete sc;
  return WS_OK;
}

int ws_scan_create(ws_context *ctx, size_t max_points, ws_scan **out)
{
  if (!ctx || !out) return invalid("ws_scan_create: NULL argument");
  if (max_points == 0) max_points = 128 * 1024;
  if (max_points > (1u << 30)) return invalid("ws_scan_create: too many points");
  ws_scan *sc = new (std::nothrow) ws_scan();
  if (!sc) return invalid("ws_scan_create: out of host memory");
  sc->ctx = ctx;
  sc->cap = max_points;
  sc->table_slots = pre_table_slots(max_points);
  const size_t blocks = (max_points + 255) / 256;
  hipError_t e = hipMalloc((void **)&sc->tmp, max_points * 3 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&sc->out, max_points * 3 * sizeof(int32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&sc->slot_of, max_points * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&sc->keys, sc->table_slots * sizeof(uint64_t));
  if (e == hipSuccess) e = hipMalloc((void **)&sc->first, sc->table_slots * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&sc->wg_count, blocks * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&sc->wg_off, blocks * sizeof(uint32_t));
  if (e == hipSuccess) e = hipMalloc((void **)&sc->counters, 64);
  if (e == hipSuccess) e = hipHostMalloc((void **)&sc->host_count, 64, hipHostMallocMapped);
  if (e == hipSuccess) e = hipHostGetDevicePointer((void **)&sc->host_count_dev, sc->host_count, 0);
  if (e != hipSuccess)
  {
    const int rc = hip_fail(e, "ws_scan_create allocation", __FILE__, __LINE__);
    ws_scan_destroy(sc);
    return rc;
  }
  *out = sc;
  return WS_OK;
}

static int scan_run(ws_scan *sc, const float *xyz_dev, size_t n, size_t stride, const float pose[16], int32_t res, size_t *n_out)
{
  int32_t M[16];
  for (int k = 0; k < 16; ++k) M[k] = (int32_t)(pose[k] * (float)MATRIX_RESOLUTION); // to_int_mat, util/util.h:8-11
  int rc = launch_scan_preprocess(sc, xyz_dev, n, stride, M, res);
  if (rc != WS_OK) return rc;
  uint32_t counters[2] = {0, 0};
  WS_HIP(hipMemcpyAsync(counters, sc->counters, sizeof counters, hipMemcpyDeviceToHost, sc->ctx->stream));
  WS_HIP(hipStreamSynchronize(sc->ctx->stream));
  sc->n_out = counters[0];
  if (n_out) *n_out = sc->n_out;
  if (counters[1] & 1u)
  {
    set_error("ws_scan_preprocess: a transformed coordinate is beyond +-2^20 mm");
    return WS_ERR_RANGE;
  }
  return WS_OK;
}

int ws_scan_preprocess_dev(ws_scan *sc, const float *xyz_dev, size_t n, size_t stride, const float pose[16], int32_t res, size_t *n_out)
{
  if (!sc || (!xyz_dev && n) || !pose) return invalid("ws_scan_preprocess_dev: NULL argument");
  if (stride < 3) return invalid("ws_scan_preprocess_dev: a point needs at least 3 floats");
  if (res < 1) return invalid("ws_scan_preprocess_dev: map_resolution must be positive");
  if (n > sc->cap) return invalid("ws_scan_preprocess_dev: more points than ws_scan_create reserved");
  return scan_run(sc, xyz_dev, n, stride, pose, res, n_out);
}

int ws_scan_preprocess(ws_scan *sc, const float *xyz_host, size_t n, size_t stride, const float pose[16], int32_t res, size_t *n_out)
{
  if (!sc || (!xyz_host && n) || !pose) return invalid("ws_scan_preprocess: NULL argument");
  if (stride < 3) return invalid("ws_scan_preprocess: a point needs at least 3 floats");
  if (res < 1) return invalid("ws_scan_preprocess: map_resolution must be positive");
  if (n > sc->cap) return invalid("ws_scan_preprocess: more points than ws_scan_create reserved");
  const size_t floats = n * stride;
  if (floats > sc->in_stage_floats)
  {
    WS_HIP(hipStreamSynchronize(sc->ctx->stream));
    if (sc->in_stage) WS_HIP(hipFree(sc->in_stage));
    sc->in_stage = nullptr;
    sc->in_stage_floats = 0;
    WS_HIP(hipMalloc((void **)&sc->in_stage, floats * sizeof(float)));
    sc->in_stage_floats = floats;
  }
  if (floats) WS_HIP(hipMemcpyAsync(sc->in_stage, xyz_host, floats * sizeof(float), hipMemcpyHostToDevice, sc->ctx->stream));
  return scan_run(sc, sc->in_stage, n, stride, pose, res, n_out);
}

const int32_t *ws_scan_points_dev(const ws_scan *sc) { return sc ? sc->out : nullptr; }

int ws_scan_download(ws_scan *sc, int32_t *xyz_host, size_t capacity_points, size_t *n_out)
{
  if (!sc || !n_out) return invalid("ws_scan_download: NULL argument");
  *n_out = sc->n_out;
  if (sc->n_out == 0) return WS_OK;
  if (!xyz_host || capacity_points < sc->n_out) return invalid("ws_scan_download: buffer too small");
  WS_HIP(hipMemcpyAsync(xyz_host, sc->out, sc->n_out * 3 * sizeof(int32_t), hipMemcpyDeviceToHost, sc->ctx->stream));
  WS_HIP(hipStreamSynchronize(sc->ctx->stream));
  return WS_OK;
}

// ------------------------------------------------------------------ measurement
int ws_prof_enable(ws_context *ctx, uint32_t class_mask)
{
  if (!ctx) return invalid("ws_prof_enable: ctx is NULL");
  WS_HIP(hipStreamSynchronize(ctx->stream));
  prof_resolve(ctx);
  ctx->prof_mask = class_mask;
  return WS_OK;
}

int ws_prof_read(ws_context *ctx, int cls, double *total_ms, int64_t *launches)
{
  if (!ctx || cls < 0 || cls >= WS_K_COUNT) return invalid("ws_prof_read: bad argument");
  WS_HIP(hipStreamSynchronize(ctx->stream));
  prof_resolve(ctx);
  if (total_ms) *total_ms = ctx->prof_ms[cls];
  if (launches) *launches = ctx->prof_n[cls];
  return WS_OK;
}

int ws_prof_reset(ws_context *ctx)
{
  if (!ctx) return invalid("ws_prof_reset: ctx is NULL");
  WS_HIP(hipStreamSynchronize(ctx->stream));
  prof_resolve(ctx);
  for (int k = 0; k < WS_K_COUNT; ++k)
  {
    ctx->prof_ms[k] = 0;
    ctx->prof_n[k] = 0;
  }
  return WS_OK;
}

} // extern "C"

namespace ws
{
int resize_records(ws_map *m, uint64_t sub_chunks)
{
  WS_HIP(hipStreamSynchronize(m->ctx->stream)); // nothing enqueued may still use the old buffers
  return map_alloc_records(m, sub_chunks);
}
} // namespace ws
