// ws_internal.h — shared host/device definitions of libwarpsense_hip (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <string>
#include <vector>

#include "warpsense_hip.h"

namespace ws
{
constexpr int MATRIX_RESOLUTION = 32768; // include/warpsense/consts.h:12-13
constexpr int WEIGHT_RESOLUTION = 64;    // include/warpsense/consts.h:9-10
constexpr int DZ_PER_DISTANCE = 100;     // (int)(tan(45/128 deg)/2 * 32768), update_tsdf.cu:49-50 (checked on the host at load)
constexpr size_t MAX_SCAN_POINTS = 1000000; // update_tsdf.h:33

constexpr int TILE_SHIFT = 6; // dirty-tile granularity: 64 consecutive voxels (256 B of a map)
constexpr uint64_t KEY_INF = ~0ull;
constexpr uint64_t KEY_CONTESTED_TAG = 0xCull << 60; // kpos of a voxel handed to the ordered fallback (never a valid key: t < 2^44)

// ring-buffer parameters passed BY VALUE to kernels (the reference chases three device pointers per
// access, device_map.h:93-101)
struct MapParams
{
  int32_t size[3];
  int32_t pos[3];
  int32_t offset[3];
};

// ---- order key layout -------------------------------------------------------------------
// t    = point(20) | ray step(16) | fan step(8)                       -> 44 bits, unique per candidate
// kpos = t << 16 | value(u16)                                          (min == earliest positive-weight candidate)
// kneg = |value|(15) << 45 | (T_MASK - t) << 1 | (value < 0)           (min == smallest |value|, latest on ties)
constexpr int T_BITS = 44;
constexpr uint64_t T_MASK = (1ull << T_BITS) - 1;

struct ContestedRecord // 16 bytes
{
  uint64_t key; // t << 17 | (weight < 0) << 16 | value(u16)
  uint32_t next;
  uint32_t pad;
};

struct TsdfCounters // device-resident, zeroed at the start of every update
{
  uint32_t contested;  // non-zero: the last resolve pass found contested voxels
  uint32_t records;    // arena records used
  uint32_t error;      // bit0 arena/list overflow, bit1 key range
  uint32_t dirty_tiles;      // length of the touched-tile list (survives until the integrate pass)
  uint32_t last_dirty_tiles; // tiles the last integrate pass streamed
  uint32_t last_contested;   // contested voxels of the last update
  uint32_t pad[2];
};

// device-resident Gauss-Newton state (tsdf_registration.cpp:28-96)
struct GnCore
{
  float T[16];      // total_transform, column-major
  int32_t center[3];
  float alpha;
  float prev[4];
  float it_weight_gradient;
  float epsilon;
  int32_t max_iterations;
  int32_t iterations;
  int32_t finished;
  int32_t error; // 1: the grid barrier of reg_loop_kernel timed out
};
struct GnState
{
  GnCore core;
  int64_t sums[44]; // last h(36) g(6) e c
};

} // namespace ws

struct ws_context
{
  int device = 0;
  hipStream_t own_stream = nullptr;
  hipStream_t stream = nullptr;
  // profiling
  uint32_t prof_mask = 0;
  struct Span
  {
    hipEvent_t a, b;
    int cls;
  };
  std::vector<Span> spans;       // recorded, not yet resolved
  std::vector<hipEvent_t> pool;  // free events
  double prof_ms[WS_K_COUNT] = {};
  int64_t prof_n[WS_K_COUNT] = {};
};

struct ws_map
{
  ws_context *ctx = nullptr;
  ws::MapParams par[2]; // [WS_MAP_AVG], [WS_MAP_NEW]
  int64_t n_vox = 0;
  int64_t n_tiles = 0;
  uint32_t *data[2] = {nullptr, nullptr};
  uint64_t *kpos = nullptr, *kneg = nullptr;
  uint8_t *dirty = nullptr;       // one flag per 64-voxel tile
  uint8_t *vstate = nullptr;      // one byte per voxel for the split scatter (keyed / free space)
  uint32_t *dirty_list = nullptr; // touched tiles of the scan in flight
  void *rays = nullptr;           // per-ray set-up records (48 B x 1 000 000)
  uint32_t *az_hist = nullptr, *az_off = nullptr, *ray_order = nullptr; // rays grouped by azimuth bin
  int32_t tau = 0, max_weight = 0, res = 0;
  bool new_is_default = false; // new_map known to be (tau,0) everywhere
  int integrate_mode = WS_INTEGRATE_SPARSE;
  int32_t *scan_dev = nullptr; // 1 000 000-point upload buffer
  // contested-voxel machinery
  ws::TsdfCounters *counters = nullptr;
  ws::ContestedRecord *arena = nullptr;
  uint32_t arena_cap = 0;
  uint32_t *contested_per_wave = nullptr; // statistics, one slot per wave of the resolve pass
  // LDS-tile scatter (tsdf_tiles.hip)
  int scatter_mode = WS_SCATTER_GLOBAL; // the LDS-tile path is exact but not yet faster
  int64_t n_tiles3d = 0;
  uint32_t *tile_count = nullptr, *tile_offset = nullptr, *tile_cursor = nullptr;
  uint64_t *tile_records = nullptr;
  uint32_t tile_records_cap = 0;
  void *tile_work = nullptr; // uint4 per work item
  uint32_t tile_work_cap = 0;
  void *tile_state = nullptr;
  uint32_t *box_stage = nullptr; // device staging for ws_map_extract_box / ws_map_insert_box
  size_t box_stage_cap = 0;
  ws::TsdfCounters *counters_host = nullptr; // pinned
};

struct ws_reg
{
  ws_context *ctx = nullptr;
  int32_t *points = nullptr;
  size_t cap = 0, n = 0;
  int64_t *partials = nullptr; // [2][32][REG_BLOCKS]
  ws::GnState *state = nullptr;      // [2] device, double buffered by launch parity
  ws::GnState *state_host = nullptr; // pinned staging
  ws::GnState *result_host = nullptr;     // pinned + mapped: the resident loop writes its final state here
  ws::GnState *result_host_dev = nullptr; // device view of result_host
  int32_t *host_flag = nullptr;      // pinned + mapped: the device sets it when the loop has finished
  int32_t *host_flag_dev = nullptr;  // device view of host_flag
  int latest = 0;                    // state buffer holding the newest state
  float *T_dev = nullptr;            // transform for ws_reg_iterate
  int64_t *sums_dev = nullptr;       // 44
  uint32_t *grid_bar = nullptr;      // [2] arrival counter + abort flag of reg_loop_kernel
  int loop_mode = 0;                 // WS_REG_LOOP_*
  int loop_supported = 0;            // the device holds the whole grid of reg_loop_kernel at once
};

// scan pre-processing buffers (App::preprocess on the device, scan_preprocess.hip)
struct ws_scan
{
  ws_context *ctx = nullptr;
  size_t cap = 0;         // points
  size_t table_slots = 0; // power of two >= 2 * cap
  float *in_stage = nullptr;
  size_t in_stage_floats = 0;
  int32_t *tmp = nullptr;
  uint32_t *slot_of = nullptr;
  uint64_t *keys = nullptr;
  uint32_t *first = nullptr;
  uint32_t *wg_count = nullptr;
  uint32_t *wg_off = nullptr;
  uint32_t *counters = nullptr;
  int32_t *out = nullptr;
  uint32_t *host_count = nullptr;     // pinned + mapped
  uint32_t *host_count_dev = nullptr;
  size_t n_out = 0;
};

namespace ws
{
void set_error(const std::string &msg);
int hip_fail(hipError_t e, const char *what, const char *file, int line);

#define WS_HIP(call)                                                     \
  do                                                                     \
  {                                                                      \
    hipError_t e__ = (call);                                             \
    if (e__ != hipSuccess) return ws::hip_fail(e__, #call, __FILE__, __LINE__); \
  } while (0)

// profiling spans around a kernel class
void prof_begin(ws_context *ctx, int cls);
void prof_end(ws_context *ctx, int cls);

// launchers implemented in the .hip files
int launch_tsdf_scatter(ws_map *m, const int32_t *xyz_dev, size_t n, const int32_t scanner_pos[3], const int32_t up[3], bool fused);
int launch_tsdf_integrate(ws_map *m);
int launch_box_copy(ws_map *m, int which, const int32_t lo[3], const int32_t ext[3], uint32_t *box_dev, bool pack);
int fill_u32(ws_context *ctx, uint32_t *dst, uint32_t value, int64_t n);
int fill_u64(ws_context *ctx, uint64_t *dst, uint64_t value, int64_t n);
int check_all_equal_host(const uint32_t *data, int64_t n, uint32_t value);

int launch_reg_accumulate(ws_reg *r, const ws_map *m, const float *T_dev_or_null, int32_t res, uint32_t flags,
                          size_t first, size_t count, int64_t *sums_dev);
int launch_reg_iteration(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, int32_t k);
int launch_reg_solve(ws_reg *r, const int64_t *sums_dev);
int launch_scan_preprocess(ws_scan *sc, const float *xyz_dev, size_t n, size_t stride, const int32_t M[16], int32_t res);
size_t pre_table_slots(size_t max_points);
int launch_reg_loop(ws_reg *r, const ws_map *m, int32_t res, uint32_t flags, const ws::GnCore &init);
int reg_loop_supported(int device);
int launch_solve6_test(ws_context *ctx, const double *A_dev, const double *b_dev, size_t n, double *x_dev, int32_t *status_dev);
size_t reg_barrier_bytes();
} // namespace ws
