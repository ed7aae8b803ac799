// ws_tiles.h — host/device interface of the LDS-staged scatter (tsdf_tiles.hip).
#pragma once

#include "ws_march.h"

namespace ws
{
constexpr int TILE_GRID_BLOCKS = 2048; // persistent grid of tile_scatter_kernel

struct TileGrid
{
  int32_t ntx, nty, ntz; // tiles per axis (8 x 8 x 16 voxels each, ring-index space)
};
inline TileGrid make_tile_grid(const MapParams &m)
{
  TileGrid g;
  g.ntx = (m.size[0] + 7) >> 3;
  g.nty = (m.size[1] + 7) >> 3;
  g.ntz = (m.size[2] + 15) >> 4;
  return g;
}

struct TileState // device resident
{
  uint32_t work_count;
  uint32_t total_records;
  uint32_t contested;
  uint32_t pad;
};

struct TileArgs
{
  const RaySetup *rays;
  uint32_t n;
  MarchFrame frame;
  TileGrid grid;
  int64_t n_tiles;
  uint32_t *tile_count;
  uint32_t *tile_offset;
  uint32_t *tile_cursor;
  uint64_t *records;
  uint32_t records_cap;
  uint4 *work;
  uint32_t work_cap;
  TileState *tile_state;
  // global structures shared with the global path
  uint64_t *kpos;
  uint64_t *kneg;
  uint8_t *dirty;
  uint32_t *new_data;
  uint32_t *avg_data;
  int32_t max_weight;
  TsdfCounters *counters;
};

int launch_tile_path(ws_map *m, const TileArgs &a, size_t n, bool fused);
} // namespace ws
