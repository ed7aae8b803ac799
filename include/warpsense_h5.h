/* warpsense_h5.h — C ABI of libwarpsense_h5.so: the global map file of the reference.
 *
 * File layout, identical to what HDF5GlobalMap writes through HighFive (src/map/hdf5_global_map.cpp):
 *   /map                      group; attributes tau, map_size_x/y/z, map_resolution, max_weight (int32)
 *                             and max_distance (float32)                                    — :207-221 write_meta
 *   /map/<cx>_<cy>_<cz>       one dataset per 64^3-voxel chunk: 262144 x uint32 raw TSDF entries,
 *                             index x*4096 + y*64 + z inside the chunk                      — :46-57, :112-121, :160-176
 *   /poses/<i>/pose           7 x float32: x y z (scaled, rounded to 3 decimals) qx qy qz qw  — :178-205 write_pose
 *
 * Optional component: built only where the HDF5 C library is installed (warpsense_amd/build.py probes for it);
 * the HIP hot path (libwarpsense_hip.so) does not depend on it.  No torch / HIP types in the signatures.
 * Every function returns 0 on success or a negative status; ws_h5_last_error() has the message.
 */
#ifndef WARPSENSE_H5_H
#define WARPSENSE_H5_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ws_h5 ws_h5;

#define WS_H5_CHUNK_SIZE 64
#define WS_H5_CHUNK_VOXELS (64 * 64 * 64)

const char *ws_h5_last_error(void);
/* HDF5GlobalMap::HDF5GlobalMap — hdf5_global_map.cpp:4-39: OpenOrCreate | Truncate, creates /map and /poses */
int ws_h5_create(const char *path, ws_h5 **out);
/* open an existing file (read-write if `writable`): test/map.cpp:95-96 reads one back this way */
int ws_h5_open(const char *path, int writable, ws_h5 **out);
int ws_h5_close(ws_h5 *f); /* flushes */
int ws_h5_flush(ws_h5 *f);
/* HDF5GlobalMap::write_meta — hdf5_global_map.cpp:207-221 */
int ws_h5_write_meta(ws_h5 *f, int32_t tau, const int32_t map_size[3], float max_distance, int32_t map_resolution,
                     int32_t max_weight);
int ws_h5_read_meta(ws_h5 *f, int32_t *tau, int32_t map_size[3], float *max_distance, int32_t *map_resolution,
                    int32_t *max_weight);
/* chunk datasets — activate_chunk / write_back, hdf5_global_map.cpp:59-121,160-176 */
int ws_h5_write_chunk(ws_h5 *f, int32_t cx, int32_t cy, int32_t cz, const uint32_t *data /* 262144 */);
int ws_h5_read_chunk(ws_h5 *f, int32_t cx, int32_t cy, int32_t cz, uint32_t *data /* 262144 */, int32_t *exists);
int ws_h5_num_chunks(ws_h5 *f, int64_t *n);
int ws_h5_list_chunks(ws_h5 *f, int32_t *chunk_pos /* n x 3 */, int64_t capacity, int64_t *n);
/* HDF5GlobalMap::write_pose — hdf5_global_map.cpp:178-199: appends /poses/<count>/pose (values already rounded) */
int ws_h5_write_pose(ws_h5 *f, const float values[7]);
int ws_h5_num_poses(ws_h5 *f, int64_t *n);
int ws_h5_read_pose(ws_h5 *f, int64_t index, float values[7]);

#ifdef __cplusplus
}
#endif
#endif
