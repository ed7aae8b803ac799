// ws_h5.cpp — the global-map file behind include/warpsense_h5.h, written with the HDF5 C API.
// Same objects, names, types and shapes as HDF5GlobalMap (src/map/hdf5_global_map.cpp) produces through HighFive:
// 1-D uint32 chunk datasets under /map, scalar int32/float32 attributes on /map, 1-D float32[7] pose datasets.
#include "warpsense_h5.h"

#include <hdf5.h>

#include <cstdio>
#include <cstring>
#include <new>
#include <string>
#include <vector>

struct ws_h5
{
  hid_t file = -1;
};

namespace
{
thread_local std::string g_err;

int fail(const std::string &msg)
{
  g_err = msg;
  return -1;
}

std::string chunk_tag(int32_t cx, int32_t cy, int32_t cz)
{
  // tag_from_chunk_pos, hdf5_global_map.cpp:46-51
  return std::to_string(cx) + "_" + std::to_string(cy) + "_" + std::to_string(cz);
}

bool exists(hid_t loc, const char *name) { return H5Lexists(loc, name, H5P_DEFAULT) > 0; }

int ensure_group(hid_t file, const char *name)
{
  if (exists(file, name)) return 0;
  hid_t g = H5Gcreate2(file, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  if (g < 0) return fail(std::string("cannot create group ") + name);
  H5Gclose(g);
  return 0;
}

int write_attr(hid_t obj, const char *name, hid_t type, const void *value)
{
  hid_t a = -1;
  if (H5Aexists(obj, name) > 0)
    a = H5Aopen(obj, name, H5P_DEFAULT);
  else
  {
    hid_t sp = H5Screate(H5S_SCALAR);
    a = H5Acreate2(obj, name, type, sp, H5P_DEFAULT, H5P_DEFAULT);
    H5Sclose(sp);
  }
  if (a < 0) return fail(std::string("cannot create attribute ") + name);
  const herr_t e = H5Awrite(a, type, value);
  H5Aclose(a);
  return e < 0 ? fail(std::string("cannot write attribute ") + name) : 0;
}

int read_attr(hid_t obj, const char *name, hid_t type, void *value)
{
  hid_t a = H5Aopen(obj, name, H5P_DEFAULT);
  if (a < 0) return fail(std::string("missing attribute ") + name);
  const herr_t e = H5Aread(a, type, value);
  H5Aclose(a);
  return e < 0 ? fail(std::string("cannot read attribute ") + name) : 0;
}

// create-or-overwrite a 1-D dataset (HighFive createDataSet(name, std::vector<T>) / DataSet::write)
int write_1d(hid_t loc, const char *name, hid_t file_type, hid_t mem_type, hsize_t n, const void *data)
{
  hid_t d = -1;
  if (exists(loc, name))
    d = H5Dopen2(loc, name, H5P_DEFAULT);
  else
  {
    hid_t sp = H5Screate_simple(1, &n, nullptr);
    d = H5Dcreate2(loc, name, file_type, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    H5Sclose(sp);
  }
  if (d < 0) return fail(std::string("cannot create dataset ") + name);
  const herr_t e = H5Dwrite(d, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
  H5Dclose(d);
  return e < 0 ? fail(std::string("cannot write dataset ") + name) : 0;
}

int read_1d(hid_t loc, const char *name, hid_t mem_type, hsize_t n, void *data)
{
  hid_t d = H5Dopen2(loc, name, H5P_DEFAULT);
  if (d < 0) return fail(std::string("cannot open dataset ") + name);
  hid_t sp = H5Dget_space(d);
  const hssize_t have = H5Sget_simple_extent_npoints(sp);
  H5Sclose(sp);
  if (have != (hssize_t)n)
  {
    H5Dclose(d);
    return fail(std::string("dataset ") + name + " has an unexpected size");
  }
  const herr_t e = H5Dread(d, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, data);
  H5Dclose(d);
  return e < 0 ? fail(std::string("cannot read dataset ") + name) : 0;
}

struct ListCtx
{
  int32_t *out;
  int64_t cap, n;
};

herr_t list_cb(hid_t, const char *name, const H5L_info_t *, void *op)
{
  ListCtx *c = static_cast<ListCtx *>(op);
  int x, y, z;
  if (std::sscanf(name, "%d_%d_%d", &x, &y, &z) == 3)
  {
    if (c->out && c->n < c->cap)
    {
      c->out[3 * c->n + 0] = x;
      c->out[3 * c->n + 1] = y;
      c->out[3 * c->n + 2] = z;
    }
    c->n += 1;
  }
  return 0;
}

int group_count(hid_t file, const char *name, int64_t *n)
{
  hid_t g = H5Gopen2(file, name, H5P_DEFAULT);
  if (g < 0) return fail(std::string("missing group ") + name);
  H5G_info_t info;
  const herr_t e = H5Gget_info(g, &info);
  H5Gclose(g);
  if (e < 0) return fail(std::string("cannot stat group ") + name);
  *n = (int64_t)info.nlinks;
  return 0;
}
} // namespace

extern "C" {

const char *ws_h5_last_error(void) { return g_err.c_str(); }

int ws_h5_create(const char *path, ws_h5 **out)
{
  if (!path || !out) return fail("ws_h5_create: NULL argument");
  H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr); // errors are reported through the return value
  ws_h5 *f = new (std::nothrow) ws_h5();
  if (!f) return fail("ws_h5_create: out of memory");
  f->file = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
  if (f->file < 0)
  {
    delete f;
    return fail(std::string("cannot create ") + path);
  }
  if (ensure_group(f->file, "/map") != 0 || ensure_group(f->file, "/poses") != 0)
  {
    H5Fclose(f->file);
    delete f;
    return -1;
  }
  *out = f;
  return 0;
}

int ws_h5_open(const char *path, int writable, ws_h5 **out)
{
  if (!path || !out) return fail("ws_h5_open: NULL argument");
  H5Eset_auto2(H5E_DEFAULT, nullptr, nullptr);
  ws_h5 *f = new (std::nothrow) ws_h5();
  if (!f) return fail("ws_h5_open: out of memory");
  f->file = H5Fopen(path, writable ? H5F_ACC_RDWR : H5F_ACC_RDONLY, H5P_DEFAULT);
  if (f->file < 0)
  {
    delete f;
    return fail(std::string("cannot open ") + path);
  }
  *out = f;
  return 0;
}

int ws_h5_flush(ws_h5 *f)
{
  if (!f) return fail("ws_h5_flush: NULL");
  return H5Fflush(f->file, H5F_SCOPE_GLOBAL) < 0 ? fail("flush failed") : 0;
}

int ws_h5_close(ws_h5 *f)
{
  if (!f) return 0;
  int rc = 0;
  if (f->file >= 0)
  {
    H5Fflush(f->file, H5F_SCOPE_GLOBAL);
    if (H5Fclose(f->file) < 0) rc = fail("close failed");
  }
  delete f;
  return rc;
}

int ws_h5_write_meta(ws_h5 *f, int32_t tau, const int32_t map_size[3], float max_distance, int32_t map_resolution, int32_t max_weight)
{
  if (!f || !map_size) return fail("ws_h5_write_meta: NULL argument");
  hid_t g = H5Gopen2(f->file, "/map", H5P_DEFAULT);
  if (g < 0) return fail("missing group /map");
  int rc = 0;
  rc |= write_attr(g, "tau", H5T_NATIVE_INT32, &tau);
  rc |= write_attr(g, "map_size_x", H5T_NATIVE_INT32, &map_size[0]);
  rc |= write_attr(g, "map_size_y", H5T_NATIVE_INT32, &map_size[1]);
  rc |= write_attr(g, "map_size_z", H5T_NATIVE_INT32, &map_size[2]);
  rc |= write_attr(g, "max_distance", H5T_NATIVE_FLOAT, &max_distance);
  rc |= write_attr(g, "map_resolution", H5T_NATIVE_INT32, &map_resolution);
  rc |= write_attr(g, "max_weight", H5T_NATIVE_INT32, &max_weight);
  H5Gclose(g);
  if (rc != 0) return -1;
  return ws_h5_flush(f);
}

int ws_h5_read_meta(ws_h5 *f, int32_t *tau, int32_t map_size[3], float *max_distance, int32_t *map_resolution, int32_t *max_weight)
{
  if (!f || !tau || !map_size || !max_distance || !map_resolution || !max_weight) return fail("ws_h5_read_meta: NULL argument");
  hid_t g = H5Gopen2(f->file, "/map", H5P_DEFAULT);
  if (g < 0) return fail("missing group /map");
  int rc = 0;
  rc |= read_attr(g, "tau", H5T_NATIVE_INT32, tau);
  rc |= read_attr(g, "map_size_x", H5T_NATIVE_INT32, &map_size[0]);
  rc |= read_attr(g, "map_size_y", H5T_NATIVE_INT32, &map_size[1]);
  rc |= read_attr(g, "map_size_z", H5T_NATIVE_INT32, &map_size[2]);
  rc |= read_attr(g, "max_distance", H5T_NATIVE_FLOAT, max_distance);
  rc |= read_attr(g, "map_resolution", H5T_NATIVE_INT32, map_resolution);
  rc |= read_attr(g, "max_weight", H5T_NATIVE_INT32, max_weight);
  H5Gclose(g);
  return rc != 0 ? -1 : 0;
}

int ws_h5_write_chunk(ws_h5 *f, int32_t cx, int32_t cy, int32_t cz, const uint32_t *data)
{
  if (!f || !data) return fail("ws_h5_write_chunk: NULL argument");
  hid_t g = H5Gopen2(f->file, "/map", H5P_DEFAULT);
  if (g < 0) return fail("missing group /map");
  const int rc = write_1d(g, chunk_tag(cx, cy, cz).c_str(), H5T_STD_U32LE, H5T_NATIVE_UINT32, WS_H5_CHUNK_VOXELS, data);
  H5Gclose(g);
  return rc;
}

int ws_h5_read_chunk(ws_h5 *f, int32_t cx, int32_t cy, int32_t cz, uint32_t *data, int32_t *exists_out)
{
  if (!f || !data || !exists_out) return fail("ws_h5_read_chunk: NULL argument");
  hid_t g = H5Gopen2(f->file, "/map", H5P_DEFAULT);
  if (g < 0) return fail("missing group /map");
  const std::string tag = chunk_tag(cx, cy, cz);
  int rc = 0;
  if (!exists(g, tag.c_str()))
    *exists_out = 0;
  else
  {
    *exists_out = 1;
    rc = read_1d(g, tag.c_str(), H5T_NATIVE_UINT32, WS_H5_CHUNK_VOXELS, data);
  }
  H5Gclose(g);
  return rc;
}

int ws_h5_num_chunks(ws_h5 *f, int64_t *n)
{
  if (!f || !n) return fail("ws_h5_num_chunks: NULL argument");
  return ws_h5_list_chunks(f, nullptr, 0, n);
}

int ws_h5_list_chunks(ws_h5 *f, int32_t *chunk_pos, int64_t capacity, int64_t *n)
{
  if (!f || !n) return fail("ws_h5_list_chunks: NULL argument");
  hid_t g = H5Gopen2(f->file, "/map", H5P_DEFAULT);
  if (g < 0) return fail("missing group /map");
  ListCtx c{chunk_pos, capacity, 0};
  const herr_t e = H5Literate(g, H5_INDEX_NAME, H5_ITER_NATIVE, nullptr, list_cb, &c);
  H5Gclose(g);
  if (e < 0) return fail("cannot list /map");
  *n = c.n;
  return 0;
}

int ws_h5_write_pose(ws_h5 *f, const float values[7])
{
  if (!f || !values) return fail("ws_h5_write_pose: NULL argument");
  int64_t count = 0;
  if (group_count(f->file, "/poses", &count) != 0) return -1;
  const std::string name = "/poses/" + std::to_string(count); // identifier = number of poses so far (:182-185)
  hid_t g = H5Gcreate2(f->file, name.c_str(), H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
  if (g < 0) return fail("cannot create " + name);
  const int rc = write_1d(g, "pose", H5T_IEEE_F32LE, H5T_NATIVE_FLOAT, 7, values);
  H5Gclose(g);
  if (rc != 0) return rc;
  return ws_h5_flush(f);
}

int ws_h5_num_poses(ws_h5 *f, int64_t *n)
{
  if (!f || !n) return fail("ws_h5_num_poses: NULL argument");
  return group_count(f->file, "/poses", n);
}

int ws_h5_read_pose(ws_h5 *f, int64_t index, float values[7])
{
  if (!f || !values) return fail("ws_h5_read_pose: NULL argument");
  const std::string name = "/poses/" + std::to_string(index);
  hid_t g = H5Gopen2(f->file, name.c_str(), H5P_DEFAULT);
  if (g < 0) return fail("missing " + name);
  const int rc = read_1d(g, "pose", H5T_NATIVE_FLOAT, 7, values);
  H5Gclose(g);
  return rc;
}

} // extern "C"
