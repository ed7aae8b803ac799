// replay.cpp — warpsense::App (include/warpsense_hip/app.hpp) over a recorded stream of sensor clouds: the C++
// counterpart of tools/replay_stream.py / warpsense_amd.App.  No ROS, no PCL.
//
//   replay <clouds.bin> <scans> <points_per_scan> <map_edge_voxels> <res> <tau> <max_weight> <shift_m> <poses_out.bin> <map_out.bin> [map.h5]
//
// clouds.bin: scans x points x 3 float32 (metres, sensor frame).  Prints one line per scan, writes the pose after
// every scan (16 floats, column-major) and the final local-map window (uint32 per voxel, ring-buffer order) after
// App::terminate(); with a last argument the global map goes to that .h5 file (needs WARPSENSE_HIP_WITH_H5).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "warpsense_hip/app.hpp"

int main(int argc, char **argv)
{
  if (argc != 11 && argc != 12)
  {
    fprintf(stderr, "usage: %s clouds.bin scans points edge res tau max_weight shift_m poses_out.bin map_out.bin [map.h5]\n", argv[0]);
    return 2;
  }
  const size_t scans = strtoull(argv[2], nullptr, 10), n = strtoull(argv[3], nullptr, 10);
  warpsense::AppParams p;
  const int edge = atoi(argv[4]);
  p.map_size[0] = p.map_size[1] = edge;
  p.map_size[2] = edge / 2;
  p.hot.map_resolution = atoi(argv[5]);
  p.hot.tau = atoi(argv[6]);
  p.hot.max_weight = atoi(argv[7]);
  p.max_distance = (float)p.hot.tau / 1000.f;
  p.shift = (float)atof(argv[8]);
  p.async_shift = getenv("WS_REPLAY_ASYNC_SHIFT") != nullptr; // the map shift off the scan path (MappingNode::shift_map_async)

  std::vector<float> clouds(scans * n * 3);
  {
    std::ifstream f(argv[1], std::ios::binary);
    f.read(reinterpret_cast<char *>(clouds.data()), (std::streamsize)(clouds.size() * sizeof(float)));
    if (!f)
    {
      fprintf(stderr, "cannot read %zu floats from %s\n", clouds.size(), argv[1]);
      return 2;
    }
  }
  warpsense::App app(p, argc == 12 ? argv[11] : "", n);
  std::ofstream poses(argv[9], std::ios::binary);
  for (size_t k = 0; k < scans; ++k)
  {
    const rmagine::Matrix4x4f &pose = app.cloud_callback(&clouds[k * n * 3], n, 3);
    poses.write(reinterpret_cast<const char *>(&pose.data[0][0]), 16 * sizeof(float));
    printf("scan %zu points %zu iterations %d updates %d shifts %d\n", k, app.last_points(), app.last_iterations(), app.n_updates(), app.n_shifts());
  }
  app.terminate();
  app.node().download();
  cuda::pause();
  auto &lm = app.local_map();
  printf("window pos %d %d %d offset %d %d %d size %d %d %d\n", lm.get_pos().x, lm.get_pos().y, lm.get_pos().z, lm.get_offset().x, lm.get_offset().y,
         lm.get_offset().z, lm.get_size().x, lm.get_size().y, lm.get_size().z);
  std::ofstream out(argv[10], std::ios::binary);
  out.write(reinterpret_cast<const char *>(lm.data().data()), (std::streamsize)(lm.data().size() * sizeof(TSDFEntry)));
  return 0;
}
