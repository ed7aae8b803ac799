// harness.cpp — the call sequence of the reference's test/pcd_registration.cpp:234-356 (build the map with
// update_tsdf, register a perturbed copy of the cloud) written against the drop-in C++ classes of
// include/warpsense_hip/{compat,mapping}.hpp.  No ROS, no PCL: clouds come in as raw int32 xyz files.
//
//   harness <scan.bin> <perturbed.bin> <n> <map_edge_voxels> <res> <tau> <max_weight> <avg_out.bin>
//
// prints "iterations <k>" and the 16 floats of the column-major result transform; writes the averaged
// map (uint32 per voxel) so a test can compare it with the oracle.
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <vector>

#include "warpsense_hip/mapping.hpp"

namespace rm = rmagine;

static std::vector<rm::Pointi> read_points(const char *path, size_t n)
{
  std::vector<rm::Pointi> pts(n);
  std::ifstream f(path, std::ios::binary);
  f.read(reinterpret_cast<char *>(pts.data()), (std::streamsize)(n * sizeof(rm::Pointi)));
  if (!f)
  {
    fprintf(stderr, "cannot read %zu points from %s\n", n, path);
    exit(2);
  }
  return pts;
}

int main(int argc, char **argv)
{
  if (argc != 9)
  {
    fprintf(stderr, "usage: %s scan.bin perturbed.bin n edge res tau max_weight avg_out.bin\n", argv[0]);
    return 2;
  }
  const size_t n = strtoull(argv[3], nullptr, 10);
  int edge = atoi(argv[4]);
  cuda::HotPathParams params;
  params.map_resolution = atoi(argv[5]);
  params.tau = atoi(argv[6]);
  params.max_weight = atoi(argv[7]);

  // HDF5LocalMap's in-memory state (src/map/hdf5_local_map.cpp:5-20): odd sizes, offset = size/2, default (tau, 0)
  if (edge % 2 == 0) edge += 1;
  rm::Pointi size(edge, edge, edge), pos(0, 0, 0), offset(edge / 2, edge / 2, edge / 2);
  std::vector<TSDFEntry> data((size_t)edge * edge * edge, TSDFEntry((int16_t)params.tau, 0));
  cuda::DeviceMap local_map(&size, &offset, data.data(), &pos);

  auto scan = read_points(argv[1], n);
  auto perturbed = read_points(argv[2], n);

  cuda::TSDFRegistration gpu(params, local_map);
  rm::Matrix4x4f pose;
  pose.setIdentity();
  gpu.update_tsdf(local_map, scan, pose); // result overload: averaged map comes back into `data`
  rm::Matrix4x4f T = gpu.register_cloud(perturbed, pose);
  cuda::pause();

  printf("iterations %d\n", gpu.last_iterations());
  for (int j = 0; j < 4; ++j)
    for (int i = 0; i < 4; ++i) printf("%.9g ", T.data[j][i]);
  printf("\n");
  std::ofstream out(argv[8], std::ios::binary);
  out.write(reinterpret_cast<const char *>(data.data()), (std::streamsize)(data.size() * sizeof(TSDFEntry)));
  return 0;
}
