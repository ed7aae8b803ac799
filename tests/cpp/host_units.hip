// host_units.hip — host-side checks of the pure functions the kernels share with the host (ws_internal.h): the per-scan split of the
// record's key bits (rec_format / make_rec) and the brick order of the voxel bytes (vbrick).  Built with hipcc, runs without a GPU.
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "ws_internal.h"

using namespace ws;

int main()
{
  // ---- vbrick: a bijection of the 1024 voxels of a tile; four consecutive z of a column (z0 % 4 == 0) stay four consecutive bytes
  std::vector<int> seen(1024, 0);
  for (uint32_t local = 0; local < 1024; ++local)
  {
    const uint32_t b = vbrick(local);
    if (b >= 1024 || seen[b]++)
    {
      printf("vbrick: %u -> %u is not a bijection\n", local, b);
      return 1;
    }
    const uint32_t lx = local >> 8, ly = (local >> 6) & 3, lz = local & 63;
    if (b != (((lz >> 3) << 7) | (lx << 5) | (ly << 3) | (lz & 7)))
    {
      printf("vbrick: layout of %u\n", local);
      return 1;
    }
    if ((lz & 3) == 0)
      for (uint32_t j = 1; j < 4; ++j)
        if (vbrick(local + j) != b + j)
        {
          printf("vbrick: z-run of %u broken\n", local);
          return 1;
        }
  }
  // ---- rec_format: the fields fill 38 bits at most, grow as the scan shrinks, and never fall below round 4's 13 | 5
  const uint64_t sizes[] = {1, 2, 3, 64, 4096, 16384, 16385, 65536, 131072, 131073, 262144, 524288, 1000000};
  int lastS = 99, lastF = 99;
  for (uint64_t n : sizes)
  {
    const RecFormat f = rec_format(n);
    int P = 1;
    while ((1ull << P) < n) ++P;
    if (P + f.S + f.F > T_BITS || f.S > 16 || f.S < 13 || f.F < 5 || f.F > 8 || f.S > lastS || f.F > lastF)
    {
      printf("rec_format(%llu) = S %d F %d (P %d)\n", (unsigned long long)n, f.S, f.F, P);
      return 1;
    }
    lastS = f.S;
    lastF = f.F;
    // make_rec: ascending in (point, step, fan), fields recoverable, value / voxel untouched by the key
    uint64_t prev = 0;
    srand(7);
    for (int it = 0; it < 20000; ++it)
    {
      const uint32_t point = (uint32_t)(rand() % (n < 1000000 ? (int)n : 1000000));
      const int32_t step = rand() % rec_max_steps(f);
      const int32_t iter_steps = 1 + rand() % rec_max_fan(f);
      const int32_t mid = (iter_steps - 1) / 2, j = rand() % iter_steps;
      const int32_t value = (rand() % 65536) - 32768;
      const uint32_t local = (uint32_t)(rand() % 1024);
      const uint64_t r = make_rec(point, step, j - mid, value, local, f.S, f.F);
      const uint64_t t = r >> REC_T_SHIFT;
      const uint64_t want_t = ((uint64_t)point << (f.S + f.F)) | ((uint64_t)step << f.F) | (uint64_t)(j - mid + (int32_t)rec_fan_mid(f.F));
      if (t != want_t || rec_value(r) != value || rec_local(r) != local ||
          rec_negative(r, (1u << f.F) - 1u, rec_fan_mid(f.F)) != (j != mid))
      {
        printf("make_rec: n %llu point %u step %d fan %d\n", (unsigned long long)n, point, step, j - mid);
        return 1;
      }
      (void)prev;
    }
  }
  if (rec_format(131072).S != 15 || rec_format(131072).F != 6 || rec_format(1000000).S != 13 || rec_format(1000000).F != 5 || rec_format(16384).S != 16 ||
      rec_format(16384).F != 8)
  {
    printf("rec_format: documented splits changed\n");
    return 1;
  }
  printf("ok\n");
  return 0;
}
