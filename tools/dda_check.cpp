// dda_check.cpp — the column-change walk of warpsense_amd/csrc/ws_dda.h against the sample-by-sample walk of the reference
// (update_tsdf.cu:65-76) on random rays: same emitting steps, same sample positions.  CPU only.
//   g++ -O2 -std=c++17 -I warpsense_amd/csrc tools/dda_check.cpp -o /tmp/dda_check && /tmp/dda_check [rays]
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "ws_dda.h"

using namespace ws;

static int32_t tdiv(int64_t a, int64_t b) { return (int32_t)(a / b); }

int main(int argc, char **argv)
{
  const long rays = argc > 1 ? atol(argv[1]) : 200000;
  std::mt19937_64 rng(12345);
  auto uni = [&](int64_t lo, int64_t hi) { return (int64_t)(rng() % (uint64_t)(hi - lo + 1)) + lo; };
  const int res_list[] = {2, 3, 7, 20, 50, 51, 64, 100, 1000};
  long long emissions = 0, specials = 0, checked = 0;
  for (long it = 0; it < rays; ++it)
  {
    const int32_t res = res_list[rng() % 9], half = res / 2;
    // ray origin within a few hundred voxels of zero (so that rays cross the cell around zero), direction up to 40 m
    int32_t pos[3], d[3];
    const int64_t span = (it % 3 == 0) ? 3 * res : 400LL * res;
    for (int c = 0; c < 3; ++c) pos[c] = (int32_t)uni(-span, span);
    const int64_t reach = (it % 5 == 0) ? 200 : 30000;
    for (int c = 0; c < 3; ++c) d[c] = (int32_t)uni(-reach, reach);
    if (it % 7 == 0) d[rng() % 3] = 0;
    if (it % 11 == 0) d[0] = d[1] = 0;
    const int64_t sq = (int64_t)d[0] * d[0] + (int64_t)d[1] * d[1] + (int64_t)d[2] * d[2];
    const int32_t dist = (int32_t)sqrtf((float)sq);
    if (dist < 2) continue;
    int32_t dmax = 0;
    for (int c = 0; c < 3; ++c) dmax = std::max(dmax, std::abs(d[c]));
    if (dmax > dist) continue; // (float rounding of the norm: such rays are not RAY_FAST)
    const int32_t tau = 600;
    const int64_t len_end = (int64_t)dist + tau;
    if (dmax * len_end >= (1LL << 31)) continue;
    const int32_t steps = (int32_t)((len_end - 1) / half) + 1;
    // reference walk
    std::vector<int32_t> ref_k;
    std::vector<int32_t> ref_p;
    int32_t prevx = 0, prevy = 0;
    for (int32_t k = 0; k < steps; ++k)
    {
      const int64_t len = 1 + (int64_t)k * half;
      int32_t proj[3];
      for (int c = 0; c < 3; ++c) proj[c] = pos[c] + tdiv((int64_t)d[c] * len, dist);
      const int32_t ix = tdiv(proj[0], res), iy = tdiv(proj[1], res);
      if (ix == prevx && iy == prevy) continue;
      prevx = ix;
      prevy = iy;
      ref_k.push_back(k);
      ref_p.push_back(proj[0]);
      ref_p.push_back(proj[1]);
      ref_p.push_back(proj[2]);
    }
    // multiply-shift constants of the ray set-up (make_fastdiv)
    int l = 0;
    while ((1LL << l) < dist) ++l;
    const int kk = 31 + l;
    const uint64_t p2 = 1ULL << kk;
    const uint64_t M = p2 / (uint64_t)dist + ((p2 % (uint64_t)dist) ? 1 : 0);
    if (M >> 32) { printf("M does not fit 32 bits: dist %d\n", dist); return 1; }
    DdaRay R;
    R.M32 = (uint32_t)M;
    R.sh = kk - 32;
    // the walk in `lanes` chunks
    const int lanes = 1 + (int)(rng() % 5);
    const int32_t kend = (int32_t)uni(0, steps);
    const int32_t ch = (kend + lanes - 1) / lanes;
    std::vector<int32_t> got_k, got_p;
    for (int c = 0; c < lanes; ++c)
    {
      const int32_t k0 = c * ch, k1 = std::min(k0 + ch, kend);
      if (k0 >= k1) continue;
      uint32_t ad[3];
      int32_t spos[3], sm[3];
      for (int a = 0; a < 3; ++a)
      {
        ad[a] = (uint32_t)std::abs(d[a]);
        sm[a] = d[a] < 0 ? -1 : 0;
        spos[a] = d[a] < 0 ? -pos[a] : pos[a];
      }
      const int32_t kinit = k0 > 0 ? k0 - 1 : 0;
      const int32_t len0 = 1 + kinit * half;
      DdaAxis wx, wy;
      dda_axis_init(wx, ad[0], spos[0], dda_q(ad[0], len0, R), dist, res, half);
      dda_axis_init(wy, ad[1], spos[1], dda_q(ad[1], len0, R), dist, res, half);
      auto emit = [&](int32_t k) {
        const int32_t len = 1 + k * half;
        got_k.push_back(k);
        for (int a = 0; a < 3; ++a)
        {
          const int32_t av = spos[a] + (int32_t)dda_q(ad[a], len, R);
          got_p.push_back((av ^ sm[a]) - sm[a]);
        }
      };
      if (k0 == 0)
      {
        // the sample k == 0 is compared with the column (0, 0) (update_tsdf.cu:65,71)
        const int32_t ax = spos[0] + (int32_t)dda_q(ad[0], 1, R), ay = spos[1] + (int32_t)dda_q(ad[1], 1, R);
        if (ax / res != 0 || ay / res != 0) emit(0);
      }
      for (;;)
      {
        const uint32_t k = std::min(wx.K, wy.K);
        if (k >= (uint32_t)k1) break;
        const bool cx = wx.K == k, cy = wy.K == k;
        if (cx)
        {
          const bool sp = wx.Ksp == k;
          dda_axis_advance(wx);
          if (sp) { dda_axis_after_zero_cell(wx, ad[0], spos[0], dist, res); ++specials; }
        }
        if (cy)
        {
          const bool sp = wy.Ksp == k;
          dda_axis_advance(wy);
          if (sp) { dda_axis_after_zero_cell(wy, ad[1], spos[1], dist, res); ++specials; }
        }
        emit((int32_t)k);
      }
    }
    // compare with the reference's emissions below kend
    size_t nref = 0;
    while (nref < ref_k.size() && ref_k[nref] < kend) ++nref;
    bool ok = got_k.size() == nref;
    for (size_t i = 0; ok && i < nref; ++i)
      ok = got_k[i] == ref_k[i] && got_p[3 * i] == ref_p[3 * i] && got_p[3 * i + 1] == ref_p[3 * i + 1] && got_p[3 * i + 2] == ref_p[3 * i + 2];
    if (!ok)
    {
      printf("MISMATCH ray %ld: res %d pos (%d %d %d) d (%d %d %d) dist %d kend %d lanes %d: %zu emissions, reference %zu\n", it, res, pos[0], pos[1], pos[2],
             d[0], d[1], d[2], dist, kend, lanes, got_k.size(), nref);
      for (size_t i = 0; i < std::min<size_t>(std::max(got_k.size(), nref), 12); ++i)
        printf("  %zu: got %d ref %d\n", i, i < got_k.size() ? got_k[i] : -1, i < nref ? ref_k[i] : -1);
      return 1;
    }
    emissions += (long long)nref;
    ++checked;
  }
  printf("ok: %lld rays, %lld emissions, %lld crossings of the cell around zero\n", checked, emissions, specials);
  return 0;
}
