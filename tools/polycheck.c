/* Validation of the sine / cosine polynomials of gn_step (registration.hip) against the C library: what the Gauss-Newton update
 * uses are (float)sin(theta) and (float)(1 - cos(theta)); both polynomial forms (separate multiply/add, fused multiply-add) must give
 * the same two floats as libm for every angle below 0.25 rad.   gcc -O2 -ffp-contract=off tools/polycheck.c -lm && ./a.out */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <stdint.h>
static void poly_sep(double theta, double *s, double *c) {
  const double z = theta * theta;
  const double p = -1.0 / 6 + z * (1.0 / 120 + z * (-1.0 / 5040 + z * (1.0 / 362880 + z * (-1.0 / 39916800 + z * (1.0 / 6227020800.0)))));
  *s = theta + theta * z * p;
  const double q = 1.0 / 24 + z * (-1.0 / 720 + z * (1.0 / 40320 + z * (-1.0 / 3628800 + z * (1.0 / 479001600.0 + z * (-1.0 / 87178291200.0)))));
  const double t = 0.5 * z, u = 1.0 - t, e = (1.0 - u) - t, ww = z * z * q;
  *c = u + (e + ww);
}
static void poly_fma(double theta, double *s, double *c) {
  const double z = theta * theta;
  double p = fma(z, 1.0 / 6227020800.0, -1.0 / 39916800);
  p = fma(z, p, 1.0 / 362880); p = fma(z, p, -1.0 / 5040); p = fma(z, p, 1.0 / 120); p = fma(z, p, -1.0 / 6);
  *s = fma(theta * z, p, theta);
  double q = fma(z, -1.0 / 87178291200.0, 1.0 / 479001600.0);
  q = fma(z, q, -1.0 / 3628800); q = fma(z, q, 1.0 / 40320); q = fma(z, q, -1.0 / 720); q = fma(z, q, 1.0 / 24);
  const double t = 0.5 * z, u = 1.0 - t, e = (1.0 - u) - t, ww = z * z * q;
  *c = u + (e + ww);
}
int main() {
  uint64_t st = 88172645463325252ull; long bad_sep_s = 0, bad_sep_c = 0, bad_fma_s = 0, bad_fma_c = 0; long n = 20000000;
  for (long i = 0; i < n; ++i) {
    st ^= st << 13; st ^= st >> 7; st ^= st << 17;
    double u = (st >> 11) * (1.0 / 9007199254740992.0);
    double theta = (i & 1) ? u * 0.25 : exp(log(1e-9) + u * (log(0.25) - log(1e-9)));
    if (theta < 1e-9) theta = 1e-9;
    double s1, c1, s2, c2; poly_sep(theta, &s1, &c1); poly_fma(theta, &s2, &c2);
    float rs = (float)sin(theta), rc = (float)(1 - cos(theta));
    bad_sep_s += (float)s1 != rs; bad_sep_c += (float)(1 - c1) != rc;
    bad_fma_s += (float)s2 != rs; bad_fma_c += (float)(1 - c2) != rc;
  }
  printf("n %ld: separate mul/add: sin %ld cos %ld mismatches; fma: sin %ld cos %ld\n", n, bad_sep_s, bad_sep_c, bad_fma_s, bad_fma_c);
  return 0;
}
