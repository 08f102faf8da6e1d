// Pins the atan2f restated in d-liom_amd/csrc/rotational_histogram.hip (glibc 2.35 sysdeps/ieee754/flt-32/s_atanf.c +
// e_atan2f.c, fdlibm; every operation rounded to float, no FMA contraction) against THIS machine's libm: the
// reference's ComputeHistogram calls std::atan2(float, float), and the device histogram is bit-identical to it only if
// the two agree.  argv[1] = number of samples.  gcc -O2 -ffp-contract=off atan2f_pin.c -lm
#include <math.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
static const float atanhi[] = {4.6364760399e-01f, 7.8539812565e-01f, 9.8279368877e-01f, 1.5707962513e+00f};
static const float atanlo[] = {5.0121582440e-09f, 3.7748947079e-08f, 3.4473217170e-08f, 7.5497894159e-08f};
static const float aT[] = {3.3333334327e-01f, -2.0000000298e-01f, 1.4285714924e-01f, -1.1111110449e-01f, 9.0908870101e-02f,
                           -7.6918758452e-02f, 6.6610731184e-02f, -5.8335702866e-02f, 4.9768779427e-02f, -3.6531571299e-02f,
                           1.6285819933e-02f};
static uint32_t bits(float f) { uint32_t u; memcpy(&u, &f, 4); return u; }
static float my_atanf(float x) {
  float w, s1, s2, z;
  int32_t ix, hx, id;
  hx = (int32_t)bits(x);
  ix = hx & 0x7fffffff;
  if (ix >= 0x4c000000) {
    if (ix > 0x7f800000) return x + x;
    if (hx > 0) return atanhi[3] + atanlo[3];
    return -atanhi[3] - atanlo[3];
  }
  if (ix < 0x3ee00000) {
    if (ix < 0x31000000) return x;
    id = -1;
  } else {
    x = fabsf(x);
    if (ix < 0x3f980000) {
      if (ix < 0x3f300000) { id = 0; x = (2.0f * x - 1.0f) / (2.0f + x); }
      else { id = 1; x = (x - 1.0f) / (x + 1.0f); }
    } else {
      if (ix < 0x401c0000) { id = 2; x = (x - 1.5f) / (1.0f + 1.5f * x); }
      else { id = 3; x = -1.0f / x; }
    }
  }
  z = x * x;
  w = z * z;
  s1 = z * (aT[0] + w * (aT[2] + w * (aT[4] + w * (aT[6] + w * (aT[8] + w * aT[10])))));
  s2 = w * (aT[1] + w * (aT[3] + w * (aT[5] + w * (aT[7] + w * aT[9]))));
  if (id < 0) return x - x * (s1 + s2);
  z = atanhi[id] - ((x * (s1 + s2) - atanlo[id]) - x);
  return (hx < 0) ? -z : z;
}
static float my_atan2f(float y, float x) {
  const float tiny = 1.0e-30f, pi_o_2 = 1.5707963705e+00f, pi = 3.1415927410e+00f, pi_lo = -8.7422776573e-08f;
  float z;
  int32_t k, m, hx, hy, ix, iy;
  hx = (int32_t)bits(x); ix = hx & 0x7fffffff;
  hy = (int32_t)bits(y); iy = hy & 0x7fffffff;
  if (ix > 0x7f800000 || iy > 0x7f800000) return x + y;
  if (hx == 0x3f800000) return my_atanf(y);
  m = ((hy >> 31) & 1) | ((hx >> 30) & 2);
  if (iy == 0) { switch (m) { case 0: case 1: return y; case 2: return pi + tiny; default: return -pi - tiny; } }
  if (ix == 0) return (hy < 0) ? -pi_o_2 - tiny : pi_o_2 + tiny;
  if (ix == 0x7f800000 || iy == 0x7f800000) return atan2f(y, x);  // not needed here
  k = (iy - ix) >> 23;
  if (k > 60) z = pi_o_2 + 0.5f * pi_lo;
  else if (hx < 0 && k < -60) z = 0.0f;
  else z = my_atanf(fabsf(y / x));
  switch (m) {
    case 0: return z;
    case 1: return -z;
    case 2: return pi - (z - pi_lo);
    default: return (z - pi_lo) - pi;
  }
}
int main(int argc, char** argv) {
  srand(7);
  long bad = 0, n = argc > 1 ? atol(argv[1]) : 2000000;
  for (long i = 0; i < n; ++i) {
    float y = (float)((rand() / (double)RAND_MAX - 0.5) * (i % 3 == 0 ? 2.0 : 60.0));
    float x = (float)((rand() / (double)RAND_MAX - 0.5) * (i % 5 == 0 ? 0.5 : 60.0));
    float a = atan2f(y, x), b = my_atan2f(y, x);
    if (bits(a) != bits(b)) { if (bad < 5) printf("diff y=%a x=%a libm=%a mine=%a\n", y, x, a, b); ++bad; }
  }
  printf("mismatches: %ld of %ld\n", bad, n);
  return bad == 0 ? 0 : 1;
}
