// Times dliom_rotational_histogram_mt (host code) on a 46 557-point cloud; argv[1] = host threads (0 = automatic).  g++ -O2 tools/hist_bench.cc -I include -L d-liom_amd -ldliom -Wl,-rpath,$PWD/d-liom_amd
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include "dliom.h"
int main(int argc, char** argv) {
  int n = 46557; std::vector<float> p(3 * n);
  srand(1);
  for (int i = 0; i < n; ++i) { float a = 6.28318f * rand() / RAND_MAX, r = 5 + 20.f * rand() / RAND_MAX; p[3*i] = r * cosf(a); p[3*i+1] = r * sinf(a); p[3*i+2] = -4 + 8.f * rand() / RAND_MAX; }
  std::vector<float> h(120);
  const int threads = argc > 1 ? atoi(argv[1]) : 0;
  for (int rep = 0; rep < 3; ++rep) {
    auto t0 = std::chrono::steady_clock::now();
    for (int k = 0; k < 10; ++k) dliom_rotational_histogram_mt(p.data(), n, 120, threads, h.data());
    printf("%.3f ms\n", std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() * 100);
  }
}
