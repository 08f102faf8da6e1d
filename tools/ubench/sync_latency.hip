// How long does the host wait for a tiny kernel?  hipStreamSynchronize vs spinning on a word the kernel writes to pinned
// host memory (after a system-scope fence).  hipcc --offload-arch=gfx950 -O2 sync_latency.hip -o sync_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
__global__ void tiny(volatile unsigned* flag, unsigned v, unsigned* sink) {
  if (threadIdx.x == 0) {
    sink[0] = v;
    __threadfence_system();
    *flag = v;
  }
}
int main() {
  hipStream_t s;
  hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
  unsigned* flag;
  hipHostMalloc((void**)&flag, 64, hipHostMallocDefault);
  unsigned* sink;
  hipMalloc((void**)&sink, 64);
  *flag = 0;
  const int reps = 2000;
  for (int mode = 0; mode < 3; ++mode) {
    double total = 0;
    for (int i = 1; i <= reps; ++i) {
      auto t0 = std::chrono::steady_clock::now();
      hipLaunchKernelGGL(tiny, dim3(1), dim3(64), 0, s, flag, (unsigned)(mode * reps + i), sink);
      if (mode == 0) hipStreamSynchronize(s);
      else if (mode == 1) { while (*(volatile unsigned*)flag != (unsigned)(mode * reps + i)) {} }
      else { unsigned h; hipMemcpyAsync(&h, sink, 4, hipMemcpyDeviceToHost, s); hipStreamSynchronize(s); }
      auto t1 = std::chrono::steady_clock::now();
      if (i > 100) total += std::chrono::duration<double>(t1 - t0).count();
      if (mode == 1) hipStreamSynchronize(s);
    }
    std::printf("%s: %.2f us per launch + wait\n", mode == 0 ? "hipStreamSynchronize" : (mode == 1 ? "spin on pinned word" : "memcpyAsync D2H + synchronize"), 1e6 * total / (reps - 100));
  }
  return 0;
}
