// A chain of K tiny dependent kernels ending in a word the host polls: plain stream launches vs ONE hipGraphLaunch of
// the same chain captured once (the kernel arguments change every repetition: hipGraphExecKernelNodeSetParams on the
// last node, as a real chain would need for its sequence numbers / pointers).  VERDICT r4 asked for hipGraph on the fixed
// part of the W-ref chain; this measures what a graph buys on this box before anything is rebuilt around it.
//   hipcc --offload-arch=gfx950 -O2 graph_latency.hip -o graph_latency
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
__global__ void link(unsigned* data, int k) {
  if (threadIdx.x == 0) data[k + 1] = data[k] + 1u;
}
__global__ void last(unsigned* data, int k, volatile unsigned* flag, unsigned v) {
  if (threadIdx.x == 0) {
    data[k + 1] = data[k] + 1u;
    __threadfence_system();
    *flag = v;
  }
}
// the insertion's case: four dependent kernels whose ~0.5 KB by-value argument changes with every call
struct Big {
  unsigned v[120];
};
__global__ void big_link(Big b, unsigned* data, int k) {
  if (threadIdx.x == 0) data[k + 1] = data[k] + b.v[k & 63];
}
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
int main() {
  hipStream_t s;
  CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
  unsigned* flag;
  CK(hipHostMalloc((void**)&flag, 64, hipHostMallocDefault));
  unsigned* data;
  CK(hipMalloc((void**)&data, 4096));
  CK(hipMemset(data, 0, 4096));
  *flag = 0;
  const int reps = 1000;
  unsigned seq = 0;
  for (int K : {2, 5, 10, 20}) {
    // ---- stream launches
    double total = 0;
    for (int i = 0; i < reps; ++i) {
      ++seq;
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < K - 1; ++k) hipLaunchKernelGGL(link, dim3(1), dim3(64), 0, s, data, k);
      hipLaunchKernelGGL(last, dim3(1), dim3(64), 0, s, data, K - 1, flag, seq);
      while (*(volatile unsigned*)flag != seq) {}
      auto t1 = std::chrono::steady_clock::now();
      if (i >= 100) total += std::chrono::duration<double>(t1 - t0).count();
    }
    const double stream_us = 1e6 * total / (reps - 100);
    // ---- graph: capture once, update the last node's arguments every repetition
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K - 1; ++k) hipLaunchKernelGGL(link, dim3(1), dim3(64), 0, s, data, k);
    unsigned v0 = 0;
    hipLaunchKernelGGL(last, dim3(1), dim3(64), 0, s, data, K - 1, flag, v0);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(g, nodes.data(), &nn));
    hipGraphNode_t tail = nullptr;
    hipKernelNodeParams kp{};
    for (hipGraphNode_t nd : nodes) {
      hipKernelNodeParams p{};
      if (hipGraphKernelNodeGetParams(nd, &p) == hipSuccess && p.func == (void*)last) { tail = nd; kp = p; }
    }
    if (tail == nullptr) { std::printf("K=%d: tail node not found\n", K); continue; }
    total = 0;
    double total_fixed = 0;
    for (int mode = 0; mode < 2; ++mode) {  // 0: with a parameter update per launch, 1: without (flag value fixed, flag reset by host)
      for (int i = 0; i < reps; ++i) {
        ++seq;
        int kk = K - 1;
        unsigned vv = mode == 0 ? seq : v0 + 1;
        void* args[4] = {&data, &kk, &flag, &vv};
        if (mode == 1) *flag = 0;
        auto t0 = std::chrono::steady_clock::now();
        if (mode == 0 || i == 0) {
          kp.kernelParams = args;
          CK(hipGraphExecKernelNodeSetParams(ge, tail, &kp));
        }
        CK(hipGraphLaunch(ge, s));
        while (*(volatile unsigned*)flag != vv) {}
        auto t1 = std::chrono::steady_clock::now();
        if (i >= 100) (mode == 0 ? total : total_fixed) += std::chrono::duration<double>(t1 - t0).count();
      }
      CK(hipStreamSynchronize(s));
    }
    std::printf("K=%2d dependent tiny kernels, launch -> completion word seen: stream launches %.2f us, hipGraph with one node's parameters updated %.2f us, hipGraph unchanged %.2f us\n",
                K, stream_us, 1e6 * total / (reps - 100), 1e6 * total_fixed / (reps - 100));
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  // ---- HOST time to enqueue four kernels with a 480-byte argument each (no wait inside the timed part): four launches
  //      against hipGraphExecKernelNodeSetParams x 4 + one hipGraphLaunch
  {
    Big b{};
    const int K = 4, reps2 = 500;
    double t_stream = 0, t_graph = 0;
    for (int i = 0; i < reps2; ++i) {
      b.v[0] = i;
      auto t0 = std::chrono::steady_clock::now();
      for (int k = 0; k < K; ++k) hipLaunchKernelGGL(big_link, dim3(256), dim3(256), 0, s, b, data, k);
      auto t1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      if (i >= 50) t_stream += std::chrono::duration<double>(t1 - t0).count();
    }
    hipGraph_t g;
    hipGraphExec_t ge;
    CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
    for (int k = 0; k < K; ++k) hipLaunchKernelGGL(big_link, dim3(256), dim3(256), 0, s, b, data, k);
    CK(hipStreamEndCapture(s, &g));
    CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
    size_t nn = 0;
    CK(hipGraphGetNodes(g, nullptr, &nn));
    std::vector<hipGraphNode_t> nodes(nn);
    CK(hipGraphGetNodes(g, nodes.data(), &nn));
    std::vector<hipKernelNodeParams> kps(nn);
    for (size_t j = 0; j < nn; ++j) CK(hipGraphKernelNodeGetParams(nodes[j], &kps[j]));
    for (int i = 0; i < reps2; ++i) {
      b.v[0] = i;
      auto t0 = std::chrono::steady_clock::now();
      for (size_t j = 0; j < nn; ++j) {
        int kk = static_cast<int>(j);
        void* args[3] = {&b, &data, &kk};
        hipKernelNodeParams kp = kps[j];
        kp.kernelParams = args;
        kp.gridDim = dim3(256 + (i & 1));  // the grid changes with the cloud's size as well
        CK(hipGraphExecKernelNodeSetParams(ge, nodes[j], &kp));
      }
      CK(hipGraphLaunch(ge, s));
      auto t1 = std::chrono::steady_clock::now();
      CK(hipStreamSynchronize(s));
      if (i >= 50) t_graph += std::chrono::duration<double>(t1 - t0).count();
    }
    std::printf("host time to enqueue 4 dependent kernels with a 480-byte argument each: 4 launches %.2f us, 4 x SetParams + hipGraphLaunch %.2f us\n",
                1e6 * t_stream / (reps2 - 50), 1e6 * t_graph / (reps2 - 50));
    hipGraphExecDestroy(ge);
    hipGraphDestroy(g);
  }
  return 0;
}
