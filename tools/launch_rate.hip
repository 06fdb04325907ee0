// GPU box: how many kernels per second the device starts and retires, by number of streams, grid size and LDS per block (what a chain of small
// launches costs apart from its work).  build: hipcc --offload-arch=gfx950 -O2 tools/launch_rate.hip -o tools/bin/launch_rate
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { std::printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
extern __shared__ float s_buf[];
__global__ void tiny(float* out, int spin)
{
  float a = threadIdx.x;
  for (int i = 0; i < spin; ++i) a = a * 1.0001f + 0.5f;
  if (a == 12345.678f) out[0] = a + s_buf[0];
}
int main()
{
  float* d; CK(hipMalloc(&d, 1024));
  CK(hipFuncSetAttribute((const void*)tiny, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
  const int K = 2000;
  struct Cfg { int blocks, threads, lds, spin; const char* name; };
  const Cfg cfgs[] = {{1, 64, 0, 0, "1 wave"}, {128, 256, 0, 0, "128 x 256"}, {128, 256, 83 * 1024, 0, "128 x 256, 83 KB LDS"}, {1024, 256, 20 * 1024, 0, "1024 x 256, 20 KB LDS"},
                      {128, 256, 83 * 1024, 20000, "128 x 256, 83 KB LDS, ~40 us of work"}};
  for (const Cfg& c : cfgs)
    for (int S : {1, 2, 4, 8}) {
      std::vector<hipStream_t> st(S);
      for (auto& s : st) CK(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
      for (int w = 0; w < 2; ++w) {
        CK(hipDeviceSynchronize());
        auto t0 = std::chrono::steady_clock::now();
        for (int k = 0; k < K; ++k) for (auto& s : st) tiny<<<c.blocks, c.threads, c.lds, s>>>(d, c.spin);
        CK(hipDeviceSynchronize());
        const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (w) std::printf("%-40s %d stream(s): %7.2f us per kernel per stream, %7.2f us per kernel in all\n", c.name, S, us / K, us / K / S);
      }
      for (auto& s : st) CK(hipStreamDestroy(s));
    }
  return 0;
}
