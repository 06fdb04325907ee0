// xcd_handoff.hip — does a consumer KERNEL read a producer kernel's output faster when its blocks run on the XCD that wrote it?
// (per-XCD placement of the march kernels behind the evaluation kernel.)  Kernel W: block b writes chunk b of a buffer.
// Kernel R (next launch on the same stream): block b reads chunk (b + shift) % G.  Blocks are dealt to XCDs round-robin by blockIdx (checked
// with HW_REG_XCC_ID), so shift 0 / 8 = the XCD that wrote the chunk (same / another CU), shift 1 / 3 = another XCD.  Reported: time of R.
// build: hipcc --offload-arch=gfx950 -O3 tools/calib/xcd_handoff.hip -o tools/bin/xcd_handoff
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t xcc_id() { uint32_t v; asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(v)); return v & 0xf; }

__global__ void __launch_bounds__(256) write_kernel(float2* buf, uint32_t per_block, float v, uint32_t* xcc)
{
  float2* p = buf + (size_t)blockIdx.x * per_block;
  for (uint32_t i = threadIdx.x; i < per_block; i += 256) p[i] = float2{v + i, v};
  if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = xcc_id();
}
// the consumer's access pattern of the march kernel: every lane 8 loads in flight of 8 bytes, stride 64 elements between them
__global__ void __launch_bounds__(256) read_kernel(const float2* buf, uint32_t per_block, uint32_t shift, float* out, uint32_t* xcc)
{
  const uint32_t src = (blockIdx.x + shift) % gridDim.x;
  const float2* p = buf + (size_t)src * per_block;
  float acc = 0.0f;
  for (uint32_t i = threadIdx.x; i < per_block; i += 256 * 8) {
    float2 v[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) v[j] = p[min(i + 256u * j, per_block - 1)];
#pragma unroll
    for (int j = 0; j < 8; ++j) acc += v[j].x * v[j].y;
  }
  if (acc == 12345.678f) out[blockIdx.x] = acc;
  if (threadIdx.x == 0 && xcc) xcc[blockIdx.x] = xcc_id();
}
// something between the two that touches other memory (the evaluation kernel is not the only writer between two marches)
__global__ void __launch_bounds__(256) noise_kernel(float* a, size_t n) { for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) a[i] = a[i] * 1.0001f + 1.0f; }

int main(int argc, char** argv)
{
  const uint32_t G = argc > 1 ? atoi(argv[1]) : 2048;
  const size_t mb = argc > 2 ? atoi(argv[2]) : 27;     // 3.4 M samples x 8 bytes of results per evaluation launch of the bench frame
  const uint32_t per_block = (uint32_t)(mb * 1024 * 1024 / 8 / G);
  float2* buf; float* out; uint32_t *xw, *xr; float* noise;
  CK(hipMalloc(&buf, (size_t)G * per_block * 8)); CK(hipMalloc(&out, G * 4)); CK(hipMalloc(&xw, G * 4)); CK(hipMalloc(&xr, G * 4));
  const size_t nn = 64u << 20; CK(hipMalloc(&noise, nn * 4)); CK(hipMemset(noise, 0, nn * 4));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  write_kernel<<<G, 256, 0, s>>>(buf, per_block, 1.0f, xw);
  read_kernel<<<G, 256, 0, s>>>(buf, per_block, 0, out, xr);
  CK(hipStreamSynchronize(s));
  std::vector<uint32_t> hw(G), hr(G); CK(hipMemcpy(hw.data(), xw, G * 4, hipMemcpyDeviceToHost)); CK(hipMemcpy(hr.data(), xr, G * 4, hipMemcpyDeviceToHost));
  uint32_t rr = 0, same = 0; for (uint32_t b = 0; b < G; ++b) { rr += hw[b] == (b & 7u); same += hw[b] == hr[b]; }
  printf("grid %u blocks, %zu MB: writer blocks on XCD blockIdx %% 8: %u of %u; reader block b on the XCD of writer block b: %u of %u\n", G, mb, rr, G, same, G);
  for (int with_noise = 0; with_noise < 2; ++with_noise)
    for (uint32_t shift : {0u, 8u, 1u, 3u, 0u, 1u}) {
      std::vector<float> ms;
      for (int rep = 0; rep < 30; ++rep) {
        write_kernel<<<G, 256, 0, s>>>(buf, per_block, (float)rep, nullptr);
        if (with_noise) noise_kernel<<<4096, 256, 0, s>>>(noise, nn);
        CK(hipEventRecord(e0, s));
        read_kernel<<<G, 256, 0, s>>>(buf, per_block, shift, out, nullptr);
        CK(hipEventRecord(e1, s));
        CK(hipEventSynchronize(e1));
        float t; CK(hipEventElapsedTime(&t, e0, e1)); ms.push_back(t);
      }
      std::sort(ms.begin(), ms.end());
      printf("%s shift %u (%s): read kernel median %.2f us, min %.2f us  -> %.0f GB/s\n", with_noise ? "write, 256 MB of other traffic, read;" : "write, read;         ", shift,
             shift % 8 == 0 ? "same XCD" : "other XCD", ms[ms.size() / 2] * 1e3, ms[0] * 1e3, (double)G * per_block * 8 / (ms[ms.size() / 2] * 1e-3) / 1e9);
    }
  return 0;
}
