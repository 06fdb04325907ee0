// Calibrates rocprofv3's FETCH_SIZE / TCC_EA0_RDREQ for the access patterns of the fused hash-grid kernel:
// random 4-, 8- and 16-byte gathers (one per lane, every lane its own random 128-B line of a table far larger than
// L2 + Infinity Cache) and a coalesced 16-B/lane streaming read as the control (MI355X_MICROARCH.md, HBM section:
// FETCH_SIZE reports 1/2 of a wide streaming read on gfx950; "other access widths are uncalibrated").
//   hipcc -O3 --offload-arch=gfx950 gather_calib.hip -o gather_calib ; rocprofv3 --pmc FETCH_SIZE -- ./gather_calib
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint64_t mix(uint64_t x)
{
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull; x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull; x ^= x >> 33;
  return x;
}

template <typename T>
__global__ void gather_kernel(const T* __restrict__ table, uint64_t n_elems_mask, uint32_t* __restrict__ out, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const T v = table[mix(i + 1) & n_elems_mask];
  const uint32_t* w = (const uint32_t*)&v;
  uint32_t s = 0;
  for (unsigned k = 0; k < sizeof(T) / 4; ++k) s ^= w[k];
  if (s == 0x12345678u) out[i] = s;  // keep the load alive, (almost) never store
}

// Request granularity of a gather miss: every lane picks a random 128-B-aligned line and reads one dword from its first
// 64-B half (HALVES == 1) or one dword from each half (HALVES == 2).  If the fabric request is the whole 128-B line the
// second load hits in L2 and TCC_EA0_RDREQ is the same for both; if it is a 64-B half, HALVES == 2 doubles RDREQ.
template <int HALVES>
__global__ void halves_kernel(const uint32_t* __restrict__ table, uint64_t n_lines_mask, uint32_t* __restrict__ out, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t* line = table + (mix(i + 1) & n_lines_mask) * 32u;  // 32 dwords = 128 B
  uint32_t s = line[0];
  if (HALVES == 2) s ^= line[16] + 1u;  // +64 B: other half of the same line
  if (s == 0x12345678u) out[i] = s;
}

__global__ void stream16_kernel(const uint4* __restrict__ table, uint32_t* __restrict__ out, uint32_t n)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint4 v = table[i];
  if ((v.x ^ v.y ^ v.z ^ v.w) == 0x12345678u) out[i] = v.x;
}

int main()
{
  const size_t bytes = 4ull << 30;  // 4 GiB table: no reuse, nothing stays in the 256 MiB Infinity Cache
  const uint32_t n = 1u << 26;      // 64 Mi lanes per kernel
  void* table; uint32_t* out;
  CHECK(hipMalloc(&table, bytes));
  CHECK(hipMalloc(&out, (size_t)n * 4));
  CHECK(hipMemset(table, 1, bytes));
  CHECK(hipMemset(out, 0, (size_t)n * 4));
  const dim3 block(256), grid(n / 256);
  for (int rep = 0; rep < 3; ++rep) {
    gather_kernel<uint32_t><<<grid, block>>>((const uint32_t*)table, bytes / 4 - 1, out, n);
    gather_kernel<uint2><<<grid, block>>>((const uint2*)table, bytes / 8 - 1, out, n);
    gather_kernel<uint4><<<grid, block>>>((const uint4*)table, bytes / 16 - 1, out, n);
    stream16_kernel<<<grid, block>>>((const uint4*)table, out, n);  // reads 1 GiB
    halves_kernel<1><<<grid, block>>>((const uint32_t*)table, bytes / 128 - 1, out, n);
    halves_kernel<2><<<grid, block>>>((const uint32_t*)table, bytes / 128 - 1, out, n);
  }
  CHECK(hipDeviceSynchronize());
  printf("lanes per kernel: %u (gather4/8/16: one random line each; stream16: %zu bytes)\n", n, (size_t)n * 16);
  return 0;
}
