// What does a 64-lane gather cost the texture addresser / L1 on gfx950, as a function of how many distinct 128-B lines its
// lanes touch and of where those lanes sit?  (Round 1 measured ~44 cycles per 64-lane dword gather with one line per lane and
// built the fused kernel's cost model on it; the LDS staging of round 2 rests on the other end: a wave whose 64 lanes read
// one or a few lines.)  Every lane issues LOADS independent 8-byte loads per loop trip from a table that stays in L2.
//   hipcc -O3 --offload-arch=gfx950 ta_model.hip -o ta_model && ./ta_model
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>

#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

__device__ __forceinline__ uint32_t mix32(uint32_t x)
{
  x ^= x >> 16; x *= 0x7feb352du; x ^= x >> 15; x *= 0x846ca68bu; x ^= x >> 16;
  return x;
}

enum Pattern { RANDOM_LINE = 0, REGION_SHUFFLED = 1, REGION_SORTED = 2, BROADCAST = 3, COALESCED_B128 = 4, QUAD_LINES = 5 };

// REGION bytes: the span all 64 lanes of a wave stay inside in one instruction (REGION_* patterns)
template <int PATTERN, int REGION>
__global__ void __launch_bounds__(256) gather_kernel(const uint8_t* __restrict__ table, uint32_t table_mask, uint32_t trips, uint32_t* __restrict__ out)
{
  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6;
  uint32_t acc = 0;
  constexpr int LOADS = 8;
  for (uint32_t t = 0; t < trips; ++t) {
    uint32_t addr[LOADS];
#pragma unroll
    for (int k = 0; k < LOADS; ++k) {
      const uint32_t wave_base = (mix32(wave * 977u + t * LOADS + k) * 128u) & table_mask & ~(uint32_t)(REGION - 1);
      uint32_t a;
      if (PATTERN == RANDOM_LINE) a = (mix32((wave * 64u + lane) * 31u + t * LOADS + k) * 8u) & table_mask;
      else if (PATTERN == REGION_SHUFFLED) a = wave_base + ((mix32(lane * 2654435761u + t * LOADS + k) * 8u) & (uint32_t)(REGION - 1));
      else if (PATTERN == REGION_SORTED) a = wave_base + ((lane * (uint32_t)(REGION / 64)) & (uint32_t)(REGION - 1) & ~7u);
      else if (PATTERN == BROADCAST) a = wave_base;
      else if (PATTERN == QUAD_LINES) a = ((mix32((wave * 16u + (lane >> 2)) * 31u + t * LOADS + k) * 128u) & table_mask) + (lane & 3u) * 8u;  // 4 adjacent lanes share a line
      else a = wave_base + lane * 16u;
      addr[k] = a;
    }
    if (PATTERN == COALESCED_B128) {
#pragma unroll
      for (int k = 0; k < LOADS; ++k) { const uint4 v = *(const uint4*)(table + addr[k]); acc ^= v.x ^ v.y ^ v.z ^ v.w; }
    } else {
#pragma unroll
      for (int k = 0; k < LOADS; ++k) { const uint2 v = *(const uint2*)(table + addr[k]); acc ^= v.x ^ v.y; }
    }
  }
  if (acc == 0x12345678u) out[wave] = acc;
}

// the LDS side: 64 lanes read 8 bytes each from a REGION-byte window of LDS at shuffled offsets (bank conflicts as they fall)
template <int REGION>
__global__ void __launch_bounds__(256) lds_kernel(uint32_t trips, uint32_t* __restrict__ out)
{
  __shared__ __attribute__((aligned(16))) uint8_t s[4 * 4096];
  const uint32_t lane = threadIdx.x & 63u, w = threadIdx.x >> 6;
  for (uint32_t i = threadIdx.x; i < 4 * 4096 / 4; i += 256) ((uint32_t*)s)[i] = i * 2654435761u;
  __syncthreads();
  uint32_t acc = 0;
  for (uint32_t t = 0; t < trips; ++t) {
#pragma unroll
    for (int k = 0; k < 8; ++k) {
      const uint32_t a = w * 4096u + ((mix32(lane * 2654435761u + t * 8u + k) * 8u) & (uint32_t)(REGION - 1));
      const uint2 v = *(const uint2*)(s + a);
      acc ^= v.x ^ v.y;
    }
  }
  if (acc == 0x12345678u) out[blockIdx.x] = acc;
}

template <typename K>
static void time_it(const char* name, K&& launch, double wave_instrs_per_cu, double clock_ghz)
{
  hipEvent_t a, b;
  CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
  launch(); launch();
  CHECK(hipDeviceSynchronize());
  CHECK(hipEventRecord(a));
  for (int r = 0; r < 5; ++r) launch();
  CHECK(hipEventRecord(b));
  CHECK(hipEventSynchronize(b));
  float ms = 0;
  CHECK(hipEventElapsedTime(&ms, a, b));
  ms /= 5;
  printf("%-58s %8.3f ms   %7.1f cycles per wave-instruction per CU (at %.1f GHz)\n", name, ms, ms * 1e-3 * clock_ghz * 1e9 / wave_instrs_per_cu, clock_ghz);
}

int main(int argc, char** argv)
{
  const size_t table_bytes = argc > 1 ? (size_t)atoll(argv[1]) << 20 : (size_t)2 << 20;   // default 2 MiB: stays in every XCD's L2
  uint8_t* table; uint32_t* out;
  CHECK(hipMalloc(&table, table_bytes));
  CHECK(hipMemset(table, 1, table_bytes));
  CHECK(hipMalloc(&out, 1 << 24));
  hipDeviceProp_t prop; CHECK(hipGetDeviceProperties(&prop, 0));
  const int cus = prop.multiProcessorCount;
  const uint32_t mask = (uint32_t)(table_bytes - 1);
  const uint32_t trips = 256;
  const int waves_per_cu = 16;
  const dim3 grid(cus * waves_per_cu / 4), block(256);
  const double instrs = (double)waves_per_cu * trips * 8;   // wave-instructions per CU
  const double ghz = 2.1;
  printf("table %zu MiB, %d CUs, %d waves per CU, %u x 8 loads per lane\n", table_bytes >> 20, cus, waves_per_cu, trips);
#define RUN(P, R, label) time_it(label, [&] { gather_kernel<P, R><<<grid, block>>>(table, mask, trips, out); }, instrs, ghz)
  RUN(RANDOM_LINE, 128, "b64 gather, every lane its own random line");
  RUN(QUAD_LINES, 128, "b64 gather, 4 adjacent lanes per line (16 lines)");
  RUN(REGION_SHUFFLED, 4096, "b64 gather, 64 lanes shuffled inside 4 KiB (32 lines)");
  RUN(REGION_SHUFFLED, 1024, "b64 gather, 64 lanes shuffled inside 1 KiB (8 lines)");
  RUN(REGION_SHUFFLED, 256, "b64 gather, 64 lanes shuffled inside 256 B (2 lines)");
  RUN(REGION_SHUFFLED, 128, "b64 gather, 64 lanes shuffled inside 128 B (1 line)");
  RUN(REGION_SORTED, 1024, "b64 gather, lanes in address order over 1 KiB");
  RUN(REGION_SORTED, 512, "b64 load, lane i -> base + 8 i (coalesced 512 B)");
  RUN(BROADCAST, 128, "b64 gather, all lanes one address");
  RUN(COALESCED_B128, 1024, "b128 load, lane i -> base + 16 i (coalesced 1 KiB)");
  time_it("LDS ds_read_b64, 64 lanes shuffled inside 1 KiB", [&] { lds_kernel<1024><<<grid, block>>>(trips, out); }, instrs, ghz);
  time_it("LDS ds_read_b64, 64 lanes shuffled inside 4 KiB", [&] { lds_kernel<4096><<<grid, block>>>(trips, out); }, instrs, ghz);
  time_it("LDS ds_read_b64, 64 lanes shuffled inside 256 B", [&] { lds_kernel<256><<<grid, block>>>(trips, out); }, instrs, ghz);
  return 0;
}
