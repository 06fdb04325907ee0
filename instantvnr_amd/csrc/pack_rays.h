// pack_rays.h — the ray lists of the streaming marcher and the packing of their survivors (render.hip), as a device function so that it
// can run either as a kernel of its own (compact_rays_kernel) or as the prologue of the evaluation kernel of the same iteration
// (network_infer.hip: both only need what march(it) wrote, and a small share of the frame is a chain of short launches).
#pragma once
#include "common.h"

namespace vnr {

struct RayList {  // per ray payload, SoA (RayMarchingData, method_raymarching.cu:59-93)
  uint32_t* pixel_index;
  float* jitter;
  float* alpha;
  vec3f* color;
  vec3i* cell;
  vec3f* t_next;
  float* next_cell_begin;
  uint32_t* sample_base;
  uint32_t* sample_count;
};

// SINGLE_SHADE_HEURISTIC only (inter_highest_*, method_raymarching.cu:84-87): per ray, the sample that has contributed most so
// far; [0] belongs to the dense ray list, [1] to the scratch list.  A kernel argument of its own, behind the others, so that
// the instances that never touch it do not drag it through their scalar registers: as members of RayList these six pointers
// cost the unshaded march kernel 6 more spilled SGPRs (90 -> 96) and the bench 3 % of its frame rate (A/B on one box,
// tools/ab/ab.sh, n = 3 each: 195-198 against 201-204 frames/s; 199-200 against 199-201 with this layout).
struct SshLists {
  vec3f* org[2];
  vec3f* color[2];
  float* alpha[2];
};

// device counters of one ray part (render.hip)
enum { C_RAYS0 = 0, C_RAYS1 = 1, C_SAMPLES0 = 2, C_SAMPLES1 = 3, C_HIT = 4, C_STAT_SAMPLES = 6, C_STAT_REFRAYS = 8, C_COUNT = 16 };

struct PackArgs {
  RayList src, dst;              // scratch list march(it) left its survivors in (group g: slots 64 g ..), dense list of march(it + 1)
  const uint32_t* ray_counts;    // per 64-ray group: survivors | rays alive when the march began to emit << 8
  uint32_t n_first;              // rays of the first march (it = 0)
  uint32_t* counters;
  int parity, first, ssh, grad;
  SshLists ssh_lists;
  uint32_t* host_alive;          // pinned: alive rays after this iteration
  uint32_t* host_stats;          // pinned copy of the counters
  uint32_t n_blocks;             // work items of pack_rays_block<WAVES> (scratch slots / (64 WAVES)); 0: nothing to pack
};

#if defined(__HIPCC__)
// Packs the 64-ray groups march_kernel left in the scratch list (group g holds ray_counts[g] rays in slots 64 g ..) into the
// dense list, in group order.  One thread per scratch slot; a work item (WAVES groups, one block of 64 WAVES threads) first sums the
// counts of all groups before it (at most 64 KiB of L2-resident counts), so no second launch and no dependency between items is
// needed.  The item that holds the last group publishes the number of alive rays; item 0 also clears the sample counter the next
// march adds to.  s_part: WAVES words of LDS; every thread of the block calls this with the same `item`.
template <int WAVES>
__device__ __forceinline__ void pack_rays_block(const PackArgs& a, uint32_t item, uint32_t* s_part)
{
  constexpr uint32_t T = 64u * WAVES;  // threads = scratch slots per item; WAVES groups
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  uint32_t* counters = a.counters;
  const int parity = a.parity;
  const uint32_t n_in = a.first ? a.n_first : counters[C_RAYS0 + parity];   // rays the march that just ran consumed
  const uint32_t n_groups = ((n_in + 255u) & ~255u) >> 6;                   // groups it wrote a count for
  if (item == 0 && tid == 0) counters[C_SAMPLES0 + (parity ^ 1)] = 0;
  // the host reads the alive-ray count and the frame statistics from pinned memory the kernel writes itself: a copy
  // engine operation between two kernels of a stream costs more than either of the small kernels
  if (n_groups == 0) {  // nothing marched: the statistics are those of the previous launch
    if (item == 0 && tid == 0) { counters[C_RAYS0 + (parity ^ 1)] = 0; *a.host_alive = 0; }
    if (item == 0 && tid >= C_HIT && tid < C_COUNT) a.host_stats[tid] = counters[tid];
    return;
  }
  const uint32_t g0 = item * (uint32_t)WAVES;
  if (g0 >= n_groups) return;
  const uint32_t* __restrict__ ray_counts = a.ray_counts;
  // a group's count: survivors in the low byte, rays that were alive when the march began to emit above it
  uint32_t sum = 0;
  for (uint32_t g = tid; g < g0; g += T) sum += ray_counts[g] & 0xffu;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
  if (lane == 0) s_part[wave] = sum;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < WAVES; ++w) before += s_part[w];
  // counts of this item's groups: lane l < WAVES of every wave holds count[g0 + l]
  const uint32_t mine = (lane < (uint32_t)WAVES && g0 + lane < n_groups) ? ray_counts[g0 + lane] & 0xffu : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < WAVES; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if ((int)lane >= d) incl += y;
  }
  const uint32_t count = __shfl(mine, (int)wave), base = before + __shfl(incl, (int)wave) - count;
  const uint32_t block_total = __shfl(incl, WAVES - 1);
  const bool last_block = g0 + (uint32_t)WAVES >= n_groups;  // (block-uniform) the item that holds the last group
  if (last_block && tid == 0) { counters[C_RAYS0 + (parity ^ 1)] = before + block_total; *a.host_alive = before + block_total; }
  if (lane < count) {
    const RayList& src = a.src;
    const RayList& dst = a.dst;
    const uint32_t from = ((g0 + wave) << 6) + lane, to = base + lane;
    dst.pixel_index[to] = src.pixel_index[from];
    dst.jitter[to] = src.jitter[from];
    dst.alpha[to] = src.alpha[from];
    dst.color[to] = src.color[from];
    dst.cell[to] = src.cell[from];
    dst.t_next[to] = src.t_next[from];
    dst.next_cell_begin[to] = src.next_cell_begin[from];
    dst.sample_base[to] = src.sample_base[from];
    dst.sample_count[to] = src.sample_count[from];
    if (a.ssh) {  // scratch [1] -> dense [0]
      const SshLists& sl = a.ssh_lists;
      sl.org[0][to] = sl.org[1][from]; sl.color[0][to] = sl.color[1][from]; sl.alpha[0][to] = sl.alpha[1][from];
    }
  }
  // Frame statistics of the march that just ran (it used to count them with three device-scope atomics per wave), summed by
  // the last item after its copies so that the code needs no more registers than the copies do
  if (last_block) {
    uint32_t asum = 0;
    for (uint32_t g = tid; g < n_groups; g += T) asum += ray_counts[g] >> 8;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) asum += __shfl_xor(asum, d);
    __syncthreads();  // s_part is read above
    if (lane == 0) s_part[wave] = asum;
    __syncthreads();
    if (tid == 0) {
      uint32_t alive_total = 0;
      for (int w = 0; w < WAVES; ++w) alive_total += s_part[w];
      const uint32_t records = counters[C_SAMPLES0 + parity];
      unsigned long long* c64 = (unsigned long long*)counters;
      c64[C_STAT_SAMPLES / 2] += (unsigned long long)(a.grad ? records >> 2 : records);
      c64[C_STAT_REFRAYS / 2] += (unsigned long long)alive_total;
      if (a.first) counters[C_HIT] += alive_total;
    }
    __syncthreads();
    if (tid >= C_HIT && tid < C_COUNT) a.host_stats[tid] = ((volatile uint32_t*)counters)[tid];
  }
}
#endif

}  // namespace vnr
