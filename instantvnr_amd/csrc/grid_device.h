// grid_device.h — device-side hash-grid arithmetic shared by the fused inference kernel, the training
// forward and the grid backward.  Semantics: core/networks/tcnn_impl_decoder.cu:7-175 (encode_one_level)
// plus tcnn's pos_fract / grid_index / fast_hash (EXTERNAL, see oracle/vnr_oracle.c for the restatement).
//
// The translation unit is compiled with -ffp-contract=off: every fused multiply-add is an explicit
// __builtin_fmaf placed exactly where the oracle has one, so the encode is bit-exact against it.
#pragma once

#include "network.h"

namespace vnr {

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float2_t __attribute__((ext_vector_type(2)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float8_t __attribute__((ext_vector_type(8)));
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef uint32_t uint4_t __attribute__((ext_vector_type(4)));

struct CornerSetup {
  float w[3];        // fractional position per dim (after the interpolation function)
  uint32_t g[3];     // lower grid corner per dim
};

__device__ __forceinline__ CornerSetup level_setup(const LevelInfo& lv, uint32_t interpolation, float x, float y, float z)
{
  CornerSetup c;
  const float in[3] = {x, y, z};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    float p = __builtin_fmaf(in[d], lv.scale, 0.5f);
    const float t = __builtin_floorf(p);
    c.g[d] = (uint32_t)(int32_t)t;
    p -= t;
    c.w[d] = p;
  }
  if (interpolation == 1) {   // Smoothstep; a real (wave-uniform) branch: as a select it costs every Linear model ~10 instructions per level
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < 3; ++d) c.w[d] = c.w[d] * c.w[d] * (3.0f - 2.0f * c.w[d]);
  }
  return c;
}

// entry index (not yet multiplied by F) of corner (px,py,pz) in level lv; exact `% size` semantics
// lv.hashed: 0 = dense over all three dimensions (Hash levels that fit their table, Dense grids), 1 = prime-XOR hash,
// 2 / 3 / 4 = Tiled grid whose stride walk stopped after 1 / 2 / 3 dimensions (EXTERNAL tcnn grid_index: the walk ends as soon as the
// stride exceeds the level's size, and only a Hash grid replaces the partial sum by the hash; the sum is then taken modulo the size)
__device__ __forceinline__ uint32_t level_index(const LevelInfo& lv, uint32_t px, uint32_t py, uint32_t pz)
{
  if (lv.hashed == 1u) {
    // size is a power of two whenever a level is hashed
    return (px ^ (py * 2654435761u) ^ (pz * 805459861u)) & (lv.size - 1u);
  }
  if (lv.hashed >= 2u) {
    uint32_t idx = px;
    if (lv.hashed >= 3u) idx += py * lv.resolution;
    if (lv.hashed >= 4u) idx += pz * (lv.resolution * lv.resolution);
    return idx % lv.size;
  }
  uint32_t idx = px + py * lv.resolution + pz * (lv.resolution * lv.resolution);
  if (idx >= lv.size) {
    idx -= lv.size;                       // in-domain coordinates wrap at most once
    if (idx >= lv.size) idx %= lv.size;   // out-of-domain inputs: keep the reference's modulo
  }
  return idx;
}

__device__ __forceinline__ float corner_weight(const CornerSetup& c, int corner)
{
  // weight = ((1 * wx) * wy) * wz in dimension order (tcnn_impl_decoder.cu:100-113)
  const float wx = (corner & 1) ? c.w[0] : 1.0f - c.w[0];
  const float wy = (corner & 2) ? c.w[1] : 1.0f - c.w[1];
  const float wz = (corner & 4) ? c.w[2] : 1.0f - c.w[2];
  return (wx * wy) * wz;
}

template <int F> struct FeatVec;
template <> struct FeatVec<1> { typedef half_t type; };
template <> struct FeatVec<2> { typedef half2_t type; };
template <> struct FeatVec<4> { typedef half4_t type; };
template <> struct FeatVec<8> { typedef half8_t type; };

// One level of the encoding for one sample: 8-corner gather + fp16-accumulated trilinear blend
// (`result[f] += (T)(weight * data)`, tcnn_impl_decoder.cu:117-122).  out[f], f < F.
template <int F>
__device__ __forceinline__ void encode_level(const LevelInfo& lv, uint32_t interpolation, const half_t* __restrict__ table,
                                             float x, float y, float z, half_t* out)
{
  typedef typename FeatVec<F>::type vec_t;
  const CornerSetup c = level_setup(lv, interpolation, x, y, z);
  const vec_t* __restrict__ base = (const vec_t*)(table + (size_t)lv.offset * F);
  vec_t v[8];
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const uint32_t idx = level_index(lv, c.g[0] + (corner & 1), c.g[1] + ((corner >> 1) & 1), c.g[2] + ((corner >> 2) & 1));
    v[corner] = base[idx];
  }
  half_t acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = (half_t)0.0f;
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const float w = corner_weight(c, corner);
    const half_t* d = (const half_t*)&v[corner];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float prod = w * (float)d[f];
      // `(T)(weight * data)` rounds twice (f32 product, then f16).  For scalar halves hipcc would otherwise
      // select v_fma_mixlo_f16, which rounds the exact product once and differs in rare halfway cases.
      if (F == 1) asm volatile("" : "+v"(prod));
      acc[f] = acc[f] + (half_t)prod;
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f) out[f] = acc[f];
}

// ------------------------------------------------------------------------------------------------
// Fast path used by the fused kernels: identical results, cheaper instruction stream.
//  * 32-bit buffer addressing (one SRSRC for the whole table, level base as scalar offset) instead of 64-bit
//    pointer arithmetic per corner;
//  * dense levels: one base index + 7 adds with 24-bit multiplies; the (rare) wrap-around / out-of-domain case
//    is detected per wave and falls back to the exact `% size` formula;
//  * hashed levels: (g+1)*P == g*P + P, so 2 full multiplies per level instead of 16.
// ------------------------------------------------------------------------------------------------
typedef __amdgpu_buffer_rsrc_t table_rsrc_t;

#ifndef VNR_HASH_QUAD
#define VNR_HASH_QUAD 1   // hashed F = 2 levels: aligned four-entry groups (gather_corners); 0 = aligned pairs (rounds 1-5), for A/B builds
#endif

__device__ __forceinline__ table_rsrc_t make_table_rsrc(const void* base, uint32_t bytes)
{
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
}

template <int F> struct RawFeat;
template <> struct RawFeat<1> {
  typedef unsigned short raw_t;
  static __device__ __forceinline__ raw_t load(table_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b16(r, voff, soff, 0); }
};
template <> struct RawFeat<2> {
  typedef uint32_t raw_t;
  static __device__ __forceinline__ raw_t load(table_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b32(r, voff, soff, 0); }
};
template <> struct RawFeat<4> {
  typedef uint32_t raw_t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ raw_t load(table_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0); }
};
template <> struct RawFeat<8> {
  typedef uint4_t raw_t;
  static __device__ __forceinline__ raw_t load(table_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }
};

// Two x-adjacent corners of one (y,z) row in ONE load when they are adjacent in memory.  The gather cost on gfx950
// is per lane-address in the texture addresser (measured: ~44 TA cycles per 64-lane dword gather, TA 87 % busy),
// not per byte, so halving the number of gather instructions is worth far more than the wider loads cost.
template <int F> struct PairFeat { static constexpr bool enabled = false; };
template <> struct PairFeat<2> {
  static constexpr bool enabled = true;
  typedef uint32_t pair_t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ pair_t load(table_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b64(r, voff, soff, 0); }
  static __device__ __forceinline__ uint32_t lo(pair_t p) { return p.x; }
  static __device__ __forceinline__ uint32_t hi(pair_t p) { return p.y; }
};
template <> struct PairFeat<4> {
  static constexpr bool enabled = true;
  typedef uint4_t pair_t;
  typedef uint32_t half_pair_t __attribute__((ext_vector_type(2)));
  static __device__ __forceinline__ pair_t load(table_rsrc_t r, uint32_t voff, uint32_t soff) { return __builtin_amdgcn_raw_buffer_load_b128(r, voff, soff, 0); }
  static __device__ __forceinline__ half_pair_t lo(pair_t p) { return half_pair_t{p.x, p.y}; }
  static __device__ __forceinline__ half_pair_t hi(pair_t p) { return half_pair_t{p.z, p.w}; }
};

// exact (slow, rare) dense indices: kept out of line so the hot instruction stream stays small
typedef uint32_t uint8x32_t __attribute__((ext_vector_type(8)));
__device__ __noinline__ uint8x32_t level_indices_exact(const LevelInfo lv, uint32_t g0, uint32_t g1, uint32_t g2)
{
  uint8x32_t idx;
#pragma unroll
  for (int corner = 0; corner < 8; ++corner)
    idx[corner] = level_index(lv, g0 + (corner & 1), g1 + ((corner >> 1) & 1), g2 + ((corner >> 2) & 1));
  return idx;
}

// GENERAL (here and below): the instance also covers what only the rarer models need: Tiled grid levels, Nearest interpolation (and, in
// infer_tile.h, the transcendental activations).  The instances without it are the instruction stream of the common models (Hash / Dense
// grid, Linear / Smoothstep, ReLU / None) and nothing else: with everything in one kernel the evaluation kernel of the bench model grew from
// 5 213 to 23 993 instructions and lost 8 % of its rate (instruction cache), although none of the added code ever ran.
template <int F, bool GENERAL = false>
__device__ __forceinline__ void gather_corners(const LevelInfo& lv, const CornerSetup& c, table_rsrc_t rsrc,
                                               typename RawFeat<F>::raw_t (&v)[8])
{
  constexpr uint32_t kBytes = (uint32_t)(F * 2);
  const uint32_t soff = lv.offset * kBytes;
  if (GENERAL && lv.hashed >= 2u) {   // Tiled grid level (wave-uniform): the exact index of every corner, out of line
    const uint8x32_t idx = level_indices_exact(lv, c.g[0], c.g[1], c.g[2]);
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) v[corner] = RawFeat<F>::load(rsrc, idx[corner] * kBytes, soff);
  } else if (lv.hashed) {
    const uint32_t mask = lv.size - 1u;
    const uint32_t hy0 = c.g[1] * 2654435761u, hy1 = hy0 + 2654435761u;
    const uint32_t hz0 = c.g[2] * 805459861u, hz1 = hz0 + 805459861u;
    const uint32_t x0 = c.g[0], x1 = c.g[0] + 1u;
    const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
    // The grouped loads below need a table that holds a whole group and starts on a group boundary: level sizes are multiples of 8 entries or
    // the whole (power-of-two) table, so that fails only for tables of fewer entries than a group, log2_hashmap_size 0 / 1 (wave-uniform, out of line).
    constexpr uint32_t kGroupMask = (F == 2 && VNR_HASH_QUAD) ? 3u : PairFeat<F>::enabled ? 1u : 0u;
    if (kGroupMask && mask < kGroupMask) {
      const uint8x32_t idx = level_indices_exact(lv, c.g[0], c.g[1], c.g[2]);
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) v[corner] = RawFeat<F>::load(rsrc, idx[corner] * kBytes, soff);
      return;
    }
#if VNR_HASH_QUAD
    if constexpr (F == 2) {
      // x + 1 = x ^ (2^(t+1) - 1) with t the trailing ones of x, and the hash of the other two coordinates is XORed onto x, so the +x
      // neighbour of entry i0 is i0 ^ 1 for even x, i0 ^ 3 for x = 1 (mod 4): both inside the aligned group of FOUR entries (16 bytes) that
      // holds i0.  One 16-byte load of that group per (y, z) row serves three lanes in four; only x = 3 (mod 4) pays a second gather (the
      // pair load above it served one lane in two).  A gather is priced per (instruction, lane address), not per byte.
      const uint32_t m = (x0 & 1u) ? 3u : 1u;
      const bool far_x = (x0 & 3u) == 3u;
      uint32_t i1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t i0 = (x0 ^ yz[q]) & mask;
        i1[q] = (x1 ^ yz[q]) & mask;
        const uint4_t g = __builtin_amdgcn_raw_buffer_load_b128(rsrc, (i0 & ~3u) * kBytes, soff, 0);
        const uint32_t k = i0 & 3u, k2 = k ^ m;
        const uint32_t a0 = (k & 1u) ? g.y : g.x, a1 = (k & 1u) ? g.w : g.z;       // entry k of its half, for either half
        const uint32_t b0 = (k & 1u) ? g.x : g.y, b1 = (k & 1u) ? g.z : g.w;       // ... and entry k ^ 1
        v[2 * q] = (k & 2u) ? a1 : a0;
        v[2 * q + 1] = (k2 & 2u) ? b1 : b0;                                        // (k2 & 1) == (k & 1) ^ 1 for m = 1 and m = 3
      }
      if (far_x) {  // divergent: one lane in four
#pragma unroll
        for (int q = 0; q < 4; ++q) v[2 * q + 1] = RawFeat<F>::load(rsrc, i1[q] * kBytes, soff);
      }
    } else
#endif
    if constexpr (PairFeat<F>::enabled) {
      // even x: (x+1)^h == (x^h)^1, i.e. the second corner is the other half of the aligned entry pair
      const bool odd_x = (x0 & 1u) != 0u;
      uint32_t i1[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t i0 = (x0 ^ yz[q]) & mask;
        i1[q] = (x1 ^ yz[q]) & mask;
        const auto pr = PairFeat<F>::load(rsrc, (i0 & ~1u) * kBytes, soff);
        const bool up = (i0 & 1u) != 0u;
        v[2 * q] = up ? PairFeat<F>::hi(pr) : PairFeat<F>::lo(pr);
        v[2 * q + 1] = up ? PairFeat<F>::lo(pr) : PairFeat<F>::hi(pr);
      }
      if (odd_x) {  // divergent: only the odd-x lanes pay for a second gather per (y,z) row
#pragma unroll
        for (int q = 0; q < 4; ++q) v[2 * q + 1] = RawFeat<F>::load(rsrc, i1[q] * kBytes, soff);
      }
    } else {
#pragma unroll
      for (int corner = 0; corner < 8; ++corner)
        v[corner] = RawFeat<F>::load(rsrc, ((((corner & 1) ? x1 : x0) ^ yz[corner >> 1]) & mask) * kBytes, soff);
    }
  } else {
    const uint32_t res = lv.resolution, res2 = lv.res2;
    const uint32_t base = c.g[0] + __umul24(c.g[1], res) + __umul24(c.g[2], res2);
    const bool bad = (c.g[0] > res) | (c.g[1] > res) | (c.g[2] > res) | (base + 1u + res + res2 >= lv.size);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) {  // wave-uniform and rare: keep the exact modulo semantics
      const uint8x32_t idx = level_indices_exact(lv, c.g[0], c.g[1], c.g[2]);
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) v[corner] = RawFeat<F>::load(rsrc, idx[corner] * kBytes, soff);
    } else if constexpr (PairFeat<F>::enabled) {
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // corners (2q, 2q+1) are entries idx and idx + 1
        const uint32_t idx = base + ((q & 1) ? res : 0u) + ((q & 2) ? res2 : 0u);
        const auto pr = PairFeat<F>::load(rsrc, idx * kBytes, soff);
        v[2 * q] = PairFeat<F>::lo(pr);
        v[2 * q + 1] = PairFeat<F>::hi(pr);
      }
    } else {
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        const uint32_t idx = base + ((corner & 1) ? 1u : 0u) + ((corner & 2) ? res : 0u) + ((corner & 4) ? res2 : 0u);
        v[corner] = RawFeat<F>::load(rsrc, idx * kBytes, soff);
      }
    }
  }
}

// Corners of one cell out of the brick image (network.h).  F = 1, 4, 8: entry (x, y, z) of a level lives at
//   ((z >> lz) nby + (y >> ly)) nbx + (x >> lx)   bricks of one 128-byte line, then (z & mz, y & my, x & mx) x-fastest inside
// (F = 2 has a brick with a repeated column of its own, first branch below).
// The step to the +1 neighbour is a constant unless the cell sits on a brick face, so one index and three selects address
// all 8 corners; corners x and x + 1 of a row come with ONE load unless x is the last column of its brick (1 lane in 4, where
// the hash pays a second gather for every odd x).  Plain 64-bit addresses: the finest level of the bench model is 4.3 GB.
// Returns false (wave-uniform) when a lane's cell is outside the level's grid, i.e. the coordinate was outside [0, 1]: the
// caller then reads the parameter blob, whose index arithmetic is defined for any coordinate.
template <int F>
__device__ __forceinline__ bool gather_corners_brick(const LevelInfo& lv, const CornerSetup& c, const uint8_t* __restrict__ image,
                                                     typename RawFeat<F>::raw_t (&v)[8])
{
  typedef typename RawFeat<F>::raw_t raw_t;
  if constexpr (F == 2) {
    // F = 2: bricks of 8 x 2 x 2 entries whose 8th column REPEATS the first column of the +x neighbour brick (7 useful columns per
    // brick).  Corners x and x + 1 of a row then always come with one 8-byte load: 4 gathers per level and no fix-up gathers, where the
    // plain 4 x 4 x 2 brick needs a second gather for every lane in the last column of its brick (1 in 4, so practically every wave
    // issued 8 gather instructions per level).  The evaluation kernel is bound by the rate of its gather requests:
    // 12.6 -> 13.3 G samples/s for 14 % more image bytes.  lv.pad1 = bricks per row = resolution / 7 + 1.
    typedef typename PairFeat<F>::pair_t pair_t;
    const uint32_t res = lv.resolution;
    const bool bad = (c.g[0] >= res) | (c.g[1] >= res) | (c.g[2] >= res);
    if (__builtin_amdgcn_ballot_w64(bad) != 0ull) return false;
    const uint32_t nbx = lv.pad1, nby = (res >> 1) + 1u;   // scalar
    const uint32_t bx = (c.g[0] * 9363u) >> 16;            // x / 7, exact for x < 13 107 (build_brick_image bricks no finer level)
    const uint32_t wx = c.g[0] - 7u * bx, wy = c.g[1] & 1u, wz = c.g[2] & 1u;
    const uint32_t brick = __umul24(__umul24(c.g[2] >> 1, nby) + (c.g[1] >> 1), nbx) + bx;
    const uint32_t e0 = brick * 32u + ((wz << 4) | (wy << 3) | wx);
    const uint32_t dy = wy ? nbx * 32u - 8u : 8u;
    const uint32_t dz = wz ? nbx * nby * 32u - 16u : 16u;
    const uint8_t* base = image + (size_t)(lv.brick - 1u) * 128u;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t e = e0 + ((q & 1) ? dy : 0u) + ((q & 2) ? dz : 0u);
      const pair_t pr = *(const pair_t*)(base + (size_t)e * 4u);
      v[2 * q] = PairFeat<F>::lo(pr);
      v[2 * q + 1] = PairFeat<F>::hi(pr);
    }
    return true;
  }
  constexpr uint32_t LX = BrickShape<F>::lx, LY = BrickShape<F>::ly, LZ = BrickShape<F>::lz;
  constexpr uint32_t MX = (1u << LX) - 1u, MY = (1u << LY) - 1u, MZ = (1u << LZ) - 1u;
  constexpr uint32_t E = 1u << (LX + LY + LZ);   // entries per brick
  constexpr uint32_t kBytes = (uint32_t)(F * 2);
  static_assert(E * kBytes == 128, "a brick is one 128-byte line");
  const uint32_t res = lv.resolution;
  const bool bad = (c.g[0] >= res) | (c.g[1] >= res) | (c.g[2] >= res);
  if (__builtin_amdgcn_ballot_w64(bad) != 0ull) return false;
  const uint32_t nbx = (res >> LX) + 1u, nby = (res >> LY) + 1u;   // scalar
  const uint32_t wx = c.g[0] & MX, wy = c.g[1] & MY, wz = c.g[2] & MZ;
  const uint32_t brick = __umul24(__umul24(c.g[2] >> LZ, nby) + (c.g[1] >> LY), nbx) + (c.g[0] >> LX);
  const uint32_t e0 = brick * E + ((wz << (LX + LY)) | (wy << LX) | wx);
  const uint32_t dx = wx == MX ? E - MX : 1u;
  const uint32_t dy = wy == MY ? nbx * E - (MY << LX) : 1u << LX;
  const uint32_t dz = wz == MZ ? nbx * nby * E - (MZ << (LX + LY)) : 1u << (LX + LY);
  const uint8_t* base = image + (size_t)(lv.brick - 1u) * 128u;
  if constexpr (PairFeat<F>::enabled) {
    typedef typename PairFeat<F>::pair_t pair_t;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const uint32_t e = e0 + ((q & 1) ? dy : 0u) + ((q & 2) ? dz : 0u);
      const pair_t pr = *(const pair_t*)(base + (size_t)e * kBytes);   // entries e, e + 1 (the image ends with a spare line)
      v[2 * q] = PairFeat<F>::lo(pr);
      v[2 * q + 1] = PairFeat<F>::hi(pr);
    }
    if (wx == MX) {  // divergent: the +x neighbour is the first column of the next brick
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const uint32_t e = e0 + ((q & 1) ? dy : 0u) + ((q & 2) ? dz : 0u) + dx;
        v[2 * q + 1] = *(const raw_t*)(base + (size_t)e * kBytes);
      }
    }
  } else {
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const uint32_t e = e0 + ((corner & 1) ? dx : 0u) + ((corner & 2) ? dy : 0u) + ((corner & 4) ? dz : 0u);
      v[corner] = *(const raw_t*)(base + (size_t)e * kBytes);
    }
  }
  return true;
}

// `(T)(weight * data)` for the two halves of a register: the fp32 products, each rounded, then both rounded to fp16.
// v_fma_mix_f32 takes the fp16 operand as it is (x y + 0 in fp32: rounds like x y; a product of -0 becomes +0, which an fp16 sum
// that starts at +0 cannot tell apart), so a pair costs 2 + 1 instructions instead of the 2 conversions + multiply + conversion
// hipcc selects for the C expression.
__device__ __forceinline__ half2_t weighted_pair(float w, uint32_t pair)
{
  float lo, hi;
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel_hi:[1,0,0]" : "=v"(lo) : "v"(pair), "v"(w));
  asm("v_fma_mix_f32 %0, %1, %2, 0 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(hi) : "v"(pair), "v"(w));
  return __builtin_convertvector(float2_t{lo, hi}, half2_t);
}

// the blend of one level from its 8 loaded corners and the weights of level_setup
template <int F>
__device__ __forceinline__ void blend_level(const float (&wd)[3], const typename FeatVec<F>::type (&v)[8], half_t* out)
{
  const float wx0 = 1.0f - wd[0], wx1 = wd[0], wy0 = 1.0f - wd[1], wy1 = wd[1], wz0 = 1.0f - wd[2], wz1 = wd[2];
  const float wxy[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
  if constexpr (F == 1) {
    half_t acc = (half_t)0.0f;
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      float prod = (wxy[corner & 3] * ((corner & 4) ? wz1 : wz0)) * (float)v[corner];
      asm volatile("" : "+v"(prod));  // see encode_level
      acc = acc + (half_t)prod;
    }
    out[0] = acc;
  } else {
    typedef uint32_t words_t __attribute__((ext_vector_type(F / 2)));
    half2_t acc[F / 2];
#pragma unroll
    for (int q = 0; q < F / 2; ++q) acc[q] = half2_t{(half_t)0.0f, (half_t)0.0f};
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      const float w = wxy[corner & 3] * ((corner & 4) ? wz1 : wz0);  // ((wx * wy) * wz)
      const words_t words = __builtin_bit_cast(words_t, v[corner]);
#pragma unroll
      for (int q = 0; q < F / 2; ++q) {
        uint32_t word;
        if constexpr (F == 2) word = __builtin_bit_cast(uint32_t, words); else word = words[q];
        acc[q] = acc[q] + weighted_pair(w, word);   // fp16 accumulate, corner by corner (tcnn_impl_decoder.cu:117-122)
      }
    }
#pragma unroll
    for (int q = 0; q < F / 2; ++q) { out[2 * q] = acc[q].x; out[2 * q + 1] = acc[q].y; }
  }
}

// quantize_threshold (tcnn_impl_decoder.cu:120: `if (fabsf(data) < quantize_threshold) data = 0.f` on every corner value before the blend)
template <int F>
__device__ __forceinline__ typename RawFeat<F>::raw_t quantize_raw(typename RawFeat<F>::raw_t raw, float qt)
{
  typedef typename FeatVec<F>::type vec_t;
  vec_t v = __builtin_bit_cast(vec_t, raw);
  if constexpr (F == 1) {
    if (fabsf((float)v) < qt) v = (half_t)0.0f;
  } else {
#pragma unroll
    for (int f = 0; f < F; ++f) if (fabsf((float)v[f]) < qt) v[f] = (half_t)0.0f;
  }
  return __builtin_bit_cast(typename RawFeat<F>::raw_t, v);
}

template <int F, bool GENERAL = false>
__device__ __forceinline__ void encode_level_fast(const LevelInfo& lv, uint32_t interpolation, table_rsrc_t rsrc, const uint8_t* image,
                                                  float x, float y, float z, half_t* out, float quantize_threshold = 0.0f)
{
  typedef typename RawFeat<F>::raw_t raw_t;
  const CornerSetup c = level_setup(lv, interpolation, x, y, z);
  if (GENERAL && interpolation == 2u) {   // Nearest (tcnn_impl_decoder.cu:73-94): the entry of the lower corner as it is, one gather
    const raw_t one = RawFeat<F>::load(rsrc, level_index(lv, c.g[0], c.g[1], c.g[2]) * (uint32_t)(F * 2), lv.offset * (uint32_t)(F * 2));
    const typename FeatVec<F>::type vv = __builtin_bit_cast(typename FeatVec<F>::type, one);
    if constexpr (F == 1) out[0] = vv;
    else {
#pragma unroll
      for (int f = 0; f < F; ++f) out[f] = vv[f];
    }
    return;
  }
  raw_t v[8];
  if (lv.brick == 0u || !gather_corners_brick<F>(lv, c, image, v)) gather_corners<F, GENERAL>(lv, c, rsrc, v);
  if (GENERAL && quantize_threshold > 0.0f) {   // wave-uniform
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) v[corner] = quantize_raw<F>(v[corner], quantize_threshold);
  }
  if constexpr (F >= 2) {   // the blend of the grouped form: same values, 4 instead of 5-6 instructions per corner and feature pair
    typename FeatVec<F>::type vv[8];
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) vv[corner] = __builtin_bit_cast(typename FeatVec<F>::type, v[corner]);
    blend_level<F>(c.w, vv, out);
    return;
  }
  const float wx0 = 1.0f - c.w[0], wx1 = c.w[0], wy0 = 1.0f - c.w[1], wy1 = c.w[1], wz0 = 1.0f - c.w[2], wz1 = c.w[2];
  const float wxy[4] = {wx0 * wy0, wx1 * wy0, wx0 * wy1, wx1 * wy1};
  half_t acc[F];
#pragma unroll
  for (int f = 0; f < F; ++f) acc[f] = (half_t)0.0f;
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) {
    const float w = wxy[corner & 3] * ((corner & 4) ? wz1 : wz0);  // ((wx * wy) * wz)
    const half_t* d = (const half_t*)&v[corner];
#pragma unroll
    for (int f = 0; f < F; ++f) {
      float prod = w * (float)d[f];
      if (F == 1) asm volatile("" : "+v"(prod));  // see encode_level
      acc[f] = acc[f] + (half_t)prod;
    }
  }
#pragma unroll
  for (int f = 0; f < F; ++f) out[f] = acc[f];
}

}  // namespace vnr
