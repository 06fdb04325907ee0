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
__device__ __forceinline__ uint32_t level_index(const LevelInfo& lv, uint32_t px, uint32_t py, uint32_t pz)
{
  if (lv.hashed) {
    // size is a power of two whenever a level is hashed
    return (px ^ (py * 2654435761u) ^ (pz * 805459861u)) & (lv.size - 1u);
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

template <int F>
__device__ __forceinline__ void gather_corners(const LevelInfo& lv, const CornerSetup& c, table_rsrc_t rsrc,
                                               typename RawFeat<F>::raw_t (&v)[8])
{
  constexpr uint32_t kBytes = (uint32_t)(F * 2);
  const uint32_t soff = lv.offset * kBytes;
  if (lv.hashed) {
    const uint32_t mask = lv.size - 1u;
    const uint32_t hy0 = c.g[1] * 2654435761u, hy1 = hy0 + 2654435761u;
    const uint32_t hz0 = c.g[2] * 805459861u, hz1 = hz0 + 805459861u;
    const uint32_t x0 = c.g[0], x1 = c.g[0] + 1u;
    const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
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

// Corners of one cell out of the brick image (network.h): entry (x, y, z) of a level lives at
//   ((z >> lz) nby + (y >> ly)) nbx + (x >> lx)   bricks of one 128-byte line, then (z & mz, y & my, x & mx) x-fastest inside.
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

// ------------------------------------------------------------------------------------------------
// Grouped form (round 2).  Two measurements set its shape (MI355X, C4 bench frame):
//  * the level-by-level kernel above finishes one level (index arithmetic, 4-5 gathers, wait, blend) before it starts the
//    next: 16 dependent round trips per tile.  Here the loads of a GROUP of levels are issued back to back and the blends
//    follow, so the round trips of a group overlap;
//  * with the brick image the kernel is bound by vector-ALU issue, not by memory: rocprofv3 counts ~2700 VALU instructions per
//    64-sample tile, SQ_ACTIVE_INST_VALU = 25 % of a wave's cycles at 3 waves per SIMD and another 38 % waiting to issue, against
//    31 % in s_waitcnt.  So everything per corner that is not the reference's arithmetic goes: corner loads are structured
//    buffer loads (`buffer_load ... idxen`: the hardware multiplies the entry index by the stride and adds a 48-bit base, so
//    there is no 64-bit address arithmetic per corner and an image larger than 4 GiB is no special case), the position's
//    floor / fraction are one conversion and one v_fract, and the interpolation weights are kept from the issue to the blend.
// Straight-line issue code: no wave-divergent fix-up gathers and no rare-path ballots between the loads of a group, so
//  * a tile with a coordinate outside [0, 1] (or NaN) takes the level-by-level path above, whole (wave-uniform, rare);
//    inside the domain a cell's lower corner is < res on every level, which is all the brick image needs, and a dense table
//    index exceeds the level size at most once (`idx -= size`);
//  * every corner has its own load.  Corners x and x + 1 of a row fall into the same 128-byte line unless the cell sits on a
//    brick face (image) / the level is hashed and x is odd (table), and a gather is priced by the distinct lines of the
//    instruction (tools/calib/ta_model.hip: ~2 cycles per line; one line per lane 129 cycles, 8 lines 25, 1 line <= 15).
// Values and blend order are those of encode_level / encode_level_fast: results are bit-identical.
typedef int int32x4_t __attribute__((ext_vector_type(4)));
// LLVM's structured buffer loads (clang has builtins for the raw forms only): (rsrc, vindex, voffset, soffset, aux)
__device__ _Float16 llvm_struct_buffer_load_f16(int32x4_t, uint32_t, uint32_t, uint32_t, int) __asm("llvm.amdgcn.struct.buffer.load.f16");
__device__ half2_t llvm_struct_buffer_load_v2f16(int32x4_t, uint32_t, uint32_t, uint32_t, int) __asm("llvm.amdgcn.struct.buffer.load.v2f16");
__device__ half4_t llvm_struct_buffer_load_v4f16(int32x4_t, uint32_t, uint32_t, uint32_t, int) __asm("llvm.amdgcn.struct.buffer.load.v4f16");
__device__ float4_t llvm_struct_buffer_load_v4f32(int32x4_t, uint32_t, uint32_t, uint32_t, int) __asm("llvm.amdgcn.struct.buffer.load.v4f32");

template <int F> struct EntryLoad;
template <> struct EntryLoad<1> { static __device__ __forceinline__ half_t load(int32x4_t r, uint32_t i) { return llvm_struct_buffer_load_f16(r, i, 0, 0, 0); } };
template <> struct EntryLoad<2> { static __device__ __forceinline__ half2_t load(int32x4_t r, uint32_t i) { return llvm_struct_buffer_load_v2f16(r, i, 0, 0, 0); } };
template <> struct EntryLoad<4> { static __device__ __forceinline__ half4_t load(int32x4_t r, uint32_t i) { return llvm_struct_buffer_load_v4f16(r, i, 0, 0, 0); } };
template <> struct EntryLoad<8> { static __device__ __forceinline__ half8_t load(int32x4_t r, uint32_t i) { return __builtin_bit_cast(half8_t, llvm_struct_buffer_load_v4f32(r, i, 0, 0, 0)); } };

// descriptor of an array of `stride`-byte entries at a wave-uniform address (scalar registers; index = entry number)
__device__ __forceinline__ int32x4_t make_entry_rsrc(uint64_t base, uint32_t stride, uint32_t n_entries)
{
  const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)base);
  const uint32_t hi = __builtin_amdgcn_readfirstlane((uint32_t)(base >> 32));
  return int32x4_t{(int)lo, (int)((hi & 0xffffu) | (stride << 16)), (int)__builtin_amdgcn_readfirstlane(n_entries), 0x00020000};
}

// `idxen` addressing multiplies index and stride in 32 bits (measured: an array beyond 4 GiB reads wrong entries), and the finest
// levels of a large model exceed that (C4: 4.3 GB of point bricks, 17 GB of cell records).  Such a level is read through a 4 GiB
// WINDOW of its array: the window starts one 2 GiB granule below the granule of the wave's first lane (wave-uniform, scalar),
// the lanes subtract the window's first entry.  A wave of a ray-marched frame spans kilobytes to megabytes; a lane outside the
// window (incoherent coordinates) reads entry 0 instead and flags the tile, which then takes the level-by-level path.
// entries_log2 = log2(2 GiB / entry bytes); `margin` = the largest distance between a cell's first and last entry.
__device__ __forceinline__ uint32_t window_index(uint32_t idx, uint32_t entries_log2, uint32_t margin, uint64_t& base, uint32_t entry_bytes, bool& outside)
{
  const uint32_t granule = __builtin_amdgcn_readfirstlane(idx) >> entries_log2;
  const uint32_t first = granule > 0u ? (granule - 1u) << entries_log2 : 0u;   // scalar
  base += (uint64_t)first * entry_bytes;
  const uint32_t rel = idx - first;
  const bool bad = rel >= (2u << entries_log2) - margin;
  outside = outside || bad;
  return bad ? 0u : rel;
}

template <int F>
__device__ __forceinline__ void issue_level_loads(const LevelInfo& lv, uint32_t interpolation, uint64_t table, uint64_t image,
                                                  float x, float y, float z, typename FeatVec<F>::type (&v)[8], float (&w)[3], bool& outside)
{
  constexpr uint32_t kBytes = (uint32_t)(F * 2);
  // pos_fract for a coordinate in [0, 1]: pos = x scale + 0.5 > 0, so truncation is floor and v_fract is pos - floor(pos), exactly
  uint32_t g[3];
  const float in[3] = {x, y, z};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float p = __builtin_fmaf(in[d], lv.scale, 0.5f);
    g[d] = (uint32_t)p;
    w[d] = __builtin_amdgcn_fractf(p);
  }
  if (interpolation == 1) {   // Smoothstep; a real (wave-uniform) branch: as a select it costs every Linear model 12 instructions per level
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < 3; ++d) w[d] = w[d] * w[d] * (3.0f - 2.0f * w[d]);
  }
  // The wave-uniform choice (brick image / hashed table / dense table) is made on entry INDICES; the eight loads follow it in
  // straight-line code.  (With the loads inside the branches the compiler reconciles the loaded registers of the two sides where
  // they join, i.e. waits for the loads there, and the group's round trips are serial again.)
  uint32_t idx[8];
  int32x4_t rsrc;
  if (lv.brick != 0u) {
    constexpr uint32_t LX = BrickShape<F>::lx, LY = BrickShape<F>::ly, LZ = BrickShape<F>::lz;
    constexpr uint32_t MX = (1u << LX) - 1u, MY = (1u << LY) - 1u, MZ = (1u << LZ) - 1u;
    constexpr uint32_t E = 1u << (LX + LY + LZ);
    const uint32_t res = lv.resolution;
    const uint32_t nbx = (res >> LX) + 1u, nby = (res >> LY) + 1u;   // scalar
    const uint32_t SY = nbx * E, SZ = SY * nby;
    // entry (x, y, z) = brick ((z >> LZ) nby + (y >> LY)) nbx + (x >> LX), then (z & MZ, y & MY, x & MX) inside it
    //                 = x + (E - 2^LX) (x >> LX)  +  (y << LX) + (SY - 2^(LX+LY)) (y >> LY)  +  (z << (LX+LY)) + (SZ - E) (z >> LZ)
    uint32_t e0 = __umul24(g[0] >> LX, E - (1u << LX)) + g[0];
    e0 = __umul24(g[1] >> LY, SY - (1u << (LX + LY))) + e0;
    e0 = __umul24(g[2] >> LZ, SZ - E) + e0;
    e0 += (g[1] << LX) + (g[2] << (LX + LY));
    const uint32_t dx = (g[0] & MX) == MX ? E - MX : 1u;
    const uint32_t dy = (g[1] & MY) == MY ? SY - (MY << LX) : 1u << LX;
    const uint32_t dz = (g[2] & MZ) == MZ ? SZ - (MZ << (LX + LY)) : 1u << (LX + LY);
    uint64_t base = image + (uint64_t)(lv.brick - 1u) * 128u;
    if ((uint64_t)SZ * ((res >> LZ) + 1u) * kBytes >= 0xfc000000ull) {   // wave-uniform: this level's array is (nearly) 4 GiB or more
      constexpr uint32_t kLog2 = F == 1 ? 30u : F == 2 ? 29u : F == 4 ? 28u : 27u;
      e0 = window_index(e0, kLog2, SZ + SY + E, base, kBytes, outside);
    }
    idx[0] = e0; idx[1] = e0 + dx; idx[2] = e0 + dy; idx[3] = idx[2] + dx;
    idx[4] = e0 + dz; idx[5] = idx[4] + dx; idx[6] = idx[4] + dy; idx[7] = idx[6] + dx;
    rsrc = make_entry_rsrc(base, kBytes, 0xffffffffu);
  } else {
    const uint32_t res = lv.resolution, res2 = lv.res2, size = lv.size;
    if (lv.hashed != 0u) {
      const uint32_t hy0 = g[1] * 2654435761u, hy1 = hy0 + 2654435761u;
      const uint32_t hz0 = g[2] * 805459861u, hz1 = hz0 + 805459861u;
      const uint32_t x1 = g[0] + 1u, mask = size - 1u;
      const uint32_t yz[4] = {hy0 ^ hz0, hy1 ^ hz0, hy0 ^ hz1, hy1 ^ hz1};
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) idx[corner] = (((corner & 1) ? x1 : g[0]) ^ yz[corner >> 1]) & mask;
    } else {
      const uint32_t lin = g[0] + __umul24(g[1], res) + __umul24(g[2], res2);
#pragma unroll
      for (int corner = 0; corner < 8; ++corner) {
        const uint32_t d = lin + ((corner & 1) ? 1u : 0u) + ((corner & 2) ? res : 0u) + ((corner & 4) ? res2 : 0u);
        idx[corner] = d >= size ? d - size : d;   // in-domain coordinates wrap at most once (level_index)
      }
    }
    rsrc = make_entry_rsrc(table + (uint64_t)lv.offset * kBytes, kBytes, size);
  }
#pragma unroll
  for (int corner = 0; corner < 8; ++corner) v[corner] = EntryLoad<F>::load(rsrc, idx[corner]);
}

// Keeps a loaded corner untouched until the blend phase: without it the compiler takes a loaded register apart in the block of
// the load (a use, hence a wait for the load, directly behind the loads of every level).
__device__ __forceinline__ void hold_until_here(half_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void hold_until_here(half2_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void hold_until_here(half4_t& v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void hold_until_here(half8_t& v) { asm volatile("" : "+v"(v)); }

// Cell records (network.h): the eight corners of a cell are the records (x, y, z) and (x, y, z + 1), 16 bytes each, in bricks
// of 2 x 2 x 2 records:  r = x + 6 (x >> 1)  +  2 y + (SY - 4) (y >> 1)  +  4 z + (SZ - 8) (z >> 1),  SY = 8 nbx, SZ = SY nby.
__device__ __forceinline__ void issue_level_records(const LevelInfo& lv, uint32_t interpolation, uint64_t image, float x, float y, float z,
                                                    half2_t (&v)[8], float (&w)[3], bool& outside)
{
  uint32_t g[3];
  const float in[3] = {x, y, z};
#pragma unroll
  for (int d = 0; d < 3; ++d) {
    const float p = __builtin_fmaf(in[d], lv.scale, 0.5f);
    g[d] = (uint32_t)p;
    w[d] = __builtin_amdgcn_fractf(p);
  }
  if (interpolation == 1) {
    asm volatile("" ::: "memory");
#pragma unroll
    for (int d = 0; d < 3; ++d) w[d] = w[d] * w[d] * (3.0f - 2.0f * w[d]);
  }
  const uint32_t nb = (lv.resolution + 1u) >> 1;   // scalar
  const uint32_t SY = nb * 8u, SZ = SY * nb;
  uint32_t r = __umul24(g[0] >> 1, 6u) + g[0];
  r = __umul24(g[1] >> 1, SY - 4u) + r;
  r = __umul24(g[2] >> 1, SZ - 8u) + r;
  r += (g[1] << 1) + (g[2] << 2);
  uint64_t base = image + (uint64_t)(lv.brick - 1u) * 128u;
  if ((uint64_t)SZ * ((lv.resolution + 2u) >> 1) * 16u >= 0xfc000000ull) r = window_index(r, 27u, SZ, base, 16u, outside);   // wave-uniform
  const uint32_t r1 = r + ((g[2] & 1u) ? SZ - 4u : 4u);
  const int32x4_t rsrc = make_entry_rsrc(base, 16u, 0xffffffffu);
  // corner (cx, cy, cz) is word cx + 2 cy of record cz: the corner order of blend_level.  (The words of a loaded register
  // quadruple are sub-registers: taking them apart costs no instruction and no wait.)
  const float4_t a = llvm_struct_buffer_load_v4f32(rsrc, r, 0, 0, 0);
  const float4_t b = llvm_struct_buffer_load_v4f32(rsrc, r1, 0, 0, 0);
  // (through scalars: hipcc of ROCm 7.2 compiles __builtin_bit_cast of an ext-vector ELEMENT to a read of element 0)
  const float a0 = a.x, a1 = a.y, a2 = a.z, a3 = a.w, b0 = b.x, b1 = b.y, b2 = b.z, b3 = b.w;
  v[0] = __builtin_bit_cast(half2_t, a0); v[1] = __builtin_bit_cast(half2_t, a1); v[2] = __builtin_bit_cast(half2_t, a2); v[3] = __builtin_bit_cast(half2_t, a3);
  v[4] = __builtin_bit_cast(half2_t, b0); v[5] = __builtin_bit_cast(half2_t, b1); v[6] = __builtin_bit_cast(half2_t, b2); v[7] = __builtin_bit_cast(half2_t, b3);
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

// the blend of one level from its 8 loaded corners and the weights of issue_level_loads
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

template <int F>
__device__ __forceinline__ void encode_level_fast(const LevelInfo& lv, uint32_t interpolation, table_rsrc_t rsrc, const uint8_t* image,
                                                  float x, float y, float z, half_t* out)
{
  typedef typename RawFeat<F>::raw_t raw_t;
  const CornerSetup c = level_setup(lv, interpolation, x, y, z);
  raw_t v[8];
  if (lv.brick == 0u || !gather_corners_brick<F>(lv, c, image, v)) gather_corners<F>(lv, c, rsrc, v);
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
