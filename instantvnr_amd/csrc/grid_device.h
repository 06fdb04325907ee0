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
    if (interpolation == 1) p = p * p * (3.0f - 2.0f * p);
    c.w[d] = p;
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

}  // namespace vnr
