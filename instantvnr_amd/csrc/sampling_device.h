// sampling_device.h — device-side replacements for the CUDA texture fetches of the reference.
// Expressions are written in exactly the order the oracle uses (oracle/vnr_oracle.c) and the unit is
// compiled with -ffp-contract=off, so ground-truth sampling and TFN lookups are bit-exact against it.
#pragma once

#include "volume.h"

namespace vnr {

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
__device__ __forceinline__ float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

// tex3D<float>(normalised coords, linear filter, clamp): xB = x*N - 0.5 (core/array.h:79; neural_sampler.cu:150-154)
__device__ __forceinline__ float tex3d(const float* __restrict__ vol, const vec3i dims, float px, float py, float pz)
{
  const float xb = px * (float)dims.x - 0.5f, yb = py * (float)dims.y - 0.5f, zb = pz * (float)dims.z - 0.5f;
  const float fx0 = __builtin_floorf(xb), fy0 = __builtin_floorf(yb), fz0 = __builtin_floorf(zb);
  const float a = xb - fx0, b = yb - fy0, g = zb - fz0;
  const int x0 = clampi((int)fx0, 0, dims.x - 1), x1 = clampi((int)fx0 + 1, 0, dims.x - 1);
  const int y0 = clampi((int)fy0, 0, dims.y - 1), y1 = clampi((int)fy0 + 1, 0, dims.y - 1);
  const int z0 = clampi((int)fz0, 0, dims.z - 1), z1 = clampi((int)fz0 + 1, 0, dims.z - 1);
  const size_t sy = (size_t)dims.x, sz = (size_t)dims.x * dims.y;
  const float v000 = vol[x0 + y0 * sy + z0 * sz], v100 = vol[x1 + y0 * sy + z0 * sz];
  const float v010 = vol[x0 + y1 * sy + z0 * sz], v110 = vol[x1 + y1 * sy + z0 * sz];
  const float v001 = vol[x0 + y0 * sy + z1 * sz], v101 = vol[x1 + y0 * sy + z1 * sz];
  const float v011 = vol[x0 + y1 * sy + z1 * sz], v111 = vol[x1 + y1 * sy + z1 * sz];
  const float c00 = v000 * (1.0f - a) + v100 * a;
  const float c10 = v010 * (1.0f - a) + v110 * a;
  const float c01 = v001 * (1.0f - a) + v101 * a;
  const float c11 = v011 * (1.0f - a) + v111 * a;
  const float c0 = c00 * (1.0f - b) + c10 * b;
  const float c1 = c01 * (1.0f - b) + c11 * b;
  return c0 * (1.0f - g) + c1 * g;
}

// sampleVolume (core/renderer/raytracing.h:105-110): nodal remap, then tex3D
__device__ __forceinline__ float sample_volume_nodal(const float* __restrict__ vol, const vec3i dims, float px, float py, float pz)
{
  const float rx = 1.0f / (float)dims.x, ry = 1.0f / (float)dims.y, rz = 1.0f / (float)dims.z;
  px = px * (1.0f - rx) + 0.5f * rx;
  py = py * (1.0f - ry) + 0.5f * ry;
  pz = pz * (1.0f - rz) + 0.5f * rz;
  return tex3d(vol, dims, px, py, pz);
}

// array1dNodal + tex1D linear filter (core/renderer/raytracing.h:71-81)
__device__ __forceinline__ void tfn_coords(int len, float v, int& i0, int& i1, float& a)
{
  v = clampf(v, 0.0f, 1.0f);
  const float t = __builtin_fmaf(v, (float)(len - 1), 0.5f) * (1.0f / (float)len);
  const float xb = t * (float)len - 0.5f;
  const float f0 = __builtin_floorf(xb);
  a = xb - f0;
  i0 = clampi((int)f0, 0, len - 1);
  i1 = clampi((int)f0 + 1, 0, len - 1);
}

// sampleTransferFunction (core/renderer/raytracing.h:147-155)
__device__ __forceinline__ void tfn_sample(const DeviceTfn& tfn, float value, vec3f& rgb, float& alpha)
{
  const float v = (clampf(value, tfn.range_lo, tfn.range_hi) - tfn.range_lo) * tfn.range_rcp_norm;
  if (tfn.n_colors > 0) {
    int i0, i1; float a;
    tfn_coords(tfn.n_colors, v, i0, i1, a);
    const vec4f c0 = tfn.colors[i0], c1 = tfn.colors[i1];
    rgb.x = c0.x * (1.0f - a) + c1.x * a;
    rgb.y = c0.y * (1.0f - a) + c1.y * a;
    rgb.z = c0.z * (1.0f - a) + c1.z * a;
  } else {
    rgb = {0, 0, 0};
  }
  if (tfn.n_alphas > 0) {
    int i0, i1; float a;
    tfn_coords(tfn.n_alphas, v, i0, i1, a);
    alpha = tfn.alphas[i0] * (1.0f - a) + tfn.alphas[i1] * a;
  } else {
    alpha = 0.0f;
  }
}

// the same lookup with both tables in LDS, read through LDS-typed pointers: a generic pointer that may or may not point to LDS
// compiles to flat_load, which travels through the texture path and is counted on both wait counters (render.hip, march_kernel)
typedef float tfn_float4_t __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(3))) tfn_float4_t* tfn_lds_colors_t;
typedef const __attribute__((address_space(3))) float* tfn_lds_alphas_t;
// `merged` (uniform; the caller's LDS copy of the colour table carries the opacity table in its 4th component, which it can when the two
// tables have the same length: tfn_tables_to_lds): one index / weight computation and one pair of 16-byte reads per sample instead of two of
// each; the same operations on the same operands as the separate lookups.
__device__ __forceinline__ void tfn_sample_lds(const DeviceTfn& tfn, tfn_lds_colors_t colors, tfn_lds_alphas_t alphas, float value, vec3f& rgb, float& alpha,
                                               bool merged = false)
{
  const float v = (clampf(value, tfn.range_lo, tfn.range_hi) - tfn.range_lo) * tfn.range_rcp_norm;
  if (merged) {
    int i0, i1; float a;
    tfn_coords(tfn.n_colors, v, i0, i1, a);
    const tfn_float4_t c0 = colors[i0], c1 = colors[i1];
    rgb.x = c0.x * (1.0f - a) + c1.x * a;
    rgb.y = c0.y * (1.0f - a) + c1.y * a;
    rgb.z = c0.z * (1.0f - a) + c1.z * a;
    alpha = c0.w * (1.0f - a) + c1.w * a;
    return;
  }
  if (tfn.n_colors > 0) {
    int i0, i1; float a;
    tfn_coords(tfn.n_colors, v, i0, i1, a);
    const tfn_float4_t c0 = colors[i0], c1 = colors[i1];
    rgb.x = c0.x * (1.0f - a) + c1.x * a;
    rgb.y = c0.y * (1.0f - a) + c1.y * a;
    rgb.z = c0.z * (1.0f - a) + c1.z * a;
  } else {
    rgb = {0, 0, 0};
  }
  if (tfn.n_alphas > 0) {
    int i0, i1; float a;
    tfn_coords(tfn.n_alphas, v, i0, i1, a);
    alpha = alphas[i0] * (1.0f - a) + alphas[i1] * a;
  } else {
    alpha = 0.0f;
  }
}

// the block's copy of the transfer function tables (the caller synchronises); returns whether the colour table carries the opacities
__device__ __forceinline__ bool tfn_tables_to_lds(const DeviceTfn& tfn, vec4f* s_colors, float* s_alphas, bool allow_merged = true)
{
  const bool merged = allow_merged && tfn.n_colors == tfn.n_alphas && tfn.n_colors > 0;
  for (int e = threadIdx.x; e < tfn.n_colors; e += blockDim.x) {
    vec4f c = tfn.colors[e];
    if (merged) c.w = tfn.alphas[e];
    s_colors[e] = c;
  }
  for (int e = threadIdx.x; e < tfn.n_alphas; e += blockDim.x) s_alphas[e] = tfn.alphas[e];
  return merged;
}

}  // namespace vnr
