// pt_device.h — the per-ray state and decisions of the delta-tracking path tracer (rendering modes 13 - 15), shared by the streaming
// kernels (render.hip pt_kernel) and the in-shader kernel (in_shader.h).  Split out of render.hip in round 5.
#pragma once

#include "march_device.h"

namespace vnr {

// ================================================================================================ path tracing (mode 14)
// Sample-streaming path tracer: core/renderer/method_pathtracing.cu:532-813 (DeltaTrackingIter with the macrocell majorants,
// iterative_take_sample, iterative_shade, raygen / shade kernels, do_path_tracing_iterative); VARYING_MAJORANT = 1 there
// (ADAPTIVE_SAMPLING is not defined in that translation unit, :24-27).  One volume sample per alive ray and iteration.
// Structure here: one kernel per iteration does shade + the delta tracking to the next tentative collision (the reference's
// raygen / shade kernels, fused), survivors stay in their 64-ray group's slots and pt_compact_kernel packs them in group order
// (the order-preserving compaction of the ray marcher) and writes the queue records the evaluation kernel reads, so the ray
// count never visits the host either.
constexpr int kPtPlanes = 26;  // dwords of state per ray, one plane each: see PtRay::load / store
struct PtRays { float* base; uint32_t stride; };

struct PtRay {
  float tnear, tfar;
  uint32_t pidx; bool shadow;
  vec3f org, dir;
  uint32_t scatter_index;
  vec3f sample_coord;
  float majorant;
  vec3f L, throughput;
  DDAState it;
  uint32_t rng;   // gdt::LCG state (EXTERNAL): next = 1664525 state + 1013904223, float = low 24 bits / 2^24
  __device__ __forceinline__ float next_float()
  {
    rng = 1664525u * rng + 1013904223u;
    return (float)(rng & 0x00FFFFFFu) / (float)0x01000000;
  }
  __device__ __forceinline__ void load(const PtRays r, uint32_t i)
  {
    const float* b = r.base + i;
    const uint32_t st = r.stride;
    const uint32_t bits = __float_as_uint(b[0]);
    shadow = (bits & 1u) != 0u; pidx = bits >> 1;
    org = {b[1 * st], b[2 * st], b[3 * st]};
    dir = {b[4 * st], b[5 * st], b[6 * st]};
    scatter_index = __float_as_uint(b[7 * st]);
    sample_coord = {b[8 * st], b[9 * st], b[10 * st]};
    majorant = b[11 * st];
    L = {b[12 * st], b[13 * st], b[14 * st]};
    throughput = {b[15 * st], b[16 * st], b[17 * st]};
    rng = __float_as_uint(b[18 * st]);
    it.t_next = {b[19 * st], b[20 * st], b[21 * st]};
    it.cell = {(int)__float_as_uint(b[22 * st]), (int)__float_as_uint(b[23 * st]), (int)__float_as_uint(b[24 * st])};
    it.next_cell_begin = b[25 * st];
  }
  __device__ __forceinline__ void store(const PtRays r, uint32_t i) const
  {
    float* b = r.base + i;
    const uint32_t st = r.stride;
    b[0] = __uint_as_float((pidx << 1) | (shadow ? 1u : 0u));
    b[1 * st] = org.x; b[2 * st] = org.y; b[3 * st] = org.z;
    b[4 * st] = dir.x; b[5 * st] = dir.y; b[6 * st] = dir.z;
    b[7 * st] = __uint_as_float(scatter_index);
    b[8 * st] = sample_coord.x; b[9 * st] = sample_coord.y; b[10 * st] = sample_coord.z;
    b[11 * st] = majorant;
    b[12 * st] = L.x; b[13 * st] = L.y; b[14 * st] = L.z;
    b[15 * st] = throughput.x; b[16 * st] = throughput.y; b[17 * st] = throughput.z;
    b[18 * st] = __uint_as_float(rng);
    b[19 * st] = it.t_next.x; b[20 * st] = it.t_next.y; b[21 * st] = it.t_next.z;
    b[22 * st] = __uint_as_float((uint32_t)it.cell.x); b[23 * st] = __uint_as_float((uint32_t)it.cell.y); b[24 * st] = __uint_as_float((uint32_t)it.cell.z);
    b[25 * st] = it.next_cell_begin;
  }
};

// DeltaTrackingIter::hashit (:545-573)
__device__ __forceinline__ bool pt_hashit(const RenderParams& p, PtRay& r, float& rayt)
{
  const vec3f m_dir = r.dir * p.mc_rcp;
  bool found_hit = false;
  float tau = -logf(1.0f - r.next_float());
  float t = r.it.next_cell_begin + r.tnear;
  while (dda_next(r.it, m_dir, r.tnear, r.tfar, p.mc_dims, [&](vec3i c, float /*t0*/, float t1) -> bool {
    r.majorant = opacity_upper_bound(p, c) * p.density_scale;
    if (fabsf(r.majorant) <= FLT_EPSILON) return true;  // next macrocell; t is not advanced, as in the reference
    tau -= (t1 - t) * (r.majorant * 1.0f);
    t = t1;
    if (tau > 0.0f) return true;
    t = t + tau / (r.majorant * 1.0f);
    found_hit = true;
    r.it.next_cell_begin = t - r.tnear;
    rayt = t;
    return false;
  })) {}
  return found_hit;
}

// uniform_sample_sphere (raytracing.h:253-270); phi = 2 * M_PI * s.x is a double expression rounded to float
__device__ __forceinline__ vec3f pt_uniform_sample_sphere(float sx, float sy)
{
  const float phi = (float)(2 * M_PI * (double)sx);
  const float cos_theta = 1.0f - 2.0f * sy;
  const float sin_theta = 2.0f * sqrtf(sy * (1.0f - sy));
  float sp, cp;
  sincosf(phi, &sp, &cp);
  return {cp * sin_theta, sp * sin_theta, cos_theta};
}

// iterative_take_sample (:598-636)
__device__ __forceinline__ bool pt_take_sample(const RenderParams& p, PtRay& r)
{
  float t;
  if (pt_hashit(p, r, t)) { r.sample_coord = r.org + t * r.dir; return true; }
  if (r.scatter_index > 0u) {  // no light accumulation for primary rays
    if (r.shadow) {
      r.L = r.L + r.throughput;   // * light_directional_rgb = 1 (instantvnr_types.h:147)
      r.shadow = false;
      const float s0 = r.next_float(), s1 = r.next_float();
      r.dir = xfm_vector(p.wto, pt_uniform_sample_sphere(s0, s1));
      if (p.pt_reset_interval) { r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE; }
      if (!intersect_box(r.tnear, r.tfar, r.org, r.dir, p.bbox_lo, p.bbox_hi)) return false;   // mode 14: the interval is not reset first
      dda_init(r.it, r.org * p.mc_rcp, r.dir * p.mc_rcp, r.tnear, p.mc_dims);
      if (pt_hashit(p, r, t)) { r.sample_coord = r.org + t * r.dir; return true; }
      // the bounce leaves the volume at once: mode 14 ends the path here without the ambient term (:631-635 falls through to
      // `return false`), the in-shader / monolithic estimator adds it on its next loop trip (:447-452, 1008-1013)
      if (p.pt_reset_interval) r.L = r.L + 1.5f * r.throughput;
    } else {
      r.L = r.L + 1.5f * r.throughput;   // light_ambient = 1.5 (instantvnr_types.h:146)
    }
  }
  return false;
}

// iterative_shade (:638-677)
__device__ __forceinline__ bool pt_shade(const RenderParams& p, const DeviceTfn& tfn, PtRay& r, float value)
{
  vec3f albedo; float a;
  tfn_sample(tfn, value, albedo, a);
  if (r.next_float() * r.majorant >= a * p.density_scale) return true;   // null collision
  if (r.shadow) {
    r.shadow = false;
    const float s0 = r.next_float(), s1 = r.next_float();
    r.dir = xfm_vector(p.wto, pt_uniform_sample_sphere(s0, s1));
    if (p.pt_reset_interval) { r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE; }
  } else {
    if (r.scatter_index > 4u) {  // russian_roulette (:366-376), russian_roulette_length = 4
      const float q = fminf(0.95f, max3f(r.throughput.x, r.throughput.y, r.throughput.z));
      if (r.next_float() > q) return false;
      r.throughput = {r.throughput.x / q, r.throughput.y / q, r.throughput.z / q};
    }
    ++r.scatter_index;
    r.org = r.sample_coord;
    r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;
    r.throughput = r.throughput * (0.6f * albedo);   // PHASE(albedo) = albedo * 0.6f (:35)
    r.shadow = true;
    r.dir = p.shadow_dir;
  }
  if (!intersect_box(r.tnear, r.tfar, r.org, r.dir, p.bbox_lo, p.bbox_hi)) return false;
  dda_init(r.it, r.org * p.mc_rcp, r.dir * p.mc_rcp, r.tnear, p.mc_dims);
  return true;
}

}  // namespace vnr
