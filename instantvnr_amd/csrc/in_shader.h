// in_shader.h — the in-shader ray marcher (rendering modes 6, 9, 12) and path tracer (mode 15) on a neural volume: kernel templates over
// the encoding shape (F, padded width), the FullyFusedMLP width W and the kernel kind (GENERAL: grid_device.h), instantiated per width
// and kind in in_shader_w{16,32,64}[g].hip and in_shader_w128.hip (round 5; the reference instantiates widths 16 / 32 / 64 x F in {1, 2, 4, 8},
// core/renderer/method_raymarching.cu:1192-1244, and refuses 128 at :1210, which comes free with the template here).
//
// Reference: network_raymarching_traceray / _transmittance / _iterator (core/renderer/method_raymarching.cu:310-356, 981-1128) with
// DeviceNeuralVolume::sample (core/networks/tcnn_impl.cu:34-102): one thread per pixel walks its ray to the end, and every step of
// the loop evaluates the network for the whole thread block (`block_any` keeps the block together).
//
// Here: one WAVE is one 8 x 8 pixel tile and the unit that stays together.  A lane carries its ray as a small state machine
// (macrocell DDA + position inside the current cell); per trip every lane that still has a ray advances to its next sample, the
// wave evaluates the network for its 64 sample points on the matrix cores (eval_tile, infer_tile.h: the code of the evaluation
// kernel, so a value is bit-identical to what the streaming path reads back), and every lane composes its own sample.  No ray
// lists, no sample queue, no launches per iteration: one launch per frame.  The price is that a lane whose ray has ended idles until
// the wave's longest ray has, which the streaming path's compaction avoids.  Measured on the bench frame (mode 6 / mode 9, ms per
// frame): whole frame 8.2 / 36.8 against 4.2 / 12.2 on the streaming path, 1/8 share 1.98 / 6.6 against 0.70 / 1.8.  So the streaming
// path stays the default for these modes and this kernel is what vnrAmdRendererSetInShaderKernel(1) selects.
//
// Arithmetic: RayMarchingIter::exec's, uninterrupted (no resume rounding; the oracle's n_iters = 512 form).  One deliberate
// difference from the reference's loop: network_raymarching_iterator lets a thread whose macrocell is EMPTY sample it anyway when
// another thread of the block has a non-empty one (its `block_any(non_empty && alive)`), at the largest adaptive step; with an
// exact macrocell those samples classify to zero opacity, and which threads share a block is an accident of the launch.  Empty
// cells are skipped here, as in the streaming modes.
#pragma once

#include "infer_tile.h"
#include "march_device.h"
#include "pt_device.h"

namespace vnr {

// the walk of one ray: dda3 over the macrocells (dda.h:140-287) with the adaptive step inside a cell, one sample per call
struct RayWalk {
  vec3f o, d;            // object space
  float t0, t1;          // the ray's interval in the volume's box
  float jitter, step;
  vec3f ts;              // |1 / (d / cell spacing)|
  DDAState it;
  float tx, ty, ss, c1;  // inside the current cell: the sample interval [tx, ty), the cell's step and end
  bool in_cell, alive;
};

__device__ __forceinline__ void walk_begin(const RenderParams& p, RayWalk& w, vec3f o, vec3f d, float t0, float t1, float jitter, float step)
{
  w.o = o; w.d = d; w.t0 = t0; w.t1 = t1; w.jitter = jitter; w.step = step;
  const vec3f m_dir = d * p.mc_rcp;
  w.ts = {fabsf(1.0f / m_dir.x), fabsf(1.0f / m_dir.y), fabsf(1.0f / m_dir.z)};
  dda_init(w.it, o * p.mc_rcp, m_dir, t0, p.mc_dims);
  w.in_cell = false;
  w.alive = t0 < t1;
  w.tx = w.ty = w.ss = w.c1 = 0.0f;
}

// -> true: [w.tx, w.ty) is the ray's next sample interval (walk_consume moves past it); false: the ray has left the volume
__device__ __forceinline__ bool walk_next(const RenderParams& p, RayWalk& w)
{
  const vec3i grid = p.mc_dims;
  // m_dir = d / spacing has d's signs (spacings are positive)
  const vec3i stop = {w.d.x > 0.0f ? grid.x : -1, w.d.y > 0.0f ? grid.y : -1, w.d.z > 0.0f ? grid.z : -1};
  const vec3i delta = {w.d.x > 0.0f ? 1 : -1, w.d.y > 0.0f ? 1 : -1, w.d.z > 0.0f ? 1 : -1};
  for (;;) {
    const float t_closest = min3f(w.it.t_next.x, w.it.t_next.y, w.it.t_next.z);
    bool leave_cell = false;
    if (w.in_cell) {
      if (w.ty > w.tx) return true;
      leave_cell = true;   // the cell's samples are used up
    } else {
      const float c0 = fmaxf(w.t0 + w.it.next_cell_begin, w.t0), c1 = fminf(w.t0 + t_closest, w.t1);
      if (c0 >= c1) break;
      const float r = opacity_upper_bound(p, w.it.cell);
      if (fabsf(r) <= FLT_EPSILON) {
        leave_cell = true;  // an empty cell
      } else {
        w.ss = adaptive_sampling_rate(w.step, r);
        w.tx = c0;
        w.ty = fminf(c1, c0 + w.ss);
        w.c1 = c1;
        w.in_cell = true;
      }
    }
    if (leave_cell) {  // dda3's advance
      w.in_cell = false;
      if (w.it.t_next.x == t_closest) { w.it.t_next.x += w.ts.x; w.it.cell.x += delta.x; if (w.it.cell.x == stop.x) break; }
      if (w.it.t_next.y == t_closest) { w.it.t_next.y += w.ts.y; w.it.cell.y += delta.y; if (w.it.cell.y == stop.y) break; }
      if (w.it.t_next.z == t_closest) { w.it.t_next.z += w.ts.z; w.it.cell.z += delta.z; if (w.it.cell.z == stop.z) break; }
      w.it.next_cell_begin = t_closest;
    }
  }
  w.alive = false;
  return false;
}

__device__ __forceinline__ void walk_consume(RayWalk& w)
{
  w.tx = w.ty;
  w.ty = fminf(w.tx + w.ss, w.c1);
}

constexpr int kInShaderStatSlots = 64;   // statistics are summed per block and spread over this many addresses (one address serialises)

// SHADE: M_NONE (mode 6), M_GRADIENT (mode 9), M_SSH (mode 12: camera walk, then a shadow walk from the strongest sample)
template <int F, int K_IN, int W, bool GENERAL, int SHADE>
__global__ void __launch_bounds__(256) in_shader_kernel(const RenderParams p, const TileNet net, unsigned long long* __restrict__ stat_samples,
                                                        uint32_t* __restrict__ stat_hits)
{
  extern __shared__ __attribute__((aligned(16))) half_t lds[];
  __shared__ uint32_t s_stat[2 * 4 + 4];   // per wave: samples (lo, hi); then hits
  {  // the weights, once per block
    const uint4_t* src = (const uint4_t*)net.packed_mlp;
    uint4_t* dst = (uint4_t*)lds;
    for (uint32_t i = threadIdx.x; i < net.lds_halves / 8; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;   // local work index: a wave is one 64-pixel tile (map_pixel)
  const table_rsrc_t rsrc = make_table_rsrc(net.table, net.table_bytes);

  uint32_t pixel = 0;
  const bool valid = i < p.n_local && map_pixel(p, i, pixel);
  vec3f org = {0, 0, 0}, dir = {0, 0, 1};
  float alpha = 0.0f;
  vec3f color = {0, 0, 0};
  vec3f h_org = {0, 0, 0}, h_color = {0, 0, 0};
  float h_alpha = 0.0f;
  float j1 = 0.0f, j2 = 0.0f;
  uint32_t rng_state = 0;
  RayWalk w;
  w.alive = false; w.in_cell = false;
  bool hit = false;
  if (valid) {
    compute_ray(p, pixel, org, dir);
    rng_state = tea_lcg_two((uint32_t)p.frame_index, pixel, j1, j2);   // rng.get_floats(): .x jitters the camera ray
    float t0 = 0.0f, t1 = VNR_FLOAT_LARGE;
    hit = intersect_box(t0, t1, org, dir, p.bbox_lo, p.bbox_hi);
    if (hit) walk_begin(p, w, org, dir, t0, t1, j1, p.step);
  }
  unsigned long long n_samples = 0;   // wave-uniform count of live samples

  // pass 0: the camera rays to their ends.  pass 1 (M_SSH, :1102-1128): one shadow ray from the strongest sample towards the light,
  // alpha only (network_raymarching_transmittance), at twice the step, jittered by the pixel's next random number.
  // (One loop for both passes rather than a function called twice: the evaluation's buffer resource and scalar level table must stay
  // in scalar registers, which they do not across a call that is not inlined.)
  constexpr int kPasses = SHADE == M_SSH ? 2 : 1;
  float pixel_alpha = 0.0f;
  bool shaded = false;
#pragma unroll 1
  for (int pass = 0; pass < kPasses; ++pass) {
    const bool shadow = pass == 1;
    if (shadow) {
      pixel_alpha = alpha;
      alpha = 0.0f;
      w.alive = false;
      shaded = valid && hit && h_alpha > 0.0f;
      if (shaded) {
        float s0 = 0.0f, s1 = VNR_FLOAT_LARGE;
        if (intersect_box(s0, s1, h_org, p.shadow_dir, p.bbox_lo, p.bbox_hi)) {
          const float j3 = (float)((1664525u * rng_state + 1013904223u) & 0x00FFFFFFu) / (float)0x01000000;   // the next get_floats().x
          walk_begin(p, w, h_org, p.shadow_dir, s0, s1, j3, 2.0f * p.step);   // raymarching_shadow_sampling_scale = 2 (instantvnr_types.h:137)
        }
      }
    }
    for (;;) {
      const bool has = w.alive && walk_next(p, w);
      const unsigned long long live = __builtin_amdgcn_ballot_w64(has);
      if (live == 0ull) break;
      n_samples += (unsigned long long)__builtin_popcountll(live);   // samples, as the streaming path counts them (gradient shading: 4 evaluations each)
      float t = 0.0f;
      vec3f c = {0.5f, 0.5f, 0.5f};   // a lane without a sample evaluates an in-domain point and drops the result
      if (has) {
        t = (1.0f - w.jitter) * w.tx + w.jitter * w.ty;   // lerp(jitter, t0, t1), instantvnr_types.h:162-166
        c = w.o + t * w.d;
      }
      const float v = eval_tile<F, K_IN, W, GENERAL>(net, lds, rsrc, c.x, c.y, c.z);
      float fgx = 0.0f, fgy = 0.0f, fgz = 0.0f;
      vec3f stp = p.grad_step;
      if (SHADE == M_GRADIENT) {  // sampleGradient (raytracing.h:128-143): forward differences, a step that would leave [0, 1] flipped
        if (p.grad_flip) {
          if (c.x + stp.x > 1.0f - FLT_EPSILON) stp.x *= -1.0f;
          if (c.y + stp.y > 1.0f - FLT_EPSILON) stp.y *= -1.0f;
          if (c.z + stp.z > 1.0f - FLT_EPSILON) stp.z *= -1.0f;
        }
        fgx = eval_tile<F, K_IN, W, GENERAL>(net, lds, rsrc, c.x + stp.x, c.y, c.z);
        fgy = eval_tile<F, K_IN, W, GENERAL>(net, lds, rsrc, c.x, c.y + stp.y, c.z);
        fgz = eval_tile<F, K_IN, W, GENERAL>(net, lds, rsrc, c.x, c.y, c.z + stp.z);
      }
      if (has) {
        vec3f rgb; float a;
        tfn_sample(p.tfn, v, rgb, a);
        a = opacity_correction(p.step_rcp, w.ty - w.tx, a);
        if (shadow) {
          alpha += (1.0f - alpha) * a;
        } else {
          if (SHADE == M_GRADIENT) rgb = gradient_shade(p, w.d, v, fgx, fgy, fgz, stp, rgb);
          if (SHADE == M_SSH && h_alpha < (1.0f - alpha) * a) { h_org = c; h_color = rgb; h_alpha = (1.0f - alpha) * a; }
          const float tr = 1.0f - alpha;
          color.x += tr * rgb.x * a; color.y += tr * rgb.y * a; color.z += tr * rgb.z * a;
          alpha += tr * a;
        }
        if (!(alpha < VNR_NEARLY_ONE)) w.alive = false;
        walk_consume(w);
      }
    }
  }
  if (SHADE == M_SSH) {
    if (shaded) {
      const float transmittance = 1.0f - alpha, k = 0.95f;
      color.x = (1.0f - k) * color.x + k * (h_color.x * pixel_alpha * transmittance);
      color.y = (1.0f - k) * color.y + k * (h_color.y * pixel_alpha * transmittance);
      color.z = (1.0f - k) * color.z + k * (h_color.z * pixel_alpha * transmittance);
    }
    alpha = pixel_alpha;
  }
  if (valid) write_pixel(p, {color.x, color.y, color.z, alpha}, pixel);

  // statistics: per block, then one add per block on one of kInShaderStatSlots addresses
  const uint32_t hits = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
  if (lane == 0) { s_stat[2 * wave] = (uint32_t)n_samples; s_stat[2 * wave + 1] = (uint32_t)(n_samples >> 32); s_stat[8 + wave] = hits; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0; uint32_t h = 0;
    for (int k = 0; k < 4; ++k) { s += (unsigned long long)s_stat[2 * k] | ((unsigned long long)s_stat[2 * k + 1] << 32); h += s_stat[8 + k]; }
    const uint32_t slot = blockIdx.x % (uint32_t)kInShaderStatSlots;
    if (s) atomicAdd(&stat_samples[slot], s);
    if (h) atomicAdd(&stat_hits[slot], h);
  }
}

// ------------------------------------------------------------------------------------------------ path tracing in shader (mode 15)
// network_path_tracing_traceray (core/renderer/method_pathtracing.cu:968-1025): the delta-tracking estimator with the network sampled
// inside the loop.  Per lane it is the streaming path tracer's own chain of decisions (pt_take_sample -> evaluate -> pt_shade,
// with p.pt_reset_interval = 1: DESIGN.md 7), so a path consumes the same random numbers and the frame is the streaming path's
// frame bit for bit; what changes is that the 60-odd iterations of a frame are trips of ONE launch instead of 3 launches each.
constexpr int kInShaderPathTracing = 4;   // render_in_shader's `shade` argument beside M_NONE / M_GRADIENT / M_SSH

template <int F, int K_IN, int W, bool GENERAL>
__global__ void __launch_bounds__(256) in_shader_pt_kernel(const RenderParams p, const TileNet net, unsigned long long* __restrict__ stat_samples,
                                                           uint32_t* __restrict__ stat_hits)
{
  extern __shared__ __attribute__((aligned(16))) half_t lds[];
  __shared__ uint32_t s_stat[2 * 4 + 4];
  {
    const uint4_t* src = (const uint4_t*)net.packed_mlp;
    uint4_t* dst = (uint4_t*)lds;
    for (uint32_t i = threadIdx.x; i < net.lds_halves / 8; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t i = blockIdx.x * 256u + threadIdx.x;
  const table_rsrc_t rsrc = make_table_rsrc(net.table, net.table_bytes);

  uint32_t pixel = 0;
  const bool valid = i < p.n_local && map_pixel(p, i, pixel);
  PtRay r;
  r.pidx = pixel; r.shadow = false;
  r.org = {0, 0, 0}; r.dir = {0, 0, 1};
  r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;
  r.scatter_index = 0; r.sample_coord = {0.5f, 0.5f, 0.5f}; r.majorant = 0.0f;
  r.L = {0, 0, 0}; r.throughput = {1, 1, 1};
  r.rng = 0;
  r.it.t_next = {0, 0, 0}; r.it.cell = {0, 0, 0}; r.it.next_cell_begin = 0.0f;
  bool alive = false, hit = false;
  if (valid) {  // iterative_raygen_kernel (:679-748)
    compute_ray(p, pixel, r.org, r.dir);
    {  // RandomTEA(frame_index, pidx): 16 TEA rounds seed the LCG
      uint32_t v0 = (uint32_t)p.frame_index, v1 = pixel, s0 = 0;
#pragma unroll
      for (int n = 0; n < 16; ++n) {
        s0 += 0x9e3779b9u;
        v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
        v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
      }
      r.rng = v0;
    }
    hit = intersect_box(r.tnear, r.tfar, r.org, r.dir, p.bbox_lo, p.bbox_hi);
    if (hit) {
      dda_init(r.it, r.org * p.mc_rcp, r.dir * p.mc_rcp, r.tnear, p.mc_dims);
      alive = pt_take_sample(p, r);
    }
  }
  unsigned long long n_samples = 0;
  for (;;) {
    const unsigned long long live = __builtin_amdgcn_ballot_w64(alive);
    if (live == 0ull) break;
    n_samples += (unsigned long long)__builtin_popcountll(live);
    const vec3f c = alive ? r.sample_coord : vec3f{0.5f, 0.5f, 0.5f};
    const float v = eval_tile<F, K_IN, W, GENERAL>(net, lds, rsrc, c.x, c.y, c.z);
    if (alive) {  // iterative_shade_kernel (:750-768); the ray's interval is recomputed on every trip, as its load does (:126-129)
      r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;
      intersect_box(r.tnear, r.tfar, r.org, r.dir, p.bbox_lo, p.bbox_hi);
      alive = pt_shade(p, p.tfn, r, v) && pt_take_sample(p, r);
    }
  }
  if (valid) write_pixel(p, {r.L.x, r.L.y, r.L.z, 1.0f}, pixel);

  const uint32_t hits = (uint32_t)__builtin_popcountll(__builtin_amdgcn_ballot_w64(hit));
  if (lane == 0) { s_stat[2 * wave] = (uint32_t)n_samples; s_stat[2 * wave + 1] = (uint32_t)(n_samples >> 32); s_stat[8 + wave] = hits; }
  __syncthreads();
  if (threadIdx.x == 0) {
    unsigned long long s = 0; uint32_t h = 0;
    for (int k = 0; k < 4; ++k) { s += (unsigned long long)s_stat[2 * k] | ((unsigned long long)s_stat[2 * k + 1] << 32); h += s_stat[8 + k]; }
    const uint32_t slot = blockIdx.x % (uint32_t)kInShaderStatSlots;
    if (s) atomicAdd(&stat_samples[slot], s);
    if (h) atomicAdd(&stat_hits[slot], h);
  }
}

// ---- one translation unit per width and kind (in_shader_w*.hip) ----------------------------------------------------------------
// model shapes with an in-shader instance (F, padded input width); the others take the streaming path
#define VNR_IN_SHADER_SHAPES(X) X(1, 16) X(1, 32) X(2, 16) X(2, 32) X(2, 64) X(4, 32) X(4, 64) X(8, 64) X(8, 128)

struct InShaderLaunch {
  uint32_t blocks;
  size_t shmem;
  hipStream_t stream;
  unsigned long long* stat_samples;
  uint32_t* stat_hits;
};

template <int W, bool GENERAL>
static bool launch_in_shader_instance(const RenderParams& p, const TileNet& net, int shade, const InShaderLaunch& l)
{
  bool launched = false;
  auto launch = [&](auto kernel) {
    // (the kernel also has 48 bytes of static LDS: the dynamic part cannot be the whole 160 KB)
    if (l.shmem > 48 * 1024) VNR_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)l.shmem));
    kernel<<<l.blocks, 256, l.shmem, l.stream>>>(p, net, l.stat_samples, l.stat_hits);
    launched = true;
  };
#define X(f, k)                                                                                \
  if (!launched && net.n_features == f && net.in_width == k) {                                 \
    if (shade == kInShaderPathTracing) launch(in_shader_pt_kernel<f, k, W, GENERAL>);         \
    else if (shade == M_GRADIENT) launch(in_shader_kernel<f, k, W, GENERAL, M_GRADIENT>);      \
    else if (shade == M_SSH) launch(in_shader_kernel<f, k, W, GENERAL, M_SSH>);                \
    else launch(in_shader_kernel<f, k, W, GENERAL, M_NONE>);                                   \
  }
  VNR_IN_SHADER_SHAPES(X)
#undef X
  return launched;
}

#define VNR_DEFINE_IN_SHADER_WIDTH(W, GENERAL, SUFFIX)                                                                          \
  bool launch_in_shader_w##W##SUFFIX(const RenderParams& p, const TileNet& net, int shade, const InShaderLaunch& l)             \
  {                                                                                                                             \
    return launch_in_shader_instance<W, GENERAL>(p, net, shade, l);                                                             \
  }

}  // namespace vnr
