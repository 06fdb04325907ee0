// march_device.h — device-side pieces every ray-marching kernel shares: the launch parameters, ray generation, the box test, the
// macrocell DDA, the adaptive step, shading, the pixel write.  Split out of render.hip in round 5 so that the in-shader kernels
// (in_shader.h) can be compiled in translation units of their own, one per FullyFusedMLP width and kernel kind (in_shader_w*.hip).
//
// Reference: core/renderer/method_raymarching.cu:544-685, core/renderer/dda.h:20-138, core/renderer/raytracing.h:9-36,147-246.
#pragma once

#include <cfloat>

#include "renderer.h"
#include "sampling_device.h"

namespace vnr {

#define VNR_FLOAT_LARGE 1e20f
#define VNR_NEARLY_ONE 0.9999f

struct RenderParams {
  vec4f* frame;
  vec4f* accumulation;
  int width, height, frame_index;
  uint32_t pixel_lo, pixel_hi;
  uint32_t il_parts, il_part, n_local;   // tile-row interleave across ranks; n_local = local index count
  uint32_t out_parts;                    // > 1: frame / accumulation hold only the tile rows r with r % out_parts == this rank, packed (Renderer::set_distributed)
  uint32_t tiles_per_row, tile_row0;     // 64-pixel tiles per band of 8 scanlines: rays of a wave are an image patch, not a scanline
  uint32_t tile_w_log2;                  // tile shape: 2^tile_w_log2 x 2^(6 - tile_w_log2) pixels (8x8, 16x4, 32x2 or 64x1)
  float bin_depth_rcp;                   // 1 / depth of one sample-sorting bin (world units)
  uint32_t tfn_in_lds;                   // the TFN tables fit in the march kernel's LDS
  uint32_t no_ranks;                     // march_kernel: a sample's rank inside its depth bin is not kept in LDS (2 bytes per sample) but claimed again from the bin's counter
  uint32_t debug_flags;                  // diagnostics builds only (-DVNR_DIAG, VNR_AMD_DEBUG_FLAGS; read through dbg()): timing ablations that render garbage: 1 no compose, 2 no TFN, 4 no sort, 8 no DDA walk, 16 no sample records, 32 separate colour / opacity lookups
  vec3f cam_pos, cam_dir, cam_hor, cam_ver;
  affine3f wto;
  vec3i vol_dims;
  const float* volume;
  vec3f bbox_lo, bbox_hi;
  float step, step_rcp;
  vec3i mc_dims;
  vec3f mc_rcp;
  const float* mc_max_opacity;
  DeviceTfn tfn;
  int n_iters;
  // gradient shading (rendering modes 7 / 8)
  affine3f otw;          // object -> world (params.transform)
  vec3f grad_step;       // 1 / dims (object.cpp:305)
  vec3f light_dir;       // LaunchParams::light_directional_dir after the flip of renderer.cpp:98-101
  uint32_t slot_cap;     // sample slots of this half's result arena (value/dt pairs first, then the gradient samples)
  uint32_t shading_mode; // 0 NO_SHADING, 1 GRADIENT_SHADING, 2 SINGLE_SHADE_HEURISTIC (the streaming kernels are templated on it; the monolithic one branches)
  // SINGLE_SHADE_HEURISTIC (modes 10 / 11): per-pixel hand-over from the camera pass to the shadow pass
  // (final_highest_*, shading_color, jitter_ssh; method_raymarching.cu:88-92) and the shadow rays' common direction
  vec3f* px_org;
  vec3f* px_color;
  float* px_alpha;
  vec4f* px_shading;
  float* px_jitter;
  vec3f shadow_dir;      // xfmVector(wto, normalize(light_directional_dir)) (:649)
  float density_scale;   // DeviceVolume::density_scale (path tracing, rendering mode 14)
  uint32_t ssh_third_draw;   // rendering mode 12: the shadow ray's jitter is the pixel's third random number
  uint32_t grad_flip;        // rendering mode 9: forward differences flip at the volume's far faces (sampleGradient)
  uint32_t pt_reset_interval;  // rendering mode 15: the in-shader estimator resets tnear / tfar before a bounce (:999-1001)
};

// The timing ablations exist in diagnostics builds only (make EXTRA=-DVNR_DIAG, tools/ab_build.sh): in the library that ships, dbg() is the
// constant 0, the branches fold away and no environment variable can make a frame wrong.
#if defined(VNR_DIAG)
__host__ __device__ __forceinline__ uint32_t dbg(const RenderParams& p) { return p.debug_flags; }
#else
__host__ __device__ __forceinline__ constexpr uint32_t dbg(const RenderParams&) { return 0u; }
#endif

// streaming kernel modes (ShadingMode, method_raymarching.cu:51-56)
enum { M_NONE = 0, M_GRADIENT = 1, M_SSH = 2, M_SHADOW = 3 };

// Result arena of one half and one parity, in floats: [slot_cap][2] = {value, t1 - t0} per sample slot, then (gradient
// shading only) [slot_cap][4] = {f(c + gx), f(c + gy), f(c + gz), unused}.  A queue record's 4th word is the absolute float
// index its result goes to, so the inference kernel and the ground-truth sampler need not know about shading modes.
__device__ __forceinline__ uint32_t arena_value_index(uint32_t slot) { return 2u * slot; }
__device__ __forceinline__ uint32_t arena_grad_index(uint32_t slot_cap, uint32_t slot) { return 2u * slot_cap + 4u * slot; }

constexpr int kDepthBins = 64;

// C_*: the device counters of a ray part (pack_rays.h)

// ------------------------------------------------------------------------------------------------ helpers
__device__ __forceinline__ float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
__device__ __forceinline__ float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

// raytracing.h:9-36
__device__ __forceinline__ bool intersect_box(float& t0, float& t1, vec3f org, vec3f dir, vec3f lower, vec3f upper)
{
  const bool sx = fabsf(dir.x) <= FLT_MIN, sy = fabsf(dir.y) <= FLT_MIN, sz = fabsf(dir.z) <= FLT_MIN;
  const vec3f rcp = {1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z};
  const vec3f lo = {sx ? VNR_FLOAT_LARGE : (lower.x - org.x) * rcp.x, sy ? VNR_FLOAT_LARGE : (lower.y - org.y) * rcp.y,
                    sz ? VNR_FLOAT_LARGE : (lower.z - org.z) * rcp.z};
  const vec3f hi = {sx ? -VNR_FLOAT_LARGE : (upper.x - org.x) * rcp.x, sy ? -VNR_FLOAT_LARGE : (upper.y - org.y) * rcp.y,
                    sz ? -VNR_FLOAT_LARGE : (upper.z - org.z) * rcp.z};
  t0 = fmaxf(t0, max3f(fminf(lo.x, hi.x), fminf(lo.y, hi.y), fminf(lo.z, hi.z)));
  t1 = fminf(t1, min3f(fmaxf(lo.x, hi.x), fmaxf(lo.y, hi.y), fmaxf(lo.z, hi.z)));
  return t1 > t0;
}

// method_raymarching.cu:658-685
__device__ __forceinline__ void compute_ray(const RenderParams& p, uint32_t pixel, vec3f& org, vec3f& dir)
{
  const uint32_t ix = pixel % (uint32_t)p.width, iy = pixel / (uint32_t)p.width;
  const float sx = ((float)ix + 0.5f) / (float)p.width, sy = ((float)iy + 0.5f) / (float)p.height;
  org = xfm_point(p.wto, p.cam_pos);
  const vec3f d = (p.cam_dir + (sx - 0.5f) * p.cam_hor) + (sy - 0.5f) * p.cam_ver;
  dir = xfm_vector(p.wto, normalize(d));
}

// local work index -> global pixel index of this rank's share of the image.  64 consecutive indices are one
// TW x TH pixel tile (TW TH = 64, TH <= 8); the 8 / TH tiles stacked in one band of 8 scanlines follow each other, then the
// next column of the band.  Bands are dealt round-robin to the ranks (il_parts, il_part); a pixel range restricts further.
__device__ __forceinline__ bool map_pixel(const RenderParams& p, uint32_t i, uint32_t& pixel)
{
  const uint32_t tile = i >> 6, l = i & 63u;
  const uint32_t tr_local = tile / p.tiles_per_row, t = tile - tr_local * p.tiles_per_row;
  const uint32_t twl = p.tile_w_log2, sub_log2 = twl - 3u;   // 8 / TH = TW / 8 stacked tiles per band column
  const uint32_t tc = t >> sub_log2, ts = t & ((1u << sub_log2) - 1u);
  const uint32_t x = (tc << twl) + (l & ((1u << twl) - 1u));
  const uint32_t y = (p.tile_row0 + tr_local * p.il_parts + p.il_part) * 8u + (ts << (6u - twl)) + (l >> twl);
  pixel = y * (uint32_t)p.width + x;
  return x < (uint32_t)p.width && y < (uint32_t)p.height && pixel >= p.pixel_lo && pixel < p.pixel_hi;
}

// gdt::LCG<16> (EXTERNAL; instantvnr_types.h:155)
__device__ __forceinline__ float tea_lcg_first(uint32_t v0, uint32_t v1)
{
  uint32_t s0 = 0;
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  const uint32_t state = 1664525u * v0 + 1013904223u;
  return (float)(state & 0x00FFFFFFu) / (float)0x01000000;
}

// both draws of rng.get_floats() (EXTERNAL OVR addition to gdt::LCG: two successive floats)
__device__ __forceinline__ uint32_t tea_lcg_two(uint32_t v0, uint32_t v1, float& a, float& b)
{
  uint32_t s0 = 0;
#pragma unroll
  for (int n = 0; n < 16; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  uint32_t state = 1664525u * v0 + 1013904223u;
  a = (float)(state & 0x00FFFFFFu) / (float)0x01000000;
  state = 1664525u * state + 1013904223u;
  b = (float)(state & 0x00FFFFFFu) / (float)0x01000000;
  return state;
}

// raytracing.h:188-194 / :166-170 / :196-207
__device__ __forceinline__ float adaptive_sampling_rate(float base_step, float max_opacity)
{
  const float scale = 15.0f * base_step;
  const float r = fabsf(clampf(max_opacity, 0.1f, 1.0f) - 1.0f);
  return fmaxf(base_step + scale * (r * r), base_step);
}
__device__ __forceinline__ float opacity_correction(float step_rcp, float distance, float opacity)
{
#if defined(VNR_FAST_POW)   // experiment (tools/ab_build.sh fastpow -DVNR_FAST_POW): v_log_f32 / v_exp_f32 instead of the ~80 instructions of powf
  return 1.0f - __builtin_amdgcn_exp2f(step_rcp * distance * __builtin_amdgcn_logf(1.0f - opacity));
#else
  return 1.0f - __builtin_powf(1.0f - opacity, step_rcp * distance);
#endif
}
__device__ __forceinline__ void write_pixel(const RenderParams& p, vec4f rgba, uint32_t pixel)
{
  if (p.out_parts > 1u) {  // compact share: tile row b of the image is tile row b / out_parts of the share
    const uint32_t y = pixel / (uint32_t)p.width, x = pixel - y * (uint32_t)p.width;
    pixel = (((y >> 3) / p.out_parts) * 8u + (y & 7u)) * (uint32_t)p.width + x;
  }
  if (p.frame_index != 1) {
    const vec4f a = p.accumulation[pixel];
    rgba = {a.x + rgba.x, a.y + rgba.y, a.z + rgba.z, a.w + rgba.w};
  }
  p.accumulation[pixel] = rgba;
  const float f = (float)p.frame_index;
  p.frame[pixel] = {rgba.x / f, rgba.y / f, rgba.z / f, rgba.w / f};
}

// ------------------------------------------------------------------------------------------------ gradient shading (modes 7 / 8)
// shade_simple_light (raytracing.h:214-222)
__device__ __forceinline__ vec3f shade_simple_light(vec3f ray_dir, vec3f normal, vec3f albedo)
{
  if (dot(normal, normal) > 1.0e-6f) {
    const vec3f n = normalize(normal);
    const float c = 0.2f + 0.8f * fabsf(-dot(ray_dir, n));
    return c * albedo;
  }
  return {0, 0, 0};
}

// shade_scivis_light (raytracing.h:224-246) with mat_gradient_shading {.6, .9, .4, 40} and light_directional_rgb = 1
// (instantvnr_types.h:142,147); the reference's light_ambient argument is unused there.  World-space vectors.
__device__ __forceinline__ vec3f shade_scivis_light(vec3f ray_dir, vec3f normal, vec3f albedo, vec3f light_dir)
{
  const float m_ambient = 0.6f, m_diffuse = 0.9f, m_specular = 0.4f, m_shininess = 40.0f;
  vec3f color = {0, 0, 0};
  if (dot(normal, normal) > 1.0e-6f) {
    const vec3f L = normalize(light_dir);
    const vec3f N = normalize(normal);
    const vec3f V = {-ray_dir.x, -ray_dir.y, -ray_dir.z};
    color = color + m_ambient * albedo;
    const float cosNL = fmaxf(dot(N, L), 0.0f);
    if (cosNL > 0.0f) {
      color = color + (m_diffuse * cosNL) * albedo;
      const vec3f H = normalize(L + V);
      const float cosNH = fmaxf(dot(N, H), 0.0f);
      const float sp = m_specular * powf(cosNH, m_shininess);
      color = color + vec3f{sp, sp, sp};
    }
  }
  const vec3f shading2 = shade_simple_light(ray_dir, normal, albedo);
  return 0.5f * shading2 + 0.5f * color;  // lerp(0.5, shading2, color)
}

// One shaded sample (method_raymarching.cu:773-788 / :440-454): object-space normal from forward differences divided by
// `step`, to world space with xfmNormal (EXTERNAL gdt: transposed inverse of the linear part = rows of wto's columns),
// shaded, then lerp(scivis_shading_scale = 0.95, albedo, shaded) (instantvnr_types.h:140).
__device__ __forceinline__ vec3f gradient_shade(const RenderParams& p, vec3f ray_dir_obj, float f, float fgx, float fgy, float fgz,
                                                vec3f step, vec3f albedo)
{
  const vec3f No = {-((fgx - f) / step.x), -((fgy - f) / step.y), -((fgz - f) / step.z)};
  const vec3f Nw = {dot(p.wto.vx, No), dot(p.wto.vy, No), dot(p.wto.vz, No)};
  const vec3f dir_w = xfm_vector(p.otw, ray_dir_obj);
  const vec3f shaded = shade_scivis_light(dir_w, Nw, albedo, p.light_dir);
  const float k = 0.95f;
  return (1.0f - k) * albedo + k * shaded;
}

// ------------------------------------------------------------------------------------------------ DDA (dda.h)
struct DDAState {
  vec3f t_next;
  vec3i cell;
  float next_cell_begin;
};

// dda.h:26-46
__device__ __forceinline__ void dda_init(DDAState& it, vec3f org, vec3f dir, float t_min, vec3i grid)
{
  const vec3f oiv = org + t_min * dir;
  const vec3f fc = {fmaxf(0.0f, fminf((float)grid.x - 1.0f, floorf(oiv.x))), fmaxf(0.0f, fminf((float)grid.y - 1.0f, floorf(oiv.y))),
                    fmaxf(0.0f, fminf((float)grid.z - 1.0f, floorf(oiv.z)))};
  const vec3f fe = {dir.x > 0.0f ? fc.x + 1.0f : fc.x, dir.y > 0.0f ? fc.y + 1.0f : fc.y, dir.z > 0.0f ? fc.z + 1.0f : fc.z};
  const vec3f ts = {fabsf(1.0f / dir.x), fabsf(1.0f / dir.y), fabsf(1.0f / dir.z)};
  it.t_next = {dir.x == 0.0f ? VNR_FLOAT_LARGE : fabsf(fe.x - oiv.x) * ts.x, dir.y == 0.0f ? VNR_FLOAT_LARGE : fabsf(fe.y - oiv.y) * ts.y,
               dir.z == 0.0f ? VNR_FLOAT_LARGE : fabsf(fe.z - oiv.z) * ts.z};
  it.cell = {(int)fc.x, (int)fc.y, (int)fc.z};
  it.next_cell_begin = 0.0f;
}

// dda.h:48-122; fn(cell, t0, t1) -> bool
template <typename F>
__device__ __forceinline__ bool dda_next(DDAState& it, vec3f dir, float t_min, float t_max, vec3i grid, F&& fn)
{
  const vec3i stop = {dir.x > 0.0f ? grid.x : -1, dir.y > 0.0f ? grid.y : -1, dir.z > 0.0f ? grid.z : -1};
  if (it.cell.x == stop.x) return false;
  if (it.cell.y == stop.y) return false;
  if (it.cell.z == stop.z) return false;
  const vec3f ts = {fabsf(1.0f / dir.x), fabsf(1.0f / dir.y), fabsf(1.0f / dir.z)};
  const vec3i delta = {dir.x > 0.0f ? 1 : -1, dir.y > 0.0f ? 1 : -1, dir.z > 0.0f ? 1 : -1};
  const float t_closest = min3f(it.t_next.x, it.t_next.y, it.t_next.z);
  const float cell_t0 = fmaxf(t_min + it.next_cell_begin, t_min);
  const float cell_t1 = fminf(t_min + t_closest, t_max);
  if (cell_t0 >= cell_t1) return false;
  const bool go = fn(it.cell, cell_t0, cell_t1);
  if (go || fmaxf(t_min + it.next_cell_begin, t_min) >= cell_t1) {
    if (it.t_next.x == t_closest) { it.t_next.x += ts.x; it.cell.x += delta.x; if (it.cell.x == stop.x) return false; }
    if (it.t_next.y == t_closest) { it.t_next.y += ts.y; it.cell.y += delta.y; if (it.cell.y == stop.y) return false; }
    if (it.t_next.z == t_closest) { it.t_next.z += ts.z; it.cell.z += delta.z; if (it.cell.z == stop.z) return false; }
    it.next_cell_begin = t_closest;
  }
  return go;
}

// dda.h:124-137
__device__ __forceinline__ bool dda_resumable(const DDAState& it, vec3f dir, float t_min, float t_max, vec3i grid)
{
  const vec3i stop = {dir.x > 0.0f ? grid.x : -1, dir.y > 0.0f ? grid.y : -1, dir.z > 0.0f ? grid.z : -1};
  if (it.cell.x == stop.x) return false;
  if (it.cell.y == stop.y) return false;
  if (it.cell.z == stop.z) return false;
  const float t_closest = min3f(it.t_next.x, it.t_next.y, it.t_next.z);
  const float cell_t0 = fmaxf(t_min + it.next_cell_begin, t_min);
  const float cell_t1 = fminf(t_min + t_closest, t_max);
  return cell_t0 < cell_t1;
}

__device__ __forceinline__ float opacity_upper_bound(const RenderParams& p, vec3i cell)
{
#if defined(VNR_WALK_NOLOAD)   // experiment: what the walk costs without its one load per macrocell (frames are garbage)
  return 0.05f + 1e-9f * (float)cell.x;
#endif
  const uint32_t idx = cell.x + cell.y * (uint32_t)p.mc_dims.x + cell.z * (uint32_t)p.mc_dims.x * (uint32_t)p.mc_dims.y;
  return p.mc_max_opacity[idx];
}

// RayMarchingIter::exec (method_raymarching.cu:555-600); body(t0, t1) -> bool.
// dda_next with the cell callback written out as one loop, for the latency of a single wave (a small frame share runs one
// wave per SIMD, and the length of the per-iteration kernel chain is what bounds it):
//  * the opacity bound of the cell the walk enters NEXT is fetched while the current cell is processed: which cell comes
//    next depends only on the DDA state, not on what the current cell holds (an empty run of cells is otherwise a chain of
//    dependent L2 round trips).  The fetch is unconditional (a walk that leaves the grid re-reads its current cell) so that
//    the compiler's wait-count bookkeeping sees one pending load on every path;
//  * the advance is written with selects instead of dda_next's three early returns.  A walk that leaves the grid through x
//    therefore also advances y / z and next_cell_begin where dda_next returns first; nothing reads that state again
//    (dda_resumable and this function test the cell against `stop` before anything else).
// Inside the grid the state (cell, t_next, next_cell_begin) goes through exactly dda_next's operations.
template <typename B>
__device__ __forceinline__ void iter_exec(const RenderParams& p, DDAState& it, vec3f dir, float t_min, float t_max, float step, B&& body)
{
  const vec3i grid = p.mc_dims;
  const vec3i stop = {dir.x > 0.0f ? grid.x : -1, dir.y > 0.0f ? grid.y : -1, dir.z > 0.0f ? grid.z : -1};
  if (it.cell.x == stop.x || it.cell.y == stop.y || it.cell.z == stop.z) return;
  const vec3f ts = {fabsf(1.0f / dir.x), fabsf(1.0f / dir.y), fabsf(1.0f / dir.z)};
  const vec3i delta = {dir.x > 0.0f ? 1 : -1, dir.y > 0.0f ? 1 : -1, dir.z > 0.0f ? 1 : -1};
  float r = opacity_upper_bound(p, it.cell);
  __builtin_amdgcn_s_waitcnt(0x0f70);  // vmcnt(0): inside the loop only the look-ahead load is in flight
  bool more = true;
  while (more) {
    const float t_closest = min3f(it.t_next.x, it.t_next.y, it.t_next.z);
    const float cell_t0 = fmaxf(t_min + it.next_cell_begin, t_min);
    const float cell_t1 = fminf(t_min + t_closest, t_max);
    if (cell_t0 >= cell_t1) break;
    // the cell after this one
    const bool bx = it.t_next.x == t_closest, by = it.t_next.y == t_closest, bz = it.t_next.z == t_closest;
    const vec3i nc = {it.cell.x + (bx ? delta.x : 0), it.cell.y + (by ? delta.y : 0), it.cell.z + (bz ? delta.z : 0)};
    const bool inside = nc.x != stop.x && nc.y != stop.y && nc.z != stop.z;
    const float r_next = opacity_upper_bound(p, inside ? nc : it.cell);
    // the cell callback of RayMarchingIter::exec
    bool go = true;
    if (!(fabsf(r) <= FLT_EPSILON)) {
      const float ss = adaptive_sampling_rate(step, r);
      float tx = cell_t0, ty = fminf(cell_t1, cell_t0 + ss);
      // (one exit: `while (ty > tx) { ...; if (!body(tx, ty)) { go = false; break; } ... }` written so that the loop's divergent lanes
      // rejoin in one place: 20 instructions and one branch per sample instead of 30 and three; what is computed once more after a full
      // batch, tx and ty, is not read again.  The frame did not notice: 3.65-3.72 against 3.67-3.68 ms, n = 3)
      bool run = ty > tx;
      while (run) {
        it.next_cell_begin = ty - t_min;
        go = body(tx, ty);
        tx = ty;
        ty = fminf(tx + ss, cell_t1);
        run = go && ty > tx;
      }
    }
    const bool adv = go || fmaxf(t_min + it.next_cell_begin, t_min) >= cell_t1;
    it.t_next.x = (adv && bx) ? it.t_next.x + ts.x : it.t_next.x;
    it.t_next.y = (adv && by) ? it.t_next.y + ts.y : it.t_next.y;
    it.t_next.z = (adv && bz) ? it.t_next.z + ts.z : it.t_next.z;
    it.cell.x = adv ? nc.x : it.cell.x;
    it.cell.y = adv ? nc.y : it.cell.y;
    it.cell.z = adv ? nc.z : it.cell.z;
    it.next_cell_begin = adv ? t_closest : it.next_cell_begin;
    more = go && inside;
    r = r_next;
  }
}

// depth bin of a sample inside its 64-ray group (gather-order counting sort of march_kernel)
__device__ __forceinline__ uint32_t depth_bin(const RenderParams& p, float t, float front)
{
  return (dbg(p) & 4u) ? 0u : min((uint32_t)kDepthBins - 1u, (uint32_t)fmaxf((t - front) * p.bin_depth_rcp, 0.0f));
}

}  // namespace vnr
