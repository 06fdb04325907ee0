// render.hip — sample-streaming ray marcher (rendering mode 5) and the monolithic ground-truth marcher
// (mode 4 on a dense volume) for gfx950, plus the Renderer host object.
//
// Reference: core/renderer/method_raymarching.cu:187-261,544-973 (streaming), :263-308,401-536 (monolithic),
// core/renderer/dda.h:20-138, core/renderer/raytracing.h:9-36,147-207, renderer.cpp:59-140.
//
// MI355X design (per-ray arithmetic identical to the reference, structure is not):
//  * reference iteration = intersect kernel (DDA walk, writes N_ITERS slots per ray) -> inference of ALL
//    N_ITERS x R slots -> compose kernel (walks the DDA a second time) -> 4-byte D2H + stream sync.
//  * here one fused `march` kernel per iteration composes the previous batch of a ray and immediately emits its
//    next batch (one DDA walk per iteration, ray state stays in registers in between), rays are compacted with
//    wave64 ballot + popcount, and SAMPLES are compacted with a wave prefix sum, so the network only ever sees
//    live samples (the reference infers stale slots).  Sample (t0,t1) pairs are staged in LDS while a lane walks.
//  * the sample count never visits the host: the fused inference kernel reads it from device memory; the host
//    launches the iteration count of the previous frame speculatively and only then looks at a pinned counter.
#include "renderer.h"

#include <algorithm>
#include <cfloat>
#include <cstdlib>

#include "dist.h"
#include "infer_tile.h"
#include "pack_rays.h"
#include "sampling_device.h"
#include "march_device.h"
#include "pt_device.h"

namespace vnr {

// In-kernel stamps of the march kernel's phases (diagnostic builds only: tools/ab_build.sh <tag> -DVNR_MARCH_STAMPS; the guide's
// "In-kernel stamps"): cycles per phase summed over the waves of every march_kernel<false> launch, read by tools/march_stamps.py
#if defined(VNR_MARCH_STAMPS)
__device__ unsigned long long g_march_stamps[16];
// one record per wave-trip of the walk kernels' launch `it == 1` (plain stores: atomics on a few addresses from every wave of a launch
// would themselves be what is measured): {s_memrealtime at the trip's start, at its end (100 MHz, one clock for the device), then
// s_memtime differences of the trip's phases}
constexpr uint32_t kWaveRecs = 65536;
__device__ unsigned long long g_wave_rec[kWaveRecs][8];
#define VNR_REALTIME(var) unsigned long long var; asm volatile("s_memrealtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define VNR_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var) :: "memory")
#define VNR_STAMP_ADD(slot, a, b) do { if (!FIRST && (threadIdx.x & 63u) == 0) atomicAdd(&g_march_stamps[slot], (b) - (a)); } while (0)
#else
#define VNR_STAMP(var)
#define VNR_STAMP_ADD(slot, a, b)
#endif

// ------------------------------------------------------------------------------------------------ streaming march kernel
// FIRST: thread = pixel of an 8x8 pixel tile (raygen, method_raymarching.cu:840-875) and emits the first batch.
// !FIRST: thread = alive ray: compose the batch inferred last iteration (:732-838), then emit the next (:687-730).
//
// Two views of one iteration's samples:
//  * result slots (values, dts): sample j of the ray in lane l of group g is slot (g n_iters + j) 64 + l, so compose needs only (base,
//    count) and a wave's loads of "sample j of my ray" are neighbours (round 2; ray-major slots before);
//  * gather order (16-byte queue records {x, y, z, result slot}, compacted over [0, n_samples)): inside the 64-ray group of a wave the samples are counting-sorted by DEPTH BIN
//    (bin = (t - t_group_front) / bin_depth, LDS-atomic histogram + wave scan).  One bin is a thin slab of an 8x8-pixel
//    frustum, i.e. a compact brick of the volume, whatever the per-ray sample index is.  The fused inference kernel
//    reads coords in this order (coherent hash-grid gathers: measured ~2x faster than ray-major order once rays have
//    drifted apart in depth) and scatters its result to values[dest[i]].
// GRAD (gradient shading, rendering mode 8): every sample puts FOUR records into the queue, itself and three forward
// offsets of grad_step (method_raymarching.cu:719-726), and compose shades with the resulting normal (:773-788).
// vd_in / vd_out are the result arenas of the previous / this iteration (layout above RenderParams' helpers).
// MODE (M_*): M_SSH is the camera pass of the single-shade heuristic (rendering mode 11): like M_NONE, but a ray remembers the sample
// that contributed most and its result goes to the per-pixel hand-over arrays instead of the frame; M_SHADOW is the second
// pass, one ray per pixel from that sample towards the light, alpha only, which finally shades and writes the pixel
// (method_raymarching.cu:789-833, 877-900, 960-973).
template <bool FIRST, int MODE>
#if defined(VNR_MARCH_WAVES_PER_EU)   // experiment (tools/ab_build.sh m96 -DVNR_MARCH_WAVES_PER_EU=5): a march wave of 96 registers fits beside the evaluation kernel's four
__attribute__((amdgpu_waves_per_eu(VNR_MARCH_WAVES_PER_EU, VNR_MARCH_WAVES_PER_EU)))   // waves of 104 on a SIMD; 232 bytes of scratch per lane; the frame 3.62 -> 3.93 ms
#endif
__global__ void __launch_bounds__(256) march_kernel(const RenderParams p, const RayList cur, const RayList nxt,
                                                    const vec2f* __restrict__ vd_in, vec4f* __restrict__ queue,
                                                    vec2f* __restrict__ vd_out, uint32_t* __restrict__ counters,
                                                    uint32_t* __restrict__ ray_counts, int parity, const SshLists ssh_lists,
                                                    uint32_t* __restrict__ tail_flag)
{
  extern __shared__ float s_t[];  // [n_iters][256] x {t0, t1}, histogram[256], claims[16], [n_iters][256] ranks (u16), then the transfer function tables
  float* s_t0 = s_t;
  float* s_t1 = s_t + (size_t)p.n_iters * 256;
  uint32_t* s_hist = (uint32_t*)(s_t + (size_t)2 * p.n_iters * 256);
  uint32_t* s_claim = s_hist + 256;             // [2][8]
  uint16_t* s_rk = (uint16_t*)(s_claim + 16);   // rank of a sample inside its depth bin (< 64 n_iters); the bin is recomputed
  // the TFN tables are read 4x per composed sample: keep them in LDS (no TA traffic) when they fit
  DeviceTfn tfn = p.tfn;
  constexpr bool GRAD = MODE == M_GRADIENT;
  tfn_lds_colors_t lds_colors = nullptr;
  tfn_lds_alphas_t lds_alphas = nullptr;
  bool tfn_merged = false;
  if (!FIRST && p.tfn_in_lds) {
    vec4f* s_colors = (vec4f*)(s_rk + (p.no_ranks ? 0 : (size_t)p.n_iters * 256));
    float* s_alphas = (float*)(s_colors + p.tfn.n_colors);
    tfn_merged = tfn_tables_to_lds(p.tfn, s_colors, s_alphas, !(dbg(p) & 32u));
    __syncthreads();
    lds_colors = (tfn_lds_colors_t)s_colors;
    lds_alphas = (tfn_lds_alphas_t)s_alphas;
  }
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t n_in = FIRST ? p.n_local : counters[C_RAYS0 + parity];
  uint32_t* n_samples_out = counters + C_SAMPLES0 + parity;
  const uint32_t n_round = (n_in + 255u) & ~255u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  // s_claim: per trip of the block's loop (double-buffered) the samples of the 4 waves, [4] the block's base
  uint32_t trip = 0;

  for (uint32_t base = blockIdx.x * 256u; base < n_round; base += gridDim.x * 256u) {
    const uint32_t i = base + tid;
    const bool active = i < n_in;
    uint32_t pixel = 0;
    float jitter = 0.0f, alpha = 0.0f;
    vec3f color = {0, 0, 0};
    DDAState it;
    it.t_next = {0, 0, 0}; it.cell = {0, 0, 0}; it.next_cell_begin = 0.0f;
    vec3f org = {0, 0, 0}, dir = {0, 0, 1}, m_dir = {0, 0, 1};
    float tmin = 0.0f, tmax = VNR_FLOAT_LARGE;
    bool alive = false;
    vec3f h_org = {0, 0, 0}, h_color = {0, 0, 0};   // M_SSH: SingleShotPayload
    float h_alpha = 0.0f;

    // the end of a ray: the pixel (M_NONE / M_GRADIENT), the hand-over to the shadow pass (M_SSH, :827-833), or the shaded
    // pixel (M_SHADOW, :820-826: lerp(scivis_shading_scale, unshaded, highest colour x pixel alpha x transmittance))
    auto finish = [&]() {
      if (MODE == M_SSH) {
        p.px_org[pixel] = h_org; p.px_color[pixel] = h_color; p.px_alpha[pixel] = h_alpha;
        p.px_shading[pixel] = {color.x, color.y, color.z, alpha};
      } else if (MODE == M_SHADOW) {
        const float transmittance = 1.0f - alpha, k = 0.95f;
        vec4f sc = p.px_shading[pixel];
        const vec3f hc = p.px_color[pixel];
        sc.x = (1.0f - k) * sc.x + k * (hc.x * sc.w * transmittance);
        sc.y = (1.0f - k) * sc.y + k * (hc.y * sc.w * transmittance);
        sc.z = (1.0f - k) * sc.z + k * (hc.z * sc.w * transmittance);
        write_pixel(p, sc, pixel);
      } else {
        write_pixel(p, {color.x, color.y, color.z, alpha}, pixel);
      }
    };

#if defined(VNR_MARCH_STAMPS)
    VNR_REALTIME(rt0);
#endif
    VNR_STAMP(st0);
    if (active) {
      if (FIRST) {
        if (map_pixel(p, i, pixel)) {
          if (MODE == M_SHADOW) {  // iterative_raygen_kernel_shadow (:877-900)
            jitter = p.px_jitter[pixel];
            org = p.px_org[pixel];
            dir = p.shadow_dir;
            color = org;           // color_or_org: a shadow ray carries its origin where a camera ray carries its colour
            m_dir = dir * p.mc_rcp;
            alive = intersect_box(tmin, tmax, org, dir, p.bbox_lo, p.bbox_hi) && p.px_alpha[pixel] > 0.0f;
            if (alive) dda_init(it, org * p.mc_rcp, m_dir, tmin, p.mc_dims);
            else write_pixel(p, p.px_shading[pixel], pixel);
          } else {
            if (MODE == M_SSH) {
              float j2;
              const uint32_t state2 = tea_lcg_two((uint32_t)p.frame_index, pixel, jitter, j2);
              // streaming: jitters.y (:866-868); in shader (mode 12): the next get_floats().x, i.e. the third draw
              p.px_jitter[pixel] = p.ssh_third_draw ? (float)((1664525u * state2 + 1013904223u) & 0x00FFFFFFu) / (float)0x01000000 : j2;
            } else {
              jitter = tea_lcg_first((uint32_t)p.frame_index, pixel);
            }
            compute_ray(p, pixel, org, dir);
            m_dir = dir * p.mc_rcp;
            alive = intersect_box(tmin, tmax, org, dir, p.bbox_lo, p.bbox_hi);
            if (alive) dda_init(it, org * p.mc_rcp, m_dir, tmin, p.mc_dims);
            else if (MODE == M_SSH) finish();   // zeros: what the reference's per-frame memset leaves for the shadow pass (:870-874)
            else write_pixel(p, {0, 0, 0, 0}, pixel);
          }
        }
      } else {
        pixel = cur.pixel_index[i];
        jitter = cur.jitter[i];
        alpha = cur.alpha[i];
        color = cur.color[i];
        it.cell = cur.cell[i];
        it.t_next = cur.t_next[i];
        it.next_cell_begin = cur.next_cell_begin[i];
        const uint32_t sb = cur.sample_base[i], sc = cur.sample_count[i];
        if (MODE == M_SHADOW) { org = color; dir = p.shadow_dir; }   // compute_ray<SHADOW> (:639-653)
        else compute_ray(p, pixel, org, dir);
        if (MODE == M_SSH) { h_org = ssh_lists.org[0][i]; h_color = ssh_lists.color[0][i]; h_alpha = ssh_lists.alpha[0][i]; }
        m_dir = dir * p.mc_rcp;
        intersect_box(tmin, tmax, org, dir, p.bbox_lo, p.bbox_hi);
        VNR_STAMP(st1);
        VNR_STAMP_ADD(0, st0, st1);   // ray state loaded
        // compose (classification, opacity correction, front-to-back blending)
        // The results were written by the evaluation kernel on other CUs (other XCDs: their L2 is not ours), so a load of them
        // costs a trip to the fabric, and a loop that loads, classifies, blends and tests for saturation sample by sample pays
        // that trip per sample: stamped on a 1/8 share of the bench frame, 2000 cycles per sample, 56 % of the kernel (DESIGN.md
        // 4.2).  The batch is composed in chunks: the chunk's results are fetched back to back (clamped index, no predicate:
        // one trip per chunk), then classified and blended in order with the reference's early exit.
        constexpr uint32_t kChunk = 8;
        const uint32_t sc_eff = (dbg(p) & 1u) ? 0u : sc;
        bool saturated = false;
#if defined(VNR_MARCH_STAMPS)
        unsigned long long acc_load = 0, acc_cls = 0, acc_blend = 0;
#endif
        // the next chunk's results are requested before this chunk is classified: next to the evaluation kernels of the other ray
        // part a trip to the results takes ~8 us (stamped), and a batch is three chunks
        vec2f ahead[kChunk];
#pragma unroll
        for (uint32_t j = 0; j < kChunk; ++j) ahead[j] = vd_in[sb + 64u * min(j, sc - 1u)];
        for (uint32_t k0 = 0; k0 < sc_eff && !saturated; k0 += kChunk) {
          vec2f chunk[kChunk];
          VNR_STAMP(sc0);
#pragma unroll
          for (uint32_t j = 0; j < kChunk; ++j) chunk[j] = ahead[j];
#pragma unroll
          for (uint32_t j = 0; j < kChunk; ++j) ahead[j] = vd_in[sb + 64u * min(k0 + kChunk + j, sc - 1u)];
#if defined(VNR_MARCH_STAMPS)
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
          VNR_STAMP(sc1);
          acc_load += sc1 - sc0;
#endif
          // classification and opacity correction of the whole chunk first: eight independent instruction streams for the
          // scheduler to interleave (a march block runs one or two waves per SIMD, where a dependent instruction issues every
          // ~10 cycles and an independent one every 4); only the blend below is sequential
          vec3f crgb[kChunk]; float ca[kChunk];
#pragma unroll
          for (uint32_t j = 0; j < kChunk; ++j) {
            if (dbg(p) & 2u) { crgb[j] = {0.5f, 0.5f, 0.5f}; ca[j] = chunk[j].x * 0.01f; }
            else if (p.tfn_in_lds) tfn_sample_lds(tfn, lds_colors, lds_alphas, chunk[j].x, crgb[j], ca[j], tfn_merged);   // uniform branch
            else tfn_sample(tfn, chunk[j].x, crgb[j], ca[j]);
            ca[j] = opacity_correction(p.step_rcp, chunk[j].y, ca[j]);
          }
#if defined(VNR_MARCH_STAMPS)
          asm volatile("" :: "v"(ca[0]), "v"(ca[7]), "v"(crgb[7].x));
          VNR_STAMP(sc2);
          acc_cls += sc2 - sc1;
#endif
#pragma unroll
          for (uint32_t j = 0; j < kChunk; ++j) {
            const uint32_t k = k0 + j;
            if (k >= sc_eff) break;
            vec3f rgb = crgb[j]; float a = ca[j];
            const vec2f vd = chunk[j];  // {network value, t1 - t0}
            if (GRAD) {  // f(c + gx), f(c + gy), f(c + gz) of this sample, written by the evaluation kernel; .w: the sample's t
              const vec4f fg = *(const vec4f*)((const float*)vd_in + arena_grad_index(p.slot_cap, sb + 64u * k));
              vec3f stp = p.grad_step;
              if (p.grad_flip) {  // in shader (mode 9): repeat sampleGradient's flip of a step that would leave [0,1] (raytracing.h:128-143)
                const vec3f c = org + fg.w * dir;
                if (c.x + stp.x > 1.0f - FLT_EPSILON) stp.x *= -1.0f;
                if (c.y + stp.y > 1.0f - FLT_EPSILON) stp.y *= -1.0f;
                if (c.z + stp.z > 1.0f - FLT_EPSILON) stp.z *= -1.0f;
              }
              rgb = gradient_shade(p, dir, vd.x, fg.x, fg.y, fg.z, stp, rgb);
            }
            if (MODE == M_SSH && h_alpha < (1.0f - alpha) * a) {  // :789-795; the sample's t is kept where GRAD keeps f(c + gx)
              const float t = ((const float*)vd_in)[arena_grad_index(p.slot_cap, sb + 64u * k)];
              h_org = org + t * dir;
              h_color = rgb;
              h_alpha = (1.0f - alpha) * a;
            }
            const float tr = 1.0f - alpha;
            alpha += tr * a;
            if (MODE != M_SHADOW) { color.x += tr * rgb.x * a; color.y += tr * rgb.y * a; color.z += tr * rgb.z * a; }
            if (!(alpha < VNR_NEARLY_ONE)) { saturated = true; break; }
          }
#if defined(VNR_MARCH_STAMPS)
          asm volatile("" :: "v"(alpha), "v"(color.x));
          VNR_STAMP(sc3);
          acc_blend += sc3 - sc2;
#endif
        }
#if defined(VNR_MARCH_STAMPS)
        VNR_STAMP_ADD(8, 0ull, acc_load); VNR_STAMP_ADD(9, 0ull, acc_cls); VNR_STAMP_ADD(10, 0ull, acc_blend);
#endif
        alive = (alpha < VNR_NEARLY_ONE) && dda_resumable(it, m_dir, tmin, tmax, p.mc_dims);
        if (!alive) finish();
      }
    }

    VNR_STAMP(st2);
    VNR_STAMP_ADD(1, st0, st2);   // ... + compose
    // emit the next batch of this ray into LDS
    uint32_t k = 0;
#if defined(VNR_MARCH_STAMPS)
    uint32_t dbg_cells = 0;
    const vec3i dbg_cell0 = it.cell;
#endif
    if (alive && !(dbg(p) & 8u)) {
      const int n_iters = p.n_iters;
      iter_exec(p, it, m_dir, tmin, tmax, p.step, [&](float t0, float t1) -> bool {
        s_t0[k * 256u + tid] = t0;
        s_t1[k * 256u + tid] = t1;
        return (int)(++k) < n_iters;
      });
      if (k == 0) finish();  // nothing left to sample: the ray is finished
#if defined(VNR_MARCH_STAMPS)
      dbg_cells = (uint32_t)(abs(it.cell.x - dbg_cell0.x) + abs(it.cell.y - dbg_cell0.y) + abs(it.cell.z - dbg_cell0.z));
#endif
    }
#if defined(VNR_MARCH_STAMPS)
    {  // macrocell steps of this trip: the wave's longest ray, the sum over its rays, its alive rays
      uint32_t mx = dbg_cells, sm = dbg_cells;
      for (int d = 32; d > 0; d >>= 1) { mx = max(mx, (uint32_t)__shfl_xor((int)mx, d)); sm += (uint32_t)__shfl_xor((int)sm, d); }
      const uint32_t n_alive = (uint32_t)__popcll(__ballot(alive));
      VNR_STAMP_ADD(11, 0ull, (unsigned long long)mx); VNR_STAMP_ADD(12, 0ull, (unsigned long long)sm); VNR_STAMP_ADD(13, 0ull, (unsigned long long)n_alive);
    }
#endif
    const bool survive = alive && k > 0;
    VNR_STAMP(st3);
    VNR_STAMP_ADD(2, st2, st3);   // DDA walk + samples to LDS

    // wave64 compaction: rays by ballot, samples by prefix sum
    const unsigned long long mask = __ballot(survive);
    const unsigned long long alive_mask = __ballot(alive);
    uint32_t incl = k;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(incl, d);
      if ((int)lane >= d) incl += y;
    }
    const uint32_t wave_samples = __shfl(incl, 63);
    const uint32_t wave_rays = (uint32_t)__popcll(mask);
    // the march the host expects to be the frame's last is launched without evaluation and packing behind it and reports here (pinned
    // host memory) whether that expectation held (launch_iteration)
    if (tail_flag && lane == 0 && wave_rays) *tail_flag = 1u;
    // Surviving rays go to the wave's OWN 64 slots of the scratch list (`nxt`) and compact_rays_kernel packs the groups in
    // wave order afterwards: an order-preserving compaction, so a 64-ray group stays a compact patch of the image over the
    // iterations.  (Claiming slots with an atomic, as the reference does, hands them out in wave-arrival order; by the third
    // iteration a group then mixes rays of distant tiles and the hash-grid gathers of its samples lose 20-35 % of their rate.)
    const uint32_t group = i >> 6;
    // Sample slots: ONE claim per block.  A device-scope atomic is resolved beyond the per-XCD L2, and atomics on one address
    // are served one after the other (about 10 ns each, measured: a first march of 131 072 pixels whose waves each issued four
    // of them took 45 us before tracing a single ray), so neither the claim is made per wave nor are the frame statistics
    // counted here: compact_rays_kernel sums them from the per-group counts.
    // The queue counter counts RECORDS (what the evaluation kernel reads); with 4 records per sample every claim is a
    // multiple of 4, so claim / 4 is a unique sample-slot base.
    uint32_t* claim = s_claim + 8u * (trip & 1u);
    if (lane == 0) {
      ray_counts[group] = wave_rays | ((uint32_t)__popcll(alive_mask) << 8);  // survivors | rays that were alive when the march began to emit
      claim[tid >> 6] = wave_samples;
    }
    __syncthreads();
    if (tid == 0) {
      const uint32_t total = claim[0] + claim[1] + claim[2] + claim[3];
      uint32_t b = 0;
      if (total) b = GRAD ? atomicAdd(n_samples_out, 4u * total) >> 2 : atomicAdd(n_samples_out, total);
      claim[4] = b;
    }
    __syncthreads();
    uint32_t smp_base = claim[4];
    for (uint32_t w = 0; w < (tid >> 6); ++w) smp_base += claim[w];
    ++trip;
    VNR_STAMP(st4);
    VNR_STAMP_ADD(3, st3, st4);   // compaction, block-wide slot claim (two barriers)
    if (wave_rays == 0 || (dbg(p) & 16u)) continue;  // wave-uniform

    // depth bins of the group: front = smallest first-sample depth among the surviving rays
    float front = survive ? s_t0[tid] : VNR_FLOAT_LARGE;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) front = fminf(front, __shfl_xor(front, d));
    // counting sort by depth bin (kDepthBins == 64: one histogram counter per lane, LDS atomics inside the wave)
    uint32_t* hist = s_hist + (tid & ~63u);
    hist[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (survive) {
      for (uint32_t j = 0; j < k; ++j) {
        const float t0 = s_t0[j * 256u + tid], t1 = s_t1[j * 256u + tid];
        const float t = (1.0f - jitter) * t0 + jitter * t1;
        const uint32_t bin = depth_bin(p, t, front);
        if (p.no_ranks) atomicAdd(&hist[bin], 1u);   // count only: the slot is claimed from the bin's counter when the record is written
        else s_rk[j * 256u + tid] = (uint16_t)atomicAdd(&hist[bin], 1u);
      }
    }
    __builtin_amdgcn_wave_barrier();
    // exclusive scan of the 64-bin histogram across the wave
    const uint32_t h = hist[lane];
    uint32_t hs = h;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(hs, d);
      if ((int)lane >= d) hs += y;
    }
    hist[lane] = smp_base + hs - h;  // first gather-order slot of bin `lane`
    __builtin_amdgcn_wave_barrier();
    VNR_STAMP(st5);
    VNR_STAMP_ADD(4, st4, st5);   // depth-bin counting sort

    if (survive) {
      const uint32_t slot = (group << 6) + (uint32_t)__popcll(mask & lt_mask);
      // Result slots: sample j of the ray in lane l of group g lives in slot (g n_iters + j) 64 + l, so that the 64 rays of a wave read
      // (compose, next launch) and write (dt below; the evaluation kernel's scatter of a depth bin) NEIGHBOURING slots: a ray-major arena
      // made every one of those instructions touch 64 lines.  The arena has n_iters slots for every ray anyway; unused ones are holes.
      const uint32_t sb = group * (uint32_t)p.n_iters * 64u + lane;
      nxt.pixel_index[slot] = pixel;
      nxt.jitter[slot] = jitter;
      nxt.alpha[slot] = alpha;
      nxt.color[slot] = color;
      nxt.cell[slot] = it.cell;
      nxt.t_next[slot] = it.t_next;
      nxt.next_cell_begin[slot] = it.next_cell_begin;
      nxt.sample_base[slot] = sb;
      nxt.sample_count[slot] = k;
      if (MODE == M_SSH) { ssh_lists.org[1][slot] = h_org; ssh_lists.color[1][slot] = h_color; ssh_lists.alpha[1][slot] = h_alpha; }
      for (uint32_t j = 0; j < k; ++j) {
        const float t0 = s_t0[j * 256u + tid], t1 = s_t1[j * 256u + tid];
        const float t = (1.0f - jitter) * t0 + jitter * t1;  // lerp(jitter, t0, t1), instantvnr_types.h:162-166
        const vec3f c = org + t * dir;
        // gather-order slot: the bin's first slot + the sample's rank in the bin; without stored ranks the bin's counter is the next free slot
        // (any order of a bin's samples will do: the evaluation of a sample does not depend on its neighbours in the queue)
        const uint32_t g = p.no_ranks ? atomicAdd(&hist[depth_bin(p, t, front)], 1u) : hist[depth_bin(p, t, front)] + s_rk[j * 256u + tid];
        // one 16-byte record per evaluation: position + the float index of the result arena its value goes to
        if (GRAD) {
          const uint32_t gi = arena_grad_index(p.slot_cap, sb + 64u * j);
          vec4f* q = queue + 4u * (size_t)g;  // the four records of a sample stay adjacent: they fall into the same grid cells
          vec3f stp = p.grad_step;
          if (p.grad_flip) {
            if (c.x + stp.x > 1.0f - FLT_EPSILON) stp.x *= -1.0f;
            if (c.y + stp.y > 1.0f - FLT_EPSILON) stp.y *= -1.0f;
            if (c.z + stp.z > 1.0f - FLT_EPSILON) stp.z *= -1.0f;
            ((float*)vd_out)[gi + 3u] = t;
          }
          q[0] = {c.x, c.y, c.z, __uint_as_float(arena_value_index(sb + 64u * j))};
          q[1] = {c.x + stp.x, c.y, c.z, __uint_as_float(gi + 0u)};
          q[2] = {c.x, c.y + stp.y, c.z, __uint_as_float(gi + 1u)};
          q[3] = {c.x, c.y, c.z + stp.z, __uint_as_float(gi + 2u)};
        } else {
          if (g < p.slot_cap) queue[g] = {c.x, c.y, c.z, __uint_as_float(arena_value_index(sb + 64u * j))};   // (always: the claims of a launch sum to at most n_local x n_iters)
        }
        vd_out[sb + 64u * j].y = t1 - t0;
        if (MODE == M_SSH) ((float*)vd_out)[arena_grad_index(p.slot_cap, sb + 64u * j)] = t;   // compose needs the position of the sample
      }
    }
    __builtin_amdgcn_wave_barrier();  // the LDS arrays are reused by the next loop trip
    VNR_STAMP(st6);
    VNR_STAMP_ADD(5, st5, st6);   // ray state + queue records + dt written
    VNR_STAMP_ADD(6, st0, st6);   // the whole trip
    VNR_STAMP_ADD(7, st0, st0 + 1ull);   // trips
#if defined(VNR_MARCH_STAMPS)
    if (!FIRST && lane == 0) {   // the longest trip of this group over the launches since the records were cleared (tools/wave_records.py)
      VNR_REALTIME(rt1);
      unsigned long long* rec = g_wave_rec[((p.il_part & 7u) << 13) | (group & 8191u)];
      if (st6 - st0 > rec[7]) {
        rec[0] = rt0; rec[1] = rt1; rec[2] = st2 - st0; rec[3] = st3 - st2; rec[4] = st4 - st3; rec[5] = st5 - st4; rec[6] = st6 - st5; rec[7] = st6 - st0;
      }
    }
#endif
  }
}

#if defined(VNR_MARCH_STAMPS)
extern "C" int vnrAmdDebugWaveRecords(unsigned long long* out, unsigned n_records, int reset)
{
  if (n_records > kWaveRecs) return 1;
  if (hipMemcpyFromSymbol(out, HIP_SYMBOL(g_wave_rec), (size_t)n_records * 8 * sizeof(unsigned long long)) != hipSuccess) return 1;
  if (reset) { void* d = nullptr; if (hipGetSymbolAddress(&d, HIP_SYMBOL(g_wave_rec)) != hipSuccess || hipMemset(d, 0, sizeof(g_wave_rec)) != hipSuccess) return 1; }
  return 0;
}
extern "C" int vnrAmdDebugMarchStamps(unsigned long long* out16, int reset)
{
  if (hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_march_stamps), sizeof(g_march_stamps)) != hipSuccess) return 1;
  if (reset) { unsigned long long z[16] = {}; (void)hipMemcpyToSymbol(HIP_SYMBOL(g_march_stamps), z, sizeof(z)); }
  return 0;
}
#endif

// iterative_sampling_groundtruth_kernel (method_raymarching.cu:902-915) over the compacted sample queue
__global__ void gt_sample_kernel(const uint32_t* __restrict__ n_ptr, const float* __restrict__ vol, vec3i dims,
                                 const vec4f* __restrict__ queue, float* __restrict__ arena)
{
  const uint32_t n = *n_ptr;  // records (4 per sample with gradient shading)
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const vec4f r = queue[i];
    arena[__float_as_uint(r.w)] = sample_volume_nodal(vol, dims, r.x, r.y, r.z);
  }
}

// pack_rays_block (pack_rays.h) as a kernel of its own: ground-truth volumes, models outside the MFMA kernels' shapes, VNR_AMD_FUSED_PACK=0
template <int WAVES>
__global__ void __launch_bounds__(64 * WAVES) compact_rays_kernel(const PackArgs a)
{
  __shared__ uint32_t s_part[WAVES];
  pack_rays_block<WAVES>(a, blockIdx.x, s_part);
}

// ------------------------------------------------------------------------------------------------ monolithic marcher (mode 4)
// raymarching_kernel / raymarching_traceray / raymarching_iterator, NO_SHADING (method_raymarching.cu:263-308,401-536)
__global__ void monolithic_kernel(const RenderParams p)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n_local) return;
  uint32_t pixel;
  if (!map_pixel(p, i, pixel)) return;
  vec3f org, dir;
  compute_ray(p, pixel, org, dir);
  float alpha = 0.0f;
  vec3f color = {0, 0, 0};
  vec3f h_org = {0, 0, 0}, h_color = {0, 0, 0};   // SINGLE_SHADE_HEURISTIC (mode 10, :414-416)
  float h_alpha = 0.0f;

  // raymarching_iterator (:270-308) = dda3 (dda.h:140-287) over the macrocells with the equalised adaptive step; one walk for
  // the camera ray, a second one (shadow = true: alpha only, raymarching_transmittance :364-398) for the single-shade ray
  auto march = [&](vec3f o, vec3f d, float t0, float t1, float jitter, float step, bool shadow) {
    const vec3f m_dir = d * p.mc_rcp;
    DDAState it;
    dda_init(it, o * p.mc_rcp, m_dir, t0, p.mc_dims);
    auto cell_fn = [&](vec3i cell, float c0, float c1) -> bool {
      const float r = opacity_upper_bound(p, cell);
      if (fabsf(r) <= FLT_EPSILON) return true;
      float ss = adaptive_sampling_rate(step, r);
      {  // sample_size_scaler :263-268
        const int N = (int)((c1 - c0) / ss + 1.0f);
        ss = (c1 - c0) / (float)N;
      }
      float tx = c0, ty = fminf(c1, c0 + ss);
      while (ty > tx) {
        const float t = (1.0f - jitter) * tx + jitter * ty;
        const vec3f c = o + t * d;
        const float v = sample_volume_nodal(p.volume, p.vol_dims, c.x, c.y, c.z);
        vec3f rgb; float a;
        tfn_sample(p.tfn, v, rgb, a);
        a = opacity_correction(p.step_rcp, ty - tx, a);
        if (shadow) {
          alpha += (1.0f - alpha) * a;
          if (!(alpha < VNR_NEARLY_ONE)) return false;
          tx = ty;
          ty = fminf(tx + ss, c1);
          continue;
        }
        if (p.shading_mode == 1u) {  // mode 7, :440-454 with sampleGradient (raytracing.h:112-126): a step leaving [0,1] is flipped
          vec3f stp = p.grad_step;
          if (c.x + stp.x > 1.0f - FLT_EPSILON) stp.x *= -1.0f;
          if (c.y + stp.y > 1.0f - FLT_EPSILON) stp.y *= -1.0f;
          if (c.z + stp.z > 1.0f - FLT_EPSILON) stp.z *= -1.0f;
          const float fgx = sample_volume_nodal(p.volume, p.vol_dims, c.x + stp.x, c.y, c.z);
          const float fgy = sample_volume_nodal(p.volume, p.vol_dims, c.x, c.y + stp.y, c.z);
          const float fgz = sample_volume_nodal(p.volume, p.vol_dims, c.x, c.y, c.z + stp.z);
          rgb = gradient_shade(p, d, v, fgx, fgy, fgz, stp, rgb);
        }
        if (p.shading_mode == 2u && h_alpha < (1.0f - alpha) * a) {  // mode 10, :455-462
          h_org = c; h_color = rgb; h_alpha = (1.0f - alpha) * a;
        }
        const float tr = 1.0f - alpha;
        color.x += tr * rgb.x * a; color.y += tr * rgb.y * a; color.z += tr * rgb.z * a;
        alpha += tr * a;
        if (!(alpha < VNR_NEARLY_ONE)) return false;
        tx = ty;
        ty = fminf(tx + ss, c1);
      }
      return true;
    };
    if (t0 < t1) {
      const vec3i stop = {m_dir.x > 0.0f ? p.mc_dims.x : -1, m_dir.y > 0.0f ? p.mc_dims.y : -1, m_dir.z > 0.0f ? p.mc_dims.z : -1};
      const vec3f ts = {fabsf(1.0f / m_dir.x), fabsf(1.0f / m_dir.y), fabsf(1.0f / m_dir.z)};
      const vec3i delta = {m_dir.x > 0.0f ? 1 : -1, m_dir.y > 0.0f ? 1 : -1, m_dir.z > 0.0f ? 1 : -1};
      for (;;) {
        const float t_closest = min3f(it.t_next.x, it.t_next.y, it.t_next.z);
        const float c0 = fmaxf(t0 + it.next_cell_begin, t0), c1 = fminf(t0 + t_closest, t1);
        if (c0 >= c1) break;
        if (!cell_fn(it.cell, c0, c1)) break;
        if (it.t_next.x == t_closest) { it.t_next.x += ts.x; it.cell.x += delta.x; if (it.cell.x == stop.x) break; }
        if (it.t_next.y == t_closest) { it.t_next.y += ts.y; it.cell.y += delta.y; if (it.cell.y == stop.y) break; }
        if (it.t_next.z == t_closest) { it.t_next.z += ts.z; it.cell.z += delta.z; if (it.cell.z == stop.z) break; }
        it.next_cell_begin = t_closest;
      }
    }
  };

  float t0 = 0.0f, t1 = VNR_FLOAT_LARGE;
  if (intersect_box(t0, t1, org, dir, p.bbox_lo, p.bbox_hi)) {
    float jitter, unused;
    tea_lcg_two((uint32_t)p.frame_index, pixel, jitter, unused);   // rng.get_floats().x; the generator advances by two draws
    march(org, dir, t0, t1, jitter, p.step, false);
    if (p.shading_mode == 2u && h_alpha > 0.0f) {  // :471-484: one shadow ray from the strongest sample towards the light
      const float pixel_alpha = alpha;
      float s0 = 0.0f, s1 = VNR_FLOAT_LARGE;
      alpha = 0.0f;
      if (intersect_box(s0, s1, h_org, p.shadow_dir, p.bbox_lo, p.bbox_hi)) {
        // third draw of the pixel's generator: state3 = a (a (a v0 + c) + c) + c, recomputed from the first two
        uint32_t v0 = (uint32_t)p.frame_index, v1 = pixel, s = 0;
#pragma unroll
        for (int n = 0; n < 16; ++n) {
          s += 0x9e3779b9u;
          v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s) ^ ((v1 >> 5) + 0xc8013ea4u);
          v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s) ^ ((v0 >> 5) + 0x7e95761eu);
        }
        uint32_t st = 1664525u * v0 + 1013904223u;
        st = 1664525u * st + 1013904223u;
        st = 1664525u * st + 1013904223u;
        const float j3 = (float)(st & 0x00FFFFFFu) / (float)0x01000000;
        march(h_org, p.shadow_dir, s0, s1, j3, 2.0f * p.step, true);   // raymarching_shadow_sampling_scale = 2 (instantvnr_types.h:137)
      }
      const float transmittance = 1.0f - alpha, k = 0.95f;
      alpha = pixel_alpha;
      color.x = (1.0f - k) * color.x + k * (h_color.x * alpha * transmittance);
      color.y = (1.0f - k) * color.y + k * (h_color.y * alpha * transmittance);
      color.z = (1.0f - k) * color.z + k * (h_color.z * alpha * transmittance);
    }
  }
  write_pixel(p, {color.x, color.y, color.z, alpha}, pixel);
}

// FIRST: iterative_raygen_kernel (:679-748), thread = pixel of an 8x8 tile; else iterative_shade_kernel (:750-768), thread = alive ray
template <bool FIRST>
__global__ void __launch_bounds__(256) pt_kernel(const RenderParams p, const PtRays dense, const PtRays scratch, const float* __restrict__ values,
                                                 uint32_t* __restrict__ counters, uint32_t* __restrict__ ray_counts, int parity)
{
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t n_in = FIRST ? p.n_local : counters[C_RAYS0 + parity];
  const uint32_t n_round = (n_in + 255u) & ~255u;
  const unsigned long long lt_mask = (1ull << lane) - 1ull;
  for (uint32_t base = blockIdx.x * 256u; base < n_round; base += gridDim.x * 256u) {
    const uint32_t i = base + tid;
    PtRay r;
    bool alive = false, finished = false;
    if (i < n_in) {
      if (FIRST) {
        uint32_t pixel;
        if (map_pixel(p, i, pixel)) {
          r.pidx = pixel; r.shadow = false;
          compute_ray(p, pixel, r.org, r.dir);
          r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;
          r.scatter_index = 0; r.sample_coord = {0, 0, 0}; r.majorant = 0.0f;
          r.L = {0, 0, 0}; r.throughput = {1, 1, 1};
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
          r.it.t_next = {0, 0, 0}; r.it.cell = {0, 0, 0}; r.it.next_cell_begin = 0.0f;
          if (intersect_box(r.tnear, r.tfar, r.org, r.dir, p.bbox_lo, p.bbox_hi)) {
            atomicAdd(counters + C_HIT, 1u);
            dda_init(r.it, r.org * p.mc_rcp, r.dir * p.mc_rcp, r.tnear, p.mc_dims);
            alive = pt_take_sample(p, r);
          }
          finished = !alive;
        }
      } else {
        r.load(dense, i);
        r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;   // load (:126-129): the interval is recomputed every iteration
        intersect_box(r.tnear, r.tfar, r.org, r.dir, p.bbox_lo, p.bbox_hi);
        alive = pt_shade(p, p.tfn, r, values[i]) && pt_take_sample(p, r);
        finished = !alive;
      }
    }
    if (finished) write_pixel(p, {r.L.x, r.L.y, r.L.z, 1.0f}, r.pidx);
    const unsigned long long mask = __ballot(alive);
    const uint32_t group = i >> 6;
    if (lane == 0) ray_counts[group] = (uint32_t)__popcll(mask);
    if (alive) r.store(scratch, (group << 6) + (uint32_t)__popcll(mask & lt_mask));
  }
}

// path tracer on a dense volume in one loop per pixel (rendering mode 13, "Decoding - Debug"): path_tracing_kernel /
// path_tracing_traceray / delta_tracking with USE_DELTA_TRACKING_ITER (method_pathtracing.cu:258-292, 420-510).  Unlike the
// streaming variant the interval is reset before a bounce (:438-439).
__global__ void pt_monolithic_kernel(const RenderParams p)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= p.n_local) return;
  uint32_t pixel;
  if (!map_pixel(p, i, pixel)) return;
  PtRay r;
  r.pidx = pixel; r.shadow = false;
  compute_ray(p, pixel, r.org, r.dir);
  r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;
  r.scatter_index = 0; r.majorant = 0.0f; r.sample_coord = {0, 0, 0};
  r.L = {0, 0, 0}; r.throughput = {1, 1, 1};
  {
    uint32_t v0 = (uint32_t)p.frame_index, v1 = pixel, s0 = 0;
#pragma unroll
    for (int n = 0; n < 16; ++n) {
      s0 += 0x9e3779b9u;
      v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
      v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
    }
    r.rng = v0;
  }
  while (intersect_box(r.tnear, r.tfar, r.org, r.dir, p.bbox_lo, p.bbox_hi)) {
    // delta_tracking: tentative collisions from hashit until a real one
    float t = r.tnear;
    vec3f albedo = {0, 0, 0};
    bool found = false;
    dda_init(r.it, r.org * p.mc_rcp, r.dir * p.mc_rcp, r.tnear, p.mc_dims);
    while (pt_hashit(p, r, t)) {
      const vec3f c = r.org + t * r.dir;
      const float v = sample_volume_nodal(p.volume, p.vol_dims, c.x, c.y, c.z);
      vec3f rgb; float a;
      tfn_sample(p.tfn, v, rgb, a);
      if (r.next_float() * r.majorant < a * p.density_scale) { albedo = rgb; found = true; break; }
    }
    const bool exited = !found;
    if (r.shadow) {
      if (exited) r.L = r.L + r.throughput;
      r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;
      const float s0 = r.next_float(), s1 = r.next_float();
      r.dir = xfm_vector(p.wto, pt_uniform_sample_sphere(s0, s1));
      r.shadow = false;
    } else {
      if (exited) {
        if (r.scatter_index > 0u) r.L = r.L + 1.5f * r.throughput;
        break;
      }
      if (r.scatter_index > 4u) {
        const float q = fminf(0.95f, max3f(r.throughput.x, r.throughput.y, r.throughput.z));
        if (r.next_float() > q) break;
        r.throughput = {r.throughput.x / q, r.throughput.y / q, r.throughput.z / q};
      }
      ++r.scatter_index;
      r.org = r.org + t * r.dir;
      r.throughput = r.throughput * (0.6f * albedo);
      r.tnear = 0.0f; r.tfar = VNR_FLOAT_LARGE;
      r.dir = p.shadow_dir;
      r.shadow = true;
    }
  }
  write_pixel(p, {r.L.x, r.L.y, r.L.z, 1.0f}, pixel);
}

// packs the survivors in group order (compact_rays_kernel's scheme) and writes one queue record per alive ray:
// {sample_coord, index of the ray} -> the evaluation kernel puts the value where the next pt_kernel reads it
__global__ void __launch_bounds__(1024) pt_compact_kernel(const PtRays src, const PtRays dst, const uint32_t* __restrict__ ray_counts,
                                                          uint32_t n_first, uint32_t* __restrict__ counters, int parity, int first,
                                                          vec4f* __restrict__ queue)
{
  __shared__ uint32_t s_part[16];
  const uint32_t tid = threadIdx.x, lane = tid & 63u, wave = tid >> 6;
  const uint32_t n_in = first ? n_first : counters[C_RAYS0 + parity];
  const uint32_t n_groups = ((n_in + 255u) & ~255u) >> 6;
  if (n_groups == 0) {
    if (blockIdx.x == 0 && tid == 0) counters[C_RAYS0 + (parity ^ 1)] = 0;
    return;
  }
  const uint32_t g0 = blockIdx.x * 16u;
  if (g0 >= n_groups) return;
  uint32_t sum = 0;
  for (uint32_t g = tid; g < g0; g += 1024u) sum += ray_counts[g];
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) sum += __shfl_xor(sum, d);
  if (lane == 0) s_part[wave] = sum;
  __syncthreads();
  uint32_t before = 0;
#pragma unroll
  for (int w = 0; w < 16; ++w) before += s_part[w];
  const uint32_t mine = (lane < 16u && g0 + lane < n_groups) ? ray_counts[g0 + lane] : 0u;
  uint32_t incl = mine;
#pragma unroll
  for (int d = 1; d < 16; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if ((int)lane >= d) incl += y;
  }
  const uint32_t count = __shfl(mine, (int)wave), base = before + __shfl(incl, (int)wave) - count;
  const uint32_t block_total = __shfl(incl, 15);
  if (g0 + 16u >= n_groups && tid == 0) {
    const uint32_t total = before + block_total;
    counters[C_RAYS0 + (parity ^ 1)] = total;
    atomicAdd((unsigned long long*)(counters + C_STAT_SAMPLES), (unsigned long long)total);
  }
  if (lane < count) {
    const uint32_t from = ((g0 + wave) << 6) + lane, to = base + lane;
#pragma unroll
    for (int k = 0; k < kPtPlanes; ++k) dst.base[(size_t)k * dst.stride + to] = src.base[(size_t)k * src.stride + from];
    queue[to] = {src.base[(size_t)8 * src.stride + from], src.base[(size_t)9 * src.stride + from], src.base[(size_t)10 * src.stride + from],
                 __uint_as_float(to)};
  }
}

}  // namespace vnr
#include "in_shader.h"   // the in-shader kernels' shapes and launch record (the kernels themselves are instantiated in in_shader_w*.hip)
#include "decoupled.h"   // walk_kernel, compose_kernel: the streaming loop with the walk off the evaluate -> compose chain
namespace vnr {

// ================================================================================================ Renderer (host)
struct PartState {
  RenderParams p;
  RayList rl[2];
  SshLists ssh;
  vec4f* queue;
  vec2f* vd[2];
  uint32_t* c;         // device counters of this half
  uint32_t* rc;        // survivors per 64-ray group
  uint32_t* hc;        // pinned ring of alive-ray counts (written by compact_rays_kernel)
  uint32_t* hs;        // pinned copy of this half's counters as of the last compact_rays_kernel
  hipStream_t s;
  size_t s_max;
  uint32_t it = 0, used = 0;
  bool done = false;
  bool tail_skipped = false;   // the last march launched has no evaluation / packing behind it yet (launch_iteration)
  hipEvent_t ev_done = nullptr;   // recorded behind the last iteration launched so far: the host (and the communication stream)
                                  // wait for THIS, not for the stream, which may already hold the head of the next frame
};

// One pass of the streaming loop between its two halves: everything launch_iteration / finish_streaming need.  With
// set_async(true) a frame stays in this state ("pending") from render() until the next call that needs its result.
// one ray part of the decoupled loop (decoupled.h)
struct DPart {
  RenderParams p;
  DRays rays;
  DRing ring[8];
  DHost* host = nullptr;
  hipStream_t sw = nullptr, se = nullptr, sc = nullptr;
  std::vector<hipEvent_t> ev_w, ev_e, ev_c;   // per iteration
  uint32_t it_w = 0, it_c = 0, target = 0;     // walks (+ evaluations) and composes enqueued so far; iterations to enqueue
  size_t s_max = 0;
  bool done = false;
  hipEvent_t& ev(std::vector<hipEvent_t>& v, uint32_t it)
  {
    if (it >= 1024u) throw std::runtime_error("internal: decoupled loop asked for the event of iteration " + std::to_string(it));   // (an index that wrapped must not become four billion events)
    while (v.size() <= it) { hipEvent_t e; VNR_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); v.push_back(e); }
    return v[it];
  }
  ~DPart() { for (auto* v : {&ev_w, &ev_e, &ev_c}) for (hipEvent_t e : *v) (void)hipEventDestroy(e); }
};

struct Renderer::StreamingFrame {
  bool pending = false;
  bool decoupled = false;
  int ahead = 2, ring = 3;
  DPart dpart[Renderer::kMaxParts];
  int slot = 0;
  int H = 0, pass_mode = 0;
  bool grad = false, ssh = false;
  NeuralVolume* nv = nullptr;
  size_t shmem = 0, shmem_compose = 0;
  uint32_t max_iterations = 240;
  uint32_t* predicted = nullptr;
  RenderParams p_all;
  uint64_t params_generation = 0;   // of the network when the frame was launched
  PartState part[Renderer::kMaxParts];
  StreamingFrame()
  {
    for (auto& p : part) {
      (void)hipEventCreateWithFlags(&p.ev_done, hipEventDisableTiming);
    }
  }
  ~StreamingFrame() { for (auto& p : part) if (p.ev_done) (void)hipEventDestroy(p.ev_done); }
  void mark(int h) { VNR_HIP_CHECK(hipEventRecord(part[h].ev_done, part[h].s)); }
};

Renderer::Renderer(std::shared_ptr<VolumeBase> volume) : volume_(std::move(volume))
{
  if (!Runtime::get().ready()) Runtime::get().init(-1);
  stream_ = Runtime::get().stream;
  // experiment (tools/two_renderers.py): a renderer whose part-0 chain does not share the runtime's stream with other renderers
  if (const char* e = std::getenv("VNR_AMD_RENDERER_OWN_STREAM")) if (std::atoi(e) != 0) { VNR_HIP_CHECK(hipStreamCreateWithFlags(&stream_, hipStreamNonBlocking)); own_stream_ = true; }
  if (const char* e = std::getenv("VNR_AMD_MARCH_RANKS")) march_ranks_ = std::atoi(e) != 0;
  if (const char* e = std::getenv("VNR_RM_N_ITERS")) { n_iters_ = std::max(1, std::min(48, std::atoi(e))); n_iters_fixed_ = true; }  // 2.5 KiB of LDS per iteration slot and block
  // streaming mode runs the rays as two halves on two streams (render_streaming); VNR_AMD_RENDER_HALVES=1: one stream
  if (const char* e = std::getenv("VNR_AMD_RENDER_HALVES")) { n_halves_ = std::max(1, std::min(kMaxParts, std::atoi(e))); n_halves_fixed_ = true; }
  if (const char* e = std::getenv("VNR_AMD_SMALL_SHARE_PARTS")) small_share_parts_ = std::max(1, std::min(kMaxParts, std::atoi(e)));
  if (const char* e = std::getenv("VNR_AMD_TILE_W")) {  // ray tile shape (diagnostics): 8 -> 8x8, 16 -> 16x4, 32 -> 32x2, 64 -> 64x1
    const int w = std::atoi(e);
    tile_w_log2_ = w == 16 ? 4u : w == 32 ? 5u : w == 64 ? 6u : 3u;
  }
  if (const char* e = std::getenv("VNR_AMD_DECOUPLED")) decoupled_mode_ = std::max(0, std::min(2, std::atoi(e)));
  if (const char* e = std::getenv("VNR_AMD_DECOUPLED_AHEAD")) decoupled_ahead_ = std::max(1, std::min(7, std::atoi(e)));
  if (const char* e = std::getenv("VNR_AMD_DECOUPLED_PARTS")) decoupled_parts_ = std::max(1, std::min(kMaxParts, std::atoi(e)));
  counters_.resize(2 * kMaxParts * C_COUNT);  // one block of counters per frame slot and half
  counters_.zero(stream_);
  VNR_HIP_CHECK(hipHostMalloc((void**)&host_counts_, 2 * kMaxParts * (256 + C_COUNT) * sizeof(uint32_t), hipHostMallocDefault));
  // (the ray parts' streams come from the process-wide pool, Runtime::part_stream, when a frame first uses them: a stream that exists takes a
  // place in the runtime's mapping of streams to its four hardware queues whether it is used or not, and a share's parts must not end up
  // sharing one: DESIGN.md 6)
  VNR_HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
}

Renderer::~Renderer()
{
  if (stream_) (void)hipStreamSynchronize(stream_);
  if (own_stream_) (void)hipStreamDestroy(stream_);
  for (int i = 1; i < kMaxParts; ++i) if (part_streams_[i]) (void)hipStreamSynchronize(part_streams_[i]);   // (pool streams: not this renderer's to destroy)
  if (ev_fork_) (void)hipEventDestroy(ev_fork_);
  for (auto& ps : d_streams_) for (hipStream_t st : ps) if (st) { (void)hipStreamSynchronize(st); (void)hipStreamDestroy(st); }
  frame_[0].reset(); frame_[1].reset();
  if (d_host_) (void)hipHostFree(d_host_);
  if (distributed_) (void)hipStreamSynchronize(Dist::get().comm_stream());
  if (ev_rendered_) (void)hipEventDestroy(ev_rendered_);
  for (int i = 0; i < 2; ++i) if (ev_gathered_[i]) (void)hipEventDestroy(ev_gathered_[i]);
  for (int i = 0; i < 2; ++i) if (host_fb_[i]) (void)hipHostFree(host_fb_[i]);
  if (host_counts_) (void)hipHostFree(host_counts_);
  for (auto& sl : events_) for (auto& v : sl) for (auto e : v) (void)hipEventDestroy(e);
}

void Renderer::resize(int w, int h)
{
  finish_pending();
  if (w <= 0 || h <= 0) throw std::runtime_error("invalid framebuffer size");
  VNR_HIP_CHECK(hipDeviceSynchronize());   // buffers are replaced below: nothing may still run on any of the renderer's streams
  width_ = w; height_ = h;
  const size_t n = (size_t)w * h;
  for (int i = 0; i < 2; ++i) { fb_[i].resize(n); fb_[i].zero(stream_); }
  accumulation_.resize(n);
  accumulation_.zero(stream_);
  if (host_fb_pixels_ != n) {
    for (int i = 0; i < 2; ++i) {
      if (host_fb_[i]) (void)hipHostFree(host_fb_[i]);
      VNR_HIP_CHECK(hipHostMalloc((void**)&host_fb_[i], n * sizeof(vec4f), hipHostMallocDefault));
      std::memset(host_fb_[i], 0, n * sizeof(vec4f));
    }
    host_fb_pixels_ = n;
  }
  if (distributed_) ensure_share_buffers();
  reset_ = true;
}

// ------------------------------------------------------------------------------------------------ distributed mode (dist.h)
// gathered [world][n_local] (slot r = the packed tile rows r, r + world, ...) -> the width x height frame
__global__ void assemble_shares_kernel(const vec4f* __restrict__ gathered, vec4f* __restrict__ full, uint32_t width, uint32_t height,
                                       uint32_t world, uint32_t n_local)
{
  const uint32_t n = width * height;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const uint32_t y = i / width, x = i - y * width, b = y >> 3;
    full[i] = gathered[(size_t)(b % world) * n_local + ((b / world) * 8u + (y & 7u)) * width + x];
  }
}

void Renderer::set_distributed(bool e)
{
  finish_pending();
  if (e && !Dist::get().active()) throw std::runtime_error("distributed rendering needs vnrAmdDistInitFromEnv / vnrAmdDistInit first");
  distributed_ = e && Dist::get().world() >= 1;
  pipe_prev_ = -1;
  if (distributed_) {
    if (!ev_rendered_) VNR_HIP_CHECK(hipEventCreateWithFlags(&ev_rendered_, hipEventDisableTiming));
    for (int i = 0; i < 2; ++i) if (!ev_gathered_[i]) VNR_HIP_CHECK(hipEventCreateWithFlags(&ev_gathered_[i], hipEventDisableTiming));
    ensure_share_buffers();
  }
  reset_ = true;
}

void Renderer::ensure_share_buffers()
{
  if (width_ <= 0 || height_ <= 0) return;
  const Dist& d = Dist::get();
  const ShareLayout l = share_layout((uint32_t)width_, (uint32_t)height_, (uint32_t)d.world());
  share_world_ = (uint32_t)d.world(); share_rank_ = (uint32_t)d.rank(); share_n_local_ = l.n_local;
  VNR_HIP_CHECK(hipStreamSynchronize(stream_));
  VNR_HIP_CHECK(hipStreamSynchronize(Dist::get().comm_stream()));
  const size_t n = (size_t)width_ * height_;
  for (int i = 0; i < 2; ++i) {
    gathered_[i].resize((size_t)share_world_ * l.n_local);
    gathered_[i].zero(stream_);
    full_[i].resize(n);
    full_[i].zero(stream_);
    gather_issued_[i] = false;
  }
  if (accumulation_.count < l.n_local) { accumulation_.resize(l.n_local); accumulation_.zero(stream_); }  // a share of a tiny image is padded past it
  VNR_HIP_CHECK(hipStreamSynchronize(stream_));
}

void Renderer::issue_gather(int buf)
{
  if (gather_issued_[buf]) return;
  Dist& d = Dist::get();
  hipStream_t comm = d.comm_stream();
  // the frame's kernels: a streaming frame's parts have their completion events (the streams themselves may already hold the next
  // frame); any other mode runs on the render stream
  const int fslot = frame_of_buffer_[buf];
  if (fslot >= 0 && frame_[fslot]) {
    for (int h = 0; h < frame_[fslot]->H; ++h) VNR_HIP_CHECK(hipStreamWaitEvent(comm, frame_[fslot]->part[h].ev_done, 0));
  } else {
    VNR_HIP_CHECK(hipEventRecord(ev_rendered_, stream_));
    VNR_HIP_CHECK(hipStreamWaitEvent(comm, ev_rendered_, 0));
  }
  vec4f* g = gathered_[buf].ptr;
  const size_t share_bytes = (size_t)share_n_local_ * sizeof(vec4f);
  d.transport().all_gather(g + (size_t)share_rank_ * share_n_local_, g, share_bytes, comm);   // in place: the own slot is already there
  const uint32_t n = (uint32_t)((size_t)width_ * height_);
  assemble_shares_kernel<<<std::min<uint32_t>(div_round_up(n, 256), (uint32_t)Runtime::get().n_cus * 8u), 256, 0, comm>>>(
      g, full_[buf].ptr, (uint32_t)width_, (uint32_t)height_, share_world_, share_n_local_);
  VNR_HIP_CHECK(hipGetLastError());
  if (!skip_download_) VNR_HIP_CHECK(hipMemcpyAsync(host_fb_[buf], full_[buf].ptr, (size_t)n * sizeof(vec4f), hipMemcpyDeviceToHost, comm));
  VNR_HIP_CHECK(hipEventRecord(ev_gathered_[buf], comm));
  gather_issued_[buf] = true;
}

const float* Renderer::render_pipelined()
{
  // frame k + 1 is enqueued (its head before the host has seen frame k complete, render_streaming), which completes frame k;
  // frame k then travels (distributed: all-gather + assembly on the communication stream) while frame k + 1 renders
  const int cur = fb_cur_;
  const bool was_async = async_, was_skip = skip_download_;
  async_ = true;
  if (!distributed_) skip_download_ = true;   // (the host copy of a pipelined frame is made when it is handed out, below)
  try { render(); } catch (...) { async_ = was_async; skip_download_ = was_skip; throw; }
  async_ = was_async; skip_download_ = was_skip;
  const float* out = nullptr;
  const int prev = pipe_prev_;
  if (prev >= 0) {
    // frame k is complete unless this frame could not be pipelined (first frame after a change: render() completed k before)
    StreamingFrame* older = frame_[slot_ ^ 1].get();
    if (older && older->pending) finish_streaming(*older);
    out = hand_out_frame(prev);
  }
  pipe_prev_ = cur;
  fb_cur_ ^= 1;
  return out;
}

const float* Renderer::flush_pipeline()
{
  if (pipe_prev_ < 0) return nullptr;
  finish_pending();
  const int prev = pipe_prev_;
  pipe_prev_ = -1;
  return hand_out_frame(prev);
}

// the complete frame in buffer `buf` as the application sees it: assembled over the ranks (distributed), on the host unless the
// output is a device framebuffer
const float* Renderer::hand_out_frame(int buf)
{
  if (distributed_) {
    issue_gather(buf);
    VNR_HIP_CHECK(hipEventSynchronize(ev_gathered_[buf]));
    return skip_download_ ? (const float*)full_[buf].ptr : (const float*)host_fb_[buf];
  }
  if (skip_download_) return (const float*)fb_[buf].ptr;
  const size_t n = (size_t)width_ * height_;
  VNR_HIP_CHECK(hipMemcpy(host_fb_[buf], fb_[buf].ptr, n * sizeof(vec4f), hipMemcpyDeviceToHost));
  return (const float*)host_fb_[buf];
}

void Renderer::set_transfer_function(const TransferFunctionData& t)
{
  finish_pending();
  // api.cpp:485-498: the volume refreshes its macrocell max-opacity, the renderer keeps the lookup tables
  volume_->set_transfer_function(t, stream_);
  tfn_.set(t, volume_->desc.range_lo, volume_->desc.range_hi, stream_);
  reset_ = true;
}

void Renderer::ensure_queues(size_t n_pixels, int n_iters, bool gradient)
{
  if (queue_pixels_ >= n_pixels && queue_iters_ >= n_iters && (queue_grad_ || !gradient)) return;
  finish_pending();   // a pending frame lives in the buffers that are about to be replaced
  VNR_HIP_CHECK(hipDeviceSynchronize());
  const size_t P = std::max(n_pixels, queue_pixels_);
  const int iters = std::max(n_iters, queue_iters_);
  const bool grad = gradient || queue_grad_;
  // everything x 2: two frame slots (renderer.h)
  q_u32_.resize(2 * 6 * P);
  ray_counts_.resize(2 * (P / 64 + 64 + 8 * kMaxParts));   // survivors per 64-ray group (P is a multiple of 64; slack for the round-up to 256 rays)
  q_f32_.resize(2 * 18 * P);
  q_i32_.resize(2 * 6 * P);
  queue_.resize(2 * P * iters * (grad ? 4 : 1));          // 4 records per sample with gradient shading
  arena_.resize(2 * 2 * P * iters * (grad ? 6 : 2));      // x2 parities; 2 (+4) result floats per sample slot
  queue_pixels_ = P;
  queue_iters_ = iters;
  queue_grad_ = grad;
}

void Renderer::finish_pending()
{
  for (int k = 1; k <= 2; ++k) {   // the older slot first
    StreamingFrame* f = frame_[(slot_ + k) & 1].get();
    if (f && f->pending) finish_streaming(*f);
  }
}

void Renderer::render()
{
  // an asynchronous single-pass frame that is still pending stays pending until the head of this frame is enqueued
  // (render_streaming); everything else completes first
  const bool in_shader = in_shader_applies();
  const bool pipeline_head = async_ && (skip_download_ || distributed_) && (mode_ == 5 || mode_ == 6 || mode_ == 8 || mode_ == 9) && !in_shader &&
                             frame_[slot_] && frame_[slot_]->pending && !(frame_[slot_ ^ 1] && frame_[slot_ ^ 1]->pending) &&
                             !reset_;   // (a frame that restarts the accumulation overwrites it: its head must not run beside the tail of the frame before)
  if (!pipeline_head) finish_pending();
  if (width_ <= 0 || height_ <= 0) return;  // renderer.cpp:63
  MacroCell& mc = volume_->macrocell();
  if (!mc.allocated()) throw std::runtime_error("volume has no macrocell");
  RenderParams p;
  p.width = width_; p.height = height_;
  const uint32_t n_pixels = (uint32_t)((size_t)width_ * height_);
  p.pixel_lo = std::min(pixel_lo_, n_pixels);
  p.pixel_hi = std::min(pixel_hi_, n_pixels);
  if (p.pixel_hi < p.pixel_lo) p.pixel_hi = p.pixel_lo;
  // rays are generated in 8x8 pixel tiles; tile rows (8 scanlines) are the unit of multi-GPU interleaving
  if (distributed_) { il_block_ = 8u * (uint32_t)width_; il_parts_ = share_world_; il_part_ = share_rank_; }
  if (il_parts_ > 1 && il_block_ != 8u * (uint32_t)width_) throw std::runtime_error("pixel interleave block must be 8 scanlines (8 * width pixels)");
  p.il_parts = il_parts_; p.il_part = il_part_;
  p.out_parts = distributed_ ? share_world_ : 1u;
  p.tile_w_log2 = tile_w_log2_;
  p.tiles_per_row = div_round_up((uint32_t)width_, 1u << p.tile_w_log2) << (p.tile_w_log2 - 3u);
  const uint32_t tr_lo = p.pixel_hi > p.pixel_lo ? (p.pixel_lo / (uint32_t)width_) / 8u : 0u;
  const uint32_t tr_hi = p.pixel_hi > p.pixel_lo ? ((p.pixel_hi - 1u) / (uint32_t)width_) / 8u + 1u : 0u;
  p.tile_row0 = il_parts_ == 1 ? tr_lo : 0u;
  const uint32_t rows_local = il_parts_ == 1 ? (tr_hi - tr_lo) : div_round_up(div_round_up((uint32_t)height_, 8), il_parts_);
  p.n_local = rows_local * p.tiles_per_row * 64u;
  p.debug_flags = 0;
#if defined(VNR_DIAG)
  if (const char* e = std::getenv("VNR_AMD_DEBUG_FLAGS")) p.debug_flags = (uint32_t)std::atoi(e);
#endif
  static const float bin_depth = std::getenv("VNR_AMD_BIN_DEPTH") ? std::max(0.5f, (float)std::atof(std::getenv("VNR_AMD_BIN_DEPTH"))) : 8.0f;   // diagnostics (3 .. 8 measure the same)
  p.bin_depth_rcp = 1.0f / bin_depth;  // 8 world units (voxels) per depth bin ~ the footprint of an 8x8 pixel tile
  // camera, renderer.cpp:87-96
  const float t = 2.0f * tanf(camera_.fovy * 0.5f * (float)M_PI / 180.0f);
  const float aspect = (float)width_ / (float)height_;
  p.cam_pos = camera_.from;
  p.cam_dir = normalize(camera_.at - camera_.from);
  p.cam_hor = (t * aspect) * normalize(cross(p.cam_dir, camera_.up));
  p.cam_ver = (1.0f / aspect) * cross(p.cam_hor, p.cam_dir);
  p.wto = affine_inverse(volume_->transform);
  p.vol_dims = volume_->desc.dims;
  // dense data to sample: the ground-truth volume, or (decoding modes 4 / 7 on a neural volume) its decoded copy
  p.volume = volume_->is_network() ? static_cast<NeuralVolume*>(volume_.get())->decoded_data()
                                   : static_cast<SimpleVolume*>(volume_.get())->d_data();
  p.bbox_lo = volume_->clipbox.lower; p.bbox_hi = volume_->clipbox.upper;
  // What the kernels cannot survive is refused here, for all of them: a ray whose direction is NaN in every component passes the slab
  // test (fminf / fmaxf drop NaNs) with t in [0, 1e30] and its DDA never advances -- a frame that does not end.  from == at, an up vector
  // along the view, a NaN field of view, a singular volume transform or a NaN clipping box all produce such rays.  (The reference
  // renders garbage or hangs; an interactive host sends these while its user drags a slider.)
  {
    auto finite3 = [](vec3f v) { return std::isfinite(v.x) && std::isfinite(v.y) && std::isfinite(v.z); };
    if (!finite3(p.cam_pos) || !finite3(p.cam_dir) || !finite3(p.cam_hor) || !finite3(p.cam_ver))
      throw std::runtime_error("degenerate camera: position, focus, up vector and field of view do not span an image plane "
                               "(from == at, up along the view direction, or a value that is not finite)");
    if (!finite3(p.wto.vx) || !finite3(p.wto.vy) || !finite3(p.wto.vz) || !finite3(p.wto.p))
      throw std::runtime_error("degenerate volume transform: the object-to-world matrix is singular or not finite");
    if (!finite3(p.bbox_lo) || !finite3(p.bbox_hi)) throw std::runtime_error("clipping box is not finite");
    if (!std::isfinite(density_scale_)) throw std::runtime_error("volume density scale is not finite");
  }
  p.step = 1.0f / sampling_rate_; p.step_rcp = sampling_rate_;  // object.cpp:303-304
  p.mc_dims = mc.dims();
  const vec3f sp = mc.spacings();
  p.mc_rcp = {1.0f / sp.x, 1.0f / sp.y, 1.0f / sp.z};
  p.mc_max_opacity = mc.d_max_opacity();
  p.tfn = tfn_.view();
  p.tfn_in_lds = ((size_t)p.tfn.n_colors * sizeof(vec4f) + (size_t)p.tfn.n_alphas * sizeof(float)) <= 24 * 1024 ? 1u : 0u;
  // A small SHARE of a frame (one rank of 8) is bound by the latency of the per-iteration kernel chain, not by throughput:
  // fewer, longer iterations (tools/share_probe.py, 1/8 of the bench frame: 24 -> 0.995 ms, 32 -> 0.910 ms, 48 -> 0.905 ms in round 1; round 5: 1/8 share
  // 24 -> 0.592, 32 -> 0.536, 40 -> 0.578, 48 -> 0.61 ms; 1/4 share 24 -> 1.028, 32 -> 1.006 ms; 1/3 share 24 -> 1.323, 32 -> 1.338 ms: 32 up to a quarter
  // of the bench frame, the same bound as the three ray parts of a small share).
  // Only for shares (distributed mode or a pixel interleave), so that an unsharded small framebuffer keeps the default; and the
  // batch size moves a few samples by an ulp (a ray interrupted inside a macrocell resumes at t_min + (t - t_min), as in the
  // reference), so a frame assembled from such shares equals the unsharded frame rendered with VNR_RM_N_ITERS=32 bit for bit and
  // the unsharded frame at the default 24 to ~4e-5 on 0.2 % of the pixels (tests/test_gpu_fullsize.py); VNR_RM_N_ITERS pins both.
  p.n_iters = (!n_iters_fixed_ && (distributed_ || il_parts_ > 1) && p.n_local <= 262144u) ? 32 : n_iters_;
  // A march block stages its batch in LDS: 8 bytes per sample, 10 with the depth sort's ranks (renderer.h march_ranks_: dropped by default
  // in round 5, the slot inside a bin is claimed from the bin's counter when the record is written; same frames)
  p.no_ranks = march_ranks_ ? 0u : 1u;
  // gradient shading (modes 7 / 8)
  p.otw = volume_->transform;
  p.grad_step = {1.0f / (float)p.vol_dims.x, 1.0f / (float)p.vol_dims.y, 1.0f / (float)p.vol_dims.z};  // object.cpp:305
  if (dot(p.cam_dir, light_dir_) > 0.0f) light_dir_ = -1.0f * light_dir_;  // renderer.cpp:98-101: flipped in place, every frame
  p.light_dir = light_dir_;
  p.shading_mode = (mode_ == 7 || mode_ == 8 || mode_ == 9) ? 1u : (mode_ == 10 || mode_ == 11 || mode_ == 12) ? 2u : 0u;
  p.grad_flip = mode_ == 9 ? 1u : 0u;
  p.pt_reset_interval = mode_ == 15 ? 1u : 0u;
  p.ssh_third_draw = mode_ == 12 ? 1u : 0u;
  p.slot_cap = 0;
  // single-shade heuristic (modes 10 / 11): shadow rays point towards the light, xfmVector(wto, normalize(dir)) (:649, :473)
  p.shadow_dir = xfm_vector(p.wto, normalize(light_dir_));
  p.density_scale = density_scale_;
  p.px_org = nullptr; p.px_color = nullptr; p.px_alpha = nullptr; p.px_shading = nullptr; p.px_jitter = nullptr;
  if (mode_ == 11 || mode_ == 12) {  // per-pixel hand-over between the two passes (final_highest_*, shading_color, jitter_ssh; :88-92)
    ssh_px_.ensure(12 * (size_t)n_pixels);
    float* b = ssh_px_.ptr;
    p.px_org = (vec3f*)b; p.px_color = (vec3f*)(b + 3 * (size_t)n_pixels); p.px_alpha = b + 6 * (size_t)n_pixels;
    p.px_jitter = b + 7 * (size_t)n_pixels; p.px_shading = (vec4f*)(b + 8 * (size_t)n_pixels);
  }
  // frame index / accumulation, renderer.cpp:103-105
  if (reset_) frame_index_ = 0;
  ++frame_index_;
  p.frame_index = frame_index_;
  p.frame = fb_[fb_cur_].ptr;
  if (distributed_) {
    // the buffer still feeds the gather of the frame rendered into it two frames ago (a no-op once that has completed)
    if (gather_issued_[fb_cur_]) VNR_HIP_CHECK(hipStreamWaitEvent(stream_, ev_gathered_[fb_cur_], 0));
    gather_issued_[fb_cur_] = false;
    p.frame = gathered_[fb_cur_].ptr + (size_t)share_rank_ * share_n_local_;
  }
  p.accumulation = accumulation_.ptr;
  if (!pipeline_head) stats_ = FrameStats();   // (pipelined: reset by render_streaming once the frame before this one has completed)

  frame_of_buffer_[fb_cur_] = -1;
  if (p.pixel_hi > p.pixel_lo && p.n_local > 0) {
    if (in_shader) {
      // VNR_RAYMARCHING_{NO_SHADING, GRADIENT_SHADING, SINGLE_SHADE_HEURISTIC}_IN_SHADER on a neural volume: the network inside the
      // marching loop, one launch per frame (in_shader.h; method_raymarching.cu:981-1249)
      // (modes 14 / 15, path tracing: in_shader_pt_kernel, method_pathtracing.cu:968-1025; the two differ by p.pt_reset_interval)
      render_in_shader(p, mode_ == 9 ? M_GRADIENT : mode_ == 12 ? M_SSH : (mode_ == 14 || mode_ == 15) ? kInShaderPathTracing : M_NONE);
    } else
    switch (mode_) {
    case 6:   // VNR_RAYMARCHING_NO_SHADING_IN_SHADER where in_shader_applies() says no (a dense volume, a model shape without an
              // in-shader instance, VNR_AMD_IN_SHADER=0): the reference evaluates the network inside the marching loop
              // (network_raymarching_traceray / _iterator, method_raymarching.cu:310-356, 1037-1100): the per-ray arithmetic is
              // mode 5's without the interruptions, i.e. mode 5's up to the last bit of the samples at batch boundaries (a ray
              // resumes at t_min + (t - t_min)): measured 4e-5 at most on 0.2 % of the pixels, two orders below what the network's
              // own arithmetic differs by.  "In shader" is an execution strategy; here it is the streaming one.
    case 9:   // VNR_RAYMARCHING_GRADIENT_SHADING_IN_SHADER (:1068-1070): mode 8 with sampleGradient's boundary flip, uninterrupted
    case 5:   // VNR_RAYMARCHING_NO_SHADING_SAMPLE_STREAMING
    case 8:   // VNR_RAYMARCHING_GRADIENT_SHADING_SAMPLE_STREAMING
      // asynchronous frames (set_async): the pass is left pending after its predicted iterations have been enqueued
      render_streaming(p, p.shading_mode == 1u ? M_GRADIENT : M_NONE, async_ && (skip_download_ || distributed_));
      break;
    case 11:  // VNR_RAYMARCHING_SINGLE_SHADE_HEURISTIC_SAMPLE_STREAMING: camera pass, then one shadow ray per pixel (:968-971)
      render_streaming(p, M_SSH);
      render_streaming(p, M_SHADOW);
      break;
    case 15:  // VNR_PATHTRACING_IN_SHADER (network_path_tracing_traceray, :968-1025): mode 13's estimator, i.e. the streaming
              // loop with the interval reset before a bounce, sampled from whatever the volume is
    case 14:  // VNR_PATHTRACING_SAMPLE_STREAMING
      render_pathtracing(p);
      break;
    case 13:  // VNR_PATHTRACING_DECODING: the same estimator in one loop per pixel, on dense (or decoded) data
      if (!p.volume)
        throw std::runtime_error(volume_->is_network() ? "rendering mode 13 traces the decoded volume: call vnrNeuralVolumeDecodeProgressive first (GetNumberOfBlobs calls = one full pass)"
                                                       : "this volume has no resident data to sample");
      pt_monolithic_kernel<<<div_round_up(p.n_local, 128), 128, 0, stream_>>>(p);
      VNR_HIP_CHECK(hipGetLastError());
      break;
    case 12: {  // VNR_RAYMARCHING_SINGLE_SHADE_HEURISTIC_IN_SHADER (network_raymarching_traceray / _transmittance, :981-1035,
                // 1037-1128): the uninterrupted camera march of mode 11, then a shadow ray at raymarching_shadow_sampling_scale = 2 x
                // the step (opacity correction keeps the step), jittered by the pixel's third random number
      render_streaming(p, M_SSH);
      RenderParams q = p;
      q.step = 2.0f * p.step;
      render_streaming(q, M_SHADOW);
      break;
    }
    case 4:   // VNR_RAYMARCHING_NO_SHADING_DECODING
    case 7:   // VNR_RAYMARCHING_GRADIENT_SHADING_DECODING
    case 10:  // VNR_RAYMARCHING_SINGLE_SHADE_HEURISTIC_DECODING
      // vnrRequireDecoding(mode) (api.h:62-88): the application decodes, vnrNeuralVolumeDecodeProgressive x GetNumberOfBlobs,
      // and these modes march whatever the decoded volume holds
      if (volume_->is_network() && !p.volume)
        throw std::runtime_error("rendering mode " + std::to_string(mode_) +
                                 " marches the decoded volume: call vnrNeuralVolumeDecodeProgressive first (GetNumberOfBlobs calls = one full pass)");
      render_monolithic(p);
      break;
    default:
      throw std::runtime_error("rendering mode " + std::to_string(mode_) +
                               " is not implemented in this build (supported: ray marching 4 - 12 and path tracing 13 - 15; 0 - 3 are the OptiX modes)");
    }
  }
  reset_ = false;
  if (!(frame_[slot_] && frame_[slot_]->pending)) completed_stats_ = stats_;   // every mode but a deferred streaming frame is complete here
  if (!skip_download_ && !distributed_) {  // renderer.cpp:133 framebuffer.download_async (distributed: the assembled frame, issue_gather)
    const size_t off = p.pixel_lo, cnt = p.pixel_hi - p.pixel_lo;  // (interleaved shares copy the covering range)
    if (cnt) VNR_HIP_CHECK(hipMemcpyAsync(host_fb_[fb_cur_] + off, fb_[fb_cur_].ptr + off, cnt * sizeof(vec4f), hipMemcpyDeviceToHost, stream_));
  }
}

// the in-shader kernels, one translation unit per FullyFusedMLP width and kind (in_shader_w*.hip); -> false: no instance for the shape
bool launch_in_shader_w16(const RenderParams&, const TileNet&, int, const InShaderLaunch&);
bool launch_in_shader_w32(const RenderParams&, const TileNet&, int, const InShaderLaunch&);
bool launch_in_shader_w64(const RenderParams&, const TileNet&, int, const InShaderLaunch&);
bool launch_in_shader_w128(const RenderParams&, const TileNet&, int, const InShaderLaunch&);
bool launch_in_shader_w16g(const RenderParams&, const TileNet&, int, const InShaderLaunch&);
bool launch_in_shader_w32g(const RenderParams&, const TileNet&, int, const InShaderLaunch&);
bool launch_in_shader_w64g(const RenderParams&, const TileNet&, int, const InShaderLaunch&);

bool Renderer::in_shader_applies() const
{
  if (mode_ != 6 && mode_ != 9 && mode_ != 12 && mode_ != 14 && mode_ != 15) return false;
  // Which execution strategy, where both exist and give the same frames (profiles/r02_in_shader_vs_streaming.txt, bench frame):
  //  * ray marching (6 / 9 / 12): the in-shader kernel takes 8.2 ms for the whole frame and 1.98 ms for a 1/8 share where the
  //    streaming path takes 4.2 and 0.70 ms (mode 9: 36.8 / 6.6 against 12.2 / 1.8): a wave marches until its longest ray has ended,
  //    which the streaming path's compaction removes.  Default: streaming.
  //  * path tracing (14 / 15): one evaluation per ray and trip, 60-odd dependent trips per frame: 5.5 ms in one launch against 6.7-7.0 ms
  //    as a chain of 3 launches per trip, and the frames are equal bit for bit.  Default: in shader.
  // vnrAmdRendererSetInShaderKernel(0 | 1) or VNR_AMD_IN_SHADER=0 | 1 force one or the other.
  static const int env = [] { const char* e = std::getenv("VNR_AMD_IN_SHADER"); return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1; }();
  const int choice = in_shader_mode_ >= 0 ? in_shader_mode_ : env;
  const bool want = choice >= 0 ? choice == 1 : (mode_ == 14 || mode_ == 15);
  if (!want || !volume_->is_network()) return false;
  const Network& net = static_cast<NeuralVolume*>(volume_.get())->network();
  // every width and kind of model has an instance (round 5) as long as its weight image fits the LDS beside the kernel's 48 static bytes
  // (deeper networks keep their weights in global memory and take the streaming path) and its encoding is one of VNR_IN_SHADER_SHAPES
  if (!net.valid() || !net.weights_in_lds() || (size_t)net.lds_halves() * 2 + 64 > 160 * 1024) return false;
  // (128 neurons: the common kind only.  The reference refuses 128 neurons in shader altogether, method_raymarching.cu:1210; the GENERAL instances
  // of that width alone were a fifth of the library's build time)
  if (net.width() == 128u && !net.common_kind()) return false;
  const uint32_t F = net.config().n_features, K = net.padded_width();
#define X(f, k) if (F == f && K == k) return true;
  VNR_IN_SHADER_SHAPES(X)
#undef X
  return false;
}

void Renderer::render_in_shader(const RenderParams& p, int shade)
{
  NeuralVolume* nv = static_cast<NeuralVolume*>(volume_.get());
  TileNet net;
  if (!nv->network().tile_net(&net, stream_)) throw std::runtime_error("internal: in-shader rendering of a model without an MFMA kernel");
  if (is_samples_.count == 0) { is_samples_.resize(kInShaderStatSlots); is_hits_.resize(kInShaderStatSlots); }
  is_samples_.zero(stream_);
  is_hits_.zero(stream_);
  InShaderLaunch l;
  l.blocks = div_round_up(p.n_local, 256);
  l.shmem = (size_t)net.lds_halves * sizeof(uint16_t);
  l.stream = stream_;
  l.stat_samples = is_samples_.ptr;
  l.stat_hits = is_hits_.ptr;
  bool launched = false;
  switch (net.width) {
  case 16: launched = net.general ? launch_in_shader_w16g(p, net, shade, l) : launch_in_shader_w16(p, net, shade, l); break;
  case 32: launched = net.general ? launch_in_shader_w32g(p, net, shade, l) : launch_in_shader_w32(p, net, shade, l); break;
  case 64: launched = net.general ? launch_in_shader_w64g(p, net, shade, l) : launch_in_shader_w64(p, net, shade, l); break;
  case 128: launched = !net.general && launch_in_shader_w128(p, net, shade, l); break;   // (128 neurons of the GENERAL kind: streaming path, in_shader_applies)
  default: break;
  }
  if (!launched) throw std::runtime_error("internal: no in-shader instance for this model shape");
  VNR_HIP_CHECK(hipGetLastError());
  unsigned long long hs[kInShaderStatSlots];
  uint32_t hh[kInShaderStatSlots];
  VNR_HIP_CHECK(hipMemcpyAsync(hs, is_samples_.ptr, sizeof(hs), hipMemcpyDeviceToHost, stream_));
  VNR_HIP_CHECK(hipMemcpyAsync(hh, is_hits_.ptr, sizeof(hh), hipMemcpyDeviceToHost, stream_));
  VNR_HIP_CHECK(hipStreamSynchronize(stream_));
  for (int k = 0; k < kInShaderStatSlots; ++k) { stats_.n_samples += hs[k]; stats_.n_rays_hit += hh[k]; }
  stats_.n_reference_slots = stats_.n_samples;
  stats_.n_iterations = 1;
}

void Renderer::render_monolithic(const RenderParams& p)
{
  monolithic_kernel<<<div_round_up(p.n_local, 128), 128, 0, stream_>>>(p);
  VNR_HIP_CHECK(hipGetLastError());
}

void Renderer::render_pathtracing(const RenderParams& p)
{
  // do_path_tracing_iterative (method_pathtracing.cu:786-806) without its per-iteration D2H + sync: the iterations the
  // previous frame needed are enqueued at once, then the alive-ray count is looked at every few iterations
  const uint32_t P = p.n_local;
  NeuralVolume* nv = volume_->is_network() ? static_cast<NeuralVolume*>(volume_.get()) : nullptr;
  if (nv && !nv->network().valid()) throw std::runtime_error("neural volume has no valid network");
  if (!nv && !p.volume) throw std::runtime_error("this volume has no resident data to sample");
  pt_rays_.ensure((size_t)2 * kPtPlanes * P);
  pt_values_.ensure(P);
  if (queue_.count < P) queue_.resize(P);
  ray_counts_.ensure(P / 64 + 64);
  const PtRays dense = {pt_rays_.ptr, P}, scratch = {pt_rays_.ptr + (size_t)kPtPlanes * P, P};
  uint32_t* c = counters_.ptr;
  hipStream_t s = stream_;
  VNR_HIP_CHECK(hipMemsetAsync(c, 0, C_COUNT * sizeof(uint32_t), s));
  const uint32_t blocks = std::min<uint32_t>(div_round_up(P, 256), 4096u);
  const uint32_t cblocks = div_round_up(P, 1024);
  uint32_t it = 0;
  auto launch = [&]() {
    const int parity = (int)(it & 1u);
    if (it == 0) {
      pt_kernel<true><<<blocks, 256, 0, s>>>(p, dense, scratch, pt_values_.ptr, c, ray_counts_.ptr, 0);
    } else {
      // values of the alive rays' sample points: the count is the one the last packing step published
      if (nv) nv->network().inference_queue((const float*)queue_.ptr, pt_values_.ptr, 1, c + C_RAYS0 + parity, P, s, 1);
      else gt_sample_kernel<<<std::min<uint32_t>(div_round_up(P, 256), (uint32_t)Runtime::get().n_cus * 8u), 256, 0, s>>>(
               c + C_RAYS0 + parity, p.volume, p.vol_dims, queue_.ptr, pt_values_.ptr);
      pt_kernel<false><<<blocks, 256, 0, s>>>(p, dense, scratch, pt_values_.ptr, c, ray_counts_.ptr, parity);
    }
    pt_compact_kernel<<<cblocks, 1024, 0, s>>>(scratch, dense, ray_counts_.ptr, P, c, parity, it == 0 ? 1 : 0, queue_.ptr);
    VNR_HIP_CHECK(hipGetLastError());
    VNR_HIP_CHECK(hipMemcpyAsync(host_counts_ + (it & 255u), c + C_RAYS0 + (parity ^ 1), sizeof(uint32_t), hipMemcpyDeviceToHost, s));
    ++it;
  };
  const uint32_t max_iterations = 1u << 16;
  while (it < std::max<uint32_t>(predicted_pt_, 1u)) launch();
  for (;;) {
    VNR_HIP_CHECK(hipStreamSynchronize(s));
    if (host_counts_[(it - 1) & 255u] == 0 || it >= max_iterations) break;
    for (int k = 0; k < 8; ++k) launch();
  }
  // iterations that had rays to shade = index of the first packing step that published zero
  uint32_t used = it;
  if (it <= 256) { while (used > 1 && host_counts_[(used - 2) & 255u] == 0) --used; }
  predicted_pt_ = used;
  uint32_t hc[C_COUNT];
  VNR_HIP_CHECK(hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost));
  stats_.n_rays_hit += hc[C_HIT];
  const uint64_t n = (uint64_t)hc[C_STAT_SAMPLES] | ((uint64_t)hc[C_STAT_SAMPLES + 1] << 32);
  stats_.n_samples += n;
  stats_.n_reference_slots += n;
  stats_.n_iterations += used > 0 ? used - 1 : 0;
}

// one iteration of one half: march(it) -> evaluate the compacted samples -> clear the counters march(it+1) appends to
void Renderer::launch_iteration(StreamingFrame& f, int h)
{
  PartState* half = f.part;
  const int pass_mode = f.pass_mode;
  const size_t shmem = f.shmem, shmem_compose = f.shmem_compose;
  const uint32_t max_iterations = f.max_iterations;
  {
    PartState& hf = half[h];
    const uint32_t it = hf.it;
    const int parity = (int)(it & 1u);
    const uint32_t P = hf.p.n_local;
    uint32_t* c = hf.c;
    const hipStream_t s_it = hf.s;
    // The march the previous frame ended with (it composes the last batch and finds no ray left to sample) is launched WITHOUT an
    // evaluation and a packing kernel behind it: both would be empty, and on a small share of the frame they are two more launches
    // on a chain of fifteen.  The march says in pinned memory whether a ray did survive; finish_streaming then launches the two
    // kernels after all (launch_tail) and goes on as if they had been there.  VNR_AMD_TAIL_SKIP=0 switches this off (diagnostics).
    static const bool tail_skip = [] { const char* e = std::getenv("VNR_AMD_TAIL_SKIP"); return !e || std::atoi(e) != 0; }();
    const bool skip_tail = tail_skip && it >= 1 && it + 1 == f.predicted[h] && it + 1 < max_iterations;
    uint32_t* tail_flag = nullptr;
    if (skip_tail) { hf.hs[0] = 0; tail_flag = hf.hs; }   // hs[0 .. C_HIT) is not used by the packing kernel's statistics
    // march(it): reads the dense ray list rl[0] (count: counter `parity`), leaves the survivors of every 64-ray group in the
    // group's slots of the scratch list rl[1] and appends their samples to queue `parity`
    {
      const bool first = it == 0;
      const uint32_t blocks = std::min<uint32_t>(div_round_up(P, 256), first ? 4096u : 2048u);
      const size_t lds = first ? shmem : shmem_compose;
      const vec2f* vd_in = hf.vd[parity ^ 1];
      vec2f* vd_out = hf.vd[parity];
#define VNR_MARCH(FIRST_, MODE_) march_kernel<FIRST_, MODE_><<<blocks, 256, lds, s_it>>>(hf.p, hf.rl[0], hf.rl[1], vd_in, hf.queue, vd_out, c, hf.rc, parity, hf.ssh, tail_flag)
      switch (pass_mode) {
      case M_GRADIENT: if (first) VNR_MARCH(true, M_GRADIENT); else VNR_MARCH(false, M_GRADIENT); break;
      case M_SSH: if (first) VNR_MARCH(true, M_SSH); else VNR_MARCH(false, M_SSH); break;
      case M_SHADOW: if (first) VNR_MARCH(true, M_SHADOW); else VNR_MARCH(false, M_SHADOW); break;
      default: if (first) VNR_MARCH(true, M_NONE); else VNR_MARCH(false, M_NONE); break;
      }
#undef VNR_MARCH
    }
    VNR_HIP_CHECK(hipGetLastError());
    if (skip_tail) {
      if (profiling_) { VNR_HIP_CHECK(hipEventRecord(events_[f.slot][h][2 * it], hf.s)); VNR_HIP_CHECK(hipEventRecord(events_[f.slot][h][2 * it + 1], hf.s)); }
      hf.tail_skipped = true;
      ++hf.it;
      if (hf.it >= max_iterations) hf.done = true;
      return;
    }
  }
  {
    PartState& hf = half[h];
    const uint32_t it = hf.it;
    launch_tail(f, h, it, hf.s);
    ++hf.it;
    if (hf.it >= max_iterations) hf.done = true;
  }
}

// evaluation of the samples march(it) emitted, then the packing of its survivors
void Renderer::launch_tail(StreamingFrame& f, int h, uint32_t it, hipStream_t s_it)
{
  PartState* half = f.part;
  const int H = f.H;
  const bool grad = f.grad, ssh = f.ssh;
  NeuralVolume* nv = f.nv;
  {
    PartState& hf = half[h];
    const int parity = (int)(it & 1u);
    const uint32_t P = hf.p.n_local;
    uint32_t* c = hf.c;
    // the packing of march(it)'s survivors into rl[0] in group order (count -> counter `parity^1`; clears the sample counter of march(it+1)):
    // as a prologue of the evaluation kernel when that is one of the MFMA kernels, else a kernel of its own behind it
    PackArgs pk;
    pk.src = hf.rl[1]; pk.dst = hf.rl[0]; pk.ray_counts = hf.rc; pk.n_first = P; pk.counters = c; pk.parity = parity; pk.first = it == 0 ? 1 : 0;
    pk.ssh = ssh ? 1 : 0; pk.grad = grad ? 1 : 0; pk.ssh_lists = hf.ssh; pk.host_alive = hf.hc + (it & 255u); pk.host_stats = hf.hs;
    pk.n_blocks = div_round_up(P, 256);
    // Only where the launch it saves is on the critical path, i.e. a part of at most 262 144 rays (a 1/2 .. 1/8 share of the bench frame):
    // a whole frame's packing kernel runs under the other part's evaluation kernel, the frame takes the same time either way, and the
    // evaluation kernel's own time (what bench.py's roofline is computed from) would carry the packing's 15-20 us per launch.
    // VNR_AMD_FUSED_PACK=0 / 2: never / always (diagnostics).
    static const int fused_pack_mode = [] { const char* e = std::getenv("VNR_AMD_FUSED_PACK"); return e ? std::atoi(e) : 1; }();
    const bool fused_pack = fused_pack_mode == 2 || (fused_pack_mode == 1 && P <= 262144u);
    bool packed = false;
    if (!s_it) s_it = hf.s;
    if (profiling_) VNR_HIP_CHECK(hipEventRecord(events_[f.slot][h][2 * it], s_it));
#if defined(VNR_DIAG)
    static const bool skip_eval = [] { const char* e = std::getenv("VNR_AMD_DEBUG_SKIP_EVAL"); return e && std::atoi(e) != 0; }();   // timing of the march / packing chain alone (frames are garbage)
#else
    constexpr bool skip_eval = false;
#endif
    if (skip_eval) {
    } else if (nv) {
      // a record's 4th word is the float index of its result in this arena (stride 1)
      packed = nv->network().inference_queue((const float*)hf.queue, (float*)hf.vd[parity], 1, c + C_SAMPLES0 + parity, hf.s_max, s_it, (uint32_t)H,
                                             fused_pack ? &pk : nullptr);
    } else {
      const uint32_t blocks = std::min<uint32_t>(div_round_up(hf.s_max, 256), (uint32_t)Runtime::get().n_cus * 8u);
      gt_sample_kernel<<<blocks, 256, 0, s_it>>>(c + C_SAMPLES0 + parity, hf.p.volume, hf.p.vol_dims, hf.queue, (float*)hf.vd[parity]);
    }
    if (profiling_) VNR_HIP_CHECK(hipEventRecord(events_[f.slot][h][2 * it + 1], s_it));
    last_schedule_[2] = packed ? 1 : 0;
    if (!packed) {
      // 1024-thread blocks need 4 free wave slots on every SIMD of one CU at once, which the other half's evaluation kernel
      // rarely leaves: a small share (where the wait shows) packs with 256-thread blocks
      static const uint32_t small_limit = [] { const char* e = std::getenv("VNR_AMD_COMPACT_SMALL_LIMIT"); return e ? (uint32_t)std::atoll(e) : 262144u; }();  // diagnostics
      if (P <= small_limit) {
        compact_rays_kernel<4><<<pk.n_blocks, 256, 0, s_it>>>(pk);
      } else {
        pk.n_blocks = div_round_up(P, 1024);
        compact_rays_kernel<16><<<pk.n_blocks, 1024, 0, s_it>>>(pk);
      }
    }
    VNR_HIP_CHECK(hipGetLastError());
  }
}

void Renderer::render_streaming(const RenderParams& p_all, int pass_mode, bool defer)
{
  if (decoupled_applies(p_all, pass_mode)) { render_decoupled(p_all, defer); return; }
  // The rank's rays are dealt to `n_halves_` independent halves (alternate local tile rows; same mechanism as the
  // multi-GPU interleave), each with its own ray lists, sample queue, counters and HIP stream.  The arithmetic per ray
  // is untouched; what changes is that march(i) of one half runs while the other half's inference kernel does: the
  // inference kernel is bound by fetched bytes and resident with 2 blocks per CU, the march kernel is latency bound
  // and fits next to it (registers and LDS: DESIGN.md 4.2).
  const uint32_t row_items = p_all.tiles_per_row * 64u;
  const uint32_t R = p_all.n_local / row_items;  // local tile rows
  // a small share is bound by the length of the kernel chain of one part, not by throughput: more, shorter chains side by side
  // (three parts: the library's stream and the pool's two, which Runtime::init creates FIRST, so they own three of the runtime's four hardware
  // queues whatever streams come later -- a rank's communication stream, an out-of-core sampler's, the application's: common.h)
  const int want = (!n_halves_fixed_ && p_all.n_local <= 524288u) ? small_share_parts_ : n_halves_;   // (half the bench frame and less: renderer.h)
  const int H = (int)std::min<uint32_t>((uint32_t)want, std::max(R, 1u));
  const uint32_t P_total = p_all.n_local;
  const bool grad = pass_mode == M_GRADIENT;
  const bool ssh = pass_mode == M_SSH;
  uint32_t* predicted = predicted_iterations_[pass_mode == M_SHADOW ? 1 : 0];
  ensure_queues(P_total, p_all.n_iters, grad || ssh || pass_mode == M_SHADOW);   // M_SSH keeps a sample's t where GRAD keeps f(c + g)
  const size_t QP = queue_pixels_;
  if (ssh) q_ssh_.ensure(14 * QP);
  const size_t QI = (size_t)queue_iters_;
  const size_t slot_floats = queue_grad_ ? 6 : 2;   // result floats per sample slot as ALLOCATED (a slice is laid out per mode)
  const size_t rec_per_slot = queue_grad_ ? 4 : 1;
  if ((size_t)P_total * QI * 6 >= (1ull << 32)) throw std::runtime_error("frame share too large for 32-bit result indices");
  NeuralVolume* nv = volume_->is_network() ? static_cast<NeuralVolume*>(volume_.get()) : nullptr;
  if (nv && !nv->network().valid()) throw std::runtime_error("neural volume has no valid network");

  // the slot the previous frame does not occupy if that one is still pending (asynchronous frames), else the same slot again
  StreamingFrame* older = frame_[slot_] && frame_[slot_]->pending ? frame_[slot_].get() : nullptr;
  const int slot = older ? slot_ ^ 1 : slot_;
  if (!frame_[slot]) frame_[slot].reset(new StreamingFrame());
  StreamingFrame& f = *frame_[slot];
  if (f.pending) throw std::runtime_error("internal: both frame slots are pending");
  f.slot = slot;
  slot_ = slot;
  f.decoupled = false;   // (the slot's previous frame may have run the decoupled loop: finish_streaming dispatches on this)
  PartState* half = f.part;
  for (int h = 0; h < kMaxParts; ++h) { half[h].it = 0; half[h].used = 0; half[h].done = false; half[h].tail_skipped = false; }
  const size_t n_groups_slot = QP / 64 + 64 + 8 * kMaxParts;
  uint32_t* const u32_base = q_u32_.ptr + (size_t)slot * 6 * QP;
  float* const f32_base = q_f32_.ptr + (size_t)slot * 18 * QP;
  int* const i32_base = q_i32_.ptr + (size_t)slot * 6 * QP;
  uint32_t* const rc_base = ray_counts_.ptr + (size_t)slot * n_groups_slot;
  vec4f* const queue_base = queue_.ptr + (size_t)slot * rec_per_slot * QP * QI;
  float* const arena_base = arena_.ptr + (size_t)slot * 2 * slot_floats * QP * QI;
  uint32_t* const counters_base = counters_.ptr + (size_t)slot * kMaxParts * C_COUNT;
  uint32_t* const host_base = host_counts_ + (size_t)slot * kMaxParts * (256 + C_COUNT);
  size_t off = 0;
  for (int h = 0; h < H; ++h) {
    PartState& hf = half[h];
    hf.p = p_all;
    if (H > 1) {
      hf.p.il_parts = p_all.il_parts * (uint32_t)H;
      hf.p.il_part = p_all.il_part + p_all.il_parts * (uint32_t)h;
      hf.p.n_local = ((R + (uint32_t)(H - 1 - h)) / (uint32_t)H) * row_items;
    }
    for (int b = 0; b < 2; ++b) {
      hf.rl[b].pixel_index = u32_base + (size_t)(0 + b) * QP + off;
      hf.rl[b].sample_base = u32_base + (size_t)(2 + b) * QP + off;
      hf.rl[b].sample_count = u32_base + (size_t)(4 + b) * QP + off;
      float* f = f32_base + (size_t)b * 9 * QP;
      hf.rl[b].jitter = f + off;
      hf.rl[b].alpha = f + QP + off;
      hf.rl[b].color = (vec3f*)(f + 2 * QP) + off;
      hf.rl[b].t_next = (vec3f*)(f + 5 * QP) + off;
      hf.rl[b].next_cell_begin = f + 8 * QP + off;
      hf.rl[b].cell = (vec3i*)(i32_base + (size_t)b * 3 * QP) + off;
      hf.ssh.org[b] = nullptr; hf.ssh.color[b] = nullptr; hf.ssh.alpha[b] = nullptr;
      if (ssh) {
        float* g = q_ssh_.ptr + (size_t)b * 7 * QP;
        hf.ssh.org[b] = (vec3f*)g + off;
        hf.ssh.color[b] = (vec3f*)(g + 3 * QP) + off;
        hf.ssh.alpha[b] = g + 6 * QP + off;
      }
      // this half's result arena of parity b: a slice of slot_floats * n_local * QI floats
      hf.vd[b] = (vec2f*)(arena_base + (size_t)b * slot_floats * QP * QI + slot_floats * off * QI);
    }
    hf.p.slot_cap = (uint32_t)((size_t)hf.p.n_local * QI);   // 2 (+ 4 with gradient shading) floats per slot fit the slice
    hf.queue = queue_base + rec_per_slot * off * QI;
    hf.c = counters_base + (size_t)h * C_COUNT;
    hf.rc = rc_base + off / 64 + (size_t)h * 8;   // halves are multiples of 64 rays; 8 groups of slack each
    hf.hc = host_base + (size_t)h * 256;
    hf.hs = host_base + kMaxParts * 256 + (size_t)h * C_COUNT;
    if (h > 0 && !part_streams_[h]) part_streams_[h] = Runtime::get().part_stream(h);
    hf.s = h == 0 ? stream_ : part_streams_[h];
    hf.s_max = (size_t)hf.p.n_local * hf.p.n_iters * (grad ? 4 : 1);   // records the evaluation kernel may see
    off += hf.p.n_local;
  }

  // (the head of a pipelined frame runs on the part streams, behind the frame before it: a head on streams of its own, released behind that
  // frame's last large evaluation, measured slower twice and was removed in round 4: docs/history/DESIGN_r01-r03.md 4.2b)
  f.params_generation = nv ? nv->network().params_generation() : 0;
  VNR_HIP_CHECK(hipMemsetAsync(counters_base, 0, (size_t)H * C_COUNT * sizeof(uint32_t), stream_));
  if (H > 1) {  // fork: the other streams start after everything queued on the render stream so far
    VNR_HIP_CHECK(hipEventRecord(ev_fork_, stream_));
    for (int h = 1; h < H; ++h) VNR_HIP_CHECK(hipStreamWaitEvent(part_streams_[h], ev_fork_, 0));
  }
  const size_t shmem = ((size_t)2 * p_all.n_iters + 1) * 256 * sizeof(float) + 16 * sizeof(uint32_t) + (p_all.no_ranks ? 0 : (size_t)p_all.n_iters * 256 * sizeof(uint16_t));
  const size_t shmem_compose = shmem + (p_all.tfn_in_lds ? (size_t)p_all.tfn.n_colors * sizeof(vec4f) + (size_t)p_all.tfn.n_alphas * sizeof(float) : 0);
  if (shmem_compose > 160 * 1024) throw std::runtime_error("VNR_RM_N_ITERS too large for the LDS of one workgroup");
  static bool lds_attr_set = false;
  if (!lds_attr_set) {  // more than the default 64 KiB of dynamic LDS needs an opt-in
#define VNR_ATTR(FIRST_, MODE_) VNR_HIP_CHECK(hipFuncSetAttribute((const void*)march_kernel<FIRST_, MODE_>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024))
    VNR_ATTR(true, M_NONE); VNR_ATTR(false, M_NONE); VNR_ATTR(true, M_GRADIENT); VNR_ATTR(false, M_GRADIENT);
    VNR_ATTR(true, M_SSH); VNR_ATTR(false, M_SSH); VNR_ATTR(true, M_SHADOW); VNR_ATTR(false, M_SHADOW);
#undef VNR_ATTR
    lds_attr_set = true;
  }
  uint32_t max_iterations = 240;
#if defined(VNR_DIAG)
  if (const char* e = std::getenv("VNR_AMD_DEBUG_MAX_ITERS")) max_iterations = std::max(1, std::min(240, std::atoi(e)));  // truncated (wrong) frames, for timing
#endif
  if (profiling_) {
    iter_ms_.assign(max_iterations, 0.0f);
    for (int h = 0; h < H; ++h)
      while (events_[slot][h].size() < 2 * max_iterations) { hipEvent_t e; VNR_HIP_CHECK(hipEventCreate(&e)); events_[slot][h].push_back(e); }
  }

  last_schedule_[0] = p_all.n_iters; last_schedule_[1] = H; last_schedule_[2] = 0; last_schedule_[3] = 0;
  f.H = H; f.pass_mode = pass_mode; f.grad = grad; f.ssh = ssh; f.nv = nv; f.shmem = shmem; f.shmem_compose = shmem_compose;
  f.max_iterations = max_iterations; f.predicted = predicted; f.p_all = p_all;
  // phase 0 (asynchronous frames with the frame before still pending): the HEAD of this frame, i.e. iteration 0 of every part:
  // ray generation, the first batch of samples, their evaluation and the packing.  The only pixels it writes are those of rays
  // that miss the volume or find nothing to sample, which contribute zero in every frame, so it may run before the frame
  // before it has written its last pixel; then that frame is completed (host), and only then come the iterations that compose.
  if (older) {
    for (int h = 0; h < H; ++h) launch_iteration(f, h);
    finish_streaming(*older);
    stats_ = FrameStats();
  }
  frame_of_buffer_[fb_cur_] = slot;
  // phase 1: the iterations the previous frame needed, launched for both halves alternately without any host sync;
  // phase 2: past that, look at the alive-ray count of the iteration just launched before launching another one.
  for (;;) {
    bool launched = false;
    for (int h = 0; h < H; ++h)
      if (!half[h].done && half[h].it < predicted[h]) { launch_iteration(f, h); launched = true; }
    if (!launched) break;
  }
  for (int h = 0; h < H; ++h) f.mark(h);
  f.pending = true;
  if (!defer) finish_streaming(f);
}

// phase 2 of a pass and its bookkeeping: past the predicted iterations, look at the alive-ray count of the iteration just
// launched before launching another one; then the statistics.  Ends the "pending" state of an asynchronous frame.
void Renderer::finish_streaming(StreamingFrame& f)
{
  if (!f.pending) return;
  if (f.decoupled) { finish_decoupled(f); return; }
  f.pending = false;
  PartState* half = f.part;
  const int H = f.H, pass_mode = f.pass_mode;
  uint32_t* predicted = f.predicted;
  const RenderParams& p_all = f.p_all;
  for (;;) {
    bool pending = false;
    for (int h = 0; h < H; ++h) {
      PartState& hf = half[h];
      if (hf.done) continue;
      if (hf.it > 0) {
        VNR_HIP_CHECK(hipEventSynchronize(hf.ev_done));
        if (hf.tail_skipped) {   // the last march went out alone (launch_iteration)
          hf.tail_skipped = false;
          if (hf.hs[0] == 0) { hf.hc[(hf.it - 1) & 255u] = 0; hf.done = true; continue; }   // as expected: no ray left, what the packing kernel would have published
          launch_tail(f, h, hf.it - 1, hf.s);   // a ray did survive: evaluate and pack after all, then look at the count like after any iteration
          f.mark(h);
          pending = true;
          continue;
        }
        if (hf.hc[(hf.it - 1) & 255u] == 0) { hf.done = true; continue; }
      }
      launch_iteration(f, h);
      f.mark(h);
      pending = true;
    }
    if (!pending) break;
  }

  // the last march that produced zero rays ends a half's frame; remember how many iterations that took
  uint32_t pass_iterations = 0;
  uint64_t n_samples = 0, n_refrays = 0;
  for (int h = 0; h < H; ++h) {
    PartState& hf = half[h];
    VNR_HIP_CHECK(hipEventSynchronize(hf.ev_done));
    uint32_t used = hf.it;
    while (used > 1 && hf.hc[(used - 2) & 255u] == 0) --used;  // trailing speculative no-op iterations
    predicted[h] = used;
    hf.used = used;
    const uint32_t* hc = hf.hs;   // every march of the frame ran before the last compact_rays_kernel
    if (pass_mode != M_SHADOW) stats_.n_rays_hit += hc[C_HIT];
    n_samples += (uint64_t)hc[C_STAT_SAMPLES] | ((uint64_t)hc[C_STAT_SAMPLES + 1] << 32);
    n_refrays += (uint64_t)hc[C_STAT_REFRAYS] | ((uint64_t)hc[C_STAT_REFRAYS + 1] << 32);
    // march(j) emits what the reference's iteration j intersects and march(j+1) composes it, so `used` marches
    // correspond to used-1 reference iterations (= inference launches with samples); halves run side by side
    pass_iterations = std::max<uint32_t>(pass_iterations, used > 0 ? used - 1 : 0);
    if (profiling_) stats_.infer_kernel_launches += used > 0 ? used - 1 : 0;
  }
  {
    uint32_t launched[kMaxParts] = {};
    for (int h = 0; h < H; ++h) launched[h] = half[h].it;
    collect_eval_profile(f, launched, nullptr);
  }
  // summed over the passes of a frame (mode 11: camera pass + shadow pass)
  stats_.n_iterations += pass_iterations;
  stats_.n_samples += n_samples;
  stats_.n_reference_slots += n_refrays * (uint64_t)p_all.n_iters;
  completed_stats_ = stats_;
}


// per-launch and union times of the evaluation kernel from the HIP events around its launches (profiling_): `launched[h]` launches of
// part h were enqueued with an event pair each
void Renderer::collect_eval_profile(StreamingFrame& f, const uint32_t* launched, const uint32_t* /*used*/)
{
  if (!profiling_) return;
  const int H = f.H;
  bool any = false;
  for (int h = 0; h < H; ++h) {
    for (uint32_t k = 0; k < launched[h]; ++k) {
      float ms = 0.0f;
      VNR_HIP_CHECK(hipEventElapsedTime(&ms, events_[f.slot][h][2 * k], events_[f.slot][h][2 * k + 1]));
      stats_.infer_kernel_ms += ms;
      if (k < iter_ms_.size()) iter_ms_[k] += ms;
      any = true;
    }
  }
  if (!any || launched[0] == 0) return;
  // the parts' launches overlap: the union of their intervals (all timed against part 0's first event) is the time the
  // evaluation kernel had the GPU or a share of it
  std::vector<std::pair<float, float>> iv;
  for (int h = 0; h < H; ++h)
    for (uint32_t k = 0; k < launched[h]; ++k) {
      float t0 = 0.0f, t1 = 0.0f;
      VNR_HIP_CHECK(hipEventElapsedTime(&t0, events_[f.slot][0][0], events_[f.slot][h][2 * k]));
      VNR_HIP_CHECK(hipEventElapsedTime(&t1, events_[f.slot][0][0], events_[f.slot][h][2 * k + 1]));
      iv.emplace_back(t0, t1);
    }
  std::sort(iv.begin(), iv.end());
  float lo = iv[0].first, hi = iv[0].second;
  double total = 0.0;
  for (size_t e = 1; e < iv.size(); ++e) {
    if (iv[e].first > hi) { total += hi - lo; lo = iv[e].first; hi = iv[e].second; }
    else hi = std::max(hi, iv[e].second);
  }
  stats_.infer_union_ms += total + (hi - lo);
}

// ------------------------------------------------------------------------------------------------ decoupled loop (decoupled.h)
bool Renderer::decoupled_applies(const RenderParams& p, int pass_mode) const
{
  if (pass_mode != M_NONE || decoupled_mode_ == 0) return false;
  if ((size_t)p.n_local * (size_t)p.n_iters >= (1ull << 28)) return false;   // 32-bit float indices into a ring slot's arena
  return decoupled_mode_ == 2 || p.n_local <= 20480u;
}

void Renderer::render_decoupled(const RenderParams& p_all, bool defer)
{
  const uint32_t row_items = p_all.tiles_per_row * 64u;
  const uint32_t R = p_all.n_local / row_items;  // local tile rows
  const int H = (int)std::min<uint32_t>((uint32_t)decoupled_parts_, std::max(R, 1u));
  const uint32_t P_total = p_all.n_local;
  const int A = decoupled_ahead_, RING = A + 1;
  last_schedule_[0] = p_all.n_iters; last_schedule_[1] = H; last_schedule_[2] = 0; last_schedule_[3] = 1;
  NeuralVolume* nv = volume_->is_network() ? static_cast<NeuralVolume*>(volume_.get()) : nullptr;
  if (nv && !nv->network().valid()) throw std::runtime_error("neural volume has no valid network");
  if (!nv && !p_all.volume) throw std::runtime_error("this volume has no resident data to sample");

  // buffers: everything x 2 frame slots (a frame's head runs beside the tail of the frame before it)
  const size_t Pt = (size_t)P_total + 256u * kMaxParts;   // every part rounded up to whole blocks
  if (d_rays_ < Pt || d_iters_ < p_all.n_iters || d_ring_ < RING) {
    finish_pending();
    VNR_HIP_CHECK(hipDeviceSynchronize());
    d_rays_ = std::max(d_rays_, Pt); d_iters_ = std::max(d_iters_, p_all.n_iters); d_ring_ = std::max(d_ring_, RING);
    const size_t words = 15 * d_rays_ + (size_t)d_ring_ * d_rays_ + (size_t)d_ring_ * kMaxParts * D_COUNT;
    for (int sl = 0; sl < 2; ++sl) {
      d_words_[sl].resize(words);
      d_words_[sl].zero(stream_);   // the counters start at zero; compose_kernel leaves them so
      d_queue_[sl].resize((size_t)d_ring_ * d_rays_ * d_iters_);
      d_arena_[sl].resize((size_t)d_ring_ * d_rays_ * d_iters_);
    }
    VNR_HIP_CHECK(hipStreamSynchronize(stream_));
  }
  if (!d_host_) {
    VNR_HIP_CHECK(hipHostMalloc(&d_host_, 2 * kMaxParts * sizeof(DHost), hipHostMallocDefault));
    std::memset(d_host_, 0, 2 * kMaxParts * sizeof(DHost));
  }
  // The walks are the frame's critical path (a chain of launches whose duration is the latency of one wave) and the evaluation kernel
  // fills every wave slot it is given: the walk and compose streams are created with the highest priority, the evaluation streams with
  // the lowest, so that a slot an evaluation block leaves goes to a waiting walk block first (VNR_AMD_DECOUPLED_PRIO=0: all equal)
  static const bool prio = [] { const char* e = std::getenv("VNR_AMD_DECOUPLED_PRIO"); return !e || std::atoi(e) != 0; }();
  int prio_least = 0, prio_greatest = 0;
  if (prio) VNR_HIP_CHECK(hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest));
  for (int h = 0; h < H; ++h)
    for (int k = 0; k < 3; ++k)
      if (!d_streams_[h][k]) VNR_HIP_CHECK(hipStreamCreateWithPriority(&d_streams_[h][k], hipStreamNonBlocking, k == 1 ? prio_least : prio_greatest));

  // the slot the previous frame does not occupy if that one is still pending (asynchronous frames), else the same slot again
  StreamingFrame* older = frame_[slot_] && frame_[slot_]->pending ? frame_[slot_].get() : nullptr;
  const int slot = older ? slot_ ^ 1 : slot_;
  if (!frame_[slot]) frame_[slot].reset(new StreamingFrame());
  StreamingFrame& f = *frame_[slot];
  if (f.pending) throw std::runtime_error("internal: both frame slots are pending");
  f.slot = slot;
  slot_ = slot;
  f.decoupled = true; f.ahead = A; f.ring = RING;
  f.H = H; f.pass_mode = M_NONE; f.grad = false; f.ssh = false; f.nv = nv; f.p_all = p_all;
  f.shmem = ((size_t)2 * p_all.n_iters + 1) * 256 * sizeof(float) + 16 * sizeof(uint32_t) + (size_t)p_all.n_iters * 256 * sizeof(uint16_t);
  f.shmem_compose = p_all.tfn_in_lds ? (size_t)p_all.tfn.n_colors * sizeof(vec4f) + (size_t)p_all.tfn.n_alphas * sizeof(float) : 0;
  if (f.shmem > 160 * 1024) throw std::runtime_error("VNR_RM_N_ITERS too large for the LDS of one workgroup");
  f.max_iterations = 240;
#if defined(VNR_DIAG)
  if (const char* e = std::getenv("VNR_AMD_DEBUG_MAX_ITERS")) f.max_iterations = std::max(1, std::min(240, std::atoi(e)));  // truncated (wrong) frames, for timing
#endif
  static bool lds_attr_set = false;
  if (!lds_attr_set) {
    VNR_HIP_CHECK(hipFuncSetAttribute((const void*)walk_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    VNR_HIP_CHECK(hipFuncSetAttribute((const void*)walk_kernel<false>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    lds_attr_set = true;
  }
  if (profiling_) {
    iter_ms_.assign(f.max_iterations, 0.0f);
    for (int h = 0; h < H; ++h)
      while (events_[slot][h].size() < 2 * f.max_iterations) { hipEvent_t e; VNR_HIP_CHECK(hipEventCreate(&e)); events_[slot][h].push_back(e); }
  }

  uint32_t* w = d_words_[slot].ptr;
  uint32_t* const rec_base = w + 15 * d_rays_;
  uint32_t* const ctr_base = rec_base + (size_t)d_ring_ * d_rays_;
  size_t off = 0;
  for (int h = 0; h < H; ++h) {
    DPart& d = f.dpart[h];
    d.p = p_all;
    if (H > 1) {
      d.p.il_parts = p_all.il_parts * (uint32_t)H;
      d.p.il_part = p_all.il_part + p_all.il_parts * (uint32_t)h;
      d.p.n_local = ((R + (uint32_t)(H - 1 - h)) / (uint32_t)H) * row_items;
    }
    d.p.slot_cap = (uint32_t)((size_t)d.p.n_local * d.p.n_iters);   // walk_kernel: capacity of a ring slot's sample queue
    const size_t Pr = ((size_t)d.p.n_local + 255u) & ~(size_t)255u;
    uint32_t* b = w + off;
    const size_t N = d_rays_;
    d.rays.pixel = b; d.rays.jitter = (float*)(b + N); d.rays.ncb = (float*)(b + 2 * N); d.rays.walking = b + 3 * N;
    d.rays.alpha = (float*)(b + 4 * N); d.rays.done = b + 5 * N;
    // the 3-word members: plane bases are carved per part (off is a ray index; 3 words per ray from the plane's start)
    d.rays.cell = (vec3i*)(w + 6 * N) + off; d.rays.t_next = (vec3f*)(w + 9 * N) + off; d.rays.color = (vec3f*)(w + 12 * N) + off;
    for (int q = 0; q < RING; ++q) {
      d.ring[q].queue = d_queue_[slot].ptr + ((size_t)q * d_rays_ + off) * d_iters_;
      d.ring[q].arena = d_arena_[slot].ptr + ((size_t)q * d_rays_ + off) * d_iters_;
      d.ring[q].rec = rec_base + (size_t)q * d_rays_ + off;
      d.ring[q].ctr = ctr_base + ((size_t)q * kMaxParts + h) * D_COUNT;
    }
    d.host = (DHost*)d_host_ + (size_t)slot * kMaxParts + h;
    d.sw = d_streams_[h][0]; d.se = d_streams_[h][1]; d.sc = d_streams_[h][2];
    d.it_w = d.it_c = 0;
    d.target = std::max<uint32_t>(decoupled_predicted_[h], 1u);
    d.s_max = (size_t)d.p.n_local * d.p.n_iters;
    d.done = false;
    f.part[h].s = d.sc;   // the stream the part's completion event is recorded on (StreamingFrame::mark)
    off += Pr;
  }
  // fork: the part streams start after everything queued on the render stream so far
  VNR_HIP_CHECK(hipEventRecord(ev_fork_, stream_));
  for (int h = 0; h < H; ++h)
    for (int k = 0; k < 3; ++k) VNR_HIP_CHECK(hipStreamWaitEvent(d_streams_[h][k], ev_fork_, 0));

  // The HEAD of a frame is everything that needs no compose: the walks and evaluations of its first `ahead` batches.  A compose
  // writes pixels the frame before it may still be accumulating into, so with a frame pending only the head is enqueued before
  // that frame has been completed (the accumulation stays ordered: composes of one part run on one stream, frame after frame).
  if (older) {
    for (bool any = true; any;) {
      any = false;
      for (int h = 0; h < H; ++h) { const uint32_t before = f.dpart[h].it_w; decoupled_step(f, h, true); any = any || f.dpart[h].it_w != before; }
    }
    finish_streaming(*older);
    stats_ = FrameStats();
  }
  frame_of_buffer_[fb_cur_] = slot;
  for (bool any = true; any;) {
    any = false;
    for (int h = 0; h < H; ++h) {
      DPart& d = f.dpart[h];
      const uint32_t bw = d.it_w, bc = d.it_c;
      decoupled_step(f, h, false);
      any = any || d.it_w != bw || d.it_c != bc;
    }
  }
  for (int h = 0; h < H; ++h) f.mark(h);
  f.pending = true;
  if (!defer) finish_streaming(f);
}

// the next launch of part h in its canonical order W0 E0 .. W(A-1) E(A-1) | C0 W(A) E(A) C1 W(A+1) E(A+1) ... up to `target` iterations
void Renderer::decoupled_step(StreamingFrame& f, int h, bool head_only)
{
  DPart& d = f.dpart[h];
  const uint32_t A = (uint32_t)f.ahead, RING = (uint32_t)f.ring;
  if (d.it_w < d.target && d.it_w < d.it_c + A) {
    const uint32_t it = d.it_w;
    const DRing& ring = d.ring[it % RING];
    // W(it) after C(it - A): the look-ahead bound, and the ring slot W(it) writes was read by E / C(it - A - 1)
    if (it >= A) VNR_HIP_CHECK(hipStreamWaitEvent(d.sw, d.ev(d.ev_c, it - A), 0));
    const uint32_t P = d.p.n_local;
    const uint32_t blocks = std::min<uint32_t>(div_round_up(P, 256), 2048u);
    if (it == 0) walk_kernel<true><<<blocks, 256, f.shmem, d.sw>>>(d.p, d.rays, ring, d.host, it);
    else walk_kernel<false><<<blocks, 256, f.shmem, d.sw>>>(d.p, d.rays, ring, d.host, it);
    VNR_HIP_CHECK(hipGetLastError());
    VNR_HIP_CHECK(hipEventRecord(d.ev(d.ev_w, it), d.sw));
    VNR_HIP_CHECK(hipStreamWaitEvent(d.se, d.ev(d.ev_w, it), 0));
    if (profiling_) VNR_HIP_CHECK(hipEventRecord(events_[f.slot][h][2 * it], d.se));
#if defined(VNR_DIAG)
    static const bool skip_eval = [] { const char* e = std::getenv("VNR_AMD_DEBUG_SKIP_EVAL"); return e && std::atoi(e) != 0; }();   // timing of the walk / compose kernels alone (frames are garbage)
#else
    constexpr bool skip_eval = false;
#endif
    if (skip_eval) {
    } else if (f.nv) {
      f.nv->network().inference_queue((const float*)ring.queue, (float*)ring.arena, 1, ring.ctr + D_SAMPLES, d.s_max, d.se, (uint32_t)f.H, nullptr);
    } else {
      const uint32_t eb = std::min<uint32_t>(div_round_up(d.s_max, 256), (uint32_t)Runtime::get().n_cus * 8u);
      gt_sample_kernel<<<eb, 256, 0, d.se>>>(ring.ctr + D_SAMPLES, d.p.volume, d.p.vol_dims, ring.queue, (float*)ring.arena);
      VNR_HIP_CHECK(hipGetLastError());
    }
    if (profiling_) VNR_HIP_CHECK(hipEventRecord(events_[f.slot][h][2 * it + 1], d.se));
    VNR_HIP_CHECK(hipEventRecord(d.ev(d.ev_e, it), d.se));
    ++d.it_w;
    return;
  }
  if (!head_only && d.it_c < d.it_w) {
    const uint32_t it = d.it_c;
    const DRing& ring = d.ring[it % RING];
    VNR_HIP_CHECK(hipStreamWaitEvent(d.sc, d.ev(d.ev_e, it), 0));
    const uint32_t blocks = std::min<uint32_t>(div_round_up(d.p.n_local, 256), 2048u);
    compose_kernel<<<blocks, 256, f.shmem_compose, d.sc>>>(d.p, d.rays, ring, d.host, it);
    VNR_HIP_CHECK(hipGetLastError());
    VNR_HIP_CHECK(hipEventRecord(d.ev(d.ev_c, it), d.sc));
    ++d.it_c;
  }
}

void Renderer::finish_decoupled(StreamingFrame& f)
{
  if (!f.pending) return;
  f.pending = false;
  const int H = f.H;
  // everything enqueued so far runs to its end; a part whose last walk left rays walking gets one more iteration at a time
  for (;;) {
    bool more = false;
    for (int h = 0; h < H; ++h) {
      DPart& d = f.dpart[h];
      if (d.done) continue;
      for (;;) {   // (a frame whose head only was enqueued: the rest of its predicted iterations)
        const uint32_t bw = d.it_w, bc = d.it_c;
        decoupled_step(f, h, false);
        if (d.it_w == bw && d.it_c == bc) break;
      }
    }
    for (int h = 0; h < H; ++h) {
      DPart& d = f.dpart[h];
      if (d.done) continue;
      if (d.it_c == 0 || d.it_w == 0) throw std::runtime_error("internal: a decoupled frame with nothing enqueued");
      VNR_HIP_CHECK(hipEventSynchronize(d.ev(d.ev_c, d.it_c - 1)));
      if (d.host->walking[(d.it_w - 1) & 255u] == 0 || d.it_w >= f.max_iterations) { d.done = true; continue; }
      ++d.target;
      more = true;
    }
    if (!more) break;
  }
  uint32_t pass_iterations = 0, launched[kMaxParts] = {};
  uint64_t n_samples = 0, n_ref = 0;
  for (int h = 0; h < H; ++h) {
    DPart& d = f.dpart[h];
    f.mark(h);
    // iterations that had walking rays to begin with = index of the first walk that left none, + 1
    uint32_t used = 0;
    while (used < d.it_w && used < 256u) { ++used; if (d.host->walking[(used - 1) & 255u] == 0) break; }
    decoupled_predicted_[h] = used;
    uint32_t with_samples = 0;
    for (uint32_t it = 0; it < used; ++it) {
      n_samples += d.host->n_smp[it & 255u];
      n_ref += d.host->n_ref[it & 255u];
      if (d.host->n_ref[it & 255u]) with_samples = it + 1;
    }
    pass_iterations = std::max(pass_iterations, with_samples);
    stats_.n_rays_hit += d.host->hit;
    launched[h] = d.it_w;
    if (profiling_) stats_.infer_kernel_launches += with_samples;
  }
  collect_eval_profile(f, launched, nullptr);
  stats_.n_iterations += pass_iterations;
  stats_.n_samples += n_samples;
  stats_.n_reference_slots += n_ref * (uint64_t)f.p_all.n_iters;
  completed_stats_ = stats_;
}

const float* Renderer::map_frame()
{
  finish_pending();
  if (distributed_) {
    issue_gather(fb_cur_);
    VNR_HIP_CHECK(hipEventSynchronize(ev_gathered_[fb_cur_]));
    const float* out = skip_download_ ? (const float*)full_[fb_cur_].ptr : (const float*)host_fb_[fb_cur_];
    fb_cur_ ^= 1;
    return out;
  }
  VNR_HIP_CHECK(hipStreamSynchronize(stream_));
  const float* out = skip_download_ ? (const float*)fb_[fb_cur_].ptr : (const float*)host_fb_[fb_cur_];
  fb_cur_ ^= 1;  // framebuffer.safe_swap
  return out;
}

}  // namespace vnr
