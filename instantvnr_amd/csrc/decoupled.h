// decoupled.h — the sample-streaming loop (rendering modes 5 / 6, method_raymarching.cu:931-958) with the WALK taken off the chain
// evaluate -> compose -> walk.  Included by render.hip behind the kernels whose device functions it uses.
//
// Where a ray takes its samples depends on its geometry, the macrocells and the adaptive step: NOT on what the network returns.
// The only thing a value can do to a ray is end it early (alpha >= 0.9999, method_raymarching.cu:797).  The reference (and
// march_kernel above) nevertheless runs intersect -> inference -> compose strictly in turn, because compose decides which rays
// are re-appended.  On a small share of a frame (one rank of 8: 131 072 rays, one wave per SIMD) that chain IS the frame time: five
// iterations of (march 70-100 us + evaluation 45-70 us), each a launch whose duration is the latency of one wave
// (profiles/r02_share_1of8_kernel_timeline.txt), while the evaluation of all the share's samples would take 0.39 ms at the
// kernel's full rate.  Here an iteration is three kernels on three streams:
//   W(i) walk      ray generation (i = 0) / resume, macrocell DDA, adaptive steps: emits batch i of every ray that is still
//                  inside the volume and not KNOWN to be saturated; depends on W(i - 1) only, plus a bounded look-ahead:
//                  W(i) waits for C(i - A), so a ray that saturates costs at most A batches of evaluations nobody reads
//   E(i) evaluate  the network (or the ground-truth sampler) on batch i's compacted, depth-sorted sample queue; after W(i)
//   C(i) compose   classification, opacity correction, front-to-back blending of batch i in ray order; ends rays (saturated, or
//                  W marked the batch as the ray's last) and writes their pixels; after E(i) and C(i - 1)
// so the walks of the next A - 1 batches run beside the evaluation of this one, and the critical path of a frame is the walks
// plus ONE evaluation and ONE compose instead of all of them in sequence.
// Per-ray arithmetic is march_kernel's statement for statement (same DDA state, same resume rounding, same blend order and
// early exit), so frames are bit-identical to the coupled path; what differs is that up to A batches behind a saturation point
// are evaluated and dropped (the coupled path drops the rest of ONE batch), and that rays never move: lane l of group g is ray
// 64 g + l for the whole frame (no ray lists, no packing: in this regime an idle lane costs nothing, a launch does).
#pragma once

namespace vnr {

struct DRays {   // per ray of a part, in place for the whole frame
  uint32_t* pixel; float* jitter; vec3i* cell; vec3f* t_next; float* ncb; uint32_t* walking;   // written by walk_kernel
  float* alpha; vec3f* color; uint32_t* done;                                                   // written by compose_kernel (reset by W(0))
};

// device counters of one iteration in flight
enum { D_SAMPLES = 0, D_WALKING, D_HIT, D_TICKET_W, D_TICKET_C, D_NREF, D_NSMP, D_COUNT = 8 };
constexpr uint32_t kRecValid = 1u << 16, kRecLast = 1u << 17;

struct DRing {   // one iteration in flight
  vec4f* queue;      // gather-order sample records {x, y, z, result index}
  vec2f* arena;      // {value, t1 - t0} per result slot (group-interleaved, as in march_kernel)
  uint32_t* rec;     // per ray: samples in this batch | kRecValid | kRecLast
  uint32_t* ctr;     // D_*
};

// pinned, per part: what the host reads after an iteration
struct DHost {
  uint32_t walking[256], emitted[256], n_ref[256], n_smp[256];
  uint32_t hit;
};

// ------------------------------------------------------------------------------------------------ W
template <bool FIRST>
__global__ void __launch_bounds__(256) walk_kernel(const RenderParams p, const DRays r, const DRing ring, DHost* __restrict__ host, uint32_t it)
{
  extern __shared__ float s_t[];  // [n_iters][256] x {t0, t1}, histogram[256], claims[16], [n_iters][256] ranks (u16)
  float* s_t0 = s_t;
  float* s_t1 = s_t + (size_t)p.n_iters * 256;
  uint32_t* s_hist = (uint32_t*)(s_t + (size_t)2 * p.n_iters * 256);
  uint32_t* s_claim = s_hist + 256;             // [2][8]
  uint16_t* s_rk = (uint16_t*)(s_claim + 16);
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t P = p.n_local;
  const uint32_t n_round = (P + 255u) & ~255u;
  uint32_t trip = 0;
  uint32_t n_walking = 0, n_hit = 0;
  vec4f* __restrict__ queue = ring.queue;
  vec2f* __restrict__ vd_out = ring.arena;

  for (uint32_t base = blockIdx.x * 256u; base < n_round; base += gridDim.x * 256u) {
    const uint32_t i = base + tid;
    const bool active = i < P;
    uint32_t pixel = 0;
    float jitter = 0.0f;
    DDAState it_;
    it_.t_next = {0, 0, 0}; it_.cell = {0, 0, 0}; it_.next_cell_begin = 0.0f;
    vec3f org = {0, 0, 0}, dir = {0, 0, 1}, m_dir = {0, 0, 1};
    float tmin = 0.0f, tmax = VNR_FLOAT_LARGE;
    bool walk = false;
    if (active) {
      if (FIRST) {  // iterative_raygen_kernel_camera (method_raymarching.cu:840-875)
        if (map_pixel(p, i, pixel)) {
          jitter = tea_lcg_first((uint32_t)p.frame_index, pixel);
          compute_ray(p, pixel, org, dir);
          m_dir = dir * p.mc_rcp;
          walk = intersect_box(tmin, tmax, org, dir, p.bbox_lo, p.bbox_hi);
          if (walk) {
            dda_init(it_, org * p.mc_rcp, m_dir, tmin, p.mc_dims);
            r.pixel[i] = pixel; r.jitter[i] = jitter;
            r.alpha[i] = 0.0f; r.color[i] = {0, 0, 0}; r.done[i] = 0u;
            ++n_hit;
          } else {
            write_pixel(p, {0, 0, 0, 0}, pixel);
          }
        }
      } else if (r.walking[i]) {
        // a ray that a compose kernel has ended meanwhile stops walking.  The flag may be a launch or two old (the look-ahead): such a
        // ray emits samples that compose_kernel will not read
        if (r.done[i]) {
          r.walking[i] = 0u;
        } else {
          pixel = r.pixel[i];
          jitter = r.jitter[i];
          it_.cell = r.cell[i];
          it_.t_next = r.t_next[i];
          it_.next_cell_begin = r.ncb[i];
          compute_ray(p, pixel, org, dir);
          m_dir = dir * p.mc_rcp;
          intersect_box(tmin, tmax, org, dir, p.bbox_lo, p.bbox_hi);
          walk = true;
        }
      }
    }
    // RayMarchingIter::exec: the next batch of this ray into LDS
    uint32_t k = 0;
    bool last = false;
#if defined(VNR_MARCH_STAMPS)
    unsigned long long w0, w1, w2, w3, w4;
    VNR_REALTIME(rt0);
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w0) :: "memory");
#endif
    if (walk) {
      const int n_iters = p.n_iters;
      iter_exec(p, it_, m_dir, tmin, tmax, p.step, [&](float t0, float t1) -> bool {
        s_t0[k * 256u + tid] = t0;
        s_t1[k * 256u + tid] = t1;
        return (int)(++k) < n_iters;
      });
      // what the coupled path finds out one launch later, after composing this batch: nothing left to sample (k == 0: the ray ends
      // with what it has), or the walk cannot be resumed (this batch is the ray's last)
      last = k == 0 || !dda_resumable(it_, m_dir, tmin, tmax, p.mc_dims);
      r.walking[i] = last ? 0u : 1u;
      if (!last) { r.cell[i] = it_.cell; r.t_next[i] = it_.t_next; r.ncb[i] = it_.next_cell_begin; ++n_walking; }
    } else if (FIRST && active) {
      r.walking[i] = 0u;
    }
    if (active) ring.rec[i] = walk ? (k | kRecValid | (last ? kRecLast : 0u)) : 0u;
#if defined(VNR_MARCH_STAMPS)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w1) :: "memory");
#endif

    // samples by wave prefix sum, one claim per block (march_kernel)
    uint32_t incl = k;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(incl, d);
      if ((int)lane >= d) incl += y;
    }
    const uint32_t wave_samples = __shfl(incl, 63);
    uint32_t* claim = s_claim + 8u * (trip & 1u);
    if (lane == 0) claim[tid >> 6] = wave_samples;
    __syncthreads();
    if (tid == 0) {
      const uint32_t total = claim[0] + claim[1] + claim[2] + claim[3];
      claim[4] = total ? atomicAdd(ring.ctr + D_SAMPLES, total) : 0u;
    }
    __syncthreads();
    uint32_t smp_base = claim[4];
    for (uint32_t w = 0; w < (tid >> 6); ++w) smp_base += claim[w];
    ++trip;
#if defined(VNR_MARCH_STAMPS)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w2) :: "memory");
    if (!FIRST && lane == 0) { atomicAdd(&g_march_stamps[0], w1 - w0); atomicAdd(&g_march_stamps[1], w2 - w1); atomicAdd(&g_march_stamps[7], 1ull); }
    if (it == 1u && lane == 0 && (i >> 6) < kWaveRecs) {
      VNR_REALTIME(rt1);
      unsigned long long* rec = g_wave_rec[i >> 6];
      rec[0] = rt0; rec[1] = rt1; rec[2] = w1 - w0; rec[3] = w2 - w1; rec[4] = 0; rec[5] = 0; rec[6] = wave_samples; rec[7] = 0;
    }
#endif
    if (wave_samples == 0) continue;  // wave-uniform

    const bool emit = k > 0;
    float front = emit ? s_t0[tid] : VNR_FLOAT_LARGE;
#pragma unroll
    for (int d = 32; d > 0; d >>= 1) front = fminf(front, __shfl_xor(front, d));
    uint32_t* hist = s_hist + (tid & ~63u);
    hist[lane] = 0;
    __builtin_amdgcn_wave_barrier();
    if (emit) {
      for (uint32_t j = 0; j < k; ++j) {
        const float t0 = s_t0[j * 256u + tid], t1 = s_t1[j * 256u + tid];
        const float t = (1.0f - jitter) * t0 + jitter * t1;
        s_rk[j * 256u + tid] = (uint16_t)atomicAdd(&hist[depth_bin(p, t, front)], 1u);
      }
    }
    __builtin_amdgcn_wave_barrier();
    const uint32_t h = hist[lane];
    uint32_t hs = h;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
      const uint32_t y = __shfl_up(hs, d);
      if ((int)lane >= d) hs += y;
    }
    hist[lane] = smp_base + hs - h;
    __builtin_amdgcn_wave_barrier();
#if defined(VNR_MARCH_STAMPS)
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(w3) :: "memory");
#endif
    if (emit) {
      const uint32_t sb = (i >> 6) * (uint32_t)p.n_iters * 64u + lane;   // sample j of this ray: slot sb + 64 j
      for (uint32_t j = 0; j < k; ++j) {
        const float t0 = s_t0[j * 256u + tid], t1 = s_t1[j * 256u + tid];
        const float t = (1.0f - jitter) * t0 + jitter * t1;  // lerp(jitter, t0, t1), instantvnr_types.h:162-166
        const vec3f c = org + t * dir;
        const uint32_t g = hist[depth_bin(p, t, front)] + s_rk[j * 256u + tid];
        // (g < slot_cap always: the claims of one launch sum to at most n_local x n_iters; the test keeps a counter that was not
        // cleared, e.g. after a launch that failed half way, from turning into a wild store)
        if (g < p.slot_cap) queue[g] = {c.x, c.y, c.z, __uint_as_float(arena_value_index(sb + 64u * j))};
        vd_out[sb + 64u * j].y = t1 - t0;
      }
    }
    __builtin_amdgcn_wave_barrier();
#if defined(VNR_MARCH_STAMPS)
    asm volatile("s_memtime %0\n\ts_waitcnt vmcnt(0) lgkmcnt(0)" : "=s"(w4) :: "memory");
    if (!FIRST && lane == 0) { atomicAdd(&g_march_stamps[2], w3 - w2); atomicAdd(&g_march_stamps[3], w4 - w3); atomicAdd(&g_march_stamps[4], w4 - w0); }
    if (it == 1u && lane == 0 && (i >> 6) < kWaveRecs) {
      VNR_REALTIME(rt1);
      unsigned long long* rec = g_wave_rec[i >> 6];
      rec[1] = rt1; rec[4] = w3 - w2; rec[5] = w4 - w3;
    }
#endif
  }

  // rays still walking (and, at i = 0, rays that hit the volume): one atomic per block, the last block to arrive publishes
  __syncthreads();
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { n_walking += __shfl_xor(n_walking, d); n_hit += __shfl_xor(n_hit, d); }
  if (lane == 0) { s_hist[tid >> 6] = n_walking; s_hist[4 + (tid >> 6)] = n_hit; }
  __syncthreads();
  if (tid == 0) {
    const uint32_t bw = s_hist[0] + s_hist[1] + s_hist[2] + s_hist[3], bh = s_hist[4] + s_hist[5] + s_hist[6] + s_hist[7];
    if (bw) atomicAdd(ring.ctr + D_WALKING, bw);
    if (FIRST && bh) atomicAdd(ring.ctr + D_HIT, bh);
    __threadfence();
    if (atomicAdd(ring.ctr + D_TICKET_W, 1u) == gridDim.x - 1u) {
      host->walking[it & 255u] = atomicAdd(ring.ctr + D_WALKING, 0u);
      host->emitted[it & 255u] = atomicAdd(ring.ctr + D_SAMPLES, 0u);
      if (FIRST) host->hit = atomicAdd(ring.ctr + D_HIT, 0u);
    }
  }
}

// ------------------------------------------------------------------------------------------------ C
// iterative_compose_kernel (method_raymarching.cu:732-838), NO_SHADING: the compose half of march_kernel on the batch W(it) emitted
__global__ void __launch_bounds__(256) compose_kernel(const RenderParams p, const DRays r, const DRing ring, DHost* __restrict__ host, uint32_t it)
{
  extern __shared__ float s_tfn[];
  __shared__ uint32_t s_red[8];
  DeviceTfn tfn = p.tfn;
  tfn_lds_colors_t lds_colors = nullptr;
  tfn_lds_alphas_t lds_alphas = nullptr;
  bool tfn_merged = false;
  if (p.tfn_in_lds) {
    vec4f* s_colors = (vec4f*)s_tfn;
    float* s_alphas = (float*)(s_colors + p.tfn.n_colors);
    tfn_merged = tfn_tables_to_lds(p.tfn, s_colors, s_alphas, !(dbg(p) & 32u));
    __syncthreads();
    lds_colors = (tfn_lds_colors_t)s_colors;
    lds_alphas = (tfn_lds_alphas_t)s_alphas;
  }
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t P = p.n_local;
  const vec2f* __restrict__ vd_in = ring.arena;
  uint32_t n_ref = 0, n_smp = 0;
  for (uint32_t i = blockIdx.x * 256u + tid; i < P; i += gridDim.x * 256u) {
    const uint32_t rec = ring.rec[i];
    if (!(rec & kRecValid) || r.done[i]) continue;   // no batch, or a batch emitted ahead of a saturation: dropped
    const uint32_t sc = rec & 0xffffu;
    ++n_ref; n_smp += sc;
    const uint32_t pixel = r.pixel[i];
    float alpha = r.alpha[i];
    vec3f color = r.color[i];
    const uint32_t sb = (i >> 6) * (uint32_t)p.n_iters * 64u + lane;
    bool saturated = false;
    if (sc) {
      constexpr uint32_t kChunk = 8;
      vec2f ahead[kChunk];
#pragma unroll
      for (uint32_t j = 0; j < kChunk; ++j) ahead[j] = vd_in[sb + 64u * min(j, sc - 1u)];
      for (uint32_t k0 = 0; k0 < sc && !saturated; k0 += kChunk) {
        vec2f chunk[kChunk];
#pragma unroll
        for (uint32_t j = 0; j < kChunk; ++j) chunk[j] = ahead[j];
#pragma unroll
        for (uint32_t j = 0; j < kChunk; ++j) ahead[j] = vd_in[sb + 64u * min(k0 + kChunk + j, sc - 1u)];
        vec3f crgb[kChunk]; float ca[kChunk];
#pragma unroll
        for (uint32_t j = 0; j < kChunk; ++j) {
          if (p.tfn_in_lds) tfn_sample_lds(tfn, lds_colors, lds_alphas, chunk[j].x, crgb[j], ca[j], tfn_merged);   // uniform branch
          else tfn_sample(tfn, chunk[j].x, crgb[j], ca[j]);
          ca[j] = opacity_correction(p.step_rcp, chunk[j].y, ca[j]);
        }
#pragma unroll
        for (uint32_t j = 0; j < kChunk; ++j) {
          if (k0 + j >= sc) break;
          const float a = ca[j];
          const float tr = 1.0f - alpha;
          alpha += tr * a;
          color.x += tr * crgb[j].x * a; color.y += tr * crgb[j].y * a; color.z += tr * crgb[j].z * a;
          if (!(alpha < VNR_NEARLY_ONE)) { saturated = true; break; }
        }
      }
    }
    if (saturated || (rec & kRecLast)) {
      write_pixel(p, {color.x, color.y, color.z, alpha}, pixel);
      r.done[i] = 1u;
    } else {
      r.alpha[i] = alpha;
      r.color[i] = color;
    }
  }
  // statistics of the iteration as the coupled path counts them (batches of rays that were alive when they were emitted), then the
  // iteration's counters are cleared for the launch that reuses this ring slot
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) { n_ref += __shfl_xor(n_ref, d); n_smp += __shfl_xor(n_smp, d); }
  if (lane == 0) { s_red[tid >> 6] = n_ref; s_red[4 + (tid >> 6)] = n_smp; }
  __syncthreads();
  if (tid == 0) {
    const uint32_t br = s_red[0] + s_red[1] + s_red[2] + s_red[3], bs = s_red[4] + s_red[5] + s_red[6] + s_red[7];
    if (br) atomicAdd(ring.ctr + D_NREF, br);
    if (bs) atomicAdd(ring.ctr + D_NSMP, bs);
    __threadfence();
    if (atomicAdd(ring.ctr + D_TICKET_C, 1u) == gridDim.x - 1u) {
      host->n_ref[it & 255u] = atomicAdd(ring.ctr + D_NREF, 0u);
      host->n_smp[it & 255u] = atomicAdd(ring.ctr + D_NSMP, 0u);
      for (int c = 0; c < D_COUNT; ++c) atomicExch(ring.ctr + c, 0u);
    }
  }
}

// (An eight-lanes-per-ray form of these two kernels, walk8_kernel / compose8_kernel, was built in round 3, gave bit-identical frames and
// three times shorter wave-trips, and lost every measurement: 3-4 x the instructions.  Removed in round 4; the numbers are in
// docs/history/DESIGN_r01-r03.md 4.2b and profiles/r03_eight_lanes_per_ray.txt.)

}  // namespace vnr
