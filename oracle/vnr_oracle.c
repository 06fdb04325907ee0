/* vnr_oracle.c — CPU oracle (plain C) for the instantvnr hot path.
 *
 * TEST INFRASTRUCTURE ONLY — see vnr_oracle.h.  PARITY UNPINNED (the
 * reference has no tests / golden vectors and cannot be built here).
 *
 * Build: gcc -O2 -ffp-contract=off -fno-fast-math (see oracle/Makefile).
 * -ffp-contract=off matters: every fused multiply-add below is an explicit
 * fmaf() placed where nvcc's default -fmad=true would contract the
 * reference's expression, and nowhere else.
 *
 * All "ref:" citations are relative to /root/reference.
 */
#include "vnr_oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define FLOAT_LARGE 1e20f   /* ref: core/instantvnr_types.h:157 */
#define NEARLY_ONE 0.9999f  /* ref: core/instantvnr_types.h:160 */

/* ------------------------------------------------------------------------ */
/* fp16                                                                      */
/* ------------------------------------------------------------------------ */

uint16_t vnro_f32_to_f16(float f)
{
  /* IEEE binary32 -> binary16, round-to-nearest-even, subnormals kept
   * (what `(__half)x` does on sm>=70 and v_cvt_f16_f32 does on gfx950). */
  uint32_t x;
  memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) { /* inf / nan */
    return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x200u | ((ax >> 13) & 0x3ffu)) : 0u));
  }
  if (ax >= 0x477ff000u) { /* >= 65520 rounds to inf */
    return (uint16_t)(sign | 0x7c00u);
  }
  if (ax < 0x33000001u) { /* <= 2^-25: rounds to zero (tie at exactly 2^-25 -> even = 0) */
    return (uint16_t)sign;
  }
  int32_t e = (int32_t)(ax >> 23) - 127;
  uint32_t m = (ax & 0x7fffffu) | 0x800000u; /* 24-bit significand */
  uint32_t shift;
  uint32_t he;
  if (e < -14) { /* subnormal half */
    shift = (uint32_t)(13 + (-14 - e));
    he = 0;
  } else {
    shift = 13;
    he = (uint32_t)(e + 15);
  }
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u);
  const uint32_t half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  uint32_t h;
  if (he == 0) {
    h = q; /* may carry into exponent 1, which is the correct encoding */
  } else {
    h = ((he - 1) << 10) + q; /* q has the implicit bit at position 10 */
  }
  return (uint16_t)(sign | h);
}

float vnro_f16_to_f32(uint16_t h)
{
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16;
  const uint32_t e = (h >> 10) & 0x1fu;
  const uint32_t m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) {
      x = sign;
    } else {
      /* subnormal: value = m * 2^-24 */
      float v = (float)m * 5.9604644775390625e-8f;
      memcpy(&x, &v, 4);
      x |= sign;
    }
  } else if (e == 31) {
    x = sign | 0x7f800000u | (m << 13);
  } else {
    x = sign | ((e + 112u) << 23) | (m << 13);
  }
  float f;
  memcpy(&f, &x, 4);
  return f;
}

/* half + half, correctly rounded (fp32 add of two halves then RNE is exact
 * double rounding-free because 24 >= 2*11+2). */
static inline uint16_t h_add(uint16_t a, uint16_t b)
{
  return vnro_f32_to_f16(vnro_f16_to_f32(a) + vnro_f16_to_f32(b));
}

/* ------------------------------------------------------------------------ */
/* hash grid encoding                                                        */
/* ------------------------------------------------------------------------ */

static inline uint32_t next_multiple_u32(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

/* EXTERNAL (tcnn `GridEncodingTemplated` ctor, encodings/grid.h, v1.4-1.5 era):
 * per-level scale/resolution as used by ref core/networks/tcnn_impl_decoder.cu:41-42,
 * level size = min(next_multiple(res^3, 8), 2^log2_hashmap_size); offsets are
 * prefix sums in entries (ref uses them as hashmap_offset_table[level]*F,
 * tcnn_impl_decoder.cu:38-39). */
uint32_t vnro_grid_make_layout(const vnro_grid_config* cfg, vnro_grid_layout* out)
{
  const float log2_pls = log2f(cfg->per_level_scale); /* ref: tcnn_device_api.h:52 */
  uint32_t offset = 0;
  for (uint32_t l = 0; l < cfg->n_levels; ++l) {
    const float scale = exp2f((float)l * log2_pls) * (float)cfg->base_resolution - 1.0f; /* ref: tcnn_impl_decoder.cu:41 */
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;                                     /* ref: tcnn_impl_decoder.cu:42 */
    const uint32_t max_params = 0xffffffffu / 2u;
    const double cube = (double)res * (double)res * (double)res;
    uint32_t n = cube > (double)max_params ? max_params : (uint32_t)cube;
    n = next_multiple_u32(n, 8u);
    /* EXTERNAL tcnn GridEncodingTemplated ctor: Dense keeps the full level; Tiled caps it at base_resolution^3; Hash at 2^log2_hashmap_size */
    if (cfg->grid_type == 2u) {
      const double tile = (double)cfg->base_resolution * cfg->base_resolution * cfg->base_resolution;
      if ((double)n > tile) n = (uint32_t)tile;
    } else if (cfg->grid_type == 0u) {
      const uint32_t cap = 1u << cfg->log2_hashmap_size;
      if (n > cap) n = cap;
    }
    out->offsets[l] = offset;
    out->scale[l] = scale;
    out->resolution[l] = res;
    offset += n;
  }
  out->offsets[cfg->n_levels] = offset;
  return offset;
}

/* EXTERNAL (tcnn `grid_index` + `fast_hash`, common_device.h / grid.h); call
 * site ref: core/networks/tcnn_impl_decoder.cu:68-69. */
uint32_t vnro_grid_index_typed(uint32_t grid_type, uint32_t hashmap_size, uint32_t resolution, const uint32_t p[3])
{
  uint32_t stride = 1;
  uint32_t index = 0;
  /* the second part of the loop condition avoids integer overflows in finer levels (upstream comment) */
  for (uint32_t dim = 0; dim < 3 && stride <= hashmap_size; ++dim) {
    index += p[dim] * stride;
    stride *= resolution;
  }
  if (grid_type == 0u /* Hash */ && hashmap_size < stride) {
    index = (p[0] * 1u) ^ (p[1] * 2654435761u) ^ (p[2] * 805459861u);
  }
  return index % hashmap_size;
}

uint32_t vnro_grid_index(uint32_t hashmap_size, uint32_t resolution, const uint32_t p[3])
{
  return vnro_grid_index_typed(0u, hashmap_size, resolution, p);
}

/* ref: core/networks/tcnn_impl_decoder.cu:7-175 (encode_one_level), Linear /
 * Smoothstep interpolation, Hash grid, quantize_threshold = 0, max_level = inf. */
static void encode_one_level(const vnro_grid_config* cfg, const vnro_grid_layout* lay,
                             const uint16_t* table, uint32_t level, const float in[3],
                             uint16_t* out /* [F] */)
{
  const uint32_t F = cfg->n_features;
  if ((float)level >= cfg->max_level + 1e-3f) {                           /* :17-35 */
    for (uint32_t f = 0; f < F; ++f) out[f] = 0;
    return;
  }
  const uint16_t* grid = table + (size_t)lay->offsets[level] * F;         /* :38 */
  const uint32_t hashmap_size = lay->offsets[level + 1] - lay->offsets[level]; /* :39 */
  const float scale = lay->scale[level];
  const uint32_t res = lay->resolution[level];

  float pos[3];
  uint32_t pos_grid[3];
  for (int d = 0; d < 3; ++d) {
    /* EXTERNAL tcnn pos_fract (call site :54/:63): pos = x*scale+0.5 (fma under
     * nvcc -fmad), floor, fract, then interpolation function. */
    float p = fmaf(in[d], scale, 0.5f);
    const float t = floorf(p);
    pos_grid[d] = (uint32_t)(int32_t)t;
    p -= t;
    if (cfg->interpolation == 1) p = p * p * (3.0f - 2.0f * p); /* smoothstep */
    pos[d] = p;
  }

  if (cfg->interpolation == 2) {                 /* Nearest (:73-94): the entry of the lower corner, no quantisation, no blend */
    const uint32_t index = vnro_grid_index_typed(cfg->grid_type, hashmap_size, res, pos_grid) * F;
    for (uint32_t f = 0; f < F; ++f) out[f] = grid[index + f];
    return;
  }

  uint16_t result[8] = {0, 0, 0, 0, 0, 0, 0, 0}; /* PerLevelVec result = {} (:96) */
  for (uint32_t idx = 0; idx < 8; ++idx) {       /* :99-123 */
    float weight = 1.0f;
    uint32_t pl[3];
    for (uint32_t d = 0; d < 3; ++d) {
      if ((idx & (1u << d)) == 0) {
        weight *= 1.0f - pos[d];
        pl[d] = pos_grid[d];
      } else {
        weight *= pos[d];
        pl[d] = pos_grid[d] + 1u;
      }
    }
    const uint32_t index = vnro_grid_index_typed(cfg->grid_type, hashmap_size, res, pl) * F;
    for (uint32_t f = 0; f < F; ++f) {
      float data = vnro_f16_to_f32(grid[index + f]);
      if (fabsf(data) < cfg->quantize_threshold) data = 0.0f;             /* :120 */
      /* result[f] += (T)(weight * data): fp16 accumulate (:119-121) */
      result[f] = h_add(result[f], vnro_f32_to_f16(weight * data));
    }
  }
  for (uint32_t f = 0; f < F; ++f) out[f] = result[f];
}

void vnro_grid_encode(const vnro_grid_config* cfg, const uint16_t* table, const float* coords,
                      size_t n, uint16_t* out, uint32_t padded_width)
{
  vnro_grid_layout lay;
  vnro_grid_make_layout(cfg, &lay);
  const uint32_t F = cfg->n_features;
  for (size_t i = 0; i < n; ++i) {
    uint16_t* row = out + i * padded_width;
    memset(row, 0, sizeof(uint16_t) * padded_width); /* pad = 0 (tcnn_impl_decoder.cu:331-336) */
    for (uint32_t l = 0; l < cfg->n_levels; ++l) {
      /* column = level*F + f  (row-major-in-sample layout, :226) */
      encode_one_level(cfg, &lay, table, l, coords + 3 * i, row + l * F);
    }
  }
}

/* ------------------------------------------------------------------------ */
/* fully fused MLP forward                                                   */
/* ------------------------------------------------------------------------ */

size_t vnro_mlp_n_params(uint32_t in_width, uint32_t width, uint32_t n_hidden_matmuls)
{
  return (size_t)width * in_width + (size_t)n_hidden_matmuls * width * width + (size_t)16 * width;
}

/* one dense layer y[o] = sum_k W[o][k] x[k]; W row-major [out][in]
 * (ref: tcnn_threadblock.h:104 reads W as col-major B with ld = WIDTH,
 * i.e. B[k][o] = W[o*WIDTH + k]).  acc_mode F32: fp32 accumulation in k
 * order.  acc_mode F16: emulates the reference's half accumulator fragment
 * (tcnn_threadblock.h:83, OUT_T = __half): each 16-wide k block is summed in
 * fp32 together with the running half accumulator and rounded to half. */
static float dense_dot(const uint16_t* w_row, const uint16_t* x, uint32_t in, int acc_mode)
{
  if (acc_mode == VNRO_ACC_F32) {
    float s = 0.0f;
    for (uint32_t k = 0; k < in; ++k) s += vnro_f16_to_f32(w_row[k]) * vnro_f16_to_f32(x[k]);
    return s;
  }
  uint16_t acc = 0;
  for (uint32_t k0 = 0; k0 < in; k0 += 16) {
    float s = vnro_f16_to_f32(acc);
    for (uint32_t k = k0; k < k0 + 16 && k < in; ++k)
      s += vnro_f16_to_f32(w_row[k]) * vnro_f16_to_f32(x[k]);
    acc = vnro_f32_to_f16(s);
  }
  return vnro_f16_to_f32(acc);
}

/* EXTERNAL tcnn warp_activation (common_device.h, v1.4-1.5 era; call sites tcnn_threadblock.h:125,308,437,497): the function is
 * evaluated in fp32 on the half value and the result rounded back to half; K_ACT = 10 for Squareplus / Softplus.  tcnn uses the
 * fast intrinsics __expf for Exponential / Sigmoid (a few fp32 ulp from expf: invisible after the rounding to half except at
 * halfway cases, covered by the tests' 2^-8 tolerance). */
static inline uint16_t act_on_f16(uint16_t h, int activation)
{
  const float x = vnro_f16_to_f32(h);
  switch (activation) {
  case VNRO_ACT_RELU: return (h & 0x8000u) ? 0 : h; /* relu(-0) = 0 too */
  case VNRO_ACT_EXPONENTIAL: return vnro_f32_to_f16(expf(x));
  case VNRO_ACT_SIGMOID: return vnro_f32_to_f16(1.0f / (1.0f + expf(-x)));
  case VNRO_ACT_SQUAREPLUS: { const float t = x * 10.0f; return vnro_f32_to_f16(0.5f * (t + sqrtf(t * t + 4.0f)) / 10.0f); }
  case VNRO_ACT_SOFTPLUS: return vnro_f32_to_f16(logf(expf(x * 10.0f) + 1.0f) / 10.0f);
  default: return h;
  }
}

static inline uint16_t apply_act_f16(float v, int activation)
{
  /* activation is applied on the half result fragment (tcnn_threadblock.h:125) */
  return act_on_f16(vnro_f32_to_f16(v), activation);
}

/* ref: tcnn_impl.cu:191-246 (layer order + weight offsets), tcnn_threadblock.h:59-144,
 * :221-328, :446-505; final half -> float cast tcnn_impl.cu:421-431. */
void vnro_mlp_forward(const uint16_t* weights, uint32_t in_width, uint32_t width,
                      uint32_t n_hidden_matmuls, int activation, int acc_mode,
                      const uint16_t* input, size_t n, float* out, uint16_t* act_out)
{
  uint16_t* a = (uint16_t*)malloc(sizeof(uint16_t) * width);
  uint16_t* b = (uint16_t*)malloc(sizeof(uint16_t) * width);
  const int output_activation = (activation >> 8) & 0xff;   /* see vnr_oracle.h */
  activation &= 0xff;
  const uint16_t* w_first = weights;
  const uint16_t* w_hidden = weights + (size_t)width * in_width;            /* first_layer_size, tcnn_impl.cu:209 */
  const uint16_t* w_last = w_hidden + (size_t)n_hidden_matmuls * width * width; /* :241 */
  for (size_t i = 0; i < n; ++i) {
    const uint16_t* x = input + i * in_width;
    for (uint32_t o = 0; o < width; ++o)
      a[o] = apply_act_f16(dense_dot(w_first + (size_t)o * in_width, x, in_width, acc_mode), activation);
    if (act_out) memcpy(act_out + ((size_t)0 * n + i) * width, a, sizeof(uint16_t) * width);
    for (uint32_t l = 0; l < n_hidden_matmuls; ++l) {
      const uint16_t* w = w_hidden + (size_t)l * width * width;
      for (uint32_t o = 0; o < width; ++o)
        b[o] = apply_act_f16(dense_dot(w + (size_t)o * width, a, width, acc_mode), activation);
      uint16_t* t = a; a = b; b = t;
      if (act_out) memcpy(act_out + ((size_t)(l + 1) * n + i) * width, a, sizeof(uint16_t) * width);
    }
    /* last layer: 16 padded outputs, only neuron 0 is used; the output activation on the half result (tcnn_threadblock.h:497) */
    const float y = dense_dot(w_last, a, width, acc_mode);
    out[i] = vnro_f16_to_f32(act_on_f16(vnro_f32_to_f16(y), output_activation));
  }
  free(a);
  free(b);
}

void vnro_network_inference(const vnro_grid_config* cfg, uint32_t width, uint32_t n_hidden_layers,
                            int activation, int acc_mode, const uint16_t* params,
                            const float* coords, size_t n, float* out)
{
  const uint32_t in_width = next_multiple_u32(cfg->n_levels * cfg->n_features, 16u);
  const uint32_t n_hidden_matmuls = n_hidden_layers - 1; /* EXTERNAL tcnn FullyFusedMLP */
  const size_t n_mlp = vnro_mlp_n_params(in_width, width, n_hidden_matmuls);
  const uint16_t* table = params + n_mlp; /* EXTERNAL: MLP first, then grid */
  const size_t chunk = 4096;
  uint16_t* enc = (uint16_t*)malloc(sizeof(uint16_t) * chunk * in_width);
  for (size_t i0 = 0; i0 < n; i0 += chunk) {
    const size_t m = (n - i0 < chunk) ? (n - i0) : chunk;
    vnro_grid_encode(cfg, table, coords + 3 * i0, m, enc, in_width);
    vnro_mlp_forward(params, in_width, width, n_hidden_matmuls, activation, acc_mode, enc, m, out + i0, NULL);
  }
  free(enc);
}

/* ------------------------------------------------------------------------ */
/* ground-truth volume sampling                                              */
/* ------------------------------------------------------------------------ */

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
static inline float clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }

/* EXTERNAL: CUDA tex3D, normalised coords, cudaFilterModeLinear, clamp
 * addressing (ref: core/array.h:79; clamp explicit in
 * device/device_nnvolume_array.cpp:449-450).  xB = x*N - 0.5, i = floor(xB),
 * a = frac(xB).  The hardware uses 9-bit fixed-point weights; the oracle uses
 * exact fp32 weights (tolerance ~2^-8 relative on interpolants vs. real CUDA). */
float vnro_tex3d(const float* vol, const int dims[3], float px, float py, float pz)
{
  const float xb = px * (float)dims[0] - 0.5f;
  const float yb = py * (float)dims[1] - 0.5f;
  const float zb = pz * (float)dims[2] - 0.5f;
  const float fx0 = floorf(xb), fy0 = floorf(yb), fz0 = floorf(zb);
  const float a = xb - fx0, b = yb - fy0, g = zb - fz0;
  const int x0 = clampi((int)fx0, 0, dims[0] - 1), x1 = clampi((int)fx0 + 1, 0, dims[0] - 1);
  const int y0 = clampi((int)fy0, 0, dims[1] - 1), y1 = clampi((int)fy0 + 1, 0, dims[1] - 1);
  const int z0 = clampi((int)fz0, 0, dims[2] - 1), z1 = clampi((int)fz0 + 1, 0, dims[2] - 1);
  const size_t sx = 1, sy = (size_t)dims[0], sz = (size_t)dims[0] * dims[1];
  const float v000 = vol[x0 * sx + y0 * sy + z0 * sz], v100 = vol[x1 * sx + y0 * sy + z0 * sz];
  const float v010 = vol[x0 * sx + y1 * sy + z0 * sz], v110 = vol[x1 * sx + y1 * sy + z0 * sz];
  const float v001 = vol[x0 * sx + y0 * sy + z1 * sz], v101 = vol[x1 * sx + y0 * sy + z1 * sz];
  const float v011 = vol[x0 * sx + y1 * sy + z1 * sz], v111 = vol[x1 * sx + y1 * sy + z1 * sz];
  const float c00 = v000 * (1.0f - a) + v100 * a;
  const float c10 = v010 * (1.0f - a) + v110 * a;
  const float c01 = v001 * (1.0f - a) + v101 * a;
  const float c11 = v011 * (1.0f - a) + v111 * a;
  const float c0 = c00 * (1.0f - b) + c10 * b;
  const float c1 = c01 * (1.0f - b) + c11 * b;
  return c0 * (1.0f - g) + c1 * g;
}

/* ref: core/renderer/raytracing.h:105-110 (sampleVolume) */
float vnro_sample_volume(const float* vol, const int dims[3], float px, float py, float pz)
{
  const float rx = 1.0f / (float)dims[0], ry = 1.0f / (float)dims[1], rz = 1.0f / (float)dims[2];
  px = px * (1.0f - rx) + 0.5f * rx;
  py = py * (1.0f - ry) + 0.5f * ry;
  pz = pz * (1.0f - rz) + 0.5f * rz;
  return vnro_tex3d(vol, dims, px, py, pz);
}

void vnro_sample_volume_batch(const float* vol, const int dims[3], const float* coords, size_t n,
                              int nodal, float* out)
{
  for (size_t i = 0; i < n; ++i) {
    const float* p = coords + 3 * i;
    out[i] = nodal ? vnro_sample_volume(vol, dims, p[0], p[1], p[2]) /* renderer */
                   : vnro_tex3d(vol, dims, p[0], p[1], p[2]);        /* sampler: neural_sampler.cu:150-154 */
  }
}

/* ------------------------------------------------------------------------ */
/* transfer function                                                         */
/* ------------------------------------------------------------------------ */

/* ref: core/renderer/raytracing.h:71-81 (array1dNodal) + EXTERNAL CUDA tex1D
 * linear filter, normalised coords, clamp. */
static void array1d_nodal(const float* data, int len, int stride, float v, float* out)
{
  if (len == 0) { for (int c = 0; c < stride; ++c) out[c] = 0.0f; return; }
  v = clampf(v, 0.0f, 1.0f);
  const float t = fmaf(v, (float)(len - 1), 0.5f) * (1.0f / (float)len);
  const float xb = t * (float)len - 0.5f;
  const float f0 = floorf(xb);
  const float a = xb - f0;
  const int i0 = clampi((int)f0, 0, len - 1), i1 = clampi((int)f0 + 1, 0, len - 1);
  for (int c = 0; c < stride; ++c) out[c] = data[i0 * stride + c] * (1.0f - a) + data[i1 * stride + c] * a;
}

/* ref: core/renderer/raytracing.h:147-155 */
void vnro_tfn_sample(const vnro_tfn* tfn, float value, float rgb[3], float* alpha)
{
  const float v = (clampf(value, tfn->range_lo, tfn->range_hi) - tfn->range_lo) * tfn->range_rcp_norm;
  float rgba[4];
  array1d_nodal(tfn->colors, tfn->n_colors, 4, v, rgba);
  float a;
  array1d_nodal(tfn->alphas, tfn->n_alphas, 1, v, &a);
  rgb[0] = rgba[0]; rgb[1] = rgba[1]; rgb[2] = rgba[2];
  *alpha = a;
}

/* ------------------------------------------------------------------------ */
/* macrocell                                                                 */
/* ------------------------------------------------------------------------ */

#define MC_SIZE (1 << VNRO_MACROCELL_SIZE_MIP)

/* ref: core/macrocell.cu:195-201 */
void vnro_macrocell_shape(const int vol_dims[3], int mc_dims[3], float mc_spacings[3])
{
  for (int d = 0; d < 3; ++d) {
    mc_dims[d] = (vol_dims[d] + MC_SIZE - 1) / MC_SIZE;
    mc_spacings[d] = (float)MC_SIZE / (float)vol_dims[d];
  }
}

/* ref: core/macrocell.cu:11-40.  Float atomicMin/Max (instantvnr_types.h:185-199)
 * are order independent, so a sequential min/max is exact. */
static void update_single_macrocell(int vx, int vy, int vz, const int mc_dims[3], float* mc, float value)
{
  const int cx = vx >> VNRO_MACROCELL_SIZE_MIP, cy = vy >> VNRO_MACROCELL_SIZE_MIP, cz = vz >> VNRO_MACROCELL_SIZE_MIP;
  if (cx < 0 || cx >= mc_dims[0]) return;
  if (cy < 0 || cy >= mc_dims[1]) return;
  if (cz < 0 || cz >= mc_dims[2]) return;
  const size_t idx = (size_t)cx + (size_t)cy * mc_dims[0] + (size_t)cz * mc_dims[1] * mc_dims[0];
  const float lo = value - 1.0f, hi = value + 1.0f;
  if (lo < mc[2 * idx]) mc[2 * idx] = lo;
  if (hi > mc[2 * idx + 1]) mc[2 * idx + 1] = hi;
}

static void update_voxel_and_neighbours(int x, int y, int z, const int mc_dims[3], float* mc, float value)
{
  const int sx = (x % MC_SIZE) == 0 ? -1 : ((x % MC_SIZE) == (MC_SIZE - 1) ? 1 : 0);
  const int sy = (y % MC_SIZE) == 0 ? -1 : ((y % MC_SIZE) == (MC_SIZE - 1) ? 1 : 0);
  const int sz = (z % MC_SIZE) == 0 ? -1 : ((z % MC_SIZE) == (MC_SIZE - 1) ? 1 : 0);
  update_single_macrocell(x, y, z, mc_dims, mc, value);
  update_single_macrocell(x + sx, y, z, mc_dims, mc, value);
  update_single_macrocell(x, y + sy, z, mc_dims, mc, value);
  update_single_macrocell(x + sx, y + sy, z, mc_dims, mc, value);
  update_single_macrocell(x, y, z + sz, mc_dims, mc, value);
  update_single_macrocell(x + sx, y, z + sz, mc_dims, mc, value);
  update_single_macrocell(x, y + sy, z + sz, mc_dims, mc, value);
  update_single_macrocell(x + sx, y + sy, z + sz, mc_dims, mc, value);
}

/* ref: core/macrocell.cu:42-73 */
void vnro_macrocell_update_explicit(const float* coords, const float* values, size_t n,
                                    const int vol_dims[3], const int mc_dims[3], float* value_range)
{
  for (size_t i = 0; i < n; ++i) {
    int v[3];
    for (int d = 0; d < 3; ++d) {
      /* clamp((uint32_t)floorf(c*dims), 0, dims-1): a negative float->uint32 cast is
       * UB in C++/saturates to 0 on CUDA; coords are in [0,1] on this path. */
      const float f = floorf(coords[3 * i + d] * (float)vol_dims[d]);
      uint32_t u = f <= 0.0f ? 0u : (f >= 4294967040.0f ? 0xffffffffu : (uint32_t)f);
      if (u > (uint32_t)(vol_dims[d] - 1)) u = (uint32_t)(vol_dims[d] - 1);
      v[d] = (int)u;
    }
    update_voxel_and_neighbours(v[0], v[1], v[2], mc_dims, value_range, values[i]);
  }
}

/* ref: core/macrocell.cu:75-111, 221-229: value = tex3D at voxel centres */
void vnro_macrocell_compute_implicit(const float* vol, const int vol_dims[3],
                                     const int mc_dims[3], float* value_range)
{
  for (int z = 0; z < vol_dims[2]; ++z)
    for (int y = 0; y < vol_dims[1]; ++y)
      for (int x = 0; x < vol_dims[0]; ++x) {
        const float fx = ((float)x + 0.5f) / (float)vol_dims[0];
        const float fy = ((float)y + 0.5f) / (float)vol_dims[1];
        const float fz = ((float)z + 0.5f) / (float)vol_dims[2];
        const float value = vnro_tex3d(vol, vol_dims, fx, fy, fz);
        update_voxel_and_neighbours(x, y, z, mc_dims, value_range, value);
      }
}

/* ref: core/macrocell.cu:153-193 */
void vnro_macrocell_max_opacity(const vnro_tfn* tfn, const float* value_range, size_t n_cells,
                                float* max_opacity)
{
  const int len = tfn->n_alphas;
  if (len <= 0) return; /* :245 */
  for (size_t i = 0; i < n_cells; ++i) {
    const float rx = value_range[2 * i] + 1.0f;
    const float ry = value_range[2 * i + 1] - 1.0f;
    const float lower = (clampf(rx, tfn->range_lo, tfn->range_hi) - tfn->range_lo) * tfn->range_rcp_norm;
    const float upper = (clampf(ry, tfn->range_lo, tfn->range_hi) - tfn->range_lo) * tfn->range_rcp_norm;
    /* uint32_t i_lower = floorf(..) - 1: float -> uint32; -1.0f converts to 0 on CUDA (saturating) */
    const float fl = floorf(fmaf(lower, (float)(len - 1), 0.5f)) - 1.0f;
    const float fu = floorf(fmaf(upper, (float)(len - 1), 0.5f)) + 1.0f;
    uint32_t il = fl <= 0.0f ? 0u : (uint32_t)fl;
    uint32_t iu = fu <= 0.0f ? 0u : (uint32_t)fu;
    if (il > (uint32_t)(len - 1)) il = (uint32_t)(len - 1);
    if (iu > (uint32_t)(len - 1)) iu = (uint32_t)(len - 1);
    float op = 0.0f;
    for (uint32_t j = il; j <= iu; ++j) op = fmaxf(op, tfn->alphas[j]);
    max_opacity[i] = op;
  }
}

/* ------------------------------------------------------------------------ */
/* RNG                                                                       */
/* ------------------------------------------------------------------------ */

/* EXTERNAL gdt::LCG<16> (gdt/random/random.h; ref use: instantvnr_types.h:155) */
void vnro_lcg_init(vnro_lcg* r, uint32_t val0, uint32_t val1)
{
  uint32_t v0 = val0, v1 = val1, s0 = 0;
  for (int n = 0; n < 16; ++n) {
    s0 += 0x9e3779b9u;
    v0 += ((v1 << 4) + 0xa341316cu) ^ (v1 + s0) ^ ((v1 >> 5) + 0xc8013ea4u);
    v1 += ((v0 << 4) + 0xad90777du) ^ (v0 + s0) ^ ((v0 >> 5) + 0x7e95761eu);
  }
  r->state = v0;
}

float vnro_lcg_next(vnro_lcg* r)
{
  r->state = 1664525u * r->state + 1013904223u;
  return (float)(r->state & 0x00FFFFFFu) / (float)0x01000000;
}

/* EXTERNAL pcg32 (tcnn default_rng_t; ref use: neural_sampler.cu:36-41) */
void vnro_pcg32_seed(vnro_pcg32* r, uint64_t initstate, uint64_t initseq)
{
  r->state = 0u;
  r->inc = (initseq << 1u) | 1u;
  vnro_pcg32_next_uint(r);
  r->state += initstate;
  vnro_pcg32_next_uint(r);
}

uint32_t vnro_pcg32_next_uint(vnro_pcg32* r)
{
  const uint64_t oldstate = r->state;
  r->state = oldstate * 0x5851f42d4c957f2dULL + r->inc;
  const uint32_t xorshifted = (uint32_t)(((oldstate >> 18u) ^ oldstate) >> 27u);
  const uint32_t rot = (uint32_t)(oldstate >> 59u);
  return (xorshifted >> rot) | (xorshifted << ((~rot + 1u) & 31));
}

float vnro_pcg32_next_float(vnro_pcg32* r)
{
  union { uint32_t u; float f; } x;
  x.u = (vnro_pcg32_next_uint(r) >> 9) | 0x3f800000u;
  return x.f - 1.0f;
}

void vnro_pcg32_advance(vnro_pcg32* r, int64_t delta_)
{
  uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = r->inc, acc_mult = 1u, acc_plus = 0u;
  uint64_t delta = (uint64_t)delta_;
  while (delta > 0) {
    if (delta & 1) {
      acc_mult *= cur_mult;
      acc_plus = acc_plus * cur_mult + cur_plus;
    }
    cur_plus = (cur_mult + 1) * cur_plus;
    cur_mult *= cur_mult;
    delta /= 2;
  }
  r->state = acc_mult * r->state + acc_plus;
}

/* ------------------------------------------------------------------------ */
/* vector helpers                                                            */
/* ------------------------------------------------------------------------ */

typedef struct { float x, y, z; } v3;
typedef struct { int x, y, z; } i3;

static inline v3 v3_make(float x, float y, float z) { v3 r = {x, y, z}; return r; }
static inline v3 v3_add(v3 a, v3 b) { return v3_make(a.x + b.x, a.y + b.y, a.z + b.z); }
static inline v3 v3_sub(v3 a, v3 b) { return v3_make(a.x - b.x, a.y - b.y, a.z - b.z); }
static inline v3 v3_mul(v3 a, v3 b) { return v3_make(a.x * b.x, a.y * b.y, a.z * b.z); }
static inline v3 v3_scale(float s, v3 a) { return v3_make(s * a.x, s * a.y, s * a.z); }
static inline float v3_dot(v3 a, v3 b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
static inline v3 v3_cross(v3 a, v3 b)
{
  return v3_make(a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x);
}
/* EXTERNAL gdt normalize: v * (1/sqrt(dot(v,v))) */
static inline v3 v3_normalize(v3 a) { return v3_scale(1.0f / sqrtf(v3_dot(a, a)), a); }
static inline float min3f(float a, float b, float c) { return fminf(fminf(a, b), c); }
static inline float max3f(float a, float b, float c) { return fmaxf(fmaxf(a, b), c); }

typedef struct { v3 vx, vy, vz, p; } affine;

static affine affine_from(const float m[12])
{
  affine a;
  a.vx = v3_make(m[0], m[1], m[2]);
  a.vy = v3_make(m[3], m[4], m[5]);
  a.vz = v3_make(m[6], m[7], m[8]);
  a.p = v3_make(m[9], m[10], m[11]);
  return a;
}
static inline v3 xfm_vector(const affine* a, v3 v)
{
  return v3_add(v3_add(v3_scale(v.x, a->vx), v3_scale(v.y, a->vy)), v3_scale(v.z, a->vz));
}
static inline v3 xfm_point(const affine* a, v3 v) { return v3_add(xfm_vector(a, v), a->p); }
/* EXTERNAL gdt AffineSpace::inverse(): linear inverse = adjoint / det, p' = -(L^-1 p) */
static affine affine_inverse(const affine* a)
{
  const v3 c0 = v3_cross(a->vy, a->vz), c1 = v3_cross(a->vz, a->vx), c2 = v3_cross(a->vx, a->vy);
  const float det = v3_dot(a->vx, c0);
  const float r = 1.0f / det;
  affine o;
  /* rows of the adjoint are the cross products; columns of the inverse: */
  o.vx = v3_scale(r, v3_make(c0.x, c1.x, c2.x));
  o.vy = v3_scale(r, v3_make(c0.y, c1.y, c2.y));
  o.vz = v3_scale(r, v3_make(c0.z, c1.z, c2.z));
  const v3 t = xfm_vector(&o, a->p);
  o.p = v3_make(-t.x, -t.y, -t.z);
  return o;
}

/* ------------------------------------------------------------------------ */
/* ray / box, camera                                                         */
/* ------------------------------------------------------------------------ */

/* ref: core/renderer/raytracing.h:9-36 */
static int intersect_box(float* _t0, float* _t1, v3 org, v3 dir, v3 lower, v3 upper)
{
  float t0 = *_t0, t1 = *_t1;
  const int sx = fabsf(dir.x) <= FLT_MIN, sy = fabsf(dir.y) <= FLT_MIN, sz = fabsf(dir.z) <= FLT_MIN;
  const v3 rcp = v3_make(1.0f / dir.x, 1.0f / dir.y, 1.0f / dir.z);
  const v3 t_lo = v3_make(sx ? FLOAT_LARGE : (lower.x - org.x) * rcp.x,
                          sy ? FLOAT_LARGE : (lower.y - org.y) * rcp.y,
                          sz ? FLOAT_LARGE : (lower.z - org.z) * rcp.z);
  const v3 t_hi = v3_make(sx ? -FLOAT_LARGE : (upper.x - org.x) * rcp.x,
                          sy ? -FLOAT_LARGE : (upper.y - org.y) * rcp.y,
                          sz ? -FLOAT_LARGE : (upper.z - org.z) * rcp.z);
  t0 = fmaxf(t0, max3f(fminf(t_lo.x, t_hi.x), fminf(t_lo.y, t_hi.y), fminf(t_lo.z, t_hi.z)));
  t1 = fminf(t1, min3f(fmaxf(t_lo.x, t_hi.x), fmaxf(t_lo.y, t_hi.y), fmaxf(t_lo.z, t_hi.z)));
  *_t0 = t0;
  *_t1 = t1;
  return t1 > t0;
}

typedef struct { v3 position, direction, horizontal, vertical; } camera_t;

/* ref: renderer.cpp:87-96 */
static camera_t make_camera(const vnro_scene* s)
{
  camera_t c;
  const v3 from = v3_make(s->cam_from[0], s->cam_from[1], s->cam_from[2]);
  const v3 at = v3_make(s->cam_at[0], s->cam_at[1], s->cam_at[2]);
  const v3 up = v3_make(s->cam_up[0], s->cam_up[1], s->cam_up[2]);
  const float t = 2.0f * tanf(s->fovy * 0.5f * (float)M_PI / 180.0f);
  const float aspect = (float)s->width / (float)s->height;
  c.position = from;
  c.direction = v3_normalize(v3_sub(at, from));
  c.horizontal = v3_scale(t * aspect, v3_normalize(v3_cross(c.direction, up)));
  c.vertical = v3_scale(1.0f / aspect, v3_cross(c.horizontal, c.direction));
  /* NOTE: ref divides by aspect (vec / float); gdt implements vec/float as multiplication by
   * rcp(float) — treated as equal within render tolerance. */
  return c;
}

typedef struct { v3 org, dir; } ray_t;

/* ref: core/renderer/method_raymarching.cu:658-685 (compute_ray) */
static ray_t compute_ray(const vnro_scene* s, const camera_t* cam, const affine* wto, uint32_t pixel)
{
  const uint32_t ix = pixel % (uint32_t)s->width, iy = pixel / (uint32_t)s->width;
  const float sx = ((float)ix + 0.5f) / (float)s->width;
  const float sy = ((float)iy + 0.5f) / (float)s->height;
  ray_t r;
  r.org = xfm_point(wto, cam->position);
  const v3 d = v3_add(v3_add(cam->direction, v3_scale(sx - 0.5f, cam->horizontal)), v3_scale(sy - 0.5f, cam->vertical));
  r.dir = xfm_vector(wto, v3_normalize(d));
  return r;
}

/* ------------------------------------------------------------------------ */
/* DDA (ref: core/renderer/dda.h)                                            */
/* ------------------------------------------------------------------------ */

typedef struct {
  v3 t_next;
  i3 cell;
  float next_cell_begin;
} dda_iter;

/* ref: dda.h:26-46 */
static void dda_init(dda_iter* it, v3 org, v3 dir, float t_min, float t_max, i3 grid)
{
  (void)t_max;
  const v3 oiv = v3_add(org, v3_scale(t_min, dir));
  const v3 fc = v3_make(fmaxf(0.0f, fminf((float)grid.x - 1.0f, floorf(oiv.x))),
                        fmaxf(0.0f, fminf((float)grid.y - 1.0f, floorf(oiv.y))),
                        fmaxf(0.0f, fminf((float)grid.z - 1.0f, floorf(oiv.z))));
  const v3 fe = v3_make(dir.x > 0.0f ? fc.x + 1.0f : fc.x, dir.y > 0.0f ? fc.y + 1.0f : fc.y,
                        dir.z > 0.0f ? fc.z + 1.0f : fc.z);
  const v3 ts = v3_make(fabsf(1.0f / dir.x), fabsf(1.0f / dir.y), fabsf(1.0f / dir.z));
  it->t_next = v3_make(dir.x == 0.0f ? FLOAT_LARGE : fabsf(fe.x - oiv.x) * ts.x,
                       dir.y == 0.0f ? FLOAT_LARGE : fabsf(fe.y - oiv.y) * ts.y,
                       dir.z == 0.0f ? FLOAT_LARGE : fabsf(fe.z - oiv.z) * ts.z);
  it->cell.x = (int)fc.x; it->cell.y = (int)fc.y; it->cell.z = (int)fc.z;
  it->next_cell_begin = 0.0f;
}

typedef int (*dda_cell_fn)(void* ctx, i3 cell, float t0, float t1);

/* ref: dda.h:48-122 */
static int dda_next(dda_iter* it, v3 dir, float t_min, float t_max, i3 grid, dda_cell_fn fn, void* ctx)
{
  const i3 stop = { dir.x > 0.0f ? grid.x : -1, dir.y > 0.0f ? grid.y : -1, dir.z > 0.0f ? grid.z : -1 };
  if (it->cell.x == stop.x) return 0;
  if (it->cell.y == stop.y) return 0;
  if (it->cell.z == stop.z) return 0;
  const v3 ts = v3_make(fabsf(1.0f / dir.x), fabsf(1.0f / dir.y), fabsf(1.0f / dir.z));
  const i3 delta = { dir.x > 0.0f ? 1 : -1, dir.y > 0.0f ? 1 : -1, dir.z > 0.0f ? 1 : -1 };
  const float t_closest = min3f(it->t_next.x, it->t_next.y, it->t_next.z);
  const float cell_t0 = fmaxf(t_min + it->next_cell_begin, t_min);
  const float cell_t1 = fminf(t_min + t_closest, t_max);
  if (cell_t0 >= cell_t1) return 0;
  const int go = fn(ctx, it->cell, cell_t0, cell_t1);
  if (go || fmaxf(t_min + it->next_cell_begin, t_min) >= cell_t1) {
    if (it->t_next.x == t_closest) { it->t_next.x += ts.x; it->cell.x += delta.x; if (it->cell.x == stop.x) return 0; }
    if (it->t_next.y == t_closest) { it->t_next.y += ts.y; it->cell.y += delta.y; if (it->cell.y == stop.y) return 0; }
    if (it->t_next.z == t_closest) { it->t_next.z += ts.z; it->cell.z += delta.z; if (it->cell.z == stop.z) return 0; }
    it->next_cell_begin = t_closest;
  }
  return go;
}

/* ref: dda.h:124-137 */
static int dda_resumable(const dda_iter* it, v3 dir, float t_min, float t_max, i3 grid)
{
  const i3 stop = { dir.x > 0.0f ? grid.x : -1, dir.y > 0.0f ? grid.y : -1, dir.z > 0.0f ? grid.z : -1 };
  if (it->cell.x == stop.x) return 0;
  if (it->cell.y == stop.y) return 0;
  if (it->cell.z == stop.z) return 0;
  const float t_closest = min3f(it->t_next.x, it->t_next.y, it->t_next.z);
  const float cell_t0 = fmaxf(t_min + it->next_cell_begin, t_min);
  const float cell_t1 = fminf(t_min + t_closest, t_max);
  return cell_t0 < cell_t1;
}

/* ref: dda.h:140-287 (dda3, monolithic traversal) */
static void dda3(v3 org, v3 dir, float t_min, float t_max, i3 grid, dda_cell_fn fn, void* ctx)
{
  if (t_min >= t_max) return;
  dda_iter it;
  dda_init(&it, org, dir, t_min, t_max, grid);
  const v3 ts = v3_make(fabsf(1.0f / dir.x), fabsf(1.0f / dir.y), fabsf(1.0f / dir.z));
  const i3 stop = { dir.x > 0.0f ? grid.x : -1, dir.y > 0.0f ? grid.y : -1, dir.z > 0.0f ? grid.z : -1 };
  const i3 delta = { dir.x > 0.0f ? 1 : -1, dir.y > 0.0f ? 1 : -1, dir.z > 0.0f ? 1 : -1 };
  for (;;) {
    const float t_closest = min3f(it.t_next.x, it.t_next.y, it.t_next.z);
    const float cell_t0 = fmaxf(t_min + it.next_cell_begin, t_min);
    const float cell_t1 = fminf(t_min + t_closest, t_max);
    if (cell_t0 >= cell_t1) return;
    if (!fn(ctx, it.cell, cell_t0, cell_t1)) return;
    if (it.t_next.x == t_closest) { it.t_next.x += ts.x; it.cell.x += delta.x; if (it.cell.x == stop.x) return; }
    if (it.t_next.y == t_closest) { it.t_next.y += ts.y; it.cell.y += delta.y; if (it.cell.y == stop.y) return; }
    if (it.t_next.z == t_closest) { it.t_next.z += ts.z; it.cell.z += delta.z; if (it.cell.z == stop.z) return; }
    it.next_cell_begin = t_closest;
  }
}

typedef struct { int* cells; float* ts; size_t n, cap; } trace_ctx;
static int trace_cell(void* c, i3 cell, float t0, float t1)
{
  trace_ctx* t = (trace_ctx*)c;
  if (t->n < t->cap) {
    t->cells[3 * t->n + 0] = cell.x; t->cells[3 * t->n + 1] = cell.y; t->cells[3 * t->n + 2] = cell.z;
    t->ts[2 * t->n + 0] = t0; t->ts[2 * t->n + 1] = t1;
  }
  t->n++;
  return 1;
}
size_t vnro_dda_trace(const float org[3], const float dir[3], float t_min, float t_max,
                      const int grid[3], int* cells, float* ts, size_t max_cells)
{
  trace_ctx c = { cells, ts, 0, max_cells };
  const i3 g = { grid[0], grid[1], grid[2] };
  dda3(v3_make(org[0], org[1], org[2]), v3_make(dir[0], dir[1], dir[2]), t_min, t_max, g, trace_cell, &c);
  return c.n;
}

/* ------------------------------------------------------------------------ */
/* marching helpers                                                          */
/* ------------------------------------------------------------------------ */

/* ref: core/renderer/raytracing.h:172-186 */
static inline float opacity_upper_bound(const vnro_scene* s, i3 cell)
{
  const size_t idx = (size_t)cell.x + (size_t)cell.y * (size_t)s->mc_dims[0] +
                     (size_t)cell.z * (size_t)s->mc_dims[0] * (size_t)s->mc_dims[1];
  return s->mc_max_opacity[idx];
}

/* ref: core/renderer/raytracing.h:188-194 */
static inline float adaptive_sampling_rate(float base_step, float max_opacity)
{
  const float scale = 15.0f * base_step;
  const float r = fabsf(clampf(max_opacity, 0.1f, 1.0f) - 1.0f);
  return fmaxf(base_step + scale * (r * r), base_step);
}

/* ref: core/renderer/raytracing.h:166-170 */
static inline float opacity_correction(float step_rcp, float distance, float opacity)
{
  return 1.0f - powf(1.0f - opacity, step_rcp * distance);
}

/* ref: core/renderer/raytracing.h:196-207 */
static void write_pixel(const vnro_scene* s, float* accumulation, float* frame, const float rgba[4], uint32_t pixel)
{
  float out[4];
  for (int c = 0; c < 4; ++c) {
    float v = rgba[c];
    if (s->frame_index != 1) v = accumulation[4 * pixel + c] + v;
    accumulation[4 * pixel + c] = v;
    out[c] = v;
  }
  for (int c = 0; c < 4; ++c) frame[4 * pixel + c] = out[c] / (float)s->frame_index;
}

/* ------------------------------------------------------------------------ */
/* sample-streaming ray marcher (mode 5)                                     */
/* ref: core/renderer/method_raymarching.cu:544-973                          */
/* ------------------------------------------------------------------------ */

typedef struct {
  uint32_t pixel_index;
  float jitter;
  float alpha;
  v3 color;
  dda_iter iter;
} payload_t;

typedef int (*sample_body_fn)(void* ctx, float t0, float t1);

typedef struct {
  const vnro_scene* s;
  dda_iter* iter;
  float t_min, step;
  sample_body_fn body;
  void* body_ctx;
} exec_ctx;

/* the lambda of RayMarchingIter::exec, ref: method_raymarching.cu:565-580 */
static int exec_cell(void* c, i3 cell, float t0, float t1)
{
  exec_ctx* e = (exec_ctx*)c;
  const float r = opacity_upper_bound(e->s, cell);
  if (fabsf(r) <= FLT_EPSILON) return 1; /* empty cell */
  const float ss = adaptive_sampling_rate(e->step, r);
  float tx = t0, ty = fminf(t1, t0 + ss);
  while (ty > tx) {
    e->iter->next_cell_begin = ty - e->t_min;
    if (!e->body(e->body_ctx, tx, ty)) return 0;
    tx = ty;
    ty = fminf(tx + ss, t1);
  }
  return 1;
}

/* ref: method_raymarching.cu:555-600 */
static void iter_exec(const vnro_scene* s, dda_iter* iter, v3 dir, float t_min, float t_max, float step,
                      sample_body_fn body, void* body_ctx)
{
  const v3 rcp = v3_make(1.0f / s->mc_spacings[0], 1.0f / s->mc_spacings[1], 1.0f / s->mc_spacings[2]);
  const v3 m_dir = v3_mul(dir, rcp);
  const i3 grid = { s->mc_dims[0], s->mc_dims[1], s->mc_dims[2] };
  exec_ctx e = { s, iter, t_min, step, body, body_ctx };
  while (dda_next(iter, m_dir, t_min, t_max, grid, exec_cell, &e)) {}
}

/* ------------------------------------------------------------------------ */
/* gradient shading (modes 7 / 8)                                            */
/* ------------------------------------------------------------------------ */

/* ref: core/renderer/raytracing.h:214-222 */
static v3 shade_simple_light(v3 ray_dir, v3 normal, v3 albedo)
{
  if (v3_dot(normal, normal) > 1.0e-6f) {
    const v3 n = v3_normalize(normal);
    const float c = 0.2f + 0.8f * fabsf(-v3_dot(ray_dir, n));   /* dot(-ray_dir, normalize(normal)) */
    return v3_scale(c, albedo);
  }
  return v3_make(0, 0, 0);
}

/* ref: core/renderer/raytracing.h:224-246; mat = mat_gradient_shading {ambient .6, diffuse .9, specular .4, shininess 40}
 * (instantvnr_types.h:142); light_diffuse = light_directional_rgb = 1 (:147); the light_ambient argument is unused there */
static v3 shade_scivis_light(v3 ray_dir, v3 normal, v3 albedo, v3 light_dir)
{
  const float m_ambient = 0.6f, m_diffuse = 0.9f, m_specular = 0.4f, m_shininess = 40.0f;
  v3 color = v3_make(0, 0, 0);
  if (v3_dot(normal, normal) > 1.0e-6f) {
    const v3 L = v3_normalize(light_dir);
    const v3 N = v3_normalize(normal);
    const v3 V = v3_make(-ray_dir.x, -ray_dir.y, -ray_dir.z);
    color = v3_add(color, v3_scale(m_ambient, albedo));
    const float cosNL = fmaxf(v3_dot(N, L), 0.0f);
    if (cosNL > 0.0f) {
      color = v3_add(color, v3_scale(m_diffuse * cosNL, albedo));   /* * light_diffuse (= 1) */
      const v3 H = v3_normalize(v3_add(L, V));
      const float cosNH = fmaxf(v3_dot(N, H), 0.0f);
      const float sp = m_specular * powf(cosNH, m_shininess);
      color = v3_add(color, v3_make(sp, sp, sp));
    }
  }
  const v3 shading2 = shade_simple_light(ray_dir, normal, albedo);
  /* lerp(0.5, shading2, color) */
  return v3_add(v3_scale(0.5f, shading2), v3_scale(0.5f, color));
}

void vnro_shade_scivis_light(const float ray_dir[3], const float normal[3], const float albedo[3], const float light_dir[3],
                             float out[3])
{
  const v3 r = shade_scivis_light(v3_make(ray_dir[0], ray_dir[1], ray_dir[2]), v3_make(normal[0], normal[1], normal[2]),
                                  v3_make(albedo[0], albedo[1], albedo[2]), v3_make(light_dir[0], light_dir[1], light_dir[2]));
  out[0] = r.x; out[1] = r.y; out[2] = r.z;
}

/* EXTERNAL gdt (embree lineage) xfmNormal(affine a, n) = transposed(inverse(a.l)) * n; given wto = inverse(otw):
 * rows of wto.l, i.e. (dot(wto.vx-row...)).  With columns vx,vy,vz of wto.l the transpose applied to n is
 * (dot(vx, n), dot(vy, n), dot(vz, n)). */
static v3 xfm_normal_with_inverse(const affine* wto, v3 n)
{
  return v3_make(v3_dot(wto->vx, n), v3_dot(wto->vy, n), v3_dot(wto->vz, n));
}

/* one shaded sample: value f at c, forward differences f(c + gs_x e_x) etc.; `step` is the divisor of the differences
 * (mode 8: grad_step, :783; mode 7: sampleGradient's possibly sign-flipped step, raytracing.h:112-126) */
static v3 gradient_shade(const vnro_scene* s, const affine* otw, const affine* wto, v3 ray_dir_obj, float f, float fgx, float fgy,
                         float fgz, v3 step, v3 albedo)
{
  /* No = -gradient */
  const v3 No = v3_make(-((fgx - f) / step.x), -((fgy - f) / step.y), -((fgz - f) / step.z));
  const v3 Nw = xfm_normal_with_inverse(wto, No);
  const v3 dir_w = xfm_vector(otw, ray_dir_obj);
  const v3 light = v3_make(s->light_dir[0], s->light_dir[1], s->light_dir[2]);
  const v3 shaded = shade_scivis_light(dir_w, Nw, albedo, light);
  /* sampleColor = lerp(scivis_shading_scale = 0.95, sampleColor, shadingColor), instantvnr_types.h:140 */
  const float k = 0.95f;
  return v3_add(v3_scale(1.0f - k, albedo), v3_scale(k, shaded));
}

typedef struct { ray_t ray; float jitter; float* coords; uint32_t n_rays, i; int k, n_iters; int gradient; v3 gs; } intersect_ctx;
static int intersect_body(void* c, float t0, float t1)
{
  intersect_ctx* x = (intersect_ctx*)c;
  /* lerp(r,a,b) = (1-r)*a + r*b  (instantvnr_types.h:162-166) */
  const float t = (1.0f - x->jitter) * t0 + x->jitter * t1;
  const v3 p = v3_add(x->ray.org, v3_scale(t, x->ray.dir));
  const size_t slot = (size_t)x->n_rays * x->k + x->i;
  float* dst = x->coords + 3 * slot;
  dst[0] = p.x; dst[1] = p.y; dst[2] = p.z;
  if (x->gradient) {  /* :719-726: three more coordinate blocks of n_rays * N_ITERS each */
    const size_t block = (size_t)x->n_rays * x->n_iters;
    float* gx = x->coords + 3 * (1 * block + slot);
    float* gy = x->coords + 3 * (2 * block + slot);
    float* gz = x->coords + 3 * (3 * block + slot);
    v3 stp = x->gs;
    if (x->gradient == 2) {  /* in shader (mode 9): sampleGradient flips a step that would leave [0,1] (raytracing.h:128-143) */
      if (p.x + stp.x > 1.0f - FLT_EPSILON) stp.x *= -1.0f;
      if (p.y + stp.y > 1.0f - FLT_EPSILON) stp.y *= -1.0f;
      if (p.z + stp.z > 1.0f - FLT_EPSILON) stp.z *= -1.0f;
    }
    gx[0] = p.x + stp.x; gx[1] = p.y; gx[2] = p.z;
    gy[0] = p.x; gy[1] = p.y + stp.y; gy[2] = p.z;
    gz[0] = p.x; gz[1] = p.y; gz[2] = p.z + stp.z;
  }
  return (++x->k) < x->n_iters;
}

typedef struct {
  const vnro_scene* s; const float* samples; uint32_t n_rays, i; int k, n_iters;
  float alpha; v3 color; float step_rcp;
  int gradient; v3 gs; const affine* otw; const affine* wto; v3 ray_dir;
  v3 ray_org; float jitter;   /* gradient == 2 recomputes the sample position to repeat sampleGradient's flip */
} compose_ctx;
static int compose_body(void* c, float t0, float t1)
{
  compose_ctx* x = (compose_ctx*)c;
  const size_t slot = (size_t)x->n_rays * x->k + x->i;
  const float value = x->samples[slot];
  float rgb[3], a;
  vnro_tfn_sample(&x->s->tfn, value, rgb, &a);
  a = opacity_correction(x->step_rcp, t1 - t0, a);
  if (x->gradient) {  /* :773-788 */
    const size_t block = (size_t)x->n_rays * x->n_iters;
    v3 stp = x->gs;
    if (x->gradient == 2) {
      const float t = (1.0f - x->jitter) * t0 + x->jitter * t1;
      const v3 p = v3_add(x->ray_org, v3_scale(t, x->ray_dir));
      if (p.x + stp.x > 1.0f - FLT_EPSILON) stp.x *= -1.0f;
      if (p.y + stp.y > 1.0f - FLT_EPSILON) stp.y *= -1.0f;
      if (p.z + stp.z > 1.0f - FLT_EPSILON) stp.z *= -1.0f;
    }
    const v3 shaded = gradient_shade(x->s, x->otw, x->wto, x->ray_dir, value, x->samples[1 * block + slot],
                                     x->samples[2 * block + slot], x->samples[3 * block + slot], stp,
                                     v3_make(rgb[0], rgb[1], rgb[2]));
    rgb[0] = shaded.x; rgb[1] = shaded.y; rgb[2] = shaded.z;
  }
  const float tr = 1.0f - x->alpha;
  x->alpha += tr * a;
  x->color.x += tr * rgb[0] * a;
  x->color.y += tr * rgb[1] * a;
  x->color.z += tr * rgb[2] * a;
  return ((++x->k) < x->n_iters) && (x->alpha < NEARLY_ONE);
}

/* SINGLE_SHADE_HEURISTIC state carried by a ray (SingleShotPayload, method_raymarching.cu:157-162) */
typedef struct { v3 org, color; float alpha; } ssh_t;

/* per-pixel results of the SSH camera pass that the shadow pass consumes (:88-92): highest-contribution sample, the
 * unshaded pixel, the second jitter */
typedef struct { v3* org; v3* color; float* alpha; float* shading; float* jitter; } ssh_pixels;

typedef struct {
  const vnro_scene* s; const float* samples; uint32_t n_rays, i; int k, n_iters;
  float alpha; v3 color; float step_rcp;
  ray_t ray; float jitter; ssh_t* ssh; int shadow;
} compose2_ctx;
/* iterative_compose_kernel<SINGLE_SHADE_HEURISTIC / SHADOW> body (:762-803) */
static int compose2_body(void* c, float t0, float t1)
{
  compose2_ctx* x = (compose2_ctx*)c;
  const size_t slot = (size_t)x->n_rays * x->k + x->i;
  const float value = x->samples[slot];
  float rgb[3], a;
  vnro_tfn_sample(&x->s->tfn, value, rgb, &a);
  a = opacity_correction(x->step_rcp, t1 - t0, a);
  if (x->ssh) {  /* remember the sample that contributes most (:789-795) */
    if (x->ssh->alpha < (1.0f - x->alpha) * a) {
      const float t = (1.0f - x->jitter) * t0 + x->jitter * t1;
      x->ssh->org = v3_add(x->ray.org, v3_scale(t, x->ray.dir));
      x->ssh->color = v3_make(rgb[0], rgb[1], rgb[2]);
      x->ssh->alpha = (1.0f - x->alpha) * a;
    }
  }
  const float tr = 1.0f - x->alpha;
  x->alpha += tr * a;
  if (!x->shadow) {
    x->color.x += tr * rgb[0] * a;
    x->color.y += tr * rgb[1] * a;
    x->color.z += tr * rgb[2] * a;
  }
  return ((++x->k) < x->n_iters) && (x->alpha < NEARLY_ONE);
}

/* One iterative_raymarching_loop<MODE> (:931-958).  pass: 0 = NO_SHADING / GRADIENT_SHADING camera rays, 2 = SINGLE_SHADE_HEURISTIC
 * camera rays (results go to `px`, no pixel is written), 3 = SHADOW rays from px->org towards the light (:877-900, 639-653) */
static void streaming_pass(const vnro_scene* s, int pass, int n_iters, vnro_value_fn fn, void* user, float* accumulation, float* frame,
                           vnro_render_stats* st, ssh_pixels* px)
{
  /* shading_mode 3 = the in-shader single-shade heuristic (rendering mode 12; network_raymarching_traceray / _transmittance,
   * method_raymarching.cu:981-1035, 1037-1128): the shadow ray marches at raymarching_shadow_sampling_scale = 2 x the step and is
   * jittered by the pixel's THIRD random number (two get_floats() calls), where the streaming variant uses the step and the second */
  const int in_shader_ssh = s->shading_mode == 3;
  const uint32_t n_pixels = (uint32_t)s->width * (uint32_t)s->height;
  const camera_t cam = make_camera(s);
  const affine otw = affine_from(s->xfm);
  const affine wto = affine_inverse(&otw);
  const v3 lo = v3_make(s->bbox_lo[0], s->bbox_lo[1], s->bbox_lo[2]);
  const v3 hi = v3_make(s->bbox_hi[0], s->bbox_hi[1], s->bbox_hi[2]);
  const float step = (pass == 3 && in_shader_ssh ? 2.0f : 1.0f) * (1.0f / s->sampling_rate), step_rcp = s->sampling_rate; /* object.cpp:303-304 */
  const v3 rcp = v3_make(1.0f / s->mc_spacings[0], 1.0f / s->mc_spacings[1], 1.0f / s->mc_spacings[2]);
  const i3 grid = { s->mc_dims[0], s->mc_dims[1], s->mc_dims[2] };
  const float shading_scale = 0.95f;  /* scivis_shading_scale, instantvnr_types.h:140 */

  /* GRADIENT_SHADING streams 4 coordinates per sample (:198, :934): the sample and three forward offsets of grad_step */
  /* shading_mode 4 = the in-shader gradient shading (rendering mode 9; :1068-1070): mode 8 with sampleGradient's boundary flip */
  const int gradient = pass == 0 ? (s->shading_mode == 1 ? 1 : s->shading_mode == 4 ? 2 : 0) : 0;
  const size_t per_sample = gradient ? 4 : 1;
  const v3 gs = v3_make(1.0f / (float)s->vol_dims[0], 1.0f / (float)s->vol_dims[1], 1.0f / (float)s->vol_dims[2]); /* object.cpp:305 */
  payload_t* cur = (payload_t*)malloc(sizeof(payload_t) * n_pixels);
  payload_t* nxt = (payload_t*)malloc(sizeof(payload_t) * n_pixels);
  ssh_t* ssh_cur = (ssh_t*)calloc(n_pixels, sizeof(ssh_t));
  ssh_t* ssh_nxt = (ssh_t*)calloc(n_pixels, sizeof(ssh_t));
  float* coords = (float*)calloc((size_t)n_pixels * n_iters * 3 * per_sample, sizeof(float));
  float* values = (float*)calloc((size_t)n_pixels * n_iters * per_sample, sizeof(float));
  /* shadow rays all point towards the light: xfmVector(wto, normalize(light_directional_dir)) (:649) */
  const v3 ldir = xfm_vector(&wto, v3_normalize(v3_make(s->light_dir[0], s->light_dir[1], s->light_dir[2])));

  /* raygen, ref: method_raymarching.cu:840-875 (camera) / :877-900 (shadow) */
  uint32_t n_rays = 0;
  const uint32_t p_lo = s->pixel_lo, p_hi = s->pixel_hi < n_pixels ? s->pixel_hi : n_pixels;
  for (uint32_t i = p_lo; i < p_hi; ++i) {
    ray_t ray;
    float jitter;
    int want = 1;
    if (s->il_parts > 1 && s->il_block > 0 && (i / s->il_block) % s->il_parts != s->il_part) continue;   /* another rank's pixel */
    if (pass == 3) {
      ray.org = px->org[i]; ray.dir = ldir;
      jitter = px->jitter[i];
      want = px->alpha[i] > 0.0f;
    } else {
      vnro_lcg rng;
      vnro_lcg_init(&rng, (uint32_t)s->frame_index, i);
      jitter = vnro_lcg_next(&rng); /* get_floats().x */
      if (pass == 2) px->jitter[i] = vnro_lcg_next(&rng); /* get_floats().y (:866-868; stored for every pixel here) */
      if (pass == 2 && in_shader_ssh) px->jitter[i] = vnro_lcg_next(&rng); /* the next get_floats().x */
      ray = compute_ray(s, &cam, &wto, i);
    }
    float tmin = 0.0f, tmax = FLOAT_LARGE;
    if (intersect_box(&tmin, &tmax, ray.org, ray.dir, lo, hi) && want) {
      payload_t p;
      p.pixel_index = i; p.jitter = jitter; p.alpha = 0.0f;
      p.color = pass == 3 ? ray.org : v3_make(0, 0, 0);   /* color_or_org: a shadow ray keeps its origin there (:639-653) */
      dda_init(&p.iter, v3_mul(ray.org, rcp), v3_mul(ray.dir, rcp), tmin, tmax, grid); /* :544-553 */
      ssh_cur[n_rays].org = v3_make(0, 0, 0); ssh_cur[n_rays].color = v3_make(0, 0, 0); ssh_cur[n_rays].alpha = 0.0f;
      cur[n_rays++] = p;
    } else if (pass == 3) {
      write_pixel(s, accumulation, frame, px->shading + 4 * (size_t)i, i);   /* :897-899 */
    } else if (pass == 2) {
      /* :870-874: nothing is written; the reference's per-frame memset leaves zeros for the shadow pass to find */
      px->org[i] = v3_make(0, 0, 0); px->color[i] = v3_make(0, 0, 0); px->alpha[i] = 0.0f;
      px->shading[4 * (size_t)i] = px->shading[4 * (size_t)i + 1] = px->shading[4 * (size_t)i + 2] = px->shading[4 * (size_t)i + 3] = 0.0f;
    } else {
      const float zero[4] = {0, 0, 0, 0};
      write_pixel(s, accumulation, frame, zero, i);
    }
  }
  if (pass != 3) st->n_rays_hit = n_rays;

  /* loop, ref: method_raymarching.cu:931-958 */
  while (n_rays > 0) {
    st->n_iterations++;
    st->n_slots += (uint64_t)n_rays * (uint64_t)n_iters;
    /* intersect, :687-730 (iterator state is NOT saved here) */
    for (uint32_t i = 0; i < n_rays; ++i) {
      payload_t p = cur[i];
      ray_t ray;
      if (pass == 3) { ray.org = p.color; ray.dir = ldir; } else ray = compute_ray(s, &cam, &wto, p.pixel_index);
      float tmin = 0.0f, tmax = FLOAT_LARGE;
      intersect_box(&tmin, &tmax, ray.org, ray.dir, lo, hi);
      intersect_ctx x = { ray, p.jitter, coords, n_rays, i, 0, n_iters, gradient, gs };
      iter_exec(s, &p.iter, ray.dir, tmin, tmax, step, intersect_body, &x);
      st->n_samples += (uint64_t)x.k;
    }
    /* inference of ALL n_iters*n_rays slots (stale coords included), :950-953; x4 blocks with gradient shading */
    fn(user, coords, (size_t)n_rays * n_iters * per_sample, values);
    /* compose, :732-838 */
    uint32_t n_next = 0;
    for (uint32_t i = 0; i < n_rays; ++i) {
      payload_t p = cur[i];
      ray_t ray;
      if (pass == 3) { ray.org = p.color; ray.dir = ldir; } else ray = compute_ray(s, &cam, &wto, p.pixel_index);
      float tmin = 0.0f, tmax = FLOAT_LARGE;
      intersect_box(&tmin, &tmax, ray.org, ray.dir, lo, hi);
      ssh_t ssh = ssh_cur[i];
      if (pass == 0) {
        compose_ctx x = { s, values, n_rays, i, 0, n_iters, p.alpha, p.color, step_rcp, gradient, gs, &otw, &wto, ray.dir, ray.org, p.jitter };
        iter_exec(s, &p.iter, ray.dir, tmin, tmax, step, compose_body, &x);
        p.alpha = x.alpha; p.color = x.color;
      } else {
        compose2_ctx x = { s, values, n_rays, i, 0, n_iters, p.alpha, pass == 3 ? v3_make(0, 0, 0) : p.color, step_rcp, ray, p.jitter,
                           pass == 2 ? &ssh : NULL, pass == 3 };
        iter_exec(s, &p.iter, ray.dir, tmin, tmax, step, compose2_body, &x);
        p.alpha = x.alpha;
        if (pass == 2) p.color = x.color;   /* a shadow ray's color_or_org stays its origin (set_ray<SHADOW>) */
      }
      const int resumable = dda_resumable(&p.iter, v3_mul(ray.dir, rcp), tmin, tmax, grid);
      if (p.alpha < NEARLY_ONE && resumable) {
        ssh_nxt[n_next] = ssh;
        nxt[n_next++] = p;
      } else if (pass == 3) {  /* :820-826 */
        const uint32_t pidx = p.pixel_index;
        const float transmittance = 1.0f - p.alpha;
        float sc[4] = { px->shading[4 * (size_t)pidx], px->shading[4 * (size_t)pidx + 1], px->shading[4 * (size_t)pidx + 2], px->shading[4 * (size_t)pidx + 3] };
        const v3 hc = px->color[pidx];
        sc[0] = (1.0f - shading_scale) * sc[0] + shading_scale * (hc.x * sc[3] * transmittance);
        sc[1] = (1.0f - shading_scale) * sc[1] + shading_scale * (hc.y * sc[3] * transmittance);
        sc[2] = (1.0f - shading_scale) * sc[2] + shading_scale * (hc.z * sc[3] * transmittance);
        write_pixel(s, accumulation, frame, sc, pidx);
      } else if (pass == 2) {  /* :827-833 */
        const uint32_t pidx = p.pixel_index;
        px->org[pidx] = ssh.org; px->color[pidx] = ssh.color; px->alpha[pidx] = ssh.alpha;
        px->shading[4 * (size_t)pidx] = p.color.x; px->shading[4 * (size_t)pidx + 1] = p.color.y;
        px->shading[4 * (size_t)pidx + 2] = p.color.z; px->shading[4 * (size_t)pidx + 3] = p.alpha;
      } else {
        const float rgba[4] = { p.color.x, p.color.y, p.color.z, p.alpha };
        write_pixel(s, accumulation, frame, rgba, p.pixel_index);
      }
    }
    payload_t* t = cur; cur = nxt; nxt = t;
    ssh_t* u = ssh_cur; ssh_cur = ssh_nxt; ssh_nxt = u;
    n_rays = n_next;
  }
  free(cur); free(nxt); free(ssh_cur); free(ssh_nxt); free(coords); free(values);
}

/* do_raymarching_iterative (:960-973): shading_mode 0 / 1 = one pass, 2 = SINGLE_SHADE_HEURISTIC = camera pass + shadow pass */
void vnro_render_streaming(const vnro_scene* s, int n_iters, vnro_value_fn fn, void* user,
                           float* accumulation, float* frame, vnro_render_stats* stats)
{
  vnro_render_stats st = {0, 0, 0, 0};
  if (s->shading_mode == 2 || s->shading_mode == 3) {
    const size_t n = (size_t)s->width * (size_t)s->height;
    ssh_pixels px;
    px.org = (v3*)calloc(n, sizeof(v3)); px.color = (v3*)calloc(n, sizeof(v3)); px.alpha = (float*)calloc(n, sizeof(float));
    px.shading = (float*)calloc(4 * n, sizeof(float)); px.jitter = (float*)calloc(n, sizeof(float));
    streaming_pass(s, 2, n_iters, fn, user, accumulation, frame, &st, &px);
    streaming_pass(s, 3, n_iters, fn, user, accumulation, frame, &st, &px);
    free(px.org); free(px.color); free(px.alpha); free(px.shading); free(px.jitter);
  } else {
    streaming_pass(s, 0, n_iters, fn, user, accumulation, frame, &st, NULL);
  }
  if (stats) *stats = st;
}

/* ------------------------------------------------------------------------ */
/* path tracing, sample streaming (rendering mode 14)                        */
/* ref: core/renderer/method_pathtracing.cu:532-813; VARYING_MAJORANT = 1     */
/* (ADAPTIVE_SAMPLING is not defined in that translation unit, :24-27)        */
/* ------------------------------------------------------------------------ */

typedef struct {
  /* Ray (:70-79) */
  float tnear, tfar;
  uint32_t pidx;
  v3 org, dir;
  int shadow;
  /* SampleStreamingPayload (:100-113) */
  uint32_t scatter_index;
  v3 sample_coord;
  float majorant;
  v3 L, throughput;
  dda_iter iter;
  vnro_lcg rng;
} pt_ray;

typedef struct { const vnro_scene* s; pt_ray* r; float tau, t, density_scale; int found_hit; float rayt; } pt_hit_ctx;

/* the lambda of DeltaTrackingIter::hashit (:558-570) */
static int pt_hit_cell(void* c, i3 cell, float t0, float t1)
{
  (void)t0;
  pt_hit_ctx* x = (pt_hit_ctx*)c;
  x->r->majorant = opacity_upper_bound(x->s, cell) * x->density_scale;
  if (fabsf(x->r->majorant) <= FLT_EPSILON) return 1;  /* move to the next macrocell (t is NOT advanced, as in the reference) */
  x->tau -= (t1 - x->t) * (x->r->majorant * 1.0f);
  x->t = t1;
  if (x->tau > 0.0f) return 1;
  x->t = x->t + x->tau / (x->r->majorant * 1.0f);
  x->found_hit = 1;
  x->r->iter.next_cell_begin = x->t - x->r->tnear;
  x->rayt = x->t;
  return 0;
}

/* DeltaTrackingIter::hashit (:545-573) */
static int pt_hashit(const vnro_scene* s, pt_ray* r, v3 rcp, i3 grid, float density_scale, float* rayt)
{
  pt_hit_ctx x;
  x.s = s; x.r = r; x.density_scale = density_scale; x.found_hit = 0; x.rayt = 0.0f;
  x.tau = -logf(1.0f - vnro_lcg_next(&r->rng));
  x.t = r->iter.next_cell_begin + r->tnear;
  const v3 m_dir = v3_mul(r->dir, rcp);
  while (dda_next(&r->iter, m_dir, r->tnear, r->tfar, grid, pt_hit_cell, &x)) {}
  *rayt = x.rayt;
  return x.found_hit;
}

/* uniform_sample_sphere (raytracing.h:253-270): phi = 2 * M_PI * s.x is evaluated in double and rounded to float */
static v3 pt_uniform_sample_sphere(float radius, float sx, float sy)
{
  const float phi = (float)(2 * M_PI * sx);
  const float cosTheta = radius * (1.f - 2.f * sy);
  const float sinTheta = 2.f * radius * sqrtf(sy * (1.f - sy));
  return v3_make(cosf(phi) * sinTheta, sinf(phi) * sinTheta, cosTheta);
}

typedef struct { const vnro_scene* s; const affine* wto; v3 lo, hi, rcp; i3 grid; float density_scale; v3 light_dir_obj;
                 int reset_interval;  /* in-shader / monolithic estimator: tnear = 0, tfar = large before a bounce (:438-439, 999-1001) */
} pt_env;

/* iterative_take_sample (:598-636) */
static int pt_take_sample(const pt_env* e, pt_ray* r)
{
  float t;
  if (pt_hashit(e->s, r, e->rcp, e->grid, e->density_scale, &t)) {
    r->sample_coord = v3_add(r->org, v3_scale(t, r->dir));
    return 1;
  }
  /* ray exits the volume, compute lighting */
  if (r->scatter_index > 0) {  /* no light accumulation for primary rays */
    if (r->shadow) {
      r->L = v3_add(r->L, v3_scale(1.0f, r->throughput));  /* light_directional_rgb = 1 (instantvnr_types.h:147) */
      r->shadow = 0;
      const float s0 = vnro_lcg_next(&r->rng), s1 = vnro_lcg_next(&r->rng);
      r->dir = xfm_vector(e->wto, pt_uniform_sample_sphere(1.f, s0, s1));
      if (e->reset_interval) { r->tnear = 0.f; r->tfar = FLOAT_LARGE; }
      if (!intersect_box(&r->tnear, &r->tfar, r->org, r->dir, e->lo, e->hi)) return 0;  /* streaming: the interval is NOT reset first */
      dda_init(&r->iter, v3_mul(r->org, e->rcp), v3_mul(r->dir, e->rcp), r->tnear, r->tfar, e->grid);
      if (pt_hashit(e->s, r, e->rcp, e->grid, e->density_scale, &t)) {
        r->sample_coord = v3_add(r->org, v3_scale(t, r->dir));
        return 1;
      }
      /* the bounce leaves the volume without a tentative collision: the streaming variant ends the path here WITHOUT the ambient term
       * (it falls through to `return false`, :631-635); the monolithic / in-shader estimator adds it on its next loop trip (:447-452) */
      if (e->reset_interval) r->L = v3_add(r->L, v3_scale(1.5f, r->throughput));
    } else {
      r->L = v3_add(r->L, v3_scale(1.5f, r->throughput));  /* light_ambient = 1.5 (instantvnr_types.h:146) */
    }
  }
  return 0;
}

/* iterative_shade (:638-677) */
static int pt_shade(const pt_env* e, pt_ray* r, float sample_value)
{
  float rgb[3], a;
  vnro_tfn_sample(&e->s->tfn, sample_value, rgb, &a);
  if (vnro_lcg_next(&r->rng) * r->majorant >= a * e->density_scale) return 1;  /* null collision */
  const v3 albedo = v3_make(rgb[0], rgb[1], rgb[2]);
  if (r->shadow) {
    r->shadow = 0;
    const float s0 = vnro_lcg_next(&r->rng), s1 = vnro_lcg_next(&r->rng);
    r->dir = xfm_vector(e->wto, pt_uniform_sample_sphere(1.f, s0, s1));
    if (e->reset_interval) { r->tnear = 0.f; r->tfar = FLOAT_LARGE; }
  } else {
    /* russian_roulette (:366-376), russian_roulette_length = 4 */
    if (r->scatter_index > 4) {
      const float q = fminf(0.95f, fmaxf(fmaxf(r->throughput.x, r->throughput.y), r->throughput.z));
      if (vnro_lcg_next(&r->rng) > q) return 0;
      r->throughput = v3_make(r->throughput.x / q, r->throughput.y / q, r->throughput.z / q);
    }
    ++r->scatter_index;
    r->org = r->sample_coord;
    r->tnear = 0.f;
    r->tfar = FLOAT_LARGE;
    r->throughput = v3_mul(r->throughput, v3_scale(0.6f, albedo));  /* PHASE(albedo) = albedo * 0.6f (:35) */
    r->shadow = 1;
    r->dir = e->light_dir_obj;
  }
  if (!intersect_box(&r->tnear, &r->tfar, r->org, r->dir, e->lo, e->hi)) return 0;
  dda_init(&r->iter, v3_mul(r->org, e->rcp), v3_mul(r->dir, e->rcp), r->tnear, r->tfar, e->grid);
  return 1;
}

/* do_path_tracing_iterative (:786-806) with iterative_raygen_kernel (:679-748) and iterative_shade_kernel (:750-768).  Every
 * iteration reloads a ray with tnear = 0, tfar = large and re-intersects the box (load, :115-143). */
void vnro_render_pathtracing(const vnro_scene* s, vnro_value_fn fn, void* user, float* accumulation, float* frame,
                             vnro_render_stats* stats)
{
  const uint32_t n_pixels = (uint32_t)s->width * (uint32_t)s->height;
  const camera_t cam = make_camera(s);
  const affine otw = affine_from(s->xfm);
  const affine wto = affine_inverse(&otw);
  pt_env e;
  e.s = s; e.wto = &wto;
  e.lo = v3_make(s->bbox_lo[0], s->bbox_lo[1], s->bbox_lo[2]);
  e.hi = v3_make(s->bbox_hi[0], s->bbox_hi[1], s->bbox_hi[2]);
  e.rcp = v3_make(1.0f / s->mc_spacings[0], 1.0f / s->mc_spacings[1], 1.0f / s->mc_spacings[2]);
  e.grid.x = s->mc_dims[0]; e.grid.y = s->mc_dims[1]; e.grid.z = s->mc_dims[2];
  e.density_scale = s->density_scale == 0.0f ? 1.0f : s->density_scale;
  e.light_dir_obj = xfm_vector(&wto, v3_normalize(v3_make(s->light_dir[0], s->light_dir[1], s->light_dir[2])));
  /* shading_mode 5: the in-shader path tracer's estimator (rendering mode 15; network_path_tracing_traceray, :968-1025), which is
   * path_tracing_traceray's: the streaming loop with the interval reset before a bounce */
  e.reset_interval = s->shading_mode == 5;
  vnro_render_stats st = {0, 0, 0, 0};
  pt_ray* rays = (pt_ray*)malloc(sizeof(pt_ray) * (n_pixels ? n_pixels : 1));
  float* coords = (float*)malloc(sizeof(float) * 3 * (n_pixels ? n_pixels : 1));
  float* values = (float*)malloc(sizeof(float) * (n_pixels ? n_pixels : 1));
  uint32_t n_rays = 0;
  const uint32_t p_lo = s->pixel_lo, p_hi = s->pixel_hi < n_pixels ? s->pixel_hi : n_pixels;
  for (uint32_t i = p_lo; i < p_hi; ++i) {
    pt_ray r;
    const ray_t cr = compute_ray(s, &cam, &wto, i);
    r.tnear = 0.f; r.tfar = FLOAT_LARGE; r.pidx = i; r.org = cr.org; r.dir = cr.dir; r.shadow = 0;
    r.scatter_index = 0; r.sample_coord = v3_make(0, 0, 0); r.majorant = 0.f;
    r.L = v3_make(0, 0, 0); r.throughput = v3_make(1, 1, 1);
    vnro_lcg_init(&r.rng, (uint32_t)s->frame_index, i);
    int alive = 0;
    if (intersect_box(&r.tnear, &r.tfar, r.org, r.dir, e.lo, e.hi)) {
      st.n_rays_hit++;
      dda_init(&r.iter, v3_mul(r.org, e.rcp), v3_mul(r.dir, e.rcp), r.tnear, r.tfar, e.grid);
      alive = pt_take_sample(&e, &r);
    }
    if (alive) rays[n_rays++] = r;
    else { const float rgba[4] = { r.L.x, r.L.y, r.L.z, 1.f }; write_pixel(s, accumulation, frame, rgba, i); }
  }
  while (n_rays > 0) {
    st.n_iterations++;
    st.n_samples += n_rays;
    st.n_slots += n_rays;
    for (uint32_t i = 0; i < n_rays; ++i) { coords[3 * i] = rays[i].sample_coord.x; coords[3 * i + 1] = rays[i].sample_coord.y; coords[3 * i + 2] = rays[i].sample_coord.z; }
    fn(user, coords, n_rays, values);
    uint32_t n_next = 0;
    for (uint32_t i = 0; i < n_rays; ++i) {
      pt_ray r = rays[i];
      r.tnear = 0.f; r.tfar = FLOAT_LARGE;                                     /* load (:126-129) */
      (void)intersect_box(&r.tnear, &r.tfar, r.org, r.dir, e.lo, e.hi);
      if (pt_shade(&e, &r, values[i]) && pt_take_sample(&e, &r)) rays[n_next++] = r;
      else { const float rgba[4] = { r.L.x, r.L.y, r.L.z, 1.f }; write_pixel(s, accumulation, frame, rgba, r.pidx); }
    }
    n_rays = n_next;
  }
  if (stats) *stats = st;
  free(rays); free(coords); free(values);
}

/* delta_tracking with USE_DELTA_TRACKING_ITER (method_pathtracing.cu:258-292): tentative collisions from hashit until a real one */
static int pt_delta_tracking(const pt_env* e, const float* vol, pt_ray* r, float* t_out, v3* albedo)
{
  float t = r->tnear;
  int found = 0;
  *albedo = v3_make(0, 0, 0);
  dda_init(&r->iter, v3_mul(r->org, e->rcp), v3_mul(r->dir, e->rcp), r->tnear, r->tfar, e->grid);
  while (pt_hashit(e->s, r, e->rcp, e->grid, e->density_scale, &t)) {
    const v3 c = v3_add(r->org, v3_scale(t, r->dir));
    const float sample = vnro_sample_volume(vol, e->s->vol_dims, c.x, c.y, c.z);
    float rgb[3], a;
    vnro_tfn_sample(&e->s->tfn, sample, rgb, &a);
    if (vnro_lcg_next(&r->rng) * r->majorant < a * e->density_scale) {
      *albedo = v3_make(rgb[0], rgb[1], rgb[2]);
      found = 1;
      break;
    }
  }
  *t_out = t;
  return found;
}

/* path tracer on a dense volume in one loop per pixel (rendering mode 13): path_tracing_kernel / path_tracing_traceray
 * (method_pathtracing.cu:420-510).  Unlike the streaming variant the interval IS reset before a bounce (:438-439). */
void vnro_render_pathtracing_monolithic(const vnro_scene* s, const float* vol, int row_lo, int row_hi, float* accumulation, float* frame)
{
  const camera_t cam = make_camera(s);
  const affine otw = affine_from(s->xfm);
  const affine wto = affine_inverse(&otw);
  pt_env e;
  e.s = s; e.wto = &wto;
  e.lo = v3_make(s->bbox_lo[0], s->bbox_lo[1], s->bbox_lo[2]);
  e.hi = v3_make(s->bbox_hi[0], s->bbox_hi[1], s->bbox_hi[2]);
  e.rcp = v3_make(1.0f / s->mc_spacings[0], 1.0f / s->mc_spacings[1], 1.0f / s->mc_spacings[2]);
  e.grid.x = s->mc_dims[0]; e.grid.y = s->mc_dims[1]; e.grid.z = s->mc_dims[2];
  e.density_scale = s->density_scale == 0.0f ? 1.0f : s->density_scale;
  e.light_dir_obj = xfm_vector(&wto, v3_normalize(v3_make(s->light_dir[0], s->light_dir[1], s->light_dir[2])));
  e.reset_interval = 1;
  for (int iy = row_lo; iy < row_hi; ++iy)
    for (int ix = 0; ix < s->width; ++ix) {
      const uint32_t pixel = (uint32_t)ix + (uint32_t)iy * (uint32_t)s->width;
      pt_ray r;
      const ray_t cr = compute_ray(s, &cam, &wto, pixel);
      r.tnear = 0.f; r.tfar = FLOAT_LARGE; r.pidx = pixel; r.org = cr.org; r.dir = cr.dir; r.shadow = 0;
      r.scatter_index = 0; r.majorant = 0.f; r.sample_coord = v3_make(0, 0, 0);
      r.L = v3_make(0, 0, 0); r.throughput = v3_make(1, 1, 1);
      vnro_lcg_init(&r.rng, (uint32_t)s->frame_index, pixel);
      while (intersect_box(&r.tnear, &r.tfar, r.org, r.dir, e.lo, e.hi)) {
        float t; v3 albedo;
        const int exited = !pt_delta_tracking(&e, vol, &r, &t, &albedo);
        if (r.shadow) {
          if (exited) r.L = v3_add(r.L, v3_scale(1.0f, r.throughput));   /* light_directional_rgb */
          r.tnear = 0.f; r.tfar = FLOAT_LARGE;
          const float s0 = vnro_lcg_next(&r.rng), s1 = vnro_lcg_next(&r.rng);
          r.dir = xfm_vector(&wto, pt_uniform_sample_sphere(1.f, s0, s1));
          r.shadow = 0;
        } else {
          if (exited) {
            if (r.scatter_index > 0) r.L = v3_add(r.L, v3_scale(1.5f, r.throughput));   /* light_ambient */
            break;
          }
          if (r.scatter_index > 4) {  /* russian_roulette */
            const float q = fminf(0.95f, fmaxf(fmaxf(r.throughput.x, r.throughput.y), r.throughput.z));
            if (vnro_lcg_next(&r.rng) > q) break;
            r.throughput = v3_make(r.throughput.x / q, r.throughput.y / q, r.throughput.z / q);
          }
          ++r.scatter_index;
          r.org = v3_add(r.org, v3_scale(t, r.dir));
          r.throughput = v3_mul(r.throughput, v3_scale(0.6f, albedo));
          r.tnear = 0.f; r.tfar = FLOAT_LARGE;
          r.dir = e.light_dir_obj;
          r.shadow = 1;
        }
      }
      const float rgba[4] = { r.L.x, r.L.y, r.L.z, 1.f };
      write_pixel(s, accumulation, frame, rgba, pixel);
    }
}

/* ------------------------------------------------------------------------ */
/* monolithic ground-truth marcher (mode 4 semantics, NO_SHADING)            */
/* ref: core/renderer/method_raymarching.cu:263-308, 401-536                 */
/* ------------------------------------------------------------------------ */

typedef struct {
  const vnro_scene* s; const float* vol; ray_t ray; float jitter, step, step_rcp;
  float alpha; v3 color;
  int gradient; v3 gs; const affine* otw; const affine* wto;
  int ssh, shadow;               /* SINGLE_SHADE_HEURISTIC camera ray (:455-462) / its transmittance ray (:364-398) */
  v3 highest_org, highest_color; float highest_alpha;
} mono_ctx;

static int mono_cell(void* c, i3 cell, float t0, float t1)
{
  mono_ctx* m = (mono_ctx*)c;
  const float r = opacity_upper_bound(m->s, cell);
  if (fabsf(r) <= FLT_EPSILON) return 1;
  /* sample_size_scaler, :263-268 */
  float ss = adaptive_sampling_rate(m->step, r);
  {
    const int32_t N = (int32_t)((t1 - t0) / ss + 1.0f);
    ss = (t1 - t0) / (float)N;
  }
  float tx = t0, ty = fminf(t1, t0 + ss);
  while (ty > tx) {
    const float t = (1.0f - m->jitter) * tx + m->jitter * ty;
    const v3 p = v3_add(m->ray.org, v3_scale(t, m->ray.dir));
    const float value = vnro_sample_volume(m->vol, m->s->vol_dims, p.x, p.y, p.z);
    float rgb[3], a;
    vnro_tfn_sample(&m->s->tfn, value, rgb, &a);
    a = opacity_correction(m->step_rcp, ty - tx, a);
    if (m->gradient) {  /* :440-454 with sampleGradient, raytracing.h:112-126: a step that would leave [0,1] is flipped */
      v3 stp = m->gs;
      if (p.x + stp.x > 1.0f - FLT_EPSILON) stp.x *= -1.0f;
      if (p.y + stp.y > 1.0f - FLT_EPSILON) stp.y *= -1.0f;
      if (p.z + stp.z > 1.0f - FLT_EPSILON) stp.z *= -1.0f;
      const float fgx = vnro_sample_volume(m->vol, m->s->vol_dims, p.x + stp.x, p.y, p.z);
      const float fgy = vnro_sample_volume(m->vol, m->s->vol_dims, p.x, p.y + stp.y, p.z);
      const float fgz = vnro_sample_volume(m->vol, m->s->vol_dims, p.x, p.y, p.z + stp.z);
      const v3 shaded = gradient_shade(m->s, m->otw, m->wto, m->ray.dir, value, fgx, fgy, fgz, stp, v3_make(rgb[0], rgb[1], rgb[2]));
      rgb[0] = shaded.x; rgb[1] = shaded.y; rgb[2] = shaded.z;
    }
    if (m->shadow) {  /* raymarching_transmittance: alpha += (1 - alpha) * sampleAlpha (:391-392) */
      m->alpha += (1.0f - m->alpha) * a;
      if (!(m->alpha < NEARLY_ONE)) return 0;
      tx = ty;
      ty = fminf(tx + ss, t1);
      continue;
    }
    if (m->ssh && m->highest_alpha < (1.0f - m->alpha) * a) {  /* :455-462 */
      m->highest_org = p;
      m->highest_color = v3_make(rgb[0], rgb[1], rgb[2]);
      m->highest_alpha = (1.0f - m->alpha) * a;
    }
    const float tr = 1.0f - m->alpha;
    m->color.x += tr * rgb[0] * a;
    m->color.y += tr * rgb[1] * a;
    m->color.z += tr * rgb[2] * a;
    m->alpha += tr * a;
    if (!(m->alpha < NEARLY_ONE)) return 0;
    tx = ty;
    ty = fminf(tx + ss, t1);
  }
  return 1;
}

void vnro_render_monolithic(const vnro_scene* s, const float* vol, int row_lo, int row_hi,
                            float* accumulation, float* frame)
{
  const camera_t cam = make_camera(s);
  const affine otw = affine_from(s->xfm);
  const affine wto = affine_inverse(&otw);
  const v3 lo = v3_make(s->bbox_lo[0], s->bbox_lo[1], s->bbox_lo[2]);
  const v3 hi = v3_make(s->bbox_hi[0], s->bbox_hi[1], s->bbox_hi[2]);
  const v3 rcp = v3_make(1.0f / s->mc_spacings[0], 1.0f / s->mc_spacings[1], 1.0f / s->mc_spacings[2]);
  const i3 grid = { s->mc_dims[0], s->mc_dims[1], s->mc_dims[2] };
  for (int iy = row_lo; iy < row_hi; ++iy)
    for (int ix = 0; ix < s->width; ++ix) {
      const uint32_t pixel = (uint32_t)ix + (uint32_t)iy * (uint32_t)s->width;
      vnro_lcg rng;
      vnro_lcg_init(&rng, (uint32_t)s->frame_index, pixel);
      mono_ctx m;
      m.s = s; m.vol = vol; m.ray = compute_ray(s, &cam, &wto, pixel);
      m.step = 1.0f / s->sampling_rate; m.step_rcp = s->sampling_rate;
      m.alpha = 0.0f; m.color = v3_make(0, 0, 0);
      m.gradient = s->shading_mode == 1;
      m.ssh = s->shading_mode == 2; m.shadow = 0;
      m.highest_org = v3_make(0, 0, 0); m.highest_color = v3_make(0, 0, 0); m.highest_alpha = 0.0f;
      m.gs = v3_make(1.0f / (float)s->vol_dims[0], 1.0f / (float)s->vol_dims[1], 1.0f / (float)s->vol_dims[2]);
      m.otw = &otw; m.wto = &wto;
      float t0 = 0.0f, t1 = FLOAT_LARGE;
      if (intersect_box(&t0, &t1, m.ray.org, m.ray.dir, lo, hi)) {
        m.jitter = vnro_lcg_next(&rng);   /* get_floats().x: the generator advances by two draws per get_floats() */
        (void)vnro_lcg_next(&rng);
        dda3(v3_mul(m.ray.org, rcp), v3_mul(m.ray.dir, rcp), t0, t1, grid, mono_cell, &m);
        if (m.ssh && m.highest_alpha > 0.0f) {  /* :471-484: one shadow ray from the strongest sample towards the light */
          const v3 ldir = xfm_vector(&wto, v3_normalize(v3_make(s->light_dir[0], s->light_dir[1], s->light_dir[2])));
          mono_ctx sh = m;
          sh.ray.org = m.highest_org; sh.ray.dir = ldir;
          sh.shadow = 1; sh.ssh = 0; sh.gradient = 0;
          sh.step = 2.0f * m.step;          /* raymarching_shadow_sampling_scale = 2 (instantvnr_types.h:137); opacityCorrection keeps self.step */
          sh.alpha = 0.0f;
          float s0 = 0.0f, s1 = FLOAT_LARGE;
          if (intersect_box(&s0, &s1, sh.ray.org, sh.ray.dir, lo, hi)) {
            sh.jitter = vnro_lcg_next(&rng);
            dda3(v3_mul(sh.ray.org, rcp), v3_mul(sh.ray.dir, rcp), s0, s1, grid, mono_cell, &sh);
          }
          const float transmittance = 1.0f - sh.alpha;
          const float k = 0.95f;            /* scivis_shading_scale */
          m.color.x = (1.0f - k) * m.color.x + k * (m.highest_color.x * m.alpha * transmittance);
          m.color.y = (1.0f - k) * m.color.y + k * (m.highest_color.y * m.alpha * transmittance);
          m.color.z = (1.0f - k) * m.color.z + k * (m.highest_color.z * m.alpha * transmittance);
        }
      }
      const float rgba[4] = { m.color.x, m.color.y, m.color.z, m.alpha };
      write_pixel(s, accumulation, frame, rgba, pixel);
    }
}

/* ------------------------------------------------------------------------ */
/* metrics                                                                   */
/* ------------------------------------------------------------------------ */

/* ref: core/network.cu:51-68 */
void vnro_generate_grid_coords(const int lower[3], const int size[3], const float rdims[3], float* coords)
{
  const size_t n = (size_t)size[0] * size[1] * size[2];
  const size_t stride = (size_t)size[0] * size[1];
  for (size_t i = 0; i < n; ++i) {
    const int x = lower[0] + (int)(i % (size_t)size[0]);
    const int y = lower[1] + (int)((i % stride) / (size_t)size[0]);
    const int z = lower[2] + (int)(i / stride);
    coords[3 * i + 0] = ((float)x + 0.5f) * rdims[0];
    coords[3 * i + 1] = ((float)y + 0.5f) * rdims[1];
    coords[3 * i + 2] = ((float)z + 0.5f) * rdims[2];
  }
}

/* ref: core/network.cu:450-471 (MSE in fp32 sums there; double here, same formula) */
double vnro_psnr(const float* pred, const float* ref, size_t n, float ref_min, float ref_max)
{
  double err = 0.0;
  for (size_t i = 0; i < n; ++i) {
    const double d = (double)pred[i] - (double)ref[i];
    err += d * d;
  }
  const double range = (double)ref_max - (double)ref_min;
  const double mse = err / (double)n;
  return 10.0 * log10(range * range / mse);
}

/* ------------------------------------------------------------------------ */
/* out-of-core training sampler.  ref: core/samplers/neural_sampler.cpp      */
/* ------------------------------------------------------------------------ */

/* ref: neural_sampler.cpp:143-156 read_typed_pointer */
static float ooc_read_typed(const uint8_t* buffer, size_t id, int type)
{
  switch (type) {
  case 0: return (float)((const uint8_t*)buffer)[id];
  case 1: return (float)((const int8_t*)buffer)[id];
  case 2: { uint16_t v; memcpy(&v, buffer + 2 * id, 2); return (float)v; }
  case 3: { int16_t v; memcpy(&v, buffer + 2 * id, 2); return (float)v; }
  case 4: { uint32_t v; memcpy(&v, buffer + 4 * id, 4); return (float)v; }
  case 5: { int32_t v; memcpy(&v, buffer + 4 * id, 4); return (float)v; }
  case 8: { float v; memcpy(&v, buffer + 4 * id, 4); return v; }
  default: { double v; memcpy(&v, buffer + 8 * id, 8); return (float)v; }
  }
}

static uint64_t ooc_flatten(const int index[3], const int grid[3])  /* :339-345 */
{
  return (uint64_t)index[0] + (uint64_t)index[1] * (uint64_t)grid[0] + (uint64_t)index[2] * (uint64_t)grid[1] * (uint64_t)grid[0];
}

static int ooc_imin(int a, int b) { return a < b ? a : b; }
static int ooc_imax(int a, int b) { return a > b ? a : b; }
static float ooc_clampf(float v, float lo, float hi) { return v < lo ? lo : (v > hi ? hi : v); }  /* gdt clamp = min(max(v, lo), hi) */

/* ref: neural_sampler.cpp:531-552 (RandomBuffer ctor) */
int vnro_ooc_geometry_make(const int dims[3], int type, vnro_ooc_geometry* g)
{
  const uint64_t STREAM_SIZE = 32 * 1024, ALIGNMENT = 512;
  switch (type) {
  case 0: case 1: g->elem = 1; break;
  case 2: case 3: g->elem = 2; break;
  case 4: case 5: case 8: g->elem = 4; break;
  case 12: g->elem = 8; break;
  default: return -1;
  }
  g->type = type;
  for (int d = 0; d < 3; ++d) g->dims[d] = dims[d];
  const uint64_t row = (uint64_t)dims[0] * g->elem;
  uint64_t rows = (STREAM_SIZE + row - 1) / row;
  if (rows > (uint64_t)dims[1]) rows = (uint64_t)dims[1];
  g->block_dims[0] = dims[0];
  g->block_dims[1] = (int)rows;
  g->block_dims[2] = ooc_imin(1, dims[2]);
  g->ghost_dims[0] = g->block_dims[0];
  g->ghost_dims[1] = ooc_imin(g->block_dims[1] + 2, dims[1]);
  g->ghost_dims[2] = ooc_imin(g->block_dims[2] + 2, dims[2]);
  g->index_space[0] = 1;
  g->index_space[1] = (dims[1] + g->block_dims[1] - 1) / g->block_dims[1];
  g->index_space[2] = (dims[2] + g->block_dims[2] - 1) / g->block_dims[2];
  const uint64_t bytes = (uint64_t)g->ghost_dims[0] * g->ghost_dims[1] * g->ghost_dims[2] * g->elem;
  g->block_size_aligned = (bytes + ALIGNMENT - 1) / ALIGNMENT * ALIGNMENT;
  return 0;
}

/* ref: neural_sampler.cpp:579-636 (submit_one_job); the asynchronous reads become memcpy from the in-memory file */
int vnro_ooc_load_block(const vnro_ooc_geometry* g, const uint8_t* file, const int block_index[3], vnro_ooc_block* b,
                        uint8_t* block_data)
{
  for (int d = 0; d < 3; ++d)
    if (block_index[d] < 0 || block_index[d] >= g->index_space[d]) return -1;
  int v0[3], v1[3], g0[3], g1[3], gd[3];
  for (int d = 0; d < 3; ++d) {
    v0[d] = block_index[d] * g->block_dims[d];
    v1[d] = ooc_imin(v0[d] + g->block_dims[d], g->dims[d]);
    g0[d] = ooc_imax(v0[d] - 1, 0);
    g1[d] = ooc_imin(v1[d] + 1, g->dims[d]);
    gd[d] = g1[d] - g0[d];
    b->index[d] = block_index[d];
    b->bounds_lo[d] = v0[d]; b->bounds_hi[d] = v1[d];
    b->ghost_lo[d] = g0[d]; b->ghost_hi[d] = g1[d];
  }
  b->offset = ooc_flatten(v0, g->dims);
  b->length = (uint64_t)(v1[0] - v0[0]) * (uint64_t)(v1[1] - v0[1]) * (uint64_t)(v1[2] - v0[2]);
  if (b->length == 0) return -1;
  if ((uint64_t)gd[0] * gd[1] * gd[2] * g->elem > g->block_size_aligned) return -1;
  for (int z = g0[2]; z < g1[2]; ++z) {
    const int slice_begin[3] = {g0[0], g0[1], z};
    const int rel[3] = {0, 0, z - g0[2]};
    const uint64_t off_b = ooc_flatten(rel, gd) * g->elem;
    const uint64_t off_f = ooc_flatten(slice_begin, g->dims) * g->elem;
    const uint64_t nbytes = (uint64_t)gd[0] * gd[1] * g->elem;
    memcpy(block_data + off_b, file + off_f, nbytes);
  }
  return 0;
}

/* ref: neural_sampler.cpp:302-329 trilinear_vkl; accessor(x, y, z) reads slab-local storage (access_voxel :653-663) and
 * normalises before interpolation (:1098-1102) */
typedef struct {
  const vnro_ooc_geometry* g;
  const vnro_ooc_block* b;
  const uint8_t* data;
  float lo, vscale;
} ooc_accessor;

static float ooc_access(const ooc_accessor* a, int x, int y, int z)
{
  const int rel[3] = {x - a->b->ghost_lo[0], y - a->b->ghost_lo[1], z - a->b->ghost_lo[2]};
  const int size[3] = {a->b->ghost_hi[0] - a->b->ghost_lo[0], a->b->ghost_hi[1] - a->b->ghost_lo[1], a->b->ghost_hi[2] - a->b->ghost_lo[2]};
  const float v = ooc_read_typed(a->data, (size_t)ooc_flatten(rel, size), a->g->type);
  return ooc_clampf((v - a->lo) * a->vscale, 0.f, 1.f);
}

static float ooc_trilinear_vkl(const float p[3], const int dims[3], const ooc_accessor* a)
{
  float w[3], iw[3];
  int i0[3], i1[3];
  for (int d = 0; d < 3; ++d) {
    const float pb = p[d] - 0.5f;
    w[d] = modff(pb, &iw[d]);
    i0[d] = ooc_imin(ooc_imax((int)iw[d], 0), dims[d] - 1);
    i1[d] = ooc_imin(ooc_imax(i0[d] + 1, 0), dims[d] - 1);
  }
  const float c000 = ooc_access(a, i0[0], i0[1], i0[2]);
  const float c001 = ooc_access(a, i1[0], i0[1], i0[2]);
  const float c010 = ooc_access(a, i0[0], i1[1], i0[2]);
  const float c011 = ooc_access(a, i1[0], i1[1], i0[2]);
  const float c100 = ooc_access(a, i0[0], i0[1], i1[2]);
  const float c101 = ooc_access(a, i1[0], i0[1], i1[2]);
  const float c110 = ooc_access(a, i0[0], i1[1], i1[2]);
  const float c111 = ooc_access(a, i1[0], i1[1], i1[2]);
  return (1 - w[0]) * (1 - w[1]) * (1 - w[2]) * c000 + w[0] * (1 - w[1]) * (1 - w[2]) * c001
       + (1 - w[0]) * w[1] * (1 - w[2]) * c010 + w[0] * w[1] * (1 - w[2]) * c011
       + (1 - w[0]) * (1 - w[1]) * w[2] * c100 + w[0] * (1 - w[1]) * w[2] * c101
       + (1 - w[0]) * w[1] * w[2] * c110 + w[0] * w[1] * w[2] * c111;
}

/* ref: neural_sampler.cpp:1066-1120 (the tbb::parallel_for body), random numbers supplied by the caller */
size_t vnro_ooc_sample(const vnro_ooc_geometry* g, const vnro_ooc_block* blocks, const uint8_t* block_data, uint64_t n_blocks,
                       float range_lo, float range_hi, const float* r_coords, const float* r_bidx, const float* r_vidx, size_t n,
                       const float lower[3], const float upper[3], float* coords, float* values)
{
  size_t out_of_range = 0;
  const float rfdims[3] = {1.f / (float)g->dims[0], 1.f / (float)g->dims[1], 1.f / (float)g->dims[2]};
  const float vscale = 1.f / (range_hi - range_lo);
  for (size_t i = 0; i < n; ++i) {
    uint64_t bidx = (uint64_t)(r_bidx[i] * (float)n_blocks);
    if (bidx >= n_blocks) { bidx = n_blocks - 1; ++out_of_range; }  /* "[aio] invalid block index" */
    const vnro_ooc_block* b = &blocks[bidx];
    uint64_t vidx = (uint64_t)(r_vidx[i] * (float)b->length);
    if (vidx >= b->length) { vidx = b->length - 1; ++out_of_range; }  /* "[aio] invalid voxel index" */
    const uint64_t index = b->offset + vidx;  /* locate_voxel */
    const uint64_t stride_y = (uint64_t)g->dims[0], stride_z = (uint64_t)g->dims[1] * (uint64_t)g->dims[0];
    const int voxel[3] = {(int)(index % stride_y), (int)((index % stride_z) / stride_y), (int)(index / stride_z)};  /* to_grid_index */
    float p[3], ccp[3];
    for (int d = 0; d < 3; ++d) {
      p[d] = r_coords[3 * i + d] + (float)voxel[d];
      coords[3 * i + d] = p[d] * rfdims[d] * (upper[d] - lower[d]) + lower[d];
      ccp[d] = ooc_clampf(p[d], 0.5f, (float)g->dims[d] - 0.5f);
    }
    const ooc_accessor acc = {g, b, block_data + bidx * g->block_size_aligned, range_lo, vscale};
    values[i] = ooc_trilinear_vkl(ccp, g->dims, &acc);
  }
  return out_of_range;
}

/* ref: neural_sampler.cpp:967-1035 with trilinear = false: nearest_vkl (:294-300) at the grid points */
void vnro_ooc_sample_grid(const vnro_ooc_geometry* g, const uint8_t* file, float range_lo, float range_hi, const int origin[3],
                          const int size[3], const float spacing[3], float* values)
{
  const float scale = 1.f / (range_hi - range_lo);
  const size_t n = (size_t)size[0] * size[1] * size[2];
  const uint64_t stride = (uint64_t)size[0] * size[1];
  for (size_t i = 0; i < n; ++i) {
    const int gp[3] = {origin[0] + (int)(i % (size_t)size[0]), origin[1] + (int)((i % stride) / (size_t)size[0]), origin[2] + (int)(i / stride)};
    int ip[3];
    for (int d = 0; d < 3; ++d) {
      const float fp = ((float)gp[d] + 0.5f) * spacing[d];
      ip[d] = (int)ooc_clampf(fp * (float)g->dims[d], 0.5f, (float)g->dims[d] - 0.5f);
    }
    const float v = ooc_read_typed(file, (size_t)ooc_flatten(ip, g->dims), g->type);
    values[i] = ooc_clampf((v - range_lo) * scale, 0.f, 1.f);
  }
}
