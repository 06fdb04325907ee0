// infer_tile.h — the evaluation of ONE wave-tile of 64 samples (hash-grid encode + MLP on the matrix cores) as device functions,
// shared by the evaluation kernels (network_infer.hip) and the in-shader ray marcher (render.hip), which evaluates the network
// inside its marching loop (method_raymarching.cu:981-1249 calls DeviceNeuralVolume::sample, tcnn_impl.cu:34-102, the same way).
// Design notes: network_infer.hip's header.
#pragma once
#include "grid_device.h"

namespace vnr {

struct float3_packed { float x, y, z; };

// ---- shapes -----------------------------------------------------------------------------------------------------------------
// FullyFusedMLP n_neurons W in {16, 32, 64, 128} (tcnn_impl.cu:315-347, tcnn_impl_network.cu:151-161 instantiate exactly these) on
// v_mfma_f32_32x32x16_f16, computed transposed: Y^T[W x samples] = Wgt[W x K] . X^T[K x samples].  A layer's output is MT tiles of
// 32 neurons; W = 16 is one tile whose rows 16 .. 31 are zero weights (the hidden k-step of 16 is exactly the instruction's K, so a
// layer costs one MFMA per 32 samples either way).  An accumulator tile (column = sample on the lane, rows = neurons in the
// registers) is, after the activation and the fp16 pack, the B operand of k-steps 2 m and 2 m + 1 of the next layer.
template <int W> struct MlpShape {
  static_assert(W == 16 || W == 32 || W == 64 || W == 128, "FullyFusedMLP widths");
  static constexpr int MT = W >= 32 ? W / 32 : 1;   // 32-row tiles of a layer's output
  static constexpr int RW = 32 * MT;                // rows of a layer in the weight image
  static constexpr int KS = W / 16;                 // k-steps of a hidden layer / of the last layer
  static constexpr int STEP = 16 * RW;              // halves of one k-step of a layer in the image: [h < 2][row < RW][j < 8]
  static constexpr int HIDDEN = KS * STEP;          // halves of a hidden layer
  static constexpr int LAST = KS * 16;              // halves of the last layer's row 0: [s < KS][h < 2][j < 8]
  static constexpr int WAVES = W == 128 ? 8 : 4;    // waves per block of the evaluation kernels (128: two waves per SIMD share one image)
};
__host__ __device__ inline uint32_t mlp_rows_padded(uint32_t W) { return W >= 32u ? W : 32u; }
__host__ __device__ inline uint32_t packed_mlp_halves(uint32_t in_width, uint32_t W, uint32_t n_hidden_matmuls)
{
  const uint32_t step = 16u * mlp_rows_padded(W);
  return (in_width / 16u) * step + n_hidden_matmuls * (W / 16u) * step + (W / 16u) * 16u;
}

// backward image of the MLP (network_infer.hip pack_mlp_kernel; network_train.hip mlp_backward_kernel)
__host__ __device__ inline uint32_t packedT_halves(uint32_t in_width, uint32_t W, uint32_t nh)
{
  const uint32_t ks = W / 16u;
  return ks * 16u + nh * ks * 16u * mlp_rows_padded(W) + ks * 16u * (((in_width + 31u) / 32u) * 32u);
}

// ---- activations ------------------------------------------------------------------------------------------------------------
// tcnn Activation as the reference dispatches it (tcnn_impl.cu:405-415, tcnn_device_api.h:274-285): None, Exponential, Sigmoid, ReLU,
// Squareplus, Softplus, applied to the fp16 result fragment (tcnn_threadblock.h:125,308,437,497).  EXTERNAL tcnn warp_activation
// (common_device.h of the v1.4 - 1.5 era): the function is evaluated in fp32 on the fp16 value and rounded back to fp16;
// K_ACT = 10 for Squareplus / Softplus.
enum : uint32_t { kActNone = 0, kActReLU = 1, kActExponential = 2, kActSigmoid = 3, kActSquareplus = 4, kActSoftplus = 5 };

__device__ __forceinline__ float act_forward_f32(float x, uint32_t act)
{
  switch (act) {
  case kActExponential: return __expf(x);
  case kActSigmoid: return 1.0f / (1.0f + __expf(-x));
  case kActSquareplus: { const float t = x * 10.0f; return 0.5f * (t + sqrtf(t * t + 4.0f)) / 10.0f; }
  case kActSoftplus: return logf(expf(x * 10.0f) + 1.0f) / 10.0f;
  default: return x;
  }
}
__device__ __forceinline__ half_t act_forward_f16(half_t v, uint32_t act)
{
  if (act == kActNone) return v;
  if (act == kActReLU) return (__builtin_bit_cast(unsigned short, v) & 0x8000u) ? (half_t)0.0f : v;
  return (half_t)act_forward_f32((float)v, act);
}
// EXTERNAL tcnn warp_activation_backward: the gradient of the activation from its OUTPUT y (the stored fp16 activation), every
// factor rounded to fp16 as tcnn's half arithmetic rounds it, times the incoming gradient d (one more fp16 rounding)
__device__ __forceinline__ half_t act_backward_f16(half_t d, half_t y, uint32_t act)
{
  switch (act) {
  case kActReLU: return y > (half_t)0.0f ? d : (half_t)0.0f;
  case kActExponential: return d * y;
  case kActSigmoid: return d * (half_t)(y * (half_t)(1.0f - (float)y));
  case kActSquareplus: { const float t = (float)y * 10.0f; return d * (half_t)(t * t / (t * t + 1.0f)); }
  case kActSoftplus: return d * (half_t)(1.0f - __expf(-(float)y * 10.0f));
  default: return d;
  }
}

__device__ __noinline__ half8_t act_general8(half8_t r, uint32_t act)
{
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = (half_t)act_forward_f32((float)r[j], act);
  return r;
}

template <bool GENERAL>
__device__ __forceinline__ half8_t pack_act(const f32x16& acc, int sh, uint32_t act)
{
  float8_t v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = acc[8 * sh + j];
  half8_t r = __builtin_convertvector(v, half8_t);
  if (act == kActReLU) {
    const half8_t zero = {0, 0, 0, 0, 0, 0, 0, 0};
    r = __builtin_elementwise_max(r, zero);
  } else if (GENERAL && act > kActReLU) {   // wave-uniform: the transcendental activations, out of line (GENERAL instances only: grid_device.h)
    r = act_general8(r, act);
  }
  return r;
}

// swaps the upper 32 lanes of a with the lower 32 lanes of b:  a' = [a.lo | b.lo],  b' = [a.hi | b.hi]
__device__ __forceinline__ void swap_halves(uint32_t& a, uint32_t& b)
{
  const auto r = __builtin_amdgcn_permlane32_swap(a, b, false, false);
  a = r[0];
  b = r[1];
}

__device__ __forceinline__ void swap_halves8(half8_t& p, half8_t& q)
{
  uint4_t a = __builtin_bit_cast(uint4_t, p), b = __builtin_bit_cast(uint4_t, q);
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    uint32_t x = a[i], y = b[i];
    swap_halves(x, y);
    a[i] = x;
    b[i] = y;
  }
  p = __builtin_bit_cast(half8_t, a);
  q = __builtin_bit_cast(half8_t, b);
}

template <int KS>
__device__ __forceinline__ void store_acts(half_t* row, const half8_t (&bf)[KS], uint32_t h)
{
  // element j of bf[s] on lane (r, h) is neuron 16 s + 8 (j>>2) + 4 h + (j&3)
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    *(half4_t*)(row + 16 * s + 4 * h) = half4_t{bf[s][0], bf[s][1], bf[s][2], bf[s][3]};
    *(half4_t*)(row + 16 * s + 8 + 4 * h) = half4_t{bf[s][4], bf[s][5], bf[s][6], bf[s][7]};
  }
}

// The MLP for ONE 32-sample column tile of the wave.  b1[s] = first-layer B fragments (k = 16 s + 8 h + j).
// Returns this lane's partial sum of the output neuron (its half of the W last-layer terms).
// `wgt`: the weight image (packed_mlp_halves), in LDS or - models whose image exceeds the LDS - in global memory.
template <int W, int S1, bool TRAIN, bool GENERAL>
__device__ __forceinline__ float mlp_column_tile(const half_t* __restrict__ wgt, const half8_t (&b1)[S1], uint32_t nh, uint32_t act,
                                                 uint32_t h, uint32_t r, half_t* acts_out, size_t n, uint32_t smp, bool smp_ok)
{
  typedef MlpShape<W> Sh;
  constexpr int MT = Sh::MT, RW = Sh::RW, KS = Sh::KS;
  f32x16 acc[MT];
#pragma unroll
  for (int m = 0; m < MT; ++m)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
#pragma unroll
  for (int s = 0; s < S1; ++s) {
#pragma unroll
    for (int m = 0; m < MT; ++m) {
      const half8_t a = *(const half8_t*)(wgt + ((s * 2 + h) * RW + m * 32 + r) * 8);
      acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b1[s], acc[m], 0, 0, 0);
    }
  }
  half8_t bf[KS];  // activations as next-layer B fragments, one per k-step
#pragma unroll
  for (int s = 0; s < KS; ++s) bf[s] = pack_act<GENERAL>(acc[s >> 1], s & 1, act);
  if (TRAIN && acts_out && smp_ok) store_acts<KS>(acts_out + (size_t)smp * W, bf, h);

  for (uint32_t layer = 0; layer < nh; ++layer) {
    const half_t* w = wgt + S1 * Sh::STEP + layer * Sh::HIDDEN;
#pragma unroll
    for (int m = 0; m < MT; ++m)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[m][e] = 0.0f;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
#pragma unroll
      for (int m = 0; m < MT; ++m) {
        const half8_t a = *(const half8_t*)(w + ((s * 2 + h) * RW + m * 32 + r) * 8);
        acc[m] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[s], acc[m], 0, 0, 0);
      }
    }
#pragma unroll
    for (int s = 0; s < KS; ++s) bf[s] = pack_act<GENERAL>(acc[s >> 1], s & 1, act);
    if (TRAIN && acts_out && smp_ok) store_acts<KS>(acts_out + ((size_t)(layer + 1) * n + smp) * W, bf, h);
  }

  // last layer: output neuron 0 only (the other 15 padded rows are never read)
  const half_t* wl = wgt + S1 * Sh::STEP + nh * Sh::HIDDEN;
  float part = 0.0f;
#pragma unroll
  for (int s = 0; s < KS; ++s) {
    const half8_t wv = *(const half8_t*)(wl + (s * 2 + h) * 8);
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const half2_t a2 = {bf[s][2 * q], bf[s][2 * q + 1]};
      const half2_t w2 = {wv[2 * q], wv[2 * q + 1]};
      part = __builtin_amdgcn_fdot2(a2, w2, part, false);
    }
  }
  return part;
}


// ---- encode: lane = sample, level wave-uniform -> feat[K_IN / 8] (the sample's K_IN features, fp16, zero padded) ---------------

template <int F, int K_IN, bool GENERAL = false>
__device__ __forceinline__ void encode_tile(const LevelInfo* levels, uint32_t n_levels, uint32_t interpolation, const table_rsrc_t& rsrc,
                                            const uint8_t* brick_image, float x, float y, float z, half8_t (&feat)[K_IN / 8],
                                            float quantize_threshold = 0.0f)
{
  constexpr int L_PAD = K_IN / F;   // levels incl. zero padding
  // The level table is re-read (scalar loads) every tile: making the pointer opaque per call keeps the
  // compiler from hoisting 16 x 6 loop-invariant scalars out of a persistent loop and spilling them.
  // (constant address space => scalar s_load_dwordx8 per level)
  typedef const __attribute__((address_space(4))) LevelInfo* const_levels_t;
  const LevelInfo* lvtab_generic = levels;
  asm volatile("" : "+s"(lvtab_generic));
  const const_levels_t lvtab = (const_levels_t)lvtab_generic;
  // wave-uniform by construction; say so, or hipcc wraps every buffer load in a waterfall loop
  auto level_consts = [&](int l) {
    LevelInfo lv;
    lv.scale = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, lvtab[l].scale)));
    lv.resolution = __builtin_amdgcn_readfirstlane(lvtab[l].resolution);
    lv.res2 = __builtin_amdgcn_readfirstlane(lvtab[l].res2);
    lv.size = __builtin_amdgcn_readfirstlane(lvtab[l].size);
    lv.offset = __builtin_amdgcn_readfirstlane(lvtab[l].offset);
    lv.hashed = __builtin_amdgcn_readfirstlane(lvtab[l].hashed);
    lv.brick = __builtin_amdgcn_readfirstlane(lvtab[l].brick);
    lv.pad1 = __builtin_amdgcn_readfirstlane(lvtab[l].pad1);
    return lv;
  };
#pragma unroll
  for (int l = 0; l < L_PAD; ++l) {
    half_t o[F];
#pragma unroll
    for (int f = 0; f < F; ++f) o[f] = (half_t)0.0f;
    if (l < (int)n_levels) encode_level_fast<F, GENERAL>(level_consts(l), interpolation, rsrc, brick_image, x, y, z, o, quantize_threshold);
#pragma unroll
    for (int f = 0; f < F; ++f) feat[(l * F + f) / 8][(l * F + f) % 8] = o[f];
    // keep the level constants (scalar registers) of at most four levels live at a time
    if ((l & 3) == 3) __builtin_amdgcn_sched_barrier(0);
  }
}

// ---- MLP on the wave's 64 samples: feat (consumed) -> this lane's sample's output, before the final rounding to fp16 ------------
// tile_base = index of the wave's first sample (training: where the activations are stored)
template <int W, int K_IN, bool TRAIN, bool GENERAL = false>
__device__ __forceinline__ float mlp_tile(const half_t* __restrict__ wgt, half8_t (&feat)[K_IN / 8], uint32_t nh, uint32_t act, uint32_t h,
                                          uint32_t r, half_t* acts_out, size_t n, uint32_t tile_base)
{
  constexpr int S1 = K_IN / 16;     // k-steps of the first layer
  // first layer operands: B = X^T via permlane32 swaps
  // before: lane (sample) holds chunks 2s (P) and 2s+1 (Q) of its own sample.
  // after : P = B fragment of column tile 0, Q = B fragment of column tile 1 (k = 16 s + 8 h + j).
#pragma unroll
  for (int s = 0; s < S1; ++s) swap_halves8(feat[2 * s], feat[2 * s + 1]);

  float part[2];
#pragma unroll
  for (int nt = 0; nt < 2; ++nt) {
    half8_t b1[S1];
#pragma unroll
    for (int s = 0; s < S1; ++s) b1[s] = feat[2 * s + nt];
    const uint32_t smp = tile_base + 32u * nt + r;
    part[nt] = mlp_column_tile<W, S1, TRAIN, GENERAL>(wgt, b1, nh, act, h, r, acts_out, n, smp, smp < n);
  }

  // combine the two lane halves: lanes < 32 get column tile 0, lanes >= 32 column tile 1
  uint32_t p0 = __builtin_bit_cast(uint32_t, part[0]), p1 = __builtin_bit_cast(uint32_t, part[1]);
  swap_halves(p0, p1);
  return __builtin_bit_cast(float, p0) + __builtin_bit_cast(float, p1);
}

// the network's output as the reference hands it out: produced in half precision, the output activation applied to the half
// (tcnn_threadblock.h:497), then cast to float (tcnn_impl.cu:421-431)
template <bool GENERAL = false>
__device__ __forceinline__ float finish_output(float y, uint32_t out_act)
{
  if (!GENERAL) return (float)(half_t)y;
  return (float)act_forward_f16((half_t)y, out_act);
}

// What a kernel other than the evaluation kernels needs to evaluate the network itself (Network::tile_net): every model whose weight
// image fits the LDS, i.e. all the reference's in-shader shapes (widths 16 / 32 / 64, method_raymarching.cu:1192-1244) and 128 too
struct TileNet {
  const LevelInfo* levels;     // per-level constants (the brick variant when the image is in use)
  uint32_t n_levels, interpolation;
  const half_t* table;         // grid part of the parameter blob
  uint32_t table_bytes;
  const uint8_t* brick_image;  // or null
  const half_t* packed_mlp;    // LDS image of the weights (pack_mlp_kernel), lds_halves halves
  uint32_t lds_halves, n_hidden_matmuls, activation, output_activation;
  uint32_t n_features, in_width, width;
  uint32_t general;            // the model needs the GENERAL instances (Network::common_kind is false)
  float quantize_threshold;
};

// the network at this lane's point, all 64 lanes of the wave taking part (inactive lanes pass any in-domain point): the value
// fused_infer_kernel<F, K_IN, W, 0, GENERAL> writes for it, bit for bit (the same encode_tile / mlp_tile / finish_output calls)
template <int F, int K_IN, int W, bool GENERAL>
__device__ __forceinline__ float eval_tile(const TileNet& net, const half_t* __restrict__ lds, const table_rsrc_t& rsrc, float x, float y, float z)
{
  const uint32_t lane = threadIdx.x & 63u;
  half8_t feat[K_IN / 8];
  encode_tile<F, K_IN, GENERAL>(net.levels, net.n_levels, net.interpolation, rsrc, net.brick_image, x, y, z, feat, GENERAL ? net.quantize_threshold : 0.0f);
  const float v = mlp_tile<W, K_IN, false, GENERAL>(lds, feat, net.n_hidden_matmuls, net.activation, lane >> 5, lane & 31u, nullptr, 0, 0);
  return finish_output<GENERAL>(v, net.output_activation);
}

}  // namespace vnr
