// network_train.hip — training step of the hash-grid + MLP network on gfx950.
//
// Replaces tcnn `Trainer::training_step` as called from core/networks/tcnn_network.h:223-252 (the source is
// in the un-vendored tiny-cuda-nn submodule: EXTERNAL).  Restated from the published upstream design:
//   forward (keeps fp16 features + hidden activations) -> L1/L2 loss with loss scale 128 -> MLP backward ->
//   weight gradients -> hash-grid backward (scatter-add) -> Adam (fp32 master weights, per-parameter step
//   count, zero-gradient grid entries skipped, l2_reg on matrix weights only) under ExponentialDecay.
// Chosen for MI355X: the gradient of the whole parameter blob lives in ONE half-precision buffer (tcnn's own gradient
// precision; packed fp16 atomics), so a data-parallel run exchanges a single tensor; the MLP backward and the weight gradients run on MFMA
// with the forward's transposed register-resident scheme; the dense coarse levels scatter through LDS tiles; since round 5 the weight
// gradients and that LDS scatter run on a side stream beside a persistent atomic scatter of the hashed levels.
#include "infer_kernel.h"

namespace vnr {

void launch_fused(int mode, const GridDevice& grid, uint32_t in_width, const FusedMlp& mlp, const LevelInfo* d_levels, const uint16_t* table, size_t table_bytes,
                  const float* coords, float* out, uint16_t* features_out, uint16_t* acts_out, size_t n, const uint32_t* d_n, size_t n_max, hipStream_t s,
                  const uint32_t* d_dest = nullptr, uint32_t queue_out_stride = 0, const uint8_t* brick_image = nullptr,
                  uint32_t sharers = 1, const struct PackArgs* pack = nullptr);

// ------------------------------------------------------------------------------------------------ pcg32
struct Pcg32 {
  uint64_t state, inc;
  __host__ __device__ Pcg32(uint64_t initstate, uint64_t initseq = 0xda3e39cb94b95bdbULL)
  {
    state = 0u;
    inc = (initseq << 1u) | 1u;
    next_uint();
    state += initstate;
    next_uint();
  }
  __host__ __device__ uint32_t next_uint()
  {
    const uint64_t old = state;
    state = old * 0x5851f42d4c957f2dULL + inc;
    const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u);
    const uint32_t rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
  }
  __host__ __device__ float next_float()
  {
    const uint32_t u = (next_uint() >> 9) | 0x3f800000u;
    return __builtin_bit_cast(float, u) - 1.0f;
  }
  __host__ __device__ void advance(uint64_t delta)
  {
    uint64_t cur_mult = 0x5851f42d4c957f2dULL, cur_plus = inc, acc_mult = 1u, acc_plus = 0u;
    while (delta > 0) {
      if (delta & 1) { acc_mult *= cur_mult; acc_plus = acc_plus * cur_mult + cur_plus; }
      cur_plus = (cur_mult + 1) * cur_plus;
      cur_mult *= cur_mult;
      delta >>= 1;
    }
    state = acc_mult * state + acc_plus;
  }
};

// EXTERNAL tcnn init: MLP Xavier-uniform per weight matrix, grid uniform(-1e-4, 1e-4)
__global__ void init_params_kernel(OptState* __restrict__ state, half_t* __restrict__ params, size_t n_mlp, size_t n_total,
                                   uint32_t in_width, uint32_t n_hidden_matmuls, uint32_t width, uint64_t seed)
{
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t e0 = t * 4;
  if (e0 >= n_total) return;
  Pcg32 rng(seed);
  rng.advance(e0);
  const size_t first = (size_t)width * in_width, hidden_end = first + (size_t)n_hidden_matmuls * width * width;
  for (size_t e = e0; e < e0 + 4 && e < n_total; ++e) {
    const float u = rng.next_float();
    float scale;
    if (e < first) scale = sqrtf(6.0f / (float)(in_width + width));
    else if (e < hidden_end) scale = sqrtf(6.0f / (float)(width + width));
    else if (e < n_mlp) scale = sqrtf(6.0f / (float)(width + 16));
    else scale = 1e-4f;
    const float v = u * (2.0f * scale) - scale;
    state[e] = OptState{v, 0.0f, 0.0f, 0u};
    params[e] = (half_t)v;
  }
}

void launch_init_params(OptState* state, uint16_t* params, size_t n_mlp, size_t n_total, uint32_t in_width,
                        uint32_t n_hidden_matmuls, uint32_t width, uint64_t seed, hipStream_t s)
{
  const size_t threads = (n_total + 3) / 4;
  init_params_kernel<<<div_round_up(threads, 256), 256, 0, s>>>(state, (half_t*)params, n_mlp, n_total, in_width,
                                                                n_hidden_matmuls, width, seed);
  VNR_HIP_CHECK(hipGetLastError());
}

// master weights <- fp16 parameters (parameters loaded from a file, or training state allocated after the fact)
__global__ void master_from_f16_kernel(const half_t* __restrict__ in, OptState* __restrict__ state, size_t n, bool reset_optimizer)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    if (reset_optimizer) state[i] = OptState{(float)in[i], 0.0f, 0.0f, 0u};
    else state[i].master = (float)in[i];
  }
}
void launch_master_from_f16(const uint16_t* params, OptState* state, size_t n, bool reset_optimizer, hipStream_t s)
{
  master_from_f16_kernel<<<min(div_round_up(n, 256), 4096u), 256, 0, s>>>((const half_t*)params, state, n, reset_optimizer);
  VNR_HIP_CHECK(hipGetLastError());
}

// ------------------------------------------------------------------------------------------------ loss
// EXTERNAL tcnn L1Loss / L2Loss: values = |d|/N (d^2/N), gradient = loss_scale * sign(d)/N (2 d/N), stored fp16.  With an output
// activation the network's backward pass first takes that gradient through it, from the OUTPUT values (EXTERNAL tcnn
// FullyFusedMLP::backward -> activation_backward_output_gpu): folded in here.
__global__ void loss_grad_kernel(const float* __restrict__ y, const float* __restrict__ target, uint32_t n, uint32_t loss_type, uint32_t out_act,
                                 half_t* __restrict__ dy, float* __restrict__ loss_partials)
{
  __shared__ float red[256];
  float acc = 0.0f;
  const float inv_n = 1.0f / (float)n;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float d = y[i] - target[i];
    float g;
    if (loss_type == 0) { acc += fabsf(d) * inv_n; g = copysignf(1.0f, d); }
    else { acc += d * d * inv_n; g = 2.0f * d; }
    dy[i] = act_backward_f16((half_t)((float)kLossScale * g * inv_n), (half_t)y[i], out_act);
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss_partials[blockIdx.x] = red[0];
}

// ------------------------------------------------------------------------------------------------ MLP backward (MFMA)
// The backward image of the weights (layout: network_infer.hip pack_mlp_kernel) is packed together with the forward image
// whenever the parameters change; kk = 16 s + 8 (j>>2) + 4 h + (j&3) is the k-order of an accumulator tile reused as B operand.
struct BackwardArgs {
  const half_t* packedT;
  const half_t* dy;      // [n] loss-scaled dL/dy
  const half_t* acts;    // [(nh+1)][n][W]
  half_t* d_out;         // [(nh+1)][n][W]  dL/d(pre-activation) of every hidden layer output
  half_t* dfeat;         // [n][in_width]
  uint32_t n, nh, activation, in_width, lds_halves;
  uint32_t weights_global;   // GENERAL instances: the image exceeds the LDS and is read from global memory
};

__device__ __forceinline__ half8_t load_frag_rowmajor(const half_t* row, int s, uint32_t h)
{
  const half4_t lo = *(const half4_t*)(row + 16 * s + 4 * h);
  const half4_t hi = *(const half4_t*)(row + 16 * s + 8 + 4 * h);
  return half8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
__device__ __forceinline__ void store_frag_rowmajor(half_t* row, int s, uint32_t h, const half8_t& v)
{
  *(half4_t*)(row + 16 * s + 4 * h) = half4_t{v[0], v[1], v[2], v[3]};
  *(half4_t*)(row + 16 * s + 8 + 4 * h) = half4_t{v[4], v[5], v[6], v[7]};
}
__device__ __forceinline__ half8_t pack_plain(const f32x16& acc, int sh)
{
  float8_t v;
#pragma unroll
  for (int j = 0; j < 8; ++j) v[j] = acc[8 * sh + j];
  return __builtin_convertvector(v, half8_t);
}
// the gradient through the activation, from the stored OUTPUT of the layer (infer_tile.h act_backward_f16)
__device__ __noinline__ half8_t act_backward_general8(half8_t d, half8_t a, uint32_t act)
{
#pragma unroll
  for (int j = 0; j < 8; ++j) d[j] = act_backward_f16(d[j], a[j], act);
  return d;
}
template <bool GENERAL>   // GENERAL: the transcendental activations too (instances of their own: grid_device.h gather_corners)
__device__ __forceinline__ half8_t act_backward8(const half8_t& d, const half8_t& a, uint32_t act)
{
  if (act == kActNone) return d;
  if (GENERAL && act > kActReLU) return act_backward_general8(d, a, act);
  half8_t r;
#pragma unroll
  for (int j = 0; j < 8; ++j) r[j] = a[j] > (half_t)0.0f ? d[j] : (half_t)0.0f;
  return r;
}

template <int W, int MTF, bool GENERAL>  // MTF = number of 32-row tiles of the feature gradient (roundup(in_width, 32) / 32)
__global__ void __launch_bounds__(256) mlp_backward_kernel(const BackwardArgs args)
{
  typedef MlpShape<W> Sh;
  constexpr int MT = Sh::MT, RW = Sh::RW, KS = Sh::KS;
  constexpr int NTB = W == 128 ? 1 : 2;   // 32-sample column tiles in flight (128 neurons: one, its accumulators are 4 tiles of registers already)
  extern __shared__ __attribute__((aligned(16))) half_t lds[];
  constexpr bool CAN_GLOBAL = GENERAL;   // an image beyond 160 KiB: 128 neurons from 6 hidden layers on, 64 from ~20, 32 from ~75, 16 from ~300
  const bool wglobal = CAN_GLOBAL && args.weights_global != 0u;
  if (!wglobal) {
    const uint4_t* src = (const uint4_t*)args.packedT;
    uint4_t* dst = (uint4_t*)lds;
    for (uint32_t i = threadIdx.x; i < args.lds_halves / 8; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const uint32_t h = lane >> 5, r = lane & 31u;
  const uint32_t n = args.n, nh = args.nh;
  const uint32_t act = args.activation;
  const uint32_t n_tiles = (n + 63u) >> 6;
  constexpr int RP = MTF * 32;

  // the body once per address space of the image (inlined twice: ds_read_b128 against global_load_dwordx4)
  auto run = [&](const half_t* __restrict__ img) __attribute__((always_inline)) {
  for (uint32_t tile = blockIdx.x * 4u + wave; tile < n_tiles; tile += gridDim.x * 4u) {
#pragma unroll
    for (int nt0 = 0; nt0 < 2; nt0 += NTB) {
      half8_t bf[KS][NTB];
      uint32_t smp[NTB];
      bool ok[NTB];
      // ---- through the last layer: d_nh[k] = Wl[0][k] * dy, through the activation of a_nh ----------------
#pragma unroll
      for (int q = 0; q < NTB; ++q) {
        smp[q] = tile * 64u + 32u * (uint32_t)(nt0 + q) + r;
        ok[q] = smp[q] < n;
        const uint32_t sc = ok[q] ? smp[q] : n - 1u;
        const float g = (float)args.dy[sc];
        const half_t* arow = args.acts + ((size_t)nh * n + sc) * W;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
          const half8_t wv = *(const half8_t*)(img + (s * 2 + h) * 8);
          half8_t d;
#pragma unroll
          for (int j = 0; j < 8; ++j) d[j] = (half_t)((float)wv[j] * g);
          d = act_backward8<GENERAL>(d, load_frag_rowmajor(arow, s, h), act);
          bf[s][q] = d;
          if (ok[q]) store_frag_rowmajor(args.d_out + ((size_t)nh * n + smp[q]) * W, s, h, d);
        }
      }
      // ---- hidden layers, last to first: d_l^T = Wh_l^T . d_{l+1}^T, through the activation --------------
      for (int layer = (int)nh - 1; layer >= 0; --layer) {
        const half_t* w = img + Sh::LAST + layer * Sh::HIDDEN;
        f32x16 acc[MT][NTB];
#pragma unroll
        for (int m = 0; m < MT; ++m)
#pragma unroll
          for (int q = 0; q < NTB; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][q][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int m = 0; m < MT; ++m) {
            const half8_t a = *(const half8_t*)(w + ((s * 2 + h) * RW + m * 32 + r) * 8);
#pragma unroll
            for (int q = 0; q < NTB; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[s][q], acc[m][q], 0, 0, 0);
          }
#pragma unroll
        for (int q = 0; q < NTB; ++q) {
          const uint32_t sc = ok[q] ? smp[q] : n - 1u;
          const half_t* arow = args.acts + ((size_t)layer * n + sc) * W;
#pragma unroll
          for (int s = 0; s < KS; ++s) {
            half8_t d = pack_plain(acc[s >> 1][q], s & 1);
            d = act_backward8<GENERAL>(d, load_frag_rowmajor(arow, s, h), act);
            bf[s][q] = d;
            if (ok[q]) store_frag_rowmajor(args.d_out + ((size_t)layer * n + smp[q]) * W, s, h, d);
          }
        }
      }
      // ---- feature gradient: dfeat^T = W1^T . d_0^T -------------------------------------------------
      {
        const half_t* w = img + Sh::LAST + nh * Sh::HIDDEN;
        f32x16 acc[MTF][NTB];
#pragma unroll
        for (int m = 0; m < MTF; ++m)
#pragma unroll
          for (int q = 0; q < NTB; ++q)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[m][q][e] = 0.0f;
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
          for (int m = 0; m < MTF; ++m) {
            const half8_t a = *(const half8_t*)(w + ((s * 2 + h) * RP + m * 32 + r) * 8);
#pragma unroll
            for (int q = 0; q < NTB; ++q) acc[m][q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bf[s][q], acc[m][q], 0, 0, 0);
          }
#pragma unroll
        for (int q = 0; q < NTB; ++q) {
          if (!ok[q]) continue;
          half_t* row = args.dfeat + (size_t)smp[q] * args.in_width;
#pragma unroll
          for (int m = 0; m < MTF; ++m)
#pragma unroll
            for (int sh = 0; sh < 2; ++sh) {
              const half8_t d = pack_plain(acc[m][q], sh);
              const uint32_t f0 = 32 * m + 16 * sh;  // features f0 + 4h + {0..3} and f0 + 8 + 4h + {0..3}
              if (f0 < args.in_width) {
                *(half4_t*)(row + f0 + 4 * h) = half4_t{d[0], d[1], d[2], d[3]};
                *(half4_t*)(row + f0 + 8 + 4 * h) = half4_t{d[4], d[5], d[6], d[7]};
              }
            }
        }
      }
    }
  }
  };
  if (CAN_GLOBAL && wglobal) run(args.packedT); else run((const half_t*)lds);
}

// ------------------------------------------------------------------------------------------------ weight gradients
// dW[out][in] = sum_b d[b][out] * x[b][in].  blockIdx.y selects the matrix: 0 = first layer (x = features),
// 1..nh = hidden (x = acts[l-1]), nh+1 = last layer row 0 (d = dy, x = acts[nh]).
struct WGradArgs {
  const half_t* features;  // [n][in_width]
  const half_t* acts;      // [(nh+1)][n][W]
  const half_t* d_all;     // [(nh+1)][n][W]
  const half_t* dy;        // [n]
  float* slab;             // [blocks][n_mlp] partial sums (fp32), summed by weight_grad_reduce_kernel into the fp16 gradient blob
  uint32_t n, nh, in_width, n_mlp;
};

// On the matrix cores: dW^T is a [W x in] product whose reduction dimension is the BATCH, so both operands are needed
// transposed ([neuron][sample] with 8 consecutive samples per lane) while the forward / backward kernels leave them row-major
// [sample][W].  A block stages 64 samples of d and x row-major in LDS (rows padded by 8 halves: lanes r = 0..31 of a transposed read
// touch 16 consecutive dwords, the two lane halves rows 8 apart = 32 banks apart); every wave owns 32 x 32 tiles of the product (not
// a share of the samples: summing the waves' partial tiles with ds_add_f32 took 12 us per block, four times the products) and runs
// the stage's four k-steps of v_mfma_f32_32x32x16_f16 on them; products of fewer than four tiles (16 / 32 neurons, narrow inputs)
// split the stage's k-steps over the waves of a tile instead and add up through LDS once at the end.  All loads of the block's 256
// samples are issued before the first product; the tiles leave as plain stores into the block's row of a slab [blocks][n_mlp], which
// weight_grad_reduce_kernel sums.  fp16 products are exact in the fp32 accumulator; only the order of the sums differs from a loop.
constexpr int kWgStage = 64;           // samples per stage
constexpr int kWgStages = 4;           // stages per block: a block owns 256 samples and issues ALL their loads before the first product

template <int W, int NT>  // NT: 32-column tiles of the input (in_width, or W for the hidden layers, padded to 32 NT)
__global__ void __launch_bounds__(256) weight_grad_mfma_kernel(const WGradArgs args, uint32_t layer_lo)
{
  typedef MlpShape<W> Sh;
  constexpr int MT = Sh::MT;
  constexpr int DS = W + 8;            // halves per staged row of d
  constexpr int DC = W / 8;            // uint4 per row of d
  constexpr int DQ = (kWgStage * DC + 255) / 256;
  constexpr int XS = 32 * NT + 8;      // halves per staged row of x
  constexpr int XC = 4 * NT;           // uint4 per row of x
  constexpr int XQ = (kWgStage * XC + 255) / 256;
  __shared__ __attribute__((aligned(16))) half_t sd[kWgStage * DS];
  __shared__ __attribute__((aligned(16))) half_t sx[kWgStage * XS];
  __shared__ float red[3 * 16 * 64];   // last layer: [sample lanes][W] (2048 floats); tile sums: [contributors - 1][tiles][16 registers][64 lanes]
  const uint32_t layer = layer_lo + blockIdx.y;
  const uint32_t n = args.n, nh = args.nh;
  const uint32_t blk0 = blockIdx.x * (uint32_t)(kWgStage * kWgStages);
  if (blk0 >= n) return;
  const uint32_t blk_end = min(n, blk0 + (uint32_t)(kWgStage * kWgStages));
  float* slab = args.slab + (size_t)blockIdx.x * args.n_mlp;   // this block's partial sums, tcnn order
  const size_t first_sz = (size_t)W * args.in_width;

  if (layer == nh + 1) {
    // last layer: dWl[0][k] = sum_b dy[b] * a_nh[b][k].  Thread = (8 neurons, one of SL sample lanes): every load of the block's
    // 256 rows is in flight at once (one dependent load per sample was 64 latencies in a row: the longest block of the launch)
    constexpr uint32_t CH = W / 8, SL = 256 / CH, ROWS = (kWgStage * kWgStages) / SL;
    const uint32_t c = threadIdx.x % CH, sr = threadIdx.x / CH;
    float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    half8_t av[ROWS];
    float gv[ROWS];
#pragma unroll
    for (uint32_t q = 0; q < ROWS; ++q) {
      const uint32_t b = blk0 + sr + SL * q;
      const bool ok = b < blk_end;
      av[q] = ok ? *(const half8_t*)(args.acts + ((size_t)nh * n + b) * W + c * 8) : half8_t{0, 0, 0, 0, 0, 0, 0, 0};
      gv[q] = ok ? (float)args.dy[b] : 0.0f;
    }
#pragma unroll
    for (uint32_t q = 0; q < ROWS; ++q)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[j] = __builtin_fmaf(gv[q], (float)av[q][j], acc[j]);
#pragma unroll
    for (int j = 0; j < 8; ++j) red[sr * W + c * 8 + j] = acc[j];
    __syncthreads();
    if (threadIdx.x < (uint32_t)W) {
      float t = 0.0f;
      for (uint32_t q = 0; q < SL; ++q) t += red[q * W + threadIdx.x];
      slab[first_sz + (size_t)nh * W * W + threadIdx.x] = t;
    }
    return;
  }
  const uint32_t in_w = layer == 0 ? args.in_width : (uint32_t)W;   // a multiple of 8 (rows are read as uint4)
  const uint32_t xc = in_w / 8;
  const half_t* dsrc = args.d_all + (size_t)layer * n * W;
  const half_t* xsrc = layer == 0 ? args.features : args.acts + (size_t)(layer - 1) * n * W;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6, h = lane >> 5, r = lane & 31u;
  constexpr int T = MT * NT;                   // 32 x 32 tiles of the product [32 MT out][32 NT in]
  constexpr int TW = T >= 4 ? T / 4 : 1;       // tiles per wave (same 32 output rows, TW column tiles)
  constexpr int WPT = T >= 4 ? 1 : 4 / T;      // waves per tile: with fewer than four tiles the waves of a tile split the stage's k-steps
  constexpr int KSW = 4 / WPT;                 // k-steps of a stage per wave
  const uint32_t t0 = T >= 4 ? wave * TW : wave % (uint32_t)T;
  const uint32_t m = t0 / NT, nt0 = t0 % NT;
  const uint32_t part = T >= 4 ? 0u : wave / (uint32_t)T;   // which share of the k-steps
  const uint32_t kbase = part * KSW;

  uint4_t rd[kWgStages][DQ], rx[kWgStages][XQ];
#pragma unroll
  for (int st = 0; st < kWgStages; ++st) {
    const uint32_t b0 = blk0 + (uint32_t)(st * kWgStage);
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      const uint32_t e = threadIdx.x + 256u * q, row = e / DC, c = e % DC, b = b0 + row;
      rd[st][q] = (row < (uint32_t)kWgStage && b < blk_end) ? *(const uint4_t*)(dsrc + (size_t)b * W + c * 8) : uint4_t{0, 0, 0, 0};
    }
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const uint32_t e = threadIdx.x + 256u * q, row = e / XC, c = e % XC, b = b0 + row;
      rx[st][q] = (row < (uint32_t)kWgStage && c < xc && b < blk_end) ? *(const uint4_t*)(xsrc + (size_t)b * in_w + c * 8) : uint4_t{0, 0, 0, 0};
    }
  }

  f32x16 acc[TW];
#pragma unroll
  for (int t = 0; t < TW; ++t)
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[t][e] = 0.0f;
  const bool row_ok = W >= 32 || r < (uint32_t)W;   // 16 neurons: rows 16 .. 31 of the one tile do not exist

#pragma unroll
  for (int st = 0; st < kWgStages; ++st) {
    if (st) __syncthreads();   // the previous stage has been consumed
#pragma unroll
    for (int q = 0; q < DQ; ++q) {
      const uint32_t e = threadIdx.x + 256u * q, row = e / DC, c = e % DC;
      if (row < (uint32_t)kWgStage) *(uint4_t*)(sd + row * DS + c * 8) = rd[st][q];
    }
#pragma unroll
    for (int q = 0; q < XQ; ++q) {
      const uint32_t e = threadIdx.x + 256u * q, row = e / XC, c = e % XC;
      if (row < (uint32_t)kWgStage) *(uint4_t*)(sx + row * XS + c * 8) = rx[st][q];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < KSW; ++ks) {
      const uint32_t k0 = 16u * (kbase + (uint32_t)ks) + 8u * h;
      half8_t a, bx[TW];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        a[j] = row_ok ? sd[(k0 + j) * DS + m * 32 + r] : (half_t)0.0f;
#pragma unroll
        for (int t = 0; t < TW; ++t) bx[t][j] = sx[(k0 + j) * XS + (nt0 + t) * 32 + r];
      }
#pragma unroll
      for (int t = 0; t < TW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, bx[t], acc[t], 0, 0, 0);
    }
  }
  if (T < 4) {   // the other shares of the k-steps hand their sums to the tile's first wave (plain LDS traffic, 4 KB each)
    __syncthreads();
    if (part > 0) {
#pragma unroll
      for (int e = 0; e < 16; ++e) red[(((part - 1u) * T + t0) * 16 + e) * 64 + lane] = acc[0][e];
    }
    __syncthreads();
    if (part > 0) return;
#pragma unroll
    for (int c = 1; c < WPT; ++c)
#pragma unroll
      for (int e = 0; e < 16; ++e) acc[0][e] += red[(((uint32_t)(c - 1) * T + t0) * 16 + e) * 64 + lane];
  }
  // element (out, in) of tile (m, nt) sits in lane (in = 32 nt + r), register e with out = 32 m + 8 (e / 4) + 4 h + e % 4
  float* g = slab + (layer == 0 ? 0 : first_sz + (size_t)(layer - 1) * W * W);
#pragma unroll
  for (int t = 0; t < TW; ++t) {
    const uint32_t col = 32u * (nt0 + t) + r;
    if (col < in_w) {
#pragma unroll
      for (int e = 0; e < 16; ++e) {
        const uint32_t out = 32u * m + 8u * (e >> 2) + 4u * h + (e & 3);
        if (W >= 32 || out < (uint32_t)W) g[(size_t)out * in_w + col] = acc[t][e];
      }
    }
  }
}

// grads[p] += sum over the blocks' partial sums, in block order: the MLP's gradient does not depend on the order in which atomics
// arrive.  (Float atomics of every block into the same 64 rows run at a fourteenth of the atomic rate, MI355X_MICROARCH.md "Global
// float atomics", contention row: 128 blocks x 16 KB per matrix took longer than the products.)
__global__ void __launch_bounds__(256) weight_grad_reduce_kernel(const float* __restrict__ slab, uint32_t n_blocks, uint32_t n_mlp, half_t* __restrict__ grads)
{
  // 64 parameters x 4 interleaved groups of slab rows per block, 16 loads in flight per thread: the sum is short, its loads' latency is all it costs
  __shared__ float part[4][64];
  const uint32_t tx = threadIdx.x & 63u, g = threadIdx.x >> 6;
  const uint32_t p = blockIdx.x * 64u + tx;
  float acc = 0.0f;
  if (p < n_mlp) {
    uint32_t b = g;
    for (; b + 60 < n_blocks; b += 64) {
      float v[16];
#pragma unroll
      for (int j = 0; j < 16; ++j) v[j] = slab[(size_t)(b + 4 * j) * n_mlp + p];
#pragma unroll
      for (int j = 0; j < 16; ++j) acc += v[j];
    }
    for (; b < n_blocks; b += 4) acc += slab[(size_t)b * n_mlp + p];
  }
  part[g][tx] = acc;
  __syncthreads();
  // the sum over the batch in fp32, rounded to the gradient's half precision once (tcnn's gradient matrices are __half)
  if (g == 0 && p < n_mlp) grads[p] = (half_t)((float)grads[p] + ((part[0][tx] + part[1][tx]) + (part[2][tx] + part[3][tx])));
}

// element type of the buffer the grid backward scatters into: the fp16 gradient blob's grid part, or (F = 1) its float image
template <int F> struct GridGrad { typedef half_t type; };
template <> struct GridGrad<1> { typedef float type; };

// F = 1: float image [lo, hi) of the grid gradients -> the fp16 blob, one rounding per entry; what it consumed is cleared for the next scatter
__global__ void fold_grid_grads_f32_kernel(float* __restrict__ image, half_t* __restrict__ grid_grads, size_t lo, size_t hi)
{
  for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (size_t)gridDim.x * blockDim.x) {
    const float v = image[i];
    if (v == 0.0f) continue;
    image[i] = 0.0f;
    grid_grads[i] = (half_t)((float)grid_grads[i] + v);   // (adds: two forward_backward calls before one optimizer step accumulate, vnr_amd.h)
  }
}

// ------------------------------------------------------------------------------------------------ grid backward
// EXTERNAL tcnn kernel_grid_backward: for every (sample, level): grad[idx*F+f] += w * dL/dfeature[f], accumulated in HALF precision
// with packed atomics as tcnn does for F > 1 (grad_t = __half, atomicAdd(__half2)).  F = 1: tcnn's grad_t is float there (one feature per
// level has no pair to pack), so the scatter adds fp32 (global_atomic_add_f32) into a float image of the grid part, and fold_grid_grads_f32_kernel
// rounds each sum ONCE into the fp16 gradient blob the optimizer and the gradient exchange read (rounds 1-5 added packed fp16 with a zero
// in the other half: 6 % low at ~2 000 adds per entry).  `grid_grads` is then that float image.
// lane = (sample, x bit, feature pair).  Memory-side float atomics are priced per 64-byte REQUEST, whatever the request carries
// (MI355X_MICROARCH.md "Global float atomics"; measured here: one lane per sample and dword 0.846 ms, (sample, feature) lanes 0.425,
// with the x bit 0.242, packed fp16 pairs 0.242: DESIGN.md 4.3), so what matters is that the lanes of one entry pair are adjacent:
// the x-neighbour of a corner is the adjacent table entry on dense levels and, for even x, on hashed levels ((x+1)^h = (x^h)^1).
// Every lane repeats the (cheap) index arithmetic of its sample.
template <int F>
__global__ void grid_backward_kernel(const GridDevice grid, const float* __restrict__ coords, const half_t* __restrict__ dfeat,
                                     uint32_t n, uint32_t in_width, typename GridGrad<F>::type* __restrict__ grid_grads, uint32_t level0)
{
  constexpr uint32_t P = F >= 2 ? (uint32_t)F / 2u : 1u;   // packed pairs per entry
  constexpr uint32_t kLanesPerSample = 2u * P;
  const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
  const uint32_t i = t / kLanesPerSample, r = t % kLanesPerSample;
  const uint32_t xb = r / P, f = 2u * (r % P);
  if (i >= n) return;
  const uint32_t level = level0 + blockIdx.y;
  const LevelInfo lv = grid.levels[level];
  float g0, g1 = 0.0f;
  if constexpr (F >= 2) {
    const half2_t g2 = *(const half2_t*)(dfeat + (size_t)i * in_width + level * F + f);
    g0 = (float)g2[0]; g1 = (float)g2[1];
  } else {
    g0 = (float)dfeat[(size_t)i * in_width + level];
  }
  if (g0 == 0.0f && g1 == 0.0f) return;
  const bool nearest = grid.interpolation == 2u;   // EXTERNAL tcnn kernel_grid_backward, Nearest: the whole gradient to the lower corner's entry
  if (nearest && xb) return;
  const CornerSetup c = level_setup(lv, grid.interpolation == 1u ? 1u : 0u, coords[3 * (size_t)i], coords[3 * (size_t)i + 1], coords[3 * (size_t)i + 2]);
  auto* base = grid_grads + (size_t)lv.offset * F + (F >= 2 ? f : 0u);
#pragma unroll
  for (int yz = 0; yz < 4; ++yz) {
    if (nearest && yz) break;
    const int corner = (int)xb | (yz << 1);
    const uint32_t idx = level_index(lv, c.g[0] + xb, c.g[1] + (uint32_t)(yz & 1), c.g[2] + (uint32_t)(yz >> 1));
    const float w = nearest ? 1.0f : corner_weight(c, corner);
    if constexpr (F >= 2) {
      const half2_t v = half2_t{(half_t)(w * g0), (half_t)(w * g1)};
      half_t* addr = base + (size_t)idx * F;
      asm volatile("global_atomic_pk_add_f16 %0, %1, off" : : "v"(addr), "v"(v) : "memory");
    } else {
      const float v = w * g0;
      float* addr = base + idx;
      asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(addr), "v"(v) : "memory");
    }
  }
}

// The same scatter as a PERSISTENT grid of a few blocks per CU that walks (level, lane) work items: memory-side atomics are issued without
// waiting for them, so a handful of waves per CU already saturates the chip's atomic request rate, and the rest of the CU stays free for
// the kernels that run beside it on the side stream (weight gradients, the dense levels' LDS scatter: forward_backward).  The plain
// kernel's grid of one block per 256 lanes takes every wave slot of the GPU while it lasts, which is why running those kernels beside it
// bought nothing in round 4.  Per lane the arithmetic is grid_backward_kernel's, statement for statement.
template <int F>
__global__ void __launch_bounds__(256) grid_backward_persistent_kernel(const GridDevice grid, const float* __restrict__ coords, const half_t* __restrict__ dfeat,
                                                                       uint32_t n, uint32_t in_width, typename GridGrad<F>::type* __restrict__ grid_grads, uint32_t level0, uint32_t n_levels)
{
  constexpr uint32_t P = F >= 2 ? (uint32_t)F / 2u : 1u;
  constexpr uint32_t kLanesPerSample = 2u * P;
  const uint32_t per_level = (n * kLanesPerSample + 255u) & ~255u;          // lanes of one level, whole blocks: the level is uniform over a block's trip
  const uint64_t total = (uint64_t)per_level * n_levels;
  const bool nearest = grid.interpolation == 2u;
  for (uint64_t w = (uint64_t)blockIdx.x * 256u + threadIdx.x; w < total; w += (uint64_t)gridDim.x * 256u) {
    const uint32_t level = level0 + (uint32_t)(w / per_level);
    const uint32_t t = (uint32_t)(w % per_level);
    const uint32_t i = t / kLanesPerSample, r = t % kLanesPerSample;
    const uint32_t xb = r / P, f = 2u * (r % P);
    if (i >= n) continue;
    const LevelInfo lv = grid.levels[level];
    float g0, g1 = 0.0f;
    if constexpr (F >= 2) {
      const half2_t g2 = *(const half2_t*)(dfeat + (size_t)i * in_width + level * F + f);
      g0 = (float)g2[0]; g1 = (float)g2[1];
    } else {
      g0 = (float)dfeat[(size_t)i * in_width + level];
    }
    if (g0 == 0.0f && g1 == 0.0f) continue;
    if (nearest && xb) continue;
    const CornerSetup c = level_setup(lv, grid.interpolation == 1u ? 1u : 0u, coords[3 * (size_t)i], coords[3 * (size_t)i + 1], coords[3 * (size_t)i + 2]);
    auto* base = grid_grads + (size_t)lv.offset * F + (F >= 2 ? f : 0u);
#pragma unroll
    for (int yz = 0; yz < 4; ++yz) {
      if (nearest && yz) break;
      const int corner = (int)xb | (yz << 1);
      const uint32_t idx = level_index(lv, c.g[0] + xb, c.g[1] + (uint32_t)(yz & 1), c.g[2] + (uint32_t)(yz >> 1));
      const float wgt = nearest ? 1.0f : corner_weight(c, corner);
      if constexpr (F >= 2) {
        const half2_t v = half2_t{(half_t)(wgt * g0), (half_t)(wgt * g1)};
        half_t* addr = base + (size_t)idx * F;
        asm volatile("global_atomic_pk_add_f16 %0, %1, off" : : "v"(addr), "v"(v) : "memory");
      } else {
        const float v = wgt * g0;
        float* addr = base + idx;
        asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(addr), "v"(v) : "memory");
      }
    }
  }
}

// The dense coarse levels through LDS.  A level of a few thousand entries takes 65 536 x 8 corner updates per step; as global atomics
// those are 262 144 memory-side requests per level whatever the level's size (one 64-byte request per x-pair segment, the floor the
// scatter above sits on).  Here a block owns a TILE of a level's table (a contiguous entry range that fits its LDS as fp32 accumulators)
// and a slice of the batch: it repeats the index arithmetic of every sample of its slice, adds the corners that fall into its tile
// with LDS atomics (fp32: one rounding to the gradient's half precision at the end instead of one per update) and flushes the tile
// with packed fp16 atomics on CONTIGUOUS entries, 16 entries per 64-byte request, skipping pairs nobody touched.  Requests per level:
// slices x entries / 16 instead of 4 x batch.  Levels whose table needs more than VNR_AMD_GRID_BWD_LDS_TILES tiles, and hashed levels,
// keep the global-atomic kernel.
struct LdsBwdItem { uint32_t level, e0, e1, s0, s1; };   // entries [e0, e1) of `level`, samples [s0, s1)

template <int F>
__global__ void __launch_bounds__(256) grid_backward_lds_kernel(const GridDevice grid, const LdsBwdItem* __restrict__ items, const float* __restrict__ coords,
                                                                const half_t* __restrict__ dfeat, uint32_t in_width, typename GridGrad<F>::type* __restrict__ grid_grads)
{
  extern __shared__ float s_acc[];   // [e1 - e0][F]
  const LdsBwdItem it = items[blockIdx.x];
  const LevelInfo lv = grid.levels[it.level];
  const uint32_t n_acc = (it.e1 - it.e0) * (uint32_t)F;
  for (uint32_t e = threadIdx.x; e < n_acc; e += blockDim.x) s_acc[e] = 0.0f;
  __syncthreads();
  const bool nearest = grid.interpolation == 2u;
  // A tile is a contiguous entry range of a dense level, i.e. a slab of grid z; a sample's eight corners have z in {gz, gz + 1}, so all of
  // them lie in [gz res^2, (gz + 2) res^2 + res].  A sample whose range misses the tile is dropped after ONE fma / floor instead of after
  // eight index computations: with 36 tiles at the fifth level a sample is looked at by 36 blocks and belongs to two of them (round 4:
  // the kernel 66 -> see DESIGN 4.3).  Ranges that reach the end of the table (indices wrap there) are never dropped.
  const uint32_t res2 = lv.resolution * lv.resolution;
  for (uint32_t i = it.s0 + threadIdx.x; i < it.s1; i += blockDim.x) {
    const float cx = coords[3 * (size_t)i], cy = coords[3 * (size_t)i + 1], cz = coords[3 * (size_t)i + 2];
    {
      const uint32_t gx = (uint32_t)(int32_t)__builtin_floorf(__builtin_fmaf(cx, lv.scale, 0.5f));
      const uint32_t gy = (uint32_t)(int32_t)__builtin_floorf(__builtin_fmaf(cy, lv.scale, 0.5f));
      const uint32_t gz = (uint32_t)(int32_t)__builtin_floorf(__builtin_fmaf(cz, lv.scale, 0.5f));
      const uint64_t lo = (uint64_t)gz * res2, hi = ((uint64_t)gz + 2u) * res2 + lv.resolution;   // [lo, hi]: every corner index before wrapping
      const bool in_domain = gx < lv.resolution && gy < lv.resolution && gz < lv.resolution;      // (coordinates outside [0, 1]: any index, never dropped)
      if (in_domain && hi < lv.size && (hi < it.e0 || lo >= it.e1)) continue;
    }
    float g[F];
    bool any = false;
#pragma unroll
    for (int f = 0; f < F; ++f) { g[f] = (float)dfeat[(size_t)i * in_width + it.level * F + f]; any = any || g[f] != 0.0f; }
    if (!any) continue;
    const CornerSetup c = level_setup(lv, grid.interpolation == 1u ? 1u : 0u, cx, cy, cz);
#pragma unroll
    for (int corner = 0; corner < 8; ++corner) {
      if (nearest && corner) break;
      const uint32_t idx = level_index(lv, c.g[0] + (corner & 1), c.g[1] + ((corner >> 1) & 1), c.g[2] + ((corner >> 2) & 1));
      if (idx < it.e0 || idx >= it.e1) continue;
      const float w = nearest ? 1.0f : corner_weight(c, corner);
#pragma unroll
      for (int f = 0; f < F; ++f) atomicAdd(&s_acc[(idx - it.e0) * F + f], w * g[f]);
    }
  }
  __syncthreads();
  if constexpr (F == 1) {   // fp32 sums into the float image, entry by entry
    float* base = grid_grads + (size_t)lv.offset + it.e0;
    for (uint32_t q = threadIdx.x; q < n_acc; q += blockDim.x) {
      const float a = s_acc[q];
      if (a == 0.0f) continue;
      float* addr = base + q;
      asm volatile("global_atomic_add_f32 %0, %1, off" : : "v"(addr), "v"(a) : "memory");
    }
    return;
  }
  // flush: one packed atomic per pair of halves that received something (e0 is even and level offsets are multiples of 8 entries)
  half_t* base = (half_t*)grid_grads + ((size_t)lv.offset + it.e0) * F;
  for (uint32_t q = threadIdx.x; 2u * q < n_acc; q += blockDim.x) {
    const float a = s_acc[2u * q], b = 2u * q + 1u < n_acc ? s_acc[2u * q + 1u] : 0.0f;
    if (a == 0.0f && b == 0.0f) continue;
    const half2_t v = half2_t{(half_t)a, (half_t)b};
    half_t* addr = base + 2u * (size_t)q;
    asm volatile("global_atomic_pk_add_f16 %0, %1, off" : : "v"(addr), "v"(v) : "memory");
  }
}

// ------------------------------------------------------------------------------------------------ Adam
// EXTERNAL tcnn adam_step (optimizers/adam.h): see header comment.  Parameters [lo, hi); also clears the gradient for the next step.
// Measured split at C4 (70 M parameters, tools/adam_probe.py, with the fp32 gradient blob of round 1): the sweep over all gradients
// alone 0.15 ms, the ~10 M touched parameters of a 65 536-sample batch 0.44 ms with master / m / v / step in four arrays.  Hence one
// 16-byte record per parameter (OptState), and no zero written over a gradient that is already zero.
__global__ void adam_kernel(size_t lo, size_t hi, size_t n_matrix, float grad_mul, float lr, float beta1, float beta2, float log2_beta1,
                            float log2_beta2, float epsilon, float l2_reg, OptState* __restrict__ state, half_t* __restrict__ params,
                            half_t* __restrict__ grads)
{
  const size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= hi) return;
  const half_t raw_h = grads[i];
  const float raw = (float)raw_h;
  if (raw != 0.0f) grads[i] = (half_t)0.0f;
  float gradient = raw * grad_mul;
  if (i >= n_matrix && gradient == 0.0f) return;  // untouched hash-grid entries are skipped entirely
  OptState st = state[i];
  const float w = st.master;
  if (i < n_matrix) gradient += l2_reg * w;       // no L2 regularisation for grid parameters
  const float m = st.m = beta1 * st.m + (1.0f - beta1) * gradient;
  const float v = st.v = beta2 * st.v + (1.0f - beta2) * (gradient * gradient);
  const uint32_t step = ++st.step;
  // beta^step as exp2(step * log2(beta)) on the native exponential (1 instruction instead of the ~80 of powf): almost every
  // wave reaches this line for a few touched lanes, so the length of this path, not memory, sets the kernel's time
  const float fs = (float)step;
  const float lr_t = lr * sqrtf(1.0f - __builtin_amdgcn_exp2f(fs * log2_beta2)) / (1.0f - __builtin_amdgcn_exp2f(fs * log2_beta1));
  const float eff = lr_t / (sqrtf(v) + epsilon);
  const float nw = w - eff * m;
  st.master = nw;
  state[i] = st;
  params[i] = (half_t)nw;
}

// The same step with the sweep and the update taken apart.  adam_kernel gives every parameter a thread: 1.1 M waves, each of which loads
// 128 bytes of gradients and then walks the ~50 instructions of the update for the few lanes whose gradient is not zero (a 65 536-sample
// batch touches ~12 % of the 70 M parameters): the kernel is bound by instruction issue, 0.32 ms (profiles/r02_train_step_timeline.txt).
// Here a lane sweeps EIGHT gradients (one 16-byte load, one 16-byte store of zeros where any was set), the wave compacts what it found
// (offset in the wave's 512 parameters and the gradient's bits, through 2 KB of LDS) and then runs the update on dense lanes: the update
// path is executed once per 64 TOUCHED parameters instead of once per 64 parameters.  Per parameter the arithmetic is adam_kernel's,
// statement for statement (bit-identical state and parameters: tests/test_gpu_train.py).
__global__ void __launch_bounds__(256) adam_compact_kernel(size_t lo, size_t hi, size_t n_matrix, float grad_mul, float lr, float beta1, float beta2,
                                                           float log2_beta1, float log2_beta2, float epsilon, float l2_reg,
                                                           OptState* __restrict__ state, half_t* __restrict__ params, half_t* __restrict__ grads)
{
  __shared__ uint32_t s_list[4][512];
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  const size_t a0 = lo & ~(size_t)7;                                   // groups of 8 parameters, aligned to 16 bytes of the gradient blob
  const size_t wave_base = a0 + ((size_t)blockIdx.x * 256u + wave * 64u) * 8u;
  const size_t base = wave_base + (size_t)lane * 8u;
  uint32_t h[4] = {0u, 0u, 0u, 0u};   // the group's 8 gradients (bits)
  uint32_t mask = 0;                   // parameters of the group to look at: in [lo, hi) and (gradient set, or a matrix weight)
  if (base < hi) {
    const bool whole = base >= lo && base + 8u <= hi;
    if (whole) {
      const uint4 v = *reinterpret_cast<const uint4*>(grads + base);
      h[0] = v.x; h[1] = v.y; h[2] = v.z; h[3] = v.w;
    } else {
      const uint16_t* g16 = reinterpret_cast<const uint16_t*>(grads);
      for (uint32_t j = 0; j < 8u; ++j)
        if (base + j >= lo && base + j < hi) h[j >> 1] |= (uint32_t)g16[base + j] << (16u * (j & 1u));
    }
    uint32_t set = 0;
#pragma unroll
    for (uint32_t j = 0; j < 8u; ++j) {
      const uint32_t bits = (h[j >> 1] >> (16u * (j & 1u))) & 0xffffu;
      const bool in = whole || (base + j >= lo && base + j < hi);
      const bool nz = (bits & 0x7fffu) != 0u;                          // raw != 0.0f (a negative zero is a zero)
      if (in && nz) set |= 1u << j;
      if (in && (nz || base + j < n_matrix)) mask |= 1u << j;
    }
    if (set) {   // clear what was set, for the next step
      if (whole) *reinterpret_cast<uint4*>(grads + base) = uint4{0u, 0u, 0u, 0u};
      else {
        uint16_t* g16 = reinterpret_cast<uint16_t*>(grads);
        for (uint32_t j = 0; j < 8u; ++j) if (set & (1u << j)) g16[base + j] = 0;
      }
    }
  }
  // wave compaction
  const uint32_t cnt = (uint32_t)__popc(mask);
  uint32_t incl = cnt;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if ((int)lane >= d) incl += y;
  }
  const uint32_t total = __shfl(incl, 63);
  if (total == 0) return;   // wave-uniform
  uint32_t pos = incl - cnt;
  uint32_t* list = s_list[wave];
#pragma unroll
  for (uint32_t j = 0; j < 8u; ++j)
    if (mask & (1u << j)) list[pos++] = ((lane * 8u + j) << 16) | ((h[j >> 1] >> (16u * (j & 1u))) & 0xffffu);
  __builtin_amdgcn_wave_barrier();
  for (uint32_t t = lane; t < total; t += 64u) {
    const uint32_t e = list[t];
    const size_t i = wave_base + (e >> 16);
    const uint16_t hb = (uint16_t)(e & 0xffffu);
    half_t raw_h;
    __builtin_memcpy(&raw_h, &hb, 2);
    const float raw = (float)raw_h;
    float gradient = raw * grad_mul;
    if (i >= n_matrix && gradient == 0.0f) continue;  // untouched hash-grid entries are skipped entirely
    OptState st = state[i];
    const float w = st.master;
    if (i < n_matrix) gradient += l2_reg * w;       // no L2 regularisation for grid parameters
    const float m = st.m = beta1 * st.m + (1.0f - beta1) * gradient;
    const float v = st.v = beta2 * st.v + (1.0f - beta2) * (gradient * gradient);
    const uint32_t step = ++st.step;
    const float fs = (float)step;
    const float lr_t = lr * sqrtf(1.0f - __builtin_amdgcn_exp2f(fs * log2_beta2)) / (1.0f - __builtin_amdgcn_exp2f(fs * log2_beta1));
    const float eff = lr_t / (sqrtf(v) + epsilon);
    const float nw = w - eff * m;
    st.master = nw;
    state[i] = st;
    params[i] = (half_t)nw;
  }
}

static void launch_adam(size_t lo, size_t hi, size_t n_matrix, float grad_mul, float lr, float beta1, float beta2, float epsilon, float l2_reg,
                        OptState* state, half_t* params, half_t* grads, hipStream_t s)
{
  const char* e = std::getenv("VNR_AMD_ADAM_COMPACT");   // (read per step: the two kernels are compared inside one process, tests/test_gpu_train.py)
  const bool compact = !e || std::atoi(e) != 0;
  const float l2b1 = (float)std::log2((double)beta1), l2b2 = (float)std::log2((double)beta2);
  if (compact) {
    const size_t groups = (hi - (lo & ~(size_t)7) + 7) / 8;
    adam_compact_kernel<<<div_round_up(groups, 256), 256, 0, s>>>(lo, hi, n_matrix, grad_mul, lr, beta1, beta2, l2b1, l2b2, epsilon, l2_reg, state, params, grads);
  } else {
    adam_kernel<<<div_round_up(hi - lo, 256), 256, 0, s>>>(lo, hi, n_matrix, grad_mul, lr, beta1, beta2, l2b1, l2b2, epsilon, l2_reg, state, params, grads);
  }
  VNR_HIP_CHECK(hipGetLastError());
}

// fp16 gradient blob -> a float copy (vnrAmdNeuralVolumeGradients: inspection and tests)
__global__ void unpack_grads_f16_kernel(const half_t* __restrict__ in, float* __restrict__ out, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (float)in[i];
}

// ------------------------------------------------------------------------------------------------ host
struct TrainScratch {  // per-Network extra buffers that do not need to live in the class interface
  DeviceBuffer<float> y{MemTag::Network};
  DeviceBuffer<uint16_t> dy{MemTag::Network};
  DeviceBuffer<uint16_t> d_all{MemTag::Network};
  DeviceBuffer<float> wgrad_slab{MemTag::Network};   // [blocks][n_mlp] partial weight gradients
  uint64_t slab_key = 0;                             // (n_mlp, n_neurons, in_width, hidden matmuls) the slab's never-written elements were zeroed for
  DeviceBuffer<uint8_t> lds_items{MemTag::Network};  // work items of grid_backward_lds_kernel (as bytes: the item type is local to this file)
  std::vector<uint8_t> lds_items_host;               // what lds_items holds: the list depends on the level sizes, n_features, tile size, batch and level range
  uint32_t loss_blocks = 0;
  DeviceBuffer<double> distance_partials{MemTag::Network};   // gradient_distance (diagnostics)
  DeviceBuffer<float> grid_grads_f32{MemTag::Network};       // n_features = 1: float image of the grid gradients (zero between steps)
  hipEvent_t fold_ev = nullptr;                              // ... and the join of its two scatter streams in front of the fold
  ~TrainScratch() { if (fold_ev) (void)hipEventDestroy(fold_ev); }
};

}  // namespace vnr

#include <map>
#include <memory>
#include <mutex>
#include <set>

namespace vnr {

// once per kernel and process: hipFuncSetAttribute stays off the step's launch path.  Keyed by the function's address (the instances of a
// kernel template share one signature, so a static flag inside a generic lambda would be shared by all of them); guarded, because two
// host threads may train two networks; the attribute belongs to the function, not to a device.
static bool first_use_of_kernel(const void* kernel)
{
  static std::mutex m;
  static std::set<const void*> done;
  std::lock_guard<std::mutex> g(m);
  return done.insert(kernel).second;
}

static std::map<const Network*, std::unique_ptr<TrainScratch>>& scratch_map()
{
  static std::map<const Network*, std::unique_ptr<TrainScratch>> m;
  return m;
}
static TrainScratch& scratch_of(const Network* n)
{
  auto& m = scratch_map();
  auto it = m.find(n);
  if (it == m.end()) it = m.emplace(n, std::make_unique<TrainScratch>()).first;
  return *it->second;
}
void network_release_scratch(const Network* n) { scratch_map().erase(n); }

void Network::ensure_training_state(hipStream_t s)
{
  if (opt_state_.count != n_params_) { opt_state_.resize(n_params_); launch_master_from_f16(params_f16_.ptr, opt_state_.ptr, n_params_, true, s); }
  // (an even number of halves: the packed fp16 atomics of the grid backward add PAIRS, and a Tiled grid of F = 1 can end on an odd element)
  if (grads_.count != grads_alloc()) { grads_.resize(grads_alloc()); grads_.zero(s); }
}

void Network::reset_master_from_params(hipStream_t s)
{
  if (opt_state_.count == n_params_) launch_master_from_f16(params_f16_.ptr, opt_state_.ptr, n_params_, false, s);
}

// 5. of the training step: dL/dfeatures (ws_dfeat_, written by the MLP backward) -> the grid part of the gradient blob.  A function of its
// own so that it can be repeated alone on the stored dL/dfeatures (vnrAmdNeuralVolumeRescatterGridGradients: diagnostics).
// which levels of the grid backward go through LDS tiles, and what the scatter costs in memory-side atomic requests (network.h)
GridBackwardPlan Network::grid_backward_plan(size_t batch) const
{
  static const bool lds_bwd = [] { const char* e = std::getenv("VNR_AMD_GRID_BWD_LDS"); return !e || std::atoi(e) != 0; }();
  static const uint32_t lds_kb = [] { const char* e = std::getenv("VNR_AMD_GRID_BWD_LDS_KB"); return e ? (uint32_t)std::max(8, std::min(144, std::atoi(e))) : 24u; }();
  static const uint32_t lds_blocks = [] { const char* e = std::getenv("VNR_AMD_GRID_BWD_LDS_BLOCKS"); return e ? (uint32_t)std::max(1, std::atoi(e)) : 768u; }();
  static const uint32_t lds_max_tiles = [] { const char* e = std::getenv("VNR_AMD_GRID_BWD_LDS_TILES"); return e ? (uint32_t)std::max(1, std::atoi(e)) : 64u; }();
  // (sweep of the three on the C4 model, profiles/r03_grid_backward_lds_sweep.txt: 24 KB tiles, ~768 blocks per level, levels of at most 64 tiles =
  // levels 0 - 4 of C4: grid backward 0.239 -> 0.18 - 0.22 ms, bimodal from run to run; larger tiles or more levels cost more in scanning
  // than their atomics saved)
  GridBackwardPlan p{};
  const uint32_t F = cfg_.n_features;
  p.tile_entries = (lds_kb * 1024u / (4u * F)) & ~15u;
  p.lds_blocks = lds_blocks;
  p.n_levels = n_active_levels();
  if (lds_bwd)
    while (p.lds_levels < p.n_levels && !grid_.levels[p.lds_levels].hashed && ((size_t)grid_.levels[p.lds_levels].offset * F) % 2 == 0 &&
           div_round_up(grid_.levels[p.lds_levels].size, p.tile_entries) <= lds_max_tiles) ++p.lds_levels;   // (the flush adds aligned pairs of halves)
  // memory-side requests of one step (MI355X_MICROARCH.md "Global float atomics": a wave instruction leaves L2 as 64-byte requests).  The atomic
  // kernel's lanes are (sample, x bit, feature pair) with the two x-neighbours of a corner pair adjacent: ONE request per (sample, level, yz corner)
  // while an entry pair fits 64 bytes; an LDS tile flushes at most (entries x F x element bytes) / 64 requests per slice of the batch.
  const uint32_t elem = F == 1 ? 4u : 2u;   // F = 1 scatters fp32 (network_train.hip GridGrad)
  p.atomic_requests = (uint64_t)batch * 4u * (p.n_levels - p.lds_levels) * std::max(1u, (2u * F * elem + 63u) / 64u);
  for (uint32_t l = 0; l < p.lds_levels; ++l) {
    const uint32_t tiles = div_round_up(grid_.levels[l].size, p.tile_entries);
    const uint32_t slices = std::max(4u, std::min(128u, p.lds_blocks / tiles));
    p.flush_requests_at_most += (uint64_t)slices * ((uint64_t)grid_.levels[l].size * F * elem / 64u);
  }
  return p;
}

void Network::scatter_grid_gradients(const float* d_coords, size_t batch, hipStream_t s, GradExchange* exchange, hipStream_t s_lds)
{
  // s_lds: the stream of the dense levels' LDS scatter; another stream than `s` means the two scatters run side by side (forward_backward
  // forks and joins), and the atomic scatter then takes its persistent form so that it leaves room on the CUs
  if (!s_lds) s_lds = s;
  const bool side_by_side = s_lds != s;
  TrainScratch& ts = scratch_of(this);
  const uint32_t n = (uint32_t)batch;
  // 5. hash-grid backward: levels [l0, l1) per launch (blockIdx.y + l0 = level)
  // levels [0, lds_levels) go through grid_backward_lds_kernel: dense, and at most kLdsBwdMaxTiles LDS tiles (VNR_AMD_GRID_BWD_LDS=0: none)
  const GridBackwardPlan plan = grid_backward_plan(batch);
  const uint32_t tile_entries = plan.tile_entries, lds_levels = plan.lds_levels, lds_blocks = plan.lds_blocks;
  // n_features = 1: the scatter's target is the float image (fold_grid_grads_f32_kernel); the fold of a level range follows its scatter on the
  // same stream(s), before anybody (optimizer, exchange) reads the blob
  float* gg32 = nullptr;
  if (cfg_.n_features == 1) {
    const size_t n_grid = grads_alloc() - n_mlp_;
    if (ts.grid_grads_f32.count != n_grid) { ts.grid_grads_f32.resize(n_grid); ts.grid_grads_f32.zero(s); VNR_HIP_CHECK(hipStreamSynchronize(s)); }
    gg32 = ts.grid_grads_f32.ptr;
  }
  auto fold = [&](uint32_t l0, uint32_t l1) {
    if (!gg32 || l0 >= l1) return;
    const size_t lo = level_range_lo(l0) - n_mlp_, hi = level_range_hi(l1) - n_mlp_;
    fold_grid_grads_f32_kernel<<<(uint32_t)std::min<size_t>(div_round_up(hi - lo, 256), (size_t)Runtime::get().n_cus * 16), 256, 0, s>>>(gg32, (half_t*)grads_.ptr + n_mlp_, lo, hi);
  };
  auto grid_backward_lds = [&](uint32_t l0, uint32_t l1) {
    // work items: every tile of every level x slices of the batch; more slices where a level has few tiles, so that ~2 blocks per CU exist
    std::vector<LdsBwdItem> items;
    for (uint32_t l = l0; l < l1; ++l) {
      const uint32_t size = grid_.levels[l].size, tiles = div_round_up(size, tile_entries);
      const uint32_t slices = std::max(4u, std::min(128u, lds_blocks / tiles));
      const uint32_t per = (uint32_t)div_round_up(batch, slices);
      for (uint32_t t = 0; t < tiles; ++t)
        for (uint32_t sl = 0; sl < slices; ++sl) {
          const uint32_t s0 = sl * per, s1 = std::min<uint32_t>(n, s0 + per);
          if (s0 < s1) items.push_back({l, t * tile_entries, std::min(size, (t + 1) * tile_entries), s0, s1});
        }
    }
    if (items.empty()) return;
    // (cached on the device while the list is the same: it depends on the level sizes, n_features, the tile size, the batch and the level range,
    // so the bytes themselves are the key: a re-configured model must not reuse the old model's tile and slice ranges)
    const size_t item_bytes = items.size() * sizeof(LdsBwdItem);
    if (ts.lds_items_host.size() != item_bytes || std::memcmp(ts.lds_items_host.data(), items.data(), item_bytes) != 0) {
      ts.lds_items.ensure(item_bytes);
      VNR_HIP_CHECK(hipMemcpyAsync(ts.lds_items.ptr, items.data(), item_bytes, hipMemcpyHostToDevice, s_lds));
      VNR_HIP_CHECK(hipStreamSynchronize(s_lds));   // pageable source
      ts.lds_items_host.assign((const uint8_t*)items.data(), (const uint8_t*)items.data() + item_bytes);
    }
    const size_t shmem = (size_t)tile_entries * cfg_.n_features * sizeof(float);
    half_t* gg = (half_t*)grads_.ptr + n_mlp_;
    auto launch = [&](auto kernel, auto* out) {
      if (first_use_of_kernel((const void*)kernel)) VNR_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
      kernel<<<(uint32_t)items.size(), 256, shmem, s_lds>>>(grid_, (const LdsBwdItem*)ts.lds_items.ptr, d_coords, (const half_t*)ws_dfeat_.ptr, in_width_, out);
    };
    switch (cfg_.n_features) {
    case 1: launch(grid_backward_lds_kernel<1>, gg32); break;
    case 2: launch(grid_backward_lds_kernel<2>, gg); break;
    case 4: launch(grid_backward_lds_kernel<4>, gg); break;
    default: launch(grid_backward_lds_kernel<8>, gg); break;
    }
  };
  auto grid_backward = [&](uint32_t l0, uint32_t l1) {
    if (l0 < lds_levels) {
      grid_backward_lds(l0, std::min(l1, lds_levels));
      l0 = std::min(l1, lds_levels);
      if (l0 >= l1) return;
    }
    const uint32_t pairs = cfg_.n_features >= 2 ? cfg_.n_features / 2 : 1u;
    const dim3 g(div_round_up((uint64_t)batch * pairs * 2, 256), l1 - l0);  // one lane per (sample, x bit, feature pair)
    half_t* gg = (half_t*)grads_.ptr + n_mlp_;
    if (side_by_side) {   // a few blocks per CU walk the same lanes (VNR_AMD_GRID_BWD_BLOCKS_PER_CU, default 2: swept 2 / 3 / 4 / 6 / 8, profiles/r05_train_overlap_ab.txt)
      static const uint32_t per_cu = [] { const char* e = std::getenv("VNR_AMD_GRID_BWD_BLOCKS_PER_CU"); return e ? (uint32_t)std::max(1, std::min(16, std::atoi(e))) : 2u; }();
      const uint32_t blocks = std::min<uint32_t>(g.x * g.y, (uint32_t)Runtime::get().n_cus * per_cu);
      switch (cfg_.n_features) {
      case 1: grid_backward_persistent_kernel<1><<<blocks, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg32, l0, l1 - l0); break;
      case 2: grid_backward_persistent_kernel<2><<<blocks, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg, l0, l1 - l0); break;
      case 4: grid_backward_persistent_kernel<4><<<blocks, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg, l0, l1 - l0); break;
      default: grid_backward_persistent_kernel<8><<<blocks, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg, l0, l1 - l0); break;
      }
      return;
    }
    switch (cfg_.n_features) {
    case 1: grid_backward_kernel<1><<<g, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg32, l0); break;
    case 2: grid_backward_kernel<2><<<g, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg, l0); break;
    case 4: grid_backward_kernel<4><<<g, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg, l0); break;
    default: grid_backward_kernel<8><<<g, 256, 0, s>>>(grid_, d_coords, (const half_t*)ws_dfeat_.ptr, n, in_width_, gg, l0); break;
    }
  };
  if (!exchange) {
    // diagnostics builds (-DVNR_DIAG; tools/train_probe.py): VNR_AMD_GRID_BWD_LEVELS="l0,l1" scatters levels [l0, l1) only, to price a level
    // (a wrong gradient by design: not in the library that ships)
#if defined(VNR_DIAG)
    static const std::pair<int, int> only = [] {
      const char* e = std::getenv("VNR_AMD_GRID_BWD_LEVELS");
      int a = -1, b = -1;
      if (e && std::sscanf(e, "%d,%d", &a, &b) == 2 && a >= 0 && b > a) return std::make_pair(a, b);
      return std::make_pair(-1, -1);
    }();
#else
    constexpr std::pair<int, int> only{-1, -1};
#endif
    // levels at or beyond max_level + 1e-3 encode to zero and receive no gradient (EXTERNAL tcnn kernel_grid_backward has the same test)
    if (only.first >= 0) grid_backward((uint32_t)only.first, std::min<uint32_t>((uint32_t)only.second, n_active_levels()));
    else if (n_active_levels() > 0) grid_backward(0, n_active_levels());
    if (gg32 && n_active_levels() > 0) {
      if (side_by_side) {   // the dense levels' LDS scatter runs on s_lds and is joined by the caller only later: the fold waits for it here
        if (!ts.fold_ev) VNR_HIP_CHECK(hipEventCreateWithFlags(&ts.fold_ev, hipEventDisableTiming));
        VNR_HIP_CHECK(hipEventRecord(ts.fold_ev, s_lds));
        VNR_HIP_CHECK(hipStreamWaitEvent(s, ts.fold_ev, 0));
      }
      fold(0, n_active_levels());
    }
  } else {
    // finest levels first (the large tables), in buckets of at least bucket_params() parameters: a bucket's exchange overlaps the
    // backward launches of the coarser levels and, afterwards, the optimizer update of the buckets before it
    for (const auto& b : exchange_level_buckets(exchange->bucket_params())) {
      if (b.first < n_active_levels()) { grid_backward(b.first, std::min(b.second, n_active_levels())); fold(b.first, std::min(b.second, n_active_levels())); }
      exchange->range_ready(level_range_lo(b.first), level_range_hi(b.second), s);
    }
  }
}

void Network::forward_backward(const float* d_coords, const float* d_targets, size_t batch, hipStream_t s, GradExchange* exchange)
{
  if (!valid()) throw std::runtime_error("network is not configured");
  if (batch == 0) return;
  // every model the reference's dispatch builds runs on the MFMA kernels (tcnn_network.h:163-252 builds whatever the model JSON asks for)
  TrainScratch& ts = scratch_of(this);
  const uint32_t nh = n_hidden_matmuls();
  const uint32_t n = (uint32_t)batch;
  const uint32_t Wn = cfg_.n_neurons;
  ensure_training_state(s);  // lazily allocated
  if (ws_batch_ != batch) {
    ws_features_.resize(batch * in_width_);
    ws_acts_.resize((size_t)(nh + 1) * batch * Wn);
    ws_dfeat_.resize(batch * in_width_);
    ts.d_all.resize((size_t)(nh + 1) * batch * Wn);
    ts.y.resize(batch);
    ts.dy.resize(batch);
    ts.loss_blocks = std::min<uint32_t>(div_round_up(batch, 256), 1024u);
    ws_loss_.resize(ts.loss_blocks);
    ws_batch_ = batch;
  }

  // 1. forward, keeping features and hidden activations
  profile_mark(0, s);
  {
    GridDevice grid = grid_;
    grid.n_levels = n_active_levels();   // masked levels encode to zero, like the padding (and receive no gradient below)
    launch_fused(2, grid, in_width_, fused_mlp(), levels_dev_.ptr, params_f16_.ptr + n_mlp_, n_grid_params() * 2, d_coords, ts.y.ptr, ws_features_.ptr,
                 ws_acts_.ptr, batch, nullptr, batch, s);
  }
  // 2. loss + output gradient
  profile_mark(1, s);
  loss_grad_kernel<<<ts.loss_blocks, 256, 0, s>>>(ts.y.ptr, d_targets, n, cfg_.loss, cfg_.output_activation, (half_t*)ts.dy.ptr, ws_loss_.ptr);
  // the slab's elements no block ever writes (rows 1 .. 15 of the padded last layer) must be zero: they are summed into the gradient.  Zeroed
  // when the slab grows or when the layout it was zeroed for changes (width, input width, depth: 64 neurons x 1 layer and 32 x 2 share n_mlp)
  auto ensure_slab = [&](size_t rows, hipStream_t zs) {
    const uint64_t key = (uint64_t)n_mlp_ | ((uint64_t)Wn << 32) | ((uint64_t)in_width_ << 40) | ((uint64_t)nh << 48);   // the whole layout: two shapes can share n_mlp
    if (ts.wgrad_slab.count < rows * n_mlp_ || ts.slab_key != key) {
      if (ts.wgrad_slab.count < rows * n_mlp_) ts.wgrad_slab.resize(rows * n_mlp_);
      ts.wgrad_slab.zero(zs);
      ts.slab_key = key;
    }
  };
  bool overlap = false;
  hipStream_t sw = s;   // the stream of the weight gradients and of the dense levels' LDS scatter (a side stream when they overlap the atomic scatter)
  {
  // 3. MLP backward (the transposed weight image was packed with the forward one when the parameters last changed)
  BackwardArgs ba;
  ba.packedT = (const half_t*)mlp_packed_T_.ptr; ba.dy = (const half_t*)ts.dy.ptr; ba.acts = (const half_t*)ws_acts_.ptr;
  ba.d_out = (half_t*)ts.d_all.ptr; ba.dfeat = (half_t*)ws_dfeat_.ptr;
  ba.n = n; ba.nh = nh; ba.activation = cfg_.activation; ba.in_width = in_width_; ba.lds_halves = lds_halves_T_;
  const bool bwd_global = (size_t)lds_halves_T_ * 2 > kLdsBytes;   // a deep network (128 neurons: >= 6 hidden layers): the image stays in global memory (GENERAL instance)
  ba.weights_global = bwd_global ? 1u : 0u;
  {
    const size_t shmem = bwd_global ? 16 : (size_t)lds_halves_T_ * 2;
    const uint32_t fit = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, kLdsBytes / shmem));
    const uint32_t blocks = std::min<uint32_t>(div_round_up(div_round_up(batch, 64), 4), (uint32_t)Runtime::get().n_cus * fit);
    const int mt = (int)((in_width_ + 31) / 32);
    auto launch = [&](auto kernel) {
      // once per kernel: no driver call on the step's launch path.  (Keyed by the function's address: the instances share one signature,
      // so a static flag inside this generic lambda would be shared by all of them.)
      if (first_use_of_kernel((const void*)kernel)) VNR_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdsBytes));
      kernel<<<blocks, 256, shmem, s>>>(ba);
    };
    const bool gen = cfg_.activation > 1u || bwd_global;
#define VNR_BWD_G(w, g) do { if (mt == 1) launch(mlp_backward_kernel<w, 1, g>); else if (mt == 2) launch(mlp_backward_kernel<w, 2, g>); \
                             else if (mt == 3) launch(mlp_backward_kernel<w, 3, g>); else launch(mlp_backward_kernel<w, 4, g>); } while (0)
#define VNR_BWD(w) do { if (gen) VNR_BWD_G(w, true); else VNR_BWD_G(w, false); } while (0)
    switch (Wn) {
    case 16: VNR_BWD(16); break;
    case 32: VNR_BWD(32); break;
    case 64: VNR_BWD(64); break;
    default: VNR_BWD(128); break;
    }
#undef VNR_BWD
#undef VNR_BWD_G
  }
  // 4. weight gradients.  They read what the MLP backward wrote and write the MLP part of the gradient blob; the grid backward reads
  // dL/dfeatures and writes the grid part: independent.  Round 5: on a side stream (with the dense levels' LDS scatter behind them) BESIDE the
  // atomic scatter of the hashed levels, which is bound by the memory side's atomic rate and, in its persistent form, leaves the CUs to
  // them.  Measured (profiles/r05_train_overlap_ab.txt, one process, alternating, 8 pairs at two persistent blocks per CU): 2.4 % of the step
  // on average, never slower -- the 25 us of weight gradients disappear, the LDS scatter does not (the two scatters slow each other down by
  // about its length).  Not with a data-parallel exchange, whose ranges become ready in stream order.
  // OPT-IN (VNR_AMD_TRAIN_OVERLAP=1), not the default: the side stream is one more HIP stream alive in the process, and a renderer whose ray
  // parts then share a hardware queue with it pays far more than this buys (a 1/8 share of the frame 0.54 -> 0.85 ms before the renderer was
  // made robust to it, profiles/r05_stream_budget.txt); a process that only trains can switch it on.
  profile_mark(2, s);
  const char* overlap_e = std::getenv("VNR_AMD_TRAIN_OVERLAP");   // (read per step: both forms are compared inside one process, tests/test_gpu_train.py)
  overlap = overlap_e && std::atoi(overlap_e) != 0 && !exchange && batch >= 8192;   // opt-in: see below
  if (overlap) {
    if (!side_stream_) {
      VNR_HIP_CHECK(hipStreamCreateWithFlags(&side_stream_, hipStreamNonBlocking));
      VNR_HIP_CHECK(hipEventCreateWithFlags(&ev_fork_, hipEventDisableTiming));
      VNR_HIP_CHECK(hipEventCreateWithFlags(&ev_join_, hipEventDisableTiming));
    }
    sw = side_stream_;
    VNR_HIP_CHECK(hipEventRecord(ev_fork_, s));
    VNR_HIP_CHECK(hipStreamWaitEvent(sw, ev_fork_, 0));
  }
  WGradArgs wa;
  wa.features = (const half_t*)ws_features_.ptr; wa.acts = (const half_t*)ws_acts_.ptr; wa.d_all = (const half_t*)ts.d_all.ptr;
  wa.dy = (const half_t*)ts.dy.ptr; wa.n = n; wa.nh = nh; wa.in_width = in_width_;
  {
    // a block per 256 samples and matrix (padded_width is a multiple of 16 and at most 128)
    if (in_width_ % 8 != 0 || in_width_ > 128) throw std::runtime_error("internal: weight gradients of an input width the MFMA kernel does not cover");
    const uint32_t nblk = div_round_up(batch, (uint64_t)(kWgStage * kWgStages));
    ensure_slab(nblk, sw);
    wa.slab = ts.wgrad_slab.ptr; wa.n_mlp = (uint32_t)n_mlp_;
    const dim3 g1(nblk, 1);
    const dim3 g2(nblk, nh + 1);  // hidden layers 1..nh and the last layer nh+1
    const int nt1 = in_width_ <= 32 ? 1 : in_width_ <= 64 ? 2 : 4;
#define VNR_WG(w, mt_hidden) do {                                                                          \
      if (nt1 == 1) weight_grad_mfma_kernel<w, 1><<<g1, 256, 0, sw>>>(wa, 0);                              \
      else if (nt1 == 2) weight_grad_mfma_kernel<w, 2><<<g1, 256, 0, sw>>>(wa, 0);                         \
      else weight_grad_mfma_kernel<w, 4><<<g1, 256, 0, sw>>>(wa, 0);                                       \
      weight_grad_mfma_kernel<w, mt_hidden><<<g2, 256, 0, sw>>>(wa, 1); } while (0)
    switch (Wn) {
    case 16: VNR_WG(16, 1); break;
    case 32: VNR_WG(32, 1); break;
    case 64: VNR_WG(64, 2); break;
    default: VNR_WG(128, 4); break;
    }
#undef VNR_WG
    weight_grad_reduce_kernel<<<div_round_up(n_mlp_, 64), 256, 0, sw>>>(ts.wgrad_slab.ptr, nblk, (uint32_t)n_mlp_, (half_t*)grads_.ptr);
  }
  }   // MFMA kernels
  profile_mark(3, s);
  if (exchange) exchange->range_ready(0, n_mlp_, s);   // the MLP's gradient travels while the grid backward runs
  scatter_grid_gradients(d_coords, batch, s, exchange, sw);
  if (overlap) {   // join: the optimizer reads both parts of the blob
    VNR_HIP_CHECK(hipEventRecord(ev_join_, sw));
    VNR_HIP_CHECK(hipStreamWaitEvent(s, ev_join_, 0));
  }
  VNR_HIP_CHECK(hipGetLastError());
  profile_mark(4, s);
}

std::vector<std::pair<uint32_t, uint32_t>> Network::exchange_level_buckets(size_t bucket) const
{
  std::vector<std::pair<uint32_t, uint32_t>> out;
  uint32_t l1 = cfg_.n_levels;
  while (l1 > 0) {
    uint32_t l0 = l1;
    size_t count = 0;
    while (l0 > 0 && (count < bucket || (size_t)grid_.levels[l0 - 1].offset * cfg_.n_features < bucket)) { --l0; count += (size_t)grid_.levels[l0].size * cfg_.n_features; }
    out.emplace_back(l0, l1);
    l1 = l0;
  }
  return out;
}

void Network::for_each_exchange_range(size_t bucket, const std::function<void(size_t, size_t)>& fn) const
{
  fn(0, n_mlp_);
  for (const auto& b : exchange_level_buckets(bucket)) fn(level_range_lo(b.first), level_range_hi(b.second));
}

void Network::optimizer_step(float grad_scale, hipStream_t s)
{
  if (grads_.count != grads_alloc()) throw std::runtime_error("optimizer_step before forward_backward");
  if (opt_sharded_)
    throw std::runtime_error("the optimizer state of this volume is sharded over the ranks (vnrAmdNeuralVolumeTrainDataParallel): call "
                             "vnrAmdNeuralVolumeSyncReplicas on every rank before a step that updates all parameters on one rank");
  launch_adam(0, n_params_, n_mlp_, grad_scale / (float)kLossScale, lr_, cfg_.beta1, cfg_.beta2, cfg_.epsilon, cfg_.l2_reg, opt_state_.ptr,
              (half_t*)params_f16_.ptr, (half_t*)grads_.ptr, s);
  optimizer_finish_step(s);
}

void Network::optimizer_step_range(size_t lo, size_t hi, float grad_scale, hipStream_t s)
{
  if (opt_state_.count != n_params_ || grads_.count != grads_alloc()) throw std::runtime_error("optimizer_step_range before forward_backward");
  if (hi > n_params_ || lo >= hi) throw std::runtime_error("optimizer_step_range: invalid parameter range");
  launch_adam(lo, hi, n_mlp_, grad_scale / (float)kLossScale, lr_, cfg_.beta1, cfg_.beta2, cfg_.epsilon, cfg_.l2_reg, opt_state_.ptr,
              (half_t*)params_f16_.ptr, (half_t*)grads_.ptr, s);
}

float* Network::grads_as_f32(hipStream_t s)
{
  if (grads_.count != grads_alloc()) throw std::runtime_error("no gradient yet: call forward_backward / TrainBegin first");
  grads_f32_.ensure(n_params_);
  unpack_grads_f16_kernel<<<(uint32_t)std::min<size_t>((n_params_ + 255) / 256, 8192), 256, 0, s>>>((const half_t*)grads_.ptr, grads_f32_.ptr, n_params_);
  VNR_HIP_CHECK(hipGetLastError());
  return grads_f32_.ptr;
}

// ------------------------------------------------------------------------------------------------ diagnostics
const void* Network::training_buffer(int which, size_t* bytes) const
{
  switch (which) {
  case 0: *bytes = grads_.count ? n_params_ * 2 : 0; return grads_.ptr;
  case 1: *bytes = ws_dfeat_.bytes(); return ws_dfeat_.ptr;
  case 2: *bytes = ws_features_.bytes(); return ws_features_.ptr;
  case 3: *bytes = ws_acts_.bytes(); return ws_acts_.ptr;
  default: throw std::runtime_error("training_buffer: 0 gradients, 1 dL/dfeatures, 2 features, 3 activations");
  }
}

void Network::rescatter_grid_gradients(const float* d_coords, size_t n, hipStream_t s)
{
  if (grads_.count != grads_alloc() || ws_batch_ != n || n == 0) throw std::runtime_error("rescatter_grid_gradients: no forward_backward of this batch size to repeat");
  VNR_HIP_CHECK(hipMemsetAsync(grads_.ptr + n_mlp_, 0, (grads_alloc() - n_mlp_) * sizeof(uint16_t), s));
  scatter_grid_gradients(d_coords, n, s, nullptr);
  VNR_HIP_CHECK(hipGetLastError());
}

// block sums of (g - ref)^2 and ref^2 in double, [blocks][2]
__global__ void __launch_bounds__(256) grad_distance_kernel(const half_t* __restrict__ g, const half_t* __restrict__ ref, size_t lo, size_t hi, double* __restrict__ out)
{
  __shared__ double red[2][256];
  double d2 = 0.0, r2 = 0.0;
  for (size_t i = lo + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < hi; i += (size_t)gridDim.x * blockDim.x) {
    const double a = (double)(float)g[i], b = (double)(float)ref[i];
    d2 += (a - b) * (a - b); r2 += b * b;
  }
  red[0][threadIdx.x] = d2; red[1][threadIdx.x] = r2;
  __syncthreads();
  for (uint32_t st = 128; st > 0; st >>= 1) {
    if (threadIdx.x < st) { red[0][threadIdx.x] += red[0][threadIdx.x + st]; red[1][threadIdx.x] += red[1][threadIdx.x + st]; }
    __syncthreads();
  }
  if (threadIdx.x == 0) { out[2 * blockIdx.x] = red[0][0]; out[2 * blockIdx.x + 1] = red[1][0]; }
}

void Network::gradient_distance(const uint16_t* d_ref, double out[4], hipStream_t s)
{
  if (grads_.count != grads_alloc()) throw std::runtime_error("gradient_distance: no gradient yet");
  constexpr uint32_t kBlocks = 256;
  DeviceBuffer<double>& part = scratch_of(this).distance_partials;   // (per network, released with its scratch: nothing static that outlives the HIP runtime)
  part.ensure(4 * kBlocks);
  grad_distance_kernel<<<kBlocks, 256, 0, s>>>((const half_t*)grads_.ptr, (const half_t*)d_ref, 0, n_mlp_, part.ptr);
  grad_distance_kernel<<<kBlocks, 256, 0, s>>>((const half_t*)grads_.ptr, (const half_t*)d_ref, n_mlp_, n_params_, part.ptr + 2 * kBlocks);
  VNR_HIP_CHECK(hipGetLastError());
  std::vector<double> h(4 * kBlocks);
  part.download(h.data(), h.size(), s);
  for (int k = 0; k < 4; ++k) out[k] = 0.0;
  for (uint32_t b = 0; b < kBlocks; ++b) { out[0] += h[2 * b]; out[1] += h[2 * b + 1]; out[2] += h[2 * kBlocks + 2 * b]; out[3] += h[2 * kBlocks + 2 * b + 1]; }
}

__global__ void pack_grads_f16_kernel(const float* __restrict__ in, half_t* __restrict__ out, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) out[i] = (half_t)in[i];
}

void Network::set_grads_from_f32(const float* host, size_t count, hipStream_t s)
{
  if (count != n_params_) throw std::runtime_error("gradient count mismatch");
  ensure_training_state(s);
  grads_f32_.ensure(n_params_);
  VNR_HIP_CHECK(hipMemcpyAsync(grads_f32_.ptr, host, count * sizeof(float), hipMemcpyHostToDevice, s));
  pack_grads_f16_kernel<<<(uint32_t)std::min<size_t>((n_params_ + 255) / 256, 8192), 256, 0, s>>>(grads_f32_.ptr, (half_t*)grads_.ptr, n_params_);
  VNR_HIP_CHECK(hipGetLastError());
  VNR_HIP_CHECK(hipStreamSynchronize(s));
}

void Network::optimizer_finish_step(hipStream_t s)
{
  ++steps_;
  // EXTERNAL tcnn ExponentialDecayOptimizer::step
  if (cfg_.has_decay && steps_ >= cfg_.decay_start && cfg_.decay_interval > 0 && steps_ % cfg_.decay_interval == 0) lr_ *= cfg_.decay_base;
  refresh_inference_weights(s);
  profile_mark(kTrainPhases, s);
}

double Network::training_loss(hipStream_t s)
{
  TrainScratch& ts = scratch_of(this);
  if (ts.loss_blocks == 0) return 0.0;
  std::vector<float> h(ts.loss_blocks);
  ws_loss_.download(h.data(), ts.loss_blocks, s);
  double sum = 0.0;
  for (float v : h) sum += v;
  return sum;
}

}  // namespace vnr
