// network_infer.hip — fused hash-grid encode + MLP inference for gfx950 (the dominant kernel of the hot path).
//
// Replaces tcnn `NetworkWithInputEncoding::inference` as called from core/network.cu:1043-1052
// (NeuralVolume::inference -> tcnn_network.h:254-271 -> tcnn_impl.cu:438-448); semantic spec:
// core/networks/tcnn_impl_decoder.cu:7-175 (encode), tcnn_threadblock.h:59-144,221-328,446-505 (MLP),
// tcnn_impl.cu:104-286 (the reference's own fused encode+MLP kernel).
//
// MI355X design (not the reference's 32-lane WMMA / shared-memory activation tiling):
//  * one wave64 owns 64 samples end to end; nothing but the weights ever touches LDS.
//  * encode: lane = sample, level is wave-uniform, so every gather instruction reads 64 spatially coherent
//    samples of ONE level (scalar level constants, best L1/L2 locality); features stay in VGPRs as fp16.
//    Gathers are buffer loads with 32-bit offsets; index arithmetic avoids 32-bit integer multiplies.
//  * MLP on v_mfma_f32_32x32x16_f16 computed TRANSPOSED: Y^T[64 x samples] = W[64 x K] . X^T[K x samples].
//    Weights are the A operand (read from an LDS image, conflict-free ds_read_b128), activations the B operand.
//    The 32x32 accumulator of one layer (column = sample on the lane, rows = neurons in registers) is, after
//    ReLU + fp16 pack, directly the B operand of the next layer (k-order permutation folded into the packed
//    weight image), so activations never leave registers.  v_permlane32_swap builds the first layer's B
//    operand from the per-lane feature vectors.  The two 32-sample column tiles of a wave run the MLP one
//    after the other to keep the accumulator footprint at 32 registers (occupancy).
//  * persistent blocks, XCD-contiguous tile ranges, sample count optionally read from device memory so the
//    ray marcher never syncs with the host.
#include "infer_tile.h"
#include "pack_rays.h"

namespace vnr {

// ------------------------------------------------------------------------------------------------
// packed (LDS image) weight layout, in halves:
//   layer 1      : [s < K_IN/16][h < 2][row < 64][j < 8]  = W1[row][16 s + 8 h + j]
//   hidden l     : [s < 4][h < 2][row < 64][j < 8]        = Wh[row][16 s + 8 (j>>2) + 4 h + (j&3)]
//   last (row 0) : [s < 4][h < 2][j < 8]                  = Wl[0][16 s + 8 (j>>2) + 4 h + (j&3)]
// ------------------------------------------------------------------------------------------------
__host__ __device__ inline uint32_t packed_mlp_halves(uint32_t in_width, uint32_t n_hidden_matmuls)
{
  return (in_width / 16) * 1024 + n_hidden_matmuls * 4096 + 64;
}

__global__ void pack_mlp_kernel(const half_t* __restrict__ params, half_t* __restrict__ packed, uint32_t in_width,
                                uint32_t n_hidden_matmuls)
{
  const uint32_t total = packed_mlp_halves(in_width, n_hidden_matmuls);
  const uint32_t l1 = (in_width / 16) * 1024;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total; e += gridDim.x * blockDim.x) {
    half_t v;
    if (e < l1) {
      const uint32_t j = e & 7, row = (e >> 3) & 63, h = (e >> 9) & 1, s = e >> 10;
      v = params[row * in_width + 16 * s + 8 * h + j];
    } else if (e < l1 + n_hidden_matmuls * 4096) {
      const uint32_t r = e - l1;
      const uint32_t layer = r >> 12, q = r & 4095;
      const uint32_t j = q & 7, row = (q >> 3) & 63, h = (q >> 9) & 1, s = q >> 10;
      const uint32_t k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      v = params[kWidth * in_width + layer * 4096 + row * 64 + k];
    } else {
      const uint32_t q = e - l1 - n_hidden_matmuls * 4096;
      const uint32_t j = q & 7, h = (q >> 3) & 1, s = q >> 4;
      const uint32_t k = 16 * s + 8 * (j >> 2) + 4 * h + (j & 3);
      v = params[kWidth * in_width + n_hidden_matmuls * 4096 + k];  // row 0 of the 16 x 64 last layer
    }
    packed[e] = v;
  }
}

void launch_pack_mlp(const uint16_t* params, uint16_t* packed, uint32_t in_width, uint32_t n_hidden_matmuls, hipStream_t s)
{
  const uint32_t total = packed_mlp_halves(in_width, n_hidden_matmuls);
  pack_mlp_kernel<<<div_round_up(total, 256), 256, 0, s>>>((const half_t*)params, (half_t*)packed, in_width, n_hidden_matmuls);
}

// ------------------------------------------------------------------------------------------------
struct InferArgs {
  const LevelInfo* levels;   // device table of per-level constants (scalar loads)
  uint32_t n_levels, interpolation;
  const half_t* table;       // grid part of the parameter blob
  uint32_t table_bytes;
  const uint8_t* brick_image;  // de-hashed copies of the levels whose LevelInfo::brick is set (network.h), or null
  const half_t* packed_mlp;  // LDS image
  const float* coords;       // [n][3]
  float* out;                // [n]
  half_t* features_out;      // encode-only / training: [n][K_IN] row-major (may be null)
  half_t* acts_out;          // training: [(nh+1)][n][64] post-activation hidden outputs (may be null)
  const uint32_t* n_ptr;     // if non-null the sample count is read from here
  const uint32_t* dest;      // if non-null, sample i's result goes to out[dest[i]]
  uint32_t queue_mode;       // 1: coords are 16-byte records {x, y, z, dest} and the result goes to out[dest * out_stride]
  uint32_t out_stride;
  uint32_t n;
  uint32_t n_hidden_matmuls;
  uint32_t activation;       // 0 none, 1 relu
  uint32_t lds_halves;
  uint32_t sharers;          // kernels of this kind expected to share the GPU (host-side launch sizing only)
  uint32_t lds_table_halves; // VNR_LDS_LEVELS experiment: halves of the table's head staged behind the weights (0: none)
  PackArgs pack;             // MODE 0, queue launches of the ray marcher: the iteration's ray packing as a prologue (pack.n_blocks > 0)
};

// MODE 0: inference (out only), 1: encode only (features_out), 2: training forward (features + acts + out)
template <int F, int K_IN, int MODE>
__global__ void __launch_bounds__(256) fused_infer_kernel(const InferArgs args)
{
  constexpr int NCHUNK = K_IN / 8;  // half8 chunks of the feature vector
  extern __shared__ __attribute__((aligned(16))) half_t lds[];

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;

  // The ray marcher's packing of this iteration's survivors (pack_rays.h) needs what the march kernel wrote and nothing this kernel
  // writes: as a prologue it costs no launch of its own on the chain march -> evaluate -> pack -> march.  Work items are dealt over the
  // blocks; the words of LDS it uses are overwritten by the weights afterwards.
  if (MODE == 0 && args.pack.n_blocks) {
    uint32_t* s_part = (uint32_t*)lds;
    for (uint32_t item = blockIdx.x; item < args.pack.n_blocks; item += gridDim.x) {
      pack_rays_block<4>(args.pack, item, s_part);
      __syncthreads();
    }
  }
  const uint32_t n = args.n_ptr ? min(*args.n_ptr, args.n) : args.n;   // (args.n: the caller's upper bound when the count lives on the device)
  const uint32_t n_tiles = (n + 63u) >> 6;
  // XCD-contiguous tile ranges: blocks with equal (blockIdx % 8) share an XCD / L2 (speed only)
  const uint32_t xcd = blockIdx.x & 7u;
  const uint32_t per_xcd = (n_tiles + 7u) >> 3;
  const uint32_t waves_per_xcd = (gridDim.x >> 3) * 4u;
  const uint32_t tile_end = min(n_tiles, (xcd + 1u) * per_xcd);
  // the grid is sized by an upper bound of the sample count: a block none of whose waves has a tile leaves at once
  if (xcd * per_xcd + (blockIdx.x >> 3) * 4u >= tile_end) return;

  if (MODE != 1) {  // stage the packed weights once per block
    const uint4_t* src = (const uint4_t*)args.packed_mlp;
    uint4_t* dst = (uint4_t*)lds;
    for (uint32_t i = threadIdx.x; i < args.lds_halves / 8; i += blockDim.x) dst[i] = src[i];
#if defined(VNR_LDS_LEVELS)
    {
      const uint4_t* ts = (const uint4_t*)args.table;
      uint4_t* td = (uint4_t*)(lds + args.lds_halves);
      for (uint32_t i = threadIdx.x; i < args.lds_table_halves / 8; i += blockDim.x) td[i] = ts[i];
    }
#endif
    __syncthreads();
  }
  const uint32_t nh = args.n_hidden_matmuls;
  const bool relu = args.activation == 1;
  const uint32_t h = lane >> 5;   // lane half
  const uint32_t r = lane & 31u;  // row (A operand) / column (B, D operands)
  const table_rsrc_t rsrc = make_table_rsrc(args.table, args.table_bytes);

  for (uint32_t tile = xcd * per_xcd + (blockIdx.x >> 3) * 4u + wave; tile < tile_end; tile += waves_per_xcd) {
    const uint32_t i = tile * 64u + lane;
    const uint32_t ic = min(i, n - 1u);
    float3_packed p;
    uint32_t out_index = i;
    if (args.queue_mode) {  // ray marcher's sample queue: one 16-byte load per sample
      const uint4_t rec = ((const uint4_t*)args.coords)[ic];
      p = {__uint_as_float(rec.x), __uint_as_float(rec.y), __uint_as_float(rec.z)};
      out_index = rec.w * args.out_stride;
    } else {
      p = ((const float3_packed*)args.coords)[ic];
      if (args.dest) out_index = args.dest[ic];
    }

    // ---- encode: lane = sample, level wave-uniform (infer_tile.h) -----------------------------------
    half8_t feat[NCHUNK];
#if defined(VNR_LDS_LEVELS)
    encode_tile<F, K_IN>(args.levels, args.n_levels, args.interpolation, rsrc, args.brick_image, p.x, p.y, p.z, feat,
                         args.lds_table_halves ? (const half_t*)lds + args.lds_halves : nullptr);
#else
    encode_tile<F, K_IN>(args.levels, args.n_levels, args.interpolation, rsrc, args.brick_image, p.x, p.y, p.z, feat);
#endif

    if (MODE != 0 && args.features_out && i < n) {
      half8_t* dst = (half8_t*)(args.features_out + (size_t)i * K_IN);
#pragma unroll
      for (int c = 0; c < NCHUNK; ++c) dst[c] = feat[c];
    }
    if (MODE == 1) continue;

    // ---- MLP on the wave's 64 samples (infer_tile.h) ------------------------------------------------
    const float y = mlp_tile<F, K_IN, MODE == 2>((const half_t*)lds, feat, nh, relu, h, r, args.acts_out, n, tile * 64u);
    // network output is produced in half precision and then cast to float (tcnn_impl.cu:421-431)
    if (i < n) args.out[out_index] = (float)(half_t)y;
  }
}

// ------------------------------------------------------------------------------------------------
template <int F, int K_IN, int MODE>
static void launch_one(const InferArgs& a, size_t n_max, hipStream_t s)
{
  const Runtime& rt = Runtime::get();
  const uint32_t n_tiles = div_round_up(n_max, 64);
  uint32_t blocks = div_round_up(n_tiles, 4);
  // persistent blocks of 4 waves; 113 registers allow 4 per CU.  How many pay depends on what limits the kernel (MI355X, C4
  // bench frame, kernel-only G samples/s):
  //   reading the hashed parameter blob (bound by fetched lines): 4 blocks 5.98, 3 blocks 6.20, 2 blocks 6.25, 1 block 4.77
  //   reading the brick image (2.4 x fewer lines, latency matters again), one stream: 2 blocks 8.1, 3 blocks 10.1, 4 blocks 11.1;
  //   two ray halves on two streams (two of these kernels share the GPU): 2 blocks 179, 3 blocks 195, 4 blocks 192 frames/s
  // so the caller says how many kernels share the GPU (`sharers`): 4 blocks alone, 3 with a second stream.
  // VNR_AMD_INFER_BLOCKS_PER_CU (1..4) overrides, for diagnostics.
  static const uint32_t forced = [] {
    const char* e = std::getenv("VNR_AMD_INFER_BLOCKS_PER_CU");
    const int v = e ? std::atoi(e) : 0;
    return (uint32_t)(v >= 1 && v <= 64 ? v : 0);
  }();
  uint32_t max_blocks = (uint32_t)rt.n_cus * (forced ? forced : (a.sharers >= 2 ? 3u : 4u));
  // The ray marcher's queue (count on the device, n_max an upper bound of which a frame fills 25-30 %): more blocks than fit,
  // so that the hardware hands them out as room appears.  With a second kernel and the march kernels of the other ray half
  // on the GPU a resident grid of fixed size either leaves room unused or waits for it with its tiles already dealt out.
  // Swept on the C4 frame and on a 1/8 share of it (gpurun_out/s3_share_sweep*.log): best at 2-3 tiles per wave, i.e.
  // 16-32 blocks per CU for the whole frame (4.59 -> 4.27 ms) and 4-6 for the share (0.77 -> 0.74 ms).
  if (!forced && a.n_ptr && a.queue_mode) max_blocks = std::min((uint32_t)rt.n_cus * 32u, std::max((uint32_t)rt.n_cus * 4u, n_tiles / 33u));
  if (blocks > max_blocks) blocks = max_blocks;
  blocks = next_multiple(blocks, 8);
  const size_t shmem = MODE == 1 ? 16 : ((size_t)a.lds_halves + a.lds_table_halves) * sizeof(uint16_t);
  auto kernel = fused_infer_kernel<F, K_IN, MODE>;
  static bool attr_set = false;
  if (!attr_set) {
    VNR_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  kernel<<<blocks, 256, shmem, s>>>(a);
  VNR_HIP_CHECK(hipGetLastError());
}

template <int MODE>
static void dispatch(uint32_t F, uint32_t K_IN, const InferArgs& a, size_t n_max, hipStream_t s)
{
#define VNR_CASE(f, k) if (F == f && K_IN == k) return launch_one<f, k, MODE>(a, n_max, s)
  VNR_CASE(1, 16); VNR_CASE(1, 32);
  VNR_CASE(2, 16); VNR_CASE(2, 32); VNR_CASE(2, 48); VNR_CASE(2, 64);
  VNR_CASE(4, 16); VNR_CASE(4, 32); VNR_CASE(4, 48); VNR_CASE(4, 64);
  VNR_CASE(8, 16); VNR_CASE(8, 32); VNR_CASE(8, 48); VNR_CASE(8, 64); VNR_CASE(8, 96); VNR_CASE(8, 128);
#undef VNR_CASE
  throw std::runtime_error("unsupported encoding shape: n_features_per_level=" + std::to_string(F) +
                           " padded width=" + std::to_string(K_IN));
}

// ------------------------------------------------------------------------------------------------ generic kernel
// Every model the reference accepts that the MFMA kernels do not cover: FullyFusedMLP n_neurons 16 / 32 / 128
// (tcnn_impl.cu:315-347 dispatches WIDTH 16, 32, 64, 128), Nearest interpolation (tcnn_impl_decoder.cu:73-94) and a non-zero
// quantize_threshold (:120).  One lane = one sample, plain loops, fp32 accumulation with the layer outputs rounded to fp16 and the
// activation applied on the fp16 value, as the reference's fragments are (tcnn_threadblock.h:83,125).  It exists so that any
// params.json the reference wrote loads, evaluates and renders; it is not a fast path (no MFMA, weights from L1 / L2).
struct GenericArgs {
  const LevelInfo* levels;
  uint32_t n_levels, n_active_levels, n_features, interpolation;
  float quantize_threshold;
  const half_t* params;     // tcnn-order blob: MLP weights then grid
  size_t n_mlp;
  uint32_t in_width, width, n_hidden_matmuls, activation;
  const float* coords;
  float* out;
  half_t* features_out;
  half_t* acts_out;         // training: [(nh + 1)][n][width] post-activation outputs of the hidden layers (may be null)
  const uint32_t* n_ptr;
  const uint32_t* dest;
  uint32_t queue_mode, out_stride, n, encode_only;
};

__global__ void __launch_bounds__(128) generic_infer_kernel(const GenericArgs a)
{
  const uint32_t n = a.n_ptr ? min(*a.n_ptr, a.n) : a.n;
  const half_t* table = a.params + a.n_mlp;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    float x, y, z;
    uint32_t out_index = i;
    if (a.queue_mode) {
      const uint4_t rec = ((const uint4_t*)a.coords)[i];
      x = __uint_as_float(rec.x); y = __uint_as_float(rec.y); z = __uint_as_float(rec.z);
      out_index = rec.w * a.out_stride;
    } else {
      x = a.coords[3 * (size_t)i]; y = a.coords[3 * (size_t)i + 1]; z = a.coords[3 * (size_t)i + 2];
      if (a.dest) out_index = a.dest[i];
    }
    half_t feat[128];
    for (uint32_t k = 0; k < a.in_width; ++k) feat[k] = (half_t)0.0f;
    const uint32_t F = a.n_features;
    for (uint32_t l = 0; l < a.n_active_levels; ++l) {   // levels at or beyond max_level + 1e-3 stay zero (tcnn_impl_decoder.cu:17-35)
      const LevelInfo lv = a.levels[l];
      const CornerSetup c = level_setup(lv, a.interpolation == 1u ? 1u : 0u, x, y, z);
      const half_t* base = table + (size_t)lv.offset * F;
      if (a.interpolation == 2u) {   // Nearest: the entry of the lower corner as it is (:73-94)
        const uint32_t idx = level_index(lv, c.g[0], c.g[1], c.g[2]);
        for (uint32_t f = 0; f < F; ++f) feat[l * F + f] = base[(size_t)idx * F + f];
        continue;
      }
      half_t acc[8];
      for (uint32_t f = 0; f < F; ++f) acc[f] = (half_t)0.0f;
      for (int corner = 0; corner < 8; ++corner) {
        const uint32_t idx = level_index(lv, c.g[0] + (corner & 1), c.g[1] + ((corner >> 1) & 1), c.g[2] + ((corner >> 2) & 1));
        const float w = corner_weight(c, corner);
        for (uint32_t f = 0; f < F; ++f) {
          float data = (float)base[(size_t)idx * F + f];
          if (fabsf(data) < a.quantize_threshold) data = 0.0f;   // :120
          float prod = w * data;
          asm volatile("" : "+v"(prod));   // (T)(weight * data) rounds twice: no fused conversion (grid_device.h encode_level)
          acc[f] = acc[f] + (half_t)prod;
        }
      }
      for (uint32_t f = 0; f < F; ++f) feat[l * F + f] = acc[f];
    }
    if (a.features_out) for (uint32_t k = 0; k < a.in_width; ++k) a.features_out[(size_t)i * a.in_width + k] = feat[k];
    if (a.encode_only) continue;
    // the MLP: first (W x in), hidden (W x W) x n, last row 0 of (16 x W); weights row-major [out][in]
    const uint32_t W = a.width;
    half_t h0[128], h1[128];
    const half_t* w = a.params;
    for (uint32_t o = 0; o < W; ++o) {
      float s = 0.0f;
      for (uint32_t k = 0; k < a.in_width; ++k) s = __builtin_fmaf((float)w[(size_t)o * a.in_width + k], (float)feat[k], s);
      half_t v = (half_t)s;
      if (a.activation == 1u && (__builtin_bit_cast(unsigned short, v) & 0x8000u)) v = (half_t)0.0f;   // ReLU on the fp16 value
      h0[o] = v;
      if (a.acts_out) a.acts_out[(size_t)i * W + o] = v;
    }
    w += (size_t)W * a.in_width;
    half_t* cur = h0; half_t* nxt = h1;
    for (uint32_t layer = 0; layer < a.n_hidden_matmuls; ++layer) {
      for (uint32_t o = 0; o < W; ++o) {
        float s = 0.0f;
        for (uint32_t k = 0; k < W; ++k) s = __builtin_fmaf((float)w[(size_t)o * W + k], (float)cur[k], s);
        half_t v = (half_t)s;
        if (a.activation == 1u && (__builtin_bit_cast(unsigned short, v) & 0x8000u)) v = (half_t)0.0f;
        nxt[o] = v;
        if (a.acts_out) a.acts_out[((size_t)(layer + 1) * n + i) * W + o] = v;
      }
      w += (size_t)W * W;
      half_t* t = cur; cur = nxt; nxt = t;
    }
    float s = 0.0f;
    for (uint32_t k = 0; k < W; ++k) s = __builtin_fmaf((float)w[k], (float)cur[k], s);
    a.out[out_index] = (float)(half_t)s;   // the network's output is produced in half precision (tcnn_impl.cu:421-431)
  }
}

void launch_generic(int mode, const GridDevice& grid, const ModelConfig& cfg, uint32_t n_active_levels, uint32_t in_width, const LevelInfo* d_levels,
                    const uint16_t* params, size_t n_mlp, const float* coords, float* out, uint16_t* features_out, size_t n, const uint32_t* d_n,
                    size_t n_max, hipStream_t s, const uint32_t* d_dest, uint32_t queue_out_stride, uint16_t* acts_out)
{
  if (n_max == 0) return;
  if (n_max > 0xffffffc0ull) throw std::runtime_error("inference batch too large (max 2^32-64 samples per call)");
  if (in_width > 128 || cfg.n_neurons > 128) throw std::runtime_error("generic network kernel: width > 128");
  GenericArgs a;
  a.levels = d_levels; a.n_levels = grid.n_levels; a.n_active_levels = n_active_levels; a.n_features = grid.n_features;
  a.interpolation = cfg.interpolation; a.quantize_threshold = cfg.quantize_threshold;
  a.params = (const half_t*)params; a.n_mlp = n_mlp; a.in_width = in_width; a.width = cfg.n_neurons;
  a.n_hidden_matmuls = cfg.n_hidden_layers - 1; a.activation = cfg.activation;
  a.coords = coords; a.out = out; a.features_out = (half_t*)features_out; a.acts_out = (half_t*)acts_out; a.n_ptr = d_n; a.dest = d_dest;
  a.queue_mode = queue_out_stride ? 1u : 0u; a.out_stride = queue_out_stride; a.n = d_n ? (uint32_t)n_max : (uint32_t)n; a.encode_only = mode == 1 ? 1u : 0u;
  const uint32_t blocks = std::min<uint32_t>(div_round_up(n_max, 128), (uint32_t)Runtime::get().n_cus * 16u);
  generic_infer_kernel<<<blocks, 128, 0, s>>>(a);
  VNR_HIP_CHECK(hipGetLastError());
}

void launch_fused(int mode, const GridDevice& grid, uint32_t in_width, uint32_t n_hidden_matmuls, uint32_t activation,
                  const LevelInfo* d_levels, const uint16_t* table, size_t table_bytes, const uint16_t* packed, uint32_t lds_halves, const float* coords,
                  float* out, uint16_t* features_out, uint16_t* acts_out, size_t n, const uint32_t* d_n, size_t n_max, hipStream_t s,
                  const uint32_t* d_dest, uint32_t queue_out_stride, const uint8_t* brick_image, uint32_t sharers, const PackArgs* pack)
{
  if (n_max == 0) { if (pack && pack->n_blocks) throw std::runtime_error("internal: ray packing fused into an empty evaluation launch"); return; }
  if (n_max > 0xffffffc0ull) throw std::runtime_error("inference batch too large (max 2^32-64 samples per call)");
  if (table_bytes >= (1ull << 32)) throw std::runtime_error("hash table >= 4 GiB is not supported");
  InferArgs a;
  a.levels = d_levels;
  a.n_levels = grid.n_levels;
  a.interpolation = grid.interpolation;
  a.table = (const half_t*)table;
  a.table_bytes = (uint32_t)table_bytes;
  a.brick_image = brick_image;
  a.sharers = sharers;
  a.packed_mlp = (const half_t*)packed;
  a.coords = coords;
  a.out = out;
  a.features_out = (half_t*)features_out;
  a.acts_out = (half_t*)acts_out;
  a.n_ptr = d_n;
  a.dest = d_dest;
  a.queue_mode = queue_out_stride ? 1u : 0u;
  a.out_stride = queue_out_stride;
  a.n = d_n ? (uint32_t)n_max : (uint32_t)n;
  a.n_hidden_matmuls = n_hidden_matmuls;
  a.activation = activation;
  a.lds_halves = lds_halves;
  a.lds_table_halves = 0;
  if (pack && mode == 0) a.pack = *pack; else a.pack.n_blocks = 0;
#if defined(VNR_LDS_LEVELS)
  if (mode == 0 && grid.n_features == 2 && grid.n_levels > VNR_LDS_LEVELS && !grid.levels[VNR_LDS_LEVELS - 1].hashed)
    a.lds_table_halves = (grid.levels[VNR_LDS_LEVELS].offset * 2u + 7u) & ~7u;   // the first levels are the head of the table
#endif
  if (mode == 0) dispatch<0>(grid.n_features, in_width, a, n_max, s);
  else if (mode == 1) dispatch<1>(grid.n_features, in_width, a, n_max, s);
  else dispatch<2>(grid.n_features, in_width, a, n_max, s);
}

}  // namespace vnr
