// infer_kernel.h — the fused hash-grid encode + MLP evaluation kernel (design notes: network_infer.hip), a template over the encoding
// shape (F, padded width) and the FullyFusedMLP width; instantiated per width in network_infer_w{16,32,64,128}.hip.
#pragma once
#include <algorithm>
#include <cstdlib>

#include "infer_tile.h"
#include "pack_rays.h"

namespace vnr {

// ------------------------------------------------------------------------------------------------
// the MLP of a launch: the forward weight image and its shape
struct FusedMlp {
  const uint16_t* packed;   // forward image (packed_mlp_halves)
  uint32_t lds_halves, width, n_hidden_matmuls, activation, output_activation;
  bool general;             // the model needs a GENERAL instance (grid_device.h gather_corners; Network::common_kind)
  bool weights_global;      // the weight image exceeds the LDS (128 neurons: >= 6 hidden layers; 64: ~20; ...): the A operands are read from global memory
  float quantize_threshold;
};

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
  uint32_t activation;       // kAct* (infer_tile.h) of the hidden layers
  uint32_t output_activation;
  uint32_t lds_halves;
  uint32_t sharers;          // kernels of this kind expected to share the GPU (host-side launch sizing only)
  uint32_t weights_global;   // GENERAL instances: the weight image does not fit the LDS and stays in global memory
  float quantize_threshold;  // GENERAL instances (tcnn_impl_decoder.cu:120)
  PackArgs pack;             // MODE 0, queue launches of the ray marcher: the iteration's ray packing as a prologue (pack.n_blocks > 0)
};

// MODE 0: inference (out only), 1: encode only (features_out), 2: training forward (features + acts + out)
// GENERAL: grid_device.h gather_corners
template <int F, int K_IN, int W, int MODE, bool GENERAL>
__global__ void __launch_bounds__(64 * MlpShape<W>::WAVES) fused_infer_kernel(const InferArgs args)
{
  constexpr int NCHUNK = K_IN / 8;  // half8 chunks of the feature vector
  constexpr uint32_t WAVES = MlpShape<W>::WAVES;
  extern __shared__ __attribute__((aligned(16))) half_t lds[];

  const uint32_t lane = threadIdx.x & 63u;
  const uint32_t wave = threadIdx.x >> 6;

  // The ray marcher's packing of this iteration's survivors (pack_rays.h) needs what the march kernel wrote and nothing this kernel
  // writes: as a prologue it costs no launch of its own on the chain march -> evaluate -> pack -> march.  Work items are dealt over the
  // blocks; the words of LDS it uses are overwritten by the weights afterwards.
  if (MODE == 0 && WAVES == 4 && args.pack.n_blocks) {
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
  const uint32_t waves_per_xcd = (gridDim.x >> 3) * WAVES;
  const uint32_t tile_end = min(n_tiles, (xcd + 1u) * per_xcd);
  // the grid is sized by an upper bound of the sample count: a block none of whose waves has a tile leaves at once
  if (xcd * per_xcd + (blockIdx.x >> 3) * WAVES >= tile_end) return;

  constexpr bool CAN_GLOBAL = GENERAL;   // an image beyond 160 KiB: 128 neurons from 6 hidden layers on, 64 from ~20, 32 from ~75, 16 from ~300
  const bool wglobal = CAN_GLOBAL && args.weights_global != 0u;
  if (MODE != 1 && !wglobal) {  // stage the packed weights once per block
    const uint4_t* src = (const uint4_t*)args.packed_mlp;
    uint4_t* dst = (uint4_t*)lds;
    for (uint32_t i = threadIdx.x; i < args.lds_halves / 8; i += blockDim.x) dst[i] = src[i];
    __syncthreads();
  }
  const uint32_t nh = args.n_hidden_matmuls;
  const uint32_t act = args.activation;
  const uint32_t h = lane >> 5;   // lane half
  const uint32_t r = lane & 31u;  // row (A operand) / column (B, D operands)
  const table_rsrc_t rsrc = make_table_rsrc(args.table, args.table_bytes);

  for (uint32_t tile = xcd * per_xcd + (blockIdx.x >> 3) * WAVES + wave; tile < tile_end; tile += waves_per_xcd) {
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
    // (Round 5 tried the corners of 8 levels in flight at once for the TRAINING forward -- one tile per wave and SIMD, 67 us for 65 536 random
    // points where this kernel's throughput would take 5: 73.8 us.  Those 8.4 M random 4-byte reads are 8.4 M distinct 64-byte sectors,
    // 0.54 GB in 67 us = 8 TB/s: the pass is at the memory system's sector rate, not waiting on a chain of round trips.  Removed.)
    half8_t feat[NCHUNK];
    encode_tile<F, K_IN, GENERAL>(args.levels, args.n_levels, args.interpolation, rsrc, args.brick_image, p.x, p.y, p.z, feat,
                                  GENERAL ? args.quantize_threshold : 0.0f);

    if (MODE != 0 && args.features_out && i < n) {
      half8_t* dst = (half8_t*)(args.features_out + (size_t)i * K_IN);
#pragma unroll
      for (int c = 0; c < NCHUNK; ++c) dst[c] = feat[c];
    }
    if (MODE == 1) continue;

    // ---- MLP on the wave's 64 samples (infer_tile.h) ------------------------------------------------
    float y;
    if (CAN_GLOBAL && wglobal)   // (two calls, not one pointer select: each keeps its address space, ds_read_b128 against global_load_dwordx4)
      y = mlp_tile<W, K_IN, MODE == 2, GENERAL>(args.packed_mlp, feat, nh, act, h, r, args.acts_out, n, tile * 64u);
    else
      y = mlp_tile<W, K_IN, MODE == 2, GENERAL>((const half_t*)lds, feat, nh, act, h, r, args.acts_out, n, tile * 64u);
    // network output is produced in half precision (output activation on the half), then cast to float (tcnn_impl.cu:421-431)
    if (i < n) args.out[out_index] = finish_output<GENERAL>(y, args.output_activation);
  }
}

// ------------------------------------------------------------------------------------------------
template <int F, int K_IN, int W, int MODE, bool GENERAL>
static void launch_one(const InferArgs& a, size_t n_max, hipStream_t s)
{
  constexpr uint32_t WAVES = MlpShape<W>::WAVES;
  const Runtime& rt = Runtime::get();
  const uint32_t n_tiles = div_round_up(n_max, 64);
  uint32_t blocks = div_round_up(n_tiles, WAVES);
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
  const size_t shmem = (MODE == 1 || a.weights_global) ? 16 : (size_t)a.lds_halves * sizeof(uint16_t);
  // (128 neurons: the image takes most of the LDS, one block of 8 waves per CU)
  const uint32_t fit = (uint32_t)std::max<size_t>(1, std::min<size_t>(4, (160 * 1024) / std::max<size_t>(shmem, 1)));
  uint32_t max_blocks = (uint32_t)rt.n_cus * (forced ? forced : std::min(fit, a.sharers >= 2 ? 3u : 4u));
  // The ray marcher's queue (count on the device, n_max an upper bound of which a frame fills 25-30 %): more blocks than fit,
  // so that the hardware hands them out as room appears.  With a second kernel and the march kernels of the other ray half
  // on the GPU a resident grid of fixed size either leaves room unused or waits for it with its tiles already dealt out.
  // Swept on the C4 frame and on a 1/8 share of it (gpurun_out/s3_share_sweep*.log): best at 2-3 tiles per wave, i.e.
  // 16-32 blocks per CU for the whole frame (4.59 -> 4.27 ms) and 4-6 for the share (0.77 -> 0.74 ms).
  if (!forced && a.n_ptr && a.queue_mode) max_blocks = std::min((uint32_t)rt.n_cus * 32u, std::max((uint32_t)rt.n_cus * 4u, n_tiles / 33u));
  if (blocks > max_blocks) blocks = max_blocks;
  blocks = next_multiple(blocks, 8);
  auto kernel = fused_infer_kernel<F, K_IN, W, MODE, GENERAL>;
  static bool attr_set = false;
  if (!attr_set) {
    VNR_HIP_CHECK(hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    attr_set = true;
  }
  kernel<<<blocks, 64 * WAVES, shmem, s>>>(a);
  VNR_HIP_CHECK(hipGetLastError());
}

template <int W, int MODE, bool GENERAL>
static void dispatch(uint32_t F, uint32_t K_IN, const InferArgs& a, size_t n_max, hipStream_t s)
{
#define VNR_CASE(f, k) if (F == f && K_IN == k) return launch_one<f, k, W, MODE, GENERAL>(a, n_max, s)
  VNR_CASE(1, 16); VNR_CASE(1, 32);
  VNR_CASE(2, 16); VNR_CASE(2, 32); VNR_CASE(2, 48); VNR_CASE(2, 64);
  VNR_CASE(4, 16); VNR_CASE(4, 32); VNR_CASE(4, 48); VNR_CASE(4, 64); VNR_CASE(4, 80); VNR_CASE(4, 96); VNR_CASE(4, 112); VNR_CASE(4, 128);
  VNR_CASE(8, 16); VNR_CASE(8, 32); VNR_CASE(8, 48); VNR_CASE(8, 64); VNR_CASE(8, 80); VNR_CASE(8, 96); VNR_CASE(8, 112); VNR_CASE(8, 128);
#undef VNR_CASE
  throw std::runtime_error("unsupported encoding shape: n_features_per_level=" + std::to_string(F) +
                           " padded width=" + std::to_string(K_IN));
}

// one translation unit per width and kind (network_infer_w*.hip) instantiates the inference and the training-forward kernels
#define VNR_DEFINE_FUSED_WIDTH(W, GENERAL, SUFFIX)                                                                        \
  void launch_fused_w##W##SUFFIX(int mode, uint32_t F, uint32_t K_IN, const InferArgs& a, size_t n_max, hipStream_t s)    \
  {                                                                                                                       \
    if (mode == 0) dispatch<W, 0, GENERAL>(F, K_IN, a, n_max, s);                                                         \
    else dispatch<W, 2, GENERAL>(F, K_IN, a, n_max, s);                                                                   \
  }

}  // namespace vnr
