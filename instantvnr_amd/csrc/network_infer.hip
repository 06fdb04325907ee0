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
#include "infer_kernel.h"

namespace vnr {

// ------------------------------------------------------------------------------------------------
// packed (LDS image) weight layouts, in halves (RW = rows padded to a multiple of 32, KS = W / 16; infer_tile.h MlpShape):
//  forward image (packed_mlp_halves):
//   layer 1      : [s < K_IN/16][h < 2][row < RW][j < 8]  = W1[row][16 s + 8 h + j]                          (0 for row >= W)
//   hidden l     : [s < KS][h < 2][row < RW][j < 8]       = Wh[row][16 s + 8 (j>>2) + 4 h + (j&3)]           (0 for row >= W)
//   last (row 0) : [s < KS][h < 2][j < 8]                 = Wl[0][16 s + 8 (j>>2) + 4 h + (j&3)]
//  backward image (packedT_halves; network_train.hip mlp_backward_kernel), kk = 16 s + 8 (j>>2) + 4 h + (j&3):
//   last row     : [s < KS][h][j]                         = Wl[0][kk]
//   hidden l     : [s < KS][h][row < RW][j]               = Wh_l[kk][row]   (transposed; 0 for row >= W)
//   first        : [s < KS][h][row < RP][j]               = W1[kk][row] or 0, RP = roundup(in_width, 32)
// Both are written by ONE launch whenever the parameters change (an optimizer step ends with it).
// ------------------------------------------------------------------------------------------------
__global__ void pack_mlp_kernel(const half_t* __restrict__ params, half_t* __restrict__ packed, half_t* __restrict__ packedT, uint32_t in_width,
                                uint32_t W, uint32_t nh)
{
  const uint32_t RW = mlp_rows_padded(W), KS = W / 16u, step = 16u * RW, hidden = KS * step;
  const uint32_t total = packed_mlp_halves(in_width, W, nh), totalT = packedT ? packedT_halves(in_width, W, nh) : 0u;
  const uint32_t l1 = (in_width / 16u) * step;
  const uint32_t first_sz = W * in_width;
  for (uint32_t e = blockIdx.x * blockDim.x + threadIdx.x; e < total + totalT; e += gridDim.x * blockDim.x) {
    half_t v = (half_t)0.0f;
    if (e < total) {
      if (e < l1) {
        const uint32_t j = e & 7u, q = e >> 3, row = q % RW, hs = q / RW, h = hs & 1u, s = hs >> 1;
        if (row < W) v = params[row * in_width + 16u * s + 8u * h + j];
      } else if (e < l1 + nh * hidden) {
        const uint32_t r = e - l1, layer = r / hidden, qq = r % hidden;
        const uint32_t j = qq & 7u, q = qq >> 3, row = q % RW, hs = q / RW, h = hs & 1u, s = hs >> 1;
        const uint32_t k = 16u * s + 8u * (j >> 2) + 4u * h + (j & 3u);
        if (row < W) v = params[first_sz + layer * W * W + row * W + k];
      } else {
        const uint32_t q = e - l1 - nh * hidden;
        const uint32_t j = q & 7u, h = (q >> 3) & 1u, s = q >> 4;
        const uint32_t k = 16u * s + 8u * (j >> 2) + 4u * h + (j & 3u);
        v = params[first_sz + nh * W * W + k];  // row 0 of the 16 x W last layer
      }
      packed[e] = v;
    } else {
      const uint32_t t = e - total;
      const uint32_t rp = ((in_width + 31u) / 32u) * 32u;
      if (t < KS * 16u) {
        const uint32_t j = t & 7u, h = (t >> 3) & 1u, s = t >> 4;
        v = params[first_sz + nh * W * W + 16u * s + 8u * (j >> 2) + 4u * h + (j & 3u)];
      } else if (t < KS * 16u + nh * hidden) {
        const uint32_t r = t - KS * 16u, layer = r / hidden, qq = r % hidden;
        const uint32_t j = qq & 7u, q = qq >> 3, row = q % RW, hs = q / RW, h = hs & 1u, s = hs >> 1;
        const uint32_t kk = 16u * s + 8u * (j >> 2) + 4u * h + (j & 3u);
        if (row < W) v = params[first_sz + layer * W * W + kk * W + row];
      } else {
        const uint32_t qq = t - KS * 16u - nh * hidden;
        const uint32_t j = qq & 7u, q = qq >> 3, row = q % rp, hs = q / rp, h = hs & 1u, s = hs >> 1;
        const uint32_t kk = 16u * s + 8u * (j >> 2) + 4u * h + (j & 3u);
        if (row < in_width) v = params[kk * in_width + row];
      }
      packedT[t] = v;
    }
  }
}

void launch_pack_mlp(const uint16_t* params, uint16_t* packed, uint16_t* packedT, uint32_t in_width, uint32_t W, uint32_t n_hidden_matmuls, hipStream_t s)
{
  const uint32_t total = packed_mlp_halves(in_width, W, n_hidden_matmuls) + (packedT ? packedT_halves(in_width, W, n_hidden_matmuls) : 0u);
  pack_mlp_kernel<<<div_round_up(total, 256), 256, 0, s>>>((const half_t*)params, (half_t*)packed, (half_t*)packedT, in_width, W, n_hidden_matmuls);
}

// ------------------------------------------------------------------------------------------------
// ------------------------------------------------------------------------------------------------ generic kernel
// The models the MFMA kernels do not cover (Network::fast_path): a non-zero quantize_threshold (tcnn_impl_decoder.cu:120; settable only
// through tcnn's own API, a params.json never carries it) and 128-neuron models whose weight image exceeds the LDS of a CU
// (n_hidden_layers >= 6, or 5 with an encoded width of 112 / 128).  Every FullyFusedMLP width (tcnn_impl.cu:315-347 dispatches
// WIDTH 16, 32, 64, 128), every activation, Nearest and Dense / Tiled grids run on the fused kernels since round 4.
// One lane = one sample, plain loops, fp32 accumulation with the layer outputs rounded to fp16 and the activation applied on the
// fp16 value, as the reference's fragments are (tcnn_threadblock.h:83,125): correct for any shape, not a fast path.
struct GenericArgs {
  const LevelInfo* levels;
  uint32_t n_levels, n_active_levels, n_features, interpolation;
  float quantize_threshold;
  const half_t* params;     // tcnn-order blob: MLP weights then grid
  size_t n_mlp;
  uint32_t in_width, width, n_hidden_matmuls, activation, output_activation;
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
      const half_t v = act_forward_f16((half_t)s, a.activation);   // the activation on the fp16 value
      h0[o] = v;
      if (a.acts_out) a.acts_out[(size_t)i * W + o] = v;
    }
    w += (size_t)W * a.in_width;
    half_t* cur = h0; half_t* nxt = h1;
    for (uint32_t layer = 0; layer < a.n_hidden_matmuls; ++layer) {
      for (uint32_t o = 0; o < W; ++o) {
        float s = 0.0f;
        for (uint32_t k = 0; k < W; ++k) s = __builtin_fmaf((float)w[(size_t)o * W + k], (float)cur[k], s);
        const half_t v = act_forward_f16((half_t)s, a.activation);
        nxt[o] = v;
        if (a.acts_out) a.acts_out[((size_t)(layer + 1) * n + i) * W + o] = v;
      }
      w += (size_t)W * W;
      half_t* t = cur; cur = nxt; nxt = t;
    }
    float s = 0.0f;
    for (uint32_t k = 0; k < W; ++k) s = __builtin_fmaf((float)w[k], (float)cur[k], s);
    a.out[out_index] = finish_output<true>(s, a.output_activation);   // the network's output is produced in half precision (tcnn_impl.cu:421-431)
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
  a.n_hidden_matmuls = cfg.n_hidden_layers - 1; a.activation = cfg.activation; a.output_activation = cfg.output_activation;
  a.coords = coords; a.out = out; a.features_out = (half_t*)features_out; a.acts_out = (half_t*)acts_out; a.n_ptr = d_n; a.dest = d_dest;
  a.queue_mode = queue_out_stride ? 1u : 0u; a.out_stride = queue_out_stride; a.n = d_n ? (uint32_t)n_max : (uint32_t)n; a.encode_only = mode == 1 ? 1u : 0u;
  const uint32_t blocks = std::min<uint32_t>(div_round_up(n_max, 128), (uint32_t)Runtime::get().n_cus * 16u);
  generic_infer_kernel<<<blocks, 128, 0, s>>>(a);
  VNR_HIP_CHECK(hipGetLastError());
}

#define VNR_DECL(name) void name(int mode, uint32_t F, uint32_t K_IN, const InferArgs& a, size_t n_max, hipStream_t s)
VNR_DECL(launch_fused_w16); VNR_DECL(launch_fused_w32); VNR_DECL(launch_fused_w64); VNR_DECL(launch_fused_w128);
VNR_DECL(launch_fused_w16g); VNR_DECL(launch_fused_w32g); VNR_DECL(launch_fused_w64g); VNR_DECL(launch_fused_w128g);
#undef VNR_DECL

void launch_fused(int mode, const GridDevice& grid, uint32_t in_width, const FusedMlp& mlp, const LevelInfo* d_levels, const uint16_t* table, size_t table_bytes,
                  const float* coords, float* out, uint16_t* features_out, uint16_t* acts_out, size_t n, const uint32_t* d_n, size_t n_max, hipStream_t s,
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
  a.packed_mlp = (const half_t*)mlp.packed;
  a.coords = coords;
  a.out = out;
  a.features_out = (half_t*)features_out;
  a.acts_out = (half_t*)acts_out;
  a.n_ptr = d_n;
  a.dest = d_dest;
  a.queue_mode = queue_out_stride ? 1u : 0u;
  a.out_stride = queue_out_stride;
  a.n = d_n ? (uint32_t)n_max : (uint32_t)n;
  a.n_hidden_matmuls = mlp.n_hidden_matmuls;
  a.activation = mlp.activation;
  a.output_activation = mlp.output_activation;
  a.lds_halves = mlp.lds_halves;
  if (pack && mode == 0) {
    if (mlp.width == 128u) throw std::runtime_error("internal: the 128-neuron evaluation kernel (8 waves per block) does not take the ray packing prologue");
    a.pack = *pack;
  } else a.pack.n_blocks = 0;
  if (mode == 1) {   // encode only: the kernel has no MLP, one instance serves every width
    if (mlp.general) return dispatch<64, 1, true>(grid.n_features, in_width, a, n_max, s);
    return dispatch<64, 1, false>(grid.n_features, in_width, a, n_max, s);
  }
  const uint32_t F = grid.n_features;
  switch (mlp.width) {
  case 16: return mlp.general ? launch_fused_w16g(mode, F, in_width, a, n_max, s) : launch_fused_w16(mode, F, in_width, a, n_max, s);
  case 32: return mlp.general ? launch_fused_w32g(mode, F, in_width, a, n_max, s) : launch_fused_w32(mode, F, in_width, a, n_max, s);
  case 64: return mlp.general ? launch_fused_w64g(mode, F, in_width, a, n_max, s) : launch_fused_w64(mode, F, in_width, a, n_max, s);
  case 128: return mlp.general ? launch_fused_w128g(mode, F, in_width, a, n_max, s) : launch_fused_w128(mode, F, in_width, a, n_max, s);
  default: throw std::runtime_error("internal: no MFMA kernel for n_neurons = " + std::to_string(mlp.width));
  }
}

}  // namespace vnr
