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
// ------------------------------------------------------------------------------------------------
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
  a.weights_global = mlp.weights_global ? 1u : 0u;
  a.quantize_threshold = mlp.quantize_threshold;
  if (mlp.weights_global && !mlp.general) throw std::runtime_error("internal: weights from global memory are a GENERAL instance's");
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
