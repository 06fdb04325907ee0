// network.h — hash-grid encoding + fully fused MLP ("NetworkWithInputEncoding") for gfx950.
//
// Replaces the reference's network engine: core/networks/tcnn_network.h:79-272 (AbstractNetwork /
// TcnnNetwork<3,1>), core/networks/tcnn_impl*.cu (inference restatement) and the un-vendored
// tiny-cuda-nn Trainer behind them (forward/backward/L1 loss/Adam+ExponentialDecay).
#pragma once

#include <functional>
#include <utility>
#include <vector>

#include "common.h"
#include "json.h"

namespace vnr {

constexpr int kMaxLevels = 32;
constexpr size_t kLdsBytes = 160 * 1024;   // LDS of a CU: what a weight image may take
constexpr int kLossScale = 128;   // tcnn default loss scale for fp16 (EXTERNAL)

struct LevelInfo {
  float scale;
  uint32_t resolution;
  uint32_t res2;    // resolution^2 (dense index stride of z)
  uint32_t size;    // entries in this level
  uint32_t offset;  // first entry of this level (entries, x F for elements)
  uint32_t hashed;  // 0: dense index, 1: prime-XOR hash (size is a power of two), 2 / 3 / 4: Tiled grid, index over 1 / 2 / 3 dimensions modulo size
  uint32_t brick;   // 0: read the parameter blob; else 1 + first 128-byte line of this level in the brick image (below)
  uint32_t pad1;    // bricked levels with F = 2: bricks per row of the image (resolution / 7 + 1); else 0.  32 bytes: one s_load_dwordx8 per level
};

// Brick image (inference only): a level whose table is hashed is ALSO kept de-hashed, as a dense array over the level's
// (res + 1)^3 grid points stored in bricks of one 128-byte line: 4x4x4 entries for F = 1, 4x2x2 for F = 4, 2x2x2 for F = 8, and
// for F = 2 8x2x2 entries of which the 8th column repeats the +x neighbour's first (grid_device.h: a cell's x-pairs never straddle).  The values are copies (image[x, y, z] = table[hash(x, y, z)]), so results are bit-identical; what changes
// is which lines a wave touches: the 8 corners of a cell fall into ~2.3 lines instead of 4-8 and neighbouring samples share
// them, where the hash scatters every (y, z) row to an unrelated line (measured on the bench frame's sample queue: 917 -> 385
// fetched bytes per sample inside a 64-sample wave).  The price is memory, 8.8 GB instead of 140 MB for the bench model,
// which is what 288 GB of HBM are for, and a rebuild (a few ms) after the parameters change, so the image is only built once
// the parameters have been left alone for a while (Network::brick_policy).
template <int F> struct BrickShape;
template <> struct BrickShape<1> { static constexpr uint32_t lx = 2, ly = 2, lz = 2; };
template <> struct BrickShape<2> { static constexpr uint32_t lx = 2, ly = 2, lz = 1; };   // (superseded for the image itself: F = 2 uses 8 x 2 x 2 bricks
                                                                                          // with a repeated column, grid_device.h gather_corners_brick; kept for the sizing code's defaults)
template <> struct BrickShape<4> { static constexpr uint32_t lx = 2, ly = 1, lz = 1; };
template <> struct BrickShape<8> { static constexpr uint32_t lx = 1, ly = 1, lz = 1; };

struct GridDevice {
  LevelInfo levels[kMaxLevels];
  uint32_t n_levels;
  uint32_t n_features;     // per level
  uint32_t interpolation;  // 0 linear, 1 smoothstep, 2 nearest
};

struct ModelConfig {
  // encoding (tcnn HashGrid; example-model.json:19-25)
  uint32_t n_levels = 8, n_features = 8, log2_hashmap_size = 19, base_resolution = 16;
  float per_level_scale = 2.0f;
  uint32_t interpolation = 0;          // 0 Linear, 1 Smoothstep, 2 Nearest (tcnn_impl_decoder.cu:73-94)
  float quantize_threshold = 0.0f;     // corner values below it in magnitude count as 0 (tcnn_impl_decoder.cu:120); tcnn default 0
  float max_level = 1000.0f;           // levels l >= max_level + 1e-3 encode to 0 (tcnn_impl_decoder.cu:17); tcnn default: no masking
  // network (FullyFusedMLP; example-model.json:26-32)
  uint32_t grid_type = 0;              // 0 Hash, 1 Dense, 2 Tiled (tcnn GridType; tcnn_impl_decoder.cu:68-69 passes it to grid_index)
  uint32_t n_neurons = 64, n_hidden_layers = 4;
  uint32_t activation = 1;         // 0 None, 1 ReLU, 2 Exponential, 3 Sigmoid, 4 Squareplus, 5 Softplus (infer_tile.h kAct*; tcnn_impl.cu:405-415)
  uint32_t output_activation = 0;  // the same set, applied to the fp16 output (tcnn_threadblock.h:497)
  // optimizer (ExponentialDecay{Adam}; example-model.json:2-15)
  float learning_rate = 5e-3f, beta1 = 0.9f, beta2 = 0.999f, epsilon = 1e-15f, l2_reg = 1e-6f;
  uint32_t decay_start = 2000, decay_interval = 1000;
  float decay_base = 0.99f;
  bool has_decay = true;
  // loss
  uint32_t loss = 0;  // 0 L1, 1 L2
};

// Optimizer state of ONE parameter, interleaved: an Adam update of a touched hash-grid entry then reads and writes one
// 16-byte record (entries of a level are adjacent, so a 128-B line holds 8 of them) instead of one element in each of
// four arrays, i.e. four lines (the ~10 M touched parameters of a step cost 3x the sweep over all 70 M).
struct alignas(16) OptState {
  float master;   // fp32 master weight
  float m, v;     // Adam first / second moment
  uint32_t step;  // per-parameter step count (tcnn adam: untouched grid entries do not advance)
};
static_assert(sizeof(OptState) == 16, "OptState must be one 16-byte record");

// Data-parallel hook (dist.h): forward_backward calls it in stream order whenever this rank's fp32 gradient of a
// contiguous parameter range is complete (the MLP first, then the hash-grid levels from the finest to the coarsest in buckets),
// so the exchange of one range runs while the backward pass of the next still does.
struct TileNet;
struct PackArgs;   // infer_tile.h: what a kernel needs to evaluate the network itself
struct FusedMlp;   // infer_kernel.h: weight image + shape of a launch

struct GradExchange {
  virtual ~GradExchange() = default;
  virtual void range_ready(size_t lo, size_t hi, hipStream_t s) = 0;
  // hash-grid levels are handed over in buckets of at least this many parameters (finest first)
  virtual size_t bucket_params() const { return (size_t)16u << 20; }   // 32 MB of fp16 per message: 4 grid buckets at C4 (10 at 4 M cost 0.03 ms more per step in launches on one rank)
};

// how the grid backward of a training step is laid out (Network::grid_backward_plan): the first `lds_levels` levels (dense, few tiles) go through
// LDS tiles of `tile_entries` entries in ~`lds_blocks` blocks per level, the rest through the global-atomic kernel; and the step's memory-side
// atomic requests by that layout, the unit its roofline is priced in (bench.py train_roofline)
struct GridBackwardPlan {
  uint32_t n_levels, lds_levels, tile_entries, lds_blocks;
  uint64_t atomic_requests, flush_requests_at_most;
};

class Network {
public:
  Network() = default;
  // tcnn_network.h:163-221 deserialize_model: builds loss/optimizer/encoding/network from the model JSON.
  // Throws on unsupported otypes (the reference swallows the tcnn error and stays invalid, :211-213).
  void configure(const Json& model, uint64_t init_seed);
  bool valid() const { return n_params_ > 0; }

  const ModelConfig& config() const { return cfg_; }
  const Json& model_json() const { return model_; }
  const GridDevice& grid() const { return grid_; }
  uint32_t padded_width() const { return in_width_; }
  uint32_t width() const { return cfg_.n_neurons; }
  uint32_t lds_halves() const { return lds_halves_; }   // halves of the forward weight image
  // Every model the reference's dispatch builds runs on the MFMA kernels (round 4): widths 16 / 32 / 64 / 128 (tcnn_impl.cu:315-347), every
  // interpolation, activation and grid type.  The common kind of model (Hash / Dense grid, Linear / Smoothstep, ReLU / None, no output
  // activation, no quantize_threshold, a weight image that fits the LDS) has kernel instances that contain nothing else
  // (grid_device.h gather_corners); everything else runs on the GENERAL instances.
  bool weights_in_lds() const { return (size_t)lds_halves_ * 2 <= kLdsBytes; }   // false: 128 neurons with >= 6 hidden layers (or 5 and an encoded width >= 112), 64 with ~20, 32 with ~75, 16 with ~300
  bool common_kind() const
  {
    return cfg_.activation <= 1u && cfg_.output_activation == 0u && cfg_.interpolation != 2u && cfg_.grid_type != 2u && cfg_.quantize_threshold == 0.0f &&
           weights_in_lds();
  }
  uint32_t n_active_levels() const;    // levels below max_level + 1e-3
  uint32_t n_hidden_matmuls() const { return cfg_.n_hidden_layers - 1; }
  size_t n_params() const { return n_params_; }
  size_t n_mlp_params() const { return n_mlp_; }
  size_t n_grid_params() const { return n_params_ - n_mlp_; }
  size_t model_size_bytes() const { return n_params_ * sizeof(uint16_t); }  // tcnn_network.h:138
  uint64_t steps() const { return steps_; }

  // parameters, tcnn order: MLP weights (first, hidden..., last[16 x W]), then grid level by level; fp16
  void set_params_f16(const uint16_t* host, size_t count, hipStream_t s);
  void get_params_f16(uint16_t* host, size_t count, hipStream_t s) const;
  // tcnn Trainer::serialize / deserialize ({n_params, params_type, params_binary}); EXTERNAL format
  Json serialize_params(hipStream_t s) const;
  void deserialize_params(const Json& j, hipStream_t s);

  // inference: coords [n][3] fp32 -> out [n] fp32.  n either by value or read on the device from d_n.
  // d_dest (optional): result of sample i is written to d_out[d_dest[i]] (the ray marcher's gather order -> result slot map).
  void inference(const float* d_coords, float* d_out, size_t n, const uint32_t* d_n, size_t n_max, hipStream_t s,
                 const uint32_t* d_dest = nullptr) const;
  // ray marcher's sample queue: records {x, y, z, dest} (16 B); result of a record goes to d_out[dest * out_stride]
  // `sharers`: streams that run such launches side by side (sizes the persistent grid, network_infer.hip)
  // for a kernel that evaluates the network inside its own loop (the in-shader ray marcher, in_shader.h): levels (brick image
  // policy applied, as for an inference launch on `s`), table, weight image.  false: this model is not one of the MFMA kernels' shapes
  bool tile_net(TileNet* out, hipStream_t s) const;
  // `pack` (pack_rays.h): the ray marcher's packing of the iteration, run as a prologue of the evaluation kernel; -> false when this
  // model's kernel cannot take it (the caller then launches the packing kernel itself)
  bool inference_queue(const float* d_records, float* d_out, uint32_t out_stride, const uint32_t* d_n, size_t n_max, hipStream_t s,
                       uint32_t sharers = 1, const PackArgs* pack = nullptr) const;
  // encode only: fp16 [n][padded_width]
  void encode(const float* d_coords, uint16_t* d_features, size_t n, hipStream_t s) const;

  // training step pieces (tcnn Trainer::training_step, EXTERNAL): forward+loss+backward into grads()
  void forward_backward(const float* d_coords, const float* d_targets, size_t batch, hipStream_t s, GradExchange* exchange = nullptr);
  // The gradient of the whole blob, loss-scaled (x 128), in HALF precision like tcnn's (its gradient matrices and, for F > 1, its
  // grid gradients are network_precision_t): the unit of the data-parallel exchange, read and cleared by the optimizer.
  uint16_t* grads_f16() { return grads_.ptr; }
  size_t grads_count() const { return grads_.count ? n_params_ : 0; }
  size_t grads_alloc() const { return n_params_ + (n_params_ & 1); }   // halves allocated: the packed atomics add aligned pairs
  float* grads_as_f32(hipStream_t s);   // a float copy for inspection (vnrAmdNeuralVolumeGradients); not an input of anything
  void optimizer_step(float grad_scale, hipStream_t s);
  // the same step in pieces, for gradients that become final range by range (data-parallel exchange, volume.hip):
  // Adam on parameters [lo, hi) with gradient grads[i] * grad_scale / loss_scale (clearing it), then the bookkeeping of ONE step
  void optimizer_step_range(size_t lo, size_t hi, float grad_scale, hipStream_t s);
  void optimizer_finish_step(hipStream_t s);
  float learning_rate() const { return lr_; }
  // replica state for data-parallel training: everything an optimizer step reads
  uint16_t* params_device() { return params_f16_.ptr; }
  OptState* opt_state_device() { return opt_state_.ptr; }
  void set_replica_state(uint64_t steps, float lr, hipStream_t s) { steps_ = steps; lr_ = lr; refresh_inference_weights(s); }
  void ensure_training_state(hipStream_t s);
  void reset_master_from_params(hipStream_t s);   // master weights <- fp16 parameters (moments and step counts kept)
  double training_loss(hipStream_t s);  // mean loss of the last forward_backward
  // counts the parameter changes that do NOT come from an optimizer step (configure, set_params_f16, deserialize_params): replicas of a
  // data-parallel run whose counters differ from what they were at the last synchronisation are synchronised again (volume.hip)
  uint64_t params_generation() const { return params_generation_; }
  // Sharded optimizer (volume.hip train_data_parallel): a rank updates master weights and moments of ITS slices only and receives the
  // other ranks' fp16 parameters, so its optimizer state of the other slices is stale until NeuralVolume::sync_replicas gathers it;
  // a full optimizer step on such a state would be wrong and throws
  void set_opt_sharded(bool e) { opt_sharded_ = e; }
  bool opt_sharded() const { return opt_sharded_; }
  // the parameter ranges a data-parallel step exchanges, in the order forward_backward hands them over: the MLP, then the hash-grid
  // levels from the finest to the coarsest in buckets of at least `bucket` parameters (levels [first, second) each)
  std::vector<std::pair<uint32_t, uint32_t>> exchange_level_buckets(size_t bucket) const;
  void for_each_exchange_range(size_t bucket, const std::function<void(size_t, size_t)>& fn) const;
  size_t level_range_lo(uint32_t level) const { return n_mlp_ + (size_t)grid_.levels[level].offset * cfg_.n_features; }
  size_t level_range_hi(uint32_t level_end) const { return n_mlp_ + ((size_t)grid_.levels[level_end - 1].offset + grid_.levels[level_end - 1].size) * cfg_.n_features; }
  // Diagnostics of the training step (tests/diag/grad_hammer.py; passive: nothing else depends on them).  training_buffer: device
  // pointer and size of 0 the fp16 gradient blob, 1 dL/dfeatures [n][padded_width] fp16, 2 the features, 3 the hidden activations
  // of the last forward_backward.  rescatter_grid_gradients: clears the grid part of the blob and repeats step 5 alone on the stored
  // dL/dfeatures.  gradient_distance: {sum (g - ref)^2, sum ref^2} of the MLP part and of the grid part against an fp16 reference blob,
  // reduced on the device on stream s (what the blob holds BEFORE any download).
  const void* training_buffer(int which, size_t* bytes) const;
  void rescatter_grid_gradients(const float* d_coords, size_t n, hipStream_t s);
  void gradient_distance(const uint16_t* d_ref, double out[4], hipStream_t s);
  void scatter_grid_gradients(const float* d_coords, size_t batch, hipStream_t s, GradExchange* exchange, hipStream_t s_lds = nullptr);   // step 5 of forward_backward
  // tests: the gradient blob from a float array (rounded to the blob's half precision)
  void set_grads_from_f32(const float* host, size_t count, hipStream_t s);

  size_t bytes_allocated() const;
  // vnrFreeTemporaryGPUMemory (api.cpp:554-557 -> tcnn's free_all_gpu_memory_arenas): drops what can be rebuilt on demand,
  // the brick image and the training workspace (not parameters, optimizer state or gradients)
  void release_temporary();
  static void release_temporary_of_all();

  // brick image policy: built on stream `s` once `brick_after` inference launches have seen unchanged parameters
  // (VNR_AMD_BRICK_AFTER, default 24 = two frames of the streaming renderer; VNR_AMD_BRICK=0 disables, =1 builds at the first
  // launch), within VNR_AMD_BRICK_MAX_GB (default 32) and a quarter of the free device memory
  // Levels finer than `cap` grid points per axis are left hashed: the image pays off where neighbouring samples share cells,
  // i.e. up to about the resolution of the volume the network represents (NeuralVolume passes twice its largest dimension).
  // A 128^3 volume with tcnn's default per_level_scale 2 has levels up to 2048^3, whose image would be 137 GB and useless.
  void set_brick_resolution_cap(uint32_t cap) { brick_res_cap_ = cap; }
  // -1: the environment's policy (VNR_AMD_BRICK, default automatic), 0: never (drops an existing image), 1: build at the next launch
  void set_brick_mode(int mode);
  // budget of the image in bytes (0: the default policy, network_host.hip build_brick_image); drops an existing image, which is rebuilt
  // within the new budget by the next launches.  Levels are taken finest first while they fit.
  void set_brick_budget(size_t bytes);
  GridBackwardPlan grid_backward_plan(size_t batch) const;
  uint32_t brick_levels_mask() const { return brick_levels_mask_; }   // bit l: level l is read from the image
  bool brick_image_in_use() const { return brick_valid_; }
  size_t brick_image_bytes() const { return brick_image_.bytes(); }
  float brick_build_ms() const { return brick_build_ms_; }
  uint64_t brick_builds() const { return brick_builds_; }              // how many times the (full) image has been built
  uint64_t brick_small_builds() const { return brick_small_builds_; }  // ... and its small tier
  int brick_tier() const { return brick_valid_ ? brick_tier_ : 0; }    // 0 none, 1 small, 2 full
  uint32_t brick_after_now() const;                                    // launches with unchanged parameters the next build waits for

private:
  void build_layout();
  void initialize_params(uint64_t seed, hipStream_t s);
  void refresh_inference_weights(hipStream_t s);
  FusedMlp fused_mlp() const;
  const LevelInfo* inference_levels(hipStream_t s, const uint8_t** image, size_t n_max) const;  // decides / builds / orders streams
  void build_brick_image(hipStream_t s, bool small) const;
  static double brick_small_budget();

  ModelConfig cfg_;
  Json model_;
  GridDevice grid_{};
  uint32_t in_width_ = 0;
  size_t n_params_ = 0, n_mlp_ = 0;
  uint64_t steps_ = 0;
  uint64_t params_generation_ = 0;
  bool opt_sharded_ = false;
  float lr_ = 0.0f;   // current learning rate (ExponentialDecay state); reset by configure() like the reference's rebuilt optimizer (tcnn_network.h:195-209)

  DeviceBuffer<uint16_t> params_f16_{MemTag::Network};   // tcnn-order blob (inference + serialisation)
  DeviceBuffer<uint16_t> mlp_packed_{MemTag::Network};   // MFMA/LDS image of the MLP weights (forward)
  DeviceBuffer<uint16_t> mlp_packed_T_{MemTag::Network}; // ... transposed, for the MLP backward
  DeviceBuffer<LevelInfo> levels_dev_{MemTag::Network};  // per-level constants, read with scalar loads
  DeviceBuffer<OptState> opt_state_{MemTag::Network};    // per parameter: fp32 master copy + Adam moments + step count (training)
  DeviceBuffer<uint16_t> grads_{MemTag::Network};        // fp16 gradient of the whole blob (ONE buffer: the all-reduce unit)
  DeviceBuffer<float> grads_f32_{MemTag::Network};       // grads_as_f32()
  // training workspace
  DeviceBuffer<uint16_t> ws_features_{MemTag::Network};  // [B][in_width]
  DeviceBuffer<uint16_t> ws_acts_{MemTag::Network};      // [(nh+1)][B][n_neurons]
  DeviceBuffer<uint16_t> ws_dfeat_{MemTag::Network};     // [B][in_width] dL/dfeatures (fp16, loss-scaled)
  DeviceBuffer<float> ws_loss_{MemTag::Network};         // [blocks] partial loss sums
  size_t ws_batch_ = 0;
  // training step: weight gradients + the dense levels' LDS scatter beside the atomic scatter (network_train.hip forward_backward)
  hipStream_t side_stream_ = nullptr;
  hipEvent_t ev_fork_ = nullptr, ev_join_ = nullptr;
  uint32_t lds_halves_ = 0, lds_halves_T_ = 0;
  // brick image (inference cache; mutable: built lazily from const inference calls)
  mutable DeviceBuffer<uint8_t> brick_image_{MemTag::Network};
  mutable DeviceBuffer<LevelInfo> levels_brick_dev_{MemTag::Network};
  mutable bool brick_valid_ = false, brick_refused_ = false;
  mutable uint32_t brick_stable_calls_ = 0;
  // An application that trains while it renders (apps/int_dual_volume.cpp:631-672) changes the parameters after every frame.  An image
  // built between two optimizer steps costs more than it saves (6.5 ms for the C4 model against 0.1 ms saved per launch), and whether the
  // base threshold is reached inside one frame depends on the frame (ray parts x iterations: 12 launches for the bench frame, 27 for a
  // three-part share of it).  So the threshold backs off: an image dropped before it served `kBrickPaysAfter` launches doubles it, an
  // image that lived longer resets it.
  static constexpr uint32_t kBrickPaysAfter = 64;
  mutable uint32_t brick_served_calls_ = 0, brick_after_scale_ = 1;
  mutable uint64_t brick_builds_ = 0;
  // Two tiers (round 6).  FULL: the policy above.  SMALL: while the parameters keep changing, the first large evaluation launch after a change
  // (capacity >= kBrickSmallMinLaunch samples: a frame, not a probe) builds the finest-first levels that fit brick_small_budget(), 0.25 ms for
  // the bench model's 0.7 GB, which the frame's ~12 launches earn back: the reference application's loop 190 -> 200 frames/s.  A small image
  // never counts towards the backoff above, and it is replaced by the full one once the parameters have been left alone.
  static constexpr size_t kBrickSmallMinLaunch = (size_t)1 << 20;
  mutable int brick_tier_ = 0;   // 0 none, 1 small, 2 full
  mutable bool brick_small_refused_ = false;   // no level fits the small budget
  mutable uint64_t brick_small_builds_ = 0;
  mutable float brick_build_ms_ = 0.0f;
  uint32_t brick_res_cap_ = 0;   // 0: no cap
  size_t brick_budget_ = 0;      // 0: default policy
  mutable uint32_t brick_levels_mask_ = 0;
  int brick_mode_ = -1;

public:
  // Per-kernel time of the training step (HIP events on the training stream, a ring of the last kTrainProfileSteps steps):
  // phases = {forward, loss + MLP backward, weight gradients, grid backward, optimizer}.  Off by default (no events recorded).
  static constexpr int kTrainPhases = 5, kTrainProfileSteps = 64;
  void set_train_profiling(bool e);
  int train_profile(double* ms_per_step /* [kTrainPhases] */);   // returns the number of steps averaged over (0: nothing recorded)
private:
  void profile_mark(int phase_boundary, hipStream_t s);
  bool train_profiling_ = false;
  std::vector<hipEvent_t> prof_events_;   // [kTrainProfileSteps][kTrainPhases + 1]
  uint64_t prof_steps_ = 0;

public:
  ~Network();
  Network(const Network&) = delete;
  Network& operator=(const Network&) = delete;
};

// layout helpers shared with tests
uint32_t grid_make_layout(const ModelConfig& cfg, GridDevice* out);

}  // namespace vnr
