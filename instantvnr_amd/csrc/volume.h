// volume.h — ground-truth volume + sampler, macrocells, transfer function and the neural volume object.
//
// Replaces core/sampler.{h,cu} + core/samplers/neural_sampler.{cpp,cu} (StaticSampler), core/macrocell.{h,cu},
// TransferFunctionObject (core/instantvnr_types.h:207-213) and NeuralVolume (core/network.{h,cu}).
// No textures on CDNA (imageSupport = 0): volumes and TFN tables are linear buffers with hand-written
// (exact fp32) linear filtering.
#pragma once

#include <memory>

#include "common.h"
#include "json.h"
#include "network.h"
#include "ooc_sampler.h"

namespace vnr { struct SceneVolume; }

namespace vnr {

constexpr int kMacrocellSizeMip = 4;  // CMakeLists.txt:61 MACROCELL_SIZE_MIP
constexpr int kMacrocellSize = 1 << kMacrocellSizeMip;

struct TransferFunctionData {  // instantvnr_types.h:66-71
  std::vector<vec3f> color;
  std::vector<vec2f> alpha;
  float range_lo = 0.0f, range_hi = 1.0f;
  bool range_set = false;
};

struct DeviceTfn {  // instantvnr_types.h:89-94 (arrays instead of textures)
  const vec4f* colors;
  const float* alphas;
  int n_colors, n_alphas;
  float range_lo, range_hi, range_rcp_norm;
};

// device-resident transfer function (object.cpp:321-348 set_transfer_function)
class TfnObject {
public:
  void set(const TransferFunctionData& t, float data_lo, float data_hi, hipStream_t s);
  DeviceTfn view() const;
  bool empty() const { return n_alphas_ == 0; }

private:
  DeviceBuffer<vec4f> colors_;
  DeviceBuffer<float> alphas_;
  int n_colors_ = 0, n_alphas_ = 0;
  float lo_ = 0.0f, hi_ = 1.0f, rcp_ = 1.0f;
};

class MacroCell {  // core/macrocell.h:7-38
public:
  void set_shape(vec3i volume_dims);                    // macrocell.cu:195-201
  void set_dims(vec3i d) { dims_ = d; }
  void set_spacings(vec3f s) { spacings_ = s; }
  void set_external(MacroCell* ext) { external_ = ext; }  // macrocell.cu:203-211
  void allocate(hipStream_t s);                          // macrocell.cu:213-219 (value ranges zero-initialised)
  bool is_external() const { return external_ != nullptr; }
  bool allocated() const { return target().value_range_.count > 0; }

  vec3i dims() const { return target().dims_; }
  vec3f spacings() const { return target().spacings_; }
  vec3i volume_dims() const { return target().volume_dims_; }
  float* d_value_range() const { return target().value_range_.ptr; }  // vec2f per cell: (min-1, max+1)
  float* d_max_opacity() const { return target().max_opacity_.ptr; }
  size_t n_cells() const { const vec3i d = dims(); return (size_t)d.x * d.y * d.z; }

  void compute_everything(const float* d_volume, hipStream_t s);                            // macrocell.cu:221-234
  void update_explicit(const float* d_coords, const float* d_values, size_t n, hipStream_t s);  // :236-241
  void update_max_opacity(const DeviceTfn& tfn, hipStream_t s);                             // :243-253
  void upload_value_range(const void* host, size_t bytes, hipStream_t s);

private:
  const MacroCell& target() const { return external_ ? *external_ : *this; }
  MacroCell& target() { return external_ ? *external_ : *this; }
  vec3i volume_dims_{0, 0, 0}, dims_{0, 0, 0};
  vec3f spacings_{1, 1, 1};
  DeviceBuffer<float> value_range_, max_opacity_;
  MacroCell* external_ = nullptr;
};

struct VolumeDesc {  // MultiVolume, instantvnr_types.h:40-56 (single timestep)
  vec3i dims{0, 0, 0};
  int type = 8;  // VALUE_TYPE_FLOAT
  float range_lo = 0.0f, range_hi = 1.0f;
};

class VolumeBase {  // VolumeContext (api_internal.h:15-22) + VolumeObject (instantvnr_types.h:215-228)
public:
  virtual ~VolumeBase() = default;
  virtual bool is_network() const = 0;
  virtual MacroCell& macrocell() = 0;
  virtual void set_transfer_function(const TransferFunctionData& t, hipStream_t s) = 0;
  VolumeDesc desc;
  box3f clipbox{{0, 0, 0}, {1, 1, 1}};
  affine3f transform{{1, 0, 0}, {0, 1, 0}, {0, 0, 1}, {0, 0, 0}};
};

// Ground-truth volume with the GPU ("StaticSampler") training sampler.
class SimpleVolume : public VolumeBase {
public:
  bool is_network() const override { return false; }
  MacroCell& macrocell() override { return mc_; }
  void set_transfer_function(const TransferFunctionData& t, hipStream_t s) override;

  // neural_sampler.cpp:223-288: load + min/max normalise to [0,1] fp32; range_lo > range_hi => from the data
  void load_host(const void* data, vec3i dims, int type, float range_lo, float range_hi, bool big_endian = false);
  void load_raw_file(const std::string& filename, vec3i dims, int type, size_t offset, bool big_endian, float range_lo, float range_hi);
  void generate_perlin(vec3i dims, uint32_t seed, int octaves, float base_frequency);
  // vnrCreateSimpleVolume(scene, "OUT_OF_CORE") (neural_sampler.cpp:1224-1227, 1043-1064): no resident ground truth and no
  // macrocell; dims() = min(1024, file dims) is the shape a neural volume trained from it takes, the transform is the file's
  void load_out_of_core(const std::string& filename, vec3i dims, int type, size_t offset, float range_lo, float range_hi,
                        uint64_t n_concurrent_blocks, uint64_t n_blocks);
  // vnrCreateSimpleVolume(scene, mode, save) (api.cpp:145-158): "GPU", "OUT_OF_CORE" or "NOTHING"; one file per time step
  void load_scene(const SceneVolume& scene, const std::string& mode, bool save_volume);
  int num_timesteps() const { return steps_.empty() ? 1 : (int)steps_.size(); }   // SimpleVolume::get_num_timesteps
  void set_current_timestep(int index);                                           // core/sampler.cu:19-26
  bool has_data() const { return data_.ptr != nullptr; }   // SimpleVolume::texture() != 0
  OutOfCoreSampler* out_of_core() { return ooc_.get(); }

  const float* d_data() const { return data_.ptr; }
  vec3i dims() const { return desc.dims; }
  float unnormalized_lo = 0.0f, unnormalized_hi = 1.0f;

  // neural_sampler.cu:130-164 StaticSampler::sample
  void take_samples(float* d_coords, float* d_values, size_t n, vec3f lower, vec3f upper, hipStream_t s);
  // neural_sampler.cu:166-198 sample_grid: voxel-centre coords of a block + values
  void take_samples_grid(float* d_coords, float* d_values, vec3i origin, vec3i size, vec3f rdims, hipStream_t s);
  void sample(const float* d_coords, float* d_values, size_t n, bool nodal, hipStream_t s) const;
  void set_sampler_seed(uint64_t seed, uint64_t stream_id) { rng_seed_ = seed; rng_stream_ = stream_id; rng_offset_ = 0; rng_user_set_ = true; }
  // data-parallel training: rank r draws pcg32 stream (default stream + r) unless the application chose a stream itself
  void set_sampler_rank(int rank) { if (!rng_user_set_) { rng_stream_ = 0xda3e39cb94b95bdbULL + (uint64_t)rank; rng_offset_ = 0; } }

private:
  void finish_load(hipStream_t s);
  DeviceBuffer<float> data_;                 // the current time step
  std::vector<DeviceBuffer<float>> steps_;   // the other time steps (entry `current_step_` is moved into data_)
  int current_step_ = 0;
  std::unique_ptr<OutOfCoreSampler> ooc_;
  MacroCell mc_;
  TfnObject tfn_;
  uint64_t rng_seed_ = 1337, rng_stream_ = 0xda3e39cb94b95bdbULL, rng_offset_ = 0;  // neural_sampler.cu:36
  bool rng_user_set_ = false;
};

class NeuralVolume : public VolumeBase {  // core/network.h:29-107, core/network.cu:143-699
public:
  NeuralVolume();
  ~NeuralVolume() override;
  bool is_network() const override { return true; }
  MacroCell& macrocell() override { return mc_; }
  void set_transfer_function(const TransferFunctionData& t, hipStream_t s) override;  // network.cu:743-760

  // network.cu:551-621
  void set_network(vec3i dims, const Json& config, SimpleVolume* reference, bool use_reference_macrocell);
  void set_model(const Json& config);                       // network.cu:737-741
  void load_params_from_json(const Json& root);             // network.cu:879-939
  void save_params_to_json(Json& root);                     // network.cu:827-857
  void train(size_t steps, bool fast_mode);                 // network.cu:769-779 + Impl::train :231-259
  // Data-parallel training over the ranks of Dist (dist.h; new work): every rank samples its own batch and runs
  // forward + backward; the gradient travels as fp16 (tcnn's own gradient precision), range by range while the backward pass of
  // the coarser levels and the optimizer update of the ranges before it run; no host synchronisation inside a step on the RCCL
  // transport.  One step equals one step on the concatenated batch (sum of the ranks' gradients / world).  The first call makes
  // the replicas identical (parameters, optimizer state, step count and learning rate of rank 0) and gives every rank its own
  // sample stream; an online macrocell is merged over the ranks at the end of the call (min / max per cell).
  void train_data_parallel(size_t steps, bool fast_mode);
  void all_reduce_gradients();                              // the TrainBegin / TrainEnd form: grads() summed over the ranks
  // ... or in one call: exchange (mean over the ranks) + update in the sharded (1) or replicated (0) shape; -1: the default shape
  void train_end_data_parallel(bool fast_mode, int sharded);
  void sync_replicas();                                     // collective: rank 0's training state to everyone, or (sharded optimizer) the ranks' slices of it to everyone
  void train_begin();                                       // sample + forward + backward
  void train_end(float grad_scale, bool fast_mode);         // optimizer + macrocell update
  void forward_backward(const float* d_coords, const float* d_targets, size_t n);  // caller-provided batch
  void mark_external_gradient() { pending_step_ = true; pending_internal_ = false; }   // the gradient blob was set by the caller (tests)
  float test_loss();                                        // network.cu:261-288
  float get_psnr(bool quiet);                               // network.cu:410-472
  float get_ssim(bool quiet);                               // network.cu:474-549 (get_mssim: mean SSIM, 7^3 uniform windows)
  void inference(size_t n, const float* d_in, float* d_out, hipStream_t s);  // network.cu:1043-1052
  int num_blobs() const { return (desc.dims.z + 15) / 16; } // network.cu:969-975
  // Decoding (rendering modes 4 / 7 on a neural volume march the DECODED dense volume).  One call decodes one blob of 16
  // z-slices at the voxel centres and moves on to the next, wrapping around (network.cu:290-326; num_blobs() calls = one
  // full pass; the blob cursor is per volume here, a function-local static in the reference).
  void decode_progressive();
  const float* decoded_data() const { return decoded_.count ? decoded_.ptr : nullptr; }
  // network.cu:328-365 / :367-405: raw fp32, z-slice by z-slice, every slice padded to a multiple of 256 values
  void save_inference_volume(const std::string& filename);
  void save_reference_volume(const std::string& filename);

  Network& network() { return net_; }
  SimpleVolume* source() { return source_; }
  uint64_t init_seed = 0;  // 0 => time(NULL) like the reference (tcnn_network.h:209)
  hipStream_t stream;

private:
  SimpleVolume* source_ = nullptr;
  std::shared_ptr<void> source_keepalive_;
  Network net_;
  MacroCell mc_;
  TfnObject tfn_;
  const size_t batch_size_ = 1u << 16;  // network.cu:183
  DeviceBuffer<float> train_x_{MemTag::Network}, train_y_{MemTag::Network}, test_y1_{MemTag::Network};
  DeviceBuffer<float> decoded_{MemTag::Network}, decode_coords_{MemTag::Network};  // dense decoded volume, coordinates of one blob
  int decode_blob_ = 0;
  bool pending_step_ = false, pending_internal_ = false;
  bool replicas_synced_ = false;
  uint64_t synced_generation_ = 0;   // Network::params_generation() at the last sync_replicas
  struct DpState;
  std::unique_ptr<DpState> dp_;
  DpState& dp_state();
  friend struct VolumeKeepAlive;

public:
  void keep_source_alive(std::shared_ptr<void> p) { source_keepalive_ = std::move(p); }
};

// marching_cubes.hip: vnrMarchingCube / vnrSaveTriangles (core/marching_cube.cuh:6-8)
size_t marching_cubes(VolumeBase& volume, float isovalue, DeviceBuffer<vec3f>& vertices);
void save_triangles_obj(const std::string& filename, const float* xyz, size_t n_vertices);

}  // namespace vnr
