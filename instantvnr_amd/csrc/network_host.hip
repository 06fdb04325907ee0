// network_host.hip — host side of the network engine: model JSON -> layout, parameter blob handling,
// (de)serialisation in tcnn's Trainer::serialize format, launch wrappers.
//
// Reference: core/networks/tcnn_network.h:157-272; EXTERNAL tiny-cuda-nn (GridEncodingTemplated ctor,
// FullyFusedMLP, Trainer::serialize/deserialize) — assumptions listed in SURVEY.md Appendix A.
#include "network.h"

#include <algorithm>
#include <cstdlib>
#include <set>

#include "infer_kernel.h"

namespace vnr {

void launch_pack_mlp(const uint16_t* params, uint16_t* packed, uint16_t* packedT, uint32_t in_width, uint32_t W, uint32_t n_hidden_matmuls, hipStream_t s);
void launch_fused(int mode, const GridDevice& grid, uint32_t in_width, const FusedMlp& mlp, const LevelInfo* d_levels, const uint16_t* table, size_t table_bytes,
                  const float* coords, float* out, uint16_t* features_out, uint16_t* acts_out, size_t n, const uint32_t* d_n, size_t n_max, hipStream_t s,
                  const uint32_t* d_dest = nullptr, uint32_t queue_out_stride = 0, const uint8_t* brick_image = nullptr,
                  uint32_t sharers = 1, const struct PackArgs* pack = nullptr);
void launch_init_params(OptState* state, uint16_t* params, size_t n_mlp, size_t n_total, uint32_t in_width,
                        uint32_t n_hidden_matmuls, uint32_t width, uint64_t seed, hipStream_t s);
// master weights <- fp16 parameters; reset_optimizer also zeroes the moments and step counts
void launch_master_from_f16(const uint16_t* params, OptState* state, size_t n, bool reset_optimizer, hipStream_t s);

// ------------------------------------------------------------------------------------------------
uint16_t f32_to_f16(float f)
{
  uint32_t x;
  std::memcpy(&x, &f, 4);
  const uint32_t sign = (x >> 16) & 0x8000u;
  const uint32_t ax = x & 0x7fffffffu;
  if (ax >= 0x7f800000u) return (uint16_t)(sign | 0x7c00u | ((ax > 0x7f800000u) ? (0x200u | ((ax >> 13) & 0x3ffu)) : 0u));
  if (ax >= 0x477ff000u) return (uint16_t)(sign | 0x7c00u);
  if (ax < 0x33000001u) return (uint16_t)sign;
  const int32_t e = (int32_t)(ax >> 23) - 127;
  const uint32_t m = (ax & 0x7fffffu) | 0x800000u;
  const uint32_t shift = e < -14 ? (uint32_t)(-1 - e) : 13u;
  const uint32_t he = e < -14 ? 0u : (uint32_t)(e + 15);
  uint32_t q = m >> shift;
  const uint32_t rem = m & ((1u << shift) - 1u), half = 1u << (shift - 1);
  if (rem > half || (rem == half && (q & 1u))) q++;
  return (uint16_t)(sign | (he == 0 ? q : ((he - 1) << 10) + q));
}

float f16_to_f32(uint16_t h)
{
  const uint32_t sign = ((uint32_t)h & 0x8000u) << 16, e = (h >> 10) & 0x1fu, m = h & 0x3ffu;
  uint32_t x;
  if (e == 0) {
    if (m == 0) x = sign;
    else { float v = (float)m * 5.9604644775390625e-8f; std::memcpy(&x, &v, 4); x |= sign; }
  } else if (e == 31) x = sign | 0x7f800000u | (m << 13);
  else x = sign | ((e + 112u) << 23) | (m << 13);
  float f;
  std::memcpy(&f, &x, 4);
  return f;
}

// EXTERNAL tcnn GridEncodingTemplated ctor (level sizing) + grid_index's hash decision; scale/resolution
// as used by core/networks/tcnn_impl_decoder.cu:41-42.
uint32_t grid_make_layout(const ModelConfig& cfg, GridDevice* out)
{
  if (cfg.n_levels > (uint32_t)kMaxLevels) throw std::runtime_error("too many hash-grid levels");
  const float log2_pls = std::log2(cfg.per_level_scale);
  uint32_t offset = 0;
  for (uint32_t l = 0; l < cfg.n_levels; ++l) {
    const float scale = exp2f((float)l * log2_pls) * (float)cfg.base_resolution - 1.0f;
    const uint32_t res = (uint32_t)ceilf(scale) + 1u;
    const uint32_t max_params = 0xffffffffu / 2u;
    const double cube = (double)res * (double)res * (double)res;
    uint32_t n = cube > (double)max_params ? max_params : (uint32_t)cube;
    n = next_multiple(n, 8u);
    // EXTERNAL tcnn GridEncodingTemplated ctor: Dense keeps the full level, Tiled caps it at base_resolution^3 (the level then repeats
    // with that period), Hash at 2^log2_hashmap_size
    if (cfg.grid_type == 2u) {
      const double tile = (double)cfg.base_resolution * cfg.base_resolution * cfg.base_resolution;
      if ((double)n > tile) n = (uint32_t)tile;
    } else if (cfg.grid_type == 0u) {
      const uint32_t cap = 1u << cfg.log2_hashmap_size;
      if (n > cap) n = cap;
    }
    if (n == 0) throw std::runtime_error("grid level without entries");
    // tcnn grid_index: the stride walk stops as soon as the stride exceeds the level's size; a Hash grid then hashes instead
    uint32_t stride = 1, dims = 0;
    for (; dims < 3 && stride <= n; ++dims) stride *= res;
    LevelInfo& lv = out->levels[l];
    lv.scale = scale;
    lv.resolution = res;
    lv.res2 = res * res;
    lv.size = n;
    lv.offset = offset;
    if (cfg.grid_type == 0u) lv.hashed = n < stride ? 1u : 0u;
    else lv.hashed = (dims == 3 && (uint64_t)res * res * res <= (uint64_t)n) ? 0u : 1u + dims;   // 2 / 3 / 4: index over 1 / 2 / 3 dimensions, modulo
    lv.brick = lv.pad1 = 0;
    if (lv.hashed == 1u && (n & (n - 1)) != 0) throw std::runtime_error("internal: hashed level with non power-of-two size");
    if ((uint64_t)offset + n > 0xffffffffull) throw std::runtime_error("grid encoding too large (more than 2^32 entries)");
    offset += n;
  }
  out->n_levels = cfg.n_levels;
  out->n_features = cfg.n_features;
  out->interpolation = cfg.interpolation;
  return offset;
}

static std::set<Network*>& live_networks()
{
  static std::set<Network*> s;
  return s;
}

static uint32_t json_u32(const Json& j, const char* key, uint32_t def) { return j.contains(key) ? (uint32_t)j.at(key).as_int() : def; }
static float json_f32(const Json& j, const char* key, float def) { return j.contains(key) ? j.at(key).as_float() : def; }
static std::string json_str(const Json& j, const char* key, const char* def) { return j.contains(key) ? j.at(key).as_string() : std::string(def); }

void Network::configure(const Json& config, uint64_t init_seed)
{
  // tcnn_network.h:167-175: only loss / encoding / network are kept in the serialised model; the optimizer
  // options persist across re-configuration (m_optimizer_opts).
  const Json loss = config.value("loss", Json::object());
  const Json enc = config.value("encoding", Json::object());
  const Json net = config.value("network", Json::object());
  ModelConfig c = cfg_;  // keeps previous optimizer options unless given
  if (config.contains("optimizer")) {
    const Json& opt = config.at("optimizer");
    const std::string ot = json_str(opt, "otype", "Adam");
    const Json* adam = &opt;
    if (ot == "ExponentialDecay") {
      c.has_decay = true;
      c.decay_start = json_u32(opt, "decay_start", 10000);
      c.decay_interval = json_u32(opt, "decay_interval", 10000);  // tcnn defaults
      c.decay_base = json_f32(opt, "decay_base", 0.33f);
      if (!opt.contains("nested")) throw std::runtime_error("ExponentialDecay optimizer needs a 'nested' optimizer");
      adam = &opt.at("nested");
    } else {
      c.has_decay = false;
    }
    if (json_str(*adam, "otype", "Adam") != "Adam") throw std::runtime_error("unsupported optimizer otype (Adam / ExponentialDecay{Adam} only)");
    c.learning_rate = json_f32(*adam, "learning_rate", 1e-3f);
    c.beta1 = json_f32(*adam, "beta1", 0.9f);
    c.beta2 = json_f32(*adam, "beta2", 0.999f);
    c.epsilon = json_f32(*adam, "epsilon", 1e-8f);
    c.l2_reg = json_f32(*adam, "l2_reg", 1e-8f);
  }
  const std::string lt = json_str(loss, "otype", "L2");
  if (lt == "L1") c.loss = 0;
  else if (lt == "L2") c.loss = 1;
  else throw std::runtime_error("unsupported loss otype: " + lt);

  const std::string et = json_str(enc, "otype", "");
  // EXTERNAL tcnn create_grid_encoding: otype Grid / HashGrid / TiledGrid / DenseGrid, "type" (default by otype) Hash / Dense / Tiled
  if (et != "HashGrid" && et != "Grid" && et != "DenseGrid" && et != "TiledGrid")
    throw std::runtime_error("unsupported encoding otype: '" + et + "' (HashGrid / Grid / DenseGrid / TiledGrid)");
  const std::string gt = json_str(enc, "type", et == "DenseGrid" ? "Dense" : et == "TiledGrid" ? "Tiled" : "Hash");
  if (gt == "Hash") c.grid_type = 0;
  else if (gt == "Dense") c.grid_type = 1;
  else if (gt == "Tiled") c.grid_type = 2;
  else throw std::runtime_error("unsupported grid type: " + gt + " (Hash / Dense / Tiled)");
  c.n_levels = json_u32(enc, "n_levels", 16);
  c.n_features = json_u32(enc, "n_features_per_level", 2);
  c.log2_hashmap_size = json_u32(enc, "log2_hashmap_size", 19);
  c.base_resolution = json_u32(enc, "base_resolution", 16);
  c.per_level_scale = json_f32(enc, "per_level_scale", 2.0f);
  const std::string it = json_str(enc, "interpolation", "Linear");
  if (it == "Linear") c.interpolation = 0;
  else if (it == "Smoothstep") c.interpolation = 1;
  else if (it == "Nearest") c.interpolation = 2;
  else throw std::runtime_error("unsupported interpolation: " + it);
  // the reference reads both from the tcnn encoding object (tcnn_device_api.h:53-54), where only tcnn's own API can set them, so
  // a params.json never carries them; here they may be given with the encoding (absent = tcnn's defaults)
  c.quantize_threshold = json_f32(enc, "quantize_threshold", 0.0f);
  c.max_level = json_f32(enc, "max_level", 1000.0f);
  if (c.n_features != 1 && c.n_features != 2 && c.n_features != 4 && c.n_features != 8)
    throw std::runtime_error("n_features_per_level must be 1, 2, 4 or 8");  // method_raymarching.cu:1241-1244
  if (c.log2_hashmap_size > 28) throw std::runtime_error("log2_hashmap_size too large");

  const std::string nt = json_str(net, "otype", "FullyFusedMLP");
  if (nt != "FullyFusedMLP") throw std::runtime_error("unsupported network otype: " + nt + " (FullyFusedMLP only)");
  c.n_neurons = json_u32(net, "n_neurons", 128);
  c.n_hidden_layers = json_u32(net, "n_hidden_layers", 5);
  if (c.n_neurons != 16 && c.n_neurons != 32 && c.n_neurons != 64 && c.n_neurons != 128)
    throw std::runtime_error("FullyFusedMLP n_neurons must be 16, 32, 64 or 128");   // tcnn_impl.cu:315-347
  if (c.n_hidden_layers < 1) throw std::runtime_error("n_hidden_layers must be >= 1");
  // the activations the reference's kernels dispatch (tcnn_impl.cu:405-415, tcnn_device_api.h:274-285; no Sine there)
  auto activation_of = [](const std::string& a, const char* what) -> uint32_t {
    if (a == "None") return kActNone;
    if (a == "ReLU") return kActReLU;
    if (a == "Exponential") return kActExponential;
    if (a == "Sigmoid") return kActSigmoid;
    if (a == "Squareplus") return kActSquareplus;
    if (a == "Softplus") return kActSoftplus;
    throw std::runtime_error(std::string("unsupported ") + what + ": " + a + " (None / ReLU / Exponential / Sigmoid / Squareplus / Softplus)");
  };
  c.activation = activation_of(json_str(net, "activation", "ReLU"), "activation");
  c.output_activation = activation_of(json_str(net, "output_activation", "None"), "output_activation");

  cfg_ = c;
  model_ = Json::object();
  model_["loss"] = loss;
  model_["encoding"] = enc;
  model_["network"] = net;
  build_layout();
  steps_ = 0;
  lr_ = cfg_.learning_rate;   // a re-configured model starts its schedule over (tcnn_network.h:195-209 rebuilds optimizer and trainer)
  initialize_params(init_seed, Runtime::get().stream);
  live_networks().insert(this);
}

void Network::build_layout()
{
  const uint32_t total_entries = grid_make_layout(cfg_, &grid_);
  in_width_ = next_multiple(cfg_.n_levels * cfg_.n_features, 16u);
  // the instances of the fused kernel (infer_kernel.h dispatch): refuse the others here, at SetModel, not at the first launch
  const uint32_t widest = cfg_.n_features == 1 ? 32u : cfg_.n_features == 2 ? 64u : 128u;   // (1 and 2 features: all of kMaxLevels = 32 levels)
  if (in_width_ > widest)
    throw std::runtime_error("unsupported encoding shape: n_features_per_level=" + std::to_string(cfg_.n_features) + " with n_levels=" +
                             std::to_string(cfg_.n_levels) + " (encoded width " + std::to_string(in_width_) + " > " + std::to_string(widest) + ")");
  const size_t W = cfg_.n_neurons;
  n_mlp_ = W * in_width_ + (size_t)n_hidden_matmuls() * W * W + (size_t)16 * W;
  n_params_ = n_mlp_ + (size_t)total_entries * cfg_.n_features;
  if ((n_params_ - n_mlp_) * sizeof(uint16_t) >= (1ull << 32)) throw std::runtime_error("hash table >= 4 GiB is not supported");
  lds_halves_ = packed_mlp_halves(in_width_, cfg_.n_neurons, n_hidden_matmuls());
  lds_halves_T_ = packedT_halves(in_width_, cfg_.n_neurons, n_hidden_matmuls());
  params_f16_.resize(n_params_);
  mlp_packed_.resize(lds_halves_);
  mlp_packed_T_.resize(lds_halves_T_);
  levels_dev_.resize(kMaxLevels);
  levels_dev_.upload(grid_.levels, kMaxLevels, Runtime::get().stream);
  VNR_HIP_CHECK(hipStreamSynchronize(Runtime::get().stream));
  // training state is allocated lazily on the first training step
  opt_state_.release(); grads_.release(); grads_f32_.release();
  ws_batch_ = 0;
}

void Network::initialize_params(uint64_t seed, hipStream_t s)
{
  opt_state_.resize(n_params_);
  launch_init_params(opt_state_.ptr, params_f16_.ptr, n_mlp_, n_params_, in_width_, n_hidden_matmuls(), cfg_.n_neurons, seed, s);
  ++params_generation_;
  opt_sharded_ = false;
  refresh_inference_weights(s);
}

uint32_t Network::n_active_levels() const
{
  uint32_t n = 0;
  while (n < cfg_.n_levels && !((float)n >= cfg_.max_level + 1e-3f)) ++n;
  return n;
}

void Network::refresh_inference_weights(hipStream_t s)
{
  // both weight images (forward; transposed for the MLP backward) in one launch
  launch_pack_mlp(params_f16_.ptr, mlp_packed_.ptr, mlp_packed_T_.ptr, in_width_, cfg_.n_neurons, n_hidden_matmuls(), s);
  // the parameters changed: the brick image is stale (it is rebuilt once they have been left alone again -- for longer if this image was
  // dropped before it had paid for itself: network.h)
  if (brick_valid_ && brick_tier_ == 2) {
    if (brick_served_calls_ < kBrickPaysAfter) brick_after_scale_ = std::min<uint32_t>(brick_after_scale_ * 2u, 1u << 16);
    else brick_after_scale_ = 1;
  }
  brick_valid_ = false;
  brick_tier_ = 0;
  brick_stable_calls_ = 0;
  brick_served_calls_ = 0;
}

static uint32_t brick_after_base()
{
  static const uint32_t after = [] { const char* e = std::getenv("VNR_AMD_BRICK_AFTER"); return e ? (uint32_t)std::max(0, std::atoi(e)) : 24u; }();
  return after;
}

uint32_t Network::brick_after_now() const { return brick_after_base() * brick_after_scale_; }

double Network::brick_small_budget()
{
  // VNR_AMD_BRICK_SMALL_GB (0: no small tier).  0.75 GiB: swept on the bench model inside the reference application's loop (render, train one step,
  // render ...; bench.py `interactive`): none 190.1, 0.25 GiB 196.4, 0.75 GiB 200.2, 1.7 GiB 195.0 frames/s -- the build is 0.25 ms at 0.7 GB.
  static const double gb = [] { const char* e = std::getenv("VNR_AMD_BRICK_SMALL_GB"); return e ? std::max(0.0, std::atof(e)) : 0.75; }();
  return gb * 1073741824.0;
}

Network::~Network()
{
  live_networks().erase(this);
  for (hipEvent_t e : prof_events_) (void)hipEventDestroy(e);
  if (side_stream_) { (void)hipStreamSynchronize(side_stream_); (void)hipStreamDestroy(side_stream_); (void)hipEventDestroy(ev_fork_); (void)hipEventDestroy(ev_join_); }
}

void Network::set_brick_mode(int mode)
{
  brick_mode_ = mode < 0 ? -1 : (mode > 0 ? 1 : 0);
  brick_refused_ = false;
  brick_small_refused_ = false;
  if (brick_mode_ == 0 && brick_valid_) {   // the next launches read the parameter blob; the image's memory goes back
    if (Runtime::get().ready()) (void)hipDeviceSynchronize();
    brick_image_.release();
    levels_brick_dev_.release();
    brick_valid_ = false;
    brick_tier_ = 0;
  }
  brick_stable_calls_ = 0;
}

void Network::set_brick_budget(size_t bytes)
{
  brick_budget_ = bytes;
  brick_refused_ = false;
  brick_small_refused_ = false;
  if (brick_valid_) {
    if (Runtime::get().ready()) (void)hipDeviceSynchronize();
    brick_image_.release();
    levels_brick_dev_.release();
    brick_valid_ = false;
    brick_tier_ = 0;
  }
  brick_levels_mask_ = 0;
  brick_stable_calls_ = 0;
}

void Network::set_train_profiling(bool e)
{
  train_profiling_ = e;
  prof_steps_ = 0;
  if (e && prof_events_.empty()) {
    prof_events_.resize((size_t)kTrainProfileSteps * (kTrainPhases + 1));
    for (hipEvent_t& ev : prof_events_) VNR_HIP_CHECK(hipEventCreate(&ev));
  }
}

void Network::profile_mark(int boundary, hipStream_t s)
{
  if (!train_profiling_) return;
  VNR_HIP_CHECK(hipEventRecord(prof_events_[(size_t)(prof_steps_ % kTrainProfileSteps) * (kTrainPhases + 1) + boundary], s));
  if (boundary == kTrainPhases) ++prof_steps_;
}

int Network::train_profile(double* ms_per_step)
{
  for (int p = 0; p < kTrainPhases; ++p) ms_per_step[p] = 0.0;
  const int n = (int)std::min<uint64_t>(prof_steps_, kTrainProfileSteps);
  if (!train_profiling_ || n == 0) return 0;
  VNR_HIP_CHECK(hipEventSynchronize(prof_events_[(size_t)((prof_steps_ - 1) % kTrainProfileSteps) * (kTrainPhases + 1) + kTrainPhases]));
  for (int k = 0; k < n; ++k)
    for (int p = 0; p < kTrainPhases; ++p) {
      float ms = 0.0f;
      VNR_HIP_CHECK(hipEventElapsedTime(&ms, prof_events_[(size_t)k * (kTrainPhases + 1) + p], prof_events_[(size_t)k * (kTrainPhases + 1) + p + 1]));
      ms_per_step[p] += ms / n;
    }
  return n;
}

void Network::release_temporary()
{
  if (Runtime::get().ready()) (void)hipDeviceSynchronize();   // nothing may still read what is freed
  brick_image_.release();
  levels_brick_dev_.release();
  brick_valid_ = false;
  brick_tier_ = 0;
  brick_stable_calls_ = 0;
  ws_features_.release(); ws_acts_.release(); ws_dfeat_.release();   // re-allocated by the next training step (ws_batch_ = 0)
  ws_batch_ = 0;
}

void Network::release_temporary_of_all()
{
  for (Network* n : live_networks()) n->release_temporary();
}

// ------------------------------------------------------------------------------------------------ brick image (network.h)
// one thread per entry of a level's image: (brick, x-fastest position inside it) -> grid point -> the entry the reference's
// index function gives that point (level_index: hash or dense, exact `% size` semantics); points beyond the grid are zero
template <int F>
__global__ void brick_build_kernel(const LevelInfo lv, const half_t* __restrict__ table, uint8_t* __restrict__ image, uint64_t n_entries)
{
  typedef typename FeatVec<F>::type vec_t;
  constexpr uint32_t LX = BrickShape<F>::lx, LY = BrickShape<F>::ly, LZ = BrickShape<F>::lz;
  constexpr uint32_t E = 1u << (LX + LY + LZ);
  const uint32_t nbx = (lv.resolution >> LX) + 1u, nby = (lv.resolution >> LY) + 1u;
  vec_t* out = (vec_t*)(image + (size_t)(lv.brick - 1u) * 128u);
  const vec_t* src = (const vec_t*)(table + (size_t)lv.offset * F);
  if constexpr (F == 2) {   // 8 x 2 x 2 entries, column 7 = column 0 of the +x neighbour (grid_device.h gather_corners_brick)
    const uint32_t gnbx = lv.pad1, gnby = (lv.resolution >> 1) + 1u;
    for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += (uint64_t)gridDim.x * blockDim.x) {
      const uint32_t w = (uint32_t)(e & 31u);
      const uint64_t b = e >> 5;
      const uint32_t bx = (uint32_t)(b % gnbx), by = (uint32_t)((b / gnbx) % gnby), bz = (uint32_t)(b / ((uint64_t)gnbx * gnby));
      const uint32_t x = bx * 7u + (w & 7u), y = (by << 1) | ((w >> 3) & 1u), z = (bz << 1) | (w >> 4);
      vec_t v;
      if (x <= lv.resolution && y <= lv.resolution && z <= lv.resolution) v = src[level_index(lv, x, y, z)];
      else v = vec_t{};
      out[e] = v;
    }
    return;
  }
  for (uint64_t e = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; e < n_entries; e += (uint64_t)gridDim.x * blockDim.x) {
    const uint32_t w = (uint32_t)(e & (E - 1u));
    const uint64_t b = e >> (LX + LY + LZ);
    const uint32_t bx = (uint32_t)(b % nbx), by = (uint32_t)((b / nbx) % nby), bz = (uint32_t)(b / ((uint64_t)nbx * nby));
    const uint32_t x = (bx << LX) | (w & ((1u << LX) - 1u)), y = (by << LY) | ((w >> LX) & ((1u << LY) - 1u)), z = (bz << LZ) | (w >> (LX + LY));
    vec_t v;
    if (x <= lv.resolution && y <= lv.resolution && z <= lv.resolution) v = src[level_index(lv, x, y, z)];
    else v = vec_t{};
    out[e] = v;
  }
}

// The same image, a ROW OF BRICKS per block (round 6).  The kernel above gives every image entry a thread that de-hashes its grid point and
// gathers 2 F bytes from an unrelated line of the table: one lane address per entry, 1.3 TB/s of image (6.5 ms for the 8.8 GB of the bench
// model, 0.53 ms for the 0.7 GB an application that trains after every frame could afford per frame).  But along x the index function is
// (x ^ h(y, z)) & mask (hashed) or x + const (dense): the 64 consecutive x of a wave read ONE aligned 64-entry block of the table, permuted.
// So a block first copies its (y, z) rows -- the 2 x 2 (4 x 4 for F = 1) rows of grid points its bricks are made of, all x -- into LDS with
// those coalesced loads, and then writes its bricks entry by entry in image order: coalesced stores of whole 128-byte lines.
template <int F>
__global__ void __launch_bounds__(256) brick_build_rows_kernel(const LevelInfo lv, const half_t* __restrict__ table, uint8_t* __restrict__ image,
                                                               uint32_t nbx, uint32_t nby, uint32_t nbz)
{
  typedef typename FeatVec<F>::type vec_t;
  constexpr uint32_t LX = F == 2 ? 3u : BrickShape<F>::lx, LY = F == 2 ? 1u : BrickShape<F>::ly, LZ = F == 2 ? 1u : BrickShape<F>::lz;
  constexpr uint32_t E = 1u << (LX + LY + LZ), ROWS = 1u << (LY + LZ);
  static_assert(E * F * 2 == 128, "a brick is one 128-byte line");
  extern __shared__ __attribute__((aligned(16))) uint8_t s_raw[];
  vec_t* s_rows = (vec_t*)s_raw;                       // [ROWS][res + 1]
  const uint32_t res = lv.resolution, nx = (res + 4u) & ~3u;   // row stride: whole groups of four grid points
  const vec_t* src = (const vec_t*)(table + (size_t)lv.offset * F);
  vec_t* out = (vec_t*)(image + (size_t)(lv.brick - 1u) * 128u);
  for (uint32_t row = blockIdx.x; row < nby * nbz; row += gridDim.x) {   // a row of bricks: (by, bz)
    const uint32_t by = row % nby, bz = row / nby;
    for (uint32_t r = 0; r < ROWS; ++r) {
      const uint32_t y = (by << LY) | (r & ((1u << LY) - 1u)), z = (bz << LZ) | (r >> LY);
      const bool inside = y <= res && z <= res;        // (block-uniform)
      if (F == 2 && lv.hashed == 1u && lv.size >= 4u && inside) {
        // four grid points x4 .. x4 + 3 of a hashed level are the aligned group of four entries at (x4 ^ h) & mask & ~3, permuted by the low
        // two bits of h (uniform over the row): one 16-byte load per lane
        const uint32_t h = (y * 2654435761u) ^ (z * 805459861u), mask = lv.size - 1u, hl = h & 3u;
        for (uint32_t x4 = threadIdx.x * 4u; x4 < nx; x4 += 1024u) {
          const uint4_t v = *(const uint4_t*)(src + (((x4 ^ h) & mask) & ~3u));
          uint4_t o;
          if (hl == 0u) o = v; else if (hl == 1u) o = uint4_t{v.y, v.x, v.w, v.z}; else if (hl == 2u) o = uint4_t{v.z, v.w, v.x, v.y}; else o = uint4_t{v.w, v.z, v.y, v.x};
          *(uint4_t*)(s_rows + r * nx + x4) = o;       // (grid points beyond res: read, never used)
        }
      } else {
        for (uint32_t x = threadIdx.x; x < res + 1u; x += 256u) s_rows[r * nx + x] = inside ? src[level_index(lv, x, y, z)] : vec_t{};
      }
    }
    __syncthreads();
    vec_t* dst = out + (size_t)row * nbx * E;
    if constexpr (F == 2) {   // four entries (16 bytes) per lane: columns 0..3 or 4..7 of a brick's row r
      for (uint32_t t4 = threadIdx.x; t4 < nbx * 8u; t4 += 256u) {
        const uint32_t b = t4 >> 3, r = (t4 >> 1) & 3u, x0 = b * 7u + (t4 & 1u) * 4u;   // the 8th column repeats the +x neighbour brick's first
        const uint32_t* rowp = (const uint32_t*)(s_rows + r * nx);
        uint4_t o;
        o.x = x0 <= res ? rowp[x0] : 0u;
        o.y = x0 + 1u <= res ? rowp[x0 + 1u] : 0u;
        o.z = x0 + 2u <= res ? rowp[x0 + 2u] : 0u;
        o.w = x0 + 3u <= res ? rowp[x0 + 3u] : 0u;
        *(uint4_t*)(dst + (size_t)t4 * 4u) = o;
      }
    } else {
      for (uint32_t t = threadIdx.x; t < nbx * E; t += 256u) {
        const uint32_t b = t / E, w = t % E;
        const uint32_t wx = w & ((1u << LX) - 1u), r = w >> LX;
        const uint32_t x = (b << LX) | wx;
        dst[t] = x <= res ? s_rows[r * nx + x] : vec_t{};
      }
    }
    __syncthreads();
  }
}

template <int F>
static void launch_brick_build(const LevelInfo& lv, const uint16_t* table, uint8_t* image, uint64_t n_entries, hipStream_t s)
{
  static const bool rows = [] { const char* e = std::getenv("VNR_AMD_BRICK_BUILD_ROWS"); return !e || std::atoi(e) != 0; }();   // 0: the per-entry kernel (A/B, tests)
  constexpr uint32_t LX = F == 2 ? 3u : BrickShape<F>::lx, LY = F == 2 ? 1u : BrickShape<F>::ly, LZ = F == 2 ? 1u : BrickShape<F>::lz;
  const uint64_t res = lv.resolution;
  const uint32_t nbx = F == 2 ? lv.pad1 : (uint32_t)(res >> LX) + 1u, nby = (uint32_t)(res >> LY) + 1u, nbz = (uint32_t)(res >> LZ) + 1u;
  const size_t shmem = (size_t)(1u << (LY + LZ)) * ((res + 4) & ~(uint64_t)3) * F * 2;
  if (rows && shmem <= 64 * 1024 && (uint64_t)nbx * nby * nbz * (128u / (F * 2)) == n_entries) {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((uint64_t)nby * nbz, 1u << 20);
    brick_build_rows_kernel<F><<<blocks, 256, shmem, s>>>(lv, (const half_t*)table, image, nbx, nby, nbz);
  } else {
    const uint32_t blocks = (uint32_t)std::min<uint64_t>((n_entries + 255) / 256, 1u << 20);
    brick_build_kernel<F><<<blocks, 256, 0, s>>>(lv, (const half_t*)table, image, n_entries);
  }
  VNR_HIP_CHECK(hipGetLastError());
}

void Network::build_brick_image(hipStream_t s, bool small) const
{
  // which levels: the hashed ones (VNR_AMD_BRICK_DENSE=1: every level), finest first, while the image stays within the budget
  static const bool dense_too = [] { const char* e = std::getenv("VNR_AMD_BRICK_DENSE"); return e && std::atoi(e) != 0; }();
  // Budget.  The image is a cache in memory nothing else of the process uses: a renderer that holds a 1024^3 fp32 volume (4.3 GB), its
  // 140 MB model and its frame buffers occupies 2 % of the 288 GB of an MI355X.  Default: 1/16 of the device's memory (18 GB), and never
  // more than a quarter of what is free when the image is built; VNR_AMD_BRICK_MAX_GB or vnrAmdNeuralVolumeSetBrickImageBudget set it.
  // What a budget buys on the bench model (profiles/r03_brick_budget_table.json): the levels are taken finest first, because the
  // finest hashed levels are the ones whose 8 corners land in 8 unrelated lines of the table.
  static const double max_gb = [] { const char* e = std::getenv("VNR_AMD_BRICK_MAX_GB"); return e ? std::atof(e) : -1.0; }();
  size_t free_b = 0, total_b = 0;
  VNR_HIP_CHECK(hipMemGetInfo(&free_b, &total_b));
  double policy = brick_budget_ ? (double)brick_budget_ : max_gb >= 0.0 ? max_gb * 1073741824.0 : (double)total_b / 16.0;
  // the small tier (network.h): what an application that changes the parameters after every frame can afford to rebuild per frame
  if (small) policy = std::min(policy, brick_small_budget());
  const uint64_t budget_lines = (uint64_t)std::min(policy, (double)(free_b + brick_image_.bytes()) / 4.0) / 128u;
  const uint32_t mask_before = brick_valid_ ? brick_levels_mask_ : 0u;
  brick_levels_mask_ = 0;
  hipEvent_t t0, t1;
  VNR_HIP_CHECK(hipEventCreate(&t0)); VNR_HIP_CHECK(hipEventCreate(&t1));
  const uint32_t F = cfg_.n_features;
  const uint32_t lx = F == 8 ? 1 : 2, ly = F >= 4 ? 1 : 2, lz = F == 1 ? 2 : 1;
  const uint64_t entries_per_line = 64u / F;
  std::vector<LevelInfo> lv(grid_.levels, grid_.levels + kMaxLevels);
  std::vector<uint64_t> lines(kMaxLevels, 0);
  uint64_t used = 0;
  // Which levels first when not all fit.  F <= 2 (bricks of 32 / 64 entries): the FINEST, whose eight corners land in eight unrelated lines of the
  // table and whose big bricks are shared by the neighbouring samples of a wave (round 3's measurement on the bench model).  F >= 4 (bricks of
  // 16 / 8 entries, 2 x 2 x 2 cells for F = 8): the COARSEST.  A brick that small pays through neighbouring samples falling into the same
  // cells, which the coarser hashed levels give and a level as fine as the volume does not, and its image (gigabytes, read from HBM) replaces
  // a table of a few megabytes that lives in the caches: the reference's example model (L8 F8 T2^19, example-model.json) on the 1024^3 bench
  // volume, finest first = levels {3, 4, 6}, 17.6 GB, 181.9 frames/s; {3, 4, 5}, 2.5 GB: 185.4; no image 179.7 (round 6).
  const bool finest_first = F <= 2;
  for (int k = 0; k < (int)grid_.n_levels; ++k) {
    const int l = finest_first ? (int)grid_.n_levels - 1 - k : k;
    if (lv[l].hashed >= 2u || (!lv[l].hashed && !dense_too)) continue;   // (a Tiled level repeats: nothing to de-hash)
    if (brick_res_cap_ && lv[l].resolution > brick_res_cap_ + 1u) continue;
    const uint64_t res = lv[l].resolution;
    uint64_t n = ((res >> lx) + 1) * ((res >> ly) + 1) * ((res >> lz) + 1);
    if (F == 2) {   // the brick with a repeated column: 7 useful columns (grid_device.h gather_corners_brick)
      if (res >= 13000) continue;   // (x * 9363) >> 16 is x / 7 below 13 107
      n = (res / 7 + 1) * ((res >> 1) + 1) * ((res >> 1) + 1);
      lv[l].pad1 = (uint32_t)(res / 7 + 1);
    }
    if (n * entries_per_line >= (1ull << 32) || used + n + 1 > budget_lines || used + n + 1 >= 0xffffffffull) continue;
    lv[l].brick = (uint32_t)used + 1u;
    lines[l] = n;
    used += n;
    brick_levels_mask_ |= 1u << l;
  }
  if (used == 0) {
    if (!small) brick_refused_ = true;   // (nothing fits the small tier: the full one is still tried when its time comes)
    else brick_small_refused_ = true;    // ... and the small one is not asked for at every launch (sizes do not change with the parameters)
    brick_valid_ = false;
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    return;
  }
  // the small image of a model whose every bricked level fits the small budget IS the full image: the upgrade has nothing to rebuild
  if (!small && brick_valid_ && mask_before == brick_levels_mask_ && brick_image_.bytes() == (used + 1) * 128u) {
    brick_tier_ = 2;
    (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
    return;
  }
  if (brick_image_.bytes() != (used + 1) * 128u) {
    // (another size: nothing may still read the old image when it is freed -- launches of other streams that took its pointer)
    if (brick_image_.bytes()) VNR_HIP_CHECK(hipDeviceSynchronize());
    brick_valid_ = false;
  }
  brick_image_.resize((used + 1) * 128u);   // + one spare line: a pair load at the last entry reads 2 entries
  VNR_HIP_CHECK(hipEventRecord(t0, s));
  for (uint32_t l = 0; l < grid_.n_levels; ++l) {
    if (!lv[l].brick) continue;
    const uint64_t n_entries = lines[l] * entries_per_line;
    const uint16_t* table = params_f16_.ptr + n_mlp_;
    switch (F) {
    case 1: launch_brick_build<1>(lv[l], table, brick_image_.ptr, n_entries, s); break;
    case 2: launch_brick_build<2>(lv[l], table, brick_image_.ptr, n_entries, s); break;
    case 4: launch_brick_build<4>(lv[l], table, brick_image_.ptr, n_entries, s); break;
    default: launch_brick_build<8>(lv[l], table, brick_image_.ptr, n_entries, s); break;
    }
  }
  VNR_HIP_CHECK(hipMemsetAsync(brick_image_.ptr + used * 128u, 0, 128, s));
  levels_brick_dev_.resize(kMaxLevels);
  // pageable host source: the copy has completed for the host when the call returns, the device side is ordered on `s`
  VNR_HIP_CHECK(hipMemcpyAsync(levels_brick_dev_.ptr, lv.data(), kMaxLevels * sizeof(LevelInfo), hipMemcpyHostToDevice, s));
  VNR_HIP_CHECK(hipEventRecord(t1, s));
  VNR_HIP_CHECK(hipEventSynchronize(t1));
  VNR_HIP_CHECK(hipEventElapsedTime(&brick_build_ms_, t0, t1));
  (void)hipEventDestroy(t0); (void)hipEventDestroy(t1);
  brick_valid_ = true;
  brick_tier_ = small ? 1 : 2;
}

const LevelInfo* Network::inference_levels(hipStream_t s, const uint8_t** image, size_t n_max) const
{
  static const int env_mode = [] { const char* e = std::getenv("VNR_AMD_BRICK"); return e ? std::atoi(e) : -1; }();   // -1 auto, 0 off, 1 at once
  const int mode = brick_mode_ >= 0 ? brick_mode_ : env_mode;
  *image = nullptr;
  if (mode == 0 || brick_refused_) return levels_dev_.ptr;
  if (brick_tier_ != 2 || !brick_valid_) {
    if (brick_stable_calls_ < 0xffffffffu) ++brick_stable_calls_;
    const bool want_full = mode == 1 || brick_stable_calls_ > brick_after_now();
    if (want_full) {
      // (an upgrade from the small tier re-reads an image other streams may still be reading: build_brick_image synchronises before it frees)
      build_brick_image(s, false);
      if (brick_valid_ && brick_tier_ == 2) { ++brick_builds_; brick_served_calls_ = 0; }
    } else if (!brick_valid_ && !brick_small_refused_ && n_max >= kBrickSmallMinLaunch && brick_small_budget() > 0.0 && cfg_.n_features <= 2u) {
      // (big bricks only: with the 2 x 2 x 2 bricks of F = 8 a per-frame image earns less than its build -- the reference's example model on the
      // bench volume: 171.0 frames/s with it, 172.9 without, round 6)
      build_brick_image(s, true);
      if (brick_valid_) ++brick_small_builds_;
    }
    if (!brick_valid_) return levels_dev_.ptr;
  }
  if (brick_served_calls_ < 0xffffffffu) ++brick_served_calls_;
  // the image was built on one stream and build_brick_image returned only after the host had seen the build complete, so launches
  // on any stream may read it (a per-launch hipStreamWaitEvent on the build's event stood here: one more API call and one more barrier
  // packet in front of every evaluation kernel, always on a completed event)
  *image = brick_image_.ptr;
  return levels_brick_dev_.ptr;
}

void Network::set_params_f16(const uint16_t* host, size_t count, hipStream_t s)
{
  if (count != n_params_) throw std::runtime_error("parameter count mismatch: got " + std::to_string(count) + ", model has " + std::to_string(n_params_));
  VNR_HIP_CHECK(hipMemcpyAsync(params_f16_.ptr, host, count * sizeof(uint16_t), hipMemcpyHostToDevice, s));
  if (opt_state_.count == n_params_) launch_master_from_f16(params_f16_.ptr, opt_state_.ptr, n_params_, false, s);  // moments kept, as before
  ++params_generation_;
  refresh_inference_weights(s);
  VNR_HIP_CHECK(hipStreamSynchronize(s));
}

void Network::get_params_f16(uint16_t* host, size_t count, hipStream_t s) const
{
  if (count != n_params_) throw std::runtime_error("parameter count mismatch");
  VNR_HIP_CHECK(hipMemcpyAsync(host, params_f16_.ptr, count * sizeof(uint16_t), hipMemcpyDeviceToHost, s));
  VNR_HIP_CHECK(hipStreamSynchronize(s));
}

// EXTERNAL tcnn Trainer::serialize(): {"n_params", "params_type": "__half", "params_binary"} (key names
// confirmed by apps/view_model.cpp:124-126); no optimizer state with the reference's default arguments.
Json Network::serialize_params(hipStream_t s) const
{
  std::vector<uint16_t> host(n_params_);
  get_params_f16(host.data(), n_params_, s);
  Json j = Json::object();
  j["n_params"] = (uint64_t)n_params_;
  j["params_type"] = "__half";
  j["params_binary"] = Json::binary(host.data(), host.size() * sizeof(uint16_t));
  return j;
}

void Network::deserialize_params(const Json& j, hipStream_t s)
{
  const size_t n = (size_t)j.at("n_params").as_int();
  if (n != n_params_) throw std::runtime_error("Can't set params because buffer has the wrong size: " + std::to_string(n) + " vs " + std::to_string(n_params_));
  const std::string type = j.at("params_type").as_string();
  const std::string& bin = j.at("params_binary").as_binary();
  std::vector<uint16_t> host(n);
  if (type == "__half") {
    if (bin.size() != n * 2) throw std::runtime_error("params_binary has the wrong size");
    std::memcpy(host.data(), bin.data(), n * 2);
  } else if (type == "float") {
    if (bin.size() != n * 4) throw std::runtime_error("params_binary has the wrong size");
    const float* f = (const float*)bin.data();
    for (size_t i = 0; i < n; ++i) host[i] = f32_to_f16(f[i]);
  } else {
    throw std::runtime_error("Unknown params_type: " + type);
  }
  set_params_f16(host.data(), n, s);
  if (j.contains("optimizer") || j.contains("step")) { /* optimizer state is not stored by the reference */ }
}

void Network::inference(const float* d_coords, float* d_out, size_t n, const uint32_t* d_n, size_t n_max, hipStream_t s,
                        const uint32_t* d_dest) const
{
  const uint8_t* image;
  const LevelInfo* levels = inference_levels(s, &image, d_n ? n_max : n);
  GridDevice grid = grid_;
  grid.n_levels = n_active_levels();   // masked levels encode to zero, like the padding
  launch_fused(0, grid, in_width_, fused_mlp(), levels, params_f16_.ptr + n_mlp_, n_grid_params() * 2, d_coords, d_out, nullptr, nullptr, n, d_n, n_max, s, d_dest, 0, image);
}

bool Network::inference_queue(const float* d_records, float* d_out, uint32_t out_stride, const uint32_t* d_n, size_t n_max, hipStream_t s,
                              uint32_t sharers, const PackArgs* pack) const
{
  const uint8_t* image;
  const LevelInfo* levels = inference_levels(s, &image, n_max);
  GridDevice grid = grid_;
  grid.n_levels = n_active_levels();
  if (cfg_.n_neurons == 128u) pack = nullptr;   // blocks of 8 waves: the caller launches the packing kernel itself
  launch_fused(0, grid, in_width_, fused_mlp(), levels, params_f16_.ptr + n_mlp_, n_grid_params() * 2, d_records, d_out, nullptr, nullptr, 0, d_n, n_max, s, nullptr,
               out_stride, image, sharers, pack);
  return pack != nullptr;
}

FusedMlp Network::fused_mlp() const
{
  // diagnostics: VNR_AMD_WEIGHTS_GLOBAL=1 reads the A operands of every 128-neuron model from global memory (what models beyond the LDS always do)
  static const bool force_global = [] { const char* e = std::getenv("VNR_AMD_WEIGHTS_GLOBAL"); return e && std::atoi(e) != 0; }();
  const bool wglobal = !weights_in_lds() || (force_global && cfg_.n_neurons == 128u);
  return FusedMlp{mlp_packed_.ptr, lds_halves_, cfg_.n_neurons, n_hidden_matmuls(), cfg_.activation, cfg_.output_activation, !common_kind() || wglobal, wglobal,
                  cfg_.quantize_threshold};
}

bool Network::tile_net(TileNet* out, hipStream_t s) const
{
  if (!weights_in_lds()) return false;   // (a deeper network reads its weights from global memory: evaluation kernels only)
  if (n_grid_params() * 2 >= (1ull << 32)) throw std::runtime_error("hash table >= 4 GiB is not supported");
  const uint8_t* image;
  const LevelInfo* levels = inference_levels(s, &image, kBrickSmallMinLaunch);   // (a frame evaluated inside the marching / tracking loop)
  out->levels = levels;
  out->n_levels = n_active_levels();
  out->interpolation = grid_.interpolation;
  out->table = (const half_t*)(params_f16_.ptr + n_mlp_);
  out->table_bytes = (uint32_t)(n_grid_params() * 2);
  out->brick_image = image;
  out->packed_mlp = (const half_t*)mlp_packed_.ptr;
  out->lds_halves = lds_halves_;
  out->n_hidden_matmuls = n_hidden_matmuls();
  out->activation = cfg_.activation;
  out->output_activation = cfg_.output_activation;
  out->n_features = grid_.n_features;
  out->in_width = in_width_;
  out->width = cfg_.n_neurons;
  out->general = common_kind() ? 0u : 1u;
  out->quantize_threshold = cfg_.quantize_threshold;
  return true;
}

void Network::encode(const float* d_coords, uint16_t* d_features, size_t n, hipStream_t s) const
{
  const uint8_t* image;
  const LevelInfo* levels = inference_levels(s, &image, n);
  GridDevice grid = grid_;
  grid.n_levels = n_active_levels();
  launch_fused(1, grid, in_width_, fused_mlp(), levels, params_f16_.ptr + n_mlp_, n_grid_params() * 2, d_coords, nullptr, d_features, nullptr, n, nullptr, n, s, nullptr, 0, image);
}

size_t Network::bytes_allocated() const
{
  return brick_image_.bytes() + params_f16_.bytes() + mlp_packed_.bytes() + mlp_packed_T_.bytes() + opt_state_.bytes() + grads_.bytes() + grads_f32_.bytes() + ws_features_.bytes() + ws_acts_.bytes() + ws_dfeat_.bytes() + ws_loss_.bytes();
}

}  // namespace vnr
