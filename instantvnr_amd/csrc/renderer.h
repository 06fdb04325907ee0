// renderer.h — MainRenderer equivalent (renderer.{h,cpp}, object.{h,cpp}, framebuffer.h) with the
// sample-streaming ray marcher (core/renderer/method_raymarching.{h,cu}, dda.h, raytracing.h).
#pragma once
#include <vector>

#include <memory>

#include "volume.h"

namespace vnr {

struct CameraData {  // instantvnr_types.h:73-83
  vec3f from{0, 0, -1}, at{0, 0, 0}, up{0, 1, 0};
  float fovy = 60.0f;
};

struct FrameStats {
  uint64_t n_samples = 0, n_reference_slots = 0;
  uint32_t n_iterations = 0, n_rays_hit = 0;
  double infer_kernel_ms = 0.0;
  uint64_t infer_kernel_launches = 0;
  double infer_union_ms = 0.0;   // time during which at least one evaluation kernel of the frame was running (any stream)
};

struct RenderParams;  // device-visible POD (render.hip)

class Renderer {
public:
  explicit Renderer(std::shared_ptr<VolumeBase> volume);  // api.cpp:419-459 vnrCreateRenderer
  ~Renderer();

  void resize(int w, int h);                         // renderer.h:96-112
  void set_camera(const CameraData& c) { finish_pending(); camera_ = c; reset_ = true; }
  void set_transfer_function(const TransferFunctionData& t);
  void set_mode(int m) { finish_pending(); mode_ = m; reset_ = true; } // renderer.h:139-146
  // the step of the march is 1 / rate (object.cpp:303-304): zero, a negative number, NaN or infinity have no step
  void set_sampling_rate(float r)
  {
    if (!(r > 0.0f) || !(r < 3.0e38f)) throw std::runtime_error("volume sampling rate must be a positive finite number");
    sampling_rate_ = r; reset_ = true;
  }
  void set_density_scale(float s) { density_scale_ = s; reset_ = true; }
  void reset_accumulation() { reset_ = true; }
  void set_device_output(bool e) { finish_pending(); skip_download_ = e; }
  void set_pixel_range(uint32_t lo, uint32_t hi) { pixel_lo_ = lo; pixel_hi_ = hi; reset_ = true; }
  void set_profiling(bool e) { profiling_ = e; }
  // Asynchronous frames (device framebuffer output, single-pass streaming modes 5 / 6 / 8 / 9): render() returns once the
  // iterations the previous frame needed are enqueued; map_frame(), stats(), the next render() and every call that changes
  // what the frame reads complete it first (further iterations if rays are still alive).  Off by default: the caller
  // must not change the volume (training, decoding, time step) between render() and map_frame().
  void set_async(bool e) { finish_pending(); async_ = e; }
  // rendering modes 6 / 9 / 12 / 14 / 15 on a neural volume: 1 the in-shader kernels, 0 the streaming path, -1 the environment's
  // choice (VNR_AMD_IN_SHADER) or, without one, whichever is faster for the mode (render.hip in_shader_applies: ray marching
  // streams, path tracing runs in shader).  Same frames (ray marching: up to the streaming path's resume rounding).
  void set_in_shader_kernel(int mode) { finish_pending(); in_shader_mode_ = mode; }
  // diagnostics: the compacted sample queue of the last iteration and per-iteration kernel times of the last frame
  const float* debug_coords() { finish_pending(); return (const float*)queue_.ptr; }  // 16-byte records {x, y, z, slot}
  const uint32_t* debug_counters() { finish_pending(); return counters_.ptr; }
  const std::vector<float>& debug_iteration_ms() { finish_pending(); return iter_ms_; }
  // which schedule the last streaming frame ran: {samples per ray and iteration, ray parts, 1 = packing fused into the evaluation kernel, 1 = decoupled loop}
  const int* debug_schedule() { finish_pending(); return last_schedule_; }
  // rank `part` of `parts` renders the pixel blocks b with b % parts == part (block = `block` consecutive pixels)
  void set_pixel_interleave(uint32_t block, uint32_t parts, uint32_t part)
  {
    if (block == 0 || parts == 0 || part >= parts) throw std::runtime_error("invalid pixel interleave");
    il_block_ = block; il_parts_ = parts; il_part_ = part; reset_ = true;
  }

  void render();                 // renderer.cpp:59-140
  const float* map_frame();      // renderer.h:84-94
  // Image-tile sharding over the ranks of Dist (dist.h; new work).  The rank renders the tile rows (8 scanlines)
  // r with r % world == rank straight into ITS slot of a [world][n_local] buffer (local pixel order: the share is compact), one
  // in-place all-gather fills the other slots and one kernel de-interleaves them into the width x height frame that map_frame()
  // returns, so vnrRender + vnrRendererMapFrame give every rank the whole frame, as on one GPU.  Global pixel indices keep the
  // random sequences and the accumulation exact: the assembled frame equals the unsharded one bit for bit.
  void set_distributed(bool e);
  bool distributed() const { return distributed_; }
  // a pipeline of depth one for throughput: enqueues frame k, gathers frame k - 1 on the communication stream while the GPU
  // renders, completes frame k and returns the assembled frame k - 1 (nullptr on the first call); flush_pipeline() gathers and
  // returns the frame still in flight.  Undistributed renderers return frame k itself.
  const float* render_pipelined();
  const float* flush_pipeline();
  const FrameStats& stats() { finish_pending(); return stats_; }
  const FrameStats& completed_stats() const { return completed_stats_; }   // of the frame completed last; completes nothing (pipelined frames)
  int width() const { return width_; }
  int height() const { return height_; }

private:
  struct StreamingFrame;   // render.hip
  void render_streaming(const RenderParams& p, int pass_mode, bool defer = false);   // one iterative_raymarching_loop<MODE> (method_raymarching.cu:931-958)
  void launch_iteration(StreamingFrame& f, int part);
  void launch_tail(StreamingFrame& f, int part, uint32_t it, hipStream_t s);
  void finish_streaming(StreamingFrame& f);
  // the same loop with the walk decoupled from evaluation and compose (decoupled.h): three kernels per iteration on three streams per
  // ray part, the walks up to `decoupled_ahead_` batches ahead of the composes.  Frames are bit-identical to render_streaming's.
  bool decoupled_applies(const RenderParams& p, int pass_mode) const;
  void render_decoupled(const RenderParams& p, bool defer);
  void decoupled_step(StreamingFrame& f, int part, bool head_only);
  void finish_decoupled(StreamingFrame& f);
  void collect_eval_profile(StreamingFrame& f, const uint32_t* launched, const uint32_t* used);
  void finish_pending();
  void render_monolithic(const RenderParams& p);
  // rendering modes 6 / 9 / 12 on a neural volume: one launch, the network evaluated inside the marching loop (in_shader.h).
  // false: this model / configuration is not covered and the caller takes the streaming path (same frames up to resume rounding)
  bool in_shader_applies() const;
  void render_in_shader(const RenderParams& p, int shade);
  void render_pathtracing(const RenderParams& p);   // do_path_tracing_iterative (method_pathtracing.cu:786-806)
  void ensure_queues(size_t n_pixels, int n_iters, bool gradient);
  void ensure_share_buffers();
  void issue_gather(int buf);    // all-gather + assemble (+ download) of the frame in gathered_[buf], on the communication stream
  const float* hand_out_frame(int buf);

  std::shared_ptr<VolumeBase> volume_;
  CameraData camera_;
  TfnObject tfn_;
  int mode_ = 5;  // api.cpp:456
  float sampling_rate_ = 1.0f, density_scale_ = 1.0f;
  int width_ = 0, height_ = 0;
  uint32_t pixel_lo_ = 0, pixel_hi_ = 0xffffffffu;
  uint32_t il_block_ = 8, il_parts_ = 1, il_part_ = 0;
  bool reset_ = true, skip_download_ = false, profiling_ = false, async_ = false;
  // Two slots of frame state (ray lists, queues, result arenas, counters, pinned rings, events): with asynchronous frames the
  // HEAD of frame k + 1 (ray generation, first batch of samples, their evaluation, packing: nothing in it writes a pixel that
  // frame k contributes to) is enqueued before the host has seen frame k complete, so the GPU has work during the host's
  // turn-around between two frames (55 us of a 0.7 ms share on one of eight GPUs)
  std::unique_ptr<StreamingFrame> frame_[2];
  int slot_ = 0;            // the slot of the frame rendered last (pending or complete)
  int frame_of_buffer_[2] = {-1, -1};   // which slot's streaming frame wrote framebuffer i (-1: another rendering mode)
  int frame_index_ = 0;
  // samples per ray and iteration, VNR_RM_N_ITERS (method_raymarching.cu:30-40; 16 there, tuned on the authors' GPU).  Frames
  // depend on it only through the last bit of samples at batch boundaries (0.2 % of the pixels, max 4e-5); on MI355X 24 is the fastest (bench workload: 16: 122, 24: 129, 32: 126 frames/s)
  int n_iters_ = 24;
  DeviceBuffer<unsigned long long> is_samples_{MemTag::Renderer};   // in-shader kernel statistics, kInShaderStatSlots each
  DeviceBuffer<uint32_t> is_hits_{MemTag::Renderer};
  int in_shader_mode_ = -1;
  // VNR_AMD_MARCH_RANKS (read when the renderer is created): 1 keeps the depth sort's ranks in LDS (2 bytes per sample), 0 (default since
  // round 5) claims the slot inside a bin a second time when the record is written.  Same frames.  Without the ranks the FIRST march of
  // a frame fits three blocks per CU instead of two (50 KB against 62 KB of LDS at 24 samples per ray): march_kernel<true> 200 -> 160 us,
  // the C4 frame 3.67 -> 3.58 ms (profiles/r05_march_occupancy.txt)
  bool march_ranks_ = false;
  bool n_iters_fixed_ = false;   // VNR_RM_N_ITERS given: no adaptation to the size of the share
  // LaunchParams::light_directional_dir (instantvnr_types.h:148): a member the reference negates IN PLACE whenever it points
  // along the view direction (renderer.cpp:98-101), so it persists across frames
  vec3f light_dir_ = {0.7f, 0.9f, 0.4f};
  uint32_t tile_w_log2_ = 3;  // ray tiles of 2^w x 2^(6-w) pixels (render.hip map_pixel)
  static constexpr int kMaxParts = 4;
  int n_halves_ = 2;  // streaming mode: rays dealt to n parts on n streams (march of one overlaps inference of another)
  bool n_halves_fixed_ = false;
  // parts of a share of at most 524 288 pixels (half of a 1024 x 1024 frame; a quarter until late round 5: three parts instead of two are worth 1.9 % on the
  // 1/2 and on the 1/3 share, 1.935 -> 1.899 and 1.345 -> 1.319 ms, and cost the whole frame 0.4 %): with pipelined frames several short chains side by side beat 2
  // (round 2: 1/4 share 1.17 -> 1.13 ms, 1/8 share 0.66 -> 0.65 ms; whole frames: no difference).  THREE since round 5, not four: the runtime maps
  // a process's HIP streams onto four hardware queues, a rank of a multi-GPU run also owns a communication stream, and with four part streams
  // plus that one two parts share a queue and run one behind the other: 1/8 share 0.555 -> 0.794 ms, 1/4 share 1.02 -> 1.37 ms with ONE more
  // stream alive in the process, against 0.539 -> 0.558 and 1.03 -> 1.06 ms with three parts (profiles/r05_stream_budget.txt).
  int small_share_parts_ = 3;
  uint32_t predicted_iterations_[2][kMaxParts] = {};   // [camera pass | shadow pass][half]
  // decoupled path (VNR_AMD_DECOUPLED): 0 never, 1 (default) for ray sets of at most 20 480 rays, 2 always.  Measured on shares of the bench
  // frame (profiles/r03_decoupled_share_sweeps.txt): 1/64 (16 384 rays) 0.203 against 0.248 ms coupled, 1/32 equal, 1/16 0.45 against 0.38,
  // 1/8 0.85 against 0.58: the walk kernel is as expensive as the evaluation (0.43 ms alone on the 1/8 share) and the two do not share a
  // CU gracefully, so taking the walks off the chain pays only where the GPU is mostly idle.  Look-ahead of the walks in batches
  // (VNR_AMD_DECOUPLED_AHEAD: 1 = walk, evaluate, compose in turn; default 3); ray parts (VNR_AMD_DECOUPLED_PARTS, default 1)
  int decoupled_mode_ = 1, decoupled_ahead_ = 3, decoupled_parts_ = 1;
  uint32_t decoupled_predicted_[kMaxParts] = {};
  hipStream_t d_streams_[kMaxParts][3] = {};           // walk, evaluate, compose
  DeviceBuffer<uint32_t> d_words_[2];                  // per frame slot: ray state, batch records, counters
  DeviceBuffer<vec4f> d_queue_[2];
  DeviceBuffer<vec2f> d_arena_[2];
  void* d_host_ = nullptr;                             // pinned DHost[2][kMaxParts]
  size_t d_rays_ = 0; int d_iters_ = 0, d_ring_ = 0;   // what the buffers are sized for
  hipStream_t stream_ = nullptr, part_streams_[kMaxParts] = {};
  hipEvent_t ev_fork_ = nullptr;
  bool own_stream_ = false;

  // framebuffer: double-buffered device + pinned host (framebuffer.h:7-98)
  DeviceBuffer<vec4f> fb_[2], accumulation_;
  vec4f* host_fb_[2] = {nullptr, nullptr};
  size_t host_fb_pixels_ = 0;
  int fb_cur_ = 0;

  // distributed mode: [world][n_local] share buffers and assembled frames, both double-buffered like fb_
  bool distributed_ = false;
  uint32_t share_world_ = 1, share_rank_ = 0, share_n_local_ = 0;
  DeviceBuffer<vec4f> gathered_[2], full_[2];
  hipEvent_t ev_rendered_ = nullptr, ev_gathered_[2] = {nullptr, nullptr};
  bool gather_issued_[2] = {false, false};
  int pipe_prev_ = -1;           // buffer of the rendered but not yet gathered frame of render_pipelined()

  // streaming queues
  DeviceBuffer<uint32_t> q_u32_;   // pixel_index[2], sample_base[2], sample_count[2]
  DeviceBuffer<float> q_f32_;      // jitter[2], alpha[2], color[2][3], t_next[2][3], next_cell_begin[2]
  DeviceBuffer<int> q_i32_;        // cell[2][3]
  DeviceBuffer<float> q_ssh_;      // single-shade heuristic: highest org[2][3], colour[2][3], alpha[2] per ray
  DeviceBuffer<float> pt_rays_, pt_values_;   // path tracing: 2 x 26 planes of ray state, one value per alive ray
  uint32_t predicted_pt_ = 0;
  DeviceBuffer<float> ssh_px_;     // ... and per pixel: org[3], colour[3], alpha, second jitter, unshaded rgba[4]
  DeviceBuffer<uint32_t> ray_counts_;  // surviving rays per 64-ray group of the last march (order-preserving compaction)
  DeviceBuffer<vec4f> queue_;      // gather-order sample records {x, y, z, result slot}
  DeviceBuffer<float> arena_;      // evaluation results, x2 (ping-pong): per slot {value, t1 - t0}, then (gradient shading) 4 more floats
  bool queue_grad_ = false;        // queues currently sized for 4 records / 6 result floats per sample
  DeviceBuffer<uint32_t> counters_;  // 2 x C_COUNT: one block per half
  uint32_t* host_counts_ = nullptr;  // pinned rings of alive-ray counts, 2 x 256
  size_t queue_pixels_ = 0;
  int queue_iters_ = 0;
  std::vector<hipEvent_t> events_[2][kMaxParts];  // per slot and half: (before, after) the evaluation kernel of each iteration
  FrameStats stats_, completed_stats_;
  std::vector<float> iter_ms_;
  int last_schedule_[4] = {0, 0, 0, 0};
};

}  // namespace vnr
