/* vnr_oracle.h — CPU oracle for the instantvnr hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under instantvnr_amd/ links, imports or
 * calls this.  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may use it, and there only as the checker.
 *
 * It is a plain-C restatement of the arithmetic of the reference
 * (VIDILabs/instantvnr); every function cites the reference file:line it
 * follows.  Where the arithmetic lives in a dependency that is NOT under
 * /root/reference (tiny-cuda-nn fork `wilsonCernWq/tiny-cuda-nn`, un-vendored
 * submodule `.gitmodules:1-3`, no pinned SHA; gdt/OVR headers), the published
 * upstream algorithm is restated and the assumption is spelled out at the
 * function.
 *
 * PARITY STATUS: **parity unpinned**.  The reference ships no tests, golden
 * vectors or fixtures for this path and cannot be compiled here (needs CUDA,
 * OptiX, OVR, tcnn).  The oracle is pinned only against (a) hand-computed
 * cases in tests/test_oracle_*.py and (b) the reference's in-tree restatement
 * of the tcnn inference kernels (core/networks/tcnn_impl_decoder.cu,
 * tcnn_threadblock.h), which is what fixes the forward semantics.
 */
#ifndef VNR_ORACLE_H
#define VNR_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- fp16 ------------------------------------------------------------- */
uint16_t vnro_f32_to_f16(float f);  /* round-to-nearest-even */
float    vnro_f16_to_f32(uint16_t h);

/* ---- hash grid -------------------------------------------------------- */
typedef struct {
  uint32_t n_levels;
  uint32_t n_features;        /* per level: 1,2,4,8 */
  uint32_t log2_hashmap_size;
  uint32_t base_resolution;
  float    per_level_scale;   /* tcnn default 2.0 */
  uint32_t interpolation;     /* 0 = Linear (default), 1 = Smoothstep, 2 = Nearest (tcnn_impl_decoder.cu:73-94) */
  float    quantize_threshold;/* corner values below it in magnitude count as 0 (:120); EXTERNAL tcnn default 0 */
  float    max_level;         /* levels l >= max_level + 1e-3 encode to 0 (:17); EXTERNAL tcnn default 1000 (no masking) */
  uint32_t grid_type;         /* 0 = Hash (default), 1 = Dense, 2 = Tiled: tcnn GridType, passed to grid_index at tcnn_impl_decoder.cu:68-69 */
} vnro_grid_config;

#define VNRO_MAX_LEVELS 32

typedef struct {
  uint32_t offsets[VNRO_MAX_LEVELS + 1]; /* in entries (x n_features for elements) */
  float    scale[VNRO_MAX_LEVELS];
  uint32_t resolution[VNRO_MAX_LEVELS];
} vnro_grid_layout;

/* returns total number of table entries (sum over levels) */
uint32_t vnro_grid_make_layout(const vnro_grid_config* cfg, vnro_grid_layout* out);

/* index of corner p in a level (entries, not elements); Hash grid */
uint32_t vnro_grid_index(uint32_t hashmap_size, uint32_t resolution, const uint32_t p[3]);
/* the same for any grid type (0 Hash, 1 Dense, 2 Tiled) */
uint32_t vnro_grid_index_typed(uint32_t grid_type, uint32_t hashmap_size, uint32_t resolution, const uint32_t p[3]);

/* coords: [n][3] fp32 AoS in [0,1]; table: fp16 [total_entries * F];
 * out: fp16 [n][padded_width] row-major (padded_width = roundup(L*F,16), pad = 0) */
void vnro_grid_encode(const vnro_grid_config* cfg, const uint16_t* table,
                      const float* coords, size_t n, uint16_t* out, uint32_t padded_width);

/* ---- fully fused MLP -------------------------------------------------- */
/* the activations the reference dispatches (tcnn_impl.cu:405-415, tcnn_device_api.h:274-285).  `activation` arguments below carry the
 * hidden activation in bits 0-7 and the OUTPUT activation in bits 8-15 (0 = None, the reference's example-model.json). */
enum { VNRO_ACT_NONE = 0, VNRO_ACT_RELU = 1, VNRO_ACT_EXPONENTIAL = 2, VNRO_ACT_SIGMOID = 3, VNRO_ACT_SQUAREPLUS = 4, VNRO_ACT_SOFTPLUS = 5 };
enum { VNRO_ACC_F32 = 0, VNRO_ACC_F16 = 1 };

/* weights: fp16, row-major [out][in] per layer: first (W x in_width),
 * n_hidden_matmuls x (W x W), last (16 x W).  input fp16 [n][in_width].
 * out fp32 [n] = output neuron 0.  If act_out != NULL it receives the
 * post-activation fp16 hidden activations: [(n_hidden_matmuls+1)][n][W]. */
void vnro_mlp_forward(const uint16_t* weights, uint32_t in_width, uint32_t width,
                      uint32_t n_hidden_matmuls, int activation, int acc_mode,
                      const uint16_t* input, size_t n, float* out, uint16_t* act_out);

size_t vnro_mlp_n_params(uint32_t in_width, uint32_t width, uint32_t n_hidden_matmuls);

/* encode + mlp, params blob = MLP weights first, then grid (tcnn order) */
void vnro_network_inference(const vnro_grid_config* cfg, uint32_t width, uint32_t n_hidden_layers,
                            int activation, int acc_mode, const uint16_t* params,
                            const float* coords, size_t n, float* out);

/* ---- ground-truth volume sampling ------------------------------------ */
/* tex3D<float>(p) on a normalised-coordinate, linear-filter, clamp texture */
float vnro_tex3d(const float* vol, const int dims[3], float px, float py, float pz);
/* renderer's sampleVolume(): nodal remap then tex3D */
float vnro_sample_volume(const float* vol, const int dims[3], float px, float py, float pz);
void  vnro_sample_volume_batch(const float* vol, const int dims[3], const float* coords, size_t n,
                               int nodal, float* out);

/* ---- transfer function ------------------------------------------------ */
typedef struct {
  const float* colors; /* [n_colors][4] rgba, a = 1 */
  int n_colors;
  const float* alphas; /* [n_alphas] */
  int n_alphas;
  float range_lo, range_hi, range_rcp_norm;
} vnro_tfn;

void vnro_tfn_sample(const vnro_tfn* tfn, float value, float rgb[3], float* alpha);

/* ---- macrocell --------------------------------------------------------- */
#define VNRO_MACROCELL_SIZE_MIP 4
void vnro_macrocell_shape(const int vol_dims[3], int mc_dims[3], float mc_spacings[3]);
void vnro_macrocell_update_explicit(const float* coords, const float* values, size_t n,
                                    const int vol_dims[3], const int mc_dims[3], float* value_range);
void vnro_macrocell_compute_implicit(const float* vol, const int vol_dims[3],
                                     const int mc_dims[3], float* value_range);
void vnro_macrocell_max_opacity(const vnro_tfn* tfn, const float* value_range, size_t n_cells,
                                float* max_opacity);

/* ---- RNG --------------------------------------------------------------- */
typedef struct { uint32_t state; } vnro_lcg;
void  vnro_lcg_init(vnro_lcg* r, uint32_t v0, uint32_t v1); /* 16-round TEA */
float vnro_lcg_next(vnro_lcg* r);

typedef struct { uint64_t state, inc; } vnro_pcg32;
void     vnro_pcg32_seed(vnro_pcg32* r, uint64_t initstate, uint64_t initseq);
uint32_t vnro_pcg32_next_uint(vnro_pcg32* r);
float    vnro_pcg32_next_float(vnro_pcg32* r);
void     vnro_pcg32_advance(vnro_pcg32* r, int64_t delta);

/* ---- rendering --------------------------------------------------------- */
typedef struct {
  /* framebuffer */
  int width, height;
  int frame_index; /* 1 for the first frame */
  /* camera (world space) */
  float cam_from[3], cam_at[3], cam_up[3], fovy;
  /* object->world transform: columns vx,vy,vz and translation p */
  float xfm[12];
  /* volume */
  int   vol_dims[3];
  float bbox_lo[3], bbox_hi[3]; /* object-space clip box */
  float sampling_rate;
  /* macrocell */
  int   mc_dims[3];
  float mc_spacings[3];
  const float* mc_max_opacity;
  /* tfn */
  vnro_tfn tfn;
  /* tiling: only pixels with pixel_lo <= index < pixel_hi are rendered (others untouched) */
  uint32_t pixel_lo, pixel_hi;
  /* shading (appended last; zero = NO_SHADING, i.e. rendering modes 4 / 5) */
  int   shading_mode;  /* 1 = GRADIENT_SHADING (modes 7 / 8; method_raymarching.cu:446-454, 719-726, 773-788);
                        * 2 = SINGLE_SHADE_HEURISTIC (modes 10 / 11; :455-484 monolithic, :789-833, 877-900 streaming + shadow pass) */
  float light_dir[3];  /* LaunchParams::light_directional_dir (instantvnr_types.h:148) AFTER the flip of renderer.cpp:98-101 */
  float density_scale; /* DeviceVolume::density_scale (vnrRendererSetVolumeDensityScale); path tracing only; 0 is read as 1 */
  /* a rank's share of a tile-sharded frame (not in the reference, which is single-GPU): with il_parts > 1 only pixels whose block
   * (index / il_block) % il_parts == il_part are rendered.  Pixels keep their GLOBAL index (random sequence, method_raymarching.cu:851), so a
   * share is the same rays as in the whole frame; what changes is which rays share an iteration count.  Streaming marcher only. */
  uint32_t il_block, il_parts, il_part;
} vnro_scene;

/* shade_scivis_light (core/renderer/raytracing.h:214-246) with mat_gradient_shading {.6, .9, .4, 40} and
 * light_directional_rgb = 1 (instantvnr_types.h:142,147); all vectors in world space; out[3] */
void vnro_shade_scivis_light(const float ray_dir[3], const float normal[3], const float albedo[3], const float light_dir[3],
                             float out[3]);

/* batch value callback: values[i] = f(coords[i]) for object-space coords */
typedef void (*vnro_value_fn)(void* user, const float* coords, size_t n, float* values);

typedef struct {
  uint64_t n_samples;     /* live samples evaluated */
  uint64_t n_slots;       /* N_ITERS * alive rays summed over iterations (reference's inferred slots) */
  uint32_t n_iterations;
  uint32_t n_rays_hit;
} vnro_render_stats;

/* mode 5 sample-streaming loop.  accumulation/frame: [w*h][4] */
void vnro_render_streaming(const vnro_scene* s, int n_iters, vnro_value_fn fn, void* user,
                           float* accumulation, float* frame, vnro_render_stats* stats);

/* sample-streaming path tracer (rendering mode 14; core/renderer/method_pathtracing.cu:532-813): delta tracking with the macrocell
 * majorants, one volume sample per alive ray and iteration through `fn`, single scattering towards the directional light plus
 * uniform bounces, Russian roulette after 4 scatters; pixel = (L, 1) through writePixelColor.  stats->n_samples = evaluations. */
void vnro_render_pathtracing(const vnro_scene* s, vnro_value_fn fn, void* user, float* accumulation, float* frame,
                             vnro_render_stats* stats);

/* the same estimator in one loop per pixel on a dense volume (rendering mode 13; method_pathtracing.cu:258-292, 420-510) */
void vnro_render_pathtracing_monolithic(const vnro_scene* s, const float* vol, int row_lo, int row_hi, float* accumulation, float* frame);

/* monolithic ground-truth marcher (mode 4 semantics) on a dense fp32 volume;
 * rows [row_lo,row_hi) only, so callers can thread over scanlines */
void vnro_render_monolithic(const vnro_scene* s, const float* vol, int row_lo, int row_hi,
                            float* accumulation, float* frame);

/* DDA traversal trace for tests: cells visited + [t0,t1] per cell (grid space already applied) */
size_t vnro_dda_trace(const float org[3], const float dir[3], float t_min, float t_max,
                      const int grid[3], int* cells, float* ts, size_t max_cells);

/* ---- metrics ------------------------------------------------------------ */
void vnro_generate_grid_coords(const int lower[3], const int size[3], const float rdims[3], float* coords);
double vnro_psnr(const float* pred, const float* ref, size_t n, float ref_min, float ref_max);

/* ---- out-of-core training sampler (core/samplers/neural_sampler.cpp:377-668, 1043-1127) ------------------ */
/* value types: 0 u8, 1 i8, 2 u16, 3 i16, 4 u32, 5 i32, 8 f32, 12 f64 (read_typed_pointer, :143-156) */
typedef struct {
  int dims[3], type;
  uint32_t elem;
  int block_dims[3], ghost_dims[3], index_space[3];
  uint64_t block_size_aligned;
} vnro_ooc_geometry;
typedef struct {  /* RandomBuffer::Block (:491-497) */
  int index[3];
  uint64_t offset, length;
  int bounds_lo[3], bounds_hi[3], ghost_lo[3], ghost_hi[3];
} vnro_ooc_block;
/* RandomBuffer ctor (:531-552); returns 0, or -1 for an unsupported type */
int  vnro_ooc_geometry_make(const int dims[3], int type, vnro_ooc_geometry* g);
/* submit_one_job (:579-636) with the file held in memory: fills *b and copies the slab (with ghosts) into block_data */
int  vnro_ooc_load_block(const vnro_ooc_geometry* g, const uint8_t* file, const int block_index[3], vnro_ooc_block* b,
                         uint8_t* block_data);
/* OutOfCoreSampler::sample (:1066-1120) for given random numbers: r_coords[3n], r_bidx[n], r_vidx[n] in [0,1).
 * block_data = n_blocks x block_size_aligned.  Returns the number of samples whose pick fell outside its range (the
 * reference throws there; the pick is clamped and counted). */
size_t vnro_ooc_sample(const vnro_ooc_geometry* g, const vnro_ooc_block* blocks, const uint8_t* block_data, uint64_t n_blocks,
                       float range_lo, float range_hi, const float* r_coords, const float* r_bidx, const float* r_vidx, size_t n,
                       const float lower[3], const float upper[3], float* coords, float* values);
/* sample_streaming_grid (:967-1035, trilinear = false) with the file held in memory; every grid point reads the file voxel it
 * falls into (the reference indexes the block it read and is only defined when grid dims == file dims) */
void vnro_ooc_sample_grid(const vnro_ooc_geometry* g, const uint8_t* file, float range_lo, float range_hi, const int origin[3],
                          const int size[3], const float spacing[3], float* values);

#ifdef __cplusplus
}
#endif
#endif
