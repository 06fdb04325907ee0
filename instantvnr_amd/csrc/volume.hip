// volume.hip — ground-truth volume / sampler, macrocell kernels, transfer function, NeuralVolume host logic.
// See volume.h for the reference mapping.
#include "volume.h"

#include <algorithm>
#include <chrono>
#include <cstdlib>
#include <ctime>
#include <fstream>
#include <thread>

#include "dist.h"
#include "sampling_device.h"
#include "scene.h"

namespace vnr {

// ================================================================================================ TfnObject
void TfnObject::set(const TransferFunctionData& t, float data_lo, float data_hi, hipStream_t s)
{
  // object.cpp:321-348
  std::vector<vec4f> c(t.color.size());
  for (size_t i = 0; i < c.size(); ++i) c[i] = {t.color[i].x, t.color[i].y, t.color[i].z, 1.0f};
  std::vector<float> a(t.alpha.size());
  for (size_t i = 0; i < a.size(); ++i) a[i] = t.alpha[i].y;
  if (!c.empty()) { colors_.resize(c.size()); colors_.upload(c.data(), c.size(), s); n_colors_ = (int)c.size(); }
  if (!a.empty()) { alphas_.resize(a.size()); alphas_.upload(a.data(), a.size(), s); n_alphas_ = (int)a.size(); }
  if (t.range_set && !(t.range_lo > t.range_hi)) {
    hi_ = std::min(data_hi, t.range_hi);
    lo_ = std::max(data_lo, t.range_lo);
  }
  rcp_ = 1.0f / (hi_ - lo_);
  VNR_HIP_CHECK(hipStreamSynchronize(s));  // host staging vectors go out of scope
}

DeviceTfn TfnObject::view() const
{
  DeviceTfn d;
  d.colors = colors_.ptr; d.alphas = alphas_.ptr; d.n_colors = n_colors_; d.n_alphas = n_alphas_;
  d.range_lo = lo_; d.range_hi = hi_; d.range_rcp_norm = rcp_;
  return d;
}

// ================================================================================================ MacroCell kernels
// float atomic min/max through integer atomics (core/instantvnr_types.h:185-199)
__device__ __forceinline__ void atomic_min_f32(float* addr, float v)
{
  if (!signbit(v)) atomicMin((int*)addr, __float_as_int(v));
  else atomicMax((unsigned int*)addr, __float_as_uint(v));
}
__device__ __forceinline__ void atomic_max_f32(float* addr, float v)
{
  if (!signbit(v)) atomicMax((int*)addr, __float_as_int(v));
  else atomicMin((unsigned int*)addr, __float_as_uint(v));
}

// macrocell.cu:11-40
__device__ __forceinline__ void update_single_macrocell(int vx, int vy, int vz, vec3i mc, float* __restrict__ cells, float value)
{
  const int cx = vx >> kMacrocellSizeMip, cy = vy >> kMacrocellSizeMip, cz = vz >> kMacrocellSizeMip;
  if (cx < 0 || cx >= mc.x) return;
  if (cy < 0 || cy >= mc.y) return;
  if (cz < 0 || cz >= mc.z) return;
  const uint32_t idx = cx + cy * mc.x + cz * mc.y * mc.x;
  atomic_min_f32(cells + 2 * (size_t)idx, value - 1.0f);
  atomic_max_f32(cells + 2 * (size_t)idx + 1, value + 1.0f);
}

__device__ __forceinline__ void update_voxel_and_neighbours(int x, int y, int z, vec3i mc, float* __restrict__ cells, float value)
{
  const int sx = (x % kMacrocellSize) == 0 ? -1 : ((x % kMacrocellSize) == (kMacrocellSize - 1) ? 1 : 0);
  const int sy = (y % kMacrocellSize) == 0 ? -1 : ((y % kMacrocellSize) == (kMacrocellSize - 1) ? 1 : 0);
  const int sz = (z % kMacrocellSize) == 0 ? -1 : ((z % kMacrocellSize) == (kMacrocellSize - 1) ? 1 : 0);
  update_single_macrocell(x, y, z, mc, cells, value);
  update_single_macrocell(x + sx, y, z, mc, cells, value);
  update_single_macrocell(x, y + sy, z, mc, cells, value);
  update_single_macrocell(x + sx, y + sy, z, mc, cells, value);
  update_single_macrocell(x, y, z + sz, mc, cells, value);
  update_single_macrocell(x + sx, y, z + sz, mc, cells, value);
  update_single_macrocell(x, y + sy, z + sz, mc, cells, value);
  update_single_macrocell(x + sx, y + sy, z + sz, mc, cells, value);
}

// macrocell.cu:42-73
__global__ void update_macrocell_explicit_kernel(uint32_t n, const float* __restrict__ coords, const float* __restrict__ values,
                                                 vec3i dims, vec3i mc, float* __restrict__ cells)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float value = values[i];
  int v[3];
  const int d[3] = {dims.x, dims.y, dims.z};
#pragma unroll
  for (int k = 0; k < 3; ++k) {
    const float f = __builtin_floorf(coords[3 * (size_t)i + k] * (float)d[k]);
    uint32_t u = f <= 0.0f ? 0u : (f >= 4294967040.0f ? 0xffffffffu : (uint32_t)f);
    if (u > (uint32_t)(d[k] - 1)) u = (uint32_t)(d[k] - 1);
    v[k] = (int)u;
  }
  update_voxel_and_neighbours(v[0], v[1], v[2], mc, cells, value);
}

// macrocell.cu:75-111 (one launch per z-slab there; one launch for the whole volume here)
__global__ void update_macrocell_implicit_kernel(uint64_t n, vec3i dims, const float* __restrict__ vol, vec3i mc, float* __restrict__ cells)
{
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint64_t stride = (uint64_t)dims.x * dims.y;
  const uint32_t x = (uint32_t)(idx % dims.x), y = (uint32_t)((idx % stride) / dims.x), z = (uint32_t)(idx / stride);
  const float fx = ((float)x + 0.5f) / (float)dims.x, fy = ((float)y + 0.5f) / (float)dims.y, fz = ((float)z + 0.5f) / (float)dims.z;
  const float value = tex3d(vol, dims, fx, fy, fz);
  update_voxel_and_neighbours((int)x, (int)y, (int)z, mc, cells, value);
}

// Same result as update_macrocell_implicit_kernel (min/max are order independent) with 16x fewer atomics:
// the 16 consecutive-x voxels of a macrocell row share their cell, so they are min/max-reduced across a
// 16-lane group first.  Requires dims.x % 16 == 0 (groups never straddle a row).
__device__ __forceinline__ void update_cell_range(int cx, int cy, int cz, vec3i mc, float* __restrict__ cells, float lo, float hi)
{
  if (cx < 0 || cx >= mc.x || cy < 0 || cy >= mc.y || cz < 0 || cz >= mc.z) return;
  const uint32_t idx = cx + cy * mc.x + cz * mc.y * mc.x;
  atomic_min_f32(cells + 2 * (size_t)idx, lo - 1.0f);
  atomic_max_f32(cells + 2 * (size_t)idx + 1, hi + 1.0f);
}

__global__ void update_macrocell_implicit_grouped_kernel(uint64_t n, vec3i dims, const float* __restrict__ vol, vec3i mc, float* __restrict__ cells)
{
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint64_t stride = (uint64_t)dims.x * dims.y;
  const int x = (int)(idx % dims.x), y = (int)((idx % stride) / dims.x), z = (int)(idx / stride);
  const float fx = ((float)x + 0.5f) / (float)dims.x, fy = ((float)y + 0.5f) / (float)dims.y, fz = ((float)z + 0.5f) / (float)dims.z;
  const float value = tex3d(vol, dims, fx, fy, fz);
  float mn = value, mx = value;
#pragma unroll
  for (int d = 8; d > 0; d >>= 1) {
    mn = fminf(mn, __shfl_xor(mn, d, 16));
    mx = fmaxf(mx, __shfl_xor(mx, d, 16));
  }
  const int sx = (x % kMacrocellSize) == 0 ? -1 : ((x % kMacrocellSize) == (kMacrocellSize - 1) ? 1 : 0);
  const int sy = (y % kMacrocellSize) == 0 ? -1 : ((y % kMacrocellSize) == (kMacrocellSize - 1) ? 1 : 0);
  const int sz = (z % kMacrocellSize) == 0 ? -1 : ((z % kMacrocellSize) == (kMacrocellSize - 1) ? 1 : 0);
  const int cx = x >> kMacrocellSizeMip;
  const int cy0 = y >> kMacrocellSizeMip, cy1 = (y + sy) >> kMacrocellSizeMip;
  const int cz0 = z >> kMacrocellSizeMip, cz1 = (z + sz) >> kMacrocellSizeMip;
  const bool ny = sy != 0 && y + sy >= 0, nz = sz != 0 && z + sz >= 0;
  if ((x & (kMacrocellSize - 1)) == 0) {  // group leader: the row's own cell (and its y/z neighbours)
    update_cell_range(cx, cy0, cz0, mc, cells, mn, mx);
    if (ny) update_cell_range(cx, cy1, cz0, mc, cells, mn, mx);
    if (nz) update_cell_range(cx, cy0, cz1, mc, cells, mn, mx);
    if (ny && nz) update_cell_range(cx, cy1, cz1, mc, cells, mn, mx);
  }
  if (sx != 0 && x + sx >= 0) {           // first / last voxel of the row segment also feeds the x neighbour
    const int cxn = (x + sx) >> kMacrocellSizeMip;
    update_cell_range(cxn, cy0, cz0, mc, cells, value, value);
    if (ny) update_cell_range(cxn, cy1, cz0, mc, cells, value, value);
    if (nz) update_cell_range(cxn, cy0, cz1, mc, cells, value, value);
    if (ny && nz) update_cell_range(cxn, cy1, cz1, mc, cells, value, value);
  }
}

// macrocell.cu:153-193 (TFN alphas staged in LDS; any block size)
__global__ void macrocell_max_opacity_kernel(uint32_t n_cells, DeviceTfn tfn, const float* __restrict__ range, float* __restrict__ out)
{
  extern __shared__ float s_alphas[];
  for (int j = threadIdx.x; j < tfn.n_alphas; j += blockDim.x) s_alphas[j] = tfn.alphas[j];
  __syncthreads();
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n_cells) return;
  const float rx = range[2 * (size_t)i] + 1.0f, ry = range[2 * (size_t)i + 1] - 1.0f;
  const int len = tfn.n_alphas;
  const float lower = (clampf(rx, tfn.range_lo, tfn.range_hi) - tfn.range_lo) * tfn.range_rcp_norm;
  const float upper = (clampf(ry, tfn.range_lo, tfn.range_hi) - tfn.range_lo) * tfn.range_rcp_norm;
  const float fl = __builtin_floorf(__builtin_fmaf(lower, (float)(len - 1), 0.5f)) - 1.0f;
  const float fu = __builtin_floorf(__builtin_fmaf(upper, (float)(len - 1), 0.5f)) + 1.0f;
  uint32_t il = fl <= 0.0f ? 0u : (uint32_t)fl, iu = fu <= 0.0f ? 0u : (uint32_t)fu;
  if (il > (uint32_t)(len - 1)) il = (uint32_t)(len - 1);
  if (iu > (uint32_t)(len - 1)) iu = (uint32_t)(len - 1);
  float op = 0.0f;
  for (uint32_t j = il; j <= iu; ++j) op = fmaxf(op, s_alphas[j]);
  out[i] = op;
}

void MacroCell::set_shape(vec3i vd)
{
  MacroCell& t = target();
  t.volume_dims_ = vd;
  t.dims_ = {(vd.x + kMacrocellSize - 1) / kMacrocellSize, (vd.y + kMacrocellSize - 1) / kMacrocellSize,
             (vd.z + kMacrocellSize - 1) / kMacrocellSize};
  t.spacings_ = {(float)kMacrocellSize / (float)vd.x, (float)kMacrocellSize / (float)vd.y, (float)kMacrocellSize / (float)vd.z};
}

void MacroCell::allocate(hipStream_t s)
{
  external_ = nullptr;
  const size_t n = (size_t)dims_.x * dims_.y * dims_.z;
  value_range_.resize(2 * n);
  value_range_.zero(s);
  max_opacity_.resize(n);
  max_opacity_.zero(s);
}

void MacroCell::compute_everything(const float* d_volume, hipStream_t s)
{
  const vec3i vd = volume_dims();
  const uint64_t n = (uint64_t)vd.x * vd.y * vd.z;
  if (vd.x % kMacrocellSize == 0)
    update_macrocell_implicit_grouped_kernel<<<div_round_up(n, 256), 256, 0, s>>>(n, vd, d_volume, dims(), d_value_range());
  else
    update_macrocell_implicit_kernel<<<div_round_up(n, 256), 256, 0, s>>>(n, vd, d_volume, dims(), d_value_range());
  VNR_HIP_CHECK(hipGetLastError());
}

void MacroCell::update_explicit(const float* d_coords, const float* d_values, size_t n, hipStream_t s)
{
  if (n == 0) return;
  update_macrocell_explicit_kernel<<<div_round_up(n, 256), 256, 0, s>>>((uint32_t)n, d_coords, d_values, volume_dims(), dims(), d_value_range());
  VNR_HIP_CHECK(hipGetLastError());
}

void MacroCell::update_max_opacity(const DeviceTfn& tfn, hipStream_t s)
{
  if (tfn.n_alphas <= 0 || !allocated()) return;  // macrocell.cu:245
  const uint32_t n = (uint32_t)n_cells();
  macrocell_max_opacity_kernel<<<div_round_up(n, 256), 256, (size_t)tfn.n_alphas * sizeof(float), s>>>(n, tfn, d_value_range(), d_max_opacity());
  VNR_HIP_CHECK(hipGetLastError());
}

void MacroCell::upload_value_range(const void* host, size_t bytes, hipStream_t s)
{
  if (bytes != n_cells() * 2 * sizeof(float)) throw std::runtime_error("macrocell data has the wrong size");
  VNR_HIP_CHECK(hipMemcpyAsync(d_value_range(), host, bytes, hipMemcpyHostToDevice, s));
  VNR_HIP_CHECK(hipStreamSynchronize(s));
}

// ================================================================================================ sampler kernels
struct Pcg32Dev {
  uint64_t state, inc;
  __device__ Pcg32Dev(uint64_t initstate, uint64_t initseq)
  {
    state = 0u; inc = (initseq << 1u) | 1u; next_uint(); state += initstate; next_uint();
  }
  __device__ uint32_t next_uint()
  {
    const uint64_t old = state;
    state = old * 0x5851f42d4c957f2dULL + inc;
    const uint32_t xs = (uint32_t)(((old >> 18u) ^ old) >> 27u), rot = (uint32_t)(old >> 59u);
    return (xs >> rot) | (xs << ((~rot + 1u) & 31));
  }
  __device__ float next_float() { const uint32_t u = (next_uint() >> 9) | 0x3f800000u; return __uint_as_float(u) - 1.0f; }
  __device__ void advance(uint64_t delta)
  {
    uint64_t cm = 0x5851f42d4c957f2dULL, cp = inc, am = 1u, ap = 0u;
    while (delta > 0) {
      if (delta & 1) { am *= cm; ap = ap * cm + cp; }
      cp = (cm + 1) * cp; cm *= cm; delta >>= 1;
    }
    state = am * state + ap;
  }
};

// neural_sampler.cu:130-164: p = lower + u * (upper - lower), value = tex3D(p) (cell-centred, clamp).
// The random stream is pcg32(seed 1337) like tcnn's generate_random_uniform; element e of the call draws the
// (offset + e)-th float of the stream (tcnn's thread-to-element mapping is EXTERNAL and not reproduced).
__global__ void take_samples_kernel(uint32_t n, uint64_t seed, uint64_t stream, uint64_t offset, vec3f lower, vec3f scale,
                                    const float* __restrict__ vol, vec3i dims, float* __restrict__ coords, float* __restrict__ values)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  Pcg32Dev rng(seed, stream);
  rng.advance(offset + 3ull * i);
  const float ux = rng.next_float(), uy = rng.next_float(), uz = rng.next_float();
  const float px = lower.x + ux * scale.x, py = lower.y + uy * scale.y, pz = lower.z + uz * scale.z;
  coords[3 * (size_t)i + 0] = px; coords[3 * (size_t)i + 1] = py; coords[3 * (size_t)i + 2] = pz;
  values[i] = tex3d(vol, dims, px, py, pz);
}

__global__ void sample_kernel(size_t n, const float* __restrict__ vol, vec3i dims, const float* __restrict__ coords,
                              float* __restrict__ values, int nodal)
{
  const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const float px = coords[3 * i], py = coords[3 * i + 1], pz = coords[3 * i + 2];
  values[i] = nodal ? sample_volume_nodal(vol, dims, px, py, pz) : tex3d(vol, dims, px, py, pz);
}

// core/network.cu:51-68 generate_coords
__global__ void generate_coords_kernel(uint32_t n, vec3i lower, vec3i size, vec3f rdims, float* __restrict__ coords)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint64_t stride = (uint64_t)size.x * size.y;
  const int x = lower.x + (int)(i % size.x), y = lower.y + (int)((i % stride) / size.x), z = lower.z + (int)(i / stride);
  coords[3 * (size_t)i + 0] = ((float)x + 0.5f) * rdims.x;
  coords[3 * (size_t)i + 1] = ((float)y + 0.5f) * rdims.y;
  coords[3 * (size_t)i + 2] = ((float)z + 0.5f) * rdims.z;
}

// seeded gradient-noise fBm (synthetic stand-in for the 1024^3 / 4096^3 volumes of BASELINE C4 / C5)
__device__ __forceinline__ uint32_t hash3(uint32_t x, uint32_t y, uint32_t z, uint32_t seed)
{
  uint32_t h = seed ^ (x * 0x9e3779b1u) ^ (y * 0x85ebca77u) ^ (z * 0xc2b2ae3du);
  h ^= h >> 16; h *= 0x7feb352du; h ^= h >> 15; h *= 0x846ca68bu; h ^= h >> 16;
  return h;
}
__device__ __forceinline__ float grad3(uint32_t h, float x, float y, float z)
{
  const uint32_t g = h & 15u;
  const float u = g < 8 ? x : y;
  const float v = g < 4 ? y : ((g == 12 || g == 14) ? x : z);
  return ((g & 1) ? -u : u) + ((g & 2) ? -v : v);
}
__device__ __forceinline__ float fade(float t) { return t * t * t * (t * (t * 6.0f - 15.0f) + 10.0f); }
__device__ float perlin3(float x, float y, float z, uint32_t seed)
{
  const float fx = floorf(x), fy = floorf(y), fz = floorf(z);
  const uint32_t ix = (uint32_t)(int)fx, iy = (uint32_t)(int)fy, iz = (uint32_t)(int)fz;
  const float rx = x - fx, ry = y - fy, rz = z - fz;
  const float u = fade(rx), v = fade(ry), w = fade(rz);
  float c[8];
#pragma unroll
  for (int k = 0; k < 8; ++k) {
    const uint32_t dx = k & 1, dy = (k >> 1) & 1, dz = (k >> 2) & 1;
    c[k] = grad3(hash3(ix + dx, iy + dy, iz + dz, seed), rx - (float)dx, ry - (float)dy, rz - (float)dz);
  }
  const float x00 = c[0] + u * (c[1] - c[0]), x10 = c[2] + u * (c[3] - c[2]);
  const float x01 = c[4] + u * (c[5] - c[4]), x11 = c[6] + u * (c[7] - c[6]);
  const float y0 = x00 + v * (x10 - x00), y1 = x01 + v * (x11 - x01);
  return y0 + w * (y1 - y0);
}
__global__ void perlin_volume_kernel(uint64_t n, vec3i dims, uint32_t seed, int octaves, float base_freq, float* __restrict__ out)
{
  const uint64_t idx = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (idx >= n) return;
  const uint64_t stride = (uint64_t)dims.x * dims.y;
  const uint32_t x = (uint32_t)(idx % dims.x), y = (uint32_t)((idx % stride) / dims.x), z = (uint32_t)(idx / stride);
  const float px = ((float)x + 0.5f) / (float)dims.x, py = ((float)y + 0.5f) / (float)dims.y, pz = ((float)z + 0.5f) / (float)dims.z;
  float sum = 0.0f, amp = 1.0f, norm = 0.0f, f = base_freq;
  for (int o = 0; o < octaves; ++o) {
    sum += amp * perlin3(px * f, py * f, pz * f, seed + 0x632be5abu * (uint32_t)o);
    norm += amp; amp *= 0.5f; f *= 2.0f;
  }
  // a soft spherical envelope leaves genuinely empty space near the corners, like real scans
  const float dx = px - 0.5f, dy = py - 0.5f, dz = pz - 0.5f;
  const float r2 = (dx * dx + dy * dy + dz * dz) * 4.0f;  // 1 at the face centres
  const float env = fminf(fmaxf(1.35f - r2, 0.0f), 1.0f);
  const float v = (0.5f + 0.85f * sum / norm) * env;
  out[idx] = fminf(fmaxf(v, 0.0f), 1.0f);
}

// ================================================================================================ SimpleVolume
void SimpleVolume::set_transfer_function(const TransferFunctionData& t, hipStream_t s)
{
  tfn_.set(t, 0.0f, 1.0f, s);
  mc_.update_max_opacity(tfn_.view(), s);
}

void SimpleVolume::finish_load(hipStream_t s)
{
  // core/sampler.cu:5-17 + neural_sampler.cpp:1270: object->world = translate(-dims/2) * scale(dims)
  const vec3f d = {(float)desc.dims.x, (float)desc.dims.y, (float)desc.dims.z};
  transform = {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}, {-d.x / 2.0f, -d.y / 2.0f, -d.z / 2.0f}};
  clipbox = {{0, 0, 0}, {1, 1, 1}};
  mc_.set_shape(desc.dims);
  mc_.allocate(s);
  mc_.compute_everything(data_.ptr, s);
  VNR_HIP_CHECK(hipStreamSynchronize(s));
}

template <typename T>
static void convert_chunk(const uint8_t* src, float* dst, size_t lo, size_t hi, bool swap, float vmin, float vmax, bool pass_minmax,
                          double* omin, double* omax)
{
  double mn = 1e300, mx = -1e300;
  for (size_t i = lo; i < hi; ++i) {
    T v;
    uint8_t b[sizeof(T)];
    std::memcpy(b, src + i * sizeof(T), sizeof(T));
    if (swap) std::reverse(b, b + sizeof(T));
    std::memcpy(&v, b, sizeof(T));
    const float f = (float)v;
    if (pass_minmax) { mn = std::min(mn, (double)v); mx = std::max(mx, (double)v); }
    else {
      const float nv = (f - vmin) / (vmax - vmin);  // neural_sampler.cpp:176-210 convert_volume
      dst[i] = nv < 0.0f ? 0.0f : (nv > 1.0f ? 1.0f : nv);
    }
  }
  if (pass_minmax) { *omin = mn; *omax = mx; }
}

static size_t type_size(int type)
{
  switch (type) {
  case 0: case 1: return 1;
  case 2: case 3: return 2;
  case 4: case 5: case 8: return 4;
  case 6: case 7: case 12: return 8;
  default: throw std::runtime_error("unknown data type");
  }
}

// StaticSampler::load (neural_sampler.cpp:223-288): typed voxels -> fp32 normalised to [0,1]; an empty range (lo > hi) is
// replaced by the min/max of the data
static std::vector<float> normalise_volume(const void* data, vec3i dims, int type, float& range_lo, float& range_hi, bool big_endian)
{
  if (dims.x <= 0 || dims.y <= 0 || dims.z <= 0) throw std::runtime_error("invalid volume dims");
  const size_t count = (size_t)dims.x * dims.y * dims.z;
  const uint8_t* src = (const uint8_t*)data;
  std::vector<float> out(count);
  const unsigned nt = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  auto run = [&](bool minmax, float lo, float hi, double* gmin, double* gmax) {
    std::vector<std::thread> th;
    std::vector<double> mins(nt, 1e300), maxs(nt, -1e300);
    for (unsigned t = 0; t < nt; ++t) {
      const size_t a = count * t / nt, b = count * (t + 1) / nt;
      th.emplace_back([&, a, b, t]() {
        switch (type) {
        case 0: convert_chunk<uint8_t>(src, out.data(), a, b, false, lo, hi, minmax, &mins[t], &maxs[t]); break;
        case 1: convert_chunk<int8_t>(src, out.data(), a, b, false, lo, hi, minmax, &mins[t], &maxs[t]); break;
        case 2: convert_chunk<uint16_t>(src, out.data(), a, b, big_endian, lo, hi, minmax, &mins[t], &maxs[t]); break;
        case 3: convert_chunk<int16_t>(src, out.data(), a, b, big_endian, lo, hi, minmax, &mins[t], &maxs[t]); break;
        case 4: convert_chunk<uint32_t>(src, out.data(), a, b, big_endian, lo, hi, minmax, &mins[t], &maxs[t]); break;
        case 5: convert_chunk<int32_t>(src, out.data(), a, b, big_endian, lo, hi, minmax, &mins[t], &maxs[t]); break;
        case 8: convert_chunk<float>(src, out.data(), a, b, big_endian, lo, hi, minmax, &mins[t], &maxs[t]); break;
        case 12: convert_chunk<double>(src, out.data(), a, b, big_endian, lo, hi, minmax, &mins[t], &maxs[t]); break;
        default: break;
        }
      });
    }
    for (auto& t : th) t.join();
    if (minmax) { *gmin = *std::min_element(mins.begin(), mins.end()); *gmax = *std::max_element(maxs.begin(), maxs.end()); }
  };
  (void)type_size(type);
  if (range_lo > range_hi) {  // minmax.is_empty(): compute from the data (neural_sampler.cpp:252-265)
    double mn, mx;
    run(true, 0, 1, &mn, &mx);
    range_lo = (float)mn; range_hi = (float)mx;
  }
  run(false, range_lo, range_hi, nullptr, nullptr);
  return out;
}

static std::vector<char> read_raw_file(const std::string& filename, vec3i dims, int type, size_t offset)
{
  // the description is checked against the FILE before anything is allocated: a scene with a wrong dimension must fail by name, not by
  // a zero-filled vector of the size it claims (found by tests/test_gpu_fuzz.py's damaged scenes)
  if (dims.x <= 0 || dims.y <= 0 || dims.z <= 0)
    throw std::runtime_error("volume dimensions must be positive: " + std::to_string(dims.x) + " x " + std::to_string(dims.y) + " x " + std::to_string(dims.z));
  const unsigned __int128 wide = (unsigned __int128)(uint32_t)dims.x * (uint32_t)dims.y * (uint32_t)dims.z * (unsigned)type_size(type);
  std::ifstream f(filename, std::ios::binary | std::ios::ate);
  if (!f) throw std::runtime_error("cannot open volume file: " + filename);
  const std::streamoff file_size = f.tellg();
  if (file_size < 0 || (unsigned __int128)offset + wide > (unsigned __int128)(uint64_t)file_size)
    throw std::runtime_error("volume file too short: " + filename + " has " + std::to_string((long long)file_size) + " bytes, the description needs " +
                             (wide > (unsigned __int128)UINT64_MAX ? std::string("more than 2^64") : std::to_string((uint64_t)wide)) + " from offset " + std::to_string(offset));
  const size_t bytes = (size_t)wide;
  f.seekg((std::streamoff)offset);
  std::vector<char> buf(bytes);
  if (!f.read(buf.data(), (std::streamsize)bytes)) throw std::runtime_error("volume file too short: " + filename);
  return buf;
}

void SimpleVolume::load_host(const void* data, vec3i dims, int type, float range_lo, float range_hi, bool big_endian)
{
  const std::vector<float> out = normalise_volume(data, dims, type, range_lo, range_hi, big_endian);
  unnormalized_lo = range_lo; unnormalized_hi = range_hi;
  desc.dims = dims; desc.type = 8; desc.range_lo = 0.0f; desc.range_hi = 1.0f;
  hipStream_t s = Runtime::get().stream;
  ooc_.reset();
  steps_.clear();
  current_step_ = 0;
  data_.resize(out.size());
  data_.upload(out.data(), out.size(), s);
  VNR_HIP_CHECK(hipStreamSynchronize(s));
  finish_load(s);
}

void SimpleVolume::load_scene(const SceneVolume& sc, const std::string& mode, bool save_volume)
{
  // SimpleVolume::load -> Sampler::load (core/sampler.cu:5-17, neural_sampler.cpp:1205-1271)
  if (sc.data.empty()) throw std::runtime_error("the scene names no volume file");
  if (mode == "GPU") {
    // StaticSampler (neural_sampler.cu:86-121): every time step is normalised on load; here they all stay in HBM, so
    // switching the time step is a pointer change plus the macrocell pass
    float lo = sc.range_lo, hi = sc.range_hi;
    {
      const std::vector<char> raw = read_raw_file(sc.data[0].filename, sc.dims, sc.type, sc.data[0].offset);
      load_host(raw.data(), sc.dims, sc.type, lo, hi, sc.data[0].bigendian);
    }
    if (save_volume) {  // neural_sampler.cu:101-108: the normalised fp32 voxels of time step 0
      std::vector<float> h(data_.count);
      data_.download(h.data(), h.size(), Runtime::get().stream);
      std::ofstream o("reference.bin", std::ios::binary | std::ios::out);
      o.write((const char*)h.data(), (std::streamsize)(h.size() * sizeof(float)));
      if (!o) throw std::runtime_error("cannot write reference.bin");
    }
    steps_.resize(sc.data.size());
    hipStream_t s = Runtime::get().stream;
    for (size_t i = 1; i < sc.data.size(); ++i) {
      float l = sc.range_lo, h = sc.range_hi;
      const std::vector<char> raw = read_raw_file(sc.data[i].filename, sc.dims, sc.type, sc.data[i].offset);
      const std::vector<float> out = normalise_volume(raw.data(), sc.dims, sc.type, l, h, sc.data[i].bigendian);
      steps_[i].resize(out.size());
      steps_[i].upload(out.data(), out.size(), s);
      VNR_HIP_CHECK(hipStreamSynchronize(s));
      unnormalized_lo = std::min(unnormalized_lo, l);   // m_value_range_unnormalized.extend (:117-118)
      unnormalized_hi = std::max(unnormalized_hi, h);
    }
  } else if (mode == "OUT_OF_CORE") {
    if (sc.data[0].bigendian) throw std::runtime_error("only support small endian");  // neural_sampler.cpp:1050
    uint64_t ncb = 1024, nb = 0;  // neural_sampler.cpp:1054-1062
    if (const char* e = std::getenv("VNR_NUM_CONCURRENT_BLOCKS")) ncb = (uint64_t)std::max(1, std::atoi(e));
    nb = ncb * 64;
    if (const char* e = std::getenv("VNR_NUM_BLOCKS")) nb = (uint64_t)std::max(1, std::atoi(e));
    load_out_of_core(sc.data[0].filename, sc.dims, sc.type, sc.data[0].offset, sc.range_lo, sc.range_hi, ncb, nb);
    steps_.clear();
    steps_.resize(sc.data.size());
  } else if (mode == "NOTHING") {  // StaticSampler(dims, type) (neural_sampler.cu:76-84): a shape without data
    data_.resize(0);
    ooc_.reset();
    steps_.clear();
    unnormalized_lo = 0.0f; unnormalized_hi = 1.0f;
    desc.dims = sc.dims; desc.type = 8; desc.range_lo = 0.0f; desc.range_hi = 1.0f;
    const vec3f d = {(float)sc.dims.x, (float)sc.dims.y, (float)sc.dims.z};
    transform = {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}, {-d.x / 2.0f, -d.y / 2.0f, -d.z / 2.0f}};
    clipbox = {{0, 0, 0}, {1, 1, 1}};
  } else if (mode == "VIRTUAL_MEMORY" || mode.rfind("OPENVKL", 0) == 0) {
    throw std::runtime_error("training mode " + mode + " is not implemented in this build (GPU, OUT_OF_CORE and NOTHING are)");
  } else {
    throw std::runtime_error("unknown mode");  // neural_sampler.cpp:1268
  }
  current_step_ = 0;
}

void SimpleVolume::set_current_timestep(int index)
{
  // SimpleVolume::set_current_timestep (core/sampler.cu:19-26) -> StaticSampler::set_current_volume_timestamp
  // (neural_sampler.cu:123-128); samplers without time steps accept index 0 only (core/sampler.h:15)
  if (index < 0 || index >= num_timesteps()) throw std::runtime_error("time step " + std::to_string(index) + " out of range");
  if (ooc_ || !data_.ptr) {
    if (index != 0) throw std::runtime_error("only support single timestep volume");
    return;
  }
  if (index != current_step_) {
    steps_[(size_t)current_step_] = std::move(data_);
    data_ = std::move(steps_[(size_t)index]);
    current_step_ = index;
  }
  hipStream_t s = Runtime::get().stream;
  if (!mc_.is_external()) mc_.compute_everything(data_.ptr, s);
  if (!tfn_.empty()) mc_.update_max_opacity(tfn_.view(), s);
  VNR_HIP_CHECK(hipStreamSynchronize(s));
}

void SimpleVolume::load_raw_file(const std::string& filename, vec3i dims, int type, size_t offset, bool big_endian, float range_lo, float range_hi)
{
  const std::vector<char> buf = read_raw_file(filename, dims, type, offset);
  load_host(buf.data(), dims, type, range_lo, range_hi, big_endian);
}

void SimpleVolume::generate_perlin(vec3i dims, uint32_t seed, int octaves, float base_frequency)
{
  if (!Runtime::get().ready()) Runtime::get().init(-1);
  hipStream_t s = Runtime::get().stream;
  const uint64_t n = (uint64_t)dims.x * dims.y * dims.z;
  data_.resize(n);
  perlin_volume_kernel<<<div_round_up(n, 256), 256, 0, s>>>(n, dims, seed, octaves, base_frequency, data_.ptr);
  VNR_HIP_CHECK(hipGetLastError());
  desc.dims = dims; desc.type = 8; desc.range_lo = 0.0f; desc.range_hi = 1.0f;
  unnormalized_lo = 0.0f; unnormalized_hi = 1.0f;
  finish_load(s);
}

void SimpleVolume::load_out_of_core(const std::string& filename, vec3i dims, int type, size_t offset, float range_lo, float range_hi,
                                    uint64_t n_concurrent_blocks, uint64_t n_blocks)
{
  data_.resize(0);
  steps_.clear();
  current_step_ = 0;
  ooc_ = std::make_unique<OutOfCoreSampler>(filename, dims, type, offset, range_lo, range_hi, n_concurrent_blocks, n_blocks);
  unnormalized_lo = range_lo; unnormalized_hi = range_hi;
  // Sampler::load (neural_sampler.cpp:1224-1227, 1270): the neural volume's grid is capped at 1024 per axis, the
  // object -> world map is the file's
  desc.dims = {std::min(1024, dims.x), std::min(1024, dims.y), std::min(1024, dims.z)};
  desc.type = type; desc.range_lo = 0.0f; desc.range_hi = 1.0f;
  const vec3f d = {(float)dims.x, (float)dims.y, (float)dims.z};
  transform = {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}, {-d.x / 2.0f, -d.y / 2.0f, -d.z / 2.0f}};
  clipbox = {{0, 0, 0}, {1, 1, 1}};
  // no texture => no ground-truth macrocell (core/sampler.cu:11-16)
}

void SimpleVolume::take_samples(float* d_coords, float* d_values, size_t n, vec3f lower, vec3f upper, hipStream_t s)
{
  if (n == 0) return;
  if (ooc_) { ooc_->sample(d_coords, d_values, n, lower, upper, rng_seed_, rng_stream_, rng_offset_, s); return; }
  const vec3f scale = upper - lower;
  take_samples_kernel<<<div_round_up(n, 256), 256, 0, s>>>((uint32_t)n, rng_seed_, rng_stream_, rng_offset_, lower, scale, data_.ptr,
                                                          desc.dims, d_coords, d_values);
  VNR_HIP_CHECK(hipGetLastError());
  rng_offset_ += 3ull * n;
}

void SimpleVolume::take_samples_grid(float* d_coords, float* d_values, vec3i origin, vec3i size, vec3f rdims, hipStream_t s)
{
  const uint32_t n = (uint32_t)((size_t)size.x * size.y * size.z);
  if (n == 0) return;
  generate_coords_kernel<<<div_round_up(n, 256), 256, 0, s>>>(n, origin, size, rdims, d_coords);
  if (ooc_) { ooc_->sample_grid(d_values, origin, size, rdims, s); return; }
  sample(d_coords, d_values, n, false, s);
}

void SimpleVolume::sample(const float* d_coords, float* d_values, size_t n, bool nodal, hipStream_t s) const
{
  if (n == 0) return;
  if (!data_.ptr) throw std::runtime_error("this volume has no resident ground truth (training mode OUT_OF_CORE)");
  sample_kernel<<<div_round_up(n, 256), 256, 0, s>>>(n, data_.ptr, desc.dims, d_coords, d_values, nodal ? 1 : 0);
  VNR_HIP_CHECK(hipGetLastError());
}

// ================================================================================================ reductions for loss / PSNR
__global__ void error_reduce_kernel(uint32_t n, const float* __restrict__ pred, const float* __restrict__ ref, int l2,
                                    float* __restrict__ partial /* [blocks][3]: sum, min(ref), max(ref) */)
{
  __shared__ float ssum[256], smin[256], smax[256];
  float sum = 0.0f, mn = 1e20f, mx = -1e20f;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const float d = pred[i] - ref[i];
    sum += l2 ? d * d : fabsf(d);
    mn = fminf(mn, ref[i]); mx = fmaxf(mx, ref[i]);
  }
  ssum[threadIdx.x] = sum; smin[threadIdx.x] = mn; smax[threadIdx.x] = mx;
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) {
      ssum[threadIdx.x] += ssum[threadIdx.x + s];
      smin[threadIdx.x] = fminf(smin[threadIdx.x], smin[threadIdx.x + s]);
      smax[threadIdx.x] = fmaxf(smax[threadIdx.x], smax[threadIdx.x + s]);
    }
    __syncthreads();
  }
  if (threadIdx.x == 0) { partial[3 * blockIdx.x] = ssum[0]; partial[3 * blockIdx.x + 1] = smin[0]; partial[3 * blockIdx.x + 2] = smax[0]; }
}

static void reduce_errors(const float* pred, const float* ref, size_t n, bool l2, double* sum, float* mn, float* mx, hipStream_t s)
{
  const uint32_t blocks = std::min<uint32_t>(div_round_up(n, 256), 512u);
  DeviceBuffer<float> partial;
  partial.resize(3 * blocks);
  error_reduce_kernel<<<blocks, 256, 0, s>>>((uint32_t)n, pred, ref, l2 ? 1 : 0, partial.ptr);
  std::vector<float> h(3 * blocks);
  partial.download(h.data(), h.size(), s);
  for (uint32_t b = 0; b < blocks; ++b) { *sum += h[3 * b]; *mn = std::min(*mn, h[3 * b + 1]); *mx = std::max(*mx, h[3 * b + 2]); }
}

// ================================================================================================ NeuralVolume
void network_release_scratch(const Network* n);


// value ranges are stored as (min - 1, max + 1) with 0 = "nothing seen yet" (macrocell.cu:35-39, 213-219): min - 1 <= 0 and
// max + 1 >= 1, so the element-wise MIN over ranks of .x and MAX of .y merge the ranks' macrocells, untouched cells included
__global__ void macrocell_merge_kernel(vec2f* __restrict__ range_max_reduced, const vec2f* __restrict__ range_min_reduced, size_t n)
{
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
    range_max_reduced[i].x = range_min_reduced[i].x;
}

// Data-parallel step, two shapes of the same arithmetic (one step = one Adam step on the MEAN of the ranks' gradients):
//  * sharded (default; ZeRO-1 shape, SURVEY.md 8e "reduce-scatter + all-gather (each peer owns 1/8)"): a range's gradient is
//    reduce-scattered, the rank runs Adam on ITS 1/world of the range and the fp16 parameters are all-gathered in place.  The same
//    bytes on the wire as an all-reduce; the optimizer sweep, its touched-state traffic and (per rank) the live optimizer state are
//    1/world of the replicated form (C4 at 8 ranks: Adam 0.30 -> ~0.04 ms of a 0.65 ms step);
//  * replicated (VNR_AMD_DP_SHARDED=0): all-reduce, every rank updates everything.
// Both exchange with the Avg reduction (ncclAvg; fp32 sum / world before the one rounding on the shm transport), so the 1 / world
// factor is applied inside the exchange and the fp16 payload cannot overflow where a single GPU would not (it shrinks instead of
// growing with the world size); Adam is per parameter, hence the two shapes give the same bits wherever the reduction does
// (tests/test_gpu_dist.py: bit-equal on the shm transport).
struct NeuralVolume::DpState : GradExchange {
  NeuralVolume* nv = nullptr;
  bool sharded = true;
  // lo .. hi: the range; [lo, lo + world * per) is exchanged slice-wise (per: a multiple of 8 parameters), the rest (only when the
  // range does not divide: world sizes that are not a power of two) is all-reduced and updated by every rank
  struct Range { size_t lo, hi, per; hipEvent_t ready, reduced, updated, gathered; };
  std::vector<Range> ranges;     // this step's ranges in the order they became ready
  struct Events { hipEvent_t e[4]; };
  std::vector<Events> pool;      // reused step after step
  size_t used = 0;

  ~DpState() override
  {
    for (auto& p : pool) for (hipEvent_t e : p.e) (void)hipEventDestroy(e);
  }
  // diagnostics (tools/dp_probe.py on one GPU): VNR_AMD_DP_EMULATE_WORLD=8 slices the ranges as rank 0 of 8 would, so that the probe times the
  // compute side of one rank of an 8-rank step (the memsets, Adam on 1/8, the launches); the collectives then only see this rank's slice and
  // the other slices' parameters are never updated: timing only
  static size_t emulated_world()
  {
    static const int e = [] { const char* v = std::getenv("VNR_AMD_DP_EMULATE_WORLD"); return v ? std::atoi(v) : 0; }();
    // only where there is nobody to disagree with (a one-rank group): with a real group of another size the slices would be cut for the wrong
    // world and the other slices' parameters would go stale
    return e > 1 && Dist::get().world() == 1 ? (size_t)e : (size_t)Dist::get().world();
  }
  void range_ready(size_t lo, size_t hi, hipStream_t s) override
  {
    Dist& d = Dist::get();
    hipStream_t comm = d.comm_stream();
    if (used == pool.size()) {
      Events ev;
      for (hipEvent_t& e : ev.e) VNR_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
      pool.push_back(ev);
    }
    const size_t world = emulated_world();
    Range r{lo, hi, sharded ? ((hi - lo) / world) & ~(size_t)7 : 0, pool[used].e[0], pool[used].e[1], pool[used].e[2], pool[used].e[3]};
    ++used;
    // the gradient blob IS the payload (fp16, in place): nothing to pack
    VNR_HIP_CHECK(hipEventRecord(r.ready, s));
    VNR_HIP_CHECK(hipStreamWaitEvent(comm, r.ready, 0));
    uint16_t* g = nv->net_.grads_f16();
    if (r.per) d.transport().reduce_scatter(g + lo, r.per, DistDType::F16, DistOp::Avg, comm);
    const size_t rest = lo + world * r.per;
    if (rest < hi) d.transport().all_reduce(g + rest, hi - rest, DistDType::F16, DistOp::Avg, comm);
    VNR_HIP_CHECK(hipEventRecord(r.reduced, comm));
    ranges.push_back(r);
  }
  // the optimizer update of the ranges handed over so far, each behind its own exchange, and (sharded) the parameters' way back
  void update(hipStream_t s)
  {
    Dist& d = Dist::get();
    hipStream_t comm = d.comm_stream();
    Network& net = nv->net_;
    const size_t world = emulated_world(), rank = (size_t)d.rank();
    for (const Range& r : ranges) {
      VNR_HIP_CHECK(hipStreamWaitEvent(s, r.reduced, 0));
      if (!r.per) { net.optimizer_step_range(r.lo, r.hi, 1.0f, s); continue; }
      const size_t mine = r.lo + rank * r.per, rest = r.lo + world * r.per;
      // what the reduce-scatter left in the other ranks' slices is unspecified: the next step accumulates into zeros
      uint16_t* g = net.grads_f16();
      if (mine > r.lo) VNR_HIP_CHECK(hipMemsetAsync(g + r.lo, 0, (mine - r.lo) * sizeof(uint16_t), s));
      if (rest > mine + r.per) VNR_HIP_CHECK(hipMemsetAsync(g + mine + r.per, 0, (rest - mine - r.per) * sizeof(uint16_t), s));
      net.optimizer_step_range(mine, mine + r.per, 1.0f, s);
      if (rest < r.hi) net.optimizer_step_range(rest, r.hi, 1.0f, s);
      VNR_HIP_CHECK(hipEventRecord(r.updated, s));
      VNR_HIP_CHECK(hipStreamWaitEvent(comm, r.updated, 0));
      uint16_t* p = net.params_device();
      d.transport().all_gather(p + mine, p + r.lo, r.per * sizeof(uint16_t), comm);   // in place
      VNR_HIP_CHECK(hipEventRecord(r.gathered, comm));
    }
    bool any = false;
    for (const Range& r : ranges) if (r.per) { VNR_HIP_CHECK(hipStreamWaitEvent(s, r.gathered, 0)); any = true; }
    if (any && world > 1) net.set_opt_sharded(true);
    ranges.clear();
    used = 0;
  }
};

NeuralVolume::NeuralVolume()
{
  if (!Runtime::get().ready()) Runtime::get().init(-1);
  stream = Runtime::get().stream;
}

NeuralVolume::~NeuralVolume() { network_release_scratch(&net_); }

void NeuralVolume::set_transfer_function(const TransferFunctionData& t, hipStream_t s)
{
  tfn_.set(t, 0.0f, 1.0f, s);
  mc_.update_max_opacity(tfn_.view(), s);  // network.cu:751
}

void NeuralVolume::set_network(vec3i dims, const Json& config, SimpleVolume* reference, bool use_reference_macrocell)
{
  source_ = reference;
  if (use_reference_macrocell && !(source_ && source_->has_data()))  // network.cu:553-560
    fprintf(stderr, "[vnr] ground truth macrocell unavailable with this training mode\n");
  use_reference_macrocell = source_ && source_->has_data() && use_reference_macrocell;
  if (source_) {  // network.cu:563-570
    desc.dims = source_->dims();
    transform = source_->transform;
  } else {
    desc.dims = dims;
    const vec3f d = {(float)dims.x, (float)dims.y, (float)dims.z};
    transform = {{d.x, 0, 0}, {0, d.y, 0}, {0, 0, d.z}, {-d.x / 2.0f, -d.y / 2.0f, -d.z / 2.0f}};
  }
  desc.type = 8; desc.range_lo = 0.0f; desc.range_hi = 1.0f;
  if (config.contains("fvsrn")) throw std::runtime_error("fvsrn is not enabled");  // network.cu:572-578
  const uint64_t seed = init_seed ? init_seed : (uint64_t)time(nullptr);             // tcnn_network.h:209
  net_.configure(config, seed);
  replicas_synced_ = false;   // (a data-parallel run re-synchronises: the clock seed may differ between ranks)
  dp_.reset();
  net_.set_brick_resolution_cap(2u * (uint32_t)std::max(desc.dims.x, std::max(desc.dims.y, desc.dims.z)));
  train_x_.resize(batch_size_ * 3);
  train_y_.resize(batch_size_);
  test_y1_.resize(batch_size_);
  if (use_reference_macrocell) {
    mc_.set_external(&reference->macrocell());
  } else {
    mc_.set_external(nullptr);
    mc_.set_shape(desc.dims);
    mc_.allocate(stream);
  }
  VNR_HIP_CHECK(hipStreamSynchronize(stream));
}

void NeuralVolume::set_model(const Json& config)
{
  const uint64_t seed = init_seed ? init_seed : (uint64_t)time(nullptr);
  net_.configure(config, seed);
  replicas_synced_ = false;
  dp_.reset();
  net_.set_brick_resolution_cap(2u * (uint32_t)std::max(desc.dims.x, std::max(desc.dims.y, desc.dims.z)));
}

void NeuralVolume::train_begin()
{
  if (!net_.valid()) return;
  if (!source_) throw std::runtime_error("missing a reference volume");  // network.cu:233-235 prints and returns
  const vec3f lower = {0, 0, 0}, upper = {1, 1, 1};  // m_lower/m_upper = full volume (network.cu:605)
  source_->take_samples(train_x_.ptr, train_y_.ptr, batch_size_, lower, upper, stream);
  net_.forward_backward(train_x_.ptr, train_y_.ptr, batch_size_, stream);
  pending_step_ = true;
  pending_internal_ = true;
}

void NeuralVolume::forward_backward(const float* d_coords, const float* d_targets, size_t n)
{
  if (!net_.valid()) throw std::runtime_error("network is not valid");
  net_.forward_backward(d_coords, d_targets, n, stream);
  pending_step_ = true;
  pending_internal_ = false;
}

void NeuralVolume::train_end(float grad_scale, bool fast_mode)
{
  if (!pending_step_) return;
  net_.optimizer_step(grad_scale, stream);
  // network.cu:249-257, 774: the macrocell is trained online unless (fast_mode && external macrocell)
  const bool update_mc = !(fast_mode && mc_.is_external());
  if (update_mc && pending_internal_ && !mc_.is_external()) mc_.update_explicit(train_x_.ptr, train_y_.ptr, batch_size_, stream);
  pending_step_ = false;
}

void NeuralVolume::train(size_t steps, bool fast_mode)
{
  if (!net_.valid()) return;
  for (size_t i = 0; i < steps; ++i) {
    train_begin();
    train_end(1.0f, fast_mode);
  }
  if (!fast_mode) mc_.update_max_opacity(tfn_.view(), stream);  // network.cu:778
}

// ------------------------------------------------------------------------------------------------ data-parallel training

NeuralVolume::DpState& NeuralVolume::dp_state()
{
  if (!dp_) {
    dp_.reset(new DpState());
    dp_->nv = this;
    const char* e = std::getenv("VNR_AMD_DP_SHARDED");
    dp_->sharded = !e || std::atoi(e) != 0;
  }
  return *dp_;
}

void NeuralVolume::all_reduce_gradients()
{
  Dist& d = Dist::get();
  if (!d.active() || !pending_step_) return;   // a one-rank group still runs the exchange (the identity): tests
  d.transport().all_reduce(net_.grads_f16(), net_.n_params(), DistDType::F16, DistOp::Sum, stream);
}

// TrainBegin / TrainEndDataParallel form of one data-parallel step on whatever the gradient blob holds: the exchange (mean over the
// ranks) and the update in one of the two shapes of DpState, nothing overlapped.  sharded < 0: the process default.
void NeuralVolume::train_end_data_parallel(bool fast_mode, int sharded)
{
  Dist& d = Dist::get();
  if (!d.active()) { train_end(1.0f, fast_mode); return; }
  // a rank-local decision in front of collectives is a hang: the ranks agree over the control plane on whether a step is pending
  // (ADVICE r03); a mismatch is an error on every rank
  if (d.world() > 1) {
    double v[2] = {pending_step_ ? 1.0 : 0.0, pending_step_ ? 0.0 : 1.0};
    d.all_reduce_host(v, 2, DistOp::Sum);
    if (v[0] > 0.0 && v[1] > 0.0)
      throw std::runtime_error("vnrAmdNeuralVolumeTrainEndDataParallel: " + std::to_string((int)v[0]) + " rank(s) hold a pending step (TrainBegin) and " +
                               std::to_string((int)v[1]) + " do not; every rank must call TrainBegin before TrainEndDataParallel");
  }
  if (!pending_step_) return;
  DpState& dp = dp_state();
  const bool was = dp.sharded;
  if (sharded >= 0) dp.sharded = sharded != 0;
  if (!dp.sharded && net_.opt_sharded()) { dp.sharded = was; throw std::runtime_error("the optimizer state is sharded: vnrAmdNeuralVolumeSyncReplicas first"); }
  dp.ranges.clear();
  dp.used = 0;
  // the same ranges as the overlapped step: the MLP, then the hash-grid levels in buckets (finest first)
  net_.for_each_exchange_range(dp.bucket_params(), [&](size_t lo, size_t hi) { dp.range_ready(lo, hi, stream); });
  dp.update(stream);
  dp.sharded = was;
  net_.optimizer_finish_step(stream);
  const bool update_mc = !(fast_mode && mc_.is_external());
  if (update_mc && pending_internal_ && !mc_.is_external()) mc_.update_explicit(train_x_.ptr, train_y_.ptr, batch_size_, stream);
  pending_step_ = false;
}

// Makes the replicas identical, collectively: every rank ends with rank 0's parameters, optimizer state, step count and learning
// rate.  After sharded steps rank 0's own copy of the optimizer state is current for ITS slices only, so the ranks' slices are
// all-gathered first; if nothing else changed (the parameters are identical already: every call of train_data_parallel leaves them so)
// that is all there is to do.
void NeuralVolume::sync_replicas()
{
  Dist& d = Dist::get();
  if (!d.active()) { replicas_synced_ = true; return; }
  net_.ensure_training_state(stream);
  Transport& t = d.transport();
  double dirty = (!replicas_synced_ || synced_generation_ != net_.params_generation()) ? 1.0 : 0.0;
  d.all_reduce_host(&dirty, 1, DistOp::Max);
  double r0_sharded = net_.opt_sharded() ? 1.0 : 0.0;
  d.broadcast_host(&r0_sharded, sizeof(r0_sharded), 0);
  if (r0_sharded > 0.0 && d.world() > 1) {
    const size_t world = (size_t)d.world(), rank = (size_t)d.rank();
    OptState* st = net_.opt_state_device();
    net_.for_each_exchange_range(dp_state().bucket_params(), [&](size_t lo, size_t hi) {
      const size_t per = ((hi - lo) / world) & ~(size_t)7;
      if (per) t.all_gather(st + lo + rank * per, st + lo, per * sizeof(OptState), stream);
    });
  }
  if (dirty > 0.0 || r0_sharded == 0.0) {
    t.broadcast(net_.params_device(), net_.n_params() * sizeof(uint16_t), 0, stream);
    t.broadcast(net_.opt_state_device(), net_.n_params() * sizeof(OptState), 0, stream);
    // gathered state next to parameters that some rank replaced: the master copies of that rank's slices would bring its parameters
    // back with the next step.  Rank 0's parameters win, as documented: master weights restart from them (what SetParams does too)
    if (r0_sharded > 0.0) net_.reset_master_from_params(stream);
  }
  net_.set_opt_sharded(false);
  double host[2] = {(double)net_.steps(), (double)net_.learning_rate()};
  d.broadcast_host(host, sizeof(host), 0);
  net_.set_replica_state((uint64_t)host[0], (float)host[1], stream);
  if (source_) source_->set_sampler_rank(d.rank());
  VNR_HIP_CHECK(hipStreamSynchronize(stream));
  replicas_synced_ = true;
  synced_generation_ = net_.params_generation();
}

void NeuralVolume::train_data_parallel(size_t steps, bool fast_mode)
{
  Dist& d = Dist::get();
  if (!d.active()) { train(steps, fast_mode); return; }
  if (!net_.valid()) return;
  if (!source_) throw std::runtime_error("missing a reference volume");
  // A replica whose parameters changed outside an optimizer step since the last synchronisation (SetParams, SetModel, a loaded
  // params.json; a model re-initialised from the clock) would train and render something else than the others, silently.  One
  // control-plane round trip per call finds out whether ANY rank is in that state, and then all of them synchronise.
  double dirty = (!replicas_synced_ || synced_generation_ != net_.params_generation()) ? 1.0 : 0.0;
  d.all_reduce_host(&dirty, 1, DistOp::Max);
  if (dirty > 0.0) sync_replicas();
  DpState& dp = dp_state();
  if (!dp.sharded && net_.opt_sharded()) sync_replicas();   // the shape was switched between calls
  const vec3f lower = {0, 0, 0}, upper = {1, 1, 1};
  for (size_t i = 0; i < steps; ++i) {
    source_->take_samples(train_x_.ptr, train_y_.ptr, batch_size_, lower, upper, stream);
    dp.ranges.clear();
    dp.used = 0;
    net_.forward_backward(train_x_.ptr, train_y_.ptr, batch_size_, stream, &dp);
    // the update of a range waits for that range's exchange only: Adam of the first ranges runs while the last ones travel, and
    // (sharded) a range's parameters travel back while the next range is updated
    dp.update(stream);
    net_.optimizer_finish_step(stream);
    const bool update_mc = !(fast_mode && mc_.is_external());
    if (update_mc && !mc_.is_external()) mc_.update_explicit(train_x_.ptr, train_y_.ptr, batch_size_, stream);
  }
  if (!mc_.is_external()) {  // every rank has seen other samples: merge the value ranges so that all ranks skip the same cells
    const size_t n = mc_.n_cells();
    DeviceBuffer<float> tmp;
    tmp.resize(2 * n);
    VNR_HIP_CHECK(hipMemcpyAsync(tmp.ptr, mc_.d_value_range(), 2 * n * sizeof(float), hipMemcpyDeviceToDevice, stream));
    d.transport().all_reduce(tmp.ptr, 2 * n, DistDType::F32, DistOp::Min, stream);
    d.transport().all_reduce(mc_.d_value_range(), 2 * n, DistDType::F32, DistOp::Max, stream);
    macrocell_merge_kernel<<<std::min<uint32_t>(div_round_up(n, 256), 4096u), 256, 0, stream>>>((vec2f*)mc_.d_value_range(), (const vec2f*)tmp.ptr, n);
    VNR_HIP_CHECK(hipGetLastError());
    VNR_HIP_CHECK(hipStreamSynchronize(stream));   // tmp goes out of scope
    mc_.update_max_opacity(tfn_.view(), stream);
  } else if (!fast_mode) {
    mc_.update_max_opacity(tfn_.view(), stream);  // network.cu:778
  }
}

float NeuralVolume::test_loss()
{
  if (!net_.valid()) return 0.0f;
  if (!source_) throw std::runtime_error("missing a reference volume");
  const vec3f lower = {0, 0, 0}, upper = {1, 1, 1};
  source_->take_samples(train_x_.ptr, train_y_.ptr, batch_size_, lower, upper, stream);
  net_.inference(train_x_.ptr, test_y1_.ptr, batch_size_, nullptr, batch_size_, stream);
  double sum = 0.0; float mn = 1e20f, mx = -1e20f;
  reduce_errors(test_y1_.ptr, train_y_.ptr, batch_size_, false, &sum, &mn, &mx, stream);  // l1_loss, network.cu:283
  return (float)(sum / (double)batch_size_);
}

float NeuralVolume::get_psnr(bool /*quiet*/)
{
  if (!source_) throw std::runtime_error("missing a reference volume");
  const vec3i dims = desc.dims;
  const vec3f rdims = {1.0f / (float)dims.x, 1.0f / (float)dims.y, 1.0f / (float)dims.z};
  const vec3i batch = {std::min(4096, dims.x), std::min(16, dims.y), std::min(16, dims.z)};  // network.cu:418
  const size_t N = (size_t)batch.x * batch.y * batch.z;
  DeviceBuffer<float> coords, pred, ref;
  coords.resize(3 * N); pred.resize(N); ref.resize(N);
  double err = 0.0; float vmin = 1e20f, vmax = -1e20f;
  for (int z = 0; z < dims.z; z += batch.z)
    for (int y = 0; y < dims.y; y += batch.y)
      for (int x = 0; x < dims.x; x += batch.x) {
        const vec3i off = {x, y, z};
        const vec3i blk = {std::min(batch.x, dims.x - x), std::min(batch.y, dims.y - y), std::min(batch.z, dims.z - z)};
        const size_t count = (size_t)blk.x * blk.y * blk.z;
        if (count == 0) continue;
        source_->take_samples_grid(coords.ptr, ref.ptr, off, blk, rdims, stream);
        net_.inference(coords.ptr, pred.ptr, count, nullptr, count, stream);
        reduce_errors(pred.ptr, ref.ptr, count, true, &err, &vmin, &vmax, stream);
      }
  const double range = (double)vmax - (double)vmin;
  const double mse = err / ((double)dims.x * dims.y * dims.z);
  return (float)(10.0 * std::log10(range * range / mse));  // network.cu:469-471
}

// compute_ssim<7> (network.cu:70-127): SSIM of one 7^3 uniform window per output voxel, reference fx vs inference fy on a
// block grid with a 3-voxel halo; sample covariance (cov_norm = NP / (NP - 1)), C1 = (K1 R)^2, C2 = (K2 R)^2.  The block's
// sum is reduced here (the reference writes the map and thrust-reduces it in fp32; the sum is kept in fp64 here).
__global__ void ssim_block_kernel(vec3i block, vec3i gdims, const float* __restrict__ fx_, const float* __restrict__ fy_, float data_range,
                                  float cov_norm, float K1, float K2, double* __restrict__ sum)
{
  constexpr int W = 7;
  __shared__ double red[256];
  const uint32_t n = (uint32_t)block.x * block.y * block.z;
  double acc = 0.0;
  for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
    const int x = (int)(i % (uint32_t)block.x), y = (int)((i / (uint32_t)block.x) % (uint32_t)block.y), z = (int)(i / ((uint32_t)block.x * block.y));
    float ux = 0.f, uy = 0.f, uxx = 0.f, uyy = 0.f, uxy = 0.f;
    for (int kz = 0; kz < W; ++kz)
      for (int ky = 0; ky < W; ++ky)
        for (int kx = 0; kx < W; ++kx) {
          const uint32_t g = (uint32_t)(x + kx) + (uint32_t)(y + ky) * gdims.x + (uint32_t)(z + kz) * gdims.x * gdims.y;
          const float fx = fx_[g], fy = fy_[g];
          ux += fx; uy += fy; uxx += fx * fx; uyy += fy * fy; uxy += fx * fy;
        }
    const float w = 1.f / (W * W * W);  // uniform filter
    ux *= w; uy *= w; uxx *= w; uyy *= w; uxy *= w;
    const float vx = cov_norm * (uxx - ux * ux), vy = cov_norm * (uyy - uy * uy), vxy = cov_norm * (uxy - ux * uy);
    const float R = data_range, C1 = (K1 * R) * (K1 * R), C2 = (K2 * R) * (K2 * R);
    const float A1 = 2 * ux * uy + C1, A2 = 2 * vxy + C2, B1 = ux * ux + uy * uy + C1, B2 = vx + vy + C2;
    acc += (double)((A1 * A2) / (B1 * B2));
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  for (uint32_t s = 128; s > 0; s >>= 1) {
    if (threadIdx.x < s) red[threadIdx.x] += red[threadIdx.x + s];
    __syncthreads();
  }
  if (threadIdx.x == 0) atomicAdd(sum, red[0]);
}

// get_mssim (network.cu:474-549): mean SSIM over the interior (dims - 6 per axis), blocks of <= 4096 x 16 x 16 voxels plus halo
float NeuralVolume::get_ssim(bool /*quiet*/)
{
  if (!source_) throw std::runtime_error("missing a reference volume");  // the reference prints this and returns -1
  if (!net_.valid()) throw std::runtime_error("neural volume has no valid network");
  constexpr int win = 7, crop = win >> 1, NP = win * win * win;
  const float K1 = 0.01f, K2 = 0.03f, data_range = 1.0f, cov_norm = (float)NP / (float)(NP - 1);
  const vec3i dims = desc.dims;
  if (dims.x < win || dims.y < win || dims.z < win) throw std::runtime_error("volume smaller than the 7^3 SSIM window");
  const vec3f rdims = {1.0f / (float)dims.x, 1.0f / (float)dims.y, 1.0f / (float)dims.z};
  const vec3i batch = {std::min(4096, dims.x), std::min(16, dims.y), std::min(16, dims.z)};
  const size_t grid_cap = (size_t)(batch.x + win - 1) * (batch.y + win - 1) * (batch.z + win - 1);
  DeviceBuffer<float> coords, pred, ref;
  DeviceBuffer<double> sum;
  coords.resize(3 * grid_cap); pred.resize(grid_cap); ref.resize(grid_cap);
  sum.resize(1); sum.zero(stream);
  for (int z = crop; z < dims.z - crop; z += batch.z)
    for (int y = crop; y < dims.y - crop; y += batch.y)
      for (int x = crop; x < dims.x - crop; x += batch.x) {
        const vec3i off = {x, y, z};
        const vec3i blk = {std::min(batch.x, dims.x - crop - x), std::min(batch.y, dims.y - crop - y), std::min(batch.z, dims.z - crop - z)};
        if (blk.x <= 0 || blk.y <= 0 || blk.z <= 0) continue;
        const vec3i goff = {off.x - crop, off.y - crop, off.z - crop};
        const vec3i gblk = {blk.x + win - 1, blk.y + win - 1, blk.z + win - 1};
        const size_t gcount = (size_t)gblk.x * gblk.y * gblk.z;
        source_->take_samples_grid(coords.ptr, ref.ptr, goff, gblk, rdims, stream);
        net_.inference(coords.ptr, pred.ptr, gcount, nullptr, gcount, stream);
        const uint32_t n = (uint32_t)blk.x * blk.y * blk.z;
        ssim_block_kernel<<<std::min<uint32_t>(div_round_up(n, 256), 2048u), 256, 0, stream>>>(blk, gblk, ref.ptr, pred.ptr, data_range, cov_norm, K1, K2,
                                                                                              sum.ptr);
        VNR_HIP_CHECK(hipGetLastError());
      }
  double total = 0.0;
  VNR_HIP_CHECK(hipMemcpyAsync(&total, sum.ptr, sizeof(double), hipMemcpyDeviceToHost, stream));
  VNR_HIP_CHECK(hipStreamSynchronize(stream));
  return (float)(total / ((double)(dims.x - win + 1) * (dims.y - win + 1) * (dims.z - win + 1)));
}

void NeuralVolume::inference(size_t n, const float* d_in, float* d_out, hipStream_t s)
{
  if (!net_.valid()) return;
  net_.inference(d_in, d_out, n, nullptr, n, s);
}

// infer_progressively_decode_volume (network.cu:290-326): m_lower = 0, m_upper = dims, so a blob is 16 whole z-slices and its
// result is contiguous in the decoded array: the inference writes straight into it (the reference copies with cudaMemcpy3D).
void NeuralVolume::decode_progressive()
{
  if (!net_.valid()) throw std::runtime_error("neural volume has no valid network");
  const vec3i dims = desc.dims;
  const size_t slice = (size_t)dims.x * dims.y, total = slice * dims.z;
  const int per_blob = 16;  // m_num_slices_per_blob, network.cu:171
  if (decoded_.count != total) { decoded_.resize(total); decoded_.zero(stream); decode_blob_ = 0; }
  if (decode_coords_.count != 3 * slice * per_blob) decode_coords_.resize(3 * slice * per_blob);
  const int b = decode_blob_;
  const int nz = std::min(per_blob, dims.z - b * per_blob);
  const size_t count = slice * (size_t)nz;
  if (count >= (1ull << 32)) throw std::runtime_error("blob too large");
  const vec3f rdims = {1.0f / (float)dims.x, 1.0f / (float)dims.y, 1.0f / (float)dims.z};
  generate_coords_kernel<<<div_round_up(count, 256), 256, 0, stream>>>((uint32_t)count, vec3i{0, 0, b * per_blob}, vec3i{dims.x, dims.y, nz}, rdims,
                                                                        decode_coords_.ptr);
  VNR_HIP_CHECK(hipGetLastError());
  net_.inference(decode_coords_.ptr, decoded_.ptr + slice * (size_t)b * per_blob, count, nullptr, count, stream);
  VNR_HIP_CHECK(hipStreamSynchronize(stream));  // the renderer reads the decoded volume on its own stream
  decode_blob_ = (b + 1) * per_blob >= dims.z ? 0 : b + 1;
}

// save_inference_volume (network.cu:328-365): z-slice by z-slice; every slice is written with its length padded to a
// multiple of 256 values, and the padding is what the network returns at the coordinates generate_coords produces for
// those indices (they wrap into slice z + 1) -- reproduced by running the same kernel over the padded count.
void NeuralVolume::save_inference_volume(const std::string& filename)
{
  if (!net_.valid()) throw std::runtime_error("neural volume has no valid network");
  const vec3i dims = desc.dims;
  const vec3f rdims = {1.0f / (float)dims.x, 1.0f / (float)dims.y, 1.0f / (float)dims.z};
  const size_t count = (((size_t)dims.x * dims.y + 255) / 256) * 256;  // util::next_multiple<size_t>(..., 256)
  DeviceBuffer<float> coords, values;
  coords.resize(3 * count); values.resize(count);
  std::vector<float> host(count);
  std::ofstream ofs(filename, std::ios::binary | std::ios::out);
  if (!ofs) throw std::runtime_error("cannot open " + filename);
  for (int z = 0; z < dims.z; ++z) {
    generate_coords_kernel<<<div_round_up(count, 256), 256, 0, stream>>>((uint32_t)count, vec3i{0, 0, z}, vec3i{dims.x, dims.y, 1}, rdims, coords.ptr);
    net_.inference(coords.ptr, values.ptr, count, nullptr, count, stream);
    VNR_HIP_CHECK(hipMemcpyAsync(host.data(), values.ptr, count * sizeof(float), hipMemcpyDeviceToHost, stream));
    VNR_HIP_CHECK(hipStreamSynchronize(stream));
    ofs.write((const char*)host.data(), (std::streamsize)(count * sizeof(float)));
  }
  if (!ofs) throw std::runtime_error("error while writing " + filename);
}

// save_reference_volume (network.cu:367-405): the (normalised) reference volume at the voxel centres, same slice format.
// Two deliberate differences: the reference ignores `filename` and always writes "reference.bin" into the working directory
// (this writes `filename`), and it writes whatever its scratch buffer holds as padding (this writes zeros).
void NeuralVolume::save_reference_volume(const std::string& filename)
{
  if (!source_) throw std::runtime_error("missing a reference volume");  // the reference prints this and returns
  const vec3i dims = desc.dims;
  const vec3f rdims = {1.0f / (float)dims.x, 1.0f / (float)dims.y, 1.0f / (float)dims.z};
  const size_t xy = (size_t)dims.x * dims.y, count = ((xy + 255) / 256) * 256;
  DeviceBuffer<float> coords, values;
  coords.resize(3 * count); values.resize(count);
  values.zero(stream);
  std::vector<float> host(count);
  std::ofstream ofs(filename, std::ios::binary | std::ios::out);
  if (!ofs) throw std::runtime_error("cannot open " + filename);
  for (int z = 0; z < dims.z; ++z) {
    source_->take_samples_grid(coords.ptr, values.ptr, vec3i{0, 0, z}, vec3i{dims.x, dims.y, 1}, rdims, stream);
    VNR_HIP_CHECK(hipMemcpyAsync(host.data(), values.ptr, count * sizeof(float), hipMemcpyDeviceToHost, stream));
    VNR_HIP_CHECK(hipStreamSynchronize(stream));
    ofs.write((const char*)host.data(), (std::streamsize)(count * sizeof(float)));
  }
  if (!ofs) throw std::runtime_error("error while writing " + filename);
}

void NeuralVolume::save_params_to_json(Json& root)
{
  const vec3i md = mc_.dims();
  const vec3f ms = mc_.spacings();
  std::vector<float> h(mc_.n_cells() * 2);
  VNR_HIP_CHECK(hipMemcpyAsync(h.data(), mc_.d_value_range(), h.size() * sizeof(float), hipMemcpyDeviceToHost, stream));
  VNR_HIP_CHECK(hipStreamSynchronize(stream));
  Json vol = Json::object(), vdims = Json::object();
  vdims["x"] = desc.dims.x; vdims["y"] = desc.dims.y; vdims["z"] = desc.dims.z;
  vol["dims"] = vdims;
  root["volume"] = vol;
  Json mc = Json::object(), mdims = Json::object(), msp = Json::object();
  mdims["x"] = md.x; mdims["y"] = md.y; mdims["z"] = md.z;
  msp["x"] = ms.x; msp["y"] = ms.y; msp["z"] = ms.z;
  mc["groundtruth"] = mc_.is_external();
  mc["dims"] = mdims;
  mc["spacings"] = msp;
  mc["data"] = Json::binary(h.data(), h.size() * sizeof(float));
  root["macrocell"] = mc;
  root["parameters"] = net_.serialize_params(stream);
  root["model"] = net_.model_json();
}

void NeuralVolume::load_params_from_json(const Json& root)
{
  if (root.contains("volume")) {
    const Json& d = root.at("volume").at("dims");
    const vec3i dims = {(int)d.at("x").as_int(), (int)d.at("y").as_int(), (int)d.at("z").as_int()};
    if (dims.x != desc.dims.x || dims.y != desc.dims.y || dims.z != desc.dims.z) throw std::runtime_error("mismatch data dimension");
  }
  if (root.contains("macrocell")) {
    const Json& m = root.at("macrocell");
    const vec3i md = {(int)m.at("dims").at("x").as_int(), (int)m.at("dims").at("y").as_int(), (int)m.at("dims").at("z").as_int()};
    const vec3f ms = {m.at("spacings").at("x").as_float(), m.at("spacings").at("y").as_float(), m.at("spacings").at("z").as_float()};
    const vec3i cd = mc_.dims();
    const vec3f cs = mc_.spacings();
    if (md.x != cd.x || md.y != cd.y || md.z != cd.z || ms.x != cs.x || ms.y != cs.y || ms.z != cs.z || !mc_.allocated()) {
      mc_.set_external(nullptr);
      mc_.set_shape(desc.dims);  // keeps volume dims for explicit updates
      mc_.set_dims(md);
      mc_.set_spacings(ms);
      mc_.allocate(stream);
    }
    const std::string& bin = m.at("data").as_binary();
    mc_.upload_value_range(bin.data(), bin.size(), stream);
    mc_.update_max_opacity(tfn_.view(), stream);
  }
  if (root.contains("model")) { net_.configure(root.at("model"), init_seed ? init_seed : 1); replicas_synced_ = false; dp_.reset(); }
  if (root.contains("parameters")) net_.deserialize_params(root.at("parameters"), stream);
  else net_.deserialize_params(root, stream);  // legacy format, network.cu:934-936
}

}  // namespace vnr
