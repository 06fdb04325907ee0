// marching_cubes.hip — isosurface extraction (vnrMarchingCube / vnrSaveTriangles, core/marching_cube.cuh:6-8, core/marching_cube.cu:147-519;
// apps/batch_isosurface.cpp:55-80) for gfx950.
//
// What is kept from the reference: the dual grid (cells between the dims voxels: dims - 1 per axis, :23-24), the classification
// `value <= isovalue` (:187-191), the vertex rule `t = |fa - fb| >= 0.001 ? (iso - fa) / (fb - fa) : 0`, v = va (1 - t) + vb t, + cell + 0.5
// (:38-44, 242-246), where the corner values come from: the voxels themselves for a simple volume (:84-92), the network at
// (cell + corner) / dims for a neural volume (:117-122), three vertices per triangle, no indexing (:497-512 writes "v" lines and
// 1-based "f" triples).  What is not: the case table is this library's own derivation (tools/gen_mc_table.py; the reference ships the
// classic hand-made table, and no fixture pins its triangle order), so triangles may come out in another order and ambiguous faces may be
// cut the other way; and the pipeline is count -> scan of BLOCK counts -> emit (two passes over the values, 8 bytes of temporary
// storage per 256 cells) instead of flag -> compact -> count -> scan -> emit with 17 bytes per cell (:292-405).
// A neural volume is evaluated once per grid NODE into a dense array (dims values), not 8 times per cell.
#include <fstream>

#include <hipcub/hipcub.hpp>

#include "mc_table.h"
#include "volume.h"

namespace vnr {

__constant__ int8_t c_mc_table[256][kMcCaseElements];
__constant__ int8_t c_mc_edges[12][2];

struct McGrid {
  const float* values;   // [dims.z][dims.y][dims.x]
  vec3i dims, dual;
  float iso;
};

__device__ __forceinline__ bool mc_cell(const McGrid& g, uint64_t index, vec3i& c, float v[8], uint32_t& case_idx)
{
  const uint64_t n = (uint64_t)g.dual.x * g.dual.y * g.dual.z;
  if (index >= n) return false;
  c = {(int)(index % g.dual.x), (int)((index / g.dual.x) % g.dual.y), (int)(index / ((uint64_t)g.dual.x * g.dual.y))};
  case_idx = 0;
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    const int64_t idx = (c.x + (i & 1)) + (int64_t)(c.y + ((i >> 1) & 1)) * g.dims.x + (int64_t)(c.z + ((i >> 2) & 1)) * g.dims.x * g.dims.y;
    v[i] = g.values[idx];
    if (v[i] <= g.iso) case_idx |= 1u << i;
  }
  return true;
}

__device__ __forceinline__ uint32_t mc_count(uint32_t case_idx)
{
  uint32_t n = 0;
  while (n < (uint32_t)kMcCaseElements && c_mc_table[case_idx][n] >= 0) ++n;
  return n;
}

__global__ void __launch_bounds__(256) mc_count_kernel(const McGrid g, unsigned long long* __restrict__ block_counts)
{
  __shared__ uint32_t s[4];
  const uint64_t index = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  vec3i c; float v[8]; uint32_t case_idx;
  uint32_t n = mc_cell(g, index, c, v, case_idx) ? mc_count(case_idx) : 0u;
#pragma unroll
  for (int d = 32; d > 0; d >>= 1) n += __shfl_xor(n, d);
  if ((threadIdx.x & 63u) == 0) s[threadIdx.x >> 6] = n;
  __syncthreads();
  if (threadIdx.x == 0) block_counts[blockIdx.x] = (unsigned long long)(s[0] + s[1] + s[2] + s[3]);
}

__global__ void __launch_bounds__(256) mc_emit_kernel(const McGrid g, const unsigned long long* __restrict__ block_offsets, vec3f* __restrict__ vertices)
{
  __shared__ uint32_t s[4];
  const uint64_t index = (uint64_t)blockIdx.x * 256u + threadIdx.x;
  const uint32_t lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
  vec3i c; float v[8]; uint32_t case_idx = 0;
  const bool valid = mc_cell(g, index, c, v, case_idx);
  const uint32_t n = valid ? mc_count(case_idx) : 0u;
  uint32_t incl = n;
#pragma unroll
  for (int d = 1; d < 64; d <<= 1) {
    const uint32_t y = __shfl_up(incl, d);
    if ((int)lane >= d) incl += y;
  }
  if (lane == 63) s[wave] = incl;
  __syncthreads();
  uint64_t base = block_offsets[blockIdx.x] + (incl - n);
  for (uint32_t w = 0; w < wave; ++w) base += s[w];
  for (uint32_t k = 0; k < n; ++k) {
    const int e = c_mc_table[case_idx][k];
    const int a = c_mc_edges[e][0], b = c_mc_edges[e][1];
    const float fa = v[a], fb = v[b];
    float t = 0.0f;
    if (fabsf(fa - fb) >= 0.001f) t = (g.iso - fa) / (fb - fa);   // lerp_verts (core/marching_cube.cu:38-44)
    const vec3f va = {(float)(a & 1), (float)((a >> 1) & 1), (float)((a >> 2) & 1)}, vb = {(float)(b & 1), (float)((b >> 1) & 1), (float)((b >> 2) & 1)};
    vec3f p = (1.0f - t) * va + t * vb;
    p = p + vec3f{(float)c.x, (float)c.y, (float)c.z} + vec3f{0.5f, 0.5f, 0.5f};   // :244-245
    vertices[base + k] = p;
  }
}

// the network at the grid nodes p = index / dims (VolumeDesc<Impl>::compute_voxel_values, :117-122)
__global__ void mc_node_coords_kernel(uint32_t n, vec3i dims, int z0, float* __restrict__ coords)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const uint32_t plane = (uint32_t)dims.x * (uint32_t)dims.y;
  const int x = (int)(i % (uint32_t)dims.x), y = (int)((i % plane) / (uint32_t)dims.x), z = z0 + (int)(i / plane);
  coords[3 * (size_t)i + 0] = (float)x / (float)dims.x;
  coords[3 * (size_t)i + 1] = (float)y / (float)dims.y;
  coords[3 * (size_t)i + 2] = (float)z / (float)dims.z;
}

// -> number of vertices (3 per triangle); `vertices` is (re)allocated to hold them
size_t marching_cubes(VolumeBase& volume, float isovalue, DeviceBuffer<vec3f>& vertices)
{
  if (!Runtime::get().ready()) Runtime::get().init(-1);
  hipStream_t s = Runtime::get().stream;
  static bool tables_up = false;
  if (!tables_up) {
    VNR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_mc_table), kMcCaseTableHost, sizeof(kMcCaseTableHost)));
    VNR_HIP_CHECK(hipMemcpyToSymbol(HIP_SYMBOL(c_mc_edges), kMcEdgeCornersHost, sizeof(kMcEdgeCornersHost)));
    tables_up = true;
  }
  const vec3i dims = volume.desc.dims;
  if (dims.x < 2 || dims.y < 2 || dims.z < 2) { vertices.resize(0); return 0; }
  McGrid g;
  g.dims = dims; g.dual = {dims.x - 1, dims.y - 1, dims.z - 1}; g.iso = isovalue;
  DeviceBuffer<float> nodes, coords;
  if (volume.is_network()) {
    NeuralVolume& nv = static_cast<NeuralVolume&>(volume);
    if (!nv.network().valid()) throw std::runtime_error("neural volume has no valid network");
    nodes.resize((size_t)dims.x * dims.y * dims.z);
    const size_t plane = (size_t)dims.x * dims.y;
    const int slab = (int)std::max<size_t>(1, std::min<size_t>((size_t)dims.z, ((size_t)1 << 24) / plane));   // <= 16 M points per inference launch
    coords.resize(3 * plane * (size_t)slab);
    for (int z0 = 0; z0 < dims.z; z0 += slab) {
      const int nz = std::min(slab, dims.z - z0);
      const size_t count = plane * (size_t)nz;
      mc_node_coords_kernel<<<div_round_up(count, 256), 256, 0, s>>>((uint32_t)count, dims, z0, coords.ptr);
      nv.network().inference(coords.ptr, nodes.ptr + plane * (size_t)z0, count, nullptr, count, s);
    }
    g.values = nodes.ptr;
  } else {
    SimpleVolume& sv = static_cast<SimpleVolume&>(volume);
    if (!sv.has_data()) throw std::runtime_error("this volume has no resident data (training mode OUT_OF_CORE / NOTHING)");
    g.values = sv.d_data();
  }
  const uint64_t n_cells = (uint64_t)g.dual.x * g.dual.y * g.dual.z;
  const uint64_t n_blocks64 = (n_cells + 255u) / 256u;
  if (n_blocks64 >= (1ull << 31)) throw std::runtime_error("volume too large for marching cubes");
  const uint32_t n_blocks = (uint32_t)n_blocks64;
  DeviceBuffer<unsigned long long> counts, offsets;
  counts.resize((size_t)n_blocks + 1);
  offsets.resize((size_t)n_blocks + 1);
  VNR_HIP_CHECK(hipMemsetAsync(counts.ptr + n_blocks, 0, sizeof(unsigned long long), s));   // the extra element: its offset is the total
  mc_count_kernel<<<n_blocks, 256, 0, s>>>(g, counts.ptr);
  VNR_HIP_CHECK(hipGetLastError());
  size_t temp_bytes = 0;
  VNR_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(nullptr, temp_bytes, counts.ptr, offsets.ptr, (int)(n_blocks + 1), s));
  DeviceBuffer<uint8_t> temp;
  temp.resize(std::max<size_t>(temp_bytes, 16));
  VNR_HIP_CHECK(hipcub::DeviceScan::ExclusiveSum(temp.ptr, temp_bytes, counts.ptr, offsets.ptr, (int)(n_blocks + 1), s));
  unsigned long long total = 0;
  VNR_HIP_CHECK(hipMemcpyAsync(&total, offsets.ptr + n_blocks, sizeof(total), hipMemcpyDeviceToHost, s));
  VNR_HIP_CHECK(hipStreamSynchronize(s));
  vertices.resize((size_t)total);
  if (total) {
    mc_emit_kernel<<<n_blocks, 256, 0, s>>>(g, offsets.ptr, vertices.ptr);
    VNR_HIP_CHECK(hipGetLastError());
  }
  VNR_HIP_CHECK(hipStreamSynchronize(s));   // the temporaries go out of scope
  return (size_t)total;
}

// vnrSaveTriangles (core/marching_cube.cu:497-519): Wavefront OBJ, one "v" line per vertex (std::to_string: "%f") and one 1-based "f" triple
// per three vertices
void save_triangles_obj(const std::string& filename, const float* xyz, size_t n_vertices)
{
  std::string str;
  str.reserve(n_vertices * 40);
  for (size_t i = 0; i < n_vertices; ++i)
    str += "v " + std::to_string(xyz[3 * i]) + " " + std::to_string(xyz[3 * i + 1]) + " " + std::to_string(xyz[3 * i + 2]) + "\n";
  for (size_t i = 0; i < n_vertices / 3; ++i) str += "f " + std::to_string(3 * i + 1) + " " + std::to_string(3 * i + 2) + " " + std::to_string(3 * i + 3) + "\n";
  std::ofstream out(filename);
  if (!out) throw std::runtime_error("cannot write " + filename);
  out.write(str.c_str(), (std::streamsize)str.length());
}

}  // namespace vnr
