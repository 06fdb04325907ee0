// common.h — shared host-side helpers of libvnr_amd (HIP runtime, errors, POD math, device buffers).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <stdexcept>
#include <string>
#include <vector>

namespace vnr {

#define VNR_HIP_CHECK(expr)                                                                          \
  do {                                                                                               \
    hipError_t _e = (expr);                                                                          \
    if (_e != hipSuccess) {                                                                          \
      throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(_e) + " at " __FILE__ \
                               ":" + std::to_string(__LINE__) + " (" #expr ")");                   \
    }                                                                                                \
  } while (0)

struct vec2i { int x, y; };
struct vec3i { int x, y, z; };
struct vec2f { float x, y; };
struct vec3f { float x, y, z; };
struct vec4f { float x, y, z, w; };
struct box3f { vec3f lower, upper; };

// object<->world affine map: columns vx, vy, vz and translation p (gdt::affine3f layout)
struct affine3f { vec3f vx, vy, vz, p; };

__host__ __device__ inline vec3f operator+(vec3f a, vec3f b) { return {a.x + b.x, a.y + b.y, a.z + b.z}; }
__host__ __device__ inline vec3f operator-(vec3f a, vec3f b) { return {a.x - b.x, a.y - b.y, a.z - b.z}; }
__host__ __device__ inline vec3f operator*(vec3f a, vec3f b) { return {a.x * b.x, a.y * b.y, a.z * b.z}; }
__host__ __device__ inline vec3f operator*(float s, vec3f a) { return {s * a.x, s * a.y, s * a.z}; }
__host__ __device__ inline float dot(vec3f a, vec3f b) { return a.x * b.x + a.y * b.y + a.z * b.z; }
__host__ __device__ inline vec3f cross(vec3f a, vec3f b)
{
  return {a.y * b.z - a.z * b.y, a.z * b.x - a.x * b.z, a.x * b.y - a.y * b.x};
}
__host__ __device__ inline vec3f normalize(vec3f a) { return (1.0f / sqrtf(dot(a, a))) * a; }
__host__ __device__ inline vec3f xfm_vector(const affine3f& a, vec3f v)
{
  return (v.x * a.vx + v.y * a.vy) + v.z * a.vz;
}
__host__ __device__ inline vec3f xfm_point(const affine3f& a, vec3f v) { return xfm_vector(a, v) + a.p; }

inline affine3f affine_inverse(const affine3f& a)
{
  const vec3f c0 = cross(a.vy, a.vz), c1 = cross(a.vz, a.vx), c2 = cross(a.vx, a.vy);
  const float r = 1.0f / dot(a.vx, c0);
  affine3f o;
  o.vx = r * vec3f{c0.x, c1.x, c2.x};
  o.vy = r * vec3f{c0.y, c1.y, c2.y};
  o.vz = r * vec3f{c0.z, c1.z, c2.z};
  const vec3f t = xfm_vector(o, a.p);
  o.p = {-t.x, -t.y, -t.z};
  return o;
}

inline affine3f affine_scale_then(const vec3f s, const affine3f& a)
{  // scale(s) * a
  affine3f o = a;
  o.vx = s * a.vx; o.vy = s * a.vy; o.vz = s * a.vz; o.p = s * a.p;
  return o;
}

// runtime ------------------------------------------------------------------------------------------
struct Runtime {
  int device = -1;
  hipStream_t stream = nullptr;  // the library's stream
  // The ray parts' streams of EVERY renderer of the process (part 0 runs on `stream`): created once, on first use.  The HIP runtime deals a
  // process's streams round-robin onto four hardware queues, in creation order; a renderer that created streams of its own landed wherever
  // the count stood, and every fourth one on the queue of `stream` itself, where two ray parts then run one behind the other (round 5,
  // profiles/r05_stream_budget.txt).  With the pool the library owns at most 1 + 3 streams for rendering whatever comes and goes; parts 1 and 2
  // are created together with `stream` (Runtime::init), BEFORE any other stream of the library (a rank's communication stream, an out-of-core
  // sampler's copy stream, the opt-in training side stream) and before whatever the application creates after vnrAmdInit: the three streams a
  // frame runs on own three hardware queues, and a later stream shares a queue only with streams that are idle while a frame renders (measured
  // with 0-3 idle extra streams: the 1/8 share 0.54-0.59 ms in every case).  The fourth part's stream (VNR_AMD_SMALL_SHARE_PARTS=4 /
  // VNR_AMD_RENDER_HALVES=4) is created on first use.
  hipStream_t part_streams[4] = {nullptr, nullptr, nullptr, nullptr};
  hipStream_t part_stream(int part);   // part >= 1
  int n_cus = 256;
  size_t bytes_renderer = 0, bytes_network = 0;
  static Runtime& get();
  void init(int device);
  bool ready() const { return device >= 0; }
};

inline hipStream_t resolve_stream(void* s) { return s ? (hipStream_t)s : Runtime::get().stream; }

enum class MemTag { Renderer, Network };

// RAII device buffer
template <typename T>
struct DeviceBuffer {
  T* ptr = nullptr;
  size_t count = 0;
  MemTag tag = MemTag::Renderer;
  DeviceBuffer() = default;
  explicit DeviceBuffer(MemTag t) : tag(t) {}
  DeviceBuffer(const DeviceBuffer&) = delete;
  DeviceBuffer& operator=(const DeviceBuffer&) = delete;
  DeviceBuffer(DeviceBuffer&& o) noexcept { *this = std::move(o); }
  DeviceBuffer& operator=(DeviceBuffer&& o) noexcept
  {
    if (this != &o) { release(); ptr = o.ptr; count = o.count; tag = o.tag; o.ptr = nullptr; o.count = 0; }
    return *this;
  }
  ~DeviceBuffer() { release(); }
  size_t& counter() { return tag == MemTag::Network ? Runtime::get().bytes_network : Runtime::get().bytes_renderer; }
  void release()
  {
    if (ptr) { (void)hipFree(ptr); counter() -= count * sizeof(T); }
    ptr = nullptr; count = 0;
  }
  void resize(size_t n)
  {
    if (n == count) return;
    release();
    if (n) {
      if (!Runtime::get().ready()) Runtime::get().init(-1);
      VNR_HIP_CHECK(hipMalloc((void**)&ptr, n * sizeof(T)));
      counter() += n * sizeof(T);
    }
    count = n;
  }
  void ensure(size_t n) { if (n > count) resize(n); }
  void zero(hipStream_t s) { if (count) VNR_HIP_CHECK(hipMemsetAsync(ptr, 0, count * sizeof(T), s)); }
  void upload(const T* h, size_t n, hipStream_t s)
  {
    ensure(n);
    if (n) VNR_HIP_CHECK(hipMemcpyAsync(ptr, h, n * sizeof(T), hipMemcpyHostToDevice, s));
  }
  void download(T* h, size_t n, hipStream_t s) const
  {
    if (n) VNR_HIP_CHECK(hipMemcpyAsync(h, ptr, n * sizeof(T), hipMemcpyDeviceToHost, s));
    VNR_HIP_CHECK(hipStreamSynchronize(s));
  }
  size_t bytes() const { return count * sizeof(T); }
};

inline uint32_t div_round_up(uint64_t a, uint64_t b) { return (uint32_t)((a + b - 1) / b); }
inline uint32_t next_multiple(uint32_t v, uint32_t d) { return ((v + d - 1) / d) * d; }

// software fp16 <-> fp32 for host-side parameter handling (round-to-nearest-even, subnormals kept)
uint16_t f32_to_f16(float f);
float f16_to_f32(uint16_t h);

}  // namespace vnr
