// ooc_sampler.hip — see ooc_sampler.h.  Reference: core/samplers/neural_sampler.cpp:302-329 (trilinear_vkl), :377-668
// (StreamLoader, RandomBuffer), :967-1035 (sample_streaming_grid), :1043-1127 (OutOfCoreSampler).
#include "ooc_sampler.h"

#include "dist.h"

#include <fcntl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <atomic>
#include <cerrno>
#include <mutex>

namespace vnr {

constexpr uint64_t kStreamSize = 32 * 1024;  // RandomBuffer::STREAM_SIZE
constexpr uint64_t kAlignment = 512;         // RandomBuffer::ALIGNMENT

static uint32_t ooc_type_size(int type)
{
  switch (type) {
  case 0: case 1: return 1;
  case 2: case 3: return 2;
  case 4: case 5: case 8: return 4;
  case 12: return 8;
  default: throw std::runtime_error("out-of-core sampler: unsupported data type");  // read_typed_pointer (:143-156)
  }
}

static void pread_all(int fd, void* dst, size_t nbytes, uint64_t offset)
{
  uint8_t* p = (uint8_t*)dst;
  while (nbytes) {
    const ssize_t r = ::pread(fd, p, nbytes, (off_t)offset);
    if (r < 0) { if (errno == EINTR) continue; throw std::runtime_error(std::string("out-of-core sampler: pread: ") + strerror(errno)); }
    if (r == 0) throw std::runtime_error("out-of-core sampler: unexpected end of the volume file");
    p += r; nbytes -= (size_t)r; offset += (uint64_t)r;
  }
}

// ------------------------------------------------------------------------------------------------ device side
struct OocPcg32 {  // tcnn's pcg32 (EXTERNAL), the stream StaticSampler draws from as well (neural_sampler.cu:36-41)
  uint64_t state, inc;
  __device__ OocPcg32(uint64_t initstate, uint64_t initseq)
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

struct OocSampleArgs {
  const uint8_t* cache;
  const OocBlock* blocks;
  uint64_t n_blocks, block_bytes;
  uint64_t excl_first, excl_count;   // asynchronous refresh: slots [excl_first, excl_first + excl_count) (wrapping) are being replaced and are not sampled
  vec3i dims;
  int type;
  float lo, vscale;
  vec3f lower, upper;
  uint64_t seed, stream, offset;
  uint32_t n;
  float* coords;
  float* values;
};

__device__ __forceinline__ float ooc_read(const uint8_t* p, int type)
{
  switch (type) {
  case 0: return (float)*(const uint8_t*)p;
  case 1: return (float)*(const int8_t*)p;
  case 2: return (float)*(const uint16_t*)p;
  case 3: return (float)*(const int16_t*)p;
  case 4: return (float)*(const uint32_t*)p;
  case 5: return (float)*(const int32_t*)p;
  case 8: return *(const float*)p;
  default: return (float)*(const double*)p;
  }
}

// one training sample per thread (neural_sampler.cpp:1087-1113)
__global__ void ooc_sample_kernel(const OocSampleArgs a)
{
  const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= a.n) return;
  OocPcg32 rng(a.seed, a.stream);
  // three consecutive generate_random_uniform calls: 3 n coordinates, n block picks, n voxel picks
  rng.advance(a.offset + 3ull * i);
  const float ux = rng.next_float(), uy = rng.next_float(), uz = rng.next_float();
  OocPcg32 rb(a.seed, a.stream);
  rb.advance(a.offset + 3ull * a.n + i);
  const float r_b = rb.next_float();
  rb.advance((uint64_t)a.n - 1ull);
  const float r_v = rb.next_float();

  // the reference throws when float rounding carries a pick to the end of its range; here it is clamped to the last element
  // (asynchronous refresh: the pick runs over the slots that are NOT being replaced, counted from the end of the range that is)
  const uint64_t n_pick = a.n_blocks - a.excl_count;
  uint64_t bidx = min((uint64_t)(r_b * (float)n_pick), n_pick - 1ull);
  if (a.excl_count) { bidx += a.excl_first + a.excl_count; if (bidx >= a.n_blocks) bidx -= a.n_blocks; }
  const OocBlock b = a.blocks[bidx];
  const uint64_t vidx = min((uint64_t)(r_v * (float)b.length), (uint64_t)b.length - 1ull);
  const uint64_t index = b.offset + vidx;  // locate_voxel, then to_grid_index (:339-345)
  const uint64_t stride_y = (uint64_t)a.dims.x, stride_z = (uint64_t)a.dims.y * (uint64_t)a.dims.x;
  const int vx = (int)(index % stride_y), vy = (int)((index % stride_z) / stride_y), vz = (int)(index / stride_z);

  // a point inside the cell, normalised to [0,1) and mapped into [lower, upper)
  const float px = ux + (float)vx, py = uy + (float)vy, pz = uz + (float)vz;
  const float fx = (float)a.dims.x, fy = (float)a.dims.y, fz = (float)a.dims.z;
  a.coords[3 * (size_t)i + 0] = (px * (1.0f / fx)) * (a.upper.x - a.lower.x) + a.lower.x;
  a.coords[3 * (size_t)i + 1] = (py * (1.0f / fy)) * (a.upper.y - a.lower.y) + a.lower.y;
  a.coords[3 * (size_t)i + 2] = (pz * (1.0f / fz)) * (a.upper.z - a.lower.z) + a.lower.z;

  // trilinear_vkl (:302-329) at clamp(p, 0.5, dims - 0.5)
  const float cx = fminf(fmaxf(px, 0.5f), fx - 0.5f), cy = fminf(fmaxf(py, 0.5f), fy - 0.5f), cz = fminf(fmaxf(pz, 0.5f), fz - 0.5f);
  const float bx = cx - 0.5f, by = cy - 0.5f, bz = cz - 0.5f;
  const float ix = truncf(bx), iy = truncf(by), iz = truncf(bz);  // std::modf: integral part, towards zero (pb >= 0)
  const float wx = bx - ix, wy = by - iy, wz = bz - iz;
  const int x0 = min(max((int)ix, 0), a.dims.x - 1), y0 = min(max((int)iy, 0), a.dims.y - 1), z0 = min(max((int)iz, 0), a.dims.z - 1);
  const int x1 = min(x0 + 1, a.dims.x - 1), y1 = min(y0 + 1, a.dims.y - 1), z1 = min(z0 + 1, a.dims.z - 1);

  const uint32_t elem = a.type <= 1 ? 1u : a.type <= 3 ? 2u : a.type == 12 ? 8u : 4u;
  const uint8_t* slab = a.cache + bidx * a.block_bytes;
  auto voxel = [&](int x, int y, int z) -> float {  // access_voxel (:653-663) + the normalising accessor (:1098-1102)
    // y and z lie inside bounds_with_ghost by construction (the picked voxel +- 1); clamped anyway so that a corrupt table
    // cannot send a load outside the slab
    const int ly = min(max(y - b.ghost_lo_y, 0), b.ghost_ny - 1), lz = min(max(z - b.ghost_lo_z, 0), b.ghost_nz - 1);
    const uint64_t at = (uint64_t)x + (uint64_t)ly * (uint64_t)a.dims.x + (uint64_t)lz * (uint64_t)b.ghost_ny * (uint64_t)a.dims.x;
    const float v = ooc_read(slab + at * elem, a.type);
    return fminf(fmaxf((v - a.lo) * a.vscale, 0.0f), 1.0f);
  };
  const float c000 = voxel(x0, y0, z0), c001 = voxel(x1, y0, z0), c010 = voxel(x0, y1, z0), c011 = voxel(x1, y1, z0);
  const float c100 = voxel(x0, y0, z1), c101 = voxel(x1, y0, z1), c110 = voxel(x0, y1, z1), c111 = voxel(x1, y1, z1);
  const float ox = 1.0f - wx, oy = 1.0f - wy, oz = 1.0f - wz;
  float r = ox * oy * oz * c000;
  r = r + wx * oy * oz * c001;
  r = r + ox * wy * oz * c010;
  r = r + wx * wy * oz * c011;
  r = r + ox * oy * wz * c100;
  r = r + wx * oy * wz * c101;
  r = r + ox * wy * wz * c110;
  r = r + wx * wy * wz * c111;
  a.values[i] = r;
}

// ------------------------------------------------------------------------------------------------ host side
OutOfCoreSampler::OutOfCoreSampler(const std::string& filename, vec3i dims, int type, size_t file_offset, float range_lo, float range_hi,
                                   uint64_t n_concurrent_blocks, uint64_t n_blocks)
    : dims_(dims), type_(type), file_offset_(file_offset), lo_(range_lo), hi_(range_hi), n_concurrent_(n_concurrent_blocks), n_blocks_(n_blocks)
{
  if (dims.x <= 0 || dims.y <= 0 || dims.z <= 0) throw std::runtime_error("invalid volume dims");
  if (n_concurrent_ == 0 || n_blocks_ == 0) throw std::runtime_error("out-of-core sampler: block counts must be positive");
  if (n_concurrent_ > n_blocks_) n_concurrent_ = n_blocks_;
  elem_ = ooc_type_size(type);
  // data-parallel training (BASELINE C5: "each rank its own slab set"): rank r of a multi-process job draws its slabs from the
  // default-seeded generator's seed + r, so rank 0 and a single process keep the reference's sequence
  if (Dist::get().world() > 1) rng_.seed(std::mt19937::default_seed + (uint32_t)Dist::get().rank());
  if (!Runtime::get().ready()) Runtime::get().init(-1);
  fd_ = ::open(filename.c_str(), O_RDONLY);
  if (fd_ < 0) throw std::runtime_error("cannot open volume file: " + filename);
  struct stat st;
  if (fstat(fd_, &st) != 0) { ::close(fd_); throw std::runtime_error("cannot stat volume file: " + filename); }
  file_size_ = (size_t)st.st_size;
  const uint64_t need = (uint64_t)dims.x * dims.y * dims.z * elem_ + file_offset_;
  if (file_size_ < need) { ::close(fd_); throw std::runtime_error("volume file too short: " + filename); }

  // RandomBuffer ctor (:531-546)
  block_dims_.x = dims.x;
  block_dims_.y = (int)std::min<uint64_t>((kStreamSize + (uint64_t)dims.x * elem_ - 1) / ((uint64_t)dims.x * elem_), (uint64_t)dims.y);
  block_dims_.z = std::min(1, dims.z);
  ghost_dims_ = {block_dims_.x, std::min(block_dims_.y + 2, dims.y), std::min(block_dims_.z + 2, dims.z)};
  index_space_ = {1, (dims.y + block_dims_.y - 1) / block_dims_.y, (dims.z + block_dims_.z - 1) / block_dims_.z};
  const uint64_t ghost_bytes = (uint64_t)ghost_dims_.x * ghost_dims_.y * ghost_dims_.z * elem_;
  block_bytes_ = (ghost_bytes + kAlignment - 1) / kAlignment * kAlignment;

  io_threads_ = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
  if (const char* e = std::getenv("VNR_AMD_OOC_IO_THREADS")) io_threads_ = (unsigned)std::max(1, std::min(64, std::atoi(e)));
  blocks_.resize(n_blocks_);
  cache_.resize(n_blocks_ * block_bytes_);
  d_blocks_.resize(n_blocks_);
  VNR_HIP_CHECK(hipHostMalloc((void**)&staging_, n_concurrent_ * block_bytes_, hipHostMallocDefault));
  VNR_HIP_CHECK(hipHostMalloc((void**)&staging_blocks_, n_concurrent_ * sizeof(OocBlock), hipHostMallocDefault));
  VNR_HIP_CHECK(hipStreamCreateWithFlags(&copy_stream_, hipStreamNonBlocking));
  VNR_HIP_CHECK(hipEventCreateWithFlags(&ev_sampled_, hipEventDisableTiming));
  VNR_HIP_CHECK(hipEventCreateWithFlags(&ev_copied_, hipEventDisableTiming));

  if (const char* e = std::getenv("VNR_AMD_OOC_ASYNC")) async_refresh_ = std::atoi(e) != 0;
  // preloading (:555-559), then the first refresh in flight
  for (uint64_t i = 0; i < n_blocks_; i += n_concurrent_) {
    submit((int64_t)i);
    wait();
  }
  submit(-1);
}

OutOfCoreSampler::~OutOfCoreSampler()
{
  if (worker_.joinable()) worker_.join();
  if (copy_stream_) { (void)hipStreamSynchronize(copy_stream_); (void)hipStreamDestroy(copy_stream_); }
  if (ev_sampled_) (void)hipEventDestroy(ev_sampled_);
  for (hipEvent_t e : throttle_) (void)hipEventDestroy(e);
  if (ev_copied_) (void)hipEventDestroy(ev_copied_);
  if (staging_) (void)hipHostFree(staging_);
  if (staging_blocks_) (void)hipHostFree(staging_blocks_);
  if (fd_ >= 0) ::close(fd_);
}

uint64_t OutOfCoreSampler::random_uint64(uint64_t count)
{
  if (count == 0) throw std::runtime_error("calling 'uint64_random' with zero range.");
  if (count == 1) return 0;  // the reference's min < max assertion would throw for a one-slab volume
  std::uniform_int_distribution<uint64_t> distribution(0, count - 1);
  return distribution(rng_);
}

void OutOfCoreSampler::plan_block(uint64_t slot, vec3i bi)
{
  // submit_one_job (:579-636), geometry part
  const vec3i v0 = {bi.x * block_dims_.x, bi.y * block_dims_.y, bi.z * block_dims_.z};
  const vec3i v1 = {std::min(v0.x + block_dims_.x, dims_.x), std::min(v0.y + block_dims_.y, dims_.y), std::min(v0.z + block_dims_.z, dims_.z)};
  const int g0y = std::max(v0.y - 1, 0), g0z = std::max(v0.z - 1, 0);
  const int g1y = std::min(v1.y + 1, dims_.y), g1z = std::min(v1.z + 1, dims_.z);
  OocBlock b;
  b.offset = (uint64_t)v0.x + (uint64_t)v0.y * (uint64_t)dims_.x + (uint64_t)v0.z * (uint64_t)dims_.y * (uint64_t)dims_.x;
  b.length = (uint32_t)((uint64_t)(v1.x - v0.x) * (v1.y - v0.y) * (v1.z - v0.z));
  b.ghost_lo_y = g0y; b.ghost_lo_z = g0z; b.ghost_ny = g1y - g0y; b.ghost_nz = g1z - g0z;
  b.index_y = bi.y; b.index_z = bi.z;
  if (b.length == 0) throw std::runtime_error("[aio] zero block");
  if ((uint64_t)dims_.x * b.ghost_ny * b.ghost_nz * elem_ > block_bytes_) throw std::runtime_error("[aio] invalid block");
  blocks_[slot] = b;
}

void OutOfCoreSampler::submit(int64_t first)
{
  if (in_flight_) throw std::runtime_error("out-of-core sampler: a refresh is already in flight");
  const uint64_t i = first < 0 ? random_uint64(n_blocks_) : (uint64_t)first;   // submit_all_jobs (:638-646)
  const uint64_t space = (uint64_t)index_space_.x * index_space_.y * index_space_.z;
  for (uint64_t j = 0; j < n_concurrent_; ++j) {
    const uint64_t index = random_uint64(space);                                // random_grid_index (:351-356)
    const uint64_t sy = (uint64_t)index_space_.x, sz = (uint64_t)index_space_.y * (uint64_t)index_space_.x;
    plan_block((i + j) % n_blocks_, vec3i{(int)(index % sy), (int)((index % sz) / sy), (int)(index / sz)});
  }
  in_flight_ = true;
  inflight_first_ = i;
  ++refreshes_;
  worker_done_.store(false, std::memory_order_relaxed);
  worker_error_.clear();
  worker_ = std::thread([this, i]() { refresh_worker(i); worker_done_.store(true, std::memory_order_release); });
}

void OutOfCoreSampler::refresh_worker(uint64_t first)
{
  try {
    VNR_HIP_CHECK(hipSetDevice(Runtime::get().device));
    VNR_HIP_CHECK(hipEventSynchronize(ev_copied_));   // the staging buffers are free once the previous refresh has been copied
    // 1. preads: slab j of this refresh goes to staging slot j, z-slice by z-slice (x-full rows are contiguous in the file).
    // 2. the slabs travel to the device in chunks while the later ones are still being read: at the reference's 1024 slabs per step a
    //    refresh is 102 MiB, i.e. ~1.5 ms of preads from the page cache on 16 threads and ~2 ms of PCIe; one after the other they
    //    were the 3.8 ms of a step, overlapped 2.8-2.9.  The copies start after the last sampling kernel that may still
    //    read the slots being replaced.  A small refresh (VNR_NUM_CONCURRENT_BLOCKS=128: 13 MiB) goes as one copy: chunking it cost 0.1 ms.
    //    (The read threads are started per refresh.  A persistent pool saves ~0.4 ms of thread starts and was measured at 2.3 ms per
    //    step in half of the runs and at 2.8-3.6 in the other half, depending on where its threads had first been placed: not kept.)
    constexpr uint64_t kChunks = 8;
    const uint64_t n_chunks = std::max<uint64_t>(1, std::min<uint64_t>(kChunks, n_concurrent_ / 128));
    const uint64_t chunk = (n_concurrent_ + n_chunks - 1) / n_chunks;
    std::atomic<uint64_t> next{0}, bytes{0};
    std::atomic<uint64_t> done[kChunks];
    for (auto& d : done) d.store(0);
    std::atomic<bool> failed{false};
    std::string err;
    std::mutex err_mutex;
    auto io = [&]() {
      try {
        for (;;) {
          const uint64_t j = next.fetch_add(1);
          if (j >= n_concurrent_ || failed.load()) break;
          const OocBlock& b = blocks_[(first + j) % n_blocks_];
          staging_blocks_[j] = b;
          uint8_t* data = staging_ + j * block_bytes_;
          const uint64_t slice_bytes = (uint64_t)dims_.x * b.ghost_ny * elem_;
          for (int z = 0; z < b.ghost_nz; ++z) {
            const uint64_t file_index = (uint64_t)b.ghost_lo_y * (uint64_t)dims_.x + (uint64_t)(b.ghost_lo_z + z) * (uint64_t)dims_.y * (uint64_t)dims_.x;
            pread_all(fd_, data + (uint64_t)z * slice_bytes, slice_bytes, file_offset_ + file_index * elem_);
          }
          bytes.fetch_add(slice_bytes * b.ghost_nz);
          done[j / chunk].fetch_add(1, std::memory_order_release);
        }
      } catch (const std::exception& e) {
        std::lock_guard<std::mutex> g(err_mutex);
        err = e.what();
        failed.store(true);
      }
    };
    std::vector<std::thread> pool;
    struct Joiner {   // the readers use this frame's variables: whatever ends it, they have come back first
      std::vector<std::thread>* pool; std::atomic<bool>* failed;
      ~Joiner() { failed->store(true); for (auto& t : *pool) if (t.joinable()) t.join(); }
    } joiner{&pool, &failed};
    const unsigned nt = (unsigned)std::min<uint64_t>(io_threads_, n_concurrent_);
    for (unsigned t = n_chunks > 1 ? 0 : 1; t < nt; ++t) pool.emplace_back(io);
    if (n_chunks == 1) io();   // nothing to orchestrate: this thread reads too
    VNR_HIP_CHECK(hipStreamWaitEvent(copy_stream_, ev_sampled_, 0));
    // staging slot j belongs to cache slot (first + j) % n_blocks: a contiguous range of j is one copy, two where it wraps
    auto copy_range = [&](uint64_t j0, uint64_t j1) {
      while (j0 < j1) {
        const uint64_t slot = (first + j0) % n_blocks_;
        const uint64_t run = std::min(j1 - j0, n_blocks_ - slot);
        VNR_HIP_CHECK(hipMemcpyAsync(cache_.ptr + slot * block_bytes_, staging_ + j0 * block_bytes_, run * block_bytes_, hipMemcpyHostToDevice, copy_stream_));
        VNR_HIP_CHECK(hipMemcpyAsync(d_blocks_.ptr + slot, staging_blocks_ + j0, run * sizeof(OocBlock), hipMemcpyHostToDevice, copy_stream_));
        j0 += run;
      }
    };
    for (uint64_t c = 0; c * chunk < n_concurrent_ && !failed.load(); ++c) {
      const uint64_t j0 = c * chunk, j1 = std::min(n_concurrent_, j0 + chunk);
      while (done[c].load(std::memory_order_acquire) < j1 - j0 && !failed.load()) std::this_thread::sleep_for(std::chrono::microseconds(20));
      if (!failed.load()) copy_range(j0, j1);
    }
    for (auto& t : pool) t.join();
    if (!err.empty()) throw std::runtime_error(err);
    bytes_read_ += bytes.load();
    VNR_HIP_CHECK(hipEventRecord(ev_copied_, copy_stream_));
  } catch (const std::exception& e) {
    worker_error_ = e.what();
  }
}

void OutOfCoreSampler::wait()
{
  if (!in_flight_) return;
  worker_.join();
  in_flight_ = false;
  if (!worker_error_.empty()) throw std::runtime_error(worker_error_);
}

std::vector<OocBlock> OutOfCoreSampler::blocks()
{
  wait();
  return blocks_;
}

void OutOfCoreSampler::sample(float* d_coords, float* d_values, size_t n, vec3f lower, vec3f upper, uint64_t rng_seed, uint64_t rng_stream,
                              uint64_t& rng_offset, hipStream_t s)
{
  if (!(lo_ < hi_)) throw std::runtime_error("a valid value range must be provided");  // :1069-1071
  if (n == 0) return;
  if (n > 0xffffffffull) throw std::runtime_error("out-of-core sampler: batch too large");
  // The reference's schedule: wait for the refresh submitted by the previous call, sample, submit the next (a step is then as long as a
  // refresh whenever a refresh takes longer than a step: 102 MiB of preads and PCIe per step at the default 1024 slabs, 3 ms against a
  // 0.65 ms step).  Asynchronous refresh (VNR_AMD_OOC_ASYNC=1 / set_async_refresh): a step never waits.  While a refresh is in flight the
  // batch is drawn from the slots that are NOT being replaced (the kernel skips the range, so no sample reads a half-written slab), and the
  // next refresh is submitted by the first call that finds the previous one complete: slabs turn over at the rate the storage delivers
  // them instead of at a fixed 1024 per step, everything else (slab geometry, random slab choice, per-sample arithmetic) is unchanged.
  bool busy = false;
  if (async_refresh_ && in_flight_) {
    if (worker_done_.load(std::memory_order_acquire)) wait();
    else busy = true;
  } else {
    wait();                                                // randbuf.wait_all_jobs()
  }
  if (!busy) VNR_HIP_CHECK(hipStreamWaitEvent(s, ev_copied_, 0));     // ... and its copy to the device
  OocSampleArgs a;
  a.cache = cache_.ptr; a.blocks = d_blocks_.ptr;
  a.n_blocks = n_blocks_; a.block_bytes = block_bytes_;
  a.excl_first = busy ? inflight_first_ : 0; a.excl_count = busy ? n_concurrent_ : 0;
  if (busy && n_concurrent_ >= n_blocks_) throw std::runtime_error("out-of-core sampler: asynchronous refresh needs more resident slabs than one refresh replaces");
  a.dims = dims_; a.type = type_;
  a.lo = lo_; a.vscale = 1.0f / (hi_ - lo_);
  a.lower = lower; a.upper = upper;
  a.seed = rng_seed; a.stream = rng_stream; a.offset = rng_offset;
  a.n = (uint32_t)n; a.coords = d_coords; a.values = d_values;
  ooc_sample_kernel<<<div_round_up(n, 256), 256, 0, s>>>(a);
  VNR_HIP_CHECK(hipGetLastError());
  rng_offset += 5ull * n;
  if (async_refresh_) {
    // A refresh is tied to positions in the training stream (its copies run behind the last kernel that may read the slots, the first
    // kernel that reads them again waits for the copies), so a host that enqueues hundreds of steps ahead of the GPU would stretch every
    // refresh over that distance.  The host therefore stays at most kAhead sampling kernels ahead of the GPU here (it needs a fraction
    // of a step's time to enqueue a step, so this costs nothing).
    constexpr uint64_t kAhead = 6;
    if (throttle_.empty()) { throttle_.resize(kAhead); for (auto& e : throttle_) VNR_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming)); }
    hipEvent_t e = throttle_[calls_ % kAhead];
    if (calls_ >= kAhead) VNR_HIP_CHECK(hipEventSynchronize(e));
    VNR_HIP_CHECK(hipEventRecord(e, s));
    ++calls_;
  }
  if (busy) { ++steps_without_refresh_; return; }          // (the refresh in flight copies behind the LAST kernel that could read its slots: recorded when it was submitted)
  VNR_HIP_CHECK(hipEventRecord(ev_sampled_, s));
  submit(-1);                                              // randbuf.submit_all_jobs()
}

void OutOfCoreSampler::sample_grid(float* d_values, vec3i origin, vec3i size, vec3f spacing, hipStream_t s)
{
  // sample_streaming_grid (:967-1035) with its `trilinear = false`: the value of the voxel that holds the grid point.  The
  // reference reads the index range [origin, origin + size) of the FILE and therefore only works when the grid is the file's
  // own (dims <= 1024); here every grid point reads the file voxel it falls into, which is the same thing in that case.
  const size_t n = (size_t)size.x * size.y * size.z;
  if (n == 0) return;
  if (!(lo_ < hi_)) throw std::runtime_error("a valid value range must be provided");
  std::vector<float> values(n);
  const float fx = (float)dims_.x, fy = (float)dims_.y, fz = (float)dims_.z;
  const float scale = 1.0f / (hi_ - lo_);
  auto file_index = [](int g, float sp, float fd) { return (int)std::min(std::max((((float)g + 0.5f) * sp) * fd, 0.5f), fd - 0.5f); };
  const int x_lo = file_index(origin.x, spacing.x, fx), x_hi = file_index(origin.x + size.x - 1, spacing.x, fx);
  const size_t rows = (size_t)size.y * size.z;
  std::atomic<size_t> next{0};
  std::string err;
  std::mutex err_mutex;
  auto work = [&]() {
    try {
      std::vector<uint8_t> row((size_t)(x_hi - x_lo + 1) * elem_);
      for (;;) {
        const size_t r = next.fetch_add(1);
        if (r >= rows) break;
        const int gy = origin.y + (int)(r % (size_t)size.y), gz = origin.z + (int)(r / (size_t)size.y);
        const int iy = file_index(gy, spacing.y, fy), iz = file_index(gz, spacing.z, fz);
        const uint64_t at = (uint64_t)x_lo + (uint64_t)iy * (uint64_t)dims_.x + (uint64_t)iz * (uint64_t)dims_.y * (uint64_t)dims_.x;
        pread_all(fd_, row.data(), row.size(), file_offset_ + at * elem_);
        float* out = values.data() + r * (size_t)size.x;
        for (int gx = 0; gx < size.x; ++gx) {
          const int ix = file_index(origin.x + gx, spacing.x, fx);
          const uint8_t* p = row.data() + (size_t)(ix - x_lo) * elem_;
          float v;
          switch (type_) {
          case 0: v = (float)*(const uint8_t*)p; break;
          case 1: v = (float)*(const int8_t*)p; break;
          case 2: { uint16_t t; std::memcpy(&t, p, 2); v = (float)t; break; }
          case 3: { int16_t t; std::memcpy(&t, p, 2); v = (float)t; break; }
          case 4: { uint32_t t; std::memcpy(&t, p, 4); v = (float)t; break; }
          case 5: { int32_t t; std::memcpy(&t, p, 4); v = (float)t; break; }
          case 8: { std::memcpy(&v, p, 4); break; }
          default: { double t; std::memcpy(&t, p, 8); v = (float)t; break; }
          }
          out[gx] = std::min(std::max((v - lo_) * scale, 0.0f), 1.0f);
        }
      }
    } catch (const std::exception& e) {
      std::lock_guard<std::mutex> g(err_mutex);
      err = e.what();
    }
  };
  std::vector<std::thread> pool;
  const unsigned nt = (unsigned)std::min<size_t>(io_threads_, rows);
  for (unsigned t = 1; t < nt; ++t) pool.emplace_back(work);
  work();
  for (auto& t : pool) t.join();
  if (!err.empty()) throw std::runtime_error(err);
  VNR_HIP_CHECK(hipMemcpyAsync(d_values, values.data(), n * sizeof(float), hipMemcpyHostToDevice, s));
  VNR_HIP_CHECK(hipStreamSynchronize(s));  // `values` is pageable and goes out of scope
}

}  // namespace vnr
