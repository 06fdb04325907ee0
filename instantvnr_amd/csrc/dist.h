// dist.h — one process per GPU: rendezvous, a small host control plane and the collectives of the two sharded paths
// (image-tile rendering: framebuffer all-gather; data-parallel training: gradient all-reduce), behind the C-ABI.
//
// The reference is single-GPU: its only device-selection code is renderer.cpp:299-304 (VNR_CUDA_DEVICE) and there is no
// collective anywhere in the tree.  This file is new work for BASELINE.json's north star ("tiles across
// the 8 GPUs of one node with a final RCCL framebuffer gather, training batches sharded with an RCCL all-reduce").
//
// Two transports behind one seam:
//  * "rccl": librccl.so of the system ROCm, opened with dlopen at vnrAmdDistInit (573 MB that a one-GPU process never
//    maps); device pointers go straight to ncclAllGather / ncclAllReduce / ncclBroadcast on the caller's HIP stream, so
//    a collective is one more operation in stream order and the host never waits for it.
//  * "shm": host-staged through one POSIX shared-memory segment of the node.  Same call sites, same buffers, same
//    stream hand-offs; the bytes go device -> segment -> device.  It exists so that MORE THAN ONE RANK can run where
//    RCCL cannot (RCCL refuses two ranks on one device: the test suite runs 2 and 4 ranks on the one GPU of a test box),
//    and as the CPU-only transport of the world-size-2 tests.
// The control plane (rendezvous, host barrier, a few doubles for the bench's MAX / SUM over ranks) is a star of stream
// sockets through rank 0: an abstract unix socket named after MASTER_PORT when MASTER_ADDR is this host, TCP otherwise.
#pragma once

#include <cstddef>
#include <cstdint>
#include <memory>
#include <string>

#include <hip/hip_runtime.h>

namespace vnr {

enum class DistDType { F32 = 0, F16 = 1, U8 = 2 };
enum class DistOp { Sum = 0, Max = 1, Min = 2, Avg = 3 };   // Avg = Sum / world (ncclAvg; the shm transport divides the fp32 sum before rounding)

class ControlPlane;

class Transport {
public:
  virtual ~Transport() = default;
  virtual const char* name() const = 0;
  // all ranks contribute `bytes` each; rank r's contribution lands at d_recv + r * bytes on every rank.
  // d_send == d_recv + rank * bytes (in place) is allowed and is what the renderer does.
  virtual void all_gather(const void* d_send, void* d_recv, size_t bytes, hipStream_t s) = 0;
  // in place on d_buf; F16 sums are rounded to fp16 once per element on the shm transport, hop by hop on RCCL
  virtual void all_reduce(void* d_buf, size_t count, DistDType t, DistOp op, hipStream_t s) = 0;
  // rank r ends up with the reduced elements [r * count_per_rank, (r + 1) * count_per_rank) of d_buf, in place at that
  // offset (the rest of d_buf is unspecified afterwards); all_gather of the same slices completes a sharded update
  virtual void reduce_scatter(void* d_buf, size_t count_per_rank, DistDType t, DistOp op, hipStream_t s) = 0;
  virtual void broadcast(void* d_buf, size_t bytes, int root, hipStream_t s) = 0;
  // ranks of the RCCL communicator as RCCL itself counts them (ncclCommCount); 0 for a transport that is not RCCL
  virtual int rccl_ranks() const { return 0; }
};

class Dist {
public:
  static Dist& get();
  bool active() const { return transport_ != nullptr; }
  int rank() const { return rank_; }
  int world() const { return world_; }
  int local_rank() const { return local_rank_; }
  const char* transport_name() const { return transport_ ? transport_->name() : "none"; }
  int rccl_ranks_seen() const { return transport_ ? transport_->rccl_ranks() : 0; }   // did RCCL see N ranks? (the bench line's `rccl_ranks_seen`)
  Transport& transport();

  // explicit init (the application did its own rendezvous): 128-byte RCCL unique id made by vnrAmdDistGetUniqueId on one
  // rank.  transport: "rccl", "shm" or null / "" = env VNR_AMD_DIST_TRANSPORT, default rccl.
  void init(int rank, int world, int local_rank, const void* unique_id, const char* transport, const char* address);
  // torchrun contract: RANK / LOCAL_RANK / WORLD_SIZE / MASTER_ADDR / MASTER_PORT
  void init_from_env();
  void finalize();

  // host control plane (blocking; never inside a timed device pipeline)
  void barrier();
  void all_reduce_host(double* values, int n, DistOp op);
  void broadcast_host(void* data, size_t bytes, int root);
  void all_gather_host(const void* mine, void* all, size_t bytes);

  hipStream_t comm_stream();   // a second stream of this process for collectives that overlap compute
  // every collective of the sharded paths once, on patterned buffers, with a deadline per collective (dist.cpp): throws with the
  // collective's name on a mismatch or a timeout; returns a one-line report
  std::string self_test(double deadline_s);

private:
  int rank_ = 0, world_ = 1, local_rank_ = 0;
  std::unique_ptr<ControlPlane> ctl_;
  std::unique_ptr<Transport> transport_;
  hipStream_t comm_stream_ = nullptr;
public:
  Dist();
  ~Dist();
};

// pure helpers of the interleaved tile sharding, shared by the renderer and the tests
struct ShareLayout {
  uint32_t block;       // pixels per block (8 scanlines)
  uint32_t n_blocks;    // blocks of the whole image
  uint32_t per_part;    // blocks per rank (rounded up)
  uint32_t n_local;     // per_part * block: pixels of one rank's share, zero padded past the image
};
ShareLayout share_layout(uint32_t width, uint32_t height, uint32_t parts);

// the 128-byte id of an RCCL communicator (ncclGetUniqueId); throws when librccl cannot be opened
void rccl_unique_id(void* out128);

}  // namespace vnr
