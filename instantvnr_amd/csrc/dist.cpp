// dist.cpp — rendezvous, host control plane and the two collective transports (dist.h).
#include "dist.h"

#include <arpa/inet.h>
#include <dlfcn.h>
#include <fcntl.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <sys/mman.h>
#include <sys/socket.h>
#include <sys/stat.h>
#include <sys/time.h>
#include <sys/un.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <stdexcept>
#include <thread>
#include <vector>

#include "common.h"

namespace vnr {

// ================================================================================================ control plane
// A star through rank 0.  Every operation is: each rank sends its payload to rank 0, rank 0 combines and answers everyone.
// Blocking, a few hundred bytes, tens of microseconds on a loopback socket: set-up and the bench's bookkeeping only.
namespace {

double now_s()
{
  return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

int dist_timeout_s()
{
  const char* e = std::getenv("VNR_AMD_DIST_TIMEOUT");
  const int v = e ? std::atoi(e) : 0;
  return v > 0 ? v : 300;
}

// The rendezvous (accept / connect / hello) is bounded by VNR_AMD_DIST_TIMEOUT.  Afterwards the sockets block: a rank may sit in a
// barrier for as long as another one computes (PSNR / SSIM over an out-of-core volume on rank 0 takes minutes), and a rank that
// dies closes its socket, which ends the wait with "a peer closed the control plane".  VNR_AMD_DIST_STEADY_TIMEOUT (seconds)
// bounds the steady state as well for jobs that want it.
void set_socket_timeouts(int fd, int seconds)
{
  timeval tv;
  tv.tv_sec = seconds;   // 0: block
  tv.tv_usec = 0;
  (void)setsockopt(fd, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));
  (void)setsockopt(fd, SOL_SOCKET, SO_SNDTIMEO, &tv, sizeof(tv));
}

// after the rendezvous a wait on the control plane is as long as the slowest rank's work, so the library does not bound it by default
// (rank 0 writing a large out-of-core file before a barrier, an interactive viewer left idle); a job that wants a rank that
// hangs without closing its socket (GPU hang, deadlocked collective) to end the others sets VNR_AMD_DIST_STEADY_TIMEOUT=<seconds>:
// bench.py sets 1800
int dist_steady_timeout_s()
{
  const char* e = std::getenv("VNR_AMD_DIST_STEADY_TIMEOUT");
  if (!e || !*e) return 0;
  const int v = std::atoi(e);
  return v > 0 ? v : 0;
}

void send_all(int fd, const void* p, size_t n)
{
  const char* c = (const char*)p;
  while (n) {
    const ssize_t k = ::send(fd, c, n, MSG_NOSIGNAL);
    if (k <= 0) {
      if (k < 0 && errno == EINTR) continue;
      throw std::runtime_error(std::string("[vnr dist] control plane send failed: ") + std::strerror(errno));
    }
    c += k; n -= (size_t)k;
  }
}

void recv_all(int fd, void* p, size_t n)
{
  char* c = (char*)p;
  while (n) {
    const ssize_t k = ::recv(fd, c, n, 0);
    if (k == 0) throw std::runtime_error("[vnr dist] a peer closed the control plane (did another rank fail?)");
    if (k < 0) {
      if (errno == EINTR) continue;
      throw std::runtime_error(std::string("[vnr dist] control plane receive failed or timed out: ") + std::strerror(errno));
    }
    c += k; n -= (size_t)k;
  }
}

bool is_loopback(const std::string& host)
{
  return host.empty() || host == "127.0.0.1" || host == "localhost" || host == "::1";
}

}  // namespace

class ControlPlane {
public:
  // address: "unix:<name>" (abstract socket, this host only) or "tcp:<host>:<port>"
  ControlPlane(int rank, int world, const std::string& address) : rank_(rank), world_(world)
  {
    if (world <= 1) return;
    try {
      connect_all(rank, world, address);
    } catch (...) {   // the destructor of a half-built object does not run: close what was opened
      close_all();
      throw;
    }
    // rendezvous done: from here on a wait is as long as the slowest rank's work
    const int steady = dist_steady_timeout_s();
    if (up_ >= 0) set_socket_timeouts(up_, steady);
    for (int fd : peers_) if (fd >= 0) set_socket_timeouts(fd, steady);
  }

  ~ControlPlane() { close_all(); }

private:
  void close_all()
  {
    if (up_ >= 0) ::close(up_);
    up_ = -1;
    for (int& fd : peers_) { if (fd >= 0) ::close(fd); fd = -1; }
  }

  void connect_all(int rank, int world, const std::string& address)
  {
    const bool is_unix = address.rfind("unix:", 0) == 0;
    std::string host;
    int port = 0;
    sockaddr_un ua;
    socklen_t ulen = 0;
    if (is_unix) {
      const std::string name = address.substr(5);
      std::memset(&ua, 0, sizeof(ua));
      ua.sun_family = AF_UNIX;
      if (name.size() + 2 > sizeof(ua.sun_path)) throw std::runtime_error("[vnr dist] unix socket name too long");
      std::memcpy(ua.sun_path + 1, name.data(), name.size());  // leading NUL: abstract namespace, gone with the process
      ulen = (socklen_t)(offsetof(sockaddr_un, sun_path) + 1 + name.size());
    } else if (address.rfind("tcp:", 0) == 0) {
      const size_t c = address.rfind(':');
      if (c == std::string::npos || c < 4) throw std::runtime_error("[vnr dist] malformed address " + address);
      host = address.substr(4, c - 4);
      port = std::atoi(address.c_str() + c + 1);
    } else {
      throw std::runtime_error("[vnr dist] address must be unix:<name> or tcp:<host>:<port>, got " + address);
    }

    if (rank == 0) {
      const int ls = ::socket(is_unix ? AF_UNIX : AF_INET, SOCK_STREAM, 0);
      if (ls < 0) throw std::runtime_error("[vnr dist] socket() failed");
      int rc;
      if (is_unix) {
        rc = ::bind(ls, (sockaddr*)&ua, ulen);
      } else {
        int one = 1;
        (void)setsockopt(ls, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
        sockaddr_in a;
        std::memset(&a, 0, sizeof(a));
        a.sin_family = AF_INET;
        a.sin_addr.s_addr = htonl(INADDR_ANY);
        a.sin_port = htons((uint16_t)port);
        rc = ::bind(ls, (sockaddr*)&a, sizeof(a));
      }
      if (rc != 0 || ::listen(ls, world) != 0) {
        const std::string why = std::strerror(errno);
        ::close(ls);
        throw std::runtime_error("[vnr dist] rank 0 cannot listen on " + address + ": " + why + " (set VNR_AMD_DIST_ADDR)");
      }
      peers_.assign(world, -1);
      timeval tv; tv.tv_sec = dist_timeout_s(); tv.tv_usec = 0;
      (void)setsockopt(ls, SOL_SOCKET, SO_RCVTIMEO, &tv, sizeof(tv));   // bounds accept()
      for (int k = 1; k < world; ++k) {
        const int fd = ::accept(ls, nullptr, nullptr);
        if (fd < 0) { ::close(ls); throw std::runtime_error("[vnr dist] rank 0 timed out waiting for " + std::to_string(world - k) + " more rank(s) on " + address); }
        set_socket_timeouts(fd, dist_timeout_s());
        if (!is_unix) { int one = 1; (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one)); }
        int32_t hello[2];
        try { recv_all(fd, hello, sizeof(hello)); } catch (...) { ::close(fd); ::close(ls); throw; }
        if (hello[1] != world || hello[0] <= 0 || hello[0] >= world || peers_[hello[0]] != -1) {
          ::close(fd); ::close(ls);
          throw std::runtime_error("[vnr dist] unexpected rank " + std::to_string(hello[0]) + " / world " + std::to_string(hello[1]) + " on the control plane");
        }
        peers_[hello[0]] = fd;
      }
      ::close(ls);
    } else {
      const double deadline = now_s() + dist_timeout_s();
      for (;;) {
        int fd = -1, rc = -1;
        if (is_unix) {
          fd = ::socket(AF_UNIX, SOCK_STREAM, 0);
          if (fd >= 0) rc = ::connect(fd, (sockaddr*)&ua, ulen);
        } else {
          addrinfo hints, *res = nullptr;
          std::memset(&hints, 0, sizeof(hints));
          hints.ai_family = AF_INET; hints.ai_socktype = SOCK_STREAM;
          if (getaddrinfo(host.c_str(), std::to_string(port).c_str(), &hints, &res) == 0 && res) {
            fd = ::socket(res->ai_family, res->ai_socktype, res->ai_protocol);
            if (fd >= 0) rc = ::connect(fd, res->ai_addr, res->ai_addrlen);
            freeaddrinfo(res);
          }
        }
        if (rc == 0) {
          set_socket_timeouts(fd, dist_timeout_s());
          if (!is_unix) { int one = 1; (void)setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one)); }
          up_ = fd;
          break;
        }
        if (fd >= 0) ::close(fd);
        if (now_s() > deadline) throw std::runtime_error("[vnr dist] rank " + std::to_string(rank) + " cannot reach rank 0 at " + address);
        std::this_thread::sleep_for(std::chrono::milliseconds(20));   // rank 0 is not listening yet
      }
      const int32_t hello[2] = {rank, world};
      send_all(up_, hello, sizeof(hello));
    }
  }

public:
  // gather `bytes` from every rank to rank 0 (in rank order), let rank 0 transform the world * bytes, send `out_bytes` back
  template <typename F>
  void exchange(const void* mine, size_t bytes, void* out, size_t out_bytes, F&& combine_on_root)
  {
    if (world_ <= 1) {
      std::vector<char> all((const char*)mine, (const char*)mine + bytes);
      combine_on_root(all.data(), (char*)out);
      return;
    }
    // an empty payload still travels as one token byte: the answer must not leave rank 0 before everyone has arrived
    char token = 1;
    if (rank_ == 0) {
      std::vector<char> all(bytes * (size_t)world_ + 1);
      if (bytes) std::memcpy(all.data(), mine, bytes);
      for (int r = 1; r < world_; ++r) {
        if (bytes) recv_all(peers_[r], all.data() + bytes * (size_t)r, bytes);
        else recv_all(peers_[r], &token, 1);
      }
      std::vector<char> res(out_bytes ? out_bytes : 1);
      combine_on_root(all.data(), res.data());
      for (int r = 1; r < world_; ++r) { if (out_bytes) send_all(peers_[r], res.data(), out_bytes); else send_all(peers_[r], &token, 1); }
      if (out_bytes) std::memcpy(out, res.data(), out_bytes);
    } else {
      if (bytes) send_all(up_, mine, bytes);
      else send_all(up_, &token, 1);
      if (out_bytes) recv_all(up_, out, out_bytes);
      else recv_all(up_, &token, 1);
    }
  }

  void barrier()
  {
    exchange(nullptr, 0, nullptr, 0, [](const char*, char*) {});
  }

private:
  int rank_, world_;
  int up_ = -1;
  std::vector<int> peers_;
};

// ================================================================================================ RCCL (dlopen)
namespace {

// The few declarations of rccl.h that are used (ROCm 7.2, /opt/rocm/include/rccl/rccl.h: ncclUniqueId is 128 opaque bytes,
// ncclDataType_t / ncclRedOp_t values as in NCCL 2.x), so that libvnr_amd.so neither needs the header nor links the library.
struct NcclUniqueId { char internal[128]; };
typedef void* NcclComm;
enum { kNcclSuccess = 0 };
enum { kNcclUint8 = 1, kNcclFloat16 = 6, kNcclFloat32 = 7 };
enum { kNcclSum = 0, kNcclMax = 2, kNcclMin = 3, kNcclAvg = 4 };

struct RcclApi {
  void* handle = nullptr;
  int (*GetUniqueId)(NcclUniqueId*) = nullptr;
  int (*CommInitRank)(NcclComm*, int, NcclUniqueId, int) = nullptr;
  int (*CommDestroy)(NcclComm) = nullptr;
  int (*CommCount)(NcclComm, int*) = nullptr;   // optional
  int (*AllGather)(const void*, void*, size_t, int, NcclComm, hipStream_t) = nullptr;
  int (*AllReduce)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*ReduceScatter)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  int (*Broadcast)(const void*, void*, size_t, int, int, NcclComm, hipStream_t) = nullptr;
  const char* (*GetErrorString)(int) = nullptr;

  static RcclApi& get()
  {
    static RcclApi api;
    if (api.handle) return api;
    // Which librccl: the one next to the HIP runtime this process already uses.  A process that imported PyTorch first runs on
    // the wheel's bundled ROCm, and the system librccl would bring a second runtime; VNR_AMD_RCCL_LIB overrides.
    std::vector<std::string> names;
    if (const char* e = std::getenv("VNR_AMD_RCCL_LIB")) names.push_back(e);
    Dl_info info;
    if (dladdr((void*)&hipGetDeviceCount, &info) && info.dli_fname) {
      std::string dir = info.dli_fname;
      const size_t slash = dir.rfind('/');
      if (slash != std::string::npos) { dir.resize(slash); names.push_back(dir + "/librccl.so.1"); names.push_back(dir + "/librccl.so"); }
    }
    names.push_back("librccl.so.1");
    names.push_back("/opt/rocm/lib/librccl.so.1");
    std::string tried;
    for (const std::string& n : names) {
      api.handle = dlopen(n.c_str(), RTLD_NOW | RTLD_LOCAL);
      if (api.handle) break;
      tried += n + " ";
    }
    if (!api.handle) throw std::runtime_error("[vnr dist] cannot open librccl (tried " + tried + "); set VNR_AMD_RCCL_LIB or VNR_AMD_DIST_TRANSPORT=shm");
    auto sym = [&](const char* name) {
      void* p = dlsym(api.handle, name);
      if (!p) throw std::runtime_error(std::string("[vnr dist] librccl lacks ") + name);
      return p;
    };
    api.GetUniqueId = (int (*)(NcclUniqueId*))sym("ncclGetUniqueId");
    api.CommInitRank = (int (*)(NcclComm*, int, NcclUniqueId, int))sym("ncclCommInitRank");
    api.CommDestroy = (int (*)(NcclComm))sym("ncclCommDestroy");
    api.CommCount = (int (*)(NcclComm, int*))dlsym(api.handle, "ncclCommCount");
    api.AllGather = (int (*)(const void*, void*, size_t, int, NcclComm, hipStream_t))sym("ncclAllGather");
    api.AllReduce = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclAllReduce");
    api.ReduceScatter = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclReduceScatter");
    api.Broadcast = (int (*)(const void*, void*, size_t, int, int, NcclComm, hipStream_t))sym("ncclBroadcast");
    api.GetErrorString = (const char* (*)(int))sym("ncclGetErrorString");
    return api;
  }
};

void rccl_check(int rc, const char* what)
{
  if (rc != kNcclSuccess) throw std::runtime_error(std::string("[vnr dist] ") + what + ": " + RcclApi::get().GetErrorString(rc));
}

int nccl_type(DistDType t) { return t == DistDType::F32 ? kNcclFloat32 : t == DistDType::F16 ? kNcclFloat16 : kNcclUint8; }
int nccl_op(DistOp op) { return op == DistOp::Sum ? kNcclSum : op == DistOp::Max ? kNcclMax : op == DistOp::Min ? kNcclMin : kNcclAvg; }
size_t dtype_bytes(DistDType t) { return t == DistDType::F32 ? 4 : t == DistDType::F16 ? 2 : 1; }

class RcclTransport : public Transport {
public:
  RcclTransport(int rank, int world, const void* unique_id) : rank_(rank)
  {
    RcclApi& api = RcclApi::get();
    NcclUniqueId id;
    std::memcpy(&id, unique_id, sizeof(id));
    rccl_check(api.CommInitRank(&comm_, world, id, rank), "ncclCommInitRank");
  }
  ~RcclTransport() override { if (comm_) (void)RcclApi::get().CommDestroy(comm_); }
  const char* name() const override { return "rccl"; }
  int rccl_ranks() const override
  {
    int n = 0;
    if (!comm_ || !RcclApi::get().CommCount || RcclApi::get().CommCount(comm_, &n) != kNcclSuccess) return 0;
    return n;
  }
  void all_gather(const void* d_send, void* d_recv, size_t bytes, hipStream_t s) override
  {
    rccl_check(RcclApi::get().AllGather(d_send, d_recv, bytes, kNcclUint8, comm_, s), "ncclAllGather");
  }
  void all_reduce(void* d_buf, size_t count, DistDType t, DistOp op, hipStream_t s) override
  {
    rccl_check(RcclApi::get().AllReduce(d_buf, d_buf, count, nccl_type(t), nccl_op(op), comm_, s), "ncclAllReduce");
  }
  void reduce_scatter(void* d_buf, size_t count_per_rank, DistDType t, DistOp op, hipStream_t s) override
  {
    char* mine = (char*)d_buf + (size_t)rank_ * count_per_rank * dtype_bytes(t);   // in place: recvbuff = sendbuff + rank * count
    rccl_check(RcclApi::get().ReduceScatter(d_buf, mine, count_per_rank, nccl_type(t), nccl_op(op), comm_, s), "ncclReduceScatter");
  }
  void broadcast(void* d_buf, size_t bytes, int root, hipStream_t s) override
  {
    rccl_check(RcclApi::get().Broadcast(d_buf, d_buf, bytes, kNcclUint8, root, comm_, s), "ncclBroadcast");
  }

private:
  int rank_;
  NcclComm comm_ = nullptr;
};

// ================================================================================================ host-staged transport
// One POSIX shared-memory segment per job: a header with a sense-reversing barrier, then one slot of `slot_bytes` per rank.
// A collective moves its buffer in chunks of at most one slot: device -> own slot (stream copy + stream sync), barrier,
// every rank reads what it needs from all slots (reductions in rank order 0 .. world-1 in fp32, so every rank computes the
// same bits), result -> device, barrier before the slots are reused.  The host waits inside every call; that is the price of
// a transport whose purpose is to run several ranks where RCCL cannot (several ranks on one GPU, or no GPU at all).
struct ShmHeader {
  std::atomic<uint32_t> arrived;
  std::atomic<uint32_t> generation;
  uint32_t world;
  uint32_t pad;
  uint64_t slot_bytes;
};

class ShmTransport : public Transport {
public:
  ShmTransport(int rank, int world, Dist& dist) : rank_(rank), world_(world)
  {
    size_t slot_mb = 8;
    if (const char* e = std::getenv("VNR_AMD_SHM_SLOT_MB")) slot_mb = (size_t)std::max(1, std::atoi(e));
    slot_bytes_ = slot_mb << 20;
    total_ = 4096 + slot_bytes_ * (size_t)world;
    char name[96] = {0};
    if (rank == 0) {
      std::snprintf(name, sizeof(name), "/vnr_amd_%d_%llx", (int)getpid(), (unsigned long long)(now_s() * 1e6));
      const int fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
      if (fd < 0 || ftruncate(fd, (off_t)total_) != 0) throw std::runtime_error(std::string("[vnr dist] cannot create the shared segment: ") + std::strerror(errno));
      base_ = (char*)mmap(nullptr, total_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      ::close(fd);
      if (base_ == MAP_FAILED) { shm_unlink(name); throw std::runtime_error("[vnr dist] mmap of the shared segment failed"); }
      ShmHeader* h = new (base_) ShmHeader();
      h->arrived.store(0); h->generation.store(0); h->world = (uint32_t)world; h->slot_bytes = slot_bytes_;
    }
    dist.broadcast_host(name, sizeof(name), 0);
    if (rank != 0) {
      const int fd = shm_open(name, O_RDWR, 0600);
      if (fd < 0) throw std::runtime_error(std::string("[vnr dist] cannot open the shared segment ") + name);
      base_ = (char*)mmap(nullptr, total_, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
      ::close(fd);
      if (base_ == MAP_FAILED) throw std::runtime_error("[vnr dist] mmap of the shared segment failed");
    }
    dist.barrier();                 // everyone has mapped it ...
    if (rank == 0) shm_unlink(name);   // ... so the name can go: the segment lives as long as a mapping does
    // page-locked, so that the stream copies are real DMA (no effect without a device: the CPU-only tests use host buffers)
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) == hipSuccess && n_dev > 0 && Runtime::get().ready()) registered_ = hipHostRegister(base_, total_, hipHostRegisterDefault) == hipSuccess;
    else (void)hipGetLastError();
  }
  ~ShmTransport() override
  {
    if (registered_) (void)hipHostUnregister(base_);
    if (base_ && base_ != MAP_FAILED) munmap(base_, total_);
  }
  const char* name() const override { return "shm"; }

  void all_gather(const void* d_send, void* d_recv, size_t bytes, hipStream_t s) override
  {
    for (size_t off = 0; off < bytes || off == 0; off += slot_bytes_) {
      const size_t n = std::min(slot_bytes_, bytes - off);
      to_slot((const char*)d_send + off, n, s);
      barrier();
      for (int r = 0; r < world_; ++r) from_host((char*)d_recv + (size_t)r * bytes + off, slot(r), n, s);
      finish(s);
      if (bytes == 0) break;
    }
  }

  void all_reduce(void* d_buf, size_t count, DistDType t, DistOp op, hipStream_t s) override
  {
    const size_t eb = dtype_bytes(t), per = slot_bytes_ / eb;
    std::vector<char> res;
    for (size_t off = 0; off < count; off += per) {
      const size_t n = std::min(per, count - off);
      to_slot((const char*)d_buf + off * eb, n * eb, s);
      barrier();
      res.resize(n * eb);
      reduce_slots(res.data(), 0, n, t, op);
      from_host((char*)d_buf + off * eb, res.data(), n * eb, s);
      finish(s);
    }
  }

  void reduce_scatter(void* d_buf, size_t count_per_rank, DistDType t, DistOp op, hipStream_t s) override
  {
    // rank r needs the sum of everyone's slice r: move slice by slice (world rounds of at most one slot each)
    const size_t eb = dtype_bytes(t), per = slot_bytes_ / eb;
    std::vector<char> res;
    for (int r = 0; r < world_; ++r) {
      for (size_t off = 0; off < count_per_rank; off += per) {
        const size_t n = std::min(per, count_per_rank - off);
        char* p = (char*)d_buf + ((size_t)r * count_per_rank + off) * eb;
        to_slot(p, n * eb, s);
        barrier();
        if (r == rank_) {
          res.resize(n * eb);
          reduce_slots(res.data(), 0, n, t, op);
          from_host(p, res.data(), n * eb, s);
        }
        finish(s);
      }
    }
  }

  void broadcast(void* d_buf, size_t bytes, int root, hipStream_t s) override
  {
    for (size_t off = 0; off < bytes; off += slot_bytes_) {
      const size_t n = std::min(slot_bytes_, bytes - off);
      if (rank_ == root) to_slot((const char*)d_buf + off, n, s);
      barrier();
      if (rank_ != root) from_host((char*)d_buf + off, slot(root), n, s);
      finish(s);
    }
  }

private:
  char* slot(int r) const { return base_ + 4096 + slot_bytes_ * (size_t)r; }
  static bool on_device(const void* p)
  {
    hipPointerAttribute_t a;
    if (hipPointerGetAttributes(&a, p) != hipSuccess) { (void)hipGetLastError(); return false; }
    return a.type == hipMemoryTypeDevice || a.type == hipMemoryTypeManaged;
  }
  void to_slot(const void* src, size_t n, hipStream_t s)
  {
    if (!n) return;
    if (on_device(src)) {
      VNR_HIP_CHECK(hipMemcpyAsync(slot(rank_), src, n, hipMemcpyDeviceToHost, s));
      VNR_HIP_CHECK(hipStreamSynchronize(s));
    } else {
      std::memcpy(slot(rank_), src, n);   // host buffers: the CPU-only tests
    }
  }
  void from_host(void* dst, const void* src, size_t n, hipStream_t s)
  {
    if (!n) return;
    if (on_device(dst)) VNR_HIP_CHECK(hipMemcpyAsync(dst, src, n, hipMemcpyHostToDevice, s));
    else std::memcpy(dst, src, n);
    pending_device_ = pending_device_ || on_device(dst);
  }
  void finish(hipStream_t s)
  {
    // the copies out of the slots (and out of `res`) must have completed before any rank overwrites a slot
    if (pending_device_) VNR_HIP_CHECK(hipStreamSynchronize(s));
    pending_device_ = false;
    barrier();
  }
  void reduce_slots(char* out, size_t first, size_t n, DistDType t, DistOp op)
  {
    const bool sum = op == DistOp::Sum || op == DistOp::Avg;
    auto combine = [op, sum](float a, float b) { return sum ? a + b : op == DistOp::Max ? (a > b ? a : b) : (a < b ? a : b); };
    const float post = op == DistOp::Avg ? 1.0f / (float)world_ : 1.0f;   // exact for power-of-two worlds
    if (t == DistDType::F32) {
      float* o = (float*)out;
      for (size_t i = 0; i < n; ++i) {
        float acc = ((const float*)slot(0))[first + i];
        for (int r = 1; r < world_; ++r) acc = combine(acc, ((const float*)slot(r))[first + i]);
        o[i] = acc * post;
      }
    } else if (t == DistDType::F16) {
      uint16_t* o = (uint16_t*)out;
      for (size_t i = 0; i < n; ++i) {
        float acc = f16_to_f32(((const uint16_t*)slot(0))[first + i]);
        for (int r = 1; r < world_; ++r) acc = combine(acc, f16_to_f32(((const uint16_t*)slot(r))[first + i]));
        o[i] = f32_to_f16(acc * post);
      }
    } else {
      uint8_t* o = (uint8_t*)out;
      for (size_t i = 0; i < n; ++i) {
        uint32_t acc = ((const uint8_t*)slot(0))[first + i];
        for (int r = 1; r < world_; ++r) {
          const uint32_t v = ((const uint8_t*)slot(r))[first + i];
          acc = sum ? acc + v : op == DistOp::Max ? std::max(acc, v) : std::min(acc, v);
        }
        o[i] = (uint8_t)(op == DistOp::Avg ? acc / (uint32_t)world_ : acc);
      }
    }
  }
  void barrier()
  {
    ShmHeader* h = (ShmHeader*)base_;
    const uint32_t gen = h->generation.load(std::memory_order_acquire);
    if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world_) {
      h->arrived.store(0, std::memory_order_relaxed);
      h->generation.store(gen + 1, std::memory_order_release);
      return;
    }
    const double deadline = now_s() + dist_timeout_s();
    uint32_t spins = 0;
    while (h->generation.load(std::memory_order_acquire) == gen) {
      if (++spins < 2000) continue;
      std::this_thread::yield();
      if ((spins & 0xfff) == 0 && now_s() > deadline) throw std::runtime_error("[vnr dist] shared-memory barrier timed out (did another rank fail?)");
    }
  }

  int rank_, world_;
  size_t slot_bytes_ = 0, total_ = 0;
  char* base_ = nullptr;
  bool registered_ = false, pending_device_ = false;
};

}  // namespace

// ================================================================================================ Dist
Dist::Dist() = default;
// Exit-time destruction of the process-wide object: librccl was opened with dlopen AFTER this object was constructed, so its own
// statics (and possibly the HIP runtime's) are gone by the time this runs, and ncclCommDestroy from here is a known hang or crash at
// exit.  A communicator (and the communication stream) that the application did not release with vnrAmdDistFinalize is therefore
// LEAKED here, deliberately: the process is ending and the driver reclaims it.  The host-staged transport owns only a mapping.
Dist::~Dist()
{
  if (transport_ && std::strcmp(transport_->name(), "rccl") == 0) (void)transport_.release();
  else transport_.reset();
  ctl_.reset();
}

Dist& Dist::get()
{
  static Dist d;
  return d;
}

Transport& Dist::transport()
{
  if (!transport_) throw std::runtime_error("[vnr dist] not initialised: call vnrAmdDistInitFromEnv / vnrAmdDistInit first");
  return *transport_;
}

void rccl_unique_id(void* out128)
{
  NcclUniqueId id;
  rccl_check(RcclApi::get().GetUniqueId(&id), "ncclGetUniqueId");
  std::memcpy(out128, &id, sizeof(id));
}

void Dist::init(int rank, int world, int local_rank, const void* unique_id, const char* transport, const char* address)
{
  if (transport_) throw std::runtime_error("[vnr dist] already initialised");
  if (world < 1 || rank < 0 || rank >= world) throw std::runtime_error("[vnr dist] invalid rank / world size");
  std::string tname = transport && *transport ? transport : "";
  if (tname.empty()) { const char* e = std::getenv("VNR_AMD_DIST_TRANSPORT"); tname = e && *e ? e : "rccl"; }
  if (tname != "rccl" && tname != "shm") throw std::runtime_error("[vnr dist] unknown transport '" + tname + "' (rccl, shm)");
  rank_ = rank; world_ = world; local_rank_ = local_rank;
  try {
    ctl_.reset(new ControlPlane(rank, world, address ? address : ""));
    if (tname == "rccl") {
      if (!Runtime::get().ready()) Runtime::get().init(local_rank);
      NcclUniqueId id;
      if (unique_id) std::memcpy(&id, unique_id, sizeof(id));
      else {
        if (rank == 0) rccl_unique_id(&id);
        broadcast_host(&id, sizeof(id), 0);
      }
      transport_.reset(new RcclTransport(rank, world, &id));
    } else {
      transport_.reset(new ShmTransport(rank, world, *this));
    }
  } catch (...) {
    transport_.reset(); ctl_.reset();
    rank_ = 0; world_ = 1; local_rank_ = 0;
    throw;
  }
}

void Dist::init_from_env()
{
  auto geti = [](const char* k, int def) { const char* e = std::getenv(k); return e && *e ? std::atoi(e) : def; };
  const int rank = geti("RANK", 0), world = geti("WORLD_SIZE", 1), local_rank = geti("LOCAL_RANK", geti("RANK", 0));
  std::string address;
  if (const char* e = std::getenv("VNR_AMD_DIST_ADDR")) address = e;
  if (address.empty()) {
    const char* ma = std::getenv("MASTER_ADDR");
    const std::string host = ma ? ma : "127.0.0.1";
    const int port = geti("MASTER_PORT", 29531);
    // MASTER_PORT itself belongs to the launcher (torchrun keeps its own store there).  On one node an abstract unix socket
    // named after it cannot collide with anything and disappears with rank 0; across nodes TCP on the next port.
    if (is_loopback(host)) address = "unix:vnr_amd_dist_" + std::to_string(port) + "_" + std::to_string((int)getuid());
    else address = "tcp:" + host + ":" + std::to_string(port + 1);
  }
  init(rank, world, local_rank, nullptr, nullptr, address.c_str());
}

void Dist::finalize()
{
  if (comm_stream_) { (void)hipStreamSynchronize(comm_stream_); (void)hipStreamDestroy(comm_stream_); comm_stream_ = nullptr; }
  transport_.reset();
  ctl_.reset();
  rank_ = 0; world_ = 1; local_rank_ = 0;
}

void Dist::barrier()
{
  if (ctl_) ctl_->barrier();
}

void Dist::all_reduce_host(double* values, int n, DistOp op)
{
  if (!ctl_ || world_ <= 1 || n <= 0) return;
  const int world = world_;
  ctl_->exchange(values, sizeof(double) * (size_t)n, values, sizeof(double) * (size_t)n, [&](const char* all, char* out) {
    const double* a = (const double*)all;
    double* o = (double*)out;
    for (int i = 0; i < n; ++i) {
      double acc = a[i];
      for (int r = 1; r < world; ++r) {
        const double v = a[(size_t)r * n + i];
        acc = (op == DistOp::Sum || op == DistOp::Avg) ? acc + v : op == DistOp::Max ? std::max(acc, v) : std::min(acc, v);
      }
      o[i] = op == DistOp::Avg ? acc / (double)world : acc;
    }
  });
}

void Dist::broadcast_host(void* data, size_t bytes, int root)
{
  if (!ctl_ || world_ <= 1 || bytes == 0) return;
  ctl_->exchange(data, bytes, data, bytes, [&](const char* all, char* out) { std::memcpy(out, all + bytes * (size_t)root, bytes); });
}

void Dist::all_gather_host(const void* mine, void* all_out, size_t bytes)
{
  if (!ctl_ || world_ <= 1) { if (bytes) std::memcpy(all_out, mine, bytes); return; }
  const size_t total = bytes * (size_t)world_;
  ctl_->exchange(mine, bytes, all_out, total, [&](const char* all, char* out) { std::memcpy(out, all, total); });
}

// ---- first-contact self-test (bench.py --gpus N runs it before its timed region) --------------------------------------------------
// Every collective the sharded paths use, once, on patterned buffers whose result every rank can compute by itself: the in-place
// all-gather of a frame share (renderer); the sharded optimizer's exchange of a range that divides nothing, sliced exactly as the training
// step slices it (reduce-scatter(Avg) of the 8-aligned slices, all-reduce(Avg) of the remainder, all-gather of the slices); broadcast
// (replica synchronisation); all-reduce(Sum) of fp16 (the unsharded exchange).  A collective
// that does not complete within `deadline_s` (hipStreamQuery polled from the host) or returns other values throws with its name; the
// caller then exits the process (a hung collective cannot be cancelled).
static uint16_t self_test_f16(uint32_t small_int)   // integers < 2048 are exact in fp16
{
  return f32_to_f16((float)small_int);
}

std::string Dist::self_test(double deadline_s)
{
  if (!active()) return "no process group";
  Transport& tr = transport();
  hipStream_t s = comm_stream();
  const int W = world_, R = rank_;
  std::string report;
  auto t_issue = std::chrono::steady_clock::now();   // set by begin() in front of every collective: the report times issue -> complete
  auto begin = [&]() { t_issue = std::chrono::steady_clock::now(); };
  auto finish = [&](const char* what) {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      const hipError_t e = hipStreamQuery(s);
      if (e == hipSuccess) break;
      if (e != hipErrorNotReady) throw std::runtime_error(std::string("[vnr dist] self-test: ") + what + " failed: " + hipGetErrorString(e));
      const double dt = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
      if (dt > deadline_s)
        throw std::runtime_error(std::string("[vnr dist] self-test: ") + what + " did not complete within " + std::to_string(deadline_s) + " s on rank " +
                                 std::to_string(R) + " of " + std::to_string(W) + " (transport " + tr.name() + ")");
      std::this_thread::sleep_for(std::chrono::microseconds(200));
    }
    const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_issue).count();
    report += std::string(what) + " ok (" + std::to_string(ms).substr(0, 6) + " ms from issue to complete); ";
  };
  auto mismatch = [&](const char* what, size_t i, double got, double want) {
    throw std::runtime_error(std::string("[vnr dist] self-test: ") + what + " returned wrong data on rank " + std::to_string(R) + " of " + std::to_string(W) +
                             " (transport " + tr.name() + "): element " + std::to_string(i) + " is " + std::to_string(got) + ", expected " + std::to_string(want));
  };
  // 1. in-place all-gather of a share: 131 072 pixels x 16 B (a 1/8 share of the 1024 x 1024 frame), slot r = pattern(r, i)
  {
    const size_t n = 131072 * 4;   // floats per rank
    DeviceBuffer<float> buf;
    buf.resize(n * (size_t)W);
    std::vector<float> host(n * (size_t)W, -1.0f);
    for (size_t i = 0; i < n; ++i) host[(size_t)R * n + i] = (float)((i * 31u + (size_t)R * 7u) % 65521u);
    VNR_HIP_CHECK(hipMemcpy(buf.ptr, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice));
    begin(); tr.all_gather(buf.ptr + (size_t)R * n, buf.ptr, n * sizeof(float), s);
    finish("all-gather (in place, 2 MiB share)");
    VNR_HIP_CHECK(hipMemcpy(host.data(), buf.ptr, host.size() * sizeof(float), hipMemcpyDeviceToHost));
    for (int q = 0; q < W; ++q)
      for (size_t i = 0; i < n; ++i) {
        const float want = (float)((i * 31u + (size_t)q * 7u) % 65521u);
        if (host[(size_t)q * n + i] != want) mismatch("all-gather", (size_t)q * n + i, host[(size_t)q * n + i], want);
      }
  }
  // 2. the sharded optimizer's exchange of ONE range exactly as volume.hip DpState::range_ready / update slice it: a range of 800 011 fp16
  //    gradients (divides nothing): per = (length / world) & ~7 elements are reduce-scattered (Avg), the remainder is all-reduced (Avg), then
  //    the slices are all-gathered in place.  Values are small integers, so every partial sum is exact in fp16 whatever the order; the mean
  //    is exact when the world is a power of two
  {
    const size_t len = 800011, per = (len / (size_t)W) & ~(size_t)7, rest = per * (size_t)W;
    DeviceBuffer<uint16_t> buf;
    buf.resize(len);
    std::vector<uint16_t> host(len);
    auto val = [](int q, size_t i) { return (uint32_t)((i * 5u + (size_t)q * 11u) % 32u); };
    for (size_t i = 0; i < len; ++i) host[i] = self_test_f16(val(R, i));
    VNR_HIP_CHECK(hipMemcpy(buf.ptr, host.data(), len * 2, hipMemcpyHostToDevice));
    if (per) {
      begin(); tr.reduce_scatter(buf.ptr, per, DistDType::F16, DistOp::Avg, s);
      finish("reduce-scatter (Avg, fp16, slices of a range that divides nothing)");
    }
    if (rest < len) {
      begin(); tr.all_reduce(buf.ptr + rest, len - rest, DistDType::F16, DistOp::Avg, s);
      finish("all-reduce (Avg, fp16) of the range's remainder");
    }
    if (per) {
      begin(); tr.all_gather(buf.ptr + (size_t)R * per, buf.ptr, per * 2, s);
      finish("all-gather of the reduced slices");
    }
    VNR_HIP_CHECK(hipMemcpy(host.data(), buf.ptr, len * 2, hipMemcpyDeviceToHost));
    const bool pow2 = (W & (W - 1)) == 0;
    for (size_t i = 0; i < len; ++i) {
      uint32_t sum = 0;
      for (int q = 0; q < W; ++q) sum += val(q, i);
      const double want = (double)sum / W, got = (double)f16_to_f32(host[i]);
      if (pow2 ? got != want : std::fabs(got - want) > 0.02 * std::max(1.0, want)) mismatch("reduce-scatter(Avg) / all-reduce(Avg) / all-gather of a range", i, got, want);
    }
  }
  // 3. broadcast from rank 0 (a length that is not a multiple of 4)
  {
    const size_t n = (1u << 20) + 3;
    DeviceBuffer<uint8_t> buf;
    buf.resize(n);
    std::vector<uint8_t> host(n);
    for (size_t i = 0; i < n; ++i) host[i] = (uint8_t)(R == 0 ? (i * 13u + 5u) : 0xEE);
    VNR_HIP_CHECK(hipMemcpy(buf.ptr, host.data(), n, hipMemcpyHostToDevice));
    begin(); tr.broadcast(buf.ptr, n, 0, s);
    finish("broadcast (1 MiB + 3 bytes)");
    VNR_HIP_CHECK(hipMemcpy(host.data(), buf.ptr, n, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i)
      if (host[i] != (uint8_t)(i * 13u + 5u)) mismatch("broadcast", i, host[i], (uint8_t)(i * 13u + 5u));
  }
  // 4. all-reduce(Sum) of fp16
  {
    const size_t n = 250001;
    DeviceBuffer<uint16_t> buf;
    buf.resize(n);
    std::vector<uint16_t> host(n);
    auto val = [](int q, size_t i) { return (uint32_t)((i * 3u + (size_t)q * 5u) % 16u); };
    for (size_t i = 0; i < n; ++i) host[i] = self_test_f16(val(R, i));
    VNR_HIP_CHECK(hipMemcpy(buf.ptr, host.data(), n * 2, hipMemcpyHostToDevice));
    begin(); tr.all_reduce(buf.ptr, n, DistDType::F16, DistOp::Sum, s);
    finish("all-reduce (Sum, fp16)");
    VNR_HIP_CHECK(hipMemcpy(host.data(), buf.ptr, n * 2, hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) {
      uint32_t sum = 0;
      for (int q = 0; q < W; ++q) sum += val(q, i);
      if ((double)f16_to_f32(host[i]) != (double)sum) mismatch("all-reduce(Sum)", i, f16_to_f32(host[i]), sum);
    }
  }
  return report;
}

hipStream_t Dist::comm_stream()
{
  if (!comm_stream_) {
    if (!Runtime::get().ready()) Runtime::get().init(-1);
    VNR_HIP_CHECK(hipStreamCreateWithFlags(&comm_stream_, hipStreamNonBlocking));
  }
  return comm_stream_;
}

ShareLayout share_layout(uint32_t width, uint32_t height, uint32_t parts)
{
  ShareLayout l;
  l.block = 8u * width;
  l.n_blocks = (height + 7u) / 8u;
  l.per_part = (l.n_blocks + parts - 1u) / parts;
  l.n_local = l.per_part * l.block;
  return l;
}

}  // namespace vnr
