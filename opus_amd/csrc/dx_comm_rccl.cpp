// The communicator's collectives over RCCL, inside the library: a CLI or JVM host needs nothing but the 128
// bytes of a unique id (made by rank 0, handed to the others by any means: a file on a shared path, an
// environment variable, a socket) to put one context per GPU into one group.
//
// What the reference does with Spark's exchanges -- CustomPartitioner for the first-level partitions
// (partitioner/DAP.scala:541-551), hashpartitioning(key, 200) under every SortMergeJoin
// (query_processor/QueryProcessor.scala:136-139) -- the engine asks of five collectives (struct dx_comm). Here
// each of them is a handful of RCCL calls queued on the context's own stream, so they are ordered with the
// kernels that fill and read the buffers and nothing waits on the host in between:
//   exchange          all-to-all with per-peer counts: ncclSend / ncclRecv pairs in one group (xGMI is a full mesh
//                     of point-to-point links, so every pair moves over its own link at once); the rank's own
//                     share is a device-to-device copy
//   allgather         the same with one send buffer for all peers (counts differ per rank)
//   allreduce_sum_dev ncclAllReduce in place
//   exchange_host /   small host vectors: staged through a pinned block and a device block owned by the
//   allreduce_sum     communicator, then ncclAllToAll / ncclAllReduce
// RCCL is loaded at run time (dlopen of librccl.so.1): single-GPU users do not need it, and a process that
// already holds an RCCL (PyTorch ships one under the same soname) shares it instead of loading a second.
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <sys/stat.h>
#include <unistd.h>

#include <chrono>
#include <condition_variable>
#include <fstream>
#include <mutex>
#include <thread>

#include "dx_state.h"

namespace dx {
namespace {

struct RcclApi {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclSend) Send = nullptr;
  decltype(&ncclRecv) Recv = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclAllToAll) AllToAll = nullptr;
};

RcclApi& rccl() {
  static RcclApi api;
  if (api.handle) return api;
  void* h = nullptr;
  for (const char* name : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
    h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
    if (h) break;
  }
  if (!h) throw DxError(DX_E_NOENT, std::string("cannot load librccl.so.1: ") + dlerror());
  auto sym = [&](const char* n) {
    void* p = dlsym(h, n);
    if (!p) throw DxError(DX_E_NOENT, std::string("librccl lacks ") + n);
    return p;
  };
  api.GetUniqueId = (decltype(api.GetUniqueId))sym("ncclGetUniqueId");
  api.CommInitRank = (decltype(api.CommInitRank))sym("ncclCommInitRank");
  api.CommDestroy = (decltype(api.CommDestroy))sym("ncclCommDestroy");
  api.GetErrorString = (decltype(api.GetErrorString))sym("ncclGetErrorString");
  api.GroupStart = (decltype(api.GroupStart))sym("ncclGroupStart");
  api.GroupEnd = (decltype(api.GroupEnd))sym("ncclGroupEnd");
  api.Send = (decltype(api.Send))sym("ncclSend");
  api.Recv = (decltype(api.Recv))sym("ncclRecv");
  api.AllReduce = (decltype(api.AllReduce))sym("ncclAllReduce");
  api.AllToAll = (decltype(api.AllToAll))sym("ncclAllToAll");
  api.handle = h;
  return api;
}

constexpr int kHostWords = 64 * 16;  // exchange_host moves at most 16 words per peer, allreduce_sum at most 1024

}  // namespace

struct RcclComm {
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  int device = 0, rank = 0, world = 1;
  uint64_t* pinned = nullptr;  // 2 * kHostWords: send half, receive half
  uint64_t* dev = nullptr;     // the same on the device
  std::string last_error;
};

namespace {

#define DX_NCCL(c, expr)                                                                         \
  do {                                                                                           \
    ncclResult_t _r = (expr);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      (c)->last_error = std::string(#expr) + ": " + rccl().GetErrorString(_r);                   \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)
// inside ncclGroupStart .. ncclGroupEnd: the group is closed before the error is returned, so that the calls that
// follow do not nest in it
#define DX_NCCL_IN_GROUP(c, expr)                                                                \
  do {                                                                                           \
    ncclResult_t _r = (expr);                                                                    \
    if (_r != ncclSuccess) {                                                                     \
      (c)->last_error = std::string(#expr) + ": " + rccl().GetErrorString(_r);                   \
      (void)rccl().GroupEnd();                                                                   \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)
#define DX_HIPRC(c, expr)                                                                        \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      (c)->last_error = std::string(#expr) + ": " + hipGetErrorString(_e);                       \
      return 1;                                                                                  \
    }                                                                                            \
  } while (0)

// counts are in elements of elem_bytes; 8-byte elements travel as 64-bit words, anything else as bytes
inline void wire(int elem_bytes, uint64_t count, ncclDataType_t& type, size_t& n) {
  if (elem_bytes == 8) { type = ncclUint64; n = count; }
  else { type = ncclUint8; n = count * (size_t)elem_bytes; }
}

int rc_exchange(void* user, const void* send_dev, const uint64_t* send_counts, void* recv_dev, const uint64_t* recv_counts, int elem_bytes) {
  RcclComm* c = static_cast<RcclComm*>(user);
  RcclApi& api = rccl();
  const uint8_t* sp = static_cast<const uint8_t*>(send_dev);
  uint8_t* rp = static_cast<uint8_t*>(recv_dev);
  size_t so = 0, ro = 0, self_so = 0, self_ro = 0;
  DX_NCCL(c, api.GroupStart());
  for (int p = 0; p < c->world; ++p) {
    ncclDataType_t ty;
    size_t ns, nr;
    wire(elem_bytes, send_counts[p], ty, ns);
    wire(elem_bytes, recv_counts[p], ty, nr);
    if (p == c->rank) {
      self_so = so;
      self_ro = ro;
    } else {
      if (ns) DX_NCCL_IN_GROUP(c, api.Send(sp + so, ns, ty, p, c->comm, c->stream));
      if (nr) DX_NCCL_IN_GROUP(c, api.Recv(rp + ro, nr, ty, p, c->comm, c->stream));
    }
    so += send_counts[p] * (size_t)elem_bytes;
    ro += recv_counts[p] * (size_t)elem_bytes;
  }
  DX_NCCL(c, api.GroupEnd());
  const size_t mine = send_counts[c->rank] * (size_t)elem_bytes;
  if (mine) DX_HIPRC(c, hipMemcpyAsync(rp + self_ro, sp + self_so, mine, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int rc_allgather(void* user, const void* send_dev, uint64_t send_count, void* recv_dev, const uint64_t* recv_counts, int elem_bytes) {
  RcclComm* c = static_cast<RcclComm*>(user);
  RcclApi& api = rccl();
  uint8_t* rp = static_cast<uint8_t*>(recv_dev);
  size_t ro = 0, self_ro = 0;
  ncclDataType_t ty;
  size_t ns;
  wire(elem_bytes, send_count, ty, ns);
  DX_NCCL(c, api.GroupStart());
  for (int p = 0; p < c->world; ++p) {
    size_t nr;
    wire(elem_bytes, recv_counts[p], ty, nr);
    if (p == c->rank) {
      self_ro = ro;
    } else {
      if (ns) DX_NCCL_IN_GROUP(c, api.Send(send_dev, ns, ty, p, c->comm, c->stream));
      if (nr) DX_NCCL_IN_GROUP(c, api.Recv(rp + ro, nr, ty, p, c->comm, c->stream));
    }
    ro += recv_counts[p] * (size_t)elem_bytes;
  }
  DX_NCCL(c, api.GroupEnd());
  const size_t mine = send_count * (size_t)elem_bytes;
  if (mine && rp + self_ro != send_dev) DX_HIPRC(c, hipMemcpyAsync(rp + self_ro, send_dev, mine, hipMemcpyDeviceToDevice, c->stream));
  return 0;
}

int rc_allreduce_dev(void* user, void* dev, uint64_t n, int elem_bytes) {
  RcclComm* c = static_cast<RcclComm*>(user);
  if (elem_bytes != 1 && elem_bytes != 8) { c->last_error = "allreduce_sum_dev: elements of 1 or 8 bytes"; return 1; }
  if (n) DX_NCCL(c, rccl().AllReduce(dev, dev, n, elem_bytes == 8 ? ncclUint64 : ncclUint8, ncclSum, c->comm, c->stream));
  return 0;
}

int rc_exchange_host(void* user, const uint64_t* send, uint64_t* recv, int n) {
  RcclComm* c = static_cast<RcclComm*>(user);
  const size_t words = (size_t)n * c->world;
  if (n < 1 || words > (size_t)kHostWords) { c->last_error = "exchange_host: too many words"; return 1; }
  memcpy(c->pinned, send, words * 8);
  DX_HIPRC(c, hipMemcpyAsync(c->dev, c->pinned, words * 8, hipMemcpyHostToDevice, c->stream));
  DX_NCCL(c, rccl().AllToAll(c->dev, c->dev + kHostWords, (size_t)n, ncclUint64, c->comm, c->stream));
  DX_HIPRC(c, hipMemcpyAsync(c->pinned + kHostWords, c->dev + kHostWords, words * 8, hipMemcpyDeviceToHost, c->stream));
  DX_HIPRC(c, hipStreamSynchronize(c->stream));
  memcpy(recv, c->pinned + kHostWords, words * 8);
  return 0;
}

int rc_allreduce_host(void* user, uint64_t* values, int n) {
  RcclComm* c = static_cast<RcclComm*>(user);
  if (n < 0 || n > kHostWords) { c->last_error = "allreduce_sum: too many values"; return 1; }
  if (!n) return 0;
  memcpy(c->pinned, values, (size_t)n * 8);
  DX_HIPRC(c, hipMemcpyAsync(c->dev, c->pinned, (size_t)n * 8, hipMemcpyHostToDevice, c->stream));
  DX_NCCL(c, rccl().AllReduce(c->dev, c->dev, (size_t)n, ncclUint64, ncclSum, c->comm, c->stream));
  DX_HIPRC(c, hipMemcpyAsync(c->pinned, c->dev, (size_t)n * 8, hipMemcpyDeviceToHost, c->stream));
  DX_HIPRC(c, hipStreamSynchronize(c->stream));
  memcpy(values, c->pinned, (size_t)n * 8);
  return 0;
}

}  // namespace

void rccl_comm_destroy(dx_ctx* ctx) {
  RcclComm* c = ctx->native_comm;
  if (!c) return;
  ctx->native_comm = nullptr;
  if (ctx->comm.user == c) {
    ctx->comm = dx_comm{};
    ctx->comm.world = 1;
  }
  (void)hipStreamSynchronize(c->stream);
  if (c->comm) (void)rccl().CommDestroy(c->comm);
  if (c->pinned) (void)hipHostFree(c->pinned);
  if (c->dev) (void)hipFree(c->dev);
  delete c;
}

const char* rccl_comm_last_error(dx_ctx* ctx) { return ctx->native_comm ? ctx->native_comm->last_error.c_str() : ""; }

void rccl_unique_id(unsigned char* out) {
  ncclUniqueId id;
  const ncclResult_t r = rccl().GetUniqueId(&id);
  if (r != ncclSuccess) throw DxError(DX_E_IO, std::string("ncclGetUniqueId: ") + rccl().GetErrorString(r));
  static_assert(sizeof id.internal == DX_RCCL_ID_BYTES, "unique id size");
  memcpy(out, id.internal, DX_RCCL_ID_BYTES);
}

// ncclCommInitRank returns when every rank of the group has called it and has no time limit of its own: a rank whose
// peers never come (one of them failed before it got here, or read the id of another job) would wait for ever. The call
// therefore runs on a thread of its own; when it has not come back after timeout_ms the caller gets DX_E_IO, and the
// thread, which cannot be called back, is left to itself with everything it uses (the process is expected to end).
void rccl_comm_create(dx_ctx* ctx, const unsigned char* id_bytes, int rank, int world, int timeout_ms) {
  if (world < 1 || world > 64 || rank < 0 || rank >= world) throw DxError(DX_E_INVAL, "rank / world out of range (at most 64 ranks)");
  rccl_comm_destroy(ctx);
  // (a group of one rank is a real RCCL communicator too: the engine runs as on a single GPU, since world == 1,
  // and dx_comm_selftest can drive the collectives through it)
  RcclApi& api = rccl();
  if (timeout_ms <= 0) {
    const char* e = getenv("DX_RCCL_INIT_TIMEOUT_MS");
    timeout_ms = e && atoi(e) > 0 ? atoi(e) : 600000;
  }
  struct Init {
    std::mutex mu;
    std::condition_variable cv;
    bool done = false, abandoned = false;
    ncclResult_t result = ncclSuccess;
    ncclComm_t comm = nullptr;
    ncclUniqueId id;
  };
  auto init = std::make_shared<Init>();
  memcpy(init->id.internal, id_bytes, DX_RCCL_ID_BYTES);
  const int device = ctx->device;
  std::thread([init, device, rank, world, &api] {
    (void)hipSetDevice(device);
    ncclComm_t comm = nullptr;
    const ncclResult_t r = api.CommInitRank(&comm, world, init->id, rank);
    std::unique_lock<std::mutex> l(init->mu);
    if (init->abandoned) {  // nobody is waiting any more: the group that formed after all is left again
      l.unlock();
      if (r == ncclSuccess && comm) (void)api.CommDestroy(comm);
      return;
    }
    init->result = r;
    init->comm = comm;
    init->done = true;
    init->cv.notify_all();
  }).detach();
  {
    std::unique_lock<std::mutex> l(init->mu);
    if (!init->cv.wait_for(l, std::chrono::milliseconds(timeout_ms), [&] { return init->done; })) {
      init->abandoned = true;
      throw DxError(DX_E_IO, "ncclCommInitRank: the " + std::to_string(world) + " ranks of the group did not all arrive within " + std::to_string(timeout_ms) + " ms");
    }
  }
  if (init->result != ncclSuccess) throw DxError(DX_E_IO, std::string("ncclCommInitRank: ") + api.GetErrorString(init->result));
  auto c = std::make_unique<RcclComm>();
  c->comm = init->comm;
  c->device = ctx->device;
  c->rank = rank;
  c->world = world;
  c->stream = ctx->stream;
  try {
    void* p = nullptr;
    DX_HIP(hipHostMalloc(&p, 2 * kHostWords * 8, hipHostMallocDefault));
    c->pinned = static_cast<uint64_t*>(p);
    DX_HIP(hipMalloc(&p, 2 * kHostWords * 8));
    c->dev = static_cast<uint64_t*>(p);
  } catch (...) {  // the group must not outlive a half-made communicator
    (void)api.CommDestroy(c->comm);
    if (c->pinned) (void)hipHostFree(c->pinned);
    throw;
  }
  dx_comm cm{};
  cm.rank = rank;
  cm.world = world;
  cm.user = c.get();
  cm.exchange_host = rc_exchange_host;
  cm.exchange = rc_exchange;
  cm.allreduce_sum = rc_allreduce_host;
  cm.allgather = rc_allgather;
  cm.allreduce_sum_dev = rc_allreduce_dev;
  cm.stream_ordered = 1;
  ctx->comm = cm;
  ctx->native_comm = c.release();
}

// The file a job's ranks meet at: `path` itself, or `path.<job in hex>` when the caller names the job (any number all
// ranks of one launch share and other launches do not: a launcher's job id, its rendezvous port and start time). With a
// job number a file left by an earlier launch has another name and is never read.
std::string rccl_rendezvous_name(const char* path, uint64_t job) {
  if (!job) return path;
  char suffix[32];
  snprintf(suffix, sizeof suffix, ".%llx", (unsigned long long)job);
  return std::string(path) + suffix;
}
void rccl_rendezvous_done(const std::string& file) { (void)unlink(file.c_str()); }

// Rank 0 removes whatever lies at the name, then writes the id there (through a temporary name, so that a reader never
// sees half of it); the others wait for the file to appear. Rank 0 removes the file again once the group has formed
// (rccl_rendezvous_done): every rank has read it by then. Without a job number in the name one case is left open: a
// file that a launch which died before its group formed has left behind, read by a rank that starts before this launch's
// rank 0 has removed it -- that rank then waits in ncclCommInitRank until its time limit and returns DX_E_IO.
void rccl_rendezvous_file(const std::string& p, int rank, int timeout_ms, unsigned char* id) {
  if (rank == 0) {
    (void)unlink(p.c_str());
    rccl_unique_id(id);
    const std::string tmp = p + ".tmp" + std::to_string((long)getpid());
    {
      std::ofstream f(tmp, std::ios::binary | std::ios::trunc);
      f.write(reinterpret_cast<const char*>(id), DX_RCCL_ID_BYTES);
      if (!f) throw DxError(DX_E_IO, "cannot write " + tmp);
    }
    if (rename(tmp.c_str(), p.c_str()) != 0) throw DxError(DX_E_IO, "cannot rename " + tmp + " to " + p);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  for (;;) {
    struct stat st;
    if (stat(p.c_str(), &st) == 0 && st.st_size == DX_RCCL_ID_BYTES) {
      std::ifstream f(p, std::ios::binary);
      f.read(reinterpret_cast<char*>(id), DX_RCCL_ID_BYTES);
      if (f.gcount() == DX_RCCL_ID_BYTES) return;
    }
    if (std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count() > timeout_ms)
      throw DxError(DX_E_IO, "no unique id appeared at " + p + " within " + std::to_string(timeout_ms) + " ms");
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
  }
}

}  // namespace dx
