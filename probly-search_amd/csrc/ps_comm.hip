// ps_comm.hip — the multi-GPU exchange of the query path behind the C ABI: one RCCL all-gather
// of the ranks' top-k blocks over xGMI (the reference has no distributed mode — what
// makes the sharding natural is that every Index::query owns its scores / visited maps,
// src/query.rs:31,37, so ranks never talk while scoring).
//
// librccl.so.1 is dlopen()ed on first use: a single-GPU process never loads it, and in a process
// that already holds an RCCL (PyTorch bundles one under the same SONAME) the loader hands back
// that copy instead of a second runtime.
//
// PS_COMM_TRANSPORT=hostshm: debugging transport through POSIX shared memory for ranks that share
// ONE GPU (RCCL rejects two ranks on one device: "Duplicate GPU detected").  It lets a 1-GPU box
// drive the N>1 code path end to end; it is blocking and never selected implicitly.
#include <dlfcn.h>
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <string>
#include <thread>

#include "ps_capi_internal.hpp"
#include "ps_errors.hpp"

namespace {

struct RcclApi {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

RcclApi& rccl() {
  static RcclApi api;
  static std::once_flag once;
  std::call_once(once, [] {
    // One RCCL per process: a host application that already carries one (PyTorch's torch/lib/librccl.so under
    // `bench.py --gpus N`, whose process group is the nccl backend) must not get a second instance with its own
    // bootstrap state next to it.  So first the instance that is already mapped, whatever its path - found in
    // /proc/self/maps, taken with RTLD_NOLOAD -, only then the system library by name.
    {
      FILE* maps = fopen("/proc/self/maps", "r");
      char line[4352];
      while (maps && !api.handle && fgets(line, sizeof(line), maps)) {
        char* path = strchr(line, '/');
        if (!path || !strstr(path, "librccl.so")) continue;
        path[strcspn(path, "\n")] = 0;
        api.handle = dlopen(path, RTLD_NOW | RTLD_GLOBAL | RTLD_NOLOAD);
      }
      if (maps) fclose(maps);
    }
    const char* names[] = {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"};
    for (const char* n : names) {
      if (api.handle) break;
      api.handle = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
    }
    if (!api.handle) return;
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.handle, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.handle, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.handle, "ncclCommDestroy");
    api.AllGather = (decltype(api.AllGather))dlsym(api.handle, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.handle, "ncclGetErrorString");
  });
  if (!api.handle || !api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllGather)
    throw ps::RcclError("librccl.so.1 could not be loaded (RCCL is required for a multi-GPU communicator)");
  return api;
}

// Path of the RCCL instance the communicators use (dladdr of one of its symbols).
std::string rccl_path() {
  RcclApi& api = rccl();
  Dl_info info;
  if (api.AllGather && dladdr(reinterpret_cast<void*>(api.AllGather), &info) && info.dli_fname) return info.dli_fname;
  return "?";
}

void rccl_check(ncclResult_t r, const char* what) {
  if (r == ncclSuccess) return;
  const char* msg = rccl().GetErrorString ? rccl().GetErrorString(r) : "?";
  throw ps::RcclError(std::string("RCCL error in ") + what + ": " + msg);
}

void hip_check(hipError_t e, const char* what) {
  if (e != hipSuccess) throw std::runtime_error(std::string("HIP error: ") + hipGetErrorString(e) + " at " + what);
}

bool hostshm_selected() {
  const char* t = getenv("PS_COMM_TRANSPORT");
  return t && strcmp(t, "hostshm") == 0;
}

// ---- debugging transport: shared-memory segment named after the communicator id -----------------
constexpr size_t SHM_HEADER = 4096;
constexpr size_t SHM_SLOT = (size_t)16 << 20;  // largest block a rank may contribute
struct ShmHeader {
  std::atomic<uint32_t> ready;    // set by rank 0 once the header is initialised
  std::atomic<uint32_t> arrived;  // barrier: ranks that reached the current generation
  std::atomic<uint32_t> generation;
};

void shm_barrier(ShmHeader* h, int world) {
  const uint32_t gen = h->generation.load(std::memory_order_acquire);
  if (h->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == (uint32_t)world) {
    h->arrived.store(0, std::memory_order_relaxed);
    h->generation.fetch_add(1, std::memory_order_release);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (h->generation.load(std::memory_order_acquire) == gen) {
    std::this_thread::yield();
    if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
      throw ps::RcclError("hostshm transport: a rank did not reach the barrier within 120 s");
  }
}

}  // namespace

struct ps_comm {
  int world = 1, rank = 0, device = 0;
  ncclComm_t nccl = nullptr;
  hipStream_t own_stream = nullptr;  // for callers that pass a NULL stream
  // hostshm debugging transport
  bool shm = false;
  std::string shm_name;
  unsigned char* shm_base = nullptr;
  size_t shm_bytes = 0;
  ~ps_comm() {  // (also the clean-up of an init that failed half way)
    if (own_stream) (void)hipStreamDestroy(own_stream);
  }
};

namespace {

template <typename Fn>
ps_status comm_guard(Fn&& fn) {
  try {
    return fn();
  } catch (const std::bad_alloc&) {
    return ps::set_error(PS_ENOMEM, "out of memory");
  } catch (const ps::RcclError& e) {
    return ps::set_error(PS_ERCCL, e.what());
  } catch (const std::invalid_argument& e) {
    return ps::set_error(PS_EINVAL, e.what());
  } catch (const std::exception& e) {
    return ps::set_error(PS_EHIP, e.what());
  }
}

void open_shm(ps_comm& c, const unsigned char* id) {
  char name[64];
  snprintf(name, sizeof(name), "/ps_comm_%02x%02x%02x%02x%02x%02x%02x%02x", id[0], id[1], id[2], id[3], id[4], id[5], id[6],
           id[7]);
  c.shm_name = name;
  c.shm_bytes = SHM_HEADER + (size_t)c.world * SHM_SLOT;
  int fd = -1;
  if (c.rank == 0) {
    fd = shm_open(name, O_CREAT | O_EXCL | O_RDWR, 0600);
    if (fd < 0) throw ps::RcclError(std::string("hostshm transport: shm_open(create) failed for ") + name);
    if (ftruncate(fd, (off_t)c.shm_bytes) != 0) { close(fd); throw ps::RcclError("hostshm transport: ftruncate failed"); }
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    for (;;) {
      fd = shm_open(name, O_RDWR, 0600);
      struct stat st;
      if (fd >= 0 && fstat(fd, &st) == 0 && (size_t)st.st_size >= c.shm_bytes) break;
      if (fd >= 0) close(fd);
      fd = -1;
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
        throw ps::RcclError("hostshm transport: rank 0 never created the segment");
      std::this_thread::sleep_for(std::chrono::milliseconds(2));
    }
  }
  void* p = mmap(nullptr, c.shm_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) throw ps::RcclError("hostshm transport: mmap failed");
  c.shm_base = (unsigned char*)p;
  ShmHeader* h = reinterpret_cast<ShmHeader*>(c.shm_base);
  if (c.rank == 0) {
    h->arrived.store(0);
    h->generation.store(0);
    h->ready.store(1, std::memory_order_release);
  } else {
    const auto t0 = std::chrono::steady_clock::now();
    while (h->ready.load(std::memory_order_acquire) != 1) {
      std::this_thread::yield();
      if (std::chrono::steady_clock::now() - t0 > std::chrono::seconds(120))
        throw ps::RcclError("hostshm transport: rank 0 never initialised the segment");
    }
  }
  shm_barrier(h, c.world);  // everybody is attached
}

}  // namespace

extern "C" {

size_t ps_topk_block_bytes(size_t n_queries, size_t top_k) {
  return n_queries * top_k * 16 + ((n_queries * 4 + 15) & ~(size_t)15);
}

const char* ps_comm_rccl_path(void) {
  static std::string path;
  static std::mutex mu;
  std::lock_guard<std::mutex> lock(mu);
  try {
    path = rccl_path();
  } catch (const std::exception& e) {
    ps::set_error(PS_ERCCL, e.what());
    return nullptr;
  }
  return path.c_str();
}

ps_status ps_comm_get_unique_id(void* id_out) {
  return comm_guard([&]() -> ps_status {
    if (!id_out) return ps::set_error(PS_EINVAL, "null id");
    static_assert(PS_COMM_ID_BYTES == NCCL_UNIQUE_ID_BYTES, "communicator id size");
    if (hostshm_selected()) {
      memset(id_out, 0, PS_COMM_ID_BYTES);
      FILE* f = fopen("/dev/urandom", "rb");
      if (!f || fread(id_out, 1, 16, f) != 16) {
        if (f) fclose(f);
        return ps::set_error(PS_ERCCL, "hostshm transport: cannot read /dev/urandom");
      }
      fclose(f);
      return PS_OK;
    }
    ncclUniqueId id;
    rccl_check(rccl().GetUniqueId(&id), "ncclGetUniqueId");
    memcpy(id_out, &id, PS_COMM_ID_BYTES);
    return PS_OK;
  });
}

ps_status ps_comm_init_rank(const void* id, int world_size, int rank, int device, ps_comm** out) {
  return comm_guard([&]() -> ps_status {
    if (!id || !out) return ps::set_error(PS_EINVAL, "null argument");
    if (world_size < 1 || rank < 0 || rank >= world_size) return ps::set_error(PS_EINVAL, "rank outside [0, world_size)");
    int n_dev = 0;
    if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev <= 0)
      return ps::set_error(PS_ENODEVICE, "no HIP device available");
    if (device < 0 || device >= n_dev) return ps::set_error(PS_EINVAL, "device index out of range");
    std::unique_ptr<ps_comm> c(new ps_comm());
    c->world = world_size;
    c->rank = rank;
    c->device = device;
    hip_check(hipSetDevice(device), "hipSetDevice");
    hip_check(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking), "hipStreamCreateWithFlags");
    if (hostshm_selected()) {
      c->shm = true;
      if (world_size > 1) open_shm(*c, (const unsigned char*)id);
    } else {
      ncclUniqueId uid;
      memcpy(&uid, id, PS_COMM_ID_BYTES);
      rccl_check(rccl().CommInitRank(&c->nccl, world_size, uid, rank), "ncclCommInitRank");
    }
    *out = c.release();
    return PS_OK;
  });
}

void ps_comm_free(ps_comm* c) {
  if (!c) return;
  (void)hipSetDevice(c->device);
  if (c->nccl) (void)rccl().CommDestroy(c->nccl);
  if (c->shm_base) {
    try {
      shm_barrier(reinterpret_cast<ShmHeader*>(c->shm_base), c->world);  // nobody still reads the segment
    } catch (...) {
    }
    munmap(c->shm_base, c->shm_bytes);
    if (c->rank == 0) shm_unlink(c->shm_name.c_str());
  }
  delete c;  // (destroys own_stream)
}

int ps_comm_world_size(const ps_comm* c) { return c ? c->world : 1; }
int ps_comm_rank(const ps_comm* c) { return c ? c->rank : 0; }

ps_status ps_comm_all_gather(ps_comm* comm, const void* d_send, void* d_recv, size_t bytes, void* hip_stream) {
  if (!d_send || !d_recv) return ps::set_error(PS_EINVAL, "null buffer");
  return comm_guard([&]() -> ps_status {
    const int world = comm ? comm->world : 1;
    const int rank = comm ? comm->rank : 0;
    hipStream_t s = (hipStream_t)hip_stream;
    static const bool force = getenv("PS_COMM_FORCE_COLLECTIVE") && *getenv("PS_COMM_FORCE_COLLECTIVE") == '1';  // tests
    if (world == 1 && !(force && comm && comm->nccl)) {  // one rank: no exchange at all
      if (d_recv != d_send) {
        if (s) hip_check(hipMemcpyAsync(d_recv, d_send, bytes, hipMemcpyDeviceToDevice, s), "hipMemcpyAsync");
        else hip_check(hipMemcpy(d_recv, d_send, bytes, hipMemcpyDeviceToDevice), "hipMemcpy");
      }
      return PS_OK;
    }
    if (comm->shm) {  // debugging transport (blocking)
      if (bytes > SHM_SLOT) return ps::set_error(PS_EUNSUPPORTED, "hostshm transport: block larger than 16 MiB");
      if (s) hip_check(hipStreamSynchronize(s), "hipStreamSynchronize");
      ShmHeader* h = reinterpret_cast<ShmHeader*>(comm->shm_base);
      unsigned char* data = comm->shm_base + SHM_HEADER;
      hip_check(hipMemcpy(data + (size_t)rank * SHM_SLOT, d_send, bytes, hipMemcpyDeviceToHost), "hipMemcpy D2H");
      shm_barrier(h, world);
      for (int r = 0; r < world; ++r)
        hip_check(hipMemcpy((unsigned char*)d_recv + (size_t)r * bytes, data + (size_t)r * SHM_SLOT, bytes,
                            hipMemcpyHostToDevice), "hipMemcpy H2D");
      shm_barrier(h, world);  // the slots may be rewritten
      return PS_OK;
    }
    // a NULL stream: the caller's data is complete (its producing call was synchronous); gather on our own stream and wait
    hipStream_t gs = s ? s : comm->own_stream;
    rccl_check(rccl().AllGather(d_send, d_recv, bytes, ncclInt8, comm->nccl, gs), "ncclAllGather");
    if (!s) hip_check(hipStreamSynchronize(gs), "hipStreamSynchronize");
    return PS_OK;
  });
}

ps_status ps_snapshot_query_batch_allgather_flat(ps_snapshot* snap, ps_comm* comm, const ps_scorer_desc* scorer,
                                                 const char* text, const uint64_t* offsets, size_t n_queries,
                                                 const double* fields_boost, size_t n_boost, ps_tokenizer_fn tokenizer,
                                                 void* user, size_t top_k, void* d_local_block, void* d_all_blocks,
                                                 void* hip_stream) {
  if (!d_local_block || !d_all_blocks) return ps::set_error(PS_EINVAL, "null block pointer");
  unsigned char* blk = (unsigned char*)d_local_block;
  const size_t nk = n_queries * top_k;
  // scoring: no collective (queries are independent); ordered on the caller's stream
  ps_status st = ps::run_device_flat(snap, scorer, text, offsets, n_queries, fields_boost, n_boost, tokenizer, user, top_k,
                                     blk, blk + nk * 8, blk + nk * 16, hip_stream);
  if (st != PS_OK) return st;
  return ps_comm_all_gather(comm, d_local_block, d_all_blocks, ps_topk_block_bytes(n_queries, top_k), hip_stream);
}

}  // extern "C"
