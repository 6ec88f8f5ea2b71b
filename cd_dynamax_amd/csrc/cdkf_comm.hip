// cdkf_comm.hip -- the ONE collective of the path behind the C ABI (include/cdkf.h, "data-parallel reduction").
//
// Trajectories share the parameters and nothing else (/root/reference/src/ssm_temissions.py:555-568: `vmap(...)(...).sum()`), so a
// data-parallel sweep exchanges 1 double (marginal_log_prob) or 1 + n_theta doubles (value_and_grad of the fit_sgd / fit_mcmc
// objective, ssm_temissions.py:550-568, 665-679) per step: an in-place ncclAllReduce(sum, double) over RCCL / xGMI on the
// stream the sweep and cdkf_ll_sum_*_dev ran on -- no host round trip between the sweep and the reduced sum.
//
// Two ways in, as SURVEY.md section 8b/e lists them:
//   * one process per GPU: cdkf_comm_init_rank (ncclCommInitRank with an id that rank 0 made and the ranks exchanged -- e.g. over
//     the small TCP rendezvous below, which needs nothing but MASTER_ADDR / a port);
//   * one process driving every GPU of the node: cdkf_comm_init_all (ncclCommInitAll) + cdkf_ll_allreduce_all (grouped).
// RCCL is loaded at first use (dlopen), so the library itself loads -- and every sweep runs -- on hosts without it.
//
// The rendezvous (cdkf_rdv_*) is host-only plumbing: a star over TCP with rank 0 at the centre, used to hand the 128-byte RCCL
// id around and for the host-side sums of a few doubles that precede a run (global batch size, timing maxima).  It is
// deterministic (rank 0 adds in rank order) and runs without a GPU, which is how the N > 1 composition is tested on CPU.
#include <arpa/inet.h>
#include <dlfcn.h>
#include <netdb.h>
#include <netinet/in.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <chrono>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

#include <rccl/rccl.h>

#include "cdkf_host.h"

using cdkf::set_error;

// ---------------------------------------------------------------- RCCL, loaded on demand ------------------------------------
namespace {

struct Rccl {
  void* handle = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*GroupStart)() = nullptr;
  ncclResult_t (*GroupEnd)() = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};

std::mutex g_rccl_mutex;
Rccl g_rccl;

// CDKF_RCCL_PATH names the library to use (the Python front-end points it at the copy that belongs to the HIP runtime the
// process already holds); otherwise the system's.
const Rccl* rccl() {
  std::lock_guard<std::mutex> lock(g_rccl_mutex);
  if (g_rccl.handle) return &g_rccl;
  const char* env = std::getenv("CDKF_RCCL_PATH");
  void* h = nullptr;
  if (env && env[0]) {
    // a library the caller NAMED is the only candidate: falling back to the system's copy would pair this process's HIP runtime with
    // an RCCL built against another one -- the failure is reported (and agreed on by all ranks, cdkf_comm_preflight) instead
    h = dlopen(env, RTLD_NOW | RTLD_LOCAL);
    if (!h) {
      set_error("RCCL is not available: CDKF_RCCL_PATH=%s: %s", env, dlerror());
      return nullptr;
    }
  } else {
    for (const char* nm : {"librccl.so.1", "librccl.so"}) {
      h = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (h) break;
    }
    if (!h) {
      set_error("RCCL is not available: %s", dlerror());
      return nullptr;
    }
  }
  Rccl r;
  r.handle = h;
#define CDKF_SYM(field, name)                                              \
  r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name));           \
  if (!r.field) {                                                          \
    set_error("RCCL symbol %s is missing: %s", name, dlerror());           \
    dlclose(h);                                                            \
    return nullptr;                                                        \
  }
  CDKF_SYM(GetUniqueId, "ncclGetUniqueId")
  CDKF_SYM(CommInitRank, "ncclCommInitRank")
  CDKF_SYM(CommInitAll, "ncclCommInitAll")
  CDKF_SYM(AllReduce, "ncclAllReduce")
  CDKF_SYM(CommDestroy, "ncclCommDestroy")
  CDKF_SYM(GroupStart, "ncclGroupStart")
  CDKF_SYM(GroupEnd, "ncclGroupEnd")
  CDKF_SYM(GetErrorString, "ncclGetErrorString")
#undef CDKF_SYM
  g_rccl = r;
  return &g_rccl;
}

#define CDKF_NCCL_CHECK(R, expr)                                                                         \
  do {                                                                                                   \
    const ncclResult_t res__ = (expr);                                                                   \
    if (res__ != ncclSuccess) {                                                                          \
      set_error("%s failed: %s (%s:%d)", #expr, (R)->GetErrorString(res__), __FILE__, __LINE__);         \
      return CDKF_EHIP;                                                                                  \
    }                                                                                                    \
  } while (0)

}  // namespace

struct cdkf_comm {
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1, device = 0;
};

static_assert(sizeof(ncclUniqueId) == CDKF_COMM_ID_BYTES, "include/cdkf.h: CDKF_COMM_ID_BYTES");

// What a rank can find out about its own chances of joining a communicator WITHOUT entering a collective: the library and its symbols
// (dlopen), the device index against the devices this process sees.  ncclCommInitRank's bootstrap has no timeout, so a rank that fails
// here while its peers are already inside it strands the job: every rank takes this step first, the ranks agree on the outcome over
// the rendezvous (a max of failure flags), and only a unanimous pass goes on to ncclCommInitRank (cd_dynamax_amd/distributed.py).
extern "C" int cdkf_comm_preflight(int device) {
  if (device < 0) {
    set_error("cdkf_comm_preflight: bad device %d", device);
    return CDKF_EINVAL;
  }
  if (!rccl()) return CDKF_EUNSUPPORTED;
  int n = 0;
  const hipError_t e = hipGetDeviceCount(&n);
  if (e != hipSuccess) {
    set_error("cdkf_comm_preflight: hipGetDeviceCount failed: %s", hipGetErrorString(e));
    return CDKF_EHIP;
  }
  if (device >= n) {
    set_error("cdkf_comm_preflight: device %d of %d visible device(s)", device, n);
    return CDKF_EHIP;
  }
  return CDKF_OK;
}

extern "C" int cdkf_comm_unique_id(void* id) {
  if (!id) {
    set_error("cdkf_comm_unique_id: NULL id");
    return CDKF_EINVAL;
  }
  const Rccl* R = rccl();
  if (!R) return CDKF_EUNSUPPORTED;
  ncclUniqueId u;
  CDKF_NCCL_CHECK(R, R->GetUniqueId(&u));
  std::memcpy(id, &u, sizeof(u));
  return CDKF_OK;
}

extern "C" int cdkf_comm_init_rank(cdkf_comm** out, const void* id, int rank, int world, int device) {
  if (!out || !id || world < 1 || rank < 0 || rank >= world || device < 0) {
    set_error("cdkf_comm_init_rank: bad arguments (rank %d of %d, device %d)", rank, world, device);
    return CDKF_EINVAL;
  }
  const Rccl* R = rccl();
  if (!R) return CDKF_EUNSUPPORTED;
  cdkf::DeviceGuard guard(device);
  if (!guard.ok()) return CDKF_EHIP;
  ncclUniqueId u;
  std::memcpy(&u, id, sizeof(u));
  cdkf_comm* c = new cdkf_comm;
  c->rank = rank;
  c->world = world;
  c->device = device;
  const ncclResult_t res = R->CommInitRank(&c->comm, world, u, rank);
  if (res != ncclSuccess) {
    set_error("ncclCommInitRank(rank %d of %d, device %d) failed: %s", rank, world, device, R->GetErrorString(res));
    delete c;
    return CDKF_EHIP;
  }
  *out = c;
  return CDKF_OK;
}

extern "C" int cdkf_comm_init_all(cdkf_comm** out, int ndev, const int* devices) {
  if (!out || ndev < 1) {
    set_error("cdkf_comm_init_all: bad arguments");
    return CDKF_EINVAL;
  }
  const Rccl* R = rccl();
  if (!R) return CDKF_EUNSUPPORTED;
  std::vector<int> devs(ndev);
  for (int i = 0; i < ndev; ++i) devs[i] = devices ? devices[i] : i;
  std::vector<ncclComm_t> comms(ndev);
  CDKF_NCCL_CHECK(R, R->CommInitAll(comms.data(), ndev, devs.data()));
  for (int i = 0; i < ndev; ++i) {
    out[i] = new cdkf_comm;
    out[i]->comm = comms[i];
    out[i]->rank = i;
    out[i]->world = ndev;
    out[i]->device = devs[i];
  }
  return CDKF_OK;
}

extern "C" int cdkf_comm_rank(const cdkf_comm* c) { return c ? c->rank : CDKF_EINVAL; }
extern "C" int cdkf_comm_world(const cdkf_comm* c) { return c ? c->world : CDKF_EINVAL; }

static int allreduce_f64(const cdkf_comm* c, double* buf, int64_t count, ncclRedOp_t op, void* stream) {
  if (!c || !buf || count < 0) {
    set_error("cdkf allreduce: bad arguments");
    return CDKF_EINVAL;
  }
  if (count == 0) return CDKF_OK;
  const Rccl* R = rccl();
  if (!R) return CDKF_EUNSUPPORTED;
  cdkf::DeviceGuard guard(c->device);
  if (!guard.ok()) return CDKF_EHIP;
  CDKF_NCCL_CHECK(R, R->AllReduce(buf, buf, (size_t)count, ncclFloat64, op, c->comm, static_cast<hipStream_t>(stream)));
  return CDKF_OK;
}

extern "C" int cdkf_ll_allreduce(const cdkf_comm* c, double* sums, int64_t count, void* stream) {
  return allreduce_f64(c, sums, count, ncclSum, stream);
}
extern "C" int cdkf_comm_allreduce_max(const cdkf_comm* c, double* values, int64_t count, void* stream) {
  return allreduce_f64(c, values, count, ncclMax, stream);
}

extern "C" int cdkf_ll_allreduce_all(cdkf_comm* const* comms, int ndev, double* const* sums, int64_t count, void* const* streams) {
  if (!comms || !sums || ndev < 1 || count < 0) {
    set_error("cdkf_ll_allreduce_all: bad arguments");
    return CDKF_EINVAL;
  }
  const Rccl* R = rccl();
  if (!R) return CDKF_EUNSUPPORTED;
  CDKF_NCCL_CHECK(R, R->GroupStart());
  for (int i = 0; i < ndev; ++i) {
    const ncclResult_t res = R->AllReduce(sums[i], sums[i], (size_t)count, ncclFloat64, ncclSum, comms[i]->comm,
                                          static_cast<hipStream_t>(streams ? streams[i] : nullptr));
    if (res != ncclSuccess) {
      R->GroupEnd();
      set_error("ncclAllReduce (device %d) failed: %s", comms[i]->device, R->GetErrorString(res));
      return CDKF_EHIP;
    }
  }
  CDKF_NCCL_CHECK(R, R->GroupEnd());
  return CDKF_OK;
}

extern "C" int cdkf_comm_destroy(cdkf_comm* c) {
  if (!c) return CDKF_OK;
  const Rccl* R = rccl();
  if (R && c->comm) R->CommDestroy(c->comm);
  delete c;
  return CDKF_OK;
}

// ---------------------------------------------------------------- TCP rendezvous (host only) --------------------------------
struct cdkf_rdv {
  int rank = 0, world = 1;
  int listen_fd = -1;
  std::vector<int> peers;  // rank 0: socket of rank r at [r] (own slot -1); other ranks: [0] = socket to rank 0
};

namespace {

bool send_all(int fd, const void* buf, size_t n) {
  const char* p = static_cast<const char*>(buf);
  while (n) {
    const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += k;
    n -= (size_t)k;
  }
  return true;
}

bool recv_all(int fd, void* buf, size_t n, int timeout_ms) {
  char* p = static_cast<char*>(buf);
  while (n) {
    pollfd pf{fd, POLLIN, 0};
    const int pr = ::poll(&pf, 1, timeout_ms);
    if (pr == 0) {
      errno = ETIMEDOUT;
      return false;
    }
    if (pr < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    const ssize_t k = ::recv(fd, p, n, 0);
    if (k == 0) {
      errno = ECONNRESET;
      return false;
    }
    if (k < 0) {
      if (errno == EINTR) continue;
      return false;
    }
    p += k;
    n -= (size_t)k;
  }
  return true;
}

void close_all(cdkf_rdv* r) {
  for (int fd : r->peers)
    if (fd >= 0) ::close(fd);
  if (r->listen_fd >= 0) ::close(r->listen_fd);
  r->peers.clear();
  r->listen_fd = -1;
}

constexpr int kIoTimeoutMs = 600000;  // a peer that says nothing for ten minutes is gone

// The hello a rank sends carries a per-job nonce: CDKF_RDV_NONCE from the environment (any 63-bit number the launcher hands to every
// rank), else one derived from the rendezvous port and the world size -- so that a stray connection, or a rank of ANOTHER job whose
// store happens to sit on the same port, is turned away with a NACK instead of taking a slot (and the run's sums with it).
// The default also folds in the job identities a launcher exports IDENTICALLY to every rank (TORCHELASTIC_RUN_ID, SLURM_JOB_ID -- FNV-1a
// over their text), so two jobs on the default port and the same world size do not share a nonce when the launcher tells them apart.
// (Round 4 hashed MASTER_ADDR too: ranks that reach one master under different spellings -- 'localhost' on rank 0, an IP elsewhere --
//  then computed different nonces and were turned away; ADVICE r4.  An identity only SOME ranks inherit still splits the job: the NACK
//  names both nonces, and CDKF_RDV_NONCE is the override -- INTEGRATION.md.)
int64_t rdv_nonce(int port, int world) {
  if (const char* e = std::getenv("CDKF_RDV_NONCE")) return (int64_t)std::strtoll(e, nullptr, 0);
  uint64_t h = 1469598103934665603ull;
  for (const char* name : {"TORCHELASTIC_RUN_ID", "SLURM_JOB_ID"})
    if (const char* v = std::getenv(name)) {
      for (const char* c = v; *c; ++c) h = (h ^ (unsigned char)*c) * 1099511628211ull;
      h = (h ^ 0xffu) * 1099511628211ull;
    }
  return (int64_t)((h >> 1) ^ ((uint64_t)0x63646b66 * 1000003 + (uint64_t)port * 4099 + (uint64_t)world));
}
struct RdvHello {
  int32_t magic, rank, world, pad;
  int64_t nonce;
};
constexpr int32_t kRdvMagic = 0x43444b52;  // "CDKR"

}  // namespace

extern "C" int cdkf_rdv_create(cdkf_rdv** out, const char* addr, int port, int rank, int world, int timeout_ms) {
  if (!out || !addr || port <= 0 || port > 65535 || world < 1 || rank < 0 || rank >= world) {
    set_error("cdkf_rdv_create: bad arguments (rank %d of %d, %s:%d)", rank, world, addr ? addr : "(null)", port);
    return CDKF_EINVAL;
  }
  if (timeout_ms <= 0) timeout_ms = 120000;
  cdkf_rdv* r = new cdkf_rdv;
  r->rank = rank;
  r->world = world;
  if (world == 1) {
    *out = r;
    return CDKF_OK;
  }
  addrinfo hints{}, *res = nullptr;
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  char portstr[16];
  std::snprintf(portstr, sizeof(portstr), "%d", port);
  if (getaddrinfo(addr, portstr, &hints, &res) != 0 || !res) {
    set_error("cdkf_rdv_create: cannot resolve %s", addr);
    delete r;
    return CDKF_EINVAL;
  }
  const auto deadline = std::chrono::steady_clock::now() + std::chrono::milliseconds(timeout_ms);
  const int one = 1;
  const int64_t nonce = rdv_nonce(port, world);
  if (rank == 0) {
    r->listen_fd = ::socket(AF_INET, SOCK_STREAM, 0);
    if (r->listen_fd >= 0) ::setsockopt(r->listen_fd, SOL_SOCKET, SO_REUSEADDR, &one, sizeof(one));
    if (r->listen_fd < 0 || ::bind(r->listen_fd, res->ai_addr, res->ai_addrlen) != 0 || ::listen(r->listen_fd, world) != 0) {
      set_error("cdkf_rdv_create: rank 0 cannot listen on %s:%d: %s", addr, port, std::strerror(errno));
      freeaddrinfo(res);
      close_all(r);
      delete r;
      return CDKF_EHIP;
    }
    r->peers.assign(world, -1);
    for (int got = 1; got < world;) {
      const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
      pollfd pf{r->listen_fd, POLLIN, 0};
      if (left <= 0 || ::poll(&pf, 1, (int)left) <= 0) {
        set_error("cdkf_rdv_create: %d of %d ranks arrived within %d ms", got, world, timeout_ms);
        freeaddrinfo(res);
        close_all(r);
        delete r;
        return CDKF_EHIP;
      }
      const int fd = ::accept(r->listen_fd, nullptr, nullptr);
      if (fd < 0) continue;
      ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
      // the hello must arrive within what is left of the deadline, two seconds at most: an idle stray connection cannot stall the
      // accept loop for long; whoever it is gets an answer -- ACK (1) or NACK (0) -- so a refused rank fails at once, with a message
      RdvHello hello{};
      const int hello_ms = (int)(left < 2000 ? left : 2000);
      const bool ok = recv_all(fd, &hello, sizeof(hello), hello_ms) && hello.magic == kRdvMagic && hello.nonce == nonce &&
                      hello.world == world && hello.rank > 0 && hello.rank < world && r->peers[hello.rank] < 0;
      const int32_t answer = ok ? 1 : 0;
      (void)send_all(fd, &answer, sizeof(answer));
      if (!ok) {
        ::close(fd);  // not one of ours, another job's, or a duplicate rank
        continue;
      }
      r->peers[hello.rank] = fd;
      ++got;
    }
  } else {
    int fd = -1;
    for (;;) {
      fd = ::socket(AF_INET, SOCK_STREAM, 0);
      if (fd >= 0 && ::connect(fd, res->ai_addr, res->ai_addrlen) == 0) break;
      if (fd >= 0) ::close(fd);
      fd = -1;
      if (std::chrono::steady_clock::now() > deadline) break;
      std::this_thread::sleep_for(std::chrono::milliseconds(50));
    }
    if (fd < 0) {
      set_error("cdkf_rdv_create: rank %d cannot reach rank 0 at %s:%d within %d ms", rank, addr, port, timeout_ms);
      freeaddrinfo(res);
      delete r;
      return CDKF_EHIP;
    }
    ::setsockopt(fd, IPPROTO_TCP, TCP_NODELAY, &one, sizeof(one));
    const RdvHello hello{kRdvMagic, rank, world, 0, nonce};
    int32_t answer = -1;
    const auto left = std::chrono::duration_cast<std::chrono::milliseconds>(deadline - std::chrono::steady_clock::now()).count();
    if (!send_all(fd, &hello, sizeof(hello)) || !recv_all(fd, &answer, sizeof(answer), (int)(left > 1000 ? left : 1000)) || answer != 1) {
      if (answer == 0)
        set_error("cdkf_rdv_create: rank 0 at %s:%d refused rank %d of %d (a duplicate rank, another world size, or another job's "
                  "rendezvous; this rank's job nonce is %lld -- it must equal rank 0's: derived from CDKF_RDV_NONCE, else from the port, "
                  "the world size and TORCHELASTIC_RUN_ID / SLURM_JOB_ID as THIS rank sees them; set CDKF_RDV_NONCE / CDKF_RDV_PORT)",
                  addr, port, rank, world, (long long)nonce);
      else
        set_error("cdkf_rdv_create: rank %d lost rank 0: %s", rank, std::strerror(errno));
      ::close(fd);
      freeaddrinfo(res);
      delete r;
      return CDKF_EHIP;
    }
    r->peers.assign(1, fd);
  }
  freeaddrinfo(res);
  *out = r;
  return CDKF_OK;
}

extern "C" int cdkf_rdv_broadcast(cdkf_rdv* r, void* buf, int64_t bytes) {
  if (!r || (!buf && bytes) || bytes < 0) {
    set_error("cdkf_rdv_broadcast: bad arguments");
    return CDKF_EINVAL;
  }
  if (r->world == 1 || bytes == 0) return CDKF_OK;
  if (r->rank == 0) {
    for (int p = 1; p < r->world; ++p)
      if (!send_all(r->peers[p], buf, (size_t)bytes)) {
        set_error("cdkf_rdv_broadcast: lost rank %d: %s", p, std::strerror(errno));
        return CDKF_EHIP;
      }
  } else if (!recv_all(r->peers[0], buf, (size_t)bytes, kIoTimeoutMs)) {
    set_error("cdkf_rdv_broadcast: rank %d lost rank 0: %s", r->rank, std::strerror(errno));
    return CDKF_EHIP;
  }
  return CDKF_OK;
}

// op: 0 sum, 1 max.  Rank 0 combines in rank order (the same bits on every run), then hands the result back.
extern "C" int cdkf_rdv_allreduce(cdkf_rdv* r, double* values, int64_t count, int op) {
  if (!r || (!values && count) || count < 0 || (op != 0 && op != 1)) {
    set_error("cdkf_rdv_allreduce: bad arguments");
    return CDKF_EINVAL;
  }
  if (r->world == 1) return CDKF_OK;
  const size_t bytes = (size_t)count * sizeof(double);
  int64_t n = count;
  if (r->rank == 0) {
    std::vector<double> in((size_t)count);
    for (int p = 1; p < r->world; ++p) {
      int64_t pn = -1;
      if (!recv_all(r->peers[p], &pn, sizeof(pn), kIoTimeoutMs) || pn != count || !recv_all(r->peers[p], in.data(), bytes, kIoTimeoutMs)) {
        set_error("cdkf_rdv_allreduce: rank %d sent %lld values where %lld were expected (or went away)", p, (long long)pn,
                  (long long)count);
        return CDKF_EHIP;
      }
      for (int64_t e = 0; e < count; ++e) values[e] = op == 0 ? values[e] + in[e] : (in[e] > values[e] ? in[e] : values[e]);
    }
    for (int p = 1; p < r->world; ++p)
      if (!send_all(r->peers[p], values, bytes)) {
        set_error("cdkf_rdv_allreduce: lost rank %d: %s", p, std::strerror(errno));
        return CDKF_EHIP;
      }
  } else {
    if (!send_all(r->peers[0], &n, sizeof(n)) || !send_all(r->peers[0], values, bytes) ||
        !recv_all(r->peers[0], values, bytes, kIoTimeoutMs)) {
      set_error("cdkf_rdv_allreduce: rank %d lost rank 0: %s", r->rank, std::strerror(errno));
      return CDKF_EHIP;
    }
  }
  return CDKF_OK;
}

extern "C" int cdkf_rdv_barrier(cdkf_rdv* r) {
  double z = 0.0;
  return cdkf_rdv_allreduce(r, &z, 1, 0);
}

extern "C" int cdkf_rdv_destroy(cdkf_rdv* r) {
  if (!r) return CDKF_OK;
  close_all(r);
  delete r;
  return CDKF_OK;
}
