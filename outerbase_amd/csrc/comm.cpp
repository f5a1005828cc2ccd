// Multi-GPU exchange of libobhip: rows are sharded over ranks, one
// process per GPU, and the only data that ever crosses ranks is
//   back end A: ONE buffer per fit -- the packed upper triangle of G_r = B_r^T B_r, the two
//               p-vectors B_r^T y_r and B_r^T 1 and the three scalars (sum y, sum y^2, n_r);
//   back end B: one p-vector (+ 2 scalars) per B^T a pass of the PCG.
// The reference is a single process (nthreads = omp_get_num_procs(), src/modandbase.cpp:464);
// this is north_star's own requirement and has no reference counterpart.
//
// Transports behind one obhip_comm:
//   RCCL  (obhip_comm_init): ncclReduceScatter + ncclAllGather, in place, on the library's
//         stream.  On a fully connected xGMI node the two halves drive all links at once; the
//         ring a single all-reduce may pick is bound by one link.  librccl is loaded at run
//         time (the copy already in the process wins), so the library neither links nor
//         needs it on one GPU.
//   host  (obhip_comm_init_host): the caller supplies "sum this HOST buffer over ranks"
//         (gloo in the one-GPU rehearsals of the tests, MPI_Allreduce under R); the library
//         stages through a pinned buffer.
//   sim   (obhip_comm_init_sim): N virtual ranks that all hold THIS rank's shard; a sum is one
//         device pass buf *= N.  No wire: it exists to time, on one GPU, the step a rank of an
//         N-GPU job runs (pack, unpack, form H, replicated solve), and its result is checkable
//         (the fit of the shard's rows repeated N times).
//
// Which collective an RCCL communicator uses is fixed when it is created and is the same on all
// ranks: OBHIP_RCCL_ALLREDUCE=1 on ANY rank (the flags are summed over the ranks at init) turns
// the reduce-scatter / all-gather pair off everywhere, and obhip_comm_selftest_dev does the same
// when the pair returns wrong sums on this node.  Ranks can therefore never issue different
// collectives on one communicator.
#include <dlfcn.h>

#include <cmath>
#include <cstdlib>
#include <cstring>

#include "obhip_internal.h"
#include "vec_ops.h"

using namespace obhip;

namespace {

// the few RCCL entry points, by their documented C signatures (rccl/rccl.h)
struct RcclId {
  char internal[OBHIP_UNIQUE_ID_BYTES];
};
typedef void *rccl_comm_t;
constexpr int kNcclSum = 0, kNcclDouble = 8;  // ncclRedOp_t / ncclDataType_t
typedef int (*nccl_get_unique_id_t)(RcclId *);
typedef int (*nccl_comm_init_rank_t)(rccl_comm_t *, int, RcclId, int);
typedef int (*nccl_comm_destroy_t)(rccl_comm_t);
typedef int (*nccl_comm_count_t)(const rccl_comm_t, int *);
typedef const char *(*nccl_get_error_string_t)(int);
typedef int (*nccl_get_version_t)(int *);
typedef int (*nccl_reduce_scatter_t)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t);
typedef int (*nccl_all_gather_t)(const void *, void *, size_t, int, rccl_comm_t, hipStream_t);
typedef int (*nccl_all_reduce_t)(const void *, void *, size_t, int, int, rccl_comm_t, hipStream_t);
typedef int (*nccl_group_t)(void);

struct Rccl {
  void *lib = nullptr;
  nccl_get_unique_id_t get_unique_id = nullptr;
  nccl_comm_init_rank_t comm_init_rank = nullptr;
  nccl_comm_destroy_t comm_destroy = nullptr;
  nccl_comm_count_t comm_count = nullptr;
  nccl_get_error_string_t error_string = nullptr;
  nccl_get_version_t get_version = nullptr;
  nccl_reduce_scatter_t reduce_scatter = nullptr;
  nccl_all_gather_t all_gather = nullptr;
  nccl_all_reduce_t all_reduce = nullptr;
  nccl_group_t group_start = nullptr, group_end = nullptr;
};

int rccl(Rccl **out) {
  static Rccl r;
  if (!r.lib) {
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char *nm : names)
      if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!r.lib)
      for (const char *nm : names)
        if ((r.lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL))) break;
    if (!r.lib) return fail(OBHIP_ERR_STATE, "RCCL (librccl.so.1) not found: multi-GPU needs it");
    r.get_unique_id = (nccl_get_unique_id_t)dlsym(r.lib, "ncclGetUniqueId");
    r.comm_init_rank = (nccl_comm_init_rank_t)dlsym(r.lib, "ncclCommInitRank");
    r.comm_destroy = (nccl_comm_destroy_t)dlsym(r.lib, "ncclCommDestroy");
    r.comm_count = (nccl_comm_count_t)dlsym(r.lib, "ncclCommCount");
    r.error_string = (nccl_get_error_string_t)dlsym(r.lib, "ncclGetErrorString");
    r.get_version = (nccl_get_version_t)dlsym(r.lib, "ncclGetVersion");
    r.reduce_scatter = (nccl_reduce_scatter_t)dlsym(r.lib, "ncclReduceScatter");
    r.all_gather = (nccl_all_gather_t)dlsym(r.lib, "ncclAllGather");
    r.all_reduce = (nccl_all_reduce_t)dlsym(r.lib, "ncclAllReduce");
    r.group_start = (nccl_group_t)dlsym(r.lib, "ncclGroupStart");
    r.group_end = (nccl_group_t)dlsym(r.lib, "ncclGroupEnd");
    if (!r.get_unique_id || !r.comm_init_rank || !r.comm_destroy || !r.comm_count ||
        !r.reduce_scatter || !r.all_gather || !r.all_reduce) {
      r.lib = nullptr;
      return fail(OBHIP_ERR_STATE, "librccl lacks an expected entry point");
    }
  }
  *out = &r;
  return 0;
}

int rccl_fail(Rccl *r, int rc, const char *what) {
  return fail(OBHIP_ERR_HIP, std::string("RCCL: ") + what + " failed: " +
                                 (r->error_string ? r->error_string(rc) : std::to_string(rc)));
}

}  // namespace

struct obhip_comm {
  int nranks = 1, rank = 0;
  int transport = OBHIP_TRANSPORT_NONE;
  int device = 0;
  // RCCL
  Rccl *r = nullptr;
  rccl_comm_t nccl = nullptr;
  int rccl_ranks = 0;
  // host transport
  obhip_host_allreduce_fn fn = nullptr;
  void *user = nullptr;
  double *pinned = nullptr;
  size_t pinned_n = 0;
  // statistics of the last exchange
  uint64_t last_bytes = 0, calls = 0;
  // RCCL: plain ncclAllReduce for every size (agreed over the ranks at init, or set by the
  // self-test); selftest: 0 not run, 1 passed, 2 passed after switching the pair off
  bool plain = false;
  int selftest = 0;
};

namespace {

// reduce-scatter + all-gather need nranks equal blocks (16-byte multiples); small or ragged
// buffers take the plain all-reduce (latency-bound anyway)
bool pair_fits(const obhip_comm *c, uint64_t count) {
  const uint64_t nr = (uint64_t)c->nranks;
  return count >= 4096 * nr && count % (2 * nr) == 0;
}

#ifdef OBHIP_TESTING
bool g_fault_inject_pair = false;
#endif

int rccl_pair(obhip_comm *c, double *d_buf, uint64_t count, hipStream_t st) {
  Rccl *r = c->r;
  const uint64_t blk = count / (uint64_t)c->nranks;
  double *mine = d_buf + (uint64_t)c->rank * blk;
  int rc = r->reduce_scatter(d_buf, mine, blk, kNcclDouble, kNcclSum, c->nccl, st);
  if (rc) return rccl_fail(r, rc, "ncclReduceScatter");
  rc = r->all_gather(mine, d_buf, blk, kNcclDouble, c->nccl, st);
  if (rc) return rccl_fail(r, rc, "ncclAllGather");
#ifdef OBHIP_TESTING
  // fault injection for the tests of the self-test's in-process switch: one wrong element.
  // Compiled into libobhip_testing.so only (csrc/Makefile); the shipping library has neither
  // this branch nor the entry point that arms it.
  if (g_fault_inject_pair) {
    double *q = d_buf + count / 2;
    OB_TRY(vmap(1, [=] __device__(uint64_t) { *q += 1.0; }));
  }
#endif
  return 0;
}

int rccl_plain(obhip_comm *c, double *d_buf, uint64_t count, hipStream_t st) {
  const int rc = c->r->all_reduce(d_buf, d_buf, count, kNcclDouble, kNcclSum, c->nccl, st);
  if (rc) return rccl_fail(c->r, rc, "ncclAllReduce");
  return 0;
}

}  // namespace

namespace obhip {

// in-place sum of count doubles over the ranks of c, enqueued on (RCCL) or synchronised
// with (host transport) the library's stream
int comm_allreduce(obhip_comm *c, double *d_buf, uint64_t count) {
  if (!c || count == 0) return 0;
  // (a one-rank communicator still goes through its transport: the sums are copies, and
  // the whole exchange can be rehearsed on one GPU)
  hipStream_t st = cur_stream();
  c->last_bytes = count * sizeof(double);
  c->calls += 1;
  if (c->transport == OBHIP_TRANSPORT_RCCL)
    return !c->plain && pair_fits(c, count) ? rccl_pair(c, d_buf, count, st) : rccl_plain(c, d_buf, count, st);
  if (c->transport == OBHIP_TRANSPORT_SIM) {
    const double nr = (double)c->nranks;
    return vmap(count, [=] __device__(uint64_t i) { d_buf[i] *= nr; });
  }
  if (c->transport == OBHIP_TRANSPORT_HOST) {
    // one rank without a callback (obhip_comm_init_host accepts that): the sum is the identity
    if (!c->fn) return c->nranks == 1 ? 0 : fail(OBHIP_ERR_STATE, "host communicator has no callback");
    if (c->pinned_n < count) {
      if (c->pinned) (void)hipHostFree(c->pinned);
      c->pinned = nullptr;
      c->pinned_n = 0;
      OB_HIP(hipHostMalloc((void **)&c->pinned, count * sizeof(double), hipHostMallocDefault));
      c->pinned_n = count;
    }
    OB_HIP(hipMemcpyAsync(c->pinned, d_buf, count * sizeof(double), hipMemcpyDeviceToHost, st));
    OB_HIP(hipStreamSynchronize(st));
    if (c->fn(c->user, c->pinned, count) != 0)
      return fail(OBHIP_ERR_HIP, "host all-reduce callback failed");
    OB_HIP(hipMemcpyAsync(d_buf, c->pinned, count * sizeof(double), hipMemcpyHostToDevice, st));
    return 0;
  }
  return fail(OBHIP_ERR_STATE, "communicator has no transport");
}

int comm_nranks(const obhip_comm *c) { return c ? c->nranks : 1; }

}  // namespace obhip

extern "C" {

int obhip_comm_unique_id(void *id) {
  if (!id) return fail(OBHIP_ERR_INVALID, "comm_unique_id: null argument");
  Rccl *r = nullptr;
  OB_TRY(rccl(&r));
  RcclId u;
  std::memset(&u, 0, sizeof u);
  const int rc = r->get_unique_id(&u);
  if (rc) return rccl_fail(r, rc, "ncclGetUniqueId");
  std::memcpy(id, &u, sizeof u);
  return 0;
}

int obhip_comm_init(obhip_comm **out, int nranks, int rank, const void *id) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || !id)
    return fail(OBHIP_ERR_INVALID, "comm_init: bad argument");
  OB_TRY(require_device());
  Rccl *r = nullptr;
  OB_TRY(rccl(&r));
  obhip_comm *c = new obhip_comm();
  c->nranks = nranks;
  c->rank = rank;
  c->transport = OBHIP_TRANSPORT_RCCL;
  c->r = r;
  (void)hipGetDevice(&c->device);
  RcclId u;
  std::memcpy(&u, id, sizeof u);
  int rc = r->comm_init_rank(&c->nccl, nranks, u, rank);
  if (rc) {
    delete c;
    return rccl_fail(r, rc, "ncclCommInitRank");
  }
  rc = r->comm_count(c->nccl, &c->rccl_ranks);
  if (rc || c->rccl_ranks != nranks) {
    (void)r->comm_destroy(c->nccl);
    delete c;
    return fail(OBHIP_ERR_STATE, "RCCL communicator reports a different rank count");
  }
  // OBHIP_RCCL_ALLREDUCE=1 (a switch for a node where the in-place reduce-scatter / all-gather
  // pair misbehaves): decided here once, and agreed -- the flags of all ranks are summed, so a
  // launcher that forwards the variable to some ranks only cannot make them issue different
  // collectives
  {
    const char *e = getenv("OBHIP_RCCL_ALLREDUCE");
    double flag = e && atoi(e) != 0 ? 1.0 : 0.0;
    DevBuf<double> f;
    int rc2 = f.upload(&flag, 1);
    hipStream_t st = cur_stream();
    if (!rc2) rc2 = rccl_plain(c, f.p, 1, st);
    if (!rc2 && (hipMemcpyAsync(&flag, f.p, sizeof flag, hipMemcpyDeviceToHost, st) != hipSuccess ||
                 hipStreamSynchronize(st) != hipSuccess))
      rc2 = fail(OBHIP_ERR_HIP, "comm_init: reading back the agreed exchange path failed");
    if (rc2) {
      (void)r->comm_destroy(c->nccl);
      delete c;
      return rc2;
    }
    c->plain = flag > 0.0;
  }
  *out = c;
  return 0;
}

int obhip_comm_init_sim(obhip_comm **out, int nranks) {
  if (!out || nranks < 1) return fail(OBHIP_ERR_INVALID, "comm_init_sim: bad argument");
  OB_TRY(require_device());
  obhip_comm *c = new obhip_comm();
  c->nranks = nranks;
  c->rank = 0;
  c->transport = OBHIP_TRANSPORT_SIM;
  (void)hipGetDevice(&c->device);
  *out = c;
  return 0;
}

int obhip_comm_exchange_path(const obhip_comm *c, uint64_t count, int *path, int *selftest) {
  if (!c) return fail(OBHIP_ERR_INVALID, "comm_exchange_path: null communicator");
  if (path) {
    if (c->transport == OBHIP_TRANSPORT_RCCL)
      *path = !c->plain && pair_fits(c, count) ? OBHIP_EXCHANGE_PAIR : OBHIP_EXCHANGE_ALLREDUCE;
    else
      *path = c->transport == OBHIP_TRANSPORT_HOST ? OBHIP_EXCHANGE_HOST
                                                   : (c->transport == OBHIP_TRANSPORT_SIM ? OBHIP_EXCHANGE_SIM : OBHIP_EXCHANGE_NONE);
  }
  if (selftest) *selftest = c->selftest;
  return 0;
}

// Every rank fills count doubles with (rank + 1) w_i, w_i = 1 + i mod 1021 (small integers: every
// partial sum is exact in any order), sums them over the ranks and compares ON THE DEVICE with
// the closed form nranks (nranks + 1) / 2 w_i -- once through the reduce-scatter / all-gather
// pair (when a buffer of this size would take it) and once through the plain all-reduce.  The
// mismatch counts are summed over the ranks, so every rank takes the same decision.
int obhip_comm_selftest_dev(obhip_comm *c, uint64_t count, int64_t *result) {
  if (!c || count == 0) return fail(OBHIP_ERR_INVALID, "comm_selftest_dev: bad argument");
  OB_TRY(require_device());
  hipStream_t st = cur_stream();
  DevBuf<double> buf, bad, red;
  OB_TRY(buf.alloc(count));
  OB_TRY(bad.alloc(2));
  OB_TRY(red.alloc((size_t)kSumBlocks));
  OB_HIP(hipMemsetAsync(bad.p, 0, 2 * sizeof(double), st));
  double *b = buf.p;
  const double mine = (double)(c->rank + 1);
  const double nr = (double)c->nranks;
  const double tot = c->transport == OBHIP_TRANSPORT_SIM ? nr * mine : nr * (nr + 1.0) / 2.0;
  auto fill = [&]() { return vmap(count, [=] __device__(uint64_t i) { b[i] = mine * (double)(1 + i % 1021); }); };
  auto check = [&](double *dst) {
    return vsum<1>(count, [=] __device__(uint64_t i, double (&acc)[1]) {
      acc[0] += b[i] == tot * (double)(1 + i % 1021) ? 0.0 : 1.0;
    }, dst, red.p);
  };
  const bool rccl_t = c->transport == OBHIP_TRANSPORT_RCCL;
  const bool try_pair = rccl_t && !c->plain && pair_fits(c, count);
  if (try_pair) {
    OB_TRY(fill());
    OB_TRY(rccl_pair(c, b, count, st));
    OB_TRY(check(bad.p));
  }
  OB_TRY(fill());
  if (rccl_t) OB_TRY(rccl_plain(c, b, count, st));
  else OB_TRY(comm_allreduce(c, b, count));
  OB_TRY(check(bad.p + 1));
  // the verdicts of all ranks (through the plain path: were that one broken, the rank that saw
  // it has a non-zero count of its own and fails below whatever the sum says)
  double own[2], all[2];
  OB_HIP(hipMemcpyAsync(own, bad.p, sizeof own, hipMemcpyDeviceToHost, st));
  OB_HIP(hipStreamSynchronize(st));
  if (rccl_t) OB_TRY(rccl_plain(c, bad.p, 2, st));
  else OB_TRY(comm_allreduce(c, bad.p, 2));
  OB_HIP(hipMemcpyAsync(all, bad.p, sizeof all, hipMemcpyDeviceToHost, st));
  OB_HIP(hipStreamSynchronize(st));
  const bool pair_bad = try_pair && (own[0] != 0.0 || all[0] != 0.0);
  const bool plain_bad = own[1] != 0.0 || all[1] != 0.0;
  if (result) {
    result[0] = 0;
    result[1] = try_pair ? (int64_t)std::max(own[0], all[0]) : -1;
    result[2] = (int64_t)std::max(own[1], all[1]);
    result[3] = pair_bad ? 1 : 0;
  }
  if (plain_bad)
    return fail(OBHIP_ERR_STATE, "comm_selftest: the all-reduce of this communicator returns wrong sums (" +
                                     std::to_string((long long)std::max(own[1], all[1])) + " of " +
                                     std::to_string((unsigned long long)count) + " elements over all ranks)" +
                                     (pair_bad ? "; so does the reduce-scatter / all-gather pair" : ""));
  if (pair_bad) c->plain = true;  // in-process switch, same decision on every rank
  c->selftest = pair_bad ? 2 : 1;
  if (result) {
    int path = 0;
    (void)obhip_comm_exchange_path(c, count, &path, nullptr);
    result[0] = path;
  }
  return 0;
}

int obhip_comm_init_host(obhip_comm **out, int nranks, int rank, obhip_host_allreduce_fn fn,
                         void *user) {
  if (!out || nranks < 1 || rank < 0 || rank >= nranks || (nranks > 1 && !fn))
    return fail(OBHIP_ERR_INVALID, "comm_init_host: bad argument");
  obhip_comm *c = new obhip_comm();
  c->nranks = nranks;
  c->rank = rank;
  c->transport = OBHIP_TRANSPORT_HOST;
  c->fn = fn;
  c->user = user;
  *out = c;
  return 0;
}

int obhip_comm_destroy(obhip_comm *c) {
  if (!c) return 0;
  if (c->transport == OBHIP_TRANSPORT_RCCL && c->nccl) {
    (void)hipStreamSynchronize(cur_stream());
    (void)c->r->comm_destroy(c->nccl);
  }
  if (c->pinned) (void)hipHostFree(c->pinned);
  delete c;
  return 0;
}

int obhip_comm_info(const obhip_comm *c, int *nranks, int *rank, int *transport, int *rccl_ranks,
                    int *rccl_version) {
  if (!c) return fail(OBHIP_ERR_INVALID, "comm_info: null communicator");
  if (nranks) *nranks = c->nranks;
  if (rank) *rank = c->rank;
  if (transport) *transport = c->transport;
  if (rccl_ranks) *rccl_ranks = c->rccl_ranks;
  if (rccl_version) {
    *rccl_version = 0;
    if (c->r && c->r->get_version) (void)c->r->get_version(rccl_version);
  }
  return 0;
}

int obhip_comm_allreduce_dev(obhip_comm *c, double *d_buf, uint64_t count) {
  if (!c || (!d_buf && count)) return fail(OBHIP_ERR_INVALID, "comm_allreduce_dev: null argument");
  ProfScope ps("exchange");
  return comm_allreduce(c, d_buf, count);
}

}  // extern "C"

// ---- the one-buffer exchange of back end A -------------------------------------------------
namespace obhip {
uint64_t normal_eq_tail(uint64_t p);
int launch_pack_normal_eq(uint64_t p, bool with_tri, const double *d_G, const double *d_g,
                          const double *d_b1, const double *d_sum2, double nlocal, double *d_buf,
                          uint64_t count);
int launch_unpack_normal_eq(uint64_t p, bool with_tri, const double *d_buf, double *d_G, double *d_g,
                            double *d_meansd);
}  // namespace obhip

extern "C" {

int obhip_normal_eq_count(uint64_t p, int nranks, uint64_t *count) {
  if (!count || nranks < 1 || p == 0) return fail(OBHIP_ERR_INVALID, "normal_eq_count: bad argument");
  const uint64_t raw = p * (p + 1) / 2 + normal_eq_tail(p);
  const uint64_t q = 2 * (uint64_t)nranks;  // equal 16-byte blocks for reduce-scatter
  *count = (raw + q - 1) / q * q;
  return 0;
}

int obhip_normal_eq_exchange_dev(obhip_comm *comm, uint64_t p, uint64_t n_local, double *d_G,
                                 double *d_g, const double *d_b1, const double *d_sum2,
                                 double *d_buf, uint64_t buf_count, double *d_meansd) {
  if (!d_G || !d_g || !d_b1 || !d_sum2 || !d_buf || !d_meansd || p == 0)
    return fail(OBHIP_ERR_INVALID, "normal_eq_exchange_dev: null argument");
  const int nr = comm_nranks(comm);
  uint64_t need = 0;
  OB_TRY(obhip_normal_eq_count(p, nr, &need));
  if (buf_count < need) return fail(OBHIP_ERR_INVALID, "normal_eq_exchange_dev: buffer too small");
  const bool many = comm != nullptr;
  // no communicator: G stays where it is, only the tail (g, b1, scalars) goes through the
  // buffer, whose triangle part is never touched
  OB_TRY(launch_pack_normal_eq(p, many, d_G, d_g, d_b1, d_sum2, (double)n_local, d_buf, need));
  if (many) {
    ProfScope ps("exchange");
    OB_TRY(comm_allreduce(comm, d_buf, need));
  }
  return launch_unpack_normal_eq(p, many, d_buf, d_G, d_g, d_meansd);
}

}  // extern "C"

// ---- exact sample quantiles of row-sharded columns -----------------------------------------
// obfit places its knots at quantiles of every input column (.genknotlist, R/fitting.R:177-185:
// quantile(x, probs), R's default type 7).  With the rows sharded over ranks no rank can sort
// the column; the order statistics are found by bisection on the (order-preserving) bit
// pattern instead: 64 rounds of "how many elements, over all ranks, are <= this midpoint",
// for all columns and all wanted order statistics at once.  Exact: the result is the element
// a sort of the whole column would put at that position, and the interpolation is R's.
namespace obhip {
int quantile_max_targets();
int launch_count_le(const double *d_x, uint64_t n, uint64_t d, const uint64_t *d_mids, int T,
                    unsigned long long *d_counts);
int launch_u64_to_f64(const unsigned long long *d_in, uint64_t n, double *d_out);
}  // namespace obhip

namespace {

double key_to_double(uint64_t k) {
  const uint64_t b = (k >> 63) ? (k & 0x7fffffffffffffffull) : ~k;
  double v;
  std::memcpy(&v, &b, sizeof v);
  return v;
}

}  // namespace

extern "C" int obhip_quantiles_dev(obhip_comm *comm, const double *d_x, uint64_t n, uint64_t d,
                                   const double *probs, uint64_t q, double *out) {
  if (!d_x || !probs || !out || d == 0 || q == 0) return fail(OBHIP_ERR_INVALID, "quantiles_dev: bad argument");
  OB_TRY(require_device());
  const int T = (int)(2 * q);
  if (T > quantile_max_targets()) return fail(OBHIP_ERR_INVALID, "quantiles_dev: too many quantiles");
  hipStream_t st = cur_stream();
  // rows of all ranks
  double ntot = (double)n;
  DevBuf<double> dsum;
  OB_TRY(dsum.alloc(d * T));
  if (comm) {
    OB_HIP(hipMemcpyAsync(dsum.p, &ntot, sizeof(double), hipMemcpyHostToDevice, st));
    OB_HIP(hipStreamSynchronize(st));
    OB_TRY(comm_allreduce(comm, dsum.p, 1));
    OB_HIP(hipMemcpyAsync(&ntot, dsum.p, sizeof(double), hipMemcpyDeviceToHost, st));
    OB_HIP(hipStreamSynchronize(st));
  }
  const uint64_t N = (uint64_t)ntot;
  if (N == 0) return fail(OBHIP_ERR_INVALID, "quantiles_dev: no rows");
  // type 7 as stats::quantile.default computes it (1-based): index = 1 + (N - 1) p,
  // lo = floor(index), h = index - lo, Q = (1 - h) x_(lo) + h x_(lo + 1) (the 4 eps fuzz of
  // quantile.default belongs to types 4-6, 8 and 9; type 7 has none); targets 2 j, 2 j + 1
  std::vector<uint64_t> want(T);
  std::vector<double> frac(q);
  for (uint64_t j = 0; j < q; ++j) {
    if (!(probs[j] >= 0.0 && probs[j] <= 1.0)) return fail(OBHIP_ERR_INVALID, "quantiles_dev: probs outside [0, 1]");
    const double index = 1.0 + (double)(N - 1) * probs[j];
    const double lo1 = std::floor(index);
    const uint64_t lo = std::min<uint64_t>((uint64_t)lo1 - 1, N - 1);
    const double h = index - lo1;
    want[2 * j] = lo;
    want[2 * j + 1] = std::min<uint64_t>(lo + 1, N - 1);
    frac[j] = h;
  }
  std::vector<uint64_t> lo(d * T, 0), hi(d * T, ~0ull), mid(d * T);
  std::vector<double> cnt(d * T);
  DevBuf<uint64_t> dmid;
  DevBuf<unsigned long long> dcnt;
  OB_TRY(dmid.alloc(d * T));
  OB_TRY(dcnt.alloc(d * T));
  for (int it = 0; it < 64; ++it) {
    for (uint64_t e = 0; e < d * T; ++e) mid[e] = lo[e] + (hi[e] - lo[e]) / 2;
    OB_HIP(hipMemcpyAsync(dmid.p, mid.data(), d * T * sizeof(uint64_t), hipMemcpyHostToDevice, st));
    OB_HIP(hipMemsetAsync(dcnt.p, 0, d * T * sizeof(unsigned long long), st));
    OB_TRY(launch_count_le(d_x, n, d, dmid.p, T, dcnt.p));
    // the counts travel as doubles (exact below 2^53) so that the one transport sums them
    OB_TRY(launch_u64_to_f64(dcnt.p, d * T, dsum.p));
    if (comm) OB_TRY(comm_allreduce(comm, dsum.p, d * T));
    OB_HIP(hipMemcpyAsync(cnt.data(), dsum.p, d * T * sizeof(double), hipMemcpyDeviceToHost, st));
    OB_HIP(hipStreamSynchronize(st));
    bool open = false;
    for (uint64_t e = 0; e < d * T; ++e) {
      if (lo[e] == hi[e]) continue;
      // smallest key with count(<= key) >= k + 1 is the k-th order statistic (0-based)
      if ((uint64_t)cnt[e] >= want[e % T] + 1) hi[e] = mid[e];
      else lo[e] = mid[e] + 1;
      open = open || lo[e] != hi[e];
    }
    if (!open) break;
  }
  for (uint64_t l = 0; l < d; ++l)
    for (uint64_t j = 0; j < q; ++j) {
      const double a = key_to_double(lo[l * T + 2 * j]), b = key_to_double(lo[l * T + 2 * j + 1]);
      out[l * q + j] = (frac[j] == 0.0 || a == b) ? a : (1.0 - frac[j]) * a + frac[j] * b;
    }
  return 0;
}

#ifdef OBHIP_TESTING
// test build only (libobhip_testing.so): arm / disarm the wrong element behind the
// reduce-scatter + all-gather pair (tests/fault_inject_worker.py)
extern "C" void obhip_testing_fault_inject_pair(int on) { g_fault_inject_pair = on != 0; }
#endif
