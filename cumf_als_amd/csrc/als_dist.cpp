// als_dist.cpp -- the multi-GPU half-iterations behind include/cumf_dist_capi.h: one process per GPU, kernels and RCCL
// collectives enqueued back to back from C++ (compute stream / communication stream, ordered by events).  What it
// replaces in the reference: the OpenMP-thread-per-GPU loop of hugewiki/hugewiki.cu:2436-2745 with its P2P copies
// into a staging buffer, cublasSaxpy accumulation on GPU 0 and three-copy broadcast of Theta.
#include <dlfcn.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <vector>

#include "als_internal.h"
#include "cumf_dist_capi.h"

namespace {

// ---- RCCL, loaded at run time ---------------------------------------------------------------------------------
struct Rccl {
  void* handle = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduceScatter) ReduceScatter = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

const Rccl* rccl() {
  static Rccl r;
  static std::once_flag once;
  std::call_once(once, [] {
    void* h = nullptr;
    if (const char* env = getenv("CUMF_RCCL_LIB")) {
      h = dlopen(env, RTLD_NOW | RTLD_GLOBAL);
      if (!h) fprintf(stderr, "cumf_comm: CUMF_RCCL_LIB=%s: %s\n", env, dlerror());
    } else {
      // the copy already in the process first (torch loads its own librccl.so: two RCCLs in one process are one too many)
      for (const char* name : {"librccl.so", "librccl.so.1"})
        if (!h) h = dlopen(name, RTLD_NOW | RTLD_NOLOAD | RTLD_GLOBAL);
      for (const char* name : {"librccl.so.1", "/opt/rocm/lib/librccl.so.1", "librccl.so"})
        if (!h) h = dlopen(name, RTLD_NOW | RTLD_GLOBAL);
      if (!h) fprintf(stderr, "cumf_comm: librccl.so.1 not found (%s); set CUMF_RCCL_LIB\n", dlerror());
    }
    if (!h) return;
    r.GetUniqueId = reinterpret_cast<decltype(r.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    r.CommInitRank = reinterpret_cast<decltype(r.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    r.CommDestroy = reinterpret_cast<decltype(r.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    r.AllGather = reinterpret_cast<decltype(r.AllGather)>(dlsym(h, "ncclAllGather"));
    r.ReduceScatter = reinterpret_cast<decltype(r.ReduceScatter)>(dlsym(h, "ncclReduceScatter"));
    r.AllReduce = reinterpret_cast<decltype(r.AllReduce)>(dlsym(h, "ncclAllReduce"));
    r.GetErrorString = reinterpret_cast<decltype(r.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    if (r.GetUniqueId && r.CommInitRank && r.CommDestroy && r.AllGather && r.ReduceScatter && r.AllReduce &&
        r.GetErrorString)
      r.handle = h;
    else
      fprintf(stderr, "cumf_comm: the RCCL library lacks an entry point\n");
  });
  return r.handle ? &r : nullptr;
}

#define DIST_NCCL_CHECK(call)                                                                                   \
  do {                                                                                                          \
    ncclResult_t res__ = (call);                                                                                \
    if (res__ != ncclSuccess) {                                                                                 \
      fprintf(stderr, "RCCL Error:\nFile = %s\nLine = %d\nReason = %s\n", __FILE__, __LINE__,                   \
              rccl()->GetErrorString(res__));                                                                   \
      return 1000 + (int)res__;                                                                                 \
    }                                                                                                           \
  } while (0)

#define DIST_CHECK(call)                                                                                        \
  do {                                                                                                          \
    int rc__ = (call);                                                                                          \
    if (rc__ != 0) return rc__;                                                                                 \
  } while (0)

enum { kLocal = 0, kRccl = 1, kCustom = 2 };

}  // namespace

struct cumf_comm {
  int kind = kLocal, rank = 0, world = 1, device = 0;
  ncclComm_t nccl = nullptr;
  cumf_transport_t custom{};
  hipStream_t comm_stream = nullptr;
  hipEvent_t ev_in = nullptr, ev_out = nullptr;  // the short collectives of cumf_comm_all_reduce_f64
};

namespace {

int comm_init_common(cumf_comm* c) {
  CUMF_HIP_CHECK(hipGetDevice(&c->device));
  CUMF_HIP_CHECK(hipStreamCreateWithFlags(&c->comm_stream, hipStreamNonBlocking));
  CUMF_HIP_CHECK(hipEventCreateWithFlags(&c->ev_in, hipEventDisableTiming));
  CUMF_HIP_CHECK(hipEventCreateWithFlags(&c->ev_out, hipEventDisableTiming));
  return 0;
}

// the three collectives on the communication stream of `c`
int all_gather_bytes(cumf_comm* c, const void* send, void* recv, size_t bytes) {
  if (bytes == 0) return 0;
  switch (c->kind) {
    case kRccl:
      DIST_NCCL_CHECK(rccl()->AllGather(send, recv, bytes, ncclChar, c->nccl, c->comm_stream));
      return 0;
    case kCustom:
      return c->custom.all_gather(c->custom.ctx, send, recv, bytes, c->comm_stream);
    default:
      if (send != recv) CUMF_HIP_CHECK(hipMemcpyAsync(recv, send, bytes, hipMemcpyDeviceToDevice, c->comm_stream));
      return 0;
  }
}

int reduce_scatter_f32(cumf_comm* c, const float* send, float* recv, size_t count) {
  if (count == 0) return 0;
  switch (c->kind) {
    case kRccl:
      DIST_NCCL_CHECK(rccl()->ReduceScatter(send, recv, count, ncclFloat, ncclSum, c->nccl, c->comm_stream));
      return 0;
    case kCustom:
      return c->custom.reduce_scatter_f32(c->custom.ctx, send, recv, count, c->comm_stream);
    default:
      CUMF_HIP_CHECK(hipMemcpyAsync(recv, send, count * sizeof(float), hipMemcpyDeviceToDevice, c->comm_stream));
      return 0;
  }
}

int all_reduce_f64(cumf_comm* c, double* buf, size_t count) {
  if (count == 0) return 0;
  switch (c->kind) {
    case kRccl:
      DIST_NCCL_CHECK(rccl()->AllReduce(buf, buf, count, ncclDouble, ncclSum, c->nccl, c->comm_stream));
      return 0;
    case kCustom:
      return c->custom.all_reduce_f64(c->custom.ctx, buf, count, c->comm_stream);
    default:
      return 0;
  }
}

// `second` runs behind everything enqueued on `first` so far
int order(hipStream_t first, hipStream_t second, hipEvent_t ev) {
  CUMF_HIP_CHECK(hipEventRecord(ev, first));
  CUMF_HIP_CHECK(hipStreamWaitEvent(second, ev, 0));
  return 0;
}

template <typename T>
int dev_alloc(T** p, size_t count) {
  *p = nullptr;
  CUMF_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(p), std::max<size_t>(count, 1) * sizeof(T)));
  return 0;
}

}  // namespace

// ---- communicator -----------------------------------------------------------------------------------------------

extern "C" int cumf_comm_unique_id(void* id) {
  static_assert(sizeof(ncclUniqueId) == CUMF_COMM_ID_BYTES, "id size");
  if (!id) return (int)hipErrorInvalidValue;
  if (!rccl()) return (int)hipErrorSharedObjectInitFailed;
  ncclUniqueId u;
  DIST_NCCL_CHECK(rccl()->GetUniqueId(&u));
  memcpy(id, &u, sizeof(u));
  return 0;
}

extern "C" int cumf_comm_create(cumf_comm_t** comm, const void* id, int rank, int world) {
  if (!comm || !id || world < 1 || rank < 0 || rank >= world) return (int)hipErrorInvalidValue;
  if (!rccl()) return (int)hipErrorSharedObjectInitFailed;
  cumf_comm* c = new cumf_comm;
  c->kind = kRccl;
  c->rank = rank;
  c->world = world;
  DIST_CHECK(comm_init_common(c));
  ncclUniqueId u;
  memcpy(&u, id, sizeof(u));
  DIST_NCCL_CHECK(rccl()->CommInitRank(&c->nccl, world, u, rank));
  *comm = c;
  return 0;
}

extern "C" int cumf_comm_create_local(cumf_comm_t** comm) {
  if (!comm) return (int)hipErrorInvalidValue;
  cumf_comm* c = new cumf_comm;
  DIST_CHECK(comm_init_common(c));
  *comm = c;
  return 0;
}

extern "C" int cumf_comm_create_custom(cumf_comm_t** comm, const cumf_transport_t* t, int rank, int world) {
  if (!comm || !t || !t->all_gather || !t->reduce_scatter_f32 || !t->all_reduce_f64 || world < 1 || rank < 0 ||
      rank >= world)
    return (int)hipErrorInvalidValue;
  cumf_comm* c = new cumf_comm;
  c->kind = kCustom;
  c->rank = rank;
  c->world = world;
  c->custom = *t;
  DIST_CHECK(comm_init_common(c));
  *comm = c;
  return 0;
}

extern "C" int cumf_comm_destroy(cumf_comm_t* c) {
  if (!c) return 0;
  if (c->comm_stream) (void)hipStreamSynchronize(c->comm_stream);
  if (c->kind == kRccl && c->nccl) (void)rccl()->CommDestroy(c->nccl);
  if (c->ev_in) (void)hipEventDestroy(c->ev_in);
  if (c->ev_out) (void)hipEventDestroy(c->ev_out);
  if (c->comm_stream) (void)hipStreamDestroy(c->comm_stream);
  delete c;
  return 0;
}

extern "C" int cumf_comm_rank(const cumf_comm_t* c) { return c ? c->rank : -1; }
extern "C" int cumf_comm_world(const cumf_comm_t* c) { return c ? c->world : -1; }
extern "C" const char* cumf_comm_transport_name(const cumf_comm_t* c) {
  return !c ? "" : c->kind == kRccl ? "rccl" : c->kind == kCustom ? "custom" : "local";
}

extern "C" int cumf_comm_all_reduce_f64(cumf_comm_t* c, double* buf, long count, void* stream) {
  if (!c || !buf || count < 0) return (int)hipErrorInvalidValue;
  if (c->world == 1 && c->kind == kLocal) return 0;
  hipStream_t s = static_cast<hipStream_t>(stream);
  DIST_CHECK(order(s, c->comm_stream, c->ev_in));
  DIST_CHECK(all_reduce_f64(c, buf, (size_t)count));
  DIST_CHECK(order(c->comm_stream, s, c->ev_out));
  return 0;
}

// ---- gather scheme ------------------------------------------------------------------------------------------------

struct cumf_dist_gather {
  cumf_comm* comm = nullptr;
  int pieces = 0, f = 0;
  std::vector<long long> pb;        // world x (pieces + 1)
  std::vector<long long> mx;        // rows of the largest piece c over the ranks
  std::vector<size_t> recv_off;     // floats, per piece
  float* send = nullptr;            // max over c of mx[c] rows
  float* recv = nullptr;            // sum over c of world * mx[c] rows
  std::vector<hipEvent_t> ev_piece; // kernel of piece c done (compute stream)
  hipEvent_t ev_done = nullptr;
  long long bound(int g, int c) const { return pb[(size_t)g * (pieces + 1) + c]; }
};

extern "C" int cumf_dist_gather_create(cumf_dist_gather_t** out, cumf_comm_t* comm, const long long* piece_bounds,
                                       int pieces, int f) {
  if (!out || !comm || !piece_bounds || pieces < 1 || f < 1) return (int)hipErrorInvalidValue;
  const int w = comm->world;
  for (int g = 0; g < w; ++g)
    for (int c = 0; c < pieces; ++c)
      if (piece_bounds[(size_t)g * (pieces + 1) + c] > piece_bounds[(size_t)g * (pieces + 1) + c + 1]) {
        fprintf(stderr, "cumf_dist_gather_create: piece bounds of rank %d decrease\n", g);
        return (int)hipErrorInvalidValue;
      }
  cumf_dist_gather* s = new cumf_dist_gather;
  s->comm = comm;
  s->pieces = pieces;
  s->f = f;
  s->pb.assign(piece_bounds, piece_bounds + (size_t)w * (pieces + 1));
  s->mx.resize(pieces);
  s->recv_off.resize(pieces + 1);
  long long mx_all = 0;
  size_t total = 0;
  for (int c = 0; c < pieces; ++c) {
    long long m = 0;
    for (int g = 0; g < w; ++g) m = std::max(m, s->bound(g, c + 1) - s->bound(g, c));
    s->mx[c] = m;
    mx_all = std::max(mx_all, m);
    s->recv_off[c] = total;
    total += (size_t)w * m * f;
  }
  s->recv_off[pieces] = total;
  DIST_CHECK(dev_alloc(&s->send, (size_t)mx_all * f));
  DIST_CHECK(dev_alloc(&s->recv, total));
  // the rows behind a short piece are sent as they are: defined once, never NaN-producing garbage
  CUMF_HIP_CHECK(hipMemset(s->send, 0, std::max<size_t>((size_t)mx_all * f, 1) * sizeof(float)));
  s->ev_piece.resize(pieces);
  for (auto& e : s->ev_piece) CUMF_HIP_CHECK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
  CUMF_HIP_CHECK(hipEventCreateWithFlags(&s->ev_done, hipEventDisableTiming));
  *out = s;
  return 0;
}

extern "C" int cumf_dist_gather_update(cumf_dist_gather_t* s, const cumf_plan_t* const* piece_plans, const int* colidx,
                                       const float* val, const float* table, float* out, float lambda, int solver,
                                       int cg_iters, double* sse_bins, void* stream) {
  if (!s || !piece_plans || !out) return (int)hipErrorInvalidValue;
  cumf_comm* c = s->comm;
  const int w = c->world, me = c->rank, f = s->f;
  hipStream_t S = static_cast<hipStream_t>(stream), C = c->comm_stream;
  float* mine = out + (size_t)s->bound(me, 0) * f;  // the plans index the slab's local rows
  // the buffers of the previous call are free: its ev_done was awaited by that call's stream; a new caller stream must
  // not overtake it either
  CUMF_HIP_CHECK(hipStreamWaitEvent(S, s->ev_done, 0));
  for (int p = 0; p < s->pieces; ++p) {
    if (piece_plans[p]) {
      if (sse_bins)
        DIST_CHECK(cumf_als_update_fused_sse(piece_plans[p], colidx, val, table, mine, f, lambda, solver, cg_iters,
                                             sse_bins, stream));
      else
        DIST_CHECK(cumf_als_update_fused(piece_plans[p], colidx, val, table, mine, f, lambda, solver, cg_iters, stream));
    }
    if (s->mx[p] == 0) continue;  // empty on every rank (the bounds are the same everywhere)
    if (w == 1 && c->kind == kLocal) continue;  // nothing to exchange: the rows are already in place
    DIST_CHECK(order(S, C, s->ev_piece[p]));
    const long long lo = s->bound(me, p), n_mine = s->bound(me, p + 1) - lo;
    const size_t piece_floats = (size_t)s->mx[p] * f;
    // a piece as long as the largest is sent from where it lies; a shorter one through the padded send buffer
    const float* src = out + (size_t)lo * f;
    if (n_mine < s->mx[p]) {
      if (n_mine > 0)
        CUMF_HIP_CHECK(hipMemcpyAsync(s->send, src, (size_t)n_mine * f * sizeof(float), hipMemcpyDeviceToDevice, C));
      src = s->send;
    }
    float* recv = s->recv + s->recv_off[p];
    DIST_CHECK(all_gather_bytes(c, src, recv, piece_floats * sizeof(float)));
    for (int g = 0; g < w; ++g) {
      const long long glo = s->bound(g, p), gn = s->bound(g, p + 1) - glo;
      if (g == me || gn == 0) continue;
      CUMF_HIP_CHECK(hipMemcpyAsync(out + (size_t)glo * f, recv + (size_t)g * piece_floats,
                                    (size_t)gn * f * sizeof(float), hipMemcpyDeviceToDevice, C));
    }
  }
  DIST_CHECK(order(C, S, s->ev_done));
  return 0;
}

extern "C" int cumf_dist_gather_destroy(cumf_dist_gather_t* s) {
  if (!s) return 0;
  if (s->comm && s->comm->comm_stream) (void)hipStreamSynchronize(s->comm->comm_stream);
  for (auto& e : s->ev_piece) (void)hipEventDestroy(e);
  if (s->ev_done) (void)hipEventDestroy(s->ev_done);
  (void)hipFree(s->send);
  (void)hipFree(s->recv);
  delete s;
  return 0;
}

// ---- reduce scheme: Theta update ------------------------------------------------------------------------------------

struct cumf_dist_reduce {
  cumf_comm* comm = nullptr;
  long n = 0;
  int f = 0, theta_batch = 1, nbuf = 1;
  long kmax = 0;
  size_t pk = 0;                      // f (f + 1) / 2
  std::vector<long> off, size;        // Theta batches (als.cu:881-890)
  float *tri[2] = {nullptr, nullptr}, *rhs[2] = {nullptr, nullptr};      // partial systems of the batch, world * k each
  float *mine[2] = {nullptr, nullptr}, *mine_rhs[2] = {nullptr, nullptr};  // this rank's k reduced systems
  float* x[2] = {nullptr, nullptr};   // their solutions (warm start in)
  float* my_tt = nullptr;             // k x f x f, unpacked for the solvers
  float* gathered = nullptr;          // world * k solved rows
  hipEvent_t ev_gram[2] = {nullptr, nullptr}, ev_rs[2] = {nullptr, nullptr}, ev_solve[2] = {nullptr, nullptr};
  hipEvent_t ev_done = nullptr;
};

extern "C" int cumf_dist_reduce_create(cumf_dist_reduce_t** out, cumf_comm_t* comm, long n, int f, int theta_batch) {
  if (!out || !comm || n < 1 || f < 1 || theta_batch < 1 || theta_batch > n) return (int)hipErrorInvalidValue;
  cumf_dist_reduce* r = new cumf_dist_reduce;
  r->comm = comm;
  r->n = n;
  r->f = f;
  r->theta_batch = theta_batch;
  r->pk = (size_t)f * (f + 1) / 2;
  const int w = comm->world;
  for (int b = 0; b < theta_batch; ++b) {
    const long bs = (b != theta_batch - 1) ? n / theta_batch : n - (long)b * (n / theta_batch);
    r->off.push_back((long)b * (n / theta_batch));
    r->size.push_back(bs);
    r->kmax = std::max(r->kmax, (bs + w - 1) / w);
  }
  r->nbuf = theta_batch > 1 ? 2 : 1;
  const size_t wk = (size_t)w * r->kmax;
  for (int i = 0; i < r->nbuf; ++i) {
    DIST_CHECK(dev_alloc(&r->tri[i], wk * r->pk));
    DIST_CHECK(dev_alloc(&r->rhs[i], wk * f));
    DIST_CHECK(dev_alloc(&r->mine[i], (size_t)r->kmax * r->pk));
    DIST_CHECK(dev_alloc(&r->mine_rhs[i], (size_t)r->kmax * f));
    DIST_CHECK(dev_alloc(&r->x[i], (size_t)r->kmax * f));
    CUMF_HIP_CHECK(hipMemset(r->x[i], 0, std::max<size_t>((size_t)r->kmax * f, 1) * sizeof(float)));
    CUMF_HIP_CHECK(hipEventCreateWithFlags(&r->ev_gram[i], hipEventDisableTiming));
    CUMF_HIP_CHECK(hipEventCreateWithFlags(&r->ev_rs[i], hipEventDisableTiming));
    CUMF_HIP_CHECK(hipEventCreateWithFlags(&r->ev_solve[i], hipEventDisableTiming));
  }
  DIST_CHECK(dev_alloc(&r->my_tt, (size_t)r->kmax * f * f));
  DIST_CHECK(dev_alloc(&r->gathered, wk * f));
  CUMF_HIP_CHECK(hipEventCreateWithFlags(&r->ev_done, hipEventDisableTiming));
  *out = r;
  return 0;
}

extern "C" int cumf_dist_reduce_update_theta(cumf_dist_reduce_t* r, const cumf_plan_t* const* batch_plans,
                                             const int* lc_rowidx, const float* lc_val, const float* XT_slab,
                                             float* thetaT, float lambda, int solver, int cg_iters,
                                             const float* reg_all, double* sse_terms, void* stream) {
  if (!r || !batch_plans || !thetaT || (reg_all == nullptr) != (sse_terms == nullptr)) return (int)hipErrorInvalidValue;
  cumf_comm* c = r->comm;
  const int w = c->world, me = c->rank, f = r->f;
  hipStream_t S = static_cast<hipStream_t>(stream), C = c->comm_stream;
  CUMF_HIP_CHECK(hipStreamWaitEvent(S, r->ev_done, 0));  // a previous call's last placement into thetaT

  // solve + all-gather of batch b, whose reduce-scatter is in flight on the communication stream
  auto finish = [&](int b) -> int {
    const int slot = b % r->nbuf;
    const long size = r->size[b], off = r->off[b], k = (size + w - 1) / w;
    const long lo = std::min((long)me * k, size), hi = std::min((long)(me + 1) * k, size), cnt = hi - lo;
    CUMF_HIP_CHECK(hipStreamWaitEvent(S, r->ev_rs[slot], 0));
    if (cnt > 0) {
      DIST_CHECK(cumf_unpack_upper(r->mine[slot], r->my_tt, cnt, f, stream));
      // CG warm start (cg.cu:36-231 iterates from the previous factors)
      CUMF_HIP_CHECK(hipMemcpyAsync(r->x[slot], thetaT + (size_t)(off + lo) * f, (size_t)cnt * f * sizeof(float),
                                    hipMemcpyDeviceToDevice, S));
      if (solver == CUMF_SOLVER_CG)
        DIST_CHECK(cumf_cg_solve_batched(r->my_tt, r->x[slot], r->mine_rhs[slot], cnt, f, cg_iters, stream));
      else
        DIST_CHECK(cumf_lu_solve_batched(r->my_tt, r->mine_rhs[slot], r->x[slot], cnt, f, stream));
      if (sse_terms)  // the solvers leave A and b intact (f <= 200)
        DIST_CHECK(cumf_quadratic_sse_terms(r->my_tt, r->mine_rhs[slot], r->x[slot], reg_all + off + lo, cnt, f,
                                            sse_terms, stream));
    }
    DIST_CHECK(order(S, C, r->ev_solve[slot]));
    DIST_CHECK(all_gather_bytes(c, r->x[slot], r->gathered, (size_t)k * f * sizeof(float)));  // hugewiki.cu:2744-2745
    CUMF_HIP_CHECK(hipMemcpyAsync(thetaT + (size_t)off * f, r->gathered, (size_t)size * f * sizeof(float),
                                  hipMemcpyDeviceToDevice, C));
    return 0;
  };

  int pending = -1;
  for (int b = 0; b < r->theta_batch; ++b) {
    const int slot = b % r->nbuf;
    const long size = r->size[b], k = (size + w - 1) / w;
    float *tri = r->tri[slot], *rhs = r->rhs[slot];
    // partial Gram / RHS over this rank's X slab (hugewiki.cu:2668-2679), as packed upper triangles
    DIST_CHECK(cumf_get_hermitian_packed(batch_plans[b], lc_rowidx, lc_val, XT_slab, tri, rhs, f, lambda, stream));
    if (size < (long)w * k) {  // the padding systems behind the last rank's share
      CUMF_HIP_CHECK(hipMemsetAsync(tri + (size_t)size * r->pk, 0, (size_t)(w * k - size) * r->pk * sizeof(float), S));
      CUMF_HIP_CHECK(hipMemsetAsync(rhs + (size_t)size * f, 0, (size_t)(w * k - size) * f * sizeof(float), S));
    }
    // the reduce-scatter of batch b starts the moment its Gram pass ends: under the solve of batch b - 1 and the Gram
    // pass of batch b + 1 (hugewiki.cu:2703-2730 is a serial copy + axpy per GPU with every other GPU idle)
    DIST_CHECK(order(S, C, r->ev_gram[slot]));
    DIST_CHECK(reduce_scatter_f32(c, tri, r->mine[slot], (size_t)k * r->pk));
    DIST_CHECK(reduce_scatter_f32(c, rhs, r->mine_rhs[slot], (size_t)k * f));
    CUMF_HIP_CHECK(hipEventRecord(r->ev_rs[slot], C));
    if (pending >= 0) DIST_CHECK(finish(pending));
    pending = b;
  }
  if (pending >= 0) DIST_CHECK(finish(pending));
  DIST_CHECK(order(C, S, r->ev_done));
  return 0;
}

extern "C" int cumf_dist_reduce_destroy(cumf_dist_reduce_t* r) {
  if (!r) return 0;
  if (r->comm && r->comm->comm_stream) (void)hipStreamSynchronize(r->comm->comm_stream);
  for (int i = 0; i < 2; ++i) {
    (void)hipFree(r->tri[i]);
    (void)hipFree(r->rhs[i]);
    (void)hipFree(r->mine[i]);
    (void)hipFree(r->mine_rhs[i]);
    (void)hipFree(r->x[i]);
    if (r->ev_gram[i]) (void)hipEventDestroy(r->ev_gram[i]);
    if (r->ev_rs[i]) (void)hipEventDestroy(r->ev_rs[i]);
    if (r->ev_solve[i]) (void)hipEventDestroy(r->ev_solve[i]);
  }
  (void)hipFree(r->my_tt);
  (void)hipFree(r->gathered);
  if (r->ev_done) (void)hipEventDestroy(r->ev_done);
  delete r;
  return 0;
}
