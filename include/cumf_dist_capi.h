/*
 * cumf_dist_capi.h -- C ABI of the multi-GPU half-iterations of libALS.so (one process per GPU).
 *
 * Replaces the in-process multi-GPU loop of the reference's compiled program hugewiki/hugewiki.cu
 * (`main`, hugewiki.cu:2248-2888): OpenMP thread per GPU, P2P cudaMemcpy into a staging buffer + cublasSaxpy on
 * GPU 0 (hugewiki.cu:2703-2730), three cudaMemcpy broadcasts of Theta (hugewiki.cu:2744-2745).  Here every rank
 * owns one GPU and the exchange steps are RCCL collectives over xGMI, enqueued from C++ back to back with the
 * kernels they depend on: no host code between a kernel and its collective, no host synchronisation inside a
 * half-iteration.  RCCL is loaded at run time (dlopen: the copy already in the process -- torch ships one -- or
 * librccl.so.1; environment CUMF_RCCL_LIB overrides), so libALS.so itself has no link-time dependency on it.
 *
 * Conventions as in cumf_als_capi.h: plain pointers and sizes, device pointers of the calling process, `stream`
 * a hipStream_t passed as void*, 0 on success / non-zero after a message on stderr, never a CPU path.
 */
#ifndef CUMF_DIST_CAPI_H_
#define CUMF_DIST_CAPI_H_

#include <stddef.h>
#include <stdint.h>

#include "cumf_als_capi.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- communicator ------------------------------------------------------------------------------------------
 * One per process (GPU).  Owns a communication stream (collectives run on it, ordered against the caller's
 * compute stream by events) and the transport:
 *   cumf_comm_create         RCCL: `id` = the CUMF_COMM_ID_BYTES bytes rank 0 obtained from cumf_comm_unique_id and
 *                            handed to every rank (file, pipe, torch.distributed broadcast ...); the communicator
 *                            is bound to the CURRENT HIP device (ncclCommInitRank).  Replaces enableP2P,
 *                            hugewiki/common.h:19-36.
 *   cumf_comm_create_local   world = 1, collectives are device copies (single-GPU runs of the same code path).
 *   cumf_comm_create_custom  caller-supplied collectives (MPI, a test harness over gloo ...).  Each callback is
 *                            called with the communication stream and must be ordered on it: either enqueue on
 *                            that stream or synchronise it, do the exchange, and return when the result is in
 *                            `recv`.  Buffers are DEVICE pointers.
 */
#define CUMF_COMM_ID_BYTES 128
typedef struct cumf_comm cumf_comm_t;
typedef struct cumf_transport {
  void* ctx;
  /* recv[g * bytes .. (g + 1) * bytes) <- rank g's send[0 .. bytes) */
  int (*all_gather)(void* ctx, const void* send, void* recv, size_t bytes, void* stream);
  /* recv[0 .. count) <- sum over ranks of their send[rank * count .. (rank + 1) * count) (fp32) */
  int (*reduce_scatter_f32)(void* ctx, const float* send, float* recv, size_t count, void* stream);
  /* buf[0 .. count) <- sum over ranks (fp64, in place) */
  int (*all_reduce_f64)(void* ctx, double* buf, size_t count, void* stream);
} cumf_transport_t;
int cumf_comm_unique_id(void* id);
int cumf_comm_create(cumf_comm_t** comm, const void* id, int rank, int world);
int cumf_comm_create_local(cumf_comm_t** comm);
int cumf_comm_create_custom(cumf_comm_t** comm, const cumf_transport_t* transport, int rank, int world);
int cumf_comm_destroy(cumf_comm_t* comm);
int cumf_comm_rank(const cumf_comm_t* comm);
int cumf_comm_world(const cumf_comm_t* comm);
/* "rccl" | "local" | "custom" */
const char* cumf_comm_transport_name(const cumf_comm_t* comm);
/* Sum of `count` doubles over the ranks, in place in DEVICE memory, ordered behind `stream` and `stream` behind it
 * (the per-GPU SSE sums of hugewiki.cu:2848-2858, the counts, the fused train-SSE terms). */
int cumf_comm_all_reduce_f64(cumf_comm_t* comm, double* buf, long count, void* stream);

/* ---- `gather` scheme: one side of the factorisation, row-sharded update + all-gather of the factor rows -----
 * (hugewiki.cu:2436-2602 assigns X row batches to GPUs and keeps X on the host between phases; here both factor
 * matrices are replicated in HBM and every rank updates a contiguous slab of rows.)
 * The slab of rank g is cut into `pieces` contiguous pieces; piece_bounds (HOST, world x (pieces + 1) global row ids,
 * row-major, identical on every rank) names them.  cumf_dist_gather_update runs, for c = 0 .. pieces-1,
 *   compute stream:  the fused half-iteration of piece c (cumf_als_update_fused[_sse] with piece_plans[c])
 *   comm stream:     behind it, ONE all-gather of piece c of every rank (padded to the largest), then the received
 *                    rows placed into `out` by device copies
 * so that the exchange of piece c runs under the kernel of piece c + 1; `stream` finally waits for the last piece.
 * out: the FULL factor matrix (rows_total x f) on every rank; this rank's rows are computed in place, the others
 * received.  piece_plans[c]: plan over the slab's LOCAL row pointer restricted to piece c (NULL = empty piece);
 * colidx / val: the slab's CSR arrays; table: the gathered factor matrix.  sse_bins: NULL or as in
 * cumf_als_update_fused_sse (local to this rank: sum them with cumf_comm_all_reduce_f64).
 */
typedef struct cumf_dist_gather cumf_dist_gather_t;
int cumf_dist_gather_create(cumf_dist_gather_t** g, cumf_comm_t* comm, const long long* piece_bounds, int pieces,
                            int f);
int cumf_dist_gather_update(cumf_dist_gather_t* g, const cumf_plan_t* const* piece_plans, const int* colidx,
                            const float* val, const float* table, float* out, float lambda, int solver,
                            int cg_iters, double* sse_bins, void* stream);
int cumf_dist_gather_destroy(cumf_dist_gather_t* g);

/* ---- `reduce` scheme: the Theta update from row-sharded X (hugewiki.cu:2611-2745) ---------------------------
 * Rank g holds its X slab and the slab-local CSC of its ratings (hugewiki.cu:2332-2340).  Per Theta batch b
 * (THETA_BATCH slices of the n columns as als.cu:881-890 cuts them):
 *   compute stream:  partial Gram + RHS of the batch over the slab, packed upper triangles (cumf_get_hermitian_packed:
 *                    lambda * n_local on the diagonal)                                      [hugewiki.cu:2668-2679]
 *   comm stream:     reduce-scatter of triangles and right-hand sides -- rank g receives the SUMMED systems
 *                    [g k, (g + 1) k) of the batch, k = ceil(size / world) -- under the Gram pass of batch b + 1
 *                                                                                       [hugewiki.cu:2703-2730]
 *   compute stream:  unpack, warm start from thetaT, batched CG / LU on the k systems      [hugewiki.cu:2732-2741]
 *   comm stream:     all-gather of the k solved rows, placed into thetaT                   [hugewiki.cu:2744-2745]
 * batch_plans[b]: plan over the slab-local CSC column pointer restricted to batch b.  reg_all / sse_terms (both NULL,
 * or n floats and one double in DEVICE memory): the fused train SSE -- *sse_terms += this rank's part of
 * sum 2 t.b - t^T A t + reg |t|^2 (cumf_quadratic_sse_terms; reg_all[v] = lambda * n_v over ALL ranks, < 0 for a column
 * without ratings); the caller all-reduces it and subtracts it from sum r^2.
 */
typedef struct cumf_dist_reduce cumf_dist_reduce_t;
int cumf_dist_reduce_create(cumf_dist_reduce_t** r, cumf_comm_t* comm, long n, int f, int theta_batch);
int cumf_dist_reduce_update_theta(cumf_dist_reduce_t* r, const cumf_plan_t* const* batch_plans, const int* lc_rowidx,
                                  const float* lc_val, const float* XT_slab, float* thetaT, float lambda, int solver,
                                  int cg_iters, const float* reg_all, double* sse_terms, void* stream);
int cumf_dist_reduce_destroy(cumf_dist_reduce_t* r);

#ifdef __cplusplus
}
#endif
#endif /* CUMF_DIST_CAPI_H_ */
