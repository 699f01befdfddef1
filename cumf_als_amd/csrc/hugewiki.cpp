// hugewiki.cpp -- the multi-GPU ALS program, one process per GPU: the counterpart of the reference's compiled
// hugewiki/hugewiki.cu `main` (hugewiki.cu:2248-2888), which runs one OpenMP thread per GPU inside one process with
// everything #define'd (hugewiki.cu:27-42).  Here:
//
//   ./hugewiki [--gpus G] SPLIT_DIR N F lambda ITERS THETA_BATCH
//
//   SPLIT_DIR   per-GPU slab files written by `python -m cumf_als_amd.convert split` -- the pre-split
//               `R_train_csr.*.bin<g>` / `R_train_csc.*.bin<g>` (slab-local row ids) / `R_test_coo.*.bin<g>` inputs of
//               hugewiki.cu:2332-2354 -- and slabs.txt (the row boundaries)
//   --gpus G    fork G ranks here (GPU g = rank g).  Without it the rank comes from the environment a launcher sets
//               (RANK, WORLD_SIZE, LOCAL_RANK; e.g. `python -m torch.distributed.run --no-python --nproc-per-node 8
//               ./hugewiki ...`), or the program runs on one GPU.
//   environment CUMF_ALS_SOLVER=cg|lu (cg), CUMF_ALS_CG_ITERS (6), CUMF_ALS_SOLVER_X / _THETA, CUMF_ALS_CG_ITERS_X / _THETA
//               (per side), CUMF_ALS_REFERENCE_SOLVERS=1 = what hugewiki.cu runs: X by CG(100) (hugewiki.cu:2569), Theta by
//               the batched LU (hugewiki.cu:2732); CUMF_DIST_ID_FILE = where rank 0 leaves the RCCL id for the others;
//               CUMF_ALS_DUMP_MODEL=<dir>: thetaT.data (rank 0) and XT.data<g> (every rank's slab) after the last iteration.
//
// X is row-sharded and stays in HBM (the reference round-trips it through the host, hugewiki.cu:2571,2641), Theta is
// replicated; the Theta update is cumf_dist_reduce_update_theta (partial Grams -> RCCL reduce-scatter -> solve ->
// all-gather).  Initialisation as hugewiki.cu:2381-2393: srand(0), theta = 0.2 rand()/RAND_MAX on every rank, X = 0.
// Prints the reference's RMSE lines from rank 0.
#include <hip/hip_runtime.h>
#include <sys/stat.h>
#include <sys/wait.h>
#include <unistd.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <thread>
#include <vector>

#include "cumf_dist_capi.h"

namespace {

#define HW_CHECK(call)                                                                                          \
  do {                                                                                                          \
    int rc__ = (int)(call);                                                                                     \
    if (rc__ != 0) {                                                                                            \
      fprintf(stderr, "hugewiki: %s failed with code %d (%s:%d)\n", #call, rc__, __FILE__, __LINE__);           \
      exit(EXIT_FAILURE);                                                                                       \
    }                                                                                                           \
  } while (0)

int env_int(const char* name, int dflt) {
  const char* e = getenv(name);
  return e && *e ? atoi(e) : dflt;
}

int env_solver(const char* name, int dflt) {
  const char* e = getenv(name);
  if (!e || !*e) return dflt;
  if (!strcmp(e, "lu") || !strcmp(e, "LU")) return CUMF_SOLVER_LU;
  if (!strcmp(e, "cg") || !strcmp(e, "CG")) return CUMF_SOLVER_CG;
  fprintf(stderr, "hugewiki: %s=%s (cg or lu)\n", name, e);
  exit(EXIT_FAILURE);
}

template <typename T>
std::vector<T> read_file(const std::string& path, long expect = -1) {
  FILE* fp = fopen(path.c_str(), "rb");
  if (!fp) {
    fprintf(stderr, "hugewiki: cannot open %s\n", path.c_str());
    exit(EXIT_FAILURE);
  }
  struct stat st;
  fstat(fileno(fp), &st);
  const size_t count = (size_t)st.st_size / sizeof(T);
  if ((size_t)st.st_size % sizeof(T) || (expect >= 0 && count != (size_t)expect)) {
    fprintf(stderr, "hugewiki: %s holds %zu elements, expected %ld\n", path.c_str(), count, expect);
    exit(EXIT_FAILURE);
  }
  std::vector<T> v(count);
  if (count && fread(v.data(), sizeof(T), count, fp) != count) {
    fprintf(stderr, "hugewiki: short read of %s\n", path.c_str());
    exit(EXIT_FAILURE);
  }
  fclose(fp);
  return v;
}

template <typename T>
T* to_device(const std::vector<T>& v) {
  T* d = nullptr;
  HW_CHECK(hipMalloc(reinterpret_cast<void**>(&d), (v.size() ? v.size() : 1) * sizeof(T)));
  if (!v.empty()) HW_CHECK(hipMemcpy(d, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  return d;
}

// the RCCL id travels through a file: rank 0 writes it (temporary name + rename), the others wait for it
std::string id_file_path() {
  if (const char* e = getenv("CUMF_DIST_ID_FILE")) return e;
  const char* port = getenv("MASTER_PORT");
  return std::string("/tmp/cumf_dist_id.") + (port ? port : std::to_string((long)getppid()).c_str());
}

void exchange_id(int rank, unsigned char* id) {
  const std::string path = id_file_path();
  if (rank == 0) {
    HW_CHECK(cumf_comm_unique_id(id));
    const std::string tmp = path + ".tmp";
    FILE* fp = fopen(tmp.c_str(), "wb");
    if (!fp || fwrite(id, 1, CUMF_COMM_ID_BYTES, fp) != CUMF_COMM_ID_BYTES) {
      fprintf(stderr, "hugewiki: cannot write %s\n", tmp.c_str());
      exit(EXIT_FAILURE);
    }
    fclose(fp);
    rename(tmp.c_str(), path.c_str());
    return;
  }
  const time_t started = time(nullptr);
  const int timeout_s = env_int("CUMF_DIST_ID_TIMEOUT", 300);
  for (;;) {
    struct stat st;
    // a file older than this process (minus launcher skew) is a leftover of an earlier run
    if (stat(path.c_str(), &st) == 0 && st.st_size == CUMF_COMM_ID_BYTES && st.st_mtime + 120 >= started) {
      FILE* fp = fopen(path.c_str(), "rb");
      if (fp && fread(id, 1, CUMF_COMM_ID_BYTES, fp) == CUMF_COMM_ID_BYTES) {
        fclose(fp);
        return;
      }
      if (fp) fclose(fp);
    }
    if (time(nullptr) - started > timeout_s) {
      fprintf(stderr, "hugewiki: rank %d waited %d s for %s\n", rank, timeout_s, path.c_str());
      exit(EXIT_FAILURE);
    }
    std::this_thread::sleep_for(std::chrono::milliseconds(20));
  }
}

double sum_over_ranks(cumf_comm_t* comm, double v, double* d_scratch) {
  HW_CHECK(hipMemcpy(d_scratch, &v, sizeof(double), hipMemcpyHostToDevice));
  HW_CHECK(cumf_comm_all_reduce_f64(comm, d_scratch, 1, nullptr));
  HW_CHECK(hipStreamSynchronize(nullptr));
  HW_CHECK(hipMemcpy(&v, d_scratch, sizeof(double), hipMemcpyDeviceToHost));
  return v;
}

int run_rank(int rank, int world, int local_rank, const std::string& dir, long n, int f, float lambda, int iters,
             int theta_batch) {
  HW_CHECK(hipSetDevice(local_rank));
  int solver = env_solver("CUMF_ALS_SOLVER", CUMF_SOLVER_CG), cg_iters = env_int("CUMF_ALS_CG_ITERS", 6);
  int solver_x = env_solver("CUMF_ALS_SOLVER_X", solver), solver_t = env_solver("CUMF_ALS_SOLVER_THETA", solver);
  int cg_x = env_int("CUMF_ALS_CG_ITERS_X", cg_iters), cg_t = env_int("CUMF_ALS_CG_ITERS_THETA", cg_iters);
  if (env_int("CUMF_ALS_REFERENCE_SOLVERS", 0)) {
    solver_x = CUMF_SOLVER_CG;
    cg_x = 100;
    solver_t = CUMF_SOLVER_LU;
  }

  cumf_comm_t* comm = nullptr;
  if (world == 1 && !env_int("CUMF_DIST_FORCE_RCCL", 0)) {  // (the switch: a one-rank RCCL communicator, for the tests)
    HW_CHECK(cumf_comm_create_local(&comm));
  } else {
    unsigned char id[CUMF_COMM_ID_BYTES];
    exchange_id(rank, id);
    HW_CHECK(cumf_comm_create(&comm, id, rank, world));
    if (rank == 0) unlink(id_file_path().c_str());  // ncclCommInitRank returns when every rank has joined
  }

  // slab boundaries and this rank's files
  std::vector<long long> bounds;
  {
    FILE* fp = fopen((dir + "/slabs.txt").c_str(), "r");
    long long b;
    while (fp && fscanf(fp, "%lld", &b) == 1) bounds.push_back(b);
    if (fp) fclose(fp);
    if ((int)bounds.size() != world + 1) {
      fprintf(stderr, "hugewiki: %s was split for %d GPUs, launched with %d\n", dir.c_str(), (int)bounds.size() - 1, world);
      return EXIT_FAILURE;
    }
  }
  const long rows = (long)(bounds[rank + 1] - bounds[rank]);
  const std::string g = std::to_string(rank);
  auto path = [&](const char* name) { return dir + "/" + name + g; };
  const auto csr_indptr = read_file<int>(path("R_train_csr.indptr.bin"), rows + 1);
  const long nnz_l = csr_indptr[rows];
  const auto csr_indices = read_file<int>(path("R_train_csr.indices.bin"), nnz_l);
  const auto csr_data = read_file<float>(path("R_train_csr.data.bin"), nnz_l);
  const auto csc_indptr = read_file<int>(path("R_train_csc.indptr.bin"), n + 1);
  const auto csc_indices = read_file<int>(path("R_train_csc.indices.bin"), nnz_l);
  const auto csc_data = read_file<float>(path("R_train_csc.data.bin"), nnz_l);
  const auto test_data = read_file<float>(path("R_test_coo.data.bin"));
  const long nnz_test_l = (long)test_data.size();
  const auto test_row = read_file<int>(path("R_test_coo.row.bin"), nnz_test_l);
  const auto test_col = read_file<int>(path("R_test_coo.col.bin"), nnz_test_l);

  int* d_colidx = to_device(csr_indices);
  float* d_val = to_device(csr_data);
  int* d_lc_row = to_device(csc_indices);
  float* d_lc_val = to_device(csc_data);
  int *d_test_row = to_device(test_row), *d_test_col = to_device(test_col);
  float* d_test_val = to_device(test_data);

  // plans: the X slab (whole), the Theta batches over the slab-local CSC (als.cu:881-890)
  cumf_plan_t* x_plan = nullptr;
  HW_CHECK(cumf_plan_create(&x_plan, csr_indptr.data(), 0, rows, 0, rows, f, 0));
  HW_CHECK(cumf_plan_set_gather_rows(x_plan, n));
  std::vector<cumf_plan_t*> t_plans(theta_batch, nullptr);
  for (int b = 0; b < theta_batch; ++b) {
    const long bs = (b != theta_batch - 1) ? n / theta_batch : n - (long)b * (n / theta_batch);
    const long off = (long)b * (n / theta_batch);
    HW_CHECK(cumf_plan_create(&t_plans[b], csc_indptr.data(), 0, n, off, off + bs, f, 0));
    HW_CHECK(cumf_plan_set_gather_rows(t_plans[b], rows));
  }
  cumf_dist_reduce_t* red = nullptr;
  HW_CHECK(cumf_dist_reduce_create(&red, comm, n, f, theta_batch));

  // factors
  float *d_thetaT = nullptr, *d_XT = nullptr;
  {
    std::vector<float> theta((size_t)n * f);
    cumf_rand_init(theta.data(), (long)theta.size(), 0.2f, 0);
    d_thetaT = to_device(theta);
    HW_CHECK(hipMalloc(reinterpret_cast<void**>(&d_XT), ((size_t)rows * f + 1) * sizeof(float)));
    HW_CHECK(hipMemset(d_XT, 0, ((size_t)rows * f + 1) * sizeof(float)));
  }

  // constants of the data: rating counts, sum r^2, lambda n_v per column over ALL ranks (fused train SSE)
  double* d_scratch = nullptr;
  HW_CHECK(hipMalloc(reinterpret_cast<void**>(&d_scratch), (size_t)(n + 2) * sizeof(double)));
  double sum_r2 = 0;
  for (float v : csr_data) sum_r2 += (double)v * v;
  const double nnz = sum_over_ranks(comm, (double)nnz_l, d_scratch);
  const double nnz_test = sum_over_ranks(comm, (double)nnz_test_l, d_scratch);
  sum_r2 = sum_over_ranks(comm, sum_r2, d_scratch);
  float* d_reg = nullptr;
  {
    std::vector<double> cnt(n);
    for (long v = 0; v < n; ++v) cnt[v] = (double)(csc_indptr[v + 1] - csc_indptr[v]);
    HW_CHECK(hipMemcpy(d_scratch, cnt.data(), (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    HW_CHECK(cumf_comm_all_reduce_f64(comm, d_scratch, n, nullptr));
    HW_CHECK(hipStreamSynchronize(nullptr));
    HW_CHECK(hipMemcpy(cnt.data(), d_scratch, (size_t)n * sizeof(double), hipMemcpyDeviceToHost));
    std::vector<float> reg(n);
    for (long v = 0; v < n; ++v) reg[v] = cnt[v] > 0 ? (float)(lambda * cnt[v]) : -1.0f;  // < 0: no ratings anywhere
    d_reg = to_device(reg);
  }
  int* d_train_row = nullptr;  // built on first use: the direct train SSE of a near-perfect fit
  double* d_terms = d_scratch + n;  // 1 double: the quadratic terms of this rank's systems; + 1: an SSE

  auto slab_sse = [&](const float* val, const int* row, const int* col, long count) {
    double sse = 0;
    HW_CHECK(hipMemset(d_terms + 1, 0, sizeof(double)));
    if (count > 0) HW_CHECK(cumf_sse(val, row, col, d_thetaT, d_XT, count, f, 0, d_terms + 1, nullptr));
    HW_CHECK(cumf_comm_all_reduce_f64(comm, d_terms + 1, 1, nullptr));
    HW_CHECK(hipStreamSynchronize(nullptr));
    HW_CHECK(hipMemcpy(&sse, d_terms + 1, sizeof(double), hipMemcpyDeviceToHost));
    return sse;
  };

  if (rank == 0)
    printf("hugewiki: %d GPU(s), transport %s, m = %lld, n = %ld, f = %d, nnz = %.0f, nnz_test = %.0f, lambda = %f, THETA_BATCH = %d\n",
           world, cumf_comm_transport_name(comm), bounds.back(), n, f, nnz, nnz_test, lambda, theta_batch);
  const auto t0 = std::chrono::steady_clock::now();
  float final_test = NAN;
  for (int it = 0; it < iters; ++it) {
    // X phase: this rank's slab from the replicated Theta (hugewiki.cu:2436-2602), no exchange
    HW_CHECK(cumf_als_update_fused(x_plan, d_colidx, d_val, d_thetaT, d_XT, f, lambda, solver_x, cg_x, nullptr));
    // Theta phase (hugewiki.cu:2611-2745) with the train SSE out of the solved systems
    HW_CHECK(hipMemsetAsync(d_terms, 0, sizeof(double), nullptr));
    HW_CHECK(cumf_dist_reduce_update_theta(red, t_plans.data(), d_lc_row, d_lc_val, d_XT, d_thetaT, lambda, solver_t, cg_t,
                                           d_reg, d_terms, nullptr));
    HW_CHECK(cumf_comm_all_reduce_f64(comm, d_terms, 1, nullptr));
    HW_CHECK(hipStreamSynchronize(nullptr));
    double terms = 0;
    HW_CHECK(hipMemcpy(&terms, d_terms, sizeof(double), hipMemcpyDeviceToHost));
    double train_sse = sum_r2 - terms;
    if (!(train_sse >= 1e-3 * sum_r2)) {  // near-perfect fit (or NaN): the identity is cancellation noise -- evaluate directly
      if (!d_train_row) {
        std::vector<int> tr((size_t)nnz_l);
        for (long u = 0; u < rows; ++u)
          for (int k = csr_indptr[u]; k < csr_indptr[u + 1]; ++k) tr[k] = (int)u;
        d_train_row = to_device(tr);
      }
      train_sse = slab_sse(d_val, d_train_row, d_colidx, nnz_l);
    }
    const double test_sse = nnz_test > 0 ? slab_sse(d_test_val, d_test_row, d_test_col, nnz_test_l) : NAN;
    final_test = (float)std::sqrt(test_sse / nnz_test);
    if (rank == 0) {
      printf("--------- Train RMSE in iter %d: %f\n", it, std::sqrt(std::max(train_sse, 0.0) / nnz));
      printf("--------- Test RMSE in iter %d: %f\n", it, final_test);
      fflush(stdout);
    }
  }
  HW_CHECK(hipDeviceSynchronize());
  if (rank == 0)
    printf("\ndoALS takes seconds: %.3f for F = %d on %d GPU(s)\n",
           std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count(), f, world);

  if (const char* dump = getenv("CUMF_ALS_DUMP_MODEL")) {  // thetaT from rank 0, every rank its X slab
    std::vector<float> x((size_t)rows * f), th;
    HW_CHECK(hipMemcpy(x.data(), d_XT, x.size() * sizeof(float), hipMemcpyDeviceToHost));
    FILE* fp = fopen((std::string(dump) + "/XT.data" + g).c_str(), "wb");
    if (fp) fwrite(x.data(), sizeof(float), x.size(), fp), fclose(fp);
    if (rank == 0) {
      th.resize((size_t)n * f);
      HW_CHECK(hipMemcpy(th.data(), d_thetaT, th.size() * sizeof(float), hipMemcpyDeviceToHost));
      fp = fopen((std::string(dump) + "/thetaT.data").c_str(), "wb");
      if (fp) fwrite(th.data(), sizeof(float), th.size(), fp), fclose(fp);
    }
  }

  HW_CHECK(cumf_dist_reduce_destroy(red));
  for (auto* p : t_plans) cumf_plan_destroy(p);
  cumf_plan_destroy(x_plan);
  cumf_release_scratch();
  HW_CHECK(cumf_comm_destroy(comm));
  for (void* p : {(void*)d_colidx, (void*)d_val, (void*)d_lc_row, (void*)d_lc_val, (void*)d_test_row, (void*)d_test_col,
                  (void*)d_test_val, (void*)d_thetaT, (void*)d_XT, (void*)d_scratch, (void*)d_reg, (void*)d_train_row})
    if (p) (void)hipFree(p);
  if (rank == 0) printf("\nALS Done.\n");
  return 0;
}

void usage() {
  printf("Usage: ./hugewiki [--gpus G] SPLIT_DIR N F lambda ITERS THETA_BATCH\n");
  printf("SPLIT_DIR: per-GPU slab files of `python -m cumf_als_amd.convert split ... --gpus G`.\n");
  printf("E.g.: ./hugewiki --gpus 8 ./data/hugewiki_split8/ 39780 100 0.05 10 3\n");
}

}  // namespace

int main(int argc, char** argv) {
  int gpus = 0, a = 1;
  if (argc >= 3 && !strcmp(argv[1], "--gpus")) {
    gpus = atoi(argv[2]);
    a = 3;
  }
  if (argc - a != 6) {
    usage();
    return 0;
  }
  const std::string dir(argv[a]);
  const long n = atol(argv[a + 1]);
  const int f = atoi(argv[a + 2]);
  const float lambda = (float)atof(argv[a + 3]);
  const int iters = atoi(argv[a + 4]), theta_batch = atoi(argv[a + 5]);

  if (gpus > 1) {
    // one process per GPU, forked before anything touches the HIP runtime
    char id_file[] = "/tmp/cumf_dist_id.XXXXXX";
    const int fd = mkstemp(id_file);
    if (fd >= 0) close(fd);
    unlink(id_file);
    setenv("CUMF_DIST_ID_FILE", id_file, 1);
    std::vector<pid_t> kids;
    for (int r = 0; r < gpus; ++r) {
      const pid_t pid = fork();
      if (pid == 0) return run_rank(r, gpus, r, dir, n, f, lambda, iters, theta_batch);
      if (pid < 0) {
        perror("fork");
        return EXIT_FAILURE;
      }
      kids.push_back(pid);
    }
    int rc = 0;
    for (pid_t pid : kids) {
      int st = 0;
      waitpid(pid, &st, 0);
      if (!WIFEXITED(st) || WEXITSTATUS(st) != 0) rc = EXIT_FAILURE;
    }
    unlink(id_file);
    return rc;
  }
  const int world = env_int("WORLD_SIZE", 1), rank = env_int("RANK", 0), local = env_int("LOCAL_RANK", rank);
  return run_rank(rank, world, local, dir, n, f, lambda, iters, theta_batch);
}
