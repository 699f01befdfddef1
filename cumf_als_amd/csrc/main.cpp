// main.cpp -- command-line entry point with the contract of the reference's main.cpp:19-172:
//   ./main M N F NNZ NNZ_TEST lambda X_BATCH THETA_BATCH DATA_DIR
// exactly ten argv, ITERS = 10 on device 0 (main.cpp:16-17), pinned host buffers, factors seeded
// srand(0) / 0.2 * rand() / RAND_MAX with X = 0 (main.cpp:72-78), the ten input files of
// main.cpp:91-103 and the stdout lines that print-test-result.sh:8-11 scrapes.  What the reference
// selects with #defines (solver, fp16 Gram storage, ...) are run-time environment variables here
// (CUMF_ALS_SOLVER, ... see INTEGRATION.md).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <memory>
#include <string>
#include <vector>

#include "als.h"
#include "host_utilities.h"

namespace {

constexpr int kDevice = 0;   // main.cpp:16
constexpr int kIters = 10;   // main.cpp:17

// page-locked host array, released with the owner
struct HostFree {
  void operator()(void* p) const { (void)hipHostFree(p); }
};
template <typename T>
using Pinned = std::unique_ptr<T[], HostFree>;

template <typename T>
Pinned<T> pinned(size_t count) {
  void* p = nullptr;
  const hipError_t e = hipHostMalloc(&p, (count ? count : 1) * sizeof(T), hipHostMallocDefault);
  if (e != hipSuccess) {
    fprintf(stderr, "HIP Error:\nFile = %s\nLine = %d\nReason = %s\n", __FILE__, __LINE__, hipGetErrorString(e));
    exit(EXIT_FAILURE);
  }
  return Pinned<T>(static_cast<T*>(p));
}

void usage() {
  printf("Usage: give M, N, F, NNZ, NNZ_TEST, lambda, X_BATCH, THETA_BATCH and DATA_DIR.\n");
  printf("E.g., for netflix data set, use: \n");
  printf("./main 17770 480189 100 99072112 1408395 0.048 1 3 ./data/netflix/ \n");
  printf("E.g., for movielens 10M data set, use: \n");
  printf("./main 71567 65133 100 9000048 1000006 0.05 1 1 ./data/ml10M/ \n");
  printf("E.g., for yahooMusic data set, use: \n");
  printf("./main 1000990 624961 100 252800275 4003960 1.4 6 3 ./data/yahoo/ \n");
}

// the train matrix three ways + the test triplets, as the ten files of main.cpp:91-103 hold them
struct Dataset {
  Pinned<int> csr_indptr, csr_indices, csc_indices, csc_indptr, coo_row;
  Pinned<float> csr_data, csc_data;
  std::vector<int> test_row, test_col;
  std::vector<float> test_data;

  Dataset(int m, int n, long nnz, long nnz_test)
      : csr_indptr(pinned<int>((size_t)m + 1)), csr_indices(pinned<int>((size_t)nnz)),
        csc_indices(pinned<int>((size_t)nnz)), csc_indptr(pinned<int>((size_t)n + 1)),
        coo_row(pinned<int>((size_t)nnz)), csr_data(pinned<float>((size_t)nnz)),
        csc_data(pinned<float>((size_t)nnz)), test_row((size_t)nnz_test + 1), test_col((size_t)nnz_test + 1),
        test_data((size_t)nnz_test + 1) {}

  void load(const std::string& dir, int m, int n, long nnz, long nnz_test) {
    auto path = [&](const char* name) { return dir + "/" + name; };
    loadCooSparseMatrixBin(path("R_test_coo.data.bin").c_str(), path("R_test_coo.row.bin").c_str(),
                           path("R_test_coo.col.bin").c_str(), test_data.data(), test_row.data(), test_col.data(),
                           nnz_test);
    loadCSRSparseMatrixBin(path("R_train_csr.data.bin").c_str(), path("R_train_csr.indptr.bin").c_str(),
                           path("R_train_csr.indices.bin").c_str(), csr_data.get(), csr_indptr.get(),
                           csr_indices.get(), m, nnz);
    loadCSCSparseMatrixBin(path("R_train_csc.data.bin").c_str(), path("R_train_csc.indices.bin").c_str(),
                           path("R_train_csc.indptr.bin").c_str(), csc_data.get(), csc_indices.get(),
                           csc_indptr.get(), n, nnz);
    loadCooSparseMatrixRowPtrBin(path("R_train_coo.row.bin").c_str(), coo_row.get(), nnz);
  }
};

void dump_model(const char* dir, const float* XT, size_t x_count, const float* thetaT, size_t theta_count) {
  // counterpart of the commented-out dump at main.cpp:149-157, on request (CUMF_ALS_DUMP_MODEL=<dir>)
  const struct {
    const char* name;
    const float* data;
    size_t count;
  } files[] = {{"/XT.data", XT, x_count}, {"/thetaT.data", thetaT, theta_count}};
  for (const auto& f : files) {
    FILE* fp = fopen((std::string(dir) + f.name).c_str(), "wb");
    if (!fp) continue;
    fwrite(f.data, sizeof(float), f.count, fp);
    fclose(fp);
  }
}

}  // namespace

int main(int argc, char** argv) {
  if (argc != 10) {
    usage();
    return 0;
  }
  const int m = atoi(argv[1]), n = atoi(argv[2]), f = atoi(argv[3]);
  const long nnz = atol(argv[4]), nnz_test = atol(argv[5]);
  const float lambda = (float)atof(argv[6]);
  const int x_batch = atoi(argv[7]), theta_batch = atoi(argv[8]);
  const std::string data_dir(argv[9]);
  // main.cpp:33 insists on f % 10 == 0 (the 10 x 10 register tile); the matrix-core tiling only needs
  // an even f, which also admits the f = 64 configuration.
  if (f % T10 != 0 && f % 16 != 0) {
    printf("F has to be a multiple of %d (or of 16)\n", T10);
    return 0;
  }
  printf("M = %d, N = %d, F = %d, NNZ = %ld, NNZ_TEST = %ld, lambda = %f\nX_BATCH = %d, THETA_BATCH = %d\nDATA_DIR = %s \n",
         m, n, f, nnz, nnz_test, lambda, x_batch, theta_batch, data_dir.c_str());
  if (hipSetDevice(kDevice) != hipSuccess) {
    fprintf(stderr, "no HIP device %d\n", kDevice);
    return EXIT_FAILURE;
  }

  // factors (main.cpp:72-78: theta from libc rand() after srand(0), X zero)
  const size_t theta_count = (size_t)n * f, x_count = (size_t)m * f;
  Pinned<float> thetaT = pinned<float>(theta_count), XT = pinned<float>(x_count);
  srand(0u);
  for (size_t k = 0; k < theta_count; ++k) thetaT[k] = 0.2 * ((float)rand() / (float)RAND_MAX);
  for (size_t k = 0; k < x_count; ++k) XT[k] = 0;

  printf("*******start loading training and testing sets to host.\n");
  Dataset d(m, n, nnz, nnz_test);
  d.load(data_dir, m, n, nnz, nnz_test);

  const double t0 = seconds();
  doALS(d.csr_indptr.get(), d.csr_indices.get(), d.csr_data.get(), d.csc_indices.get(), d.csc_indptr.get(),
        d.csc_data.get(), d.coo_row.get(), thetaT.get(), XT.get(), d.test_row.data(), d.test_col.data(),
        d.test_data.data(), m, n, f, nnz, nnz_test, lambda, kIters, x_batch, theta_batch, kDevice);
  printf("\ndoALS takes seconds: %.3f for F = %d\n", seconds() - t0, f);

  if (const char* dir = getenv("CUMF_ALS_DUMP_MODEL")) dump_model(dir, XT.get(), x_count, thetaT.get(), theta_count);
  printf("\nALS Done.\n");
  return 0;
}
