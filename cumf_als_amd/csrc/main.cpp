// main.cpp -- command-line entry point, same contract as the reference's main.cpp:19-172:
//   ./main M N F NNZ NNZ_TEST lambda X_BATCH THETA_BATCH DATA_DIR
// exactly ten argv, fixed ITERS = 10 and DEVICEID = 0 (main.cpp:16-17), pinned host
// buffers, factor initialisation srand(0) / 0.2*rand()/RAND_MAX / X = 0 (main.cpp:72-78),
// the ten input files of main.cpp:91-103 and the stdout lines that
// print-test-result.sh:8-11 scrapes.  Solver choice etc. are run-time environment
// variables (CUMF_ALS_SOLVER, ... see INTEGRATION.md) instead of #defines.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <string>

#include "als.h"
#include "host_utilities.h"

#define DEVICEID 0
#define ITERS 10

template <typename T>
static T* pinned(size_t count) {
  T* p = nullptr;
  hipError_t e = hipHostMalloc(reinterpret_cast<void**>(&p), (count ? count : 1) * sizeof(T), hipHostMallocDefault);
  if (e != hipSuccess) {
    fprintf(stderr, "HIP Error:\nFile = %s\nLine = %d\nReason = %s\n", __FILE__, __LINE__, hipGetErrorString(e));
    exit(EXIT_FAILURE);
  }
  return p;
}

int main(int argc, char** argv) {
  if (argc != 10) {
    printf("Usage: give M, N, F, NNZ, NNZ_TEST, lambda, X_BATCH, THETA_BATCH and DATA_DIR.\n");
    printf("E.g., for netflix data set, use: \n");
    printf("./main 17770 480189 100 99072112 1408395 0.048 1 3 ./data/netflix/ \n");
    printf("E.g., for movielens 10M data set, use: \n");
    printf("./main 71567 65133 100 9000048 1000006 0.05 1 1 ./data/ml10M/ \n");
    printf("E.g., for yahooMusic data set, use: \n");
    printf("./main 1000990 624961 100 252800275 4003960 1.4 6 3 ./data/yahoo/ \n");
    return 0;
  }
  const int f = atoi(argv[3]);
  // main.cpp:33 insists on f % 10 == 0 (the 10x10 register tile); the matrix-core
  // tiling only needs an even f, which also admits the f = 64 configuration.
  if (f % T10 != 0 && f % 16 != 0) {
    printf("F has to be a multiple of %d (or of 16)\n", T10);
    return 0;
  }
  const int m = atoi(argv[1]);
  const int n = atoi(argv[2]);
  const long nnz = atol(argv[4]);
  const long nnz_test = atol(argv[5]);
  const float lambda = (float)atof(argv[6]);
  const int X_BATCH = atoi(argv[7]);
  const int THETA_BATCH = atoi(argv[8]);
  const std::string DATA_DIR(argv[9]);
  printf("M = %d, N = %d, F = %d, NNZ = %ld, NNZ_TEST = %ld, lambda = %f\nX_BATCH = %d, THETA_BATCH = %d\nDATA_DIR = %s \n",
         m, n, f, nnz, nnz_test, lambda, X_BATCH, THETA_BATCH, DATA_DIR.c_str());

  if (hipSetDevice(DEVICEID) != hipSuccess) {
    fprintf(stderr, "no HIP device %d\n", DEVICEID);
    return EXIT_FAILURE;
  }
  int* csrRowIndexHostPtr = pinned<int>((size_t)m + 1);
  int* csrColIndexHostPtr = pinned<int>((size_t)nnz);
  float* csrValHostPtr = pinned<float>((size_t)nnz);
  float* cscValHostPtr = pinned<float>((size_t)nnz);
  int* cscRowIndexHostPtr = pinned<int>((size_t)nnz);
  int* cscColIndexHostPtr = pinned<int>((size_t)n + 1);
  int* cooRowIndexHostPtr = pinned<int>((size_t)nnz);
  float* thetaTHost = pinned<float>((size_t)n * f);
  float* XTHost = pinned<float>((size_t)m * f);

  // initialise thetaT on host (main.cpp:72-78)
  srand(0u);
  for (long k = 0; k < (long)n * f; k++) thetaTHost[k] = 0.2 * ((float)rand() / (float)RAND_MAX);
  for (long k = 0; k < (long)m * f; k++) XTHost[k] = 0;
  printf("*******start loading training and testing sets to host.\n");
  int* cooRowIndexTestHostPtr = (int*)malloc((size_t)(nnz_test ? nnz_test : 1) * sizeof(int));
  int* cooColIndexTestHostPtr = (int*)malloc((size_t)(nnz_test ? nnz_test : 1) * sizeof(int));
  float* cooValHostTestPtr = (float*)malloc((size_t)(nnz_test ? nnz_test : 1) * sizeof(float));

  loadCooSparseMatrixBin((DATA_DIR + "/R_test_coo.data.bin").c_str(), (DATA_DIR + "/R_test_coo.row.bin").c_str(),
                         (DATA_DIR + "/R_test_coo.col.bin").c_str(), cooValHostTestPtr, cooRowIndexTestHostPtr,
                         cooColIndexTestHostPtr, nnz_test);
  loadCSRSparseMatrixBin((DATA_DIR + "/R_train_csr.data.bin").c_str(), (DATA_DIR + "/R_train_csr.indptr.bin").c_str(),
                         (DATA_DIR + "/R_train_csr.indices.bin").c_str(), csrValHostPtr, csrRowIndexHostPtr,
                         csrColIndexHostPtr, m, nnz);
  loadCSCSparseMatrixBin((DATA_DIR + "/R_train_csc.data.bin").c_str(), (DATA_DIR + "/R_train_csc.indices.bin").c_str(),
                         (DATA_DIR + "/R_train_csc.indptr.bin").c_str(), cscValHostPtr, cscRowIndexHostPtr,
                         cscColIndexHostPtr, n, nnz);
  loadCooSparseMatrixRowPtrBin((DATA_DIR + "/R_train_coo.row.bin").c_str(), cooRowIndexHostPtr, nnz);

  double t0 = seconds();
  doALS(csrRowIndexHostPtr, csrColIndexHostPtr, csrValHostPtr, cscRowIndexHostPtr, cscColIndexHostPtr, cscValHostPtr,
        cooRowIndexHostPtr, thetaTHost, XTHost, cooRowIndexTestHostPtr, cooColIndexTestHostPtr, cooValHostTestPtr, m,
        n, f, nnz, nnz_test, lambda, ITERS, X_BATCH, THETA_BATCH, DEVICEID);
  printf("\ndoALS takes seconds: %.3f for F = %d\n", seconds() - t0, f);

  const char* dump = getenv("CUMF_ALS_DUMP_MODEL");  // counterpart of the commented-out dump at main.cpp:149-157
  if (dump) {
    FILE* xfile = fopen((std::string(dump) + "/XT.data").c_str(), "wb");
    FILE* thetafile = fopen((std::string(dump) + "/thetaT.data").c_str(), "wb");
    if (xfile && thetafile) {
      fwrite(XTHost, sizeof(float), (size_t)m * f, xfile);
      fwrite(thetaTHost, sizeof(float), (size_t)n * f, thetafile);
    }
    if (xfile) fclose(xfile);
    if (thetafile) fclose(thetafile);
  }

  (void)hipHostFree(csrRowIndexHostPtr);
  (void)hipHostFree(csrColIndexHostPtr);
  (void)hipHostFree(csrValHostPtr);
  (void)hipHostFree(cscValHostPtr);
  (void)hipHostFree(cscRowIndexHostPtr);
  (void)hipHostFree(cscColIndexHostPtr);
  (void)hipHostFree(cooRowIndexHostPtr);
  (void)hipHostFree(XTHost);
  (void)hipHostFree(thetaTHost);
  free(cooRowIndexTestHostPtr);
  free(cooColIndexTestHostPtr);
  free(cooValHostTestPtr);
  printf("\nALS Done.\n");
  return 0;
}
