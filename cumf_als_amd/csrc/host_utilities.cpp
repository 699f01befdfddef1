// host_utilities.cpp -- raw binary loaders of the reference's on-disk format
// (same four entry points as the reference's host_utilities.h:31-40).
//
// Each file is a headerless little-endian array of 4-byte elements
// (host_utilities.cpp:33-35).  The reference prints "Unable to open file!" and returns
// with the buffers untouched on a missing file (host_utilities.cpp:27-31) and ignores
// short reads; these loaders stop the process with a message instead, so a bad
// DATA_DIR cannot silently produce garbage factors.
#include "host_utilities.h"

#include <cstdio>
#include <cstdlib>

namespace {

void read_exact(const char* path, void* dst, long count) {
  FILE* fp = fopen(path, "rb");
  if (!fp) {
    fprintf(stderr, "Unable to open file! %s\n", path);
    exit(EXIT_FAILURE);
  }
  const size_t got = count > 0 ? fread(dst, 4, (size_t)count, fp) : 0;
  fclose(fp);
  if ((long)got != count) {
    fprintf(stderr, "Short read: %s holds %zu 4-byte elements, expected %ld\n", path, got, count);
    exit(EXIT_FAILURE);
  }
}

}  // namespace

void loadCSRSparseMatrixBin(const char* dataFile, const char* rowFile, const char* colFile, float* data, int* row,
                            int* col, const int m, const long nnz) {
  read_exact(rowFile, row, (long)m + 1);  // indptr  (host_utilities.cpp:33)
  read_exact(colFile, col, nnz);          // indices (host_utilities.cpp:34)
  read_exact(dataFile, data, nnz);        // data    (host_utilities.cpp:35)
}

void loadCSCSparseMatrixBin(const char* dataFile, const char* rowFile, const char* colFile, float* data, int* row,
                            int* col, const int n, const long nnz) {
  read_exact(rowFile, row, nnz);          // row ids (host_utilities.cpp:57)
  read_exact(colFile, col, (long)n + 1);  // indptr  (host_utilities.cpp:58)
  read_exact(dataFile, data, nnz);
}

void loadCooSparseMatrixRowPtrBin(const char* rowFile, int* row, const long nnz) { read_exact(rowFile, row, nnz); }

void loadCooSparseMatrixBin(const char* dataFile, const char* rowFile, const char* colFile, float* data, int* row,
                            int* col, const long nnz) {
  read_exact(rowFile, row, nnz);
  read_exact(colFile, col, nnz);
  read_exact(dataFile, data, nnz);
}
