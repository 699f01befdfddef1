/*
 * host_utilities.h -- timing + raw binary loaders of the on-disk format
 * (replaces the reference's host_utilities.h:25-40; same names and signatures).
 *
 * Files are headerless little-endian arrays of 4-byte elements
 * (host_utilities.cpp:33-35, 57-59, 71, 90-92).  Unlike the reference, which
 * prints "Unable to open file!" and silently returns (host_utilities.cpp:27-31),
 * these loaders abort with a message on a missing or short file.
 */
#ifndef HOST_UTILITIES_H_
#define HOST_UTILITIES_H_
#include <sys/time.h>

inline double seconds() {
  struct timeval tp;
  gettimeofday(&tp, 0);
  return ((double)tp.tv_sec + (double)tp.tv_usec * 1.e-6);
}

void loadCSRSparseMatrixBin(const char* dataFile, const char* rowFile, const char* colFile,
                            float* data, int* row, int* col, const int m, const long nnz);

void loadCSCSparseMatrixBin(const char* dataFile, const char* rowFile, const char* colFile,
                            float* data, int* row, int* col, const int n, const long nnz);

void loadCooSparseMatrixRowPtrBin(const char* rowFile, int* row, const long nnz);

void loadCooSparseMatrixBin(const char* dataFile, const char* rowFile, const char* colFile,
                            float* data, int* row, int* col, const long nnz);

#endif /* HOST_UTILITIES_H_ */
