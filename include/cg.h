/*
 * cg.h -- inner operator API of the reference's solver (cg.h:30, cg.cu:682-686),
 * the boundary `hugewiki/hugewiki.cu:2569` links against through `../cg.o`.
 *
 *   void updateXWithCGHost(float* A, float* x, float* b, int batchSize, int f, float cgIter)
 *
 * DEVICE pointers; synchronous (returns after the kernel has finished, like the
 * reference's cudaDeviceSynchronize at cg.cu:686).  Exported with C++ linkage
 * under the reference's mangled name `_Z17updateXWithCGHostPfS_S_iif`.
 *
 *   void updateXWithCGHost_tt_fp16(float* A, float* x, float* b, int batchSize, int f, float cgIter)
 *
 * the same with A stored as fp16 (cg.h:32, cg.cu:235-429, 641-644: the reference passes the half
 * buffer through a float* and casts it back), mangled `_Z25updateXWithCGHost_tt_fp16PfS_S_iif`.
 *
 *   void alsUpdateFeature100Host(int batch_offset, const int* csrRowIndex, const int* csrColIndex, float lambda,
 *                                int m, int F, const float* thetaT, float* XT, float* ythetaT, int cgIter)
 *
 * the reference's fused Gram + CG host (cg.h:34-36, cg.cu:1190-1197; disabled at its call site, als.cu:809-812),
 * mangled `_Z23alsUpdateFeature100HostiPKiS0_fiiPKfPfS3_i`: for the CSR rows batch_offset .. m - 1, the Gram of
 * the row's thetaT columns + lambda * n_row * I, then cgIter warm-started CG steps on A x = ythetaT from x = XT.
 * DEVICE pointers, synchronous; XT / ythetaT are batch-local, F floats per system (the reference kernel's
 * ythetaT stride of blockDim.x = 64, cg.cu:941, is a defect of the disabled path and is not reproduced).  The
 * fused path that also forms the right-hand side is `cumf_als_update_fused` in cumf_als_capi.h.
 */
#ifndef CG_H_
#define CG_H_

void updateXWithCGHost(float* A, float* x, float* b, const int batchSize, const int f, const float cgIter);
void updateXWithCGHost_tt_fp16(float* A, float* x, float* b, const int batchSize, const int f, const float cgIter);
void alsUpdateFeature100Host(const int batch_offset, const int* csrRowIndex, const int* csrColIndex,
                             const float lambda, const int m, const int F, const float* thetaT, float* XT,
                             float* ythetaT, int cgIter);

#endif /* CG_H_ */
