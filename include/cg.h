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
 * The disabled fused kernel host (`alsUpdateFeature100Host`, cg.h:34-36) is not provided; the fused
 * path is `cumf_als_update_fused` in cumf_als_capi.h.
 */
#ifndef CG_H_
#define CG_H_

void updateXWithCGHost(float* A, float* x, float* b, const int batchSize, const int f, const float cgIter);
void updateXWithCGHost_tt_fp16(float* A, float* x, float* b, const int batchSize, const int f, const float cgIter);

#endif /* CG_H_ */
