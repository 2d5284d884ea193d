// f32 MFMA fused-reduce kernels (placeholder until the tiled kernels land).
#include <hip/hip_runtime.h>
#include "mm_common.h"
extern "C" int mm_mfma_supported(int d) { (void)d; return 0; }
int mm_launch_qred_mfma(const char*, const MMModelLayout&, char*, const MMWorkspaceLayout&,
                        int, int, int, int, int*, int*, int*, hipStream_t) { return MM_E_DIM; }
