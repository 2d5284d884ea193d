// f32 MFMA fused-reduce kernel for the off-diagonal pairs (placeholder until the tiled kernel lands).
#include <hip/hip_runtime.h>
#include "mm_common.h"
extern "C" int mm_mfma_supported(int d) { (void)d; return 0; }
int mm_launch_qred_mfma(const char*, const MMModelLayout&, char*, const MMWorkspaceLayout&,
                        int, int, int, int*, hipStream_t) { return MM_E_DIM; }
