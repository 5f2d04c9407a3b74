// Optional HIP-event timing of kernel launches inside the library, per kernel class, on the stream the kernel is
// launched on (bench.py's roofline leg: torch.cuda.Event only sees torch's current stream, and the native Block
// executor issues its launches from C++).  Off by default: a begin/end pair costs two hipEventRecord calls.
#pragma once
#include <hip/hip_runtime.h>

#define CDSEG_PROF_ATTENTION 0
#define CDSEG_PROF_CONV 1       // weight-stationary k = 3 convs of the wide stages (conv.hip: HBM / gather bound)
#define CDSEG_PROF_CONV_DEEP 2  // gathered-GEMM k = 3 convs, C >= 128 (gemm.hip: MFMA / LDS-DMA bound)
#define CDSEG_PROF_CLASSES 3

struct CdsegProfToken {
  hipEvent_t e0;
  int cls;
};

bool cdseg_prof_begin(int cls, hipStream_t s, CdsegProfToken* tok);  // false: profiling off (tok untouched)
void cdseg_prof_end(const CdsegProfToken& tok, hipStream_t s);
