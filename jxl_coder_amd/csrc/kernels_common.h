// jxl_coder_amd/csrc/kernels_common.h — shared by the kernels_*.hip translation units.
#pragma once
#include <stdlib.h>
#include <algorithm>
#include "kernels.h"

namespace jxlamd {

struct SyncBlock { __device__ void operator()() const { __syncthreads(); } };

// A frame whose earlier stage raised an error flag is not processed further: its placement data / coefficient offsets may be
// incomplete or stale (no kernel may index with them).  Kernel boundaries order the flag's stores before this load.
__device__ __forceinline__ bool frame_failed(const DevBuffers &B) { return *B.err != 0; }
}  // namespace jxlamd
