// Long-sequence build of the fused encoder bwdward pass: the same kernel source with the per-sequence working set
// in a global-memory scratch (SRFRD_BUF_GLOBAL) instead of LDS, for shapes that do not fit 160 KiB.
#define SRFRD_BUF_GLOBAL 1
#include "srfrd_enc_common.h"
#include "srfrd_encoder_bwd_kernel.inc"

#include <cstring>

// args: the caller's srfrd::EncArgs (identical layout) with scratch / scratch_stride filled in
extern "C" int srfrd_long_launch_bwd(const void* args, int grid, int threads, void* stream) {
  srfrd_long::EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  hipLaunchKernelGGL((srfrd_long::encoder_bwd_kernel<0, 0, 0>), dim3(grid), dim3(threads), 0, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
