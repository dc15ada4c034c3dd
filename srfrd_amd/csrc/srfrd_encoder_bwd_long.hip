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
  // as much of the working set as fits goes to LDS (flat addressing), the rest to the caller's scratch
  a.lds_floats = (srfrd_long::kLdsLimit / 4) - 64;
  const char* mode = getenv("SRFRD_CARVE");
  a.carve_mode = mode ? atoi(mode) : 1;
  static bool s_attr = false;
  if (!s_attr) {
    if (hipFuncSetAttribute((const void*)srfrd_long::encoder_bwd_kernel<0, 0, 0>, hipFuncAttributeMaxDynamicSharedMemorySize,
                            srfrd_long::kLdsLimit) != hipSuccess)
      return SRFRD_E_DEVICE;
    s_attr = true;
  }
  hipLaunchKernelGGL((srfrd_long::encoder_bwd_kernel<0, 0, 0>), dim3(grid), dim3(threads), (size_t)a.lds_floats * 4,
                     (hipStream_t)stream, a);
  return (int)hipGetLastError();
}
