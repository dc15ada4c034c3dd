// Long-sequence build of the fused encoder bwdward pass: the same kernel source with the per-sequence working set
// in a global-memory scratch (SRFRD_BUF_GLOBAL) instead of LDS, for shapes that do not fit 160 KiB.
#define SRFRD_BUF_GLOBAL 1
#include "srfrd_enc_common.h"
#include "srfrd_encoder_bwd_kernel.inc"

#include <cstring>

// args: the caller's srfrd::EncArgs (identical layout) with scratch / scratch_stride filled in
template <class K>
static int launch_long(K kernel, const srfrd_long::EncArgs& a, int grid, int threads, void* stream) {
  static bool s_attr = false;           // one per instantiation
  if (!s_attr) {
    if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, srfrd_long::kLdsLimit) != hipSuccess)
      return SRFRD_E_DEVICE;
    s_attr = true;
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), (size_t)a.lds_floats * 4, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

// variant 1: SASRec, hidden 50, seq_len 100, fused training step (BASELINE configs[3] geometry) - compile-time shape
extern "C" int srfrd_long_launch_bwd(const void* args, int grid, int threads, int variant, void* stream) {
  srfrd_long::EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  // as much of the working set as fits goes to LDS (flat addressing), the rest to the caller's scratch
  a.lds_floats = (srfrd_long::kLdsLimit / 4) - 64;
  const char* mode = getenv("SRFRD_CARVE");
  a.carve_mode = mode ? atoi(mode) : 1;
  if (variant == 1 && threads == 512)
    return launch_long(srfrd_long::encoder_bwd_kernel<50, 112, 8, 100, SRFRD_SASREC, 1, 50>, a, grid, threads, stream);
  return launch_long(srfrd_long::encoder_bwd_kernel<0, 0, 0>, a, grid, threads, stream);
}
