// Ragged fused encoder forward at the reference's default geometry (srfrd_encoder_fwd_ragged_kernel.inc): the forward of
// reference SRFR_model.py:92-142 / :192-239 / :473-530 / :621-666 computed on the rows a left-padded sequence really has.
// srfrd_encoder_fwd (srfrd_encoder_fwd.hip) dispatches here when shape and mode qualify.  Also home of srfrd_seq_order, the
// per-batch sequence lengths both ragged kernels schedule their sequences by.
#include "srfrd_enc_common.h"

#include "srfrd_encoder_fwd_ragged_kernel.inc"

#include <cstring>

namespace srfrd {

// ---- srfrd_seq_order: first non-pad position of every sequence (what the ragged kernels rank the batch by) ----
// One wave per sequence reads its ids (coalesced) and ballots the first non-zero: one memory round trip, every CU in parallel.
__global__ void __launch_bounds__(256) seq_order_kernel(const int64_t* __restrict__ ids, int B, int L, int G, int* sched) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  if (blockIdx.x == 0 && threadIdx.x == 0) sched[kSchedG] = G;
  const int b = blockIdx.x * 4 + wave;
  if (b >= B) return;
  int t0 = L;
  for (int base = 0; base < L && t0 == L; base += 64) {
    const int t = base + lane;
    const bool nz = t < L && ids[(int64_t)b * L + t] != 0;
    const unsigned long long m = __ballot(nz);
    if (m) t0 = base + (int)__builtin_ctzll(m);
  }
  if (lane == 0) sched[kSchedT0 + b] = t0;
}

}  // namespace srfrd

using namespace srfrd;

extern "C" int64_t srfrd_sched_ints(int B) { return B > 0 ? sched_ints(B) : 0; }

extern "C" int srfrd_seq_order(const int64_t* input_ids, int B, int L, int pair_stride, int32_t* sched, void* stream) {
  if (!input_ids || !sched || B <= 0 || L <= 0 || pair_stride <= 0) return SRFRD_E_ARG;
  hipLaunchKernelGGL(seq_order_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, input_ids, B, L, pair_stride, sched);
  return (int)hipGetLastError();
}

// kind_variant: 0 SASRec (50 + 0), 1 SRFR (45 + 5), 2 SRFRN (45 + 5), 3 SRFU_* (50 + 0, kind read at run time); train: the
// fused-training instantiation (targets, checkpoints, loss sums, dropout all on).  Geometry is fixed: hidden 50, seq_len 50.
extern "C" int srfrd_fwd_ragged_launch(const void* args, int kind_variant, int train, int grid, void* stream) {
  EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  if (a.dm.D != 50 || a.L != 50 || a.dm.n_heads != 1 || a.dm.n_blocks > SRFRD_MAX_BLOCKS ) return SRFRD_E_UNSUPPORTED;
#ifndef SRFRD_STAMPS
  if (a.dbg) return SRFRD_E_UNSUPPORTED;       // debug taps want every row of every intermediate: the full kernels
#endif
  const Geom g = make_geom(50, 50);
  const int64_t lds = fwd_lds_floats(g, a.dm.n_blocks) * 4;
  if (lds > kLdsLimit) return SRFRD_E_UNSUPPORTED;
#define SRFRD_RG(K, DI) (train ? launch_enc(encoder_fwd_ragged_kernel<K, 1, DI>, grid, 512, lds, stream, a) \
                               : launch_enc(encoder_fwd_ragged_kernel<K, 0, DI>, grid, 512, lds, stream, a))
  switch (kind_variant) {
    case 0: return SRFRD_RG(SRFRD_SASREC, 50);
    case 1: return SRFRD_RG(SRFRD_SRFR, 45);
    case 2: return SRFRD_RG(SRFRD_SRFRN, 45);
    case 3: return SRFRD_RG(-1, 50);
  }
#undef SRFRD_RG
  return SRFRD_E_UNSUPPORTED;
}
