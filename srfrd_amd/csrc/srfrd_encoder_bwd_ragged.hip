// Ragged fused encoder backward at the reference's default geometry (srfrd_encoder_bwd_ragged_kernel.inc): the autograd pass
// behind `loss.backward()` (reference trainer.py:40) computed on the rows a left-padded sequence really has, reading the
// checkpoints of the ragged forward.  srfrd_encoder_bwd (srfrd_encoder_bwd.hip) dispatches here when ragged_pair() holds.
#include "srfrd_enc_common.h"

#include "srfrd_encoder_fwd_ragged_kernel.inc"      // kRagSH, rag_take (the forward kernel template itself is not instantiated here)
#include "srfrd_encoder_bwd_ragged_kernel.inc"

#include <cstring>

using namespace srfrd;

// kind_variant: 0 SASRec (50 + 0), 1 SRFR (45 + 5), 2 SRFRN (45 + 5), 3 SRFU_* (50 + 0, kind read at run time)
extern "C" int srfrd_bwd_ragged_launch(const void* args, int grid, int kind_variant, void* stream) {
  EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  if (a.dm.D != 50 || a.L != 50 || a.dm.n_heads != 1 || a.dm.n_blocks > SRFRD_MAX_BLOCKS ) return SRFRD_E_UNSUPPORTED;
#ifndef SRFRD_STAMPS
  if (a.dbg) return SRFRD_E_UNSUPPORTED;       // debug taps want every row of every intermediate: the full kernels
#endif
  const int64_t lds = bwd_ragged_lds_floats(a.dm.n_blocks) * 4;
  if (lds > kLdsLimit) return SRFRD_E_UNSUPPORTED;
  const bool rmw = a.B > grid;          // some workgroup takes a second sequence: its slab entries are read-modify-written
#define SRFRD_RB(K, DI) (rmw ? launch_enc(encoder_bwd_ragged_kernel<K, DI, true>, grid, 512, lds, stream, a) \
                             : launch_enc(encoder_bwd_ragged_kernel<K, DI, false>, grid, 512, lds, stream, a))
  switch (kind_variant) {
    case 0: return SRFRD_RB(SRFRD_SASREC, 50);
    case 1: return SRFRD_RB(SRFRD_SRFR, 45);
    case 2: return SRFRD_RB(SRFRD_SRFRN, 45);
    case 3: return SRFRD_RB(-1, 50);
  }
#undef SRFRD_RB
  return SRFRD_E_UNSUPPORTED;
}
