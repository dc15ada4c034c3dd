// Slot-placed, query-chunked fused encoder backward (srfrd_encoder_bwd_slots_kernel.inc): the LDS-resident backward of
// the fused training step at seq_len 100 (BASELINE configs[3]); srfrd_encoder_bwd (srfrd_encoder_bwd.hip) dispatches
// here when shape and mode qualify, every other long-sequence case runs the global-scratch build.
#include "srfrd_enc_common.h"

#include "srfrd_encoder_bwd_slots_kernel.inc"

#include <cstring>

using namespace srfrd;

// kind_variant: 0 SASRec (50 + 0), 1 SRFR (45 + 5), 2 SRFRN (45 + 5), 3 SRFU_* (50 + 0, kind read at run time).
// Returns SRFRD_E_UNSUPPORTED when no instantiation serves (L, kind_variant): the caller falls back.
extern "C" int srfrd_bwd_slots_launch(const void* args, int grid, int L, int kind_variant, void* stream) {
  EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  if (a.dm.D != 50 || a.dm.n_blocks > SRFRD_MAX_BLOCKS) return SRFRD_E_UNSUPPORTED;
  const int64_t lds = slots_lds_floats(L, 50, a.dm.n_blocks) * 4;
  if (lds > kLdsLimit) return SRFRD_E_UNSUPPORTED;
  const bool rmw = a.B > grid;          // some workgroup takes a second sequence: its slab entries are read-modify-written
#define SRFRD_SL(LL, K, DI) (rmw ? launch_enc(encoder_bwd_slots_kernel<50, LL, K, DI, true>, grid, kSlotWaves * 64, lds, stream, a) \
                                 : launch_enc(encoder_bwd_slots_kernel<50, LL, K, DI, false>, grid, kSlotWaves * 64, lds, stream, a))
  if (L == 100) {
    switch (kind_variant) {
      case 0: return SRFRD_SL(100, SRFRD_SASREC, 50);
      case 1: return SRFRD_SL(100, SRFRD_SRFR, 45);
      case 2: return SRFRD_SL(100, SRFRD_SRFRN, 45);
      case 3: return SRFRD_SL(100, -1, 50);
    }
  }
  if (L == 50) {
    switch (kind_variant) {
      case 0: return SRFRD_SL(50, SRFRD_SASREC, 50);
      case 1: return SRFRD_SL(50, SRFRD_SRFR, 45);
      case 2: return SRFRD_SL(50, SRFRD_SRFRN, 45);
      case 3: return SRFRD_SL(50, -1, 50);
    }
  }
#undef SRFRD_SL
  return SRFRD_E_UNSUPPORTED;
}
