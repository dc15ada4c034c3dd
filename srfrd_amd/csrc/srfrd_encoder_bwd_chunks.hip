// Row-chunked fused encoder backward (srfrd_encoder_bwd_chunks_kernel.inc): the LDS-resident backward of the fused training
// step at seq_len 101..208 (BASELINE configs[4] trains at seq_len 200); srfrd_encoder_bwd dispatches here when shape and mode
// qualify and the caller's scratch holds the three [L][50] intermediates per workgroup.
#include "srfrd_enc_common.h"

#include "srfrd_encoder_bwd_chunks_kernel.inc"

#include <cstring>

using namespace srfrd;

// kind_variant: 0 SASRec (50 + 0), 1 SRFR (45 + 5), 2 SRFRN (45 + 5), 3 SRFU_* (50 + 0, kind read at run time).
// `args` carries scratch / scratch_stride.  Returns SRFRD_E_UNSUPPORTED when the shape is outside the kernel's scope.
extern "C" int srfrd_bwd_chunks_launch(const void* args, int grid, int kind_variant, void* stream) {
  EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  if (a.dm.D != 50 || a.dm.n_blocks > SRFRD_MAX_BLOCKS || a.L > 208 || a.L < 17) return SRFRD_E_UNSUPPORTED;
  const int64_t lds = chunks_lds_floats(a.L, 50, a.dm.n_blocks) * 4;
  if (lds > kLdsLimit || !a.scratch || a.scratch_stride < chunks_scratch_floats(a.L, 50)) return SRFRD_E_UNSUPPORTED;
  const bool rmw = a.B > grid;
#define SRFRD_CK(K, DI) (rmw ? launch_enc(encoder_bwd_chunks_kernel<50, K, DI, true>, grid, kCkWaves * 64, lds, stream, a) \
                             : launch_enc(encoder_bwd_chunks_kernel<50, K, DI, false>, grid, kCkWaves * 64, lds, stream, a))
  switch (kind_variant) {
    case 0: return SRFRD_CK(SRFRD_SASREC, 50);
    case 1: return SRFRD_CK(SRFRD_SRFR, 45);
    case 2: return SRFRD_CK(SRFRD_SRFRN, 45);
    case 3: return SRFRD_CK(-1, 50);
  }
#undef SRFRD_CK
  return SRFRD_E_UNSUPPORTED;
}
