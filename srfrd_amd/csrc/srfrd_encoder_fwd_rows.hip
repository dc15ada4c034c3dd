// Row-owner inference forward for long sequences (srfrd_encoder_fwd_rows_kernel.inc): eval-mode hidden states (all
// positions, or the last one for ranking) at seq_len 113..208 - BASELINE configs[4] (seq_len 200) - with K / V resident in
// LDS; srfrd_encoder_fwd / srfrd_encoder_fwd_last dispatch here when shape and mode qualify, every other long-sequence
// case (training, target logits, debug taps) runs the global-scratch build.
#include "srfrd_enc_common.h"

#include "srfrd_encoder_fwd_rows_kernel.inc"

#include <cstring>

using namespace srfrd;

// kind_variant: 0 SASRec (50 + 0), 1 SRFR (45 + 5), 2 SRFRN (45 + 5), 3 SRFU_* (50 + 0, kind read at run time);
// mode 0: eval-mode hidden states only, 1: dropout / checkpoints / target logits / loss sums as requested in `args`.
// Returns SRFRD_E_UNSUPPORTED when the shape is outside the kernel's scope: the caller falls back.
extern "C" int srfrd_fwd_rows_launch(const void* args, int kind_variant, int mode, void* stream) {
  EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  if (a.dm.D != 50 || a.dm.n_blocks > SRFRD_MAX_BLOCKS || a.L > 16 * kRowMaxTiles) return SRFRD_E_UNSUPPORTED;
  const int64_t lds = rows_lds_floats(a.L, 50, a.dm.n_blocks) * 4;
  if (lds > kLdsLimit) return SRFRD_E_UNSUPPORTED;
  int grid = num_cu();
  if (grid > a.B) grid = a.B;
  const int thr = kRowWaves * 64;
#define SRFRD_ROWS(K, DI) (mode ? launch_enc(encoder_fwd_rows_kernel<50, K, DI, 1>, grid, thr, lds, stream, a) \
                               : launch_enc(encoder_fwd_rows_kernel<50, K, DI, 0>, grid, thr, lds, stream, a))
  switch (kind_variant) {
    case 0: return SRFRD_ROWS(SRFRD_SASREC, 50);
    case 1: return SRFRD_ROWS(SRFRD_SRFR, 45);
    case 2: return SRFRD_ROWS(SRFRD_SRFRN, 45);
    case 3: return SRFRD_ROWS(-1, 50);
  }
#undef SRFRD_ROWS
  return SRFRD_E_UNSUPPORTED;
}
