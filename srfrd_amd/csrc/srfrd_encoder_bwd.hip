// Fused SRFRD encoder BACKWARD for MI355X (gfx950): per sequence, walks the blocks in reverse with the working set in
// LDS - LayerNorms recomputed from the forward's checkpoints, q / k / v, the attention probabilities (sign-coded with
// their dropout mask), the attention output and the FFN activation read back from them - back-propagates on the fp32
// matrix cores, scatters item-row gradients with float atomics and accumulates every dense-parameter gradient in a
// per-workgroup slab.  Replaces the autograd pass behind `loss.backward()` (reference trainer.py:40).
#include "srfrd_enc_common.h"

#include "srfrd_encoder_bwd_kernel.inc"

namespace srfrd {
}  // namespace srfrd

using namespace srfrd;

extern "C" int srfrd_long_launch_bwd(const void* args, int grid, int threads, int variant, void* stream);   // srfrd_encoder_bwd_long.hip
extern "C" int srfrd_bwd_slots_launch(const void* args, int grid, int L, int kind_variant, void* stream);    // srfrd_encoder_bwd_slots.hip
extern "C" int srfrd_bwd_chunks_launch(const void* args, int grid, int kind_variant, void* stream);          // srfrd_encoder_bwd_chunks.hip
extern "C" int srfrd_bwd_ragged_launch(const void* args, int grid, int kind_variant, void* stream);          // srfrd_encoder_bwd_ragged.hip

// kind_variant of the slot-placed / row-chunked kernels (0 SASRec 50 + 0, 1 SRFR 45 + 5, 2 SRFRN 45 + 5, 3 SRFU_* 50 + 0), or -1
static int slots_variant(const srfrd_layout* lay) {
  if (lay->D != 50 || lay->n_heads != 1) return -1;
  if (lay->kind == SRFRD_SASREC) return 0;
  if (lay->kind == SRFRD_SRFR && lay->d_item == 45) return 1;
  if (lay->kind == SRFRD_SRFRN && lay->d_item == 45) return 2;
  if (lay->kind >= SRFRD_SRFU_B && lay->d_item == 50) return 3;
  return -1;
}

// The reference's default geometry (seq_len 50, hidden 50: BASELINE configs[1] / [2]) runs the slot-placed backward with its
// six [52][54] slots = 76 KB of LDS: TWO workgroups per CU, where the first-generation kernel's ten matrices (150 KB) allow
// one.  A function of the shape only (never of the environment switches below: callers size `grad_slabs` with it).
static bool two_per_cu(const srfrd_layout* lay, int L) { return L == 50 && slots_variant(lay) >= 0; }

extern "C" int srfrd_bwd_grid(const srfrd_layout* lay, int B, int L) {
  if (!lay || B <= 0 || L <= 0) return SRFRD_E_ARG;
  const int wgs = num_cu() * (two_per_cu(lay, L) ? 2 : 1);
  return B < wgs ? B : wgs;
}

static int encoder_bwd_impl(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                            const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                            const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                            double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                            const float* hidden, const float* pos_logits, const float* neg_logits,
                            const float* save_x, const float* save_h1, const float* save_aux, const float* d_hidden,
                            const float* d_pos,
                            const float* d_neg, int fused_bce, float* grad_table, float* table_contrib, float* grad_slabs,
                            float* scratch, int64_t scratch_floats, float* dbg, int dbg_seq, const int32_t* sched, int sched_mode,
                            void* stream) {
  EncArgs a = {};
  a.sched = sched_mode != 0 ? sched : nullptr;
  a.sched_mode = a.sched ? sched_mode : 0;
  a.ragged_off = getenv("SRFRD_RAGGED_FULL_ROWS") != nullptr;
  { const char* e = getenv("SRFRD_LONG_PRIO"); a.long_prio = e ? atoi(e) : 0; }
  int rc = fill_args(a, lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L,
                     dropout_p, seed, seed_dev, seq_index0);
  if (rc) return rc;
  if (!hidden || !save_x || !save_h1 || !save_aux || !grad_table || !grad_slabs) return SRFRD_E_ARG;
  if (fused_bce && !(pos_ids && neg_ids && pos_logits && neg_logits)) return SRFRD_E_ARG;
  a.c_hidden = hidden; a.c_pl = pos_logits; a.c_nl = neg_logits; a.c_save_x = save_x; a.c_save_h1 = save_h1; a.c_save_aux = save_aux;
  a.d_hidden = d_hidden; a.d_pos = d_pos; a.d_neg = d_neg; a.fused_bce = fused_bce;
  a.grad_table = grad_table; a.grad_slabs = grad_slabs; a.contrib = table_contrib;
  a.dbg = dbg; a.dbg_seq = dbg_seq;
  srfrd_debug_shape(lay, L, &a.dbg_slot, nullptr);
  const Geom g = make_geom(L, lay->D);
  const int64_t lds = bwd_lds_floats(g, lay->n_blocks) * 4;
  const int grid = srfrd_bwd_grid(lay, B, L);
#ifdef SRFRD_STAMPS
  const bool taps = false;                     // (diagnostic build: `dbg` receives the phase stamps)
#else
  const bool taps = dbg != nullptr;
#endif
  if (!taps && ragged_pair(lay, L)) {
    // seq_len 50, hidden 50: the ragged pair (the forward wrote checkpoints for the rows of the computed tiles only)
    rc = srfrd_bwd_ragged_launch(&a, grid, ragged_variant(lay), stream);
    if (rc != SRFRD_E_UNSUPPORTED) return rc;
  }
  if (two_per_cu(lay, L) && !taps && getenv("SRFRD_NO_SLOTS50") == nullptr && getenv("SRFRD_GENERIC") == nullptr &&
      getenv("SRFRD_NO_LSPEC") == nullptr) {
    // seq_len 50 (fused training step or autograd backward): the slot-placed kernel, two workgroups per CU.  (The switches
    // select the first-generation kernel on the same grid - one sequence per workgroup, half of them resident at a time.)
    rc = srfrd_bwd_slots_launch(&a, grid, L, slots_variant(lay), stream);
    if (rc != SRFRD_E_UNSUPPORTED) return rc;
  }
  if (lds > kLdsLimit && !taps && lay->D == 50 && lay->n_heads == 1 &&
      getenv("SRFRD_NO_SLOTS") == nullptr && getenv("SRFRD_GENERIC") == nullptr) {
    // a long sequence (fused training step or autograd backward): the slot-placed, query-chunked LDS-resident kernel where
    // one is built, the row-chunked one otherwise
    const int kv = slots_variant(lay);
    if (kv >= 0) {
      rc = getenv("SRFRD_NO_SLOT_KERNEL") ? SRFRD_E_UNSUPPORTED : srfrd_bwd_slots_launch(&a, grid, L, kv, stream);
      if (rc != SRFRD_E_UNSUPPORTED) return rc;
      if (getenv("SRFRD_NO_CHUNKS") == nullptr && scratch) {     // other lengths up to 208: the row-chunked kernel
        const int64_t stride = (bwd_lds_floats(g, lay->n_blocks) + 2 * kSlack + 63) & ~63ll;
        if (scratch_floats >= stride * grid) {
          a.scratch = scratch;
          a.scratch_stride = stride;
          rc = srfrd_bwd_chunks_launch(&a, grid, kv, stream);
          if (rc != SRFRD_E_UNSUPPORTED) return rc;
        }
      }
    }
  }
  if (lds > kLdsLimit) {                       // long sequence: working set in the caller's global scratch
    const int64_t stride = (bwd_lds_floats(g, lay->n_blocks) + 2 * kSlack + 63) & ~63ll;
    if (!scratch || scratch_floats < stride * grid) return SRFRD_E_UNSUPPORTED;
    a.scratch = scratch;
    a.scratch_stride = stride;
    const bool c4 = lay->kind == SRFRD_SASREC && lay->D == 50 && lay->n_heads == 1 && L == 100 && pos_ids && neg_ids && fused_bce && !d_hidden &&
                    dropout_p > 0.0 && !dbg && getenv("SRFRD_NO_LSPEC") == nullptr && getenv("SRFRD_GENERIC") == nullptr;
    return srfrd_long_launch_bwd(&a, grid, 512, c4 ? 1 : 0, stream);
  }
  const int threads = env_threads("SRFRD_BWD_THREADS", 512);
  const bool spec = getenv("SRFRD_GENERIC") == nullptr && threads == 512 && lay->D == 50 && lay->n_heads == 1;
  if (spec && g.LP == 64 && L == 50 && getenv("SRFRD_NO_LSPEC") == nullptr)
  {
    const bool train = pos_ids && neg_ids && fused_bce && !d_hidden && dropout_p > 0.0 && !dbg && getenv("SRFRD_NO_TSPEC") == nullptr;
    const bool kspec = getenv("SRFRD_NO_KSPEC") == nullptr;
#define SRFRD_LAUNCH(K, DI) (train ? launch_enc(encoder_bwd_kernel<50, 64, 8, 50, K, 1, DI>, grid, threads, lds, stream, a) \
                                  : launch_enc(encoder_bwd_kernel<50, 64, 8, 50, K, 0, DI>, grid, threads, lds, stream, a))
    if (kspec && lay->kind == SRFRD_SASREC) return SRFRD_LAUNCH(SRFRD_SASREC, 50);
    if (kspec && lay->kind >= SRFRD_SRFU_B && lay->d_item == 50) return SRFRD_LAUNCH(-1, 50);
    if (kspec && lay->kind == SRFRD_SRFRN && lay->d_item == 45) return SRFRD_LAUNCH(SRFRD_SRFRN, 45);
    if (kspec && lay->kind == SRFRD_SRFR && lay->d_item == 45) return SRFRD_LAUNCH(SRFRD_SRFR, 45);
#undef SRFRD_LAUNCH
    return launch_enc(encoder_bwd_kernel<50, 64, 8, 50>, grid, threads, lds, stream, a);
  }
  if (spec && g.LP == 64) return launch_enc(encoder_bwd_kernel<50, 64, 8>, grid, threads, lds, stream, a);
  if (spec && g.LP == 32) return launch_enc(encoder_bwd_kernel<50, 32, 8>, grid, threads, lds, stream, a);
  return launch_enc(encoder_bwd_kernel<0, 0, 0>, grid, threads, lds, stream, a);
}

extern "C" int srfrd_encoder_bwd(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                                 const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                                 const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                                 double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                                 const float* hidden, const float* pos_logits, const float* neg_logits,
                                 const float* save_x, const float* save_h1, const float* save_aux, const float* d_hidden,
                                 const float* d_pos,
                                 const float* d_neg, int fused_bce, float* grad_table, float* table_contrib, float* grad_slabs,
                                 float* scratch, int64_t scratch_floats, float* dbg, int dbg_seq, void* stream) {
  return encoder_bwd_impl(lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L, dropout_p,
                          seed, seed_dev, seq_index0, hidden, pos_logits, neg_logits, save_x, save_h1, save_aux, d_hidden, d_pos,
                          d_neg, fused_bce, grad_table, table_contrib, grad_slabs, scratch, scratch_floats, dbg, dbg_seq, nullptr, 0,
                          stream);
}

extern "C" int srfrd_encoder_bwd_sched(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                                       const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                                       const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                                       double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                                       const float* hidden, const float* pos_logits, const float* neg_logits,
                                       const float* save_x, const float* save_h1, const float* save_aux, const float* d_hidden,
                                       const float* d_pos, const float* d_neg, int fused_bce, float* grad_table,
                                       float* table_contrib, float* grad_slabs, float* scratch, int64_t scratch_floats,
                                       const int32_t* sched, int sched_mode, void* stream) {
  if (sched_mode < 0 || sched_mode > 1 || (sched_mode != 0 && !sched)) return SRFRD_E_ARG;
  return encoder_bwd_impl(lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L, dropout_p,
                          seed, seed_dev, seq_index0, hidden, pos_logits, neg_logits, save_x, save_h1, save_aux, d_hidden, d_pos,
                          d_neg, fused_bce, grad_table, table_contrib, grad_slabs, scratch, scratch_floats, nullptr, 0, sched,
                          sched_mode, stream);
}

