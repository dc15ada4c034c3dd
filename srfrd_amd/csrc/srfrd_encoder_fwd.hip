// Fused SRFRD encoder FORWARD for MI355X (gfx950).
//
// One persistent workgroup walks sequences b = blockIdx.x, blockIdx.x + gridDim.x, ...; the whole per-sequence
// working set (embedded inputs, LN outputs, Q/K/V, the L x L scores, FFN hidden) stays in LDS between phases, so HBM
// sees only ids, the gathered embedding rows, the outputs and (training) the per-block checkpoints.  Reference
// arithmetic being reproduced: SURVEY.md 3.4 / reference SRFR_model.py:92-142 (SRFR), :192-239 (SRFRN), :473-530
// (SRFU_*), :621-666 (SASRec); torch multi_head_attention_forward explicit path (q from LN(x), k = v from x,
// q * sqrt(1/d_h), additive causal -inf mask, softmax, dropout on P, P v, out_proj); residuals on the LayerNormed
// tensors; eps = 1e-8.
#include "srfrd_enc_common.h"

#include "srfrd_encoder_fwd_kernel.inc"

namespace srfrd {
// ================================================================================================
// weight packing: canonical (N, K) row-major weights -> MFMA B-fragment order, both product forms
// ================================================================================================
__global__ void __launch_bounds__(256) pack_weights_kernel(const srfrd_layout ly, const float* __restrict__ dense,
                                                          float* __restrict__ packed, uint32_t* state, double lr, double b1,
                                                          double b2) {
  // the last launch of a fused train step also advances the optimizer state for the NEXT step (saves a launch)
  if (state != nullptr && blockIdx.x == 0 && threadIdx.x == 0) step_advance(state, lr, b1, b2);
  const int mf = blockIdx.x, mat = mf >> 1, form = mf & 1;
  const int nb6 = ly.n_blocks * 6;
  const float* W;
  int N, K;
  if (mat < nb6) {
    const srfrd_block_off o = ly.blk[mat / 6];
    const int m = mat % 6, D = ly.D;
    W = dense + (m < 3 ? o.in_w + (int64_t)m * D * D : m == 3 ? o.out_w : m == 4 ? o.c1_w : o.c2_w);
    N = K = D;
  } else {
    if (ly.off_lc_w < 0) return;
    W = dense + ly.off_lc_w;
    N = ly.d_item;
    K = ly.D;
  }
  for (int idx = threadIdx.x; idx < kPackFloats; idx += blockDim.x) {
    const int s = idx & 3, lane = (idx >> 2) & 63, kc = (idx >> 8) & 3, nt = idx >> 10;
    const int k = kc * 16 + 4 * s + (lane >> 4), n = nt * 16 + (lane & 15);
    float v;
    if (form == 0) v = (n < N && k < K) ? W[n * K + k] : 0.f;      // B(k, n) = W[n][k]   (x W^T)
    else v = (k < N && n < K) ? W[k * K + n] : 0.f;                // B(k, n) = W[k][n]   (dy W)
    packed[(int64_t)mf * kPackFloats + idx] = v;
  }
}

}  // namespace srfrd

using namespace srfrd;

extern "C" int srfrd_long_launch_fwd(const void* args, int grid, int threads, void* stream);   // srfrd_encoder_fwd_long.hip
extern "C" int srfrd_fwd_rows_launch(const void* args, int kind_variant, int mode, void* stream);   // srfrd_encoder_fwd_rows.hip
extern "C" int srfrd_fwd_ragged_launch(const void* args, int kind_variant, int train, int grid, void* stream);   // srfrd_encoder_fwd_ragged.hip

extern "C" int64_t srfrd_aux_floats(const srfrd_layout* lay, int B, int L) {
  if (!lay || B <= 0 || L <= 0) return SRFRD_E_ARG;
  const Geom g = make_geom(L, lay->D);
  return (int64_t)lay->n_blocks * B * aux_seq_floats(L, g.LP, lay->D, lay->n_heads);
}

extern "C" int srfrd_scratch_floats(const srfrd_layout* lay, int B, int L, int64_t* fwd_floats, int64_t* bwd_floats) {
  if (!lay || B <= 0 || L <= 0) return SRFRD_E_ARG;
  const Geom g = make_geom(L, lay->D);
  const int64_t f = fwd_lds_floats(g, lay->n_blocks), bw = bwd_lds_floats(g, lay->n_blocks);
  int gf = num_cu();
  if (gf > B) gf = B;
  if (fwd_floats) *fwd_floats = f * 4 <= kLdsLimit ? 0 : ((f + 2 * kSlack + 63) & ~63ll) * gf;
  if (bwd_floats) *bwd_floats = bw * 4 <= kLdsLimit ? 0 : ((bw + 2 * kSlack + 63) & ~63ll) * srfrd_bwd_grid(lay, B, L);
  return 0;
}

extern "C" int srfrd_lds_bytes(const srfrd_layout* lay, int L, int64_t* fwd_bytes, int64_t* bwd_bytes) {
  if (!lay || L <= 0) return SRFRD_E_ARG;
  const Geom g = make_geom(L, lay->D);
  const int64_t f = fwd_lds_floats(g, lay->n_blocks) * 4, bw = bwd_lds_floats(g, lay->n_blocks) * 4;
  if (fwd_bytes) *fwd_bytes = f <= kLdsLimit ? f : 0;
  if (bwd_bytes) *bwd_bytes = bw <= kLdsLimit ? bw : 0;
  return 0;
}

extern "C" int64_t srfrd_packed_floats(const srfrd_layout* lay) {
  if (!lay) return 0;
  return (int64_t)(lay->n_blocks * 6 + 1) * 2 * kPackFloats;
}

extern "C" int srfrd_pack_weights(const srfrd_layout* lay, const float* dense, float* packed, uint32_t* state, double lr,
                                  double beta1, double beta2, void* stream) {
  if (!lay || !dense || !packed) return SRFRD_E_ARG;
  if (lay->D > SRFRD_MAX_D) return SRFRD_E_UNSUPPORTED;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((lay->n_blocks * 6 + 1) * 2), dim3(256), 0, (hipStream_t)stream, *lay, dense,
                     packed, state, lr, beta1, beta2);
  return (int)hipGetLastError();
}

extern "C" int srfrd_debug_shape(const srfrd_layout* lay, int L, int64_t* slot_floats, int32_t* n_slots) {
  if (!lay || L <= 0) return SRFRD_E_ARG;
  if (slot_floats) *slot_floats = (int64_t)L * (L > lay->D ? L : lay->D);
  if (n_slots) *n_slots = 1 + 8 * lay->n_blocks;
  return 0;
}

static int encoder_fwd_impl(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                            const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                            const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                            double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                            float* hidden, float* pos_logits, float* neg_logits, float* save_x, float* save_h1,
                            float* save_aux, float* loss_part, float* scratch, int64_t scratch_floats, float* dbg, int dbg_seq,
                            int last_only, const int32_t* sched, int sched_mode, void* stream) {
  EncArgs a = {};
  a.last_only = last_only;
  a.sched = sched_mode != 0 ? sched : nullptr;
  a.sched_mode = a.sched ? sched_mode : 0;
  a.ragged_off = getenv("SRFRD_RAGGED_FULL_ROWS") != nullptr;
  { const char* e = getenv("SRFRD_LONG_PRIO"); a.long_prio = e ? atoi(e) : 0; }
  int rc = fill_args(a, lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L,
                     dropout_p, seed, seed_dev, seq_index0);
  if (rc) return rc;
  if (!hidden || (pos_ids && !pos_logits) || (neg_ids && !neg_logits)) return SRFRD_E_ARG;
  if (loss_part && !(pos_ids && neg_ids)) return SRFRD_E_ARG;
  a.hidden = hidden; a.pos_logits = pos_logits; a.neg_logits = neg_logits;
  if ((save_x != nullptr) != (save_h1 != nullptr) || (save_x != nullptr) != (save_aux != nullptr)) return SRFRD_E_ARG;
  a.save_x = save_x; a.save_h1 = save_h1; a.save_aux = save_aux; a.loss_part = loss_part;
  a.dbg = dbg; a.dbg_seq = dbg_seq;
  srfrd_debug_shape(lay, L, &a.dbg_slot, nullptr);
  const Geom g = make_geom(L, lay->D);
  const int64_t lds = fwd_lds_floats(g, lay->n_blocks) * 4;
  const bool plain = !pos_ids && !neg_ids && !save_x && !loss_part && dropout_p == 0.0;       // eval-mode hidden states only
  // the row-owner kernel (K / V resident in LDS): every long sequence it covers.  (Measured against the first-generation
  // kernel where both fit: 185 vs 170 us per 512-sequence training forward at seq_len 100, 117 vs 59 us at seq_len 50 - with
  // 7 or 4 row tiles it runs one or two waves per SIMD and their dependent chains are exposed; SRFRD_ROWS_ALWAYS selects it
  // anyway, for tests.)
  const bool rows_wanted = lds > kLdsLimit || getenv("SRFRD_ROWS_ALWAYS") != nullptr;
  if (rows_wanted && !dbg && lay->D == 50 && lay->n_heads == 1 && getenv("SRFRD_NO_ROWS") == nullptr && getenv("SRFRD_GENERIC") == nullptr) {
    int kv = -1;
    if (lay->kind == SRFRD_SASREC) kv = 0;
    else if (lay->kind == SRFRD_SRFR && lay->d_item == 45) kv = 1;
    else if (lay->kind == SRFRD_SRFRN && lay->d_item == 45) kv = 2;
    else if (lay->kind >= SRFRD_SRFU_B && lay->d_item == 50) kv = 3;
    if (kv >= 0) {
      rc = srfrd_fwd_rows_launch(&a, kv, plain ? 0 : 1, stream);
      if (rc != SRFRD_E_UNSUPPORTED) return rc;
    }
  }
  if (lds > kLdsLimit) {                       // long sequence: working set in the caller's global scratch
    int grid = num_cu();
    if (grid > B) grid = B;
    const int64_t stride = (fwd_lds_floats(g, lay->n_blocks) + 2 * kSlack + 63) & ~63ll;
    if (!scratch || scratch_floats < stride * grid) return SRFRD_E_UNSUPPORTED;
    a.scratch = scratch;
    a.scratch_stride = stride;
    return srfrd_long_launch_fwd(&a, grid, 256, stream);
  }
  const int per_cu = (int)(kLdsLimit / lds) > 2 ? 2 : (int)(kLdsLimit / lds);
  int grid = num_cu() * (per_cu < 1 ? 1 : per_cu);
  if (grid > B) grid = B;
  // The reference's default geometry: the ragged kernel (rows of the left-padded sequence only).  A forward that writes
  // training checkpoints takes it exactly when the backward of this (layout, length) will be the ragged one (ragged_pair);
  // debug taps want every row of every intermediate: the full kernel.
#ifdef SRFRD_STAMPS
  const bool taps_f = false;                   // (diagnostic build: `dbg` receives the phase stamps)
#else
  const bool taps_f = dbg != nullptr;
#endif
  if (!taps_f && ragged_pair(lay, L)) {
    const bool train = pos_ids && neg_ids && save_x && loss_part && dropout_p > 0.0 && getenv("SRFRD_NO_TSPEC") == nullptr;
    rc = srfrd_fwd_ragged_launch(&a, ragged_variant(lay), train ? 1 : 0, grid, stream);
    if (rc != SRFRD_E_UNSUPPORTED) return rc;
  }
  // 8 waves per workgroup measured fastest for the forward (95 vs 118 us at C2 with 4 waves)
  const int threads = env_threads("SRFRD_FWD_THREADS", 512);
  // (several attention heads: the generic instantiation only)
  const bool spec = getenv("SRFRD_GENERIC") == nullptr && lay->D == 50 && lay->n_heads == 1;
  if (spec && threads == 512 && g.LP == 64 && L == 50 && getenv("SRFRD_NO_LSPEC") == nullptr)
  {
    const bool train = pos_ids && neg_ids && save_x && loss_part && dropout_p > 0.0 && !dbg && getenv("SRFRD_NO_TSPEC") == nullptr;
    const bool kspec = getenv("SRFRD_NO_KSPEC") == nullptr;
#define SRFRD_LAUNCH(K, DI) (train ? launch_enc(encoder_fwd_kernel<50, 64, 8, 50, K, 1, DI>, grid, threads, lds, stream, a) \
                                  : launch_enc(encoder_fwd_kernel<50, 64, 8, 50, K, 0, DI>, grid, threads, lds, stream, a))
    if (kspec && lay->kind == SRFRD_SASREC) return SRFRD_LAUNCH(SRFRD_SASREC, 50);
    if (kspec && lay->kind >= SRFRD_SRFU_B && lay->d_item == 50) return SRFRD_LAUNCH(-1, 50);
    if (kspec && lay->kind == SRFRD_SRFRN && lay->d_item == 45) return SRFRD_LAUNCH(SRFRD_SRFRN, 45);
    if (kspec && lay->kind == SRFRD_SRFR && lay->d_item == 45) return SRFRD_LAUNCH(SRFRD_SRFR, 45);
#undef SRFRD_LAUNCH
    return launch_enc(encoder_fwd_kernel<50, 64, 8, 50>, grid, threads, lds, stream, a);
  }
  if (spec && threads == 512 && L == 100 && lay->kind == SRFRD_SASREC && getenv("SRFRD_NO_LSPEC") == nullptr) {
    // BASELINE configs[3] geometry (seq_len 100): still LDS-resident in the forward, one workgroup per CU.  The training
    // instantiation runs 16 waves (7 x 4 tiles per weight GEMM: two rounds instead of three and a half; 0.468 -> 0.462 ms
    // per C4 step - the 128-register budget of a 1024-thread workgroup takes back most of what the extra waves give)
    const bool train = pos_ids && neg_ids && save_x && loss_part && dropout_p > 0.0 && !dbg;
    if (train && getenv("SRFRD_FWD_THREADS") == nullptr)
      return launch_enc(encoder_fwd_kernel<50, 112, 16, 100, SRFRD_SASREC, 1, 50>, grid, 1024, lds, stream, a);
    return train ? launch_enc(encoder_fwd_kernel<50, 112, 8, 100, SRFRD_SASREC, 1, 50>, grid, threads, lds, stream, a)
                 : launch_enc(encoder_fwd_kernel<50, 112, 8, 100, SRFRD_SASREC, 0, 50>, grid, threads, lds, stream, a);
  }
  if (spec && threads == 512 && g.LP == 64) return launch_enc(encoder_fwd_kernel<50, 64, 8>, grid, threads, lds, stream, a);
  if (spec && threads == 512 && g.LP == 32) return launch_enc(encoder_fwd_kernel<50, 32, 8>, grid, threads, lds, stream, a);
  if (spec && threads == 256 && g.LP == 64) return launch_enc(encoder_fwd_kernel<50, 64, 4>, grid, threads, lds, stream, a);
  return launch_enc(encoder_fwd_kernel<0, 0, 0>, grid, threads, lds, stream, a);
}

extern "C" int srfrd_encoder_fwd(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                                 const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                                 const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                                 double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                                 float* hidden, float* pos_logits, float* neg_logits, float* save_x, float* save_h1,
                                 float* save_aux, float* loss_part, float* scratch, int64_t scratch_floats, float* dbg, int dbg_seq,
                                 void* stream) {
  return encoder_fwd_impl(lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L, dropout_p,
                          seed, seed_dev, seq_index0, hidden, pos_logits, neg_logits, save_x, save_h1, save_aux, loss_part, scratch,
                          scratch_floats, dbg, dbg_seq, 0, nullptr, 0, stream);
}

extern "C" int srfrd_encoder_fwd_sched(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                                       const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                                       const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                                       double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                                       float* hidden, float* pos_logits, float* neg_logits, float* save_x, float* save_h1,
                                       float* save_aux, float* loss_part, float* scratch, int64_t scratch_floats,
                                       const int32_t* sched, int sched_mode, void* stream) {
  if (sched_mode < 0 || sched_mode > 1 || (sched_mode != 0 && !sched)) return SRFRD_E_ARG;
  return encoder_fwd_impl(lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L, dropout_p,
                          seed, seed_dev, seq_index0, hidden, pos_logits, neg_logits, save_x, save_h1, save_aux, loss_part, scratch,
                          scratch_floats, nullptr, 0, 0, sched, sched_mode, stream);
}

extern "C" int srfrd_encoder_fwd_last(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                                      const int64_t* input_ids, const int64_t* fake_ids, int B, int L, float* hidden_last,
                                      float* scratch, int64_t scratch_floats, void* stream) {
  return encoder_fwd_impl(lay, item_table, dense, packed, input_ids, fake_ids, nullptr, nullptr, nullptr, nullptr, B, L, 0.0, 0,
                          nullptr, 0, hidden_last, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, scratch, scratch_floats,
                          nullptr, 0, 1, nullptr, 0, stream);
}

extern "C" int srfrd_layout_init(srfrd_layout* lay, int kind, int n_items, int max_len, int d_item, int d_fake,
                                 int n_labels, int n_blocks, int n_heads) {
  if (!lay || kind < 0 || kind > SRFRD_SRFU_R || n_items < 1 || max_len < 1 || d_item < 1 || n_blocks < 0 ||
      n_blocks > SRFRD_MAX_BLOCKS || n_heads < 1)
    return SRFRD_E_ARG;
  const bool has_fake = kind == SRFRD_SRFR || kind == SRFRD_SRFRN;
  const bool is_srfu = kind >= SRFRD_SRFU_B;
  if (has_fake && d_fake < 1) return SRFRD_E_ARG;
  if (is_srfu && n_labels < 1) return SRFRD_E_ARG;
  srfrd_layout l = {};
  l.kind = kind; l.n_items = n_items; l.max_len = max_len; l.d_item = d_item;
  l.d_fake = has_fake ? d_fake : 0;
  l.D = d_item + l.d_fake;
  l.d_out = kind == SRFRD_SRFR ? d_item : l.D;
  l.n_labels = is_srfu ? n_labels : 0;
  l.n_blocks = n_blocks; l.n_heads = n_heads;
  if (l.D % n_heads != 0) return SRFRD_E_ARG;
  const int64_t D = l.D;
  int64_t off = 0;
  l.off_pos = off; off += (int64_t)max_len * d_item;
  l.side_rows = has_fake ? 3 : (is_srfu ? n_labels : 0);
  l.side_cols = has_fake ? d_fake : (is_srfu ? l.D : 0);
  l.off_side = off; off += (int64_t)l.side_rows * l.side_cols;
  for (int i = 0; i < n_blocks; ++i) {
    srfrd_block_off& o = l.blk[i];
    o.ln1_w = off; off += D; o.ln1_b = off; off += D;
    o.in_w = off; off += 3 * D * D; o.in_b = off; off += 3 * D;
    o.out_w = off; off += D * D; o.out_b = off; off += D;
    o.ln2_w = off; off += D; o.ln2_b = off; off += D;
    o.c1_w = off; off += D * D; o.c1_b = off; off += D;
    o.c2_w = off; off += D * D; o.c2_b = off; off += D;
  }
  if (kind == SRFRD_SRFR) {
    l.off_lc_w = off; off += (int64_t)d_item * D;
    l.off_lc_b = off; off += d_item;
  } else {
    l.off_lc_w = -1; l.off_lc_b = -1;
  }
  l.off_ll_w = off; off += l.d_out;
  l.off_ll_b = off; off += l.d_out;
  l.n_dense = off;
  l.n_table = (int64_t)(n_items + 1) * d_item;
  *lay = l;
  return 0;
}
