/*
 * srfrd_hip.h  --  C ABI of libsrfrd_hip.so, the MI355X (gfx950) implementation of the SRFRD hot path.
 *
 * The reference (oss0430/SRFRD) has no FFI: its hot path is the Python nn.Module API
 * `model(user_ids, input_ids, fake_ids, positive_ids, positive_fake_ids, negative_ids, negative_fake_ids)`
 * (reference SRFR_model.py:92, :192, :473, :651), `model.predict(...)` (:144, :241, :532, :668) and the
 * train step of reference trainer.py:27-41.  Every device operation those calls perform in stock torch
 * ops is replaced by the launchers below; `srfrd_amd/` (Python) keeps the reference's module / state_dict
 * surface on top of them (see INTEGRATION.md for the binding a reference maintainer would add).
 *
 * Conventions
 *  - plain pointers and sizes only; every pointer is a DEVICE pointer unless marked [host];
 *  - the caller owns every buffer; launchers only enqueue work on `stream` (hipStream_t passed as void*) and never
 *    allocate or synchronise => graph-capturable.  The only process-wide state is a mutex-protected cache of per-device
 *    facts (CU count, which kernels already have their > 64 KiB dynamic-LDS opt-in on which device), so launchers may
 *    be called from several host threads and for several devices of one process;
 *  - return value: 0 = ok, < 0 = argument / capability error (SRFRD_E_*), > 0 = hipError_t of the launch;
 *  - ids are int64 (torch LongTensor, reference trainer.py:29), row-major (B, L), left-padded with 0;
 *  - all floating-point data is fp32 (dtype "f32"); integer label / rank work is exact.
 */
#ifndef SRFRD_HIP_H
#define SRFRD_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SRFRD_MAX_BLOCKS 8
#define SRFRD_MAX_D 64          /* hidden width limit of the fused kernels (one lane per channel) */

enum srfrd_kind {               /* reference classes, SRFR_model.py */
  SRFRD_SASREC = 0,             /* :572 */
  SRFRD_SRFR = 1,               /* :53  */
  SRFRD_SRFRN = 2,              /* :154 */
  SRFRD_SRFU_B = 3,             /* :543 */
  SRFRD_SRFU_F = 4,             /* :553 */
  SRFRD_SRFU_R = 5              /* :562 */
};

enum srfrd_error {
  SRFRD_E_ARG = -1,             /* bad argument (null pointer, size out of range) */
  SRFRD_E_UNSUPPORTED = -2,     /* configuration outside what the fused kernels cover (D > 64, L too long for LDS and no scratch given) */
  SRFRD_E_DEVICE = -3           /* not a gfx950 device / LDS attribute could not be set */
};

/* Per-block element offsets into the dense parameter vector (SURVEY Appendix B names in comments). */
typedef struct srfrd_block_off {
  int64_t ln1_w, ln1_b;         /* attention_layernorms.i.{weight,bias}          (D)      */
  int64_t in_w, in_b;           /* attention_layers.i.in_proj_{weight,bias}      (3D,D),(3D) */
  int64_t out_w, out_b;         /* attention_layers.i.out_proj.{weight,bias}     (D,D),(D) */
  int64_t ln2_w, ln2_b;         /* forward_layernorms.i.{weight,bias}            (D)      */
  int64_t c1_w, c1_b;           /* forward_layers.i.conv1.{weight,bias}          (D,D,1),(D) */
  int64_t c2_w, c2_b;           /* forward_layers.i.conv2.{weight,bias}          (D,D,1),(D) */
} srfrd_block_off;

/*
 * Model geometry + the canonical layout of the "dense" parameter vector: every parameter except the item
 * table, concatenated in this order.  The same offsets index the parameters, a dense-gradient slab and the
 * Adam moments, so one descriptor serves forward, backward, reduce and optimizer.
 */
typedef struct srfrd_layout {
  int32_t kind, n_items, max_len, d_item, d_fake, D, d_out, n_labels, n_blocks, n_heads;
  int32_t side_rows, side_cols; /* fake_embed (3,d_fake) | user_label_embed (n_labels,D) | none (0,0) */
  int32_t table_bf16;           /* 0: every `item_table` argument is the fp32 parameter (n_items+1, d_item).  1: it is the
                                   bf16 SHADOW of it (uint16_t, same shape; BASELINE configs[1] / [4] "bf16" table): the
                                   forward / backward / ranking GATHERS read half the bytes; gradients, the fp32 master
                                   and Adam are unchanged (the optimizer launchers rewrite the shadow) */
  int32_t reserved0;
  int64_t off_pos;              /* pos_embed / pos_emb                  (max_len, d_item) */
  int64_t off_side;             /* fake_embed / user_label_embed        (side_rows, side_cols) */
  srfrd_block_off blk[SRFRD_MAX_BLOCKS];
  int64_t off_lc_w, off_lc_b;   /* last_conv.{weight,bias}  (d_item,D,1),(d_item)   SRFR only, else -1 */
  int64_t off_ll_w, off_ll_b;   /* last_layernorm.{weight,bias}         (d_out) */
  int64_t n_dense;              /* elements in the dense vector */
  int64_t n_table;              /* (n_items + 1) * d_item */
} srfrd_layout;

/* [host] fills `lay`; replaces the per-class constructors' shape logic (reference SRFR_model.py:54-90, 155-190,
 * 431-466, 573-615). */
int srfrd_layout_init(srfrd_layout* lay, int kind, int n_items, int max_len, int d_item, int d_fake,
                      int n_labels, int n_blocks, int n_heads);

/* [host] capability query: dynamic LDS bytes the first-generation fused forward / backward kernels need for sequence
 * length L (0 if that working set does not fit the 160 KiB LDS of a gfx950 CU: such shapes run the long-sequence kernels -
 * LDS-resident for hidden width 50 up to L = 208, a global-scratch build beyond - and need srfrd_scratch_floats). */
int srfrd_lds_bytes(const srfrd_layout* lay, int L, int64_t* fwd_bytes, int64_t* bwd_bytes);

/* [host] floats of global scratch the forward / backward need for (B, L): 0 when the first-generation working set fits
 * LDS; otherwise one slice per workgroup: the long-sequence kernels keep what crosses their passes there (the row-chunked
 * backward: three [L][D] intermediates), the global-scratch build its whole working set - every shape runs on the GPU. */
int srfrd_scratch_floats(const srfrd_layout* lay, int B, int L, int64_t* fwd_floats, int64_t* bwd_floats);

/* [host] floats of the forward's `save_aux` checkpoint buffer for (B, L): per block and sequence the FFN hidden
 * activation relu(drop(.)), the attention output P v, the scaled queries, the keys and the values (L, D each) and the
 * attention probabilities (n_heads, L, LP = L rounded up to 16), the latter sign-coded with the attention-dropout mask (a
 * dropped entry is stored negated); sequence-major (all blocks of one sequence are contiguous). */
int64_t srfrd_aux_floats(const srfrd_layout* lay, int B, int L);

/* [host] number of persistent workgroups the backward launches for batch B at sequence length L (= rows of `grad_slabs`):
 * a function of the layout and the shape only. */
int srfrd_bwd_grid(const srfrd_layout* lay, int B, int L);

/* [host] floats per debug-tap slot and number of slots (tests only). */
int srfrd_debug_shape(const srfrd_layout* lay, int L, int64_t* slot_floats, int32_t* n_slots);

/*
 * Weight packing.  The fused kernels read every (out, in) weight matrix of the encoder (in_proj's three slices,
 * out_proj, conv1, conv2, last_conv) from `packed`: a copy pre-swizzled into v_mfma_f32_16x16x4_f32 B-fragment order
 * in both product forms (x W^T for the forward, dy W for the backward).  Call after every parameter update and before
 * srfrd_encoder_fwd / _bwd.  packed holds srfrd_packed_floats(lay) floats.  If `state` != NULL the same launch also
 * performs srfrd_step_begin's advance for the NEXT step (the fused train step ends with this launch), else lr / betas
 * are ignored.
 */
int64_t srfrd_packed_floats(const srfrd_layout* lay);
int srfrd_pack_weights(const srfrd_layout* lay, const float* dense, float* packed,
                       uint32_t* state, double lr, double beta1, double beta2, void* stream);

/*
 * Fused forward: embedding gather (+pos, +fake / user-label channel, pad mask), n_blocks x
 * {LN -> causal self-attention -> +res -> LN -> PW-FFN -> mask}, (last_conv), last LN, pos/neg logits and the
 * masked-BCE partial sums.  Replaces reference SRFR_model.py:92-142 (and the SRFRN / SRFU / SASRec twins)
 * plus the loss terms of reference trainer.py:36-38.
 *
 *  item_table (n_items+1, d_item) fp32 - or its bf16 shadow when lay->table_bf16 -, dense (n_dense) parameters;
 *    packed: srfrd_pack_weights(dense)
 *  input_ids, fake_ids (B,L); fake_ids may be NULL (SASRec ignores it; SRFR/SRFRN treat NULL as all-zero,
 *    reference SRFR_model.py:27-28)
 *  pos_ids/neg_ids (B,L) or NULL (no logits, reference :126-136); pos_fake/neg_fake used by SRFRN only
 *  dropout_p > 0 selects train mode; masks come from the counter hash of srfrd_rng.h keyed by
 *    (seed, site, seq_index0 + b, row, col); if seed_dev != NULL the seed is read from device memory
 *  hidden (B,L,d_out), pos_logits/neg_logits (B,L) outputs (logit pointers may be NULL iff the id pointer is)
 *  save_x (B, n_blocks+1, L, D): block inputs and the last block's output; save_h1 (B, n_blocks, L, D):
 *    post-attention residual; save_aux: srfrd_aux_floats() floats (see there); all three NULL for inference.
 *    Sequence-major and opaque to the caller: written here, read back by srfrd_encoder_bwd
 *  loss_part (B,3) or NULL: per sequence {sum softplus(-pos), sum softplus(neg), count} over pos_ids != 0
 *  scratch / scratch_floats: srfrd_scratch_floats() floats of workspace (NULL / 0 when that is 0)
 *  dbg / dbg_seq: debug taps of one sequence (tests only; NULL otherwise)
 */
int srfrd_encoder_fwd(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                      const int64_t* input_ids, const int64_t* fake_ids,
                      const int64_t* pos_ids, const int64_t* pos_fake,
                      const int64_t* neg_ids, const int64_t* neg_fake,
                      int B, int L, double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                      float* hidden, float* pos_logits, float* neg_logits,
                      float* save_x, float* save_h1, float* save_aux, float* loss_part,
                      float* scratch, int64_t scratch_floats,
                      float* dbg, int dbg_seq, void* stream);

/*
 * Sequence -> workgroup schedule of the seq_len-50 ("ragged") encoder kernels.  At the reference's default geometry
 * (hidden 50, maxlen 50: trainer.py:123-129) srfrd_encoder_fwd / _bwd compute only the rows a left-padded sequence really
 * has, so a launch is as slow as its slowest CU - two workgroups share a CU, and a long sequence should share it with a
 * short one.  srfrd_seq_order (one small launch per batch) writes every sequence's first non-pad position into `sched`
 * (int32[srfrd_sched_ints(B)], caller-owned, valid for THIS batch's input_ids only); the _sched forms of the encoder entry
 * points take it:
 *   sched_mode 0  no schedule (== srfrd_encoder_fwd / _bwd: workgroup x takes sequences x, x + grid, ...)
 *   sched_mode 1  length order: workgroup x takes the sequence of rank perm(x) (longest first, ties by index), perm = the
 *                 `pair_stride` longest first, then the shortest ascending (the workgroup that joins the longest sequence's
 *                 CU brings the shortest), the rest descending; every workgroup selects its sequence from the B lengths
 *                 itself (no sort launch)
 * Results never depend on the schedule beyond the summation order of the per-workgroup dense-gradient slabs, which is a
 * function of the batch.  Other geometries ignore the schedule.
 * Reference counterpart: none (the reference runs stock torch ops over the full padded batch, SRFR_model.py:92-142).
 */
int64_t srfrd_sched_ints(int B);
int srfrd_seq_order(const int64_t* input_ids, int B, int L, int pair_stride, int32_t* sched, void* stream);
int srfrd_encoder_fwd_sched(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                            const int64_t* input_ids, const int64_t* fake_ids,
                            const int64_t* pos_ids, const int64_t* pos_fake,
                            const int64_t* neg_ids, const int64_t* neg_fake,
                            int B, int L, double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                            float* hidden, float* pos_logits, float* neg_logits,
                            float* save_x, float* save_h1, float* save_aux, float* loss_part,
                            float* scratch, int64_t scratch_floats,
                            const int32_t* sched, int sched_mode, void* stream);
int srfrd_encoder_bwd_sched(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                            const int64_t* input_ids, const int64_t* fake_ids,
                            const int64_t* pos_ids, const int64_t* pos_fake,
                            const int64_t* neg_ids, const int64_t* neg_fake,
                            int B, int L, double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                            const float* hidden, const float* pos_logits, const float* neg_logits,
                            const float* save_x, const float* save_h1, const float* save_aux,
                            const float* d_hidden, const float* d_pos, const float* d_neg, int fused_bce,
                            float* grad_table, float* table_contrib, float* grad_slabs,
                            float* scratch, int64_t scratch_floats,
                            const int32_t* sched, int sched_mode, void* stream);

/* Inference forward for ranking: the encoder state of the LAST position only, hidden_last (B, d_out) - what the reference's
 * predict() takes from log2feats (SRFR_model.py:668-681: `log_feats[:, -1, :]`).  Same arithmetic as srfrd_encoder_fwd in
 * eval mode (row L - 1 of its `hidden`, bit for bit); the last block computes queries, attention rows, output projection
 * and FFN for the one 16-row tile holding that position.  Feed it to srfrd_predict_logits / srfrd_logits_topk with L = 1. */
int srfrd_encoder_fwd_last(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                           const int64_t* input_ids, const int64_t* fake_ids, int B, int L, float* hidden_last,
                           float* scratch, int64_t scratch_floats, void* stream);

/*
 * Fused backward of srfrd_encoder_fwd: LayerNorms are recomputed in LDS from save_x / save_h1; q / k / v, the FFN hidden
 * activation, the attention probabilities (with their dropout mask) and the attention output are read back from
 * save_aux.  Replaces the
 * autograd pass behind `loss.backward()` (reference trainer.py:40).
 *
 *  fused_bce != 0: d(pos_logits) = (sigmoid(pos)-1)[pos_ids!=0], d(neg_logits) = sigmoid(neg)[pos_ids!=0]
 *    (SUM reduction: the 1/count of the two means is applied by srfrd_adam_step via `stats`), d_hidden = 0;
 *  fused_bce == 0: upstream gradients d_hidden (B,L,d_out) / d_pos / d_neg (B,L) are read (each may be NULL).
 *  grad_table (n_items+1, d_item): += by float atomics (caller zeroes it; row 0 never written: padding_idx)
 *  table_contrib: NULL, or (3, B, L, d_item) floats for the DETERMINISTIC item-table scatter: instead of the atomics every
 *    (target kind {pos, neg, input}, sequence, position) writes its row contribution here (zeros where it has none) and
 *    srfrd_table_reduce sums the rows of each item in a fixed order into grad_table (bitwise reproducible)
 *  grad_slabs (srfrd_bwd_grid(lay, B, L), n_dense): per-workgroup partial dense gradients (fully overwritten)
 */
int srfrd_encoder_bwd(const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                      const int64_t* input_ids, const int64_t* fake_ids,
                      const int64_t* pos_ids, const int64_t* pos_fake,
                      const int64_t* neg_ids, const int64_t* neg_fake,
                      int B, int L, double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                      const float* hidden, const float* pos_logits, const float* neg_logits,
                      const float* save_x, const float* save_h1, const float* save_aux,
                      const float* d_hidden, const float* d_pos, const float* d_neg, int fused_bce,
                      float* grad_table, float* table_contrib, float* grad_slabs,
                      float* scratch, int64_t scratch_floats,
                      float* dbg, int dbg_seq, void* stream);

/*
 * Deterministic item-table scatter, second half (the first is srfrd_encoder_bwd with table_contrib): n = 3 * B * L keys
 * (the item id each contribution row belongs to: pos ids, neg ids, input ids, in table_contrib's row order) sorted
 * ascending by a STABLE sort, with `order[i]` = the contribution row that sorted position i came from.  One wave per
 * run of equal keys adds that item's rows in sorted order and stores the sum to grad_table[key] (key 0 = padding_idx is
 * skipped; rows without contributions are not touched: the caller keeps them zero).  The summation order is a function
 * of the ids alone => the table gradient is bitwise reproducible, unlike the float-atomic form.
 */
int srfrd_table_reduce(const int64_t* sorted_keys, const int64_t* order, const float* table_contrib, int64_t n, int d_item,
                       float* grad_table, void* stream);

/* Sums the per-workgroup slabs into grad_dense (n_dense) in a fixed order (bitwise reproducible) and, if
 * loss_part != NULL, reduces it into stats[0..3] = {sum softplus(-pos), sum softplus(neg), count, 0}; with loss_out != NULL
 * (single rank: stats are already global) it also writes the loss, making srfrd_loss_finalize unnecessary. */
int srfrd_reduce_dense(const float* grad_slabs, int n_slabs, int64_t n_dense, float* grad_dense,
                       const float* loss_part, int B, float* stats, float* loss_out, void* stream);

/* The loss half of srfrd_reduce_dense on its own: stats[0..3] = {sum softplus(-pos), sum softplus(neg), count, 0} from the
 * forward's loss_part (B,3); loss_out (may be NULL) = the loss of these statistics.  The data-parallel step runs it right
 * after the forward so that the 16-byte all-reduce of the statistics overlaps the backward kernel. */
int srfrd_loss_stats(const float* loss_part, int B, float* stats, float* loss_out, void* stream);

/*
 * Optimizer state advance (one thread): t += 1, step_size = lr / (1 - b1^t), bc2_sqrt = sqrt(1 - b2^t)
 * (double precision, as torch.optim.Adam computes them on the host), next dropout seed.
 *  state: uint32[32] {t, base_seed, step_seed, -, float step_size, float bc2_sqrt, ticket root, -, 16 ticket shards, -...}
 *         (words 6 and 8..23 are the hand-off counters of srfrd_adam_pack_step, zero between launches; callers that
 *         never use that entry point may pass 8 words)
 */
int srfrd_step_begin(uint32_t* state, double lr, double beta1, double beta2, void* stream);

/*
 * Dense Adam over the flat vector [table | dense] (torch.optim.Adam semantics, reference trainer.py:41, :390):
 *   g = grad[i] * gscale;  m = b1 m + (1-b1) g;  v = b2 v + (1-b2) g^2;  p -= step_size * m / (sqrt(v)/bc2_sqrt + eps)
 * gscale = 1 / stats[2] if stats != NULL (mean over non-pad targets, trainer.py:36-38) else 1.
 * The first n_zero gradient elements (the table, accumulated by atomics) are re-zeroed in the same pass.
 * [i0, i1) = the slice this rank updates (sharded optimizer); pass 0, n for all.
 * table_bf16 (may be NULL): bf16 shadow of the first n_table parameters (the item table); stepped elements below n_table
 * are also written there, rounded to nearest even.
 */
int srfrd_adam_step(float* param, float* grad, float* m, float* v, int64_t n, int64_t i0, int64_t i1,
                    int64_t n_zero, double beta1, double beta2, double eps,
                    const uint32_t* state, const float* stats, uint16_t* table_bf16, int64_t n_table, void* stream);

/*
 * Fused tail of the train step: srfrd_adam_step over the whole flat vector [table | pad | dense] (n floats, the dense
 * part starting at n_table_pad), the stepped encoder weights written straight into `packed` in both fragment forms
 * (what srfrd_pack_weights(dense) would produce; `packed` must have been packed once before), and the optimizer-state
 * advance of srfrd_step_begin for the NEXT step, done by the last block to finish; table_bf16 (may be NULL) = the bf16 shadow
 * of the item table (lay->n_table elements), rewritten with the stepped rows in the same pass.  `state` holds 32 words here:
 * state[6] and state[8..23] are the ticket counters of that hand-off (zero between launches).  Replaces optimizer.step() of reference trainer.py:41.
 */
int srfrd_adam_pack_step(const srfrd_layout* lay, float* param, float* grad, float* m, float* v, int64_t n,
                         int64_t n_table_pad, int64_t n_zero, double lr, double beta1, double beta2, double eps,
                         uint32_t* state, const float* stats, float* packed, uint16_t* table_bf16, void* stream);

/* bf16 shadow of an fp32 vector (round to nearest even): out[i] = bf16(src[i]), i < n.  Builds / refreshes the item-table
 * shadow that lay->table_bf16 = 1 launches gather from (the fused optimizer keeps it current by itself). */
int srfrd_table_to_bf16(const float* src, int64_t n, uint16_t* out, void* stream);

/* loss = stats[0]/stats[2] + stats[1]/stats[2] -> loss_out[0] (reference trainer.py:36-38). */
/* The `l2_emb * ||p||_2` terms of reference trainer.py:39 (one Frobenius norm per parameter tensor) for the fused step.
 * srfrd_l2_norms: the n_seg parameter tensors are [seg_off[k], seg_off[k] + seg_len[k]) of `param` (device int64 arrays;
 *   segment 0 = the item table, the others lie at or above n_table_pad); partial: 240 + n_seg floats of workspace;
 *   l2buf[0] = l2_emb / ||table||, l2buf[1] = l2_emb * sum_k ||p_k|| (add it to the loss); dense_scale (n_dense floats,
 *   zero-initialised by the caller once): l2_emb / ||p|| of the tensor each dense element belongs to.
 * srfrd_l2_apply: grad[i] += stats[2] * (l2_emb / ||p||) * param[i] over [i0, i1) (stats[2] = the count the Adam step
 *   divides by; NULL: 1) - enqueue between the gradient exchange and the Adam step of that range. */
int srfrd_l2_norms(const float* param, const int64_t* seg_off, const int64_t* seg_len, int n_seg, int64_t n_table_pad,
                   double l2_emb, float* partial, float* l2buf, float* dense_scale, void* stream);
int srfrd_l2_apply(float* grad, const float* param, int64_t i0, int64_t i1, int64_t n_table_pad, const float* l2buf,
                   const float* dense_scale, const float* stats, void* stream);

int srfrd_loss_finalize(const float* stats, float* loss_out, void* stream);

/* get_Labels (reference SRFR_model.py:546-570) and SRFRN.predict's user label (:244); labels int64 (B).
 * kind = SRFRD_SRFU_B / _F / _R, or SRFRD_SRFRN for the predict label. */
int srfrd_user_labels(int kind, const int64_t* fake_ids, int B, int L, int64_t* labels, void* stream);

/*
 * Id validation.  The reference's nn.Embedding raises IndexError for an id outside its table (SRFR_model.py:10-12, 402-404,
 * 590-591); the fused kernels instead CLAMP every id into range (item ids to [0, n_items], fake ids to [0, 2]) so that no
 * input can read or scatter out of bounds, and this launcher reports the violation: err_word[0] |= 1 if any of the (up to
 * three, each may be NULL) item-id arrays holds an id outside [0, n_items], |= 2 if any fake-id array holds one outside
 * [0, fake_hi].  Each array has n elements.  The host reads the word when it next synchronises (srfrd_amd: check_ids()).
 */
int srfrd_check_ids(const int64_t* item_a, const int64_t* item_b, const int64_t* item_c,
                    const int64_t* fake_a, const int64_t* fake_b, const int64_t* fake_c,
                    int64_t n, int64_t n_items, int64_t fake_hi, uint32_t* err_word, void* stream);

/*
 * predict (reference SRFR_model.py:144-152 and twins): logits[b][i] = <hidden[b, L-1, :], E[cand]> with
 * E = item row (SRFRN: item row || fake_embed[user_label[b]]).  cand is (n_cand) shared by all users
 * (cand_stride = 0) or (B, n_cand) per user (cand_stride = n_cand).  logits (B, n_cand).
 */
int srfrd_predict_logits(const srfrd_layout* lay, const void* item_table, const float* dense,
                         const float* hidden, int B, int L, const int64_t* cand, int n_cand, int64_t cand_stride,
                         const int64_t* user_label, float* logits, void* stream);

/*
 * Full-catalog ranking: top-k items of <hidden[b, L-1, :], E[i]> over i in [item_lo, item_hi) with the logits
 * never written to HBM.  Ties break to the lower item id (stable descending sort).  topk_idx int64 (B,k),
 * topk_val (B,k).  workspace: srfrd_topk_workspace_bytes().  exclude_pad != 0 skips item 0.
 */
int64_t srfrd_topk_workspace_bytes(int B, int k, int64_t n_rows);
int srfrd_logits_topk(const srfrd_layout* lay, const void* item_table, const float* dense,
                      const float* hidden, int B, int L, int64_t item_lo, int64_t item_hi, int exclude_pad,
                      const int64_t* user_label, int k, int64_t* topk_idx, float* topk_val,
                      void* workspace, void* stream);

/*
 * Merge of per-shard top-k lists: the catalog's rows split into shards (one per GPU for the row-sharded table of BASELINE
 * configs[4], or just to bound the ranking workspace), each ranked by srfrd_logits_topk over its [item_lo, item_hi).
 * cand_idx int64 / cand_val (B, n_cand) hold the shards' lists side by side (n_cand = shards * k <= 4096; idx < 0 marks an
 * empty slot); topk_idx / topk_val (B, k) receive the k best in stable descending order (value desc, item id asc): what
 * ONE srfrd_logits_topk over the whole catalog returns, ties across shard boundaries included.
 */
int srfrd_topk_merge(const int64_t* cand_idx, const float* cand_val, int B, int n_cand, int k,
                     int64_t* topk_idx, float* topk_val, void* stream);

/* HR@10 / NDCG@10 inputs (reference utils.py:589-597): rank[b] = #{i >= 1 : logits[b][i] > logits[b][0]};
 * metric_acc[0] += [rank<10] / log2(rank+2), metric_acc[1] += [rank<10], metric_acc[2] += 1 (double[3]). */
int srfrd_eval_rank(const float* logits, int B, int n_cand, int32_t* rank, double* metric_acc, void* stream);

/*
 * Device-side batch sampler with the layout and semantics of reference utils.py:21-57 (sample_function_fr /
 * WarpSampler_fr): per sampled user (uniform among users with > 1 interaction) the most recent `L` training items
 * left-padded with 0, pos[t] = the next item, neg[t] = a uniform item outside the user's history wherever pos[t] != 0,
 * rsq / prs the fake(1)/real(2) ids, nrs = 1 where set.  Histories are CSR over user ids 0..usernum:
 * user_ptr int64 (usernum + 2), items / reviews int32.  out_packed int64 (6, B, L) = [seq, rsq, pos, prs, neg, nrs].
 * Randomness: counter hash of (seed, batch_index, row, position, try) - reproducible, unlike the reference's
 * unseeded worker processes (utils.py:79).
 */
int srfrd_sample_batch(const int64_t* user_ptr, const int32_t* items, const int32_t* reviews, int usernum, int itemnum,
                       int B, int L, uint32_t seed, uint32_t batch_index, int64_t* out_user, int64_t* out_packed,
                       void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SRFRD_HIP_H */
