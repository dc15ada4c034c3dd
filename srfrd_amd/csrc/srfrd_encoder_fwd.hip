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

namespace srfrd {

// ================================================================================================
// forward
// ================================================================================================
// D_, LP_, NW_ > 0: geometry and wave count fixed at compile time (strides become immediates, tile loops
// resolve statically); 0: read at run time (the generic instantiation covers every other shape).
template <int D_, int LP_, int NW_>
__global__ void __launch_bounds__(NW_ > 0 ? NW_ * 64 : 512) encoder_fwd_kernel(const EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Dims& ly = a.dm;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = NW_ > 0 ? NW_ : (int)(blockDim.x >> 6), nthr = nw << 6;
  const int L = a.L;
  const int D = D_ > 0 ? D_ : ly.D;
  const int LP = LP_ > 0 ? LP_ : ((L + 15) & ~15);
  const int DK = (D + 3) & ~3, DS = DK + 2, SLD = LP + 2, NT = (D + 15) >> 4, MT = LP >> 4;
  Geom g;
  g.L = L; g.LP = LP; g.D = D; g.DK = DK; g.DS = DS; g.SLD = SLD; g.NT = NT; g.MT = MT;
  const int szA = LP * DS, szX = imax(szA, LP * SLD);
  lds_f* const lds0 = (lds_f*)smem;
  lds_f* bXS = lds0;
  lds_f* bQN = bXS + szX;
  lds_f* bQ = bQN + szA;
  lds_f* bK = bQ + szA;
  lds_f* bV = bK + szA;
  lds_f* tail = bV + szA + kSlack;
  lds_i* s_in = (lds_i*)tail;
  lds_f* s_keep = tail + LP;
  lds_i* s_pid = (lds_i*)(tail + 2 * LP);
  lds_i* s_nid = (lds_i*)(tail + 3 * LP);
  lds_f* s_misc = tail + 4 * LP;          // 64 floats
  lds_f* s_ln = s_misc + 64;              // LayerNorm parameter cache
  {
    const int total = (int)fwd_lds_floats(g, ly.n_blocks);
    for (int i = tid; i < total; i += nthr) lds0[i] = 0.f;
  }
  __syncthreads();
  fill_ln_cache(s_ln, a.dense, ly);

  const float* P = a.dense;
  const float* table = a.table;
  const int kind = ly.kind;
  const bool is_sas = kind == SRFRD_SASREC;
  const bool has_fake = kind == SRFRD_SRFR || kind == SRFRD_SRFRN;
  const bool is_srfu = kind >= SRFRD_SRFU_B;
  const int di = ly.d_item, dfk = ly.d_fake, dout = ly.d_out;
  const float sqrtD = sqrtf((float)di);
  const float qscale = a.qscale;
  const uint32_t seed = a.seed_dev ? *a.seed_dev : a.seed;
  const int B = a.B;
  // packed weight of matrix `mat` (block*6 + {Wq,Wk,Wv,Wo,W1,W2}; n_blocks*6 = last_conv); form 0: x W^T, 1: dy W
  auto pk = [&](int mat, int form) {
    return PackedB{reinterpret_cast<const float4*>(a.packed) + ((int64_t)mat * 2 + form) * (kPackFloats / 4)};
  };

  STAMP_INIT
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int64_t rowbase = (int64_t)b * L;
    const uint32_t seq = (uint32_t)(a.seq0 + b);
    for (int t = tid; t < LP; t += nthr) {
      const int id = t < L ? (int)a.in_ids[rowbase + t] : 0;
      s_in[t] = id;
      s_keep[t] = id != 0 ? 1.f : 0.f;
      s_pid[t] = (t < L && a.pos_ids) ? (int)a.pos_ids[rowbase + t] : 0;
      s_nid[t] = (t < L && a.neg_ids) ? (int)a.neg_ids[rowbase + t] : 0;
    }
    if (is_srfu && wave == 0) {
      const int lab = user_label_wave(kind, a.fk_ids ? a.fk_ids + rowbase : nullptr, L, ly.n_labels);
      if (lane == 0) ((lds_i*)s_misc)[0] = lab;
    }
    __syncthreads();

    // ---- embedding: gather + position (+ side channel) + pad mask          
    {
      const DropSite dsE = drop_site(a.drop_on && is_sas, seed, SITE_EMB, seq, a.drop_thr, a.drop_scale);
      const int lab = is_srfu ? ((lds_i*)s_misc)[0] : 0;
      const int q = tid & 3;
      for (int t = tid >> 2; t < L; t += nthr >> 2) {                  // one DPP quad per position
        const int id = s_in[t];
        const float keep = s_keep[t];
        const int f = (has_fake && a.fk_ids) ? (int)a.fk_ids[rowbase + t] : 0;
#pragma unroll
        for (int j = 0; j < kQC; ++j) {
          const int c = q + 4 * j;
          if (c < D) {
            float v;
            if (has_fake) {
              if (c < di) v = table[(int64_t)id * di + c] + P[ly.off_pos + t * di + c];
              else v = P[ly.off_side + f * dfk + (c - di)];
            } else {
              v = table[(int64_t)id * di + c];
              if (is_sas) v *= sqrtD;
              v += P[ly.off_pos + t * di + c];
              if (is_srfu) v += P[ly.off_side + lab * D + c];
              if (is_sas) v *= drop_mul(dsE, t, c);
            }
            v *= keep;
            bXS[t * DS + c] = v;
            if (a.save_x) a.save_x[(rowbase + t) * D + c] = v;
          }
        }
      }
    }
    __syncthreads();
    tap(a, b, 0, bXS, L, D, DS);

    for (int i = 0; i < ly.n_blocks; ++i) {
      const BlkOff o = blk_off(ly.blk0 + i * ly.blk_stride, D);
      const int tb = 1 + 8 * i;
      launder(bXS); launder(bQN); launder(bQ); launder(bK); launder(bV);
      // weight fragments are requested one phase ahead of the GEMM that consumes them
      const WFrag wq = load_wfrag(pk(i * 6 + 0, 0), P + o.in_b, D, NT);
      const WFrag wk = load_wfrag(pk(i * 6 + 1, 0), P + o.in_b + D, D, NT);
      const WFrag wv = load_wfrag(pk(i * 6 + 2, 0), P + o.in_b + 2 * D, D, NT);
      ln_rows(nw, bXS, bQN, L, DS, D, s_ln + (4 * i + 0) * 64, s_ln + (4 * i + 1) * 64);
      __syncthreads();
      tap(a, b, tb + 0, bQN, L, D, DS);
      // q = (LN(x) Wq^T + bq) * sqrt(1/d_h);  k = x Wk^T + bk;  v = x Wv^T + bv
      gemm_packed(nw, MT, NT, DK, Mat{bQN, DS}, wq, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v * qscale; });
      gemm_packed(nw, MT, NT, DK, Mat{bXS, DS}, wk, [&](int r, int c, float v) { if (c < D) bK[r * DS + c] = v; });
      gemm_packed(nw, MT, NT, DK, Mat{bXS, DS}, wv, [&](int r, int c, float v) { if (c < D) bV[r * DS + c] = v; });
      const WFrag wo = load_wfrag(pk(i * 6 + 3, 0), P + o.out_b, D, NT);
      const WFrag w1 = load_wfrag(pk(i * 6 + 4, 0), P + o.c1_b, D, NT);
      __syncthreads();
      tap(a, b, tb + 1, bQ, L, D, DS);
      tap(a, b, tb + 2, bK, L, D, DS);
      tap(a, b, tb + 3, bV, L, D, DS);
      // S = q k^T on the lower-triangular tiles (x is dead: S overlays it)
      gemm_tiles<1>(nw, MT, MT, DK, Mat{bQ, DS}, MatT{bK, DS}, [&](int r, int c, float v) { bXS[r * SLD + c] = v; });
      __syncthreads();
      // causal softmax (+ attention dropout); keys j > r get exact zeros up to LP
      {
        const DropSite dsA = drop_site(a.drop_on, seed, site_attn(i), seq, a.drop_thr, a.drop_scale);
        softmax_rows<true>(nw, bXS, L, SLD, LP, dsA);
      }
      __syncthreads();
      tap(a, b, tb + 4, bXS, L, L, SLD);
      // o = P v  (q is dead: o overlays it)
      gemm_tiles<2>(nw, MT, NT, LP, Mat{bXS, SLD}, Mat{bV, DS}, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v; });
      __syncthreads();
      // h1 = LN(x) + (o Wo^T + bo)
      gemm_packed(nw, MT, NT, DK, Mat{bQ, DS}, wo, [&](int r, int c, float v) {
        if (c < D) {
          const float h = bQN[r * DS + c] + v;
          bXS[r * DS + c] = h;
          if (a.save_h1 && r < L) a.save_h1[((int64_t)i * B * L + rowbase + r) * D + c] = h;
        }
      });
      __syncthreads();
      tap(a, b, tb + 5, bXS, L, D, DS);
      ln_rows(nw, bXS, bQN, L, DS, D, s_ln + (4 * i + 2) * 64, s_ln + (4 * i + 3) * 64);
      __syncthreads();
      tap(a, b, tb + 6, bQN, L, D, DS);
      // PW-FFN: y = (drop2(relu(drop1(h2 W1^T + b1)) W2^T + b2) + h2) * keep
      const DropSite ds1 = drop_site(a.drop_on, seed, site_ffn1(i), seq, a.drop_thr, a.drop_scale);
      const DropSite ds2 = drop_site(a.drop_on, seed, site_ffn2(i), seq, a.drop_thr, a.drop_scale);
      const WFrag w2 = load_wfrag(pk(i * 6 + 5, 0), P + o.c2_b, D, NT);
      gemm_packed(nw, MT, NT, DK, Mat{bQN, DS}, w1, [&](int r, int c, float v) {
        if (c < D) bQ[r * DS + c] = fmaxf(v * drop_mul(ds1, r, c), 0.f);
      });
      __syncthreads();
      gemm_packed(nw, MT, NT, DK, Mat{bQ, DS}, w2, [&](int r, int c, float v) {
        if (c < D) {
          const float y = (v * drop_mul(ds2, r, c) + bQN[r * DS + c]) * s_keep[r];
          bXS[r * DS + c] = y;
          if (a.save_x && r < L) a.save_x[((int64_t)(i + 1) * B * L + rowbase + r) * D + c] = y;
        }
      });
      __syncthreads();
      tap(a, b, tb + 7, bXS, L, D, DS);
    }

    // ---- head: (last_conv) -> last LayerNorm -> hidden, pos/neg logits, BCE partial sums
    const lds_f* hin = bXS;
    if (kind == SRFRD_SRFR) {
      const WFrag wl = load_wfrag(pk(ly.n_blocks * 6, 0), P + ly.off_lc_b, di, (di + 15) >> 4);
      gemm_packed(nw, MT, (di + 15) >> 4, DK, Mat{bXS, DS}, wl, [&](int r, int c, float v) { if (c < di) bQ[r * DS + c] = v; });
      __syncthreads();
      hin = bQ;
    }
    ln_rows(nw, hin, bQN, L, DS, dout, s_ln + (4 * ly.n_blocks) * 64, s_ln + (4 * ly.n_blocks + 1) * 64);
    __syncthreads();
    {
      float sp = 0.f, sn = 0.f, cnt = 0.f;
      const int q = tid & 3;
      const bool srfrn = kind == SRFRD_SRFRN;
      for (int t = tid >> 2; t < L; t += nthr >> 2) {                  // one DPP quad per position
        const int pid = s_pid[t], nid = s_nid[t];
        const int pf = (srfrn && a.pos_ids) ? (int)a.pos_fk[rowbase + t] : 0;
        const int nf = (srfrn && a.neg_ids) ? (int)a.neg_fk[rowbase + t] : 0;
        float ap = 0.f, an = 0.f;
#pragma unroll
        for (int j = 0; j < kQC; ++j) {
          const int c = q + 4 * j;
          if (c < dout) {
            const float h = bQN[t * DS + c];
            a.hidden[(rowbase + t) * dout + c] = h;
            if (a.pos_ids) ap += h * (c < di ? table[(int64_t)pid * di + c] : P[ly.off_side + pf * dfk + (c - di)]);
            if (a.neg_ids) an += h * (c < di ? table[(int64_t)nid * di + c] : P[ly.off_side + nf * dfk + (c - di)]);
          }
        }
        const float pl = quad_sum(ap), nl = quad_sum(an);
        if (q == 0) {
          if (a.pos_ids) a.pos_logits[rowbase + t] = pl;
          if (a.neg_ids) a.neg_logits[rowbase + t] = nl;
          if (a.loss_part && pid != 0) {          // trainer.py:36-38: both terms indexed by pos != 0
            sp += softplus_f(-pl);
            sn += softplus_f(nl);
            cnt += 1.f;
          }
        }
      }
      if (a.loss_part) {
        sp = wave_sum(sp); sn = wave_sum(sn); cnt = wave_sum(cnt);
        if (lane == 0) {
          s_misc[8 + wave * 3 + 0] = sp;
          s_misc[8 + wave * 3 + 1] = sn;
          s_misc[8 + wave * 3 + 2] = cnt;
        }
        __syncthreads();
        if (tid < 3) {
          float s = 0.f;
          for (int w = 0; w < nw; ++w) s += s_misc[8 + w * 3 + tid];
          a.loss_part[(int64_t)b * 3 + tid] = s;
        }
      }
    }
    __syncthreads();
  }
}

// ================================================================================================
// weight packing: canonical (N, K) row-major weights -> MFMA B-fragment order, both product forms
// ================================================================================================
__global__ void __launch_bounds__(256) pack_weights_kernel(const srfrd_layout ly, const float* __restrict__ dense,
                                                          float* __restrict__ packed) {
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

extern "C" int srfrd_pack_weights(const srfrd_layout* lay, const float* dense, float* packed, void* stream) {
  if (!lay || !dense || !packed) return SRFRD_E_ARG;
  if (lay->D > SRFRD_MAX_D) return SRFRD_E_UNSUPPORTED;
  hipLaunchKernelGGL(pack_weights_kernel, dim3((lay->n_blocks * 6 + 1) * 2), dim3(256), 0, (hipStream_t)stream, *lay, dense,
                     packed);
  return (int)hipGetLastError();
}

extern "C" int srfrd_debug_shape(const srfrd_layout* lay, int L, int64_t* slot_floats, int32_t* n_slots) {
  if (!lay || L <= 0) return SRFRD_E_ARG;
  if (slot_floats) *slot_floats = (int64_t)L * (L > lay->D ? L : lay->D);
  if (n_slots) *n_slots = 1 + 8 * lay->n_blocks;
  return 0;
}

extern "C" int srfrd_encoder_fwd(const srfrd_layout* lay, const float* item_table, const float* dense, const float* packed,
                                 const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                                 const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                                 double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                                 float* hidden, float* pos_logits, float* neg_logits, float* save_x, float* save_h1,
                                 float* loss_part, float* dbg, int dbg_seq, void* stream) {
  EncArgs a = {};
  int rc = fill_args(a, lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L,
                     dropout_p, seed, seed_dev, seq_index0);
  if (rc) return rc;
  if (!hidden || (pos_ids && !pos_logits) || (neg_ids && !neg_logits)) return SRFRD_E_ARG;
  if (loss_part && !(pos_ids && neg_ids)) return SRFRD_E_ARG;
  a.hidden = hidden; a.pos_logits = pos_logits; a.neg_logits = neg_logits;
  a.save_x = save_x; a.save_h1 = save_h1; a.loss_part = loss_part;
  a.dbg = dbg; a.dbg_seq = dbg_seq;
  srfrd_debug_shape(lay, L, &a.dbg_slot, nullptr);
  const Geom g = make_geom(L, lay->D);
  const int64_t lds = fwd_lds_floats(g, lay->n_blocks) * 4;
  if (lds > kLdsLimit) return SRFRD_E_UNSUPPORTED;
  const int per_cu = (int)(kLdsLimit / lds) > 2 ? 2 : (int)(kLdsLimit / lds);
  int grid = num_cu() * (per_cu < 1 ? 1 : per_cu);
  if (grid > B) grid = B;
  const int threads = env_threads("SRFRD_FWD_THREADS", 256);
  const bool spec = getenv("SRFRD_GENERIC") == nullptr && threads == 256 && lay->D == 50;
  if (spec && g.LP == 64) return launch_enc(encoder_fwd_kernel<50, 64, 4>, grid, threads, lds, stream, a);
  if (spec && g.LP == 32) return launch_enc(encoder_fwd_kernel<50, 32, 4>, grid, threads, lds, stream, a);
  return launch_enc(encoder_fwd_kernel<0, 0, 0>, grid, threads, lds, stream, a);
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
