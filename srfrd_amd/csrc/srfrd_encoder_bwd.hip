// Fused SRFRD encoder BACKWARD for MI355X (gfx950): recomputes each block's internals in LDS from the forward's
// checkpoints (block inputs, post-attention residual), back-propagates through them on the fp32 matrix cores, scatters
// item-row gradients with float atomics and accumulates every dense-parameter gradient in a per-workgroup slab.
// Replaces the autograd pass behind `loss.backward()` (reference trainer.py:40).
#include "srfrd_enc_common.h"

namespace srfrd {

// ================================================================================================
// backward
// ================================================================================================
// column sums over rows < L of an LDS matrix, added into a slab vector by ONE wave (fixed owner => the
// read-modify-write on the slab is race-free and order-deterministic)
__device__ __forceinline__ void colsum_to_slab(int owner_wave, const lds_f* buf, int ld, int rows, int cols, float* dst) {
  if (__builtin_amdgcn_readfirstlane(threadIdx.x >> 6) != owner_wave) return;
  const int lane = threadIdx.x & 63;
  if (lane < cols) {
    float s = 0.f;
    for (int t = 0; t < rows; ++t) s += buf[t * ld + lane];
    dst[lane] += s;
  }
}

// per-wave (dgamma, dbeta) partials -> LDS -> wave 0 sums in wave order -> slab
__device__ __forceinline__ void ln_param_grads_to_slab(lds_f* s_red, float dg, float db, int cols, float* dst_w, float* dst_b) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  s_red[(wave * 2 + 0) * 64 + lane] = dg;
  s_red[(wave * 2 + 1) * 64 + lane] = db;
  __syncthreads();
  if (wave == 0 && lane < cols) {
    float sg = 0.f, sb = 0.f;
    for (int w = 0; w < nw; ++w) {
      sg += s_red[(w * 2 + 0) * 64 + lane];
      sb += s_red[(w * 2 + 1) * 64 + lane];
    }
    dst_w[lane] += sg;
    dst_b[lane] += sb;
  }
  __syncthreads();
}

// D_, LP_, NW_ > 0: geometry and wave count fixed at compile time (strides become immediates, tile loops
// resolve statically); 0: read at run time (the generic instantiation covers every other shape).
template <int D_, int LP_, int NW_>
__global__ void __launch_bounds__(NW_ > 0 ? NW_ * 64 : 512) encoder_bwd_kernel(const EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Dims& ly = a.dm;
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int nw = NW_ > 0 ? NW_ : (int)(blockDim.x >> 6), nthr = nw << 6;
  const int L = a.L;
  const int D = D_ > 0 ? D_ : ly.D;
  const int LP = LP_ > 0 ? LP_ : ((L + 15) & ~15);
  const int DK = (D + 3) & ~3, DS = DK + 2, SLD = LP + 2, NT = (D + 15) >> 4, MT = LP >> 4;
  const int LK = (L + 3) & ~3;           // token k-range of the gradient GEMMs: rows >= L of every gradient matrix are zero
  Geom g;
  g.L = L; g.LP = LP; g.D = D; g.DK = DK; g.DS = DS; g.SLD = SLD; g.NT = NT; g.MT = MT;
  const int szA = LP * DS, szS = imax(LP * SLD, szA);
  lds_f* const lds0 = (lds_f*)smem;
  lds_f* bX = lds0;
  lds_f* bQN = bX + szA;
  lds_f* bQ = bQN + szA;
  lds_f* bK = bQ + szA;
  lds_f* bV = bK + szA;
  lds_f* bO = bV + szA;
  lds_f* bG = bO + szA;
  lds_f* bT = bG + szA;
  lds_f* S1 = bT + szA;
  lds_f* S2 = S1 + szS;
  lds_f* tail = S2 + szS + kSlack;
  lds_i* s_in = (lds_i*)tail;
  lds_f* s_keep = tail + LP;
  lds_i* s_pid = (lds_i*)(tail + 2 * LP);
  lds_i* s_nid = (lds_i*)(tail + 3 * LP);
  lds_i* s_fk = (lds_i*)(tail + 4 * LP);
  lds_i* s_pfk = (lds_i*)(tail + 5 * LP);
  lds_i* s_nfk = (lds_i*)(tail + 6 * LP);
  lds_f* s_dpl = tail + 7 * LP;
  lds_f* s_dnl = tail + 8 * LP;
  lds_f* s_misc = tail + 10 * LP;        // 64
  lds_f* s_ln = s_misc + 64;             // LayerNorm parameter cache
  lds_f* s_lng = s_ln + ln_cache_floats(ly.n_blocks);   // LayerNorm parameter-gradient accumulators (same indexing)
  {
    const int total = (int)bwd_lds_floats(g, ly.n_blocks);
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
  float* slab = a.grad_slabs + (int64_t)blockIdx.x * ly.n_dense;
  for (int64_t i = tid; i < ly.n_dense; i += nthr) slab[i] = 0.f;
  __syncthreads();
  const bool fold_bias = (D & 15) != 0;       // a spare padded column exists in the last n-tile
  const float keep_scale = a.drop_on ? a.drop_scale : 1.0f;

  STAMP_INIT
  for (int b = blockIdx.x; b < B; b += gridDim.x) {
    const int64_t rowbase = (int64_t)b * L;
    const uint32_t seq = (uint32_t)(a.seq0 + b);
    const int rmw = b != (int)blockIdx.x;      // first sequence of this workgroup: the slab is still all zero
    for (int t = tid; t < LP; t += nthr) {
      const bool in = t < L;
      const int id = in ? (int)a.in_ids[rowbase + t] : 0;
      const int pid = (in && a.pos_ids) ? (int)a.pos_ids[rowbase + t] : 0;
      const int nid = (in && a.neg_ids) ? (int)a.neg_ids[rowbase + t] : 0;
      s_in[t] = id;
      s_keep[t] = id != 0 ? 1.f : 0.f;
      s_pid[t] = pid;
      s_nid[t] = nid;
      s_fk[t] = (in && a.fk_ids) ? (int)a.fk_ids[rowbase + t] : 0;
      s_pfk[t] = (in && a.pos_fk) ? (int)a.pos_fk[rowbase + t] : 0;
      s_nfk[t] = (in && a.neg_fk) ? (int)a.neg_fk[rowbase + t] : 0;
      float dp = 0.f, dn = 0.f;
      if (in) {
        if (a.fused_bce) {
          if (pid != 0) {
            dp = sigmoid_f(a.c_pl[rowbase + t]) - 1.0f;
            dn = sigmoid_f(a.c_nl[rowbase + t]);
          }
        } else {
          if (a.d_pos) dp = a.d_pos[rowbase + t];
          if (a.d_neg) dn = a.d_neg[rowbase + t];
        }
      }
      s_dpl[t] = dp;
      s_dnl[t] = dn;
    }
    if (is_srfu && wave == 0) {
      const int lab = user_label_wave(kind, a.fk_ids ? a.fk_ids + rowbase : nullptr, L, ly.n_labels);
      if (lane == 0) ((lds_i*)s_misc)[0] = lab;
    }
    // final block output (input of last_conv / last LayerNorm) -> bX ; gradient pad rows must be exact zeros
    for (int i = tid; i < L * D; i += nthr) {
      const int t = i / D, c = i - t * D;
      bX[t * DS + c] = a.c_save_x[((int64_t)ly.n_blocks * B * L + rowbase + t) * D + c];
    }
    for (int i = tid; i < (LP - L) * DS; i += nthr) bG[L * DS + i] = 0.f;
    __syncthreads();

    const lds_f* lnin = bX;
    if (kind == SRFRD_SRFR) {
      const WFrag wl = load_wfrag(pk(ly.n_blocks * 6, 0), P + ly.off_lc_b, di, (di + 15) >> 4);
      gemm_packed(nw, MT, (di + 15) >> 4, DK, Mat{bX, DS}, wl, [&](int r, int c, float v) { if (c < di) bQ[r * DS + c] = v; });
      __syncthreads();
      lnin = bQ;
    }
    // ---- logits backward.  Pass 1: dh -> bG and the hidden rows -> bT.  Threads walk (position, channel) linearly, so a
    // wave reads each gathered item row as one contiguous 4*d_item-byte segment (the rows sit anywhere in the table).
    {
      const bool srfrn = kind == SRFRD_SRFRN;
      int t = tid / dout, c = tid - t * dout;
      const int dt = nthr / dout, dc = nthr - dt * dout;
      for (; t < L; ) {
        const float h = a.c_hidden[(rowbase + t) * dout + c];
        float dh = a.d_hidden ? a.d_hidden[(rowbase + t) * dout + c] : 0.f;
        const float dp = s_dpl[t], dn = s_dnl[t];
        if (c < di) {
          if (a.pos_ids) dh += dp * table[(int64_t)s_pid[t] * di + c];
          if (a.neg_ids) dh += dn * table[(int64_t)s_nid[t] * di + c];
        } else if (srfrn) {
          if (a.pos_ids) dh += dp * P[ly.off_side + s_pfk[t] * dfk + (c - di)];
          if (a.neg_ids) dh += dn * P[ly.off_side + s_nfk[t] * dfk + (c - di)];
        }
        bT[t * DS + c] = h;
        bG[t * DS + c] = dh;
        t += dt; c += dc;
        if (c >= dout) { c -= dout; ++t; }
      }
      if (dout < D)                                    // SRFR: channels d_out..D-1 of the LN input gradient are zero
        for (int idx = tid; idx < L * (D - dout); idx += nthr) {
          const int tt = idx / (D - dout), cc = dout + idx - tt * (D - dout);
          bG[tt * DS + cc] = 0.f;
        }
    }
    __syncthreads();
    // Pass 2: item-table scatter (float atomics, no return), again linear in (position, channel): every wave-instruction
    // adds to at most two contiguous row segments - the shape the memory-side atomic units run at full rate for.
    // Row 0 (padding_idx) receives none.
    {
      int t = tid / di, c = tid - t * di;
      const int dt = nthr / di, dc = nthr - dt * di;
      for (; t < L; ) {
        const float h = bT[t * DS + c];
        const float dp = s_dpl[t], dn = s_dnl[t];
        const int pid = s_pid[t], nid = s_nid[t];
        if (a.pos_ids && pid != 0 && dp != 0.f) atomicAdd(&a.grad_table[(int64_t)pid * di + c], dp * h);
        if (a.neg_ids && nid != 0 && dn != 0.f) atomicAdd(&a.grad_table[(int64_t)nid * di + c], dn * h);
        t += dt; c += dc;
        if (c >= di) { c -= di; ++t; }
      }
    }
    if (kind == SRFRD_SRFRN && wave == (1 % nw) && lane < dfk && (a.pos_ids || a.neg_ids)) {
      for (int f = 1; f <= 2; ++f) {      // fake_embed rows 1 (fake) and 2 (real); row 0 is padding_idx
        float s = 0.f;
        for (int t = 0; t < L; ++t) {
          const float w = (s_pfk[t] == f ? s_dpl[t] : 0.f) + (s_nfk[t] == f ? s_dnl[t] : 0.f);
          s += w * bT[t * DS + di + lane];
        }
        slab[ly.off_side + f * dfk + lane] += s;
      }
    }
    // ---- last LayerNorm backward: dx -> bK, g * xhat -> bO; dgamma / dbeta as ones-row GEMMs on the matrix cores
    ln_bwd_rows<false>(nw, bG, lnin, bK, bO, L, LP, DS, dout, s_ln + (4 * ly.n_blocks) * 64);
    __syncthreads();
    gemm_tiles<0>(nw, 1, (dout + 15) >> 4, LK, OnesRow{}, Mat{bG, DS},
                  [=](int r, int c, float v) { if (r == 0 && c < dout) s_lng[(4 * ly.n_blocks + 1) * 64 + c] += v; });
    gemm_tiles<0>(nw, 1, (dout + 15) >> 4, LK, OnesRow{}, Mat{bO, DS},
                  [=](int r, int c, float v) { if (r == 0 && c < dout) s_lng[(4 * ly.n_blocks + 0) * 64 + c] += v; });
    __syncthreads();
    { lds_f* t_ = bG; bG = bK; bK = t_; }
    if (kind == SRFRD_SRFR) {             // hc = hf Wlc^T + blc
      gemm_slab(nw, (di + 15) >> 4, NT, LK, MatT{bG, DS}, MatOnes{bX, DS, D},
                SlabWB{slab + ly.off_lc_w, fold_bias ? slab + ly.off_lc_b : nullptr, di, D, rmw});
      if (!fold_bias) colsum_to_slab(0, bG, DS, L, di, slab + ly.off_lc_b);
      const WFrag wln = load_wfrag(pk(ly.n_blocks * 6, 1), nullptr, 0, NT);
      gemm_packed(nw, MT, NT, (di + 3) & ~3, Mat{bG, DS}, wln, [&](int r, int c, float v) { if (c < D) bT[r * DS + c] = v; });
      __syncthreads();
      lds_f* t_ = bG; bG = bT; bT = t_;
    }
    tap(a, b, 0, bG, L, D, DS);

    for (int i = ly.n_blocks - 1; i >= 0; --i) {
      const BlkOff o = blk_off(ly.blk0 + i * ly.blk_stride, D);
      const int tb = 1 + 4 * i;
      const DropSite dsA = drop_site(a.drop_on, seed, site_attn(i), seq, a.drop_thr, a.drop_scale);
      const DropSite ds1 = drop_site(a.drop_on, seed, site_ffn1(i), seq, a.drop_thr, a.drop_scale);
      const DropSite ds2 = drop_site(a.drop_on, seed, site_ffn2(i), seq, a.drop_thr, a.drop_scale);
      launder(bX); launder(bQN); launder(bQ); launder(bK); launder(bV); launder(bO); launder(bG); launder(bT);
      launder(S1); launder(S2);
      // ================= FFN half: y = (drop2(a2) + h2) * keep, a2 = relu(drop1(h2 W1^T + b1)) W2^T + b2
      const WFrag w1t = load_wfrag(pk(i * 6 + 4, 0), P + o.c1_b, D, NT);
      const WFrag w2n = load_wfrag(pk(i * 6 + 5, 1), nullptr, 0, NT);
      const WFrag w1n = load_wfrag(pk(i * 6 + 4, 1), nullptr, 0, NT);
      // old slab values of this block's FFN weight gradients, requested three phases before they are needed
      const SlabWB sl_w2{slab + o.c2_w, fold_bias ? slab + o.c2_b : nullptr, D, D, rmw};
      const SlabWB sl_w1{slab + o.c1_w, fold_bias ? slab + o.c1_b : nullptr, D, D, rmw};
      const SlabPre pre_w2 = slab_preload(nw, NT, NT, sl_w2);
      const SlabPre pre_w1 = slab_preload(nw, NT, NT, sl_w1);
      for (int idx = tid; idx < L * D; idx += nthr) {
        const int t = idx / D, c = idx - t * D;
        bG[t * DS + c] *= s_keep[t];
        bX[t * DS + c] = a.c_save_h1[((int64_t)i * B * L + rowbase + t) * D + c];
      }
      __syncthreads();
      ln_rows(nw, bX, bQN, L, DS, D, s_ln + (4 * i + 2) * 64, s_ln + (4 * i + 3) * 64);      // h2
      for (int idx = tid; idx < LP * D; idx += nthr) {                             // dA2 = drop2'(dy)
        const int t = idx / D, c = idx - t * D;
        bK[t * DS + c] = t < L ? bG[t * DS + c] * drop_mul(ds2, t, c) : 0.f;
      }
      __syncthreads();
      gemm_packed(nw, MT, NT, DK, Mat{bQN, DS}, w1t, [&](int r, int c, float v) {
        if (c < D) bQ[r * DS + c] = fmaxf(v * drop_mul(ds1, r, c), 0.f);                        // r = relu(drop1(a1))
      });
      __syncthreads();
      gemm_slab(nw, NT, NT, LK, MatT{bK, DS}, MatOnes{bQ, DS, D}, sl_w2, pre_w2);                   // dW2 += dA2^T r (+ db2)
      if (!fold_bias) colsum_to_slab(0, bK, DS, L, D, slab + o.c2_b);
      gemm_packed(nw, MT, NT, DK, Mat{bK, DS}, w2n, [&](int r, int c, float v) {
        if (c < D) bV[r * DS + c] = bQ[r * DS + c] > 0.f ? v * keep_scale : 0.f;               // dA1
      });
      __syncthreads();
      gemm_slab(nw, NT, NT, LK, MatT{bV, DS}, MatOnes{bQN, DS, D}, sl_w1, pre_w1);                  // dW1 += dA1^T h2 (+ db1)
      if (!fold_bias) colsum_to_slab(1 % nw, bV, DS, L, D, slab + o.c1_b);
      gemm_packed(nw, MT, NT, DK, Mat{bV, DS}, w1n, [&](int r, int c, float v) { if (c < D) bG[r * DS + c] += v; });   // dh2 = dy + dA1 W1
      __syncthreads();
      ln_bwd_rows<false>(nw, bG, bX, bT, bV, L, LP, DS, D, s_ln + (4 * i + 2) * 64);               // dh1 -> bT
      __syncthreads();
      gemm_tiles<0>(nw, 1, NT, LK, OnesRow{}, Mat{bG, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) s_lng[(4 * i + 3) * 64 + c] += v; });
      gemm_tiles<0>(nw, 1, NT, LK, OnesRow{}, Mat{bV, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) s_lng[(4 * i + 2) * 64 + c] += v; });
      __syncthreads();
      { lds_f* t_ = bG; bG = bT; bT = t_; }
      tap(a, b, tb + 0, bG, L, D, DS);
      // ================= attention half: h1 = LN1(x) + (P v) Wo^T + bo
      for (int idx = tid; idx < L * D; idx += nthr) {
        const int t = idx / D, c = idx - t * D;
        bX[t * DS + c] = a.c_save_x[((int64_t)i * B * L + rowbase + t) * D + c];
      }
      __syncthreads();
      const WFrag wq = load_wfrag(pk(i * 6 + 0, 0), P + o.in_b, D, NT);
      const WFrag wk = load_wfrag(pk(i * 6 + 1, 0), P + o.in_b + D, D, NT);
      const WFrag wv = load_wfrag(pk(i * 6 + 2, 0), P + o.in_b + 2 * D, D, NT);
      ln_rows(nw, bX, bQN, L, DS, D, s_ln + (4 * i + 0) * 64, s_ln + (4 * i + 1) * 64);
      __syncthreads();
      gemm_packed(nw, MT, NT, DK, Mat{bQN, DS}, wq, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v * qscale; });
      gemm_packed(nw, MT, NT, DK, Mat{bX, DS}, wk, [&](int r, int c, float v) { if (c < D) bK[r * DS + c] = v; });
      gemm_packed(nw, MT, NT, DK, Mat{bX, DS}, wv, [&](int r, int c, float v) { if (c < D) bV[r * DS + c] = v; });
      const WFrag won = load_wfrag(pk(i * 6 + 3, 1), nullptr, 0, NT);
      const SlabWB sl_wo{slab + o.out_w, fold_bias ? slab + o.out_b : nullptr, D, D, rmw};
      const SlabPre pre_wo = slab_preload(nw, NT, NT, sl_wo);
      __syncthreads();
      gemm_tiles<1>(nw, MT, MT, DK, Mat{bQ, DS}, MatT{bK, DS}, [&](int r, int c, float v) { S1[r * SLD + c] = v; });
      __syncthreads();
      softmax_rows<false>(nw, S1, L, SLD, LP, dsA, S2);              // S1 <- P (for dS), S2 <- dropout(P) (for o and dv)
      __syncthreads();
      gemm_tiles<2>(nw, MT, NT, LP, Mat{S2, SLD}, Mat{bV, DS},
                    [&](int r, int c, float v) { if (c < D) bO[r * DS + c] = v; });            // o = drop(P) v
      __syncthreads();
      gemm_slab(nw, NT, NT, LK, MatT{bG, DS}, MatOnes{bO, DS, D}, sl_wo, pre_wo);                   // dWo += dh1^T o (+ dbo)
      if (!fold_bias) colsum_to_slab(2 % nw, bG, DS, L, D, slab + o.out_b);
      __syncthreads();
      gemm_packed(nw, MT, NT, DK, Mat{bG, DS}, won, [&](int r, int c, float v) { if (c < D) bO[r * DS + c] = v; });    // do = dh1 Wo
      const WFrag wqn = load_wfrag(pk(i * 6 + 0, 1), nullptr, 0, NT);
      const WFrag wkn = load_wfrag(pk(i * 6 + 1, 1), nullptr, 0, NT);
      const WFrag wvn = load_wfrag(pk(i * 6 + 2, 1), nullptr, 0, NT);
      const SlabWB sl_wq{slab + o.in_w, fold_bias ? slab + o.in_b : nullptr, D, D, rmw};
      const SlabWB sl_wk{slab + o.in_w + D * D, fold_bias ? slab + o.in_b + D : nullptr, D, D, rmw};
      const SlabWB sl_wv{slab + o.in_w + 2 * D * D, fold_bias ? slab + o.in_b + 2 * D : nullptr, D, D, rmw};
      const SlabPre pre_wq = slab_preload(nw, NT, NT, sl_wq);
      const SlabPre pre_wk = slab_preload(nw, NT, NT, sl_wk);
      const SlabPre pre_wv = slab_preload(nw, NT, NT, sl_wv);
      __syncthreads();
      gemm_tiles<3>(nw, MT, NT, LK, MatT{S2, SLD}, Mat{bO, DS},
                    [&](int r, int c, float v) { if (c < D) bT[r * DS + c] = v; });            // dv = drop(P)^T do
      __syncthreads();
      gemm_tiles<1>(nw, MT, MT, DK, Mat{bO, DS}, MatT{bV, DS}, [&](int r, int c, float v) { S2[r * SLD + c] = v; });  // dPd = do v^T
      __syncthreads();
      softmax_bwd_rows(nw, S2, S1, L, SLD, LP, dsA);                 // dS = P * (dP - sum_j dP_j P_j), dP = mask * dPd
      __syncthreads();
      lds_f* dKb = S1;                                           // P is dead: dk overlays it as [LP][DS]
      gemm_tiles<2>(nw, MT, NT, LP, Mat{S2, SLD}, Mat{bK, DS},
                    [&](int r, int c, float v) { if (c < D) bO[r * DS + c] = v * qscale; });   // dq (pre-scale)
      gemm_tiles<3>(nw, MT, NT, LK, MatT{S2, SLD}, Mat{bQ, DS},
                    [&](int r, int c, float v) { if (c < D) dKb[r * DS + c] = v; });           // dk = dS^T q
      __syncthreads();
      gemm_slab(nw, NT, NT, LK, MatT{bO, DS}, MatOnes{bQN, DS, D}, sl_wq, pre_wq);                  // dWq (+ dbq)
      gemm_slab(nw, NT, NT, LK, MatT{dKb, DS}, MatOnes{bX, DS, D}, sl_wk, pre_wk);                  // dWk (+ dbk)
      gemm_slab(nw, NT, NT, LK, MatT{bT, DS}, MatOnes{bX, DS, D}, sl_wv, pre_wv);                   // dWv (+ dbv)
      if (!fold_bias) {
        colsum_to_slab(0, bO, DS, L, D, slab + o.in_b);
        colsum_to_slab(1 % nw, dKb, DS, L, D, slab + o.in_b + D);
        colsum_to_slab(2 % nw, bT, DS, L, D, slab + o.in_b + 2 * D);
      }
      gemm_packed(nw, MT, NT, DK, Mat{bO, DS}, wqn, [&](int r, int c, float v) { if (c < D) bG[r * DS + c] += v; });   // dLN1 = dh1 + dq Wq
      gemm_packed(nw, MT, NT, DK, Mat{dKb, DS}, wkn, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v; });   // dx  = dk Wk
      gemm_packed(nw, MT, NT, DK, Mat{bT, DS}, wvn, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] += v; });   //     + dv Wv
      __syncthreads();
      ln_bwd_rows<true>(nw, bG, bX, bQ, S2, L, LP, DS, D, s_ln + (4 * i + 0) * 64);                //     + LN1 bwd
      __syncthreads();
      gemm_tiles<0>(nw, 1, NT, LK, OnesRow{}, Mat{bG, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) s_lng[(4 * i + 1) * 64 + c] += v; });
      gemm_tiles<0>(nw, 1, NT, LK, OnesRow{}, Mat{S2, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) s_lng[(4 * i + 0) * 64 + c] += v; });
      __syncthreads();
      lds_f* t_ = bG; bG = bQ; bQ = t_;
      tap(a, b, tb + 1, bG, L, D, DS);
    }

    // ---- embedding backward: item rows (atomics), position table, side channel
    {
      const DropSite dsE = drop_site(a.drop_on && is_sas, seed, SITE_EMB, seq, a.drop_thr, a.drop_scale);
      {
        int t = tid / di, c = tid - t * di;
        const int dt = nthr / di, dc = nthr - dt * di;
        for (; t < L; ) {
          float gv = bG[t * DS + c] * s_keep[t];
          if (is_sas) gv *= drop_mul(dsE, t, c);
          const int id = s_in[t];
          if (id != 0) atomicAdd(&a.grad_table[(int64_t)id * di + c], is_sas ? gv * sqrtD : gv);
          slab[ly.off_pos + t * di + c] += gv;          // (t, c) is owned by the same thread for every sequence
          t += dt; c += dc;
          if (c >= di) { c -= di; ++t; }
        }
      }
      if (has_fake && wave == (1 % nw) && lane < dfk) {
        for (int f = 1; f <= 2; ++f) {
          float s = 0.f;
          for (int t = 0; t < L; ++t)
            if (s_fk[t] == f) s += bG[t * DS + di + lane] * s_keep[t];
          slab[ly.off_side + f * dfk + lane] += s;
        }
      }
      if (is_srfu && wave == (1 % nw) && lane < D) {
        const int lab = ((lds_i*)s_misc)[0];
        float s = 0.f;
        for (int t = 0; t < L; ++t) s += bG[t * DS + lane] * s_keep[t];
        slab[ly.off_side + lab * D + lane] += s;
      }
    }
    __syncthreads();
  }
  // LayerNorm parameter gradients: LDS accumulators -> slab (same vector indexing as the parameter cache)
  for (int idx = tid; idx < (4 * ly.n_blocks + 2) * 64; idx += nthr) {
    const int vec = idx >> 6, c = idx & 63;
    if (vec < 4 * ly.n_blocks) {
      const BlkOff o = blk_off(ly.blk0 + (vec >> 2) * ly.blk_stride, D);
      const int sel = vec & 3;
      if (c < D) slab[(sel == 0 ? o.ln1_w : sel == 1 ? o.ln1_b : sel == 2 ? o.ln2_w : o.ln2_b) + c] = s_lng[idx];
    } else if (c < dout) {
      slab[(vec == 4 * ly.n_blocks ? ly.off_ll_w : ly.off_ll_b) + c] = s_lng[idx];
    }
  }
}

}  // namespace srfrd

using namespace srfrd;

extern "C" int srfrd_bwd_grid(int B) {
  if (B <= 0) return SRFRD_E_ARG;
  const int cu = num_cu();
  return B < cu ? B : cu;
}

extern "C" int srfrd_encoder_bwd(const srfrd_layout* lay, const float* item_table, const float* dense, const float* packed,
                                 const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids,
                                 const int64_t* pos_fake, const int64_t* neg_ids, const int64_t* neg_fake, int B, int L,
                                 double dropout_p, uint32_t seed, const uint32_t* seed_dev, int64_t seq_index0,
                                 const float* hidden, const float* pos_logits, const float* neg_logits,
                                 const float* save_x, const float* save_h1, const float* d_hidden, const float* d_pos,
                                 const float* d_neg, int fused_bce, float* grad_table, float* grad_slabs, float* dbg,
                                 int dbg_seq, void* stream) {
  EncArgs a = {};
  int rc = fill_args(a, lay, item_table, dense, packed, input_ids, fake_ids, pos_ids, pos_fake, neg_ids, neg_fake, B, L,
                     dropout_p, seed, seed_dev, seq_index0);
  if (rc) return rc;
  if (!hidden || !save_x || !save_h1 || !grad_table || !grad_slabs) return SRFRD_E_ARG;
  if (fused_bce && !(pos_ids && neg_ids && pos_logits && neg_logits)) return SRFRD_E_ARG;
  a.c_hidden = hidden; a.c_pl = pos_logits; a.c_nl = neg_logits; a.c_save_x = save_x; a.c_save_h1 = save_h1;
  a.d_hidden = d_hidden; a.d_pos = d_pos; a.d_neg = d_neg; a.fused_bce = fused_bce;
  a.grad_table = grad_table; a.grad_slabs = grad_slabs;
  a.dbg = dbg; a.dbg_seq = dbg_seq;
  srfrd_debug_shape(lay, L, &a.dbg_slot, nullptr);
  const Geom g = make_geom(L, lay->D);
  const int64_t lds = bwd_lds_floats(g, lay->n_blocks) * 4;
  if (lds > kLdsLimit) return SRFRD_E_UNSUPPORTED;
  const int grid = srfrd_bwd_grid(B);
  const int threads = env_threads("SRFRD_BWD_THREADS", 512);
  const bool spec = getenv("SRFRD_GENERIC") == nullptr && threads == 512 && lay->D == 50;
  if (spec && g.LP == 64) return launch_enc(encoder_bwd_kernel<50, 64, 8>, grid, threads, lds, stream, a);
  if (spec && g.LP == 32) return launch_enc(encoder_bwd_kernel<50, 32, 8>, grid, threads, lds, stream, a);
  return launch_enc(encoder_bwd_kernel<0, 0, 0>, grid, threads, lds, stream, a);
}

