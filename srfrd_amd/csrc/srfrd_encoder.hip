// Fused SRFRD encoder forward / backward for MI355X (gfx950).
//
// One persistent workgroup walks sequences b = blockIdx.x, blockIdx.x + gridDim.x, ...; the whole
// per-sequence working set (embedded inputs, LN outputs, Q/K/V, the L x L scores, FFN hidden) stays in
// LDS between phases, so HBM sees only ids, the gathered embedding rows, the outputs and (training) the
// per-block checkpoints.  Reference arithmetic being reproduced: SURVEY.md 3.4 / reference
// SRFR_model.py:92-142 (SRFR), :192-239 (SRFRN), :473-530 (SRFU_*), :621-666 (SASRec); torch
// multi_head_attention_forward explicit path (q from LN(x), k = v from x, q * sqrt(1/d_h), additive causal
// -inf mask, softmax, dropout on P, P v, out_proj); residuals on the LayerNormed tensors; eps = 1e-8.
#include "srfrd_dev.h"

namespace srfrd {

struct EncArgs {
  Dims dm;
  const float* table;
  const float* dense;
  const float* packed;      // srfrd_pack_weights output (MFMA-fragment-ordered weights)
  const int64_t *in_ids, *fk_ids, *pos_ids, *pos_fk, *neg_ids, *neg_fk;
  int B, L;
  uint32_t seed;
  const uint32_t* seed_dev;
  uint32_t drop_thr;
  float drop_scale;
  int drop_on;
  int64_t seq0;
  float qscale;
  // forward outputs
  float *hidden, *pos_logits, *neg_logits, *save_x, *save_h1, *loss_part;
  // backward inputs / outputs
  const float *c_hidden, *c_pl, *c_nl, *c_save_x, *c_save_h1, *d_hidden, *d_pos, *d_neg;
  int fused_bce;
  float *grad_table, *grad_slabs;
  // debug taps
  float* dbg;
  int dbg_seq;
  int64_t dbg_slot;
};

// Optimisation barrier on a wave-uniform pointer: stops LLVM from hoisting the per-call-site address arithmetic of
// ~35 inlined GEMMs out of the sequence / block loops (which costs > 256 VGPRs and spills).
__device__ __forceinline__ void launder(lds_f*& p) {
  asm volatile("" : "+s"(p));
}
__device__ __forceinline__ void launder(const float*& p) {
  asm volatile("" : "+s"(p));
}
__device__ __forceinline__ void launder(float*& p) {
  asm volatile("" : "+s"(p));
}

// Diagnostic build only (tools/phase_profile.py compiles a copy with -DSRFRD_STAMPS and a STAMP(n) after every
// workgroup barrier): thread 0 adds the s_memtime delta of each phase into a per-workgroup table that aliases the
// debug-tap buffer.  The shipped library contains no stamp.
#ifdef SRFRD_STAMPS
#define STAMP_INIT unsigned long long* stamp_acc = (unsigned long long*)a.dbg + (int64_t)blockIdx.x * 128; \
                   unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#define STAMP(id) do { if (threadIdx.x == 0 && a.dbg) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                       stamp_acc[id] += t_ - stamp_prev; stamp_prev = t_; } } while (0)
#else
#define STAMP_INIT
#define STAMP(id) do {} while (0)
#endif

// LayerNorm weights / biases -> LDS once per workgroup (read by every row pass of every sequence)
__device__ __forceinline__ void fill_ln_cache(lds_f* s_ln, const float* P, const Dims& ly) {
  const int D = ly.D;
  for (int idx = threadIdx.x; idx < (4 * ly.n_blocks + 2) * 64; idx += blockDim.x) {
    const int vec = idx >> 6, c = idx & 63;
    float v = 0.f;
    if (vec < 4 * ly.n_blocks) {
      const BlkOff o = blk_off(ly.blk0 + (vec >> 2) * ly.blk_stride, D);
      const int sel = vec & 3;
      const int off = sel == 0 ? o.ln1_w : sel == 1 ? o.ln1_b : sel == 2 ? o.ln2_w : o.ln2_b;
      if (c < D) v = P[off + c];
    } else if (c < ly.d_out) {
      v = P[(vec == 4 * ly.n_blocks ? ly.off_ll_w : ly.off_ll_b) + c];
    }
    s_ln[idx] = v;
  }
}

__device__ __forceinline__ void tap(const EncArgs& a, int b, int slot, const lds_f* buf, int rows, int cols, int ld) {
#ifdef SRFRD_STAMPS
  return;
#endif
  if (a.dbg == nullptr || b != a.dbg_seq) return;
  float* dst = a.dbg + (int64_t)slot * a.dbg_slot;
  for (int i = threadIdx.x; i < rows * cols; i += blockDim.x) {
    const int r = i / cols, c = i - r * cols;
    dst[i] = buf[r * ld + c];
  }
}

// user label of one sequence from its fake/real ids (reference SRFR_model.py:546-570); wave-uniform result
__device__ __forceinline__ int user_label_wave(int kind, const int64_t* fk_row, int L, int n_labels) {
  const int lane = threadIdx.x & 63;
  int n1 = 0, n2 = 0;
  if (fk_row != nullptr)
    for (int t = lane; t < L; t += 64) {
      const int f = (int)fk_row[t];
      n1 += (f == 1);
      n2 += (f == 2);
    }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    n1 += __shfl_xor(n1, o, 64);
    n2 += __shfl_xor(n2, o, 64);
  }
  int lab;
  if (kind == SRFRD_SRFU_B) lab = (n1 < n2) ? 1 : 2;                       // round-half-even(1.5) = 2 on ties
  else if (kind == SRFRD_SRFU_F) lab = n1;
  else if (kind == SRFRD_SRFU_R) {
    const int tot = n1 + n2;                                               // all-pad row: reference is 0/0; guarded to 0
    lab = tot == 0 ? 0 : (int)floorf(((float)n1 / (float)tot) * 10.0f);
  } else lab = (n1 > n2) ? 2 : 1;                                          // SRFRN.predict: int() truncation, tie -> 1
  if (n_labels > 0) lab = min(max(lab, 0), n_labels - 1);
  return lab;
}

// ================================================================================================
// forward
// ================================================================================================
__global__ void __launch_bounds__(512) encoder_fwd_kernel(const EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Dims& ly = a.dm;
  const Geom g = make_geom(a.L, ly.D);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6, nthr = blockDim.x;
  const int L = g.L, LP = g.LP, D = g.D, DS = g.DS, SLD = g.SLD, NT = g.NT, MT = g.MT, DK = g.DK;
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
      ln_rows(bXS, bQN, L, DS, D, s_ln + (4 * i + 0) * 64, s_ln + (4 * i + 1) * 64);
      __syncthreads();
      tap(a, b, tb + 0, bQN, L, D, DS);
      // q = (LN(x) Wq^T + bq) * sqrt(1/d_h);  k = x Wk^T + bk;  v = x Wv^T + bv
      gemm_packed(MT, NT, DK, Mat{bQN, DS}, wq, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v * qscale; });
      gemm_packed(MT, NT, DK, Mat{bXS, DS}, wk, [&](int r, int c, float v) { if (c < D) bK[r * DS + c] = v; });
      gemm_packed(MT, NT, DK, Mat{bXS, DS}, wv, [&](int r, int c, float v) { if (c < D) bV[r * DS + c] = v; });
      const WFrag wo = load_wfrag(pk(i * 6 + 3, 0), P + o.out_b, D, NT);
      const WFrag w1 = load_wfrag(pk(i * 6 + 4, 0), P + o.c1_b, D, NT);
      __syncthreads();
      tap(a, b, tb + 1, bQ, L, D, DS);
      tap(a, b, tb + 2, bK, L, D, DS);
      tap(a, b, tb + 3, bV, L, D, DS);
      // S = q k^T on the lower-triangular tiles (x is dead: S overlays it)
      gemm_tiles<1>(MT, MT, DK, Mat{bQ, DS}, MatT{bK, DS}, [&](int r, int c, float v) { bXS[r * SLD + c] = v; });
      __syncthreads();
      // causal softmax (+ attention dropout); keys j > r get exact zeros up to LP
      {
        const DropSite dsA = drop_site(a.drop_on, seed, site_attn(i), seq, a.drop_thr, a.drop_scale);
        softmax_rows<true>(bXS, L, SLD, LP, dsA);
      }
      __syncthreads();
      tap(a, b, tb + 4, bXS, L, L, SLD);
      // o = P v  (q is dead: o overlays it)
      gemm_tiles<2>(MT, NT, LP, Mat{bXS, SLD}, Mat{bV, DS}, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v; });
      __syncthreads();
      // h1 = LN(x) + (o Wo^T + bo)
      gemm_packed(MT, NT, DK, Mat{bQ, DS}, wo, [&](int r, int c, float v) {
        if (c < D) {
          const float h = bQN[r * DS + c] + v;
          bXS[r * DS + c] = h;
          if (a.save_h1 && r < L) a.save_h1[((int64_t)i * B * L + rowbase + r) * D + c] = h;
        }
      });
      __syncthreads();
      tap(a, b, tb + 5, bXS, L, D, DS);
      ln_rows(bXS, bQN, L, DS, D, s_ln + (4 * i + 2) * 64, s_ln + (4 * i + 3) * 64);
      __syncthreads();
      tap(a, b, tb + 6, bQN, L, D, DS);
      // PW-FFN: y = (drop2(relu(drop1(h2 W1^T + b1)) W2^T + b2) + h2) * keep
      const DropSite ds1 = drop_site(a.drop_on, seed, site_ffn1(i), seq, a.drop_thr, a.drop_scale);
      const DropSite ds2 = drop_site(a.drop_on, seed, site_ffn2(i), seq, a.drop_thr, a.drop_scale);
      const WFrag w2 = load_wfrag(pk(i * 6 + 5, 0), P + o.c2_b, D, NT);
      gemm_packed(MT, NT, DK, Mat{bQN, DS}, w1, [&](int r, int c, float v) {
        if (c < D) bQ[r * DS + c] = fmaxf(v * drop_mul(ds1, r, c), 0.f);
      });
      __syncthreads();
      gemm_packed(MT, NT, DK, Mat{bQ, DS}, w2, [&](int r, int c, float v) {
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
      gemm_packed(MT, (di + 15) >> 4, DK, Mat{bXS, DS}, wl, [&](int r, int c, float v) { if (c < di) bQ[r * DS + c] = v; });
      __syncthreads();
      hin = bQ;
    }
    ln_rows(hin, bQN, L, DS, dout, s_ln + (4 * ly.n_blocks) * 64, s_ln + (4 * ly.n_blocks + 1) * 64);
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

__global__ void __launch_bounds__(512) encoder_bwd_kernel(const EncArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const Dims& ly = a.dm;
  const Geom g = make_geom(a.L, ly.D);
  const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), nw = blockDim.x >> 6, nthr = blockDim.x;
  const int L = g.L, LP = g.LP, D = g.D, DS = g.DS, SLD = g.SLD, NT = g.NT, MT = g.MT, DK = g.DK;
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
      gemm_packed(MT, (di + 15) >> 4, DK, Mat{bX, DS}, wl, [&](int r, int c, float v) { if (c < di) bQ[r * DS + c] = v; });
      __syncthreads();
      lnin = bQ;
    }
    // ---- logits backward.  Pass 1 (one quad per position): dh -> bG, hidden rows staged in bT.
    {
      const int q = tid & 3;
      const bool srfrn = kind == SRFRD_SRFRN;
      for (int t = tid >> 2; t < L; t += nthr >> 2) {
        const float dp = s_dpl[t], dn = s_dnl[t];
        const int pid = s_pid[t], nid = s_nid[t], pf = s_pfk[t], nf = s_nfk[t];
#pragma unroll
        for (int j = 0; j < kQC; ++j) {
          const int c = q + 4 * j;
          if (c < D) {
            float dh = 0.f;
            if (c < dout) {
              bT[t * DS + c] = a.c_hidden[(rowbase + t) * dout + c];
              if (a.d_hidden) dh = a.d_hidden[(rowbase + t) * dout + c];
              if (c < di) {
                if (a.pos_ids) dh += dp * table[(int64_t)pid * di + c];
                if (a.neg_ids) dh += dn * table[(int64_t)nid * di + c];
              } else if (srfrn) {
                if (a.pos_ids) dh += dp * P[ly.off_side + pf * dfk + (c - di)];
                if (a.neg_ids) dh += dn * P[ly.off_side + nf * dfk + (c - di)];
              }
            }
            bG[t * DS + c] = dh;
          }
        }
      }
    }
    __syncthreads();
    // Pass 2 (one wave per position, lane = channel): item-table scatter as whole 4*d_item-byte row segments, the
    // access shape float atomics run at full rate for; row 0 (padding_idx) receives none.
    for (int t = wave; t < L; t += nw) {
      if (lane < di) {
        const float h = bT[t * DS + lane];
        const float dp = s_dpl[t], dn = s_dnl[t];
        const int pid = s_pid[t], nid = s_nid[t];
        if (a.pos_ids && pid != 0 && dp != 0.f) atomicAdd(&a.grad_table[(int64_t)pid * di + lane], dp * h);
        if (a.neg_ids && nid != 0 && dn != 0.f) atomicAdd(&a.grad_table[(int64_t)nid * di + lane], dn * h);
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
    ln_bwd_rows<false>(bG, lnin, bK, bO, L, LP, DS, dout, s_ln + (4 * ly.n_blocks) * 64);
    __syncthreads();
    gemm_tiles<0>(1, (dout + 15) >> 4, LP, OnesRow{}, Mat{bG, DS},
                  [=](int r, int c, float v) { if (r == 0 && c < dout) slab[ly.off_ll_b + c] += v; });
    gemm_tiles<0>(1, (dout + 15) >> 4, LP, OnesRow{}, Mat{bO, DS},
                  [=](int r, int c, float v) { if (r == 0 && c < dout) slab[ly.off_ll_w + c] += v; });
    __syncthreads();
    { lds_f* t_ = bG; bG = bK; bK = t_; }
    if (kind == SRFRD_SRFR) {             // hc = hf Wlc^T + blc
      gemm_slab((di + 15) >> 4, NT, LP, MatT{bG, DS}, MatOnes{bX, DS, D},
                SlabWB{slab + ly.off_lc_w, fold_bias ? slab + ly.off_lc_b : nullptr, di, D, rmw});
      if (!fold_bias) colsum_to_slab(0, bG, DS, L, di, slab + ly.off_lc_b);
      const WFrag wln = load_wfrag(pk(ly.n_blocks * 6, 1), nullptr, 0, NT);
      gemm_packed(MT, NT, (di + 3) & ~3, Mat{bG, DS}, wln, [&](int r, int c, float v) { if (c < D) bT[r * DS + c] = v; });
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
      for (int idx = tid; idx < L * D; idx += nthr) {
        const int t = idx / D, c = idx - t * D;
        bG[t * DS + c] *= s_keep[t];
        bX[t * DS + c] = a.c_save_h1[((int64_t)i * B * L + rowbase + t) * D + c];
      }
      __syncthreads();
      ln_rows(bX, bQN, L, DS, D, s_ln + (4 * i + 2) * 64, s_ln + (4 * i + 3) * 64);      // h2
      for (int idx = tid; idx < LP * D; idx += nthr) {                             // dA2 = drop2'(dy)
        const int t = idx / D, c = idx - t * D;
        bK[t * DS + c] = t < L ? bG[t * DS + c] * drop_mul(ds2, t, c) : 0.f;
      }
      __syncthreads();
      gemm_packed(MT, NT, DK, Mat{bQN, DS}, w1t, [&](int r, int c, float v) {
        if (c < D) bQ[r * DS + c] = fmaxf(v * drop_mul(ds1, r, c), 0.f);                        // r = relu(drop1(a1))
      });
      __syncthreads();
      gemm_slab(NT, NT, LP, MatT{bK, DS}, MatOnes{bQ, DS, D},                                   // dW2 += dA2^T r (+ db2)
                SlabWB{slab + o.c2_w, fold_bias ? slab + o.c2_b : nullptr, D, D, rmw});
      if (!fold_bias) colsum_to_slab(0, bK, DS, L, D, slab + o.c2_b);
      gemm_packed(MT, NT, DK, Mat{bK, DS}, w2n, [&](int r, int c, float v) {
        if (c < D) bV[r * DS + c] = bQ[r * DS + c] > 0.f ? v * keep_scale : 0.f;               // dA1
      });
      __syncthreads();
      gemm_slab(NT, NT, LP, MatT{bV, DS}, MatOnes{bQN, DS, D},                                  // dW1 += dA1^T h2 (+ db1)
                SlabWB{slab + o.c1_w, fold_bias ? slab + o.c1_b : nullptr, D, D, rmw});
      if (!fold_bias) colsum_to_slab(1 % nw, bV, DS, L, D, slab + o.c1_b);
      gemm_packed(MT, NT, DK, Mat{bV, DS}, w1n, [&](int r, int c, float v) { if (c < D) bG[r * DS + c] += v; });   // dh2 = dy + dA1 W1
      __syncthreads();
      ln_bwd_rows<false>(bG, bX, bT, bV, L, LP, DS, D, s_ln + (4 * i + 2) * 64);               // dh1 -> bT
      __syncthreads();
      gemm_tiles<0>(1, NT, LP, OnesRow{}, Mat{bG, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) slab[o.ln2_b + c] += v; });
      gemm_tiles<0>(1, NT, LP, OnesRow{}, Mat{bV, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) slab[o.ln2_w + c] += v; });
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
      ln_rows(bX, bQN, L, DS, D, s_ln + (4 * i + 0) * 64, s_ln + (4 * i + 1) * 64);
      __syncthreads();
      gemm_packed(MT, NT, DK, Mat{bQN, DS}, wq, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v * qscale; });
      gemm_packed(MT, NT, DK, Mat{bX, DS}, wk, [&](int r, int c, float v) { if (c < D) bK[r * DS + c] = v; });
      gemm_packed(MT, NT, DK, Mat{bX, DS}, wv, [&](int r, int c, float v) { if (c < D) bV[r * DS + c] = v; });
      const WFrag won = load_wfrag(pk(i * 6 + 3, 1), nullptr, 0, NT);
      __syncthreads();
      gemm_tiles<1>(MT, MT, DK, Mat{bQ, DS}, MatT{bK, DS}, [&](int r, int c, float v) { S1[r * SLD + c] = v; });
      __syncthreads();
      softmax_rows<false>(S1, L, SLD, LP, dsA);                  // P (dropout NOT folded in: applied on load)
      __syncthreads();
      gemm_tiles<2>(MT, NT, LP, MatDrop{S1, SLD, dsA}, Mat{bV, DS},
                    [&](int r, int c, float v) { if (c < D) bO[r * DS + c] = v; });            // o = drop(P) v
      __syncthreads();
      gemm_slab(NT, NT, LP, MatT{bG, DS}, MatOnes{bO, DS, D},                                   // dWo += dh1^T o (+ dbo)
                SlabWB{slab + o.out_w, fold_bias ? slab + o.out_b : nullptr, D, D, rmw});
      if (!fold_bias) colsum_to_slab(2 % nw, bG, DS, L, D, slab + o.out_b);
      __syncthreads();
      gemm_packed(MT, NT, DK, Mat{bG, DS}, won, [&](int r, int c, float v) { if (c < D) bO[r * DS + c] = v; });    // do = dh1 Wo
      const WFrag wqn = load_wfrag(pk(i * 6 + 0, 1), nullptr, 0, NT);
      const WFrag wkn = load_wfrag(pk(i * 6 + 1, 1), nullptr, 0, NT);
      const WFrag wvn = load_wfrag(pk(i * 6 + 2, 1), nullptr, 0, NT);
      __syncthreads();
      gemm_tiles<1>(MT, MT, DK, Mat{bO, DS}, MatT{bV, DS}, [&](int r, int c, float v) { S2[r * SLD + c] = v; });  // dPd = do v^T
      gemm_tiles<3>(MT, NT, LP, MatDropT{S1, SLD, dsA}, Mat{bO, DS},
                    [&](int r, int c, float v) { if (c < D) bT[r * DS + c] = v; });            // dv = drop(P)^T do
      __syncthreads();
      softmax_bwd_rows(S2, S1, L, SLD, LP, dsA);                 // dS = P * (dP - sum_j dP_j P_j), dP = mask * dPd
      __syncthreads();
      lds_f* dKb = S1;                                           // P is dead: dk overlays it as [LP][DS]
      gemm_tiles<2>(MT, NT, LP, Mat{S2, SLD}, Mat{bK, DS},
                    [&](int r, int c, float v) { if (c < D) bO[r * DS + c] = v * qscale; });   // dq (pre-scale)
      gemm_tiles<3>(MT, NT, LP, MatT{S2, SLD}, Mat{bQ, DS},
                    [&](int r, int c, float v) { if (c < D) dKb[r * DS + c] = v; });           // dk = dS^T q
      __syncthreads();
      gemm_slab(NT, NT, LP, MatT{bO, DS}, MatOnes{bQN, DS, D},                                  // dWq (+ dbq)
                SlabWB{slab + o.in_w, fold_bias ? slab + o.in_b : nullptr, D, D, rmw});
      gemm_slab(NT, NT, LP, MatT{dKb, DS}, MatOnes{bX, DS, D},                                  // dWk (+ dbk)
                SlabWB{slab + o.in_w + D * D, fold_bias ? slab + o.in_b + D : nullptr, D, D, rmw});
      gemm_slab(NT, NT, LP, MatT{bT, DS}, MatOnes{bX, DS, D},                                   // dWv (+ dbv)
                SlabWB{slab + o.in_w + 2 * D * D, fold_bias ? slab + o.in_b + 2 * D : nullptr, D, D, rmw});
      if (!fold_bias) {
        colsum_to_slab(0, bO, DS, L, D, slab + o.in_b);
        colsum_to_slab(1 % nw, dKb, DS, L, D, slab + o.in_b + D);
        colsum_to_slab(2 % nw, bT, DS, L, D, slab + o.in_b + 2 * D);
      }
      gemm_packed(MT, NT, DK, Mat{bO, DS}, wqn, [&](int r, int c, float v) { if (c < D) bG[r * DS + c] += v; });   // dLN1 = dh1 + dq Wq
      gemm_packed(MT, NT, DK, Mat{dKb, DS}, wkn, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] = v; });   // dx  = dk Wk
      gemm_packed(MT, NT, DK, Mat{bT, DS}, wvn, [&](int r, int c, float v) { if (c < D) bQ[r * DS + c] += v; });   //     + dv Wv
      __syncthreads();
      ln_bwd_rows<true>(bG, bX, bQ, S2, L, LP, DS, D, s_ln + (4 * i + 0) * 64);                //     + LN1 bwd
      __syncthreads();
      gemm_tiles<0>(1, NT, LP, OnesRow{}, Mat{bG, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) slab[o.ln1_b + c] += v; });
      gemm_tiles<0>(1, NT, LP, OnesRow{}, Mat{S2, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) slab[o.ln1_w + c] += v; });
      __syncthreads();
      lds_f* t_ = bG; bG = bQ; bQ = t_;
      tap(a, b, tb + 1, bG, L, D, DS);
    }

    // ---- embedding backward: item rows (atomics), position table, side channel
    {
      const DropSite dsE = drop_site(a.drop_on && is_sas, seed, SITE_EMB, seq, a.drop_thr, a.drop_scale);
      for (int t = wave; t < L; t += nw) {
        if (lane < D) {
          float gv = bG[t * DS + lane] * s_keep[t];
          if (is_sas) gv *= drop_mul(dsE, t, lane);
          const int id = s_in[t];
          if (lane < di) {
            if (id != 0) atomicAdd(&a.grad_table[(int64_t)id * di + lane], is_sas ? gv * sqrtD : gv);
            slab[ly.off_pos + t * di + lane] += gv;
          }
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

// ================================================================================================
// host side
// ================================================================================================
static int g_num_cu = 0;
static int num_cu() {
  if (g_num_cu == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0)
      g_num_cu = prop.multiProcessorCount;
    else
      g_num_cu = 256;
  }
  return g_num_cu;
}

// block size override for tuning runs (multiple of 64, <= 512)
static int env_threads(const char* name, int dflt) {
  const char* e = getenv(name);
  if (!e) return dflt;
  const int v = atoi(e);
  return (v >= 64 && v <= 512 && (v & 63) == 0) ? v : dflt;
}

static int fill_args(EncArgs& a, const srfrd_layout* lay, const float* item_table, const float* dense, const float* packed,
                     const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids, const int64_t* pos_fake,
                     const int64_t* neg_ids, const int64_t* neg_fake, int B, int L, double dropout_p, uint32_t seed,
                     const uint32_t* seed_dev, int64_t seq_index0) {
  if (!lay || !item_table || !dense || !packed || !input_ids || B <= 0 || L <= 0) return SRFRD_E_ARG;
  if (lay->D > SRFRD_MAX_D || lay->n_heads != 1 || lay->n_blocks > SRFRD_MAX_BLOCKS) return SRFRD_E_UNSUPPORTED;
  if (L > lay->max_len) return SRFRD_E_ARG;
  if (dropout_p < 0.0 || dropout_p >= 1.0) return SRFRD_E_ARG;
  if (lay->kind == SRFRD_SRFRN && ((pos_ids && !pos_fake) || (neg_ids && !neg_fake))) return SRFRD_E_ARG;
  Dims& d = a.dm;
  d.kind = lay->kind; d.d_item = lay->d_item; d.d_fake = lay->d_fake; d.D = lay->D; d.d_out = lay->d_out;
  d.n_labels = lay->n_labels; d.n_blocks = lay->n_blocks;
  d.off_pos = (int)lay->off_pos; d.off_side = (int)lay->off_side;
  d.blk0 = lay->n_blocks > 0 ? (int)lay->blk[0].ln1_w : 0;
  d.blk_stride = blk_stride_of(lay->D);
  for (int i = 0; i < lay->n_blocks; ++i) {        // the kernels recompute block offsets arithmetically: check the table agrees
    const BlkOff o = blk_off(d.blk0 + i * d.blk_stride, lay->D);
    const srfrd_block_off& t = lay->blk[i];
    if (t.ln1_w != o.ln1_w || t.ln1_b != o.ln1_b || t.in_w != o.in_w || t.in_b != o.in_b || t.out_w != o.out_w ||
        t.out_b != o.out_b || t.ln2_w != o.ln2_w || t.ln2_b != o.ln2_b || t.c1_w != o.c1_w || t.c1_b != o.c1_b ||
        t.c2_w != o.c2_w || t.c2_b != o.c2_b)
      return SRFRD_E_ARG;
  }
  d.off_lc_w = (int)lay->off_lc_w; d.off_lc_b = (int)lay->off_lc_b; d.off_ll_w = (int)lay->off_ll_w; d.off_ll_b = (int)lay->off_ll_b;
  d.n_dense = (int)lay->n_dense;
  a.table = item_table;
  a.dense = dense;
  a.packed = packed;
  a.in_ids = input_ids; a.fk_ids = fake_ids; a.pos_ids = pos_ids; a.pos_fk = pos_fake; a.neg_ids = neg_ids; a.neg_fk = neg_fake;
  a.B = B; a.L = L;
  a.seed = seed; a.seed_dev = seed_dev;
  a.drop_on = dropout_p > 0.0;
  double thr = dropout_p * 4294967296.0;
  a.drop_thr = thr >= 4294967295.0 ? 0xFFFFFFFFu : (uint32_t)thr;
  a.drop_scale = (float)(1.0 / (1.0 - dropout_p));
  a.seq0 = seq_index0;
  a.qscale = (float)sqrt(1.0 / (double)(lay->D / lay->n_heads));
  return 0;
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

extern "C" int srfrd_bwd_grid(int B) {
  if (B <= 0) return SRFRD_E_ARG;
  const int cu = num_cu();
  return B < cu ? B : cu;
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
  static int64_t s_attr = 0;
  if (lds > s_attr) {
    if (hipFuncSetAttribute((const void*)encoder_fwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SRFRD_E_DEVICE;
    s_attr = lds;
  }
  const int per_cu = (int)(kLdsLimit / lds) > 2 ? 2 : (int)(kLdsLimit / lds);
  int grid = num_cu() * (per_cu < 1 ? 1 : per_cu);
  if (grid > B) grid = B;
  hipLaunchKernelGGL(encoder_fwd_kernel, dim3(grid), dim3(env_threads("SRFRD_FWD_THREADS", 256)), (size_t)lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
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
  static int64_t s_attr = 0;
  if (lds > s_attr) {
    if (hipFuncSetAttribute((const void*)encoder_bwd_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
      return SRFRD_E_DEVICE;
    s_attr = lds;
  }
  const int grid = srfrd_bwd_grid(B);
  hipLaunchKernelGGL(encoder_bwd_kernel, dim3(grid), dim3(env_threads("SRFRD_BWD_THREADS", 512)), (size_t)lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
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
