// Device-side building blocks of the fused encoder kernels (gfx950 / CDNA4, wave64).
//
// Everything a sequence needs lives in LDS as row-major fp32 matrices [LP][DS]:
//   LP = L rounded up to 16 (MFMA row tiles), DK = D rounded up to 4 (MFMA k-steps), DS = DK + 2.
// DS == 2 (mod 4) makes the 16-row x 2-k footprint of one v_mfma_f32_16x16x4_f32 A/B operand read
// (ds_read_b32, two 32-lane groups) hit 32 distinct banks: row*DS mod 32 walks the 16 even banks.
// Score matrices are [LP][SLD], SLD = LP + 2 (same property).
//
// GEMMs run on the fp32-input matrix cores (v_mfma_f32_16x16x4_f32): bit-for-bit a k-ordered fmaf
// chain, so the 1e-4 fp32 parity bar holds while the VALU stays free for LayerNorm / softmax / dropout.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/srfrd_hip.h"
#include "srfrd_rng.h"

// Diagnostic build (tools/race_skew.py, -DSRFRD_SKEW=<wave mask>): every workgroup barrier is followed by a ~2 us sleep of
// the waves whose bit is set in the mask.  Results must not change: a wave that reads a buffer early in a barrier interval
// while another wave writes it late in the same interval (a missing barrier that normally "wins" by a wide margin) turns
// into a test failure once the reader is the delayed one.
#ifdef SRFRD_SKEW
__device__ __forceinline__ void srfrd_skewed_barrier() {
  __syncthreads();
  // (wave-uniform test on a scalar register: a per-lane condition compiles to an EXEC mask around the sleeps, which
  // execute - they are scalar instructions - whatever the mask holds: every wave would sleep)
  const int w_ = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6)) & 15;
  if ((SRFRD_SKEW >> w_) & 1) {
    __builtin_amdgcn_s_sleep(40);
    __builtin_amdgcn_s_sleep(40);
  }
}
#define __syncthreads() srfrd_skewed_barrier()
#endif

// Two builds of the same sources: the default keeps a sequence's working set in LDS; with SRFRD_BUF_GLOBAL the same
// kernels address a per-workgroup scratch in global memory instead (sequences too long for the 160 KiB LDS: slower,
// but every shape runs on the GPU).  The variants live in different namespaces so both link into one library.
#ifdef SRFRD_BUF_GLOBAL
#define SRFRD_NS srfrd_long
#else
#define SRFRD_NS srfrd
#endif

namespace SRFRD_NS {
using namespace srfrd;   // srfrd_rng.h

typedef float f32x4 __attribute__((ext_vector_type(4)));
#ifdef SRFRD_BUF_GLOBAL
typedef float lds_f;
typedef int lds_i;
#else
// LDS-resident matrices are addressed through explicit address_space(3) pointers: a generic `float*` that the
// compiler cannot trace back to __shared__ becomes flat_load/flat_store with 64-bit address arithmetic.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) int lds_i;
#endif

constexpr float kLnEps = 1e-8f;   // reference SRFR_model.py:77,80,86
constexpr int kLdsLimit = 160 * 1024;
constexpr int kSlack = 64;        // floats of slack behind the last matrix (tile overreads stay in-bounds)

struct Geom {
  int L, LP, D, DK, DS, SLD, NT, MT;
};

__host__ __device__ __forceinline__ Geom make_geom(int L, int D) {
  Geom g;
  g.L = L;
  g.LP = (L + 15) & ~15;
  g.D = D;
  g.DK = (D + 3) & ~3;
  g.DS = g.DK + 2;
  g.SLD = g.LP + 2;
  g.NT = (D + 15) >> 4;
  g.MT = g.LP >> 4;
  return g;
}

__host__ __device__ __forceinline__ int imax(int a, int b) { return a > b ? a : b; }

// bf16 <-> f32 (round to nearest even on the way down: hipcc lowers the cast to v_cvt_pk_bf16_f32, NaN stays NaN)
__device__ __forceinline__ float bf16_to_f32(uint16_t h) { return __builtin_bit_cast(float, (uint32_t)h << 16); }
__device__ __forceinline__ uint16_t f32_to_bf16(float x) { return __builtin_bit_cast(uint16_t, (__bf16)x); }
// The item table as the gathers see it: the fp32 parameter, or its bf16 shadow (srfrd_layout::table_bf16) at half the bytes
struct ItemTable {
  const float* f;
  const uint16_t* h;
  __device__ __forceinline__ float operator()(int64_t i) const { return h != nullptr ? bf16_to_f32(h[i]) : f[i]; }
};

// An id as the kernels use it: clamped (in 64 bits, before the narrowing) into [0, hi], so that no input - an item id
// beyond the model's table, a negative id, a fake id above 2 - can turn a gather into an out-of-bounds read or a
// gradient scatter into a write over a neighbouring parameter.  Out-of-range ids are REPORTED by srfrd_check_ids (the
// reference's nn.Embedding raises IndexError for them); the clamp only keeps the launch memory-safe until then.
__device__ __forceinline__ int clamp_id(int64_t v, int hi) {
  return (int)(v < 0 ? 0 : (v > (int64_t)hi ? (int64_t)hi : v));
}

// Compact, all-int32 view of srfrd_layout passed to the kernels (the full descriptor with its 8 x 12 int64 block
// table costs hundreds of SGPRs; block offsets are affine in the block index, so they are recomputed instead).
struct Dims {
  int kind, d_item, d_fake, D, d_out, n_labels, n_blocks, n_items, n_heads;
  int off_pos, off_side, blk0, blk_stride, off_lc_w, off_lc_b, off_ll_w, off_ll_b, n_dense;
};
struct BlkOff {
  int ln1_w, ln1_b, in_w, in_b, out_w, out_b, ln2_w, ln2_b, c1_w, c1_b, c2_w, c2_b;
};
__host__ __device__ __forceinline__ BlkOff blk_off(int base, int D) {
  const int DD = D * D;
  BlkOff o;
  o.ln1_w = base;          o.ln1_b = o.ln1_w + D;
  o.in_w = o.ln1_b + D;    o.in_b = o.in_w + 3 * DD;
  o.out_w = o.in_b + 3 * D; o.out_b = o.out_w + DD;
  o.ln2_w = o.out_b + D;   o.ln2_b = o.ln2_w + D;
  o.c1_w = o.ln2_b + D;    o.c1_b = o.c1_w + DD;
  o.c2_w = o.c1_b + D;     o.c2_b = o.c2_w + DD;
  return o;
}
__host__ __device__ __forceinline__ int blk_stride_of(int D) { return 10 * D + 6 * D * D; }
// LDS cache of the LayerNorm parameters: vector v of block i at (4 i + v) * 64, the last LayerNorm's two after them
__host__ __device__ __forceinline__ int ln_cache_floats(int n_blocks) { return (4 * n_blocks + 2) * 64; }

// LDS floats: forward = XS + 4 matrices; backward = 8 matrices + 2 score matrices; + slack + per-row scalars
__host__ __device__ __forceinline__ int64_t fwd_lds_floats(const Geom& g, int n_blocks) {
  return (int64_t)imax(g.LP * g.DS, g.LP * g.SLD) + 4ll * g.LP * g.DS + kSlack + 4ll * g.LP + 64 + ln_cache_floats(n_blocks);
}
__host__ __device__ __forceinline__ int64_t bwd_lds_floats(const Geom& g, int n_blocks) {
  return 8ll * g.LP * g.DS + 2ll * imax(g.LP * g.SLD, g.LP * g.DS) + 2 * kSlack + 10ll * g.LP + 64 + 2ll * ln_cache_floats(n_blocks);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

// ---------------------------------------------------------------------------------------------
// operand accessors: A(row, k) and B(k, col)
// ---------------------------------------------------------------------------------------------
struct Mat {            // element (r, c) of a row-major LDS matrix
  const lds_f* p;
  int ld;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[r * ld + c]; }
};
struct MatT {           // element (r, c) of the transpose of a row-major LDS matrix
  const lds_f* p;
  int ld;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[c * ld + r]; }
};
struct MatCols {        // row-major LDS matrix with the columns outside [c0, c1) read as zero: one attention head's slice
  const lds_f* p;       // of q / do as the A operand of a full-width product (multi-head path of the generic kernels)
  int ld, c0, c1;
  __device__ __forceinline__ float operator()(int r, int c) const { return (c >= c0 && c < c1) ? p[r * ld + c] : 0.f; }
};
struct MatTPos {        // transpose of an LDS matrix with negative entries read as zero: the sign-coded attention
  const lds_f* p;       // probabilities (dropped entries stored negated) read as the kept ones
  int ld;
  __device__ __forceinline__ float operator()(int r, int c) const { return fmaxf(p[c * ld + r], 0.f); }
};

// Weights pre-swizzled into MFMA B-fragment order by srfrd_pack_weights: for strip nt and 16-deep k-chunk kc, lane l
// holds the float4 {B(16kc + 4s + (l>>4), 16nt + (l&15))}_{s=0..3}, zero outside the matrix.  One coalesced
// global_load_dwordx4 per lane per chunk replaces four predicated scalar gathers, and a whole strip (<= 4 chunks
// at D <= 64) is requested up front, so a weight GEMM exposes one L2 latency instead of one per k-chunk.
constexpr int kPackKC = 4;                     // k-chunks per strip (K <= 64)
constexpr int kPackFloats = 4 * kPackKC * 256; // floats per packed matrix-form (4 strips)
struct PackedB {
  const float4* p;
  __device__ __forceinline__ float4 chunk(int nt, int kc, int lane) const { return p[(nt * kPackKC + kc) * 64 + lane]; }
};

// dW epilogue target: a (R x C) weight-gradient block at float offset `w` of this workgroup's slab (+ the bias
// gradient at offset `b`, or -1: it is column C of the product, fed by the ones column the X operand carries);
// `rmw == 0` for the first sequence of the workgroup (store), 1 afterwards (add).  Offsets, not pointers: the
// stores take the scalar-base + 32-bit-offset form.
struct SlabWB {
  float* base;
  int w, b;
  int R, C, rmw;
};

// ---------------------------------------------------------------------------------------------
// tiled GEMM on v_mfma_f32_16x16x4_f32.   C(16 mt.., 16 nt..) = sum_k A(row, k) * B(k, col)
// A operand: lane l holds A[m0 + (l & 15)][k + (l >> 4)];  B operand: B[k + (l >> 4)][n0 + (l & 15)]
// C/D: register r of lane l is element (m0 + 4 * (l >> 4) + r, n0 + (l & 15))
// TRI: 0 full; 1 skip tiles with nt > mt (lower-triangular C); 2 A is lower-triangular (k beyond the group's last
//      diagonal block is skipped; inside the range the stored zeros of A do the masking); 3 A is the transpose of a
//      lower-triangular matrix (k before the group's first diagonal block is skipped).
//
// Work split: a wave owns one 16-column strip (n-tile) and walks its row tiles in groups of G = 4 / 2 / 1, so a
// B fragment (a weight column block from global/L1, or an LDS matrix) is loaded once per k-step and feeds G
// independent accumulator chains (the 16x16x4 MFMA has a 40-cycle dependent latency against a 32-cycle issue
// interval).  The inner body is branch-free: 4 B loads + 4G A loads are issued together, then 4G MFMAs.
// When there are more waves than strips, the waves sharing a strip take interleaved row tiles.
// ---------------------------------------------------------------------------------------------
// When k is the ROW index of both underlying LDS matrices (A a transposed view, B a plain matrix: every dW GEMM, dv, dk,
// the ones-row column sums), the lanes of one operand fetch read 16 consecutive floats of rows k, k+1, k+2, k+3: with the
// row strides that keep the (row, k) views conflict-free (54 and 66 floats: 22 and 2 banks) those runs overlap in the
// 32 banks.  The k-steps of a 16-deep chunk may be taken in any order, so such GEMMs pair the rows that are 8 apart in
// a half-wave (step s reads rows s, s+8 | s+4, s+12): 8 x 22 and 8 x 2 are both 16 banks - the two runs tile the banks.
template <class AL> struct k_is_row { static constexpr bool value = false; };
template <> struct k_is_row<MatT> { static constexpr bool value = true; };
template <> struct k_is_row<MatTPos> { static constexpr bool value = true; };
template <class AL, class BL> struct k_perm { static constexpr bool value = false; };
template <class AL> struct k_perm<AL, Mat> { static constexpr bool value = k_is_row<AL>::value; };

template <int G, class AL, class BL>
__device__ __forceinline__ void mma_group(f32x4 (&acc)[G], const AL& a, const BL& b, int mrow, int mstride, int ncol,
                                          int k0, int k1, int lq) {
  constexpr bool KP = k_perm<AL, BL>::value;
  const int kofs = KP ? ((lq & 1) << 3) + ((lq >> 1) << 2) : lq;      // this lane's k within step 0 of a chunk
  constexpr int kstep = KP ? 1 : 4;                                   // ... and the distance to the next step
  const int nfull = (k1 - k0) >> 4;      // full 16-deep chunks; the fragments of chunk c+1 are requested before the
  if (nfull > 0) {                       // MFMAs of chunk c are issued (double buffering), so LDS latency is hidden
    float bv[4], av[4][G];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      bv[s] = b(k0 + kstep * s + kofs, ncol);
#pragma unroll
      for (int j = 0; j < G; ++j) av[s][j] = a(mrow + j * mstride, k0 + kstep * s + kofs);
    }
    for (int c = 0; c < nfull; ++c) {
      const int kn = k0 + (min(c + 1, nfull - 1) << 4);     // the last pass re-reads its own chunk (harmless)
      float bn[4], an[4][G];
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bn[s] = b(kn + kstep * s + kofs, ncol);
#pragma unroll
        for (int j = 0; j < G; ++j) an[s][j] = a(mrow + j * mstride, kn + kstep * s + kofs);
      }
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < G; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][j], bv[s], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s) {
        bv[s] = bn[s];
#pragma unroll
        for (int j = 0; j < G; ++j) av[s][j] = an[s][j];
      }
    }
  }
  for (int k = k0 + (nfull << 4); k < k1; k += 4) {
    const float bt = b(k + lq, ncol);
    float at[G];
#pragma unroll
    for (int j = 0; j < G; ++j) at[j] = a(mrow + j * mstride, k + lq);
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(at[j], bt, acc[j], 0, 0, 0);
  }
}

template <int TRI, int G, class AL, class BL, class EPI>
__device__ __forceinline__ void gemm_group(int mt, int mgroups, int n0, int k_end, const AL& a, const BL& b, const EPI& epi,
                                           int li, int lq) {
  int k0 = 0, k1 = k_end;
  if (TRI == 2) k1 = min(k_end, (mt + (G - 1) * mgroups + 1) << 4);
  if (TRI == 3) k0 = mt << 4;
  f32x4 acc[G];
#pragma unroll
  for (int j = 0; j < G; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  mma_group<G>(acc, a, b, (mt << 4) + li, mgroups << 4, n0 + li, k0, k1, lq);
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int r0 = ((mt + j * mgroups) << 4) + (lq << 2), c = n0 + li;
    epi(r0 + 0, c, acc[j][0]);
    epi(r0 + 1, c, acc[j][1]);
    epi(r0 + 2, c, acc[j][2]);
    epi(r0 + 3, c, acc[j][3]);
  }
}

constexpr int kMW = 4;     // row tiles a wave accumulates at once

// tiles of this wave in a (m_tiles x n_tiles) GEMM with m_tiles <= kMW * mgroups: unit, strip, first tile, count
struct WaveTiles {
  int active, nt, g, mgroups, n;
};
__device__ __forceinline__ WaveTiles wave_tiles(int nw, int m_tiles, int n_tiles) {
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  WaveTiles t;
  t.mgroups = nw > n_tiles ? nw / n_tiles : 1;
  t.active = wave < n_tiles * t.mgroups;
  t.nt = wave % n_tiles;
  t.g = wave / n_tiles;
  // (when the row tiles divide evenly - every specialised shape - the count is a constant and the G = 1 / 3 code
  // paths of the callers fold away)
  if (m_tiles % t.mgroups == 0) t.n = t.active ? m_tiles / t.mgroups : 0;
  else t.n = t.active && t.g < m_tiles ? (m_tiles - t.g + t.mgroups - 1) / t.mgroups : 0;
  return t;
}

// Old slab values of the (<= kMW) row tiles a wave owns in a dW GEMM.  slab_preload() requests them at the start of
// the phase (clamped offsets, no branch per element); they become the accumulators' initial value, so the
// read-modify-write costs the wave nothing but the registers: the loads land under the GEMMs that precede the dW
// GEMM in the phase, the result leaves as plain stores.  (The alternatives measured slower: load + add + store
// after the MFMA chain makes every store wait - vmcnt retires in order - for the one before it; no-return float
// atomics execute at the memory side at ~1.3 TB/s chip-wide, 13 us of the backward at 31 MB per launch.)
struct SlabPre {
  float v[kMW][4];
};
struct SlabCol {        // this lane's column of the target: offset of row 0, row stride, validity
  int col0, rs;
  bool ok;
};
__device__ __forceinline__ SlabCol slab_col(const SlabWB& sl, int c) {
  const bool in_w = c < sl.C, in_b = c == sl.C && sl.b >= 0;
  SlabCol s;
  s.col0 = in_w ? sl.w + c : (in_b ? sl.b : sl.w);
  s.rs = in_w ? sl.C : (in_b ? 1 : 0);
  s.ok = in_w || in_b;
  return s;
}
__device__ __forceinline__ SlabPre slab_preload(int nw, int m_tiles, int n_tiles, const SlabWB& sl) {
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const WaveTiles t = wave_tiles(nw, m_tiles, n_tiles);
  const SlabCol sc = slab_col(sl, (t.nt << 4) + li);
  SlabPre p;
#pragma unroll
  for (int j = 0; j < kMW; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) p.v[j][e] = 0.f;
  if (sl.rmw) {
#pragma unroll
    for (int j = 0; j < kMW; ++j)
      if (j < t.n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const int r = ((t.g + j * t.mgroups) << 4) + (lq << 2) + e;
          // clamped, unconditional: an element outside the target starts from some in-range value, and since it is
          // never stored that is as good as zero (a select here made hipcc wait for each load before the next)
          p.v[j][e] = sl.base[(unsigned)(sc.col0 + min(r, sl.R - 1) * sc.rs)];     // (unsigned: scalar base + 32-bit offset addressing)
        }
      }
  }
  return p;
}
// Makes the wave wait for a preload HERE (before the first slab store of the phase): a first use behind another dW
// GEMM's stores would wait for their acknowledgement as well.
__device__ __forceinline__ void slab_ready(SlabPre& p) {
#pragma unroll
  for (int j = 0; j < kMW; ++j)
#pragma unroll
    for (int e = 0; e < 4; ++e) asm volatile("" : "+v"(p.v[j][e]));
}

template <int G, int J0, class AL, class BL>
__device__ __forceinline__ void gemm_group_slab(int mt, int mgroups, int n0, int k_end, const AL& a, const BL& b,
                                                const SlabWB& sl, const SlabPre& pre, int li, int lq) {
  f32x4 acc[G];
#pragma unroll
  for (int j = 0; j < G; ++j) acc[j] = f32x4{pre.v[J0 + j][0], pre.v[J0 + j][1], pre.v[J0 + j][2], pre.v[J0 + j][3]};
  mma_group<G>(acc, a, b, (mt << 4) + li, mgroups << 4, n0 + li, 0, k_end, lq);
  // element (r, c): r < R rows of the weight gradient; column C is the bias gradient
  const SlabCol sc = slab_col(sl, n0 + li);
  if (sc.ok) {
#pragma unroll
    for (int j = 0; j < G; ++j)
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        const int r = ((mt + j * mgroups) << 4) + (lq << 2) + e;
        if (r < sl.R) sl.base[(unsigned)(sc.col0 + r * sc.rs)] = acc[j][e];
      }
  }
}

// m_tiles <= kMW * mgroups and n_tiles <= number of waves (true for every weight-gradient GEMM: <= 4 x 4 tiles)
template <class AL, class BL>
__device__ __forceinline__ void gemm_slab(int nw, int m_tiles, int n_tiles, int k_end, AL a, BL b, SlabWB sl, const SlabPre& pre) {
  const int lane = threadIdx.x & 63, li = lane & 15, lq = lane >> 4;
  const WaveTiles t = wave_tiles(nw, m_tiles, n_tiles);
  const int n0 = t.nt << 4;
  if (t.n >= 4) gemm_group_slab<4, 0>(t.g, t.mgroups, n0, k_end, a, b, sl, pre, li, lq);
  else {
    if (t.n >= 2) gemm_group_slab<2, 0>(t.g, t.mgroups, n0, k_end, a, b, sl, pre, li, lq);
    if (t.n == 3) gemm_group_slab<1, 2>(t.g + 2 * t.mgroups, t.mgroups, n0, k_end, a, b, sl, pre, li, lq);
    if (t.n == 1) gemm_group_slab<1, 0>(t.g, t.mgroups, n0, k_end, a, b, sl, pre, li, lq);
  }
}
template <class AL, class BL>
__device__ __forceinline__ void gemm_slab(int nw, int m_tiles, int n_tiles, int k_end, AL a, BL b, SlabWB sl) {
  SlabPre pre = slab_preload(nw, m_tiles, n_tiles, sl);
  gemm_slab(nw, m_tiles, n_tiles, k_end, a, b, sl, pre);
}

// Weight fragments of the strip this wave owns in a packed GEMM, requested ahead of use (one phase early where the
// call site allows) so that the L2 latency of the weights never sits on the critical path.
struct WFrag {
  float4 q[kPackKC];
  float bias;
};
// (tid_: the caller's - possibly laundered - copy of threadIdx.x: a load whose address derives from an opaque value cannot be
// hoisted out of the caller's loop, where it would pin 53 registers for the loop's whole life)
__device__ __forceinline__ WFrag load_wfrag(PackedB b, const float* bias, int nbias, int n_tiles, int tid_) {
  const int lane = tid_ & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid_ >> 6);
  const int nt = wave % n_tiles;
  WFrag f;
#pragma unroll
  for (int kc = 0; kc < kPackKC; ++kc) f.q[kc] = b.chunk(nt, kc, lane);
  const int c = (nt << 4) + (lane & 15);
  f.bias = (bias != nullptr && c < nbias) ? bias[c] : 0.f;
  return f;
}
__device__ __forceinline__ WFrag load_wfrag(PackedB b, const float* bias, int nbias, int n_tiles) {
  return load_wfrag(b, bias, nbias, n_tiles, (int)threadIdx.x);
}

// packed-weight GEMM: C = A * Bpacked (+ bias[col]); k_end <= 64, a multiple of 4.  Exactly k_end / 4 MFMA k-steps are
// issued (13 at D = 50, not 16): the step count per 16-deep chunk is wave-uniform and, in the specialised kernels, a
// compile-time constant, so the chunk bodies stay branch-free there.
template <int G, class AL, class EPI>
__device__ __forceinline__ void gemm_group_packed(int mt, int mgroups, int nt, int ksteps, const AL& a, const WFrag& w,
                                                  const EPI& epi, int lane) {
  const int li = lane & 15, lq = lane >> 4;
  const int n0 = nt << 4;
  f32x4 acc[G];
#pragma unroll
  for (int j = 0; j < G; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
  const int mrow = (mt << 4) + li, mstride = mgroups << 4;
  float av[4][G];
#pragma unroll
  for (int s = 0; s < 4; ++s)
    if (s < ksteps)
#pragma unroll
      for (int j = 0; j < G; ++j) av[s][j] = a(mrow + j * mstride, 4 * s + lq);
#pragma unroll
  for (int kc = 0; kc < kPackKC; ++kc) {
    const int nst = ksteps - 4 * kc;           // k-steps of this chunk (>= 4: full)
    if (nst > 0) {
      const int nnext = nst - 4;               // k-steps of the next chunk
      float an[4][G];
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < nnext)
#pragma unroll
          for (int j = 0; j < G; ++j) an[s][j] = a(mrow + j * mstride, ((kc + 1) << 4) + 4 * s + lq);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
        if (s < nst)
#pragma unroll
          for (int j = 0; j < G; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][j], w.q[kc][s], acc[j], 0, 0, 0);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int j = 0; j < G; ++j) av[s][j] = an[s][j];
    }
  }
#pragma unroll
  for (int j = 0; j < G; ++j) {
    const int r0 = ((mt + j * mgroups) << 4) + (lq << 2), c = n0 + li;
    epi(r0 + 0, c, acc[j][0] + w.bias);
    epi(r0 + 1, c, acc[j][1] + w.bias);
    epi(r0 + 2, c, acc[j][2] + w.bias);
    epi(r0 + 3, c, acc[j][3] + w.bias);
  }
}

// n_tiles <= number of waves (every wave owns at most one strip: the one load_wfrag fetched for it)
template <class AL, class EPI>
__device__ __forceinline__ void gemm_packed(int nw, int m_tiles, int n_tiles, int k_end, AL a, const WFrag& w, EPI epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int mgroups = nw > n_tiles ? nw / n_tiles : 1;
  if (wave >= n_tiles * mgroups) return;
  const int nt = wave % n_tiles, g = wave / n_tiles;
  const int nchunks = k_end >> 2;      // MFMA k-steps (k_end is a multiple of 4)
  int mt = g;
  while (mt + 3 * mgroups < m_tiles) {
    gemm_group_packed<4>(mt, mgroups, nt, nchunks, a, w, epi, lane);
    mt += 4 * mgroups;
  }
  if (mt + mgroups < m_tiles) {
    gemm_group_packed<2>(mt, mgroups, nt, nchunks, a, w, epi, lane);
    mt += 2 * mgroups;
  }
  if (mt < m_tiles) gemm_group_packed<1>(mt, mgroups, nt, nchunks, a, w, epi, lane);
}

template <int TRI, class AL, class BL, class EPI>
__device__ __forceinline__ void gemm_tiles(int nw, int m_tiles, int n_tiles, int k_end, AL a, BL b, EPI epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int li = lane & 15, lq = lane >> 4;
  const int mgroups = nw > n_tiles ? nw / n_tiles : 1;
  const int units = n_tiles * mgroups;
  for (int u = wave; u < units; u += nw) {
    const int nt = u % n_tiles, g = u / n_tiles;
    const int n0 = nt << 4;
    if (TRI >= 2 && mgroups == 2 && m_tiles == 4) {
      // triangular A, four row tiles, two waves per strip: tile t runs (t + 1) or (4 - t) sixteen-deep chunks, so the
      // folded pairs {0, 3} and {1, 2} cost the same, where the strided pairs {0, 2} / {1, 3} sharing one k-range cost
      // 13 + 13 against 9 + 9 k-steps (dv, dk) or 12 + 12 against 16 + 16 (dq) - the busier wave sets the phase
      gemm_group<TRI, 1>(g, mgroups, n0, k_end, a, b, epi, li, lq);
      gemm_group<TRI, 1>(3 - g, mgroups, n0, k_end, a, b, epi, li, lq);
      continue;
    }
    int mt = g;
    if (TRI == 1 && mt < nt) mt += ((nt - mt + mgroups - 1) / mgroups) * mgroups;   // first own row tile on/below the diagonal
    while (mt + 3 * mgroups < m_tiles) {
      gemm_group<TRI, 4>(mt, mgroups, n0, k_end, a, b, epi, li, lq);
      mt += 4 * mgroups;
    }
    if (mt + mgroups < m_tiles) {
      gemm_group<TRI, 2>(mt, mgroups, n0, k_end, a, b, epi, li, lq);
      mt += 2 * mgroups;
    }
    if (mt < m_tiles) gemm_group<TRI, 1>(mt, mgroups, n0, k_end, a, b, epi, li, lq);
  }
}

// ---------------------------------------------------------------------------------------------
// row-wise ops: kRL (4 or 8) lanes per row, lane q owns columns q, q + kRL, q + 2 kRL, ...  With four, a 256-thread
// workgroup therefore covers 64 rows per pass, every row's loads are in flight at once, and the reductions are two
// DPP quad_perm adds (VALU rate) instead of six dependent ds_bpermute shuffles per row per reduction.
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ float quad_sum(float v) {
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));  // lane ^ 1
  v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true));  // lane ^ 2
  return v;
}
__device__ __forceinline__ float quad_max(float v) {
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true)));
  v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xF, 0xF, true)));
  return v;
}

// Lanes that share a row in the row-wise passes: 4 (one DPP quad) or 8 (two quads, one more row_half_mirror step).  With
// L = 50 rows, four lanes per row keep only 200 of a workgroup's 512 lanes (waves 0-3, one per SIMD) busy; eight put
// two half-as-long waves on every SIMD.
#ifndef SRFRD_ROW_LANES
#define SRFRD_ROW_LANES 8
#endif
constexpr int kRL = SRFRD_ROW_LANES;
__device__ __forceinline__ float row_sum(float v) {
  v = quad_sum(v);
  if (kRL == 8) v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true));   // lane ^ 7
  return v;
}
__device__ __forceinline__ float row_max(float v) {
  v = quad_max(v);
  if (kRL == 8) v = fmaxf(v, __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xF, 0xF, true)));
  return v;
}
constexpr int kQC = SRFRD_MAX_D / kRL;    // columns per lane (D <= 64)

// Y[r] = LayerNorm(X[r]) for r < rows   (biased variance, eps inside the sqrt: torch.nn.LayerNorm)
__device__ __forceinline__ void ln_rows(int nw, const lds_f* X, lds_f* Y, int rows, int ld, int D, const lds_f* w,
                                        const lds_f* bia) {
  const int q = threadIdx.x & (kRL - 1), rpp = (nw << 6) / kRL;
  const float invD = 1.0f / (float)D;
  float wl[kQC], bl[kQC];
#pragma unroll
  for (int j = 0; j < kQC; ++j) {
    const int c = q + kRL * j;
    wl[j] = c < D ? w[c] : 0.f;
    bl[j] = c < D ? bia[c] : 0.f;
  }
  for (int r = threadIdx.x / kRL; r < rows; r += rpp) {
    float x[kQC];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kQC; ++j) {
      const int c = q + kRL * j;
      x[j] = c < D ? X[r * ld + c] : 0.f;
      s += x[j];
    }
    const float mu = row_sum(s) * invD;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < kQC; ++j) {
      const int c = q + kRL * j;
      x[j] = c < D ? x[j] - mu : 0.f;
      v += x[j] * x[j];
    }
    const float rstd = 1.0f / sqrtf(row_sum(v) * invD + kLnEps);
#pragma unroll
    for (int j = 0; j < kQC; ++j) {
      const int c = q + kRL * j;
      if (c < D) Y[r * ld + c] = x[j] * rstd * wl[j] + bl[j];
    }
  }
}

// LayerNorm backward, rows < rows.  g = GY[r] (upstream), x = X[r]:
//   dx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat));  OUT[r] = (ACCUM ? OUT[r] : 0) + dx   (OUT must not alias GY)
//   GXH[r] = g * xhat  -- its column sums are dgamma, those of GY are dbeta; both are taken on the matrix cores by the
//   caller (ones-row GEMM), so this pass needs no cross-row reduction.
template <bool ACCUM>
__device__ __forceinline__ void ln_bwd_rows(int nw, const lds_f* GY, const lds_f* X, lds_f* OUT, lds_f* GXH, int rows, int LP,
                                            int ld, int D, const lds_f* w) {
  const int q = threadIdx.x & (kRL - 1), rpp = (nw << 6) / kRL;
  for (int i = threadIdx.x; i < (LP - rows) * ld; i += (nw << 6)) GXH[rows * ld + i] = 0.f;   // padding rows feed a k-sum
  const float invD = 1.0f / (float)D;
  float wl[kQC];
#pragma unroll
  for (int j = 0; j < kQC; ++j) {
    const int c = q + kRL * j;
    wl[j] = c < D ? w[c] : 0.f;
  }
  for (int r = threadIdx.x / kRL; r < rows; r += rpp) {
    float x[kQC], g[kQC];
    float s = 0.f;
#pragma unroll
    for (int j = 0; j < kQC; ++j) {
      const int c = q + kRL * j;
      x[j] = c < D ? X[r * ld + c] : 0.f;
      g[j] = c < D ? GY[r * ld + c] : 0.f;
      s += x[j];
    }
    const float mu = row_sum(s) * invD;
    float v = 0.f;
#pragma unroll
    for (int j = 0; j < kQC; ++j) {
      const int c = q + kRL * j;
      x[j] = c < D ? x[j] - mu : 0.f;
      v += x[j] * x[j];
    }
    const float rstd = 1.0f / sqrtf(row_sum(v) * invD + kLnEps);
    float s1 = 0.f, s2 = 0.f;
#pragma unroll
    for (int j = 0; j < kQC; ++j) {
      x[j] *= rstd;                       // xhat
      const float gw = g[j] * wl[j];
      s1 += gw;
      s2 += gw * x[j];
    }
    const float m1 = row_sum(s1) * invD, m2 = row_sum(s2) * invD;
#pragma unroll
    for (int j = 0; j < kQC; ++j) {
      const int c = q + kRL * j;
      if (c < D) {
        const float dx = rstd * (g[j] * wl[j] - m1 - x[j] * m2);
        if (ACCUM) OUT[r * ld + c] += dx;
        else OUT[r * ld + c] = dx;
        GXH[r * ld + c] = g[j] * x[j];
      }
    }
  }
}

struct OnesRow {        // A operand whose row 0 is all ones (rows 1..15 zero): C row 0 = column sums of B
  __device__ __forceinline__ float operator()(int r, int) const { return r == 0 ? 1.0f : 0.0f; }
};
template <> struct k_is_row<OnesRow> { static constexpr bool value = true; };   // (no LDS on the A side: B decides)

// causal softmax of score rows r < rows in place; keys j > r get exact zeros up to LP.  MASKED folds the attention
// dropout multiplier into the stored probabilities (forward); the backward keeps P unmasked and masks on load.
// A quad owns a row; each lane keeps its (up to kSMJ) elements in registers, so the row is read once and written
// once and the loops are fully unrolled (no per-element LDS round trip on the dependency chain).
constexpr int kSMJ = 128 / kRL;     // elements per lane: rows up to 128 keys
// MASKED = true : S <- mask * P (forward).   MASKED = false: S <- P and, if S_masked != nullptr, S_masked <- mask * P
// gsave (forward, training): the probabilities also go to global memory as [rows][LP], SIGN-CODED with the dropout
// mask - a dropped entry is stored negated (P >= 0, so |.| is P and the sign bit is the mask; exact zeros above the
// diagonal).  The backward pass reads them back instead of recomputing q k^T, the softmax and the mask hash.
template <bool MASKED>
__device__ __forceinline__ void softmax_rows(int nw, lds_f* S, int rows, int sld, int LP, const DropSite& ds,
                                             lds_f* S_masked = nullptr, float* gsave = nullptr, int row0 = 0) {
  const int q = threadIdx.x & (kRL - 1), rpp = (nw << 6) / kRL;
  const int nj = LP / kRL;                       // elements per lane (LP is a multiple of 16)
  if (nj > kSMJ) {                              // rows longer than 4 * kSMJ keys: streaming three-pass form
    for (int r = row0 + threadIdx.x / kRL; r < rows; r += rpp) {
      lds_f* row = S + r * sld;
      float m = -INFINITY;
      for (int j = q; j <= r; j += kRL) m = fmaxf(m, row[j]);
      m = row_max(m);
      float s = 0.f;
      for (int j = q; j <= r; j += kRL) {
        const float e = __expf(row[j] - m);
        row[j] = e;
        s += e;
      }
      s = row_sum(s);
      const float inv = 1.0f / s;                 // one division per row (a per-element IEEE division is ~10 VALU ops)
      for (int j = q; j < LP; j += kRL) {
        float p = j <= r ? row[j] * inv : 0.f;
        if (MASKED) {
          const float mul = drop_mul(ds, r, j);
          if (gsave != nullptr) gsave[(int64_t)r * LP + j] = mul != 0.f ? p : -p;
          p *= mul;
        }
        row[j] = p;
        if (!MASKED && S_masked != nullptr) S_masked[r * sld + j] = p * drop_mul(ds, r, j);
      }
    }
    return;
  }
  for (int r = row0 + threadIdx.x / kRL; r < rows; r += rpp) {
    lds_f* row = S + r * sld;
    float x[kSMJ];
    float m = -INFINITY;
#pragma unroll
    for (int i = 0; i < kSMJ; ++i) {
      const int j = q + kRL * i;
      x[i] = (i < nj && j <= r) ? row[j] : -INFINITY;
      m = fmaxf(m, x[i]);
    }
    m = row_max(m);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < kSMJ; ++i) {
      const int j = q + kRL * i;
      x[i] = (i < nj && j <= r) ? __expf(x[i] - m) : 0.f;      // v_exp_f32 path: ~1e-7 relative, far inside the 1e-4 bar
      s += x[i];
    }
    s = row_sum(s);
    const float inv = 1.0f / s;                   // one division per row (a per-element IEEE division is ~10 VALU ops)
#pragma unroll
    for (int i = 0; i < kSMJ; ++i) {
      const int j = q + kRL * i;
      if (i < nj) {
        float p = x[i] * inv;                    // exact zero above the diagonal (x = 0)
        if (MASKED) {
          const float mul = drop_mul(ds, r, j);
          if (gsave != nullptr) gsave[(int64_t)r * LP + j] = mul != 0.f ? p : -p;
          p *= mul;
        }
        row[j] = p;
        if (!MASKED && S_masked != nullptr) S_masked[r * sld + j] = p * drop_mul(ds, r, j);
      }
    }
  }
}

// dS = P * (dP - sum_j dP_j P_j), dP = mask * dPd, in place in dPd; rows >= rows (padding) are zeroed up to LP rows.
// Pm holds the sign-coded probabilities of the forward pass (see softmax_rows): P = |Pm|, kept <=> Pm > 0
// (a kept P that underflowed to +0 reads as dropped: both give dS = 0).  `scale` = 1 / (1 - p), or 1 without dropout.
__device__ __forceinline__ void softmax_bwd_rows(int nw, lds_f* dPd, const lds_f* Pm, int rows, int sld, int LP, float scale) {
  const int q = threadIdx.x & (kRL - 1), rpp = (nw << 6) / kRL;
  const int nj = LP / kRL;
  if (nj > kSMJ) {                              // long rows: streaming form
    for (int r = threadIdx.x / kRL; r < LP; r += rpp) {
      lds_f* drow = dPd + r * sld;
      const lds_f* prow = Pm + r * sld;
      if (r < rows) {
        float acc = 0.f;
        for (int j = q; j <= r; j += kRL) {
          const float pm = prow[j];
          const float d = pm > 0.f ? drow[j] * scale : 0.f;
          drow[j] = d;
          acc += d * pm;                          // (d = 0 wherever pm <= 0)
        }
        acc = row_sum(acc);
        for (int j = q; j < LP; j += kRL) drow[j] = j <= r ? fabsf(prow[j]) * (drow[j] - acc) : 0.f;
      } else {
        for (int j = q; j < LP; j += kRL) drow[j] = 0.f;
      }
    }
    return;
  }
  for (int r = threadIdx.x / kRL; r < LP; r += rpp) {
    lds_f* drow = dPd + r * sld;
    const lds_f* prow = Pm + r * sld;
    float dp[kSMJ], p[kSMJ];
    float acc = 0.f;
    const bool live = r < rows;
#pragma unroll
    for (int i = 0; i < kSMJ; ++i) {
      const int j = q + kRL * i;
      const bool on = live && i < nj && j <= r;
      const float pm = on ? prow[j] : 0.f;
      p[i] = fabsf(pm);
      dp[i] = pm > 0.f ? drow[j] * scale : 0.f;
      acc += dp[i] * p[i];
    }
    acc = row_sum(acc);
#pragma unroll
    for (int i = 0; i < kSMJ; ++i) {
      const int j = q + kRL * i;
      if (i < nj) drow[j] = p[i] * (dp[i] - acc);
    }
  }
}

// ---------------------------------------------------------------------------------------------
// user labels (reference SRFR_model.py:546-570 get_Labels, :244 SRFRN.predict): ONE routine for the standalone
// srfrd_user_labels kernel and the label the encoder kernels derive in place, so the two can never disagree.
// The ratio label is torch's float32 expression floor(fl(fl(n1 / (n1 + n2)) * 10)).  Written in floating point it
// would be at the mercy of the translation unit's relaxed flags (__graft_entry__.ENCODER_FLAGS turn the division into
// x * rcp(y) - `#pragma clang fp reciprocal(off)` does not stop that on this compiler - and at exact-decile ratios
// that lands one ulp below the integer: the floor drops by one, a different embedding row, not a 1e-4 error).  It is
// therefore computed in integers: (10 n1) / (n1 + n2) equals the float32 expression for every 0 <= n1 <= n1 + n2 <= 2048
// (checked exhaustively against numpy / torch float32; tests/test_host_logic.py repeats the check).
// ---------------------------------------------------------------------------------------------
__device__ __forceinline__ int user_label_from_counts(int kind, int n1, int n2) {
  if (kind == SRFRD_SRFU_B) return (n1 < n2) ? 1 : 2;                       // round-half-even(1.5) = 2 on ties
  if (kind == SRFRD_SRFU_F) return n1;
  if (kind == SRFRD_SRFU_R) {
    const int tot = n1 + n2;                                               // all-pad row: reference is 0/0; guarded to 0
    if (tot == 0) return 0;
    return (10 * n1) / tot;
  }
  return (n1 > n2) ? 2 : 1;                                                // SRFRN.predict: int() truncation, tie -> 1
}
// wave-uniform label of one sequence from its fake(1) / real(2) ids; n_labels > 0 clamps it into the label table (the
// encoder's gather must stay in bounds; the reference would raise an index error instead)
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
  int lab = user_label_from_counts(kind, n1, n2);
  if (n_labels > 0) lab = min(max(lab, 0), n_labels - 1);
  return lab;
}

// optimizer state advance (one thread): t += 1, Adam bias corrections in double precision as torch computes them on
// the host, dropout seed of the step.  state: uint32[8] {t, base_seed, step_seed, -, f32 step_size, f32 bc2_sqrt, -, -}
__device__ __forceinline__ void step_advance(uint32_t* state, double lr, double b1, double b2) {
  const uint32_t t = state[0] + 1u;
  state[0] = t;
  state[2] = step_seed(state[1], t);
  const double bc1 = 1.0 - pow(b1, (double)t);
  const double bc2 = 1.0 - pow(b2, (double)t);
  ((float*)state)[4] = (float)(lr / bc1);
  ((float*)state)[5] = (float)sqrt(bc2);
}

__device__ __forceinline__ float softplus_f(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigmoid_f(float z) { return 1.0f / (1.0f + expf(-z)); }

}  // namespace SRFRD_NS
