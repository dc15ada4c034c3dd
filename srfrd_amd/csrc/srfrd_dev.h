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

namespace srfrd {

typedef float f32x4 __attribute__((ext_vector_type(4)));
// LDS-resident matrices are addressed through explicit address_space(3) pointers: a generic `float*` that the
// compiler cannot trace back to __shared__ becomes flat_load/flat_store with 64-bit address arithmetic.
typedef __attribute__((address_space(3))) float lds_f;
typedef __attribute__((address_space(3))) int lds_i;

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

// LDS floats: forward = XS + 4 matrices; backward = 8 matrices + 2 score matrices; + slack + per-row scalars
__host__ __device__ __forceinline__ int64_t fwd_lds_floats(const Geom& g) {
  return (int64_t)imax(g.LP * g.DS, g.LP * g.SLD) + 4ll * g.LP * g.DS + kSlack + 4ll * g.LP + 64;
}
__host__ __device__ __forceinline__ int64_t bwd_lds_floats(const Geom& g) {
  return 8ll * g.LP * g.DS + 2ll * imax(g.LP * g.SLD, g.LP * g.DS) + kSlack + 10ll * g.LP + 64 + 8ll * 2 * 64;
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
struct MatDrop {        // P with the attention-dropout multiplier applied on load (row = query, col = key)
  const lds_f* p;
  int ld;
  DropSite ds;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[r * ld + c] * drop_mul(ds, r, c); }
};
struct MatDropT {       // transpose of the above: element (key, query)
  const lds_f* p;
  int ld;
  DropSite ds;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[c * ld + r] * drop_mul(ds, c, r); }
};
struct MatOnes {        // [M | 1]: column `one_col` reads 1.0 - folds a column-sum (bias gradient) into a dW GEMM
  const lds_f* p;
  int ld, one_col;
  __device__ __forceinline__ float operator()(int r, int c) const { return c == one_col ? 1.0f : p[r * ld + c]; }
};
struct WgtNT {          // B(k, n) = W[n][k]  (y = x W^T, torch Linear / Conv1d(k=1) weight (N, K)); 0 outside
  const float* w;
  int N, K;
  __device__ __forceinline__ float operator()(int k, int n) const {
    const float v = w[min(n, N - 1) * K + min(k, K - 1)];     // clamped address + select: no branch in the MFMA loop
    return (n < N && k < K) ? v : 0.0f;
  }
};
struct WgtNN {          // B(k, n) = W[k][n]  (dx = dy W, weight (K, N)); 0 outside
  const float* w;
  int K, N;
  __device__ __forceinline__ float operator()(int k, int n) const {
    const float v = w[min(k, K - 1) * N + min(n, N - 1)];
    return (k < K && n < N) ? v : 0.0f;
  }
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
template <int G, class AL, class BL>
__device__ __forceinline__ void mma_group(f32x4 (&acc)[G], const AL& a, const BL& b, int mrow, int mstride, int ncol,
                                          int k0, int k1, int lq) {
  int k = k0;
  for (; k + 16 <= k1; k += 16) {
    float bv[4], av[4][G];
#pragma unroll
    for (int s = 0; s < 4; ++s) bv[s] = b(k + 4 * s + lq, ncol);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < G; ++j) av[s][j] = a(mrow + j * mstride, k + 4 * s + lq);
    // keep all 4 + 4G loads in flight ahead of the MFMAs (hipcc otherwise re-serialises load -> wait -> mfma)
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int j = 0; j < G; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[s][j], bv[s], acc[j], 0, 0, 0);
    __builtin_amdgcn_sched_barrier(0);
  }
  for (; k < k1; k += 4) {
    const float bv = b(k + lq, ncol);
    float av[G];
#pragma unroll
    for (int j = 0; j < G; ++j) av[j] = a(mrow + j * mstride, k + lq);
#pragma unroll
    for (int j = 0; j < G; ++j) acc[j] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[j], bv, acc[j], 0, 0, 0);
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

template <int TRI, class AL, class BL, class EPI>
__device__ __forceinline__ void gemm_tiles(int m_tiles, int n_tiles, int k_end, AL a, BL b, EPI epi) {
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const int nw = blockDim.x >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int mgroups = nw > n_tiles ? nw / n_tiles : 1;
  const int units = n_tiles * mgroups;
  for (int u = wave; u < units; u += nw) {
    const int nt = u % n_tiles, g = u / n_tiles;
    const int n0 = nt << 4;
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
// row-wise ops: one wave per row, lane = channel (D <= 64)
// ---------------------------------------------------------------------------------------------
// Y[r] = LayerNorm(X[r]) for r < rows   (biased variance, eps inside the sqrt: torch.nn.LayerNorm)
__device__ __forceinline__ void ln_rows(const lds_f* X, lds_f* Y, int rows, int ld, int D, const float* w,
                                        const float* bia) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const float wl = lane < D ? w[lane] : 0.f, bl = lane < D ? bia[lane] : 0.f;
  const float invD = 1.0f / (float)D;
  for (int r = wave; r < rows; r += nw) {
    const float x = lane < D ? X[r * ld + lane] : 0.f;
    const float mu = wave_sum(x) * invD;
    const float xc = lane < D ? x - mu : 0.f;
    const float var = wave_sum(xc * xc) * invD;
    const float rstd = 1.0f / sqrtf(var + kLnEps);
    if (lane < D) Y[r * ld + lane] = xc * rstd * wl + bl;
  }
}

// LayerNorm backward for rows < rows:  G[r] <- dX (in place, or G[r] += dX when ACCUM) given upstream G? no:
//   g = GY[r] (upstream), x = X[r];  dx = rstd * (g*w - mean(g*w) - xhat * mean(g*w*xhat))
//   OUT[r] = (ACCUM ? OUT[r] : 0) + dx ;  per-lane partial sums of dgamma = g*xhat, dbeta = g are returned in
//   (dg, db) accumulated over the rows this wave handled.
template <bool ACCUM>
__device__ __forceinline__ void ln_bwd_rows(const lds_f* GY, const lds_f* X, lds_f* OUT, int rows, int ld, int D,
                                            const float* w, float& dg, float& db) {
  const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), nw = blockDim.x >> 6;
  const float wl = lane < D ? w[lane] : 0.f;
  const float invD = 1.0f / (float)D;
  for (int r = wave; r < rows; r += nw) {
    const float x = lane < D ? X[r * ld + lane] : 0.f;
    const float g = lane < D ? GY[r * ld + lane] : 0.f;
    const float mu = wave_sum(x) * invD;
    const float xc = lane < D ? x - mu : 0.f;
    const float var = wave_sum(xc * xc) * invD;
    const float rstd = 1.0f / sqrtf(var + kLnEps);
    const float xh = xc * rstd;
    const float gw = g * wl;
    const float m1 = wave_sum(gw) * invD;
    const float m2 = wave_sum(gw * xh) * invD;
    const float dx = rstd * (gw - m1 - xh * m2);
    dg += g * xh;
    db += g;
    if (lane < D) {
      if (ACCUM) OUT[r * ld + lane] += dx;
      else OUT[r * ld + lane] = dx;
    }
  }
}

__device__ __forceinline__ float softplus_f(float z) { return fmaxf(z, 0.f) + log1pf(expf(-fabsf(z))); }
__device__ __forceinline__ float sigmoid_f(float z) { return 1.0f / (1.0f + expf(-z)); }

}  // namespace srfrd
