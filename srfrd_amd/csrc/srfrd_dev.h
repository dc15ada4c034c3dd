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
  const float* p;
  int ld;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[r * ld + c]; }
};
struct MatT {           // element (r, c) of the transpose of a row-major LDS matrix
  const float* p;
  int ld;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[c * ld + r]; }
};
struct MatDrop {        // P with the attention-dropout multiplier applied on load (row = query, col = key)
  const float* p;
  int ld;
  DropSite ds;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[r * ld + c] * drop_mul(ds, r, c); }
};
struct MatDropT {       // transpose of the above: element (key, query)
  const float* p;
  int ld;
  DropSite ds;
  __device__ __forceinline__ float operator()(int r, int c) const { return p[c * ld + r] * drop_mul(ds, c, r); }
};
struct WgtNT {          // B(k, n) = W[n][k]  (y = x W^T, torch Linear / Conv1d(k=1) weight (N, K)); 0 outside
  const float* w;
  int N, K;
  __device__ __forceinline__ float operator()(int k, int n) const { return (n < N && k < K) ? w[n * K + k] : 0.0f; }
};
struct WgtNN {          // B(k, n) = W[k][n]  (dx = dy W, weight (K, N)); 0 outside
  const float* w;
  int K, N;
  __device__ __forceinline__ float operator()(int k, int n) const { return (k < K && n < N) ? w[k * N + n] : 0.0f; }
};

// ---------------------------------------------------------------------------------------------
// tiled GEMM on v_mfma_f32_16x16x4_f32.   C(16 mt.., 16 nt..) = sum_k A(row, k) * B(k, col)
// A operand: lane l holds A[m0 + (l & 15)][k + (l >> 4)];  B operand: B[k + (l >> 4)][n0 + (l & 15)]
// C/D: register r of lane l is element (m0 + 4 * (l >> 4) + r, n0 + (l & 15))
// TRI: 0 full; 1 skip tiles with nt > mt (lower-triangular C); 2 k < 16 (mt + 1) (A lower-triangular);
//      3 k >= 16 mt (A = transpose of a lower-triangular matrix)
// ---------------------------------------------------------------------------------------------
template <int TRI, class AL, class BL, class EPI>
__device__ __forceinline__ void gemm_tiles(int m_tiles, int n_tiles, int k_end, AL a, BL b, EPI epi) {
  const int lane = threadIdx.x & 63;
  const int wave = threadIdx.x >> 6;
  const int nw = blockDim.x >> 6;
  const int li = lane & 15, lq = lane >> 4;
  const int total = m_tiles * n_tiles;
  for (int t = wave; t < total; t += nw) {
    const int mt = t / n_tiles, nt = t - mt * n_tiles;
    if (TRI == 1 && nt > mt) continue;
    const int m0 = mt << 4, n0 = nt << 4;
    int k0 = 0, k1 = k_end;
    if (TRI == 2) k1 = min(k_end, (mt + 1) << 4);
    if (TRI == 3) k0 = mt << 4;
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    for (int k = k0; k < k1; k += 4) {
      const float av = a(m0 + li, k + lq);
      const float bv = b(k + lq, n0 + li);
      acc = __builtin_amdgcn_mfma_f32_16x16x4f32(av, bv, acc, 0, 0, 0);
    }
    const int r0 = m0 + (lq << 2), c = n0 + li;
    epi(r0 + 0, c, acc[0]);
    epi(r0 + 1, c, acc[1]);
    epi(r0 + 2, c, acc[2]);
    epi(r0 + 3, c, acc[3]);
  }
}

// ---------------------------------------------------------------------------------------------
// row-wise ops: one wave per row, lane = channel (D <= 64)
// ---------------------------------------------------------------------------------------------
// Y[r] = LayerNorm(X[r]) for r < rows   (biased variance, eps inside the sqrt: torch.nn.LayerNorm)
__device__ __forceinline__ void ln_rows(const float* X, float* Y, int rows, int ld, int D, const float* w,
                                        const float* bia) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
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
__device__ __forceinline__ void ln_bwd_rows(const float* GY, const float* X, float* OUT, int rows, int ld, int D,
                                            const float* w, float& dg, float& db) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
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
