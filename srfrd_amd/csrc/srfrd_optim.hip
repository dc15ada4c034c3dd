// Gradient reduction, loss finalisation and the dense Adam step (torch.optim.Adam semantics,
// reference trainer.py:36-41, :390).  All streaming, HBM-bound, float4-wide.
#include "srfrd_dev.h"

namespace srfrd {

// grad_dense[i] = sum_w slabs[w][i]: a block owns 64 columns; its 4 waves each sum an interleaved quarter of the
// slabs (4 loads in flight per lane), then the 4 partials are added in wave order - a fixed summation tree, so the
// result is bitwise reproducible.  Block 0 also reduces the BCE partials.  (A float4 form - 16 columns x 16 slab groups per
// block, a quarter of the load instructions - measured the same 10 us by events: the kernel waits on L2 / MALL, not on issue.)
__global__ void __launch_bounds__(256) reduce_dense_kernel(const float* __restrict__ slabs, int n_slabs, int64_t n_dense,
                                                          float* __restrict__ grad_dense, const float* __restrict__ loss_part,
                                                          int B, float* __restrict__ stats, float* __restrict__ loss_out) {
  __shared__ float part[4][64];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int64_t i = (int64_t)blockIdx.x * 64 + lane;
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  if (i < n_dense) {
    int w = wave;
    for (; w + 28 < n_slabs; w += 32) {                   // eight loads in flight per lane
      const float a0 = slabs[(int64_t)w * n_dense + i], a1 = slabs[(int64_t)(w + 4) * n_dense + i];
      const float a2 = slabs[(int64_t)(w + 8) * n_dense + i], a3 = slabs[(int64_t)(w + 12) * n_dense + i];
      const float a4 = slabs[(int64_t)(w + 16) * n_dense + i], a5 = slabs[(int64_t)(w + 20) * n_dense + i];
      const float a6 = slabs[(int64_t)(w + 24) * n_dense + i], a7 = slabs[(int64_t)(w + 28) * n_dense + i];
      s0 += a0; s1 += a1; s2 += a2; s3 += a3;
      s0 += a4; s1 += a5; s2 += a6; s3 += a7;
    }
    for (; w + 12 < n_slabs; w += 16) {
      s0 += slabs[(int64_t)w * n_dense + i];
      s1 += slabs[(int64_t)(w + 4) * n_dense + i];
      s2 += slabs[(int64_t)(w + 8) * n_dense + i];
      s3 += slabs[(int64_t)(w + 12) * n_dense + i];
    }
    for (; w < n_slabs; w += 4) s0 += slabs[(int64_t)w * n_dense + i];
  }
  part[wave][lane] = (s0 + s1) + (s2 + s3);
  __syncthreads();
  if (wave == 0 && i < n_dense) grad_dense[i] = (part[0][lane] + part[1][lane]) + (part[2][lane] + part[3][lane]);
  if (blockIdx.x == 0 && loss_part != nullptr && stats != nullptr) {
    __shared__ float red[4][3];
    float a0 = 0.f, a1 = 0.f, a2 = 0.f;
    for (int b = threadIdx.x; b < B; b += blockDim.x) {
      a0 += loss_part[(int64_t)b * 3 + 0];
      a1 += loss_part[(int64_t)b * 3 + 1];
      a2 += loss_part[(int64_t)b * 3 + 2];
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
    if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; }
    __syncthreads();
    if (threadIdx.x < 3) {
      float s = 0.f;
      for (int w = 0; w < 4; ++w) s += red[w][threadIdx.x];
      stats[threadIdx.x] = s;
      red[0][threadIdx.x] = s;
    }
    if (threadIdx.x == 3) stats[3] = 0.f;
    if (loss_out != nullptr) {               // single-rank fast path: the loss of reference trainer.py:36-38 right here
      __syncthreads();
      if (threadIdx.x == 0) loss_out[0] = red[0][0] / red[0][2] + red[0][1] / red[0][2];
    }
  }
}

// stats[0..3] = {sum softplus(-pos), sum softplus(neg), count, 0} from the forward's per-sequence partials: the
// data-parallel step reduces them on their own, right after the forward, so their all-reduce overlaps the backward
__global__ void __launch_bounds__(256) loss_stats_kernel(const float* __restrict__ loss_part, int B, float* __restrict__ stats,
                                                        float* __restrict__ loss_out) {
  __shared__ float red[4][3];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  float a0 = 0.f, a1 = 0.f, a2 = 0.f;
  for (int b = threadIdx.x; b < B; b += blockDim.x) {
    a0 += loss_part[(int64_t)b * 3 + 0];
    a1 += loss_part[(int64_t)b * 3 + 1];
    a2 += loss_part[(int64_t)b * 3 + 2];
  }
  a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2);
  if (lane == 0) { red[wave][0] = a0; red[wave][1] = a1; red[wave][2] = a2; }
  __syncthreads();
  if (threadIdx.x < 3) {
    float s = 0.f;
    for (int w = 0; w < 4; ++w) s += red[w][threadIdx.x];
    stats[threadIdx.x] = s;
    red[0][threadIdx.x] = s;
  }
  if (threadIdx.x == 3) stats[3] = 0.f;
  __syncthreads();
  if (loss_out != nullptr && threadIdx.x == 0) loss_out[0] = red[0][0] / red[0][2] + red[0][1] / red[0][2];
}

// deterministic item-table scatter: one wave per sorted position; the wave at the head of a run of equal keys owns the item
__global__ void __launch_bounds__(256) table_reduce_kernel(const int64_t* __restrict__ keys, const int64_t* __restrict__ order,
                                                          const float* __restrict__ contrib, int64_t n, int di,
                                                          float* __restrict__ grad_table) {
  const int lane = threadIdx.x & 63;
  const int64_t i = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (i >= n) return;
  const int64_t key = keys[i];
  if (key == 0 || (i > 0 && keys[i - 1] == key)) return;          // padding_idx row, or not the head of its run
  float acc = 0.f;
  for (int64_t j = i; j < n && keys[j] == key; ++j)
    if (lane < di) acc += contrib[order[j] * di + lane];
  if (lane < di) grad_table[key * di + lane] = acc;
}

__global__ void step_begin_kernel(uint32_t* state, double lr, double b1, double b2) {
  if (threadIdx.x == 0 && blockIdx.x == 0) step_advance(state, lr, b1, b2);
}

__device__ __forceinline__ void adam_one(float& p, float g, float& m, float& v, float b1, float b2, float eps,
                                         float step_size, float bc2s) {
  m = m * b1 + (1.0f - b1) * g;           // exp_avg.lerp_(grad, 1 - beta1) == exp_avg*b1 + (1-b1)*g up to rounding
  v = v * b2 + ((1.0f - b2) * g) * g;     // exp_avg_sq.mul_(b2).addcmul_(g, g, 1 - b2)
  const float denom = sqrtf(v) / bc2s + eps;
  p = p - step_size * (m / denom);
}

// bf16 shadow of four stepped item-table elements starting at i (i is a multiple of 4: one 8-byte store when all four are
// inside the table)
__device__ __forceinline__ void shadow_store4(uint16_t* __restrict__ shadow, int64_t i, int64_t n_shadow, const float4& p4) {
  const uint16_t h0 = f32_to_bf16(p4.x), h1 = f32_to_bf16(p4.y), h2 = f32_to_bf16(p4.z), h3 = f32_to_bf16(p4.w);
  if (i + 3 < n_shadow) {
    *reinterpret_cast<uint2*>(shadow + i) = make_uint2((uint32_t)h0 | ((uint32_t)h1 << 16), (uint32_t)h2 | ((uint32_t)h3 << 16));
  } else {
    shadow[i] = h0;
    if (i + 1 < n_shadow) shadow[i + 1] = h1;
    if (i + 2 < n_shadow) shadow[i + 2] = h2;
  }
}

__global__ void __launch_bounds__(256) table_to_bf16_kernel(const float* __restrict__ src, int64_t n, uint16_t* __restrict__ out) {
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t nvec = n >> 2;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nvec; q += stride)
    shadow_store4(out, q << 2, n, *reinterpret_cast<const float4*>(src + (q << 2)));
  for (int64_t i = (nvec << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) out[i] = f32_to_bf16(src[i]);
}

__global__ void __launch_bounds__(256) adam_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m,
                                                  float* __restrict__ v, int64_t i0, int64_t i1, int64_t n_zero, float b1,
                                                  float b2, float eps, const uint32_t* __restrict__ state,
                                                  const float* __restrict__ stats, uint16_t* __restrict__ shadow, int64_t n_shadow) {
  const float step_size = ((const float*)state)[4];
  const float bc2s = ((const float*)state)[5];
  const float gscale = stats ? 1.0f / stats[2] : 1.0f;
  // i0 is a multiple of 4 (host guarantees), so float4 lanes are aligned
  const int64_t nvec = (i1 - i0) >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nvec; q += stride) {
    const int64_t i = i0 + (q << 2);
    float4 p4 = *reinterpret_cast<float4*>(param + i);
    float4 g4 = *reinterpret_cast<const float4*>(grad + i);
    float4 m4 = *reinterpret_cast<float4*>(m + i);
    float4 v4 = *reinterpret_cast<float4*>(v + i);
    adam_one(p4.x, g4.x * gscale, m4.x, v4.x, b1, b2, eps, step_size, bc2s);
    adam_one(p4.y, g4.y * gscale, m4.y, v4.y, b1, b2, eps, step_size, bc2s);
    adam_one(p4.z, g4.z * gscale, m4.z, v4.z, b1, b2, eps, step_size, bc2s);
    adam_one(p4.w, g4.w * gscale, m4.w, v4.w, b1, b2, eps, step_size, bc2s);
    *reinterpret_cast<float4*>(param + i) = p4;
    *reinterpret_cast<float4*>(m + i) = m4;
    *reinterpret_cast<float4*>(v + i) = v4;
    if (i + 3 < n_zero) *reinterpret_cast<float4*>(grad + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    else if (i < n_zero) for (int k = 0; k < 4; ++k) if (i + k < n_zero) grad[i + k] = 0.f;
    if (shadow != nullptr && i < n_shadow) shadow_store4(shadow, i, n_shadow, p4);
  }
  // scalar tail
  const int64_t tail0 = i0 + (nvec << 2);
  for (int64_t i = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += stride) {
    float p = param[i], mm = m[i], vv = v[i];
    adam_one(p, grad[i] * gscale, mm, vv, b1, b2, eps, step_size, bc2s);
    param[i] = p; m[i] = mm; v[i] = vv;
    if (i < n_zero) grad[i] = 0.f;
    if (shadow != nullptr && i < n_shadow) shadow[i] = f32_to_bf16(p);
  }
}

// Packed-weight positions of dense parameter j (if it belongs to a weight matrix): the fused optimizer writes the
// stepped value straight into both MFMA-fragment forms (the layout srfrd_pack_weights produces), so the train step
// needs no separate re-pack launch.  Zero padding of the packed buffer is never touched.
__device__ __forceinline__ void pack_scatter(const Dims& d, int64_t j, float val, float* __restrict__ packed) {
  int mat, N, K, e;
  const int D = d.D, DD = D * D;
  if (d.off_lc_w >= 0 && j >= d.off_lc_w && j < d.off_lc_w + (int64_t)d.d_item * D) {
    mat = d.n_blocks * 6; N = d.d_item; K = D; e = (int)(j - d.off_lc_w);
  } else {
    if (j < d.blk0 || j >= d.blk0 + (int64_t)d.n_blocks * d.blk_stride) return;
    const int b = (int)((j - d.blk0) / d.blk_stride), off = (int)(j - d.blk0) - b * d.blk_stride;
    const BlkOff o = blk_off(0, D);
    int m;
    if (off >= o.in_w && off < o.in_w + 3 * DD) { m = (off - o.in_w) / DD; e = (off - o.in_w) - m * DD; }
    else if (off >= o.out_w && off < o.out_w + DD) { m = 3; e = off - o.out_w; }
    else if (off >= o.c1_w && off < o.c1_w + DD) { m = 4; e = off - o.c1_w; }
    else if (off >= o.c2_w && off < o.c2_w + DD) { m = 5; e = off - o.c2_w; }
    else return;
    mat = b * 6 + m; N = D; K = D;
  }
  (void)N;
  const int n = e / K, k = e - n * K;             // W[n][k]
  auto slot = [](int kk, int nn) {                // fragment index of B(kk, nn): see pack_weights_kernel
    const int rem = kk & 15;
    return ((nn >> 4) << 10) | ((kk >> 4) << 8) | ((((rem & 3) << 4) | (nn & 15)) << 2) | (rem >> 2);
  };
  float* base = packed + (int64_t)mat * 2 * kPackFloats;
  base[slot(k, n)] = val;                         // form 0: B(k, n) = W[n][k]   (x W^T)
  base[kPackFloats + slot(n, k)] = val;           // form 1: B(n, k) = W[n][k]   (dy W)
}

// Adam over the whole flat vector + re-pack of the stepped encoder weights + (last block to finish) the optimizer-state
// advance for the next step: one launch where the train step used to end with two.
__global__ void __launch_bounds__(512) adam_pack_kernel(float* __restrict__ param, float* __restrict__ grad, float* __restrict__ m,
                                                       float* __restrict__ v, int64_t n, int64_t n_tab, int64_t n_zero, float b1,
                                                       float b2, float eps, uint32_t* __restrict__ state,
                                                       const float* __restrict__ stats, Dims dims, float* __restrict__ packed,
                                                       double lr, double b1d, double b2d, uint16_t* __restrict__ shadow,
                                                       int64_t n_shadow) {
  const float step_size = ((const float*)state)[4];
  const float bc2s = ((const float*)state)[5];
  const float gscale = stats ? 1.0f / stats[2] : 1.0f;
  const int64_t nvec = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t w_lo = n_tab + dims.blk0;          // first dense index that can be an encoder weight
  for (int64_t q = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; q < nvec; q += stride) {
    const int64_t i = q << 2;
    float4 p4 = *reinterpret_cast<float4*>(param + i);
    float4 g4 = *reinterpret_cast<const float4*>(grad + i);
    float4 m4 = *reinterpret_cast<float4*>(m + i);
    float4 v4 = *reinterpret_cast<float4*>(v + i);
    adam_one(p4.x, g4.x * gscale, m4.x, v4.x, b1, b2, eps, step_size, bc2s);
    adam_one(p4.y, g4.y * gscale, m4.y, v4.y, b1, b2, eps, step_size, bc2s);
    adam_one(p4.z, g4.z * gscale, m4.z, v4.z, b1, b2, eps, step_size, bc2s);
    adam_one(p4.w, g4.w * gscale, m4.w, v4.w, b1, b2, eps, step_size, bc2s);
    *reinterpret_cast<float4*>(param + i) = p4;
    *reinterpret_cast<float4*>(m + i) = m4;
    *reinterpret_cast<float4*>(v + i) = v4;
    if (i + 3 < n_zero) *reinterpret_cast<float4*>(grad + i) = make_float4(0.f, 0.f, 0.f, 0.f);
    else if (i < n_zero) for (int k = 0; k < 4; ++k) if (i + k < n_zero) grad[i + k] = 0.f;
    if (shadow != nullptr && i < n_shadow) shadow_store4(shadow, i, n_shadow, p4);
    if (i + 3 >= w_lo) {
      pack_scatter(dims, i - n_tab, p4.x, packed);
      pack_scatter(dims, i + 1 - n_tab, p4.y, packed);
      pack_scatter(dims, i + 2 - n_tab, p4.z, packed);
      pack_scatter(dims, i + 3 - n_tab, p4.w, packed);
    }
  }
  for (int64_t i = (nvec << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {     // scalar tail
    float p = param[i], mm = m[i], vv = v[i];
    adam_one(p, grad[i] * gscale, mm, vv, b1, b2, eps, step_size, bc2s);
    param[i] = p; m[i] = mm; v[i] = vv;
    if (i < n_zero) grad[i] = 0.f;
    if (shadow != nullptr && i < n_shadow) shadow[i] = f32_to_bf16(p);
    if (i >= w_lo) pack_scatter(dims, i - n_tab, p, packed);
  }
  // The block that finishes last advances (t, bias corrections, dropout seed): every block has read them by then.
  // Two-level ticket (16 shard counters state[8..23], root state[6]): a single counter taking one returning atomic per
  // block serialises at ~11 ns each - 45 us for 4096 blocks, measured.  No agent-scope fence: nothing is published
  // to other blocks (an agent release would write back the XCD's dirty L2 lines once per block), and every thread's
  // reads of `state` returned before its stores above could issue.
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned shard = blockIdx.x & 15u, n_shards = gridDim.x < 16u ? gridDim.x : 16u;
    const unsigned in_shard = (gridDim.x - shard + 15u) >> 4;
    if (atomicAdd(&state[8 + shard], 1u) == in_shard - 1u) {
      state[8 + shard] = 0;
      if (atomicAdd(&state[6], 1u) == n_shards - 1u) {
        state[6] = 0;
        step_advance(state, lr, b1d, b2d);
      }
    }
  }
}

// ---- l2_emb * sum_p ||p||_2 of reference trainer.py:39 (one Frobenius norm per parameter tensor), for the fused step.
// Pass A: workgroup g < n_tw sums the squares of its contiguous share of segment 0 (the item table), workgroup n_tw + k - 1
// those of dense segment k: fixed shares, fixed in-block tree.  Pass B (one workgroup): norms in segment order,
// l2buf = {l2 / ||table||, l2 * sum of norms}, and dense_scale[j] = l2 / ||p|| of the tensor dense element j belongs to
// (0 in the alignment gaps and for an all-zero tensor, where torch's norm backward is 0 as well).
__global__ void __launch_bounds__(256) l2_partial_kernel(const float* __restrict__ param, const int64_t* __restrict__ seg_off,
                                                        const int64_t* __restrict__ seg_len, int n_tw, float* __restrict__ partial) {
  __shared__ float red[4];
  const int g = blockIdx.x;
  int64_t lo, hi;
  if (g < n_tw) {
    const int64_t n = seg_len[0], per = ((n + n_tw - 1) / n_tw + 3) & ~3ll;
    lo = seg_off[0] + g * per;
    hi = seg_off[0] + (g + 1) * per < seg_off[0] + n ? seg_off[0] + (g + 1) * per : seg_off[0] + n;
  } else {
    const int k = g - n_tw + 1;
    lo = seg_off[k];
    hi = lo + seg_len[k];
  }
  float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
  int64_t i = lo + threadIdx.x;
  for (; i + 768 < hi; i += 1024) {
    const float a = param[i], b = param[i + 256], c = param[i + 512], d = param[i + 768];
    s0 += a * a; s1 += b * b; s2 += c * c; s3 += d * d;
  }
  for (; i < hi; i += 256) { const float a = param[i]; s0 += a * a; }
  float s = wave_sum((s0 + s1) + (s2 + s3));
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  if (threadIdx.x == 0) partial[g] = (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256) l2_finish_kernel(const float* __restrict__ partial, const int64_t* __restrict__ seg_off,
                                                       const int64_t* __restrict__ seg_len, int n_seg, int n_tw,
                                                       int64_t n_table_pad, float l2, float* __restrict__ l2buf,
                                                       float* __restrict__ dense_scale) {
  __shared__ float s_scale[128];
  __shared__ float s_tab[4];
  float t = 0.f;
  for (int g = threadIdx.x; g < n_tw; g += 256) t += partial[g];       // (a thread owns fixed entries: a fixed tree)
  t = wave_sum(t);
  if ((threadIdx.x & 63) == 0) s_tab[threadIdx.x >> 6] = t;
  __syncthreads();
  if (threadIdx.x == 0) {
    float total = 0.f;
    for (int k = 0; k < n_seg; ++k) {
      const float sq = k == 0 ? (s_tab[0] + s_tab[1]) + (s_tab[2] + s_tab[3]) : partial[n_tw + k - 1];
      const float nrm = sqrtf(sq);
      total += nrm;
      s_scale[k] = nrm > 0.f ? l2 / nrm : 0.f;
    }
    l2buf[0] = s_scale[0];
    l2buf[1] = l2 * total;
  }
  __syncthreads();
  for (int k = 1; k < n_seg; ++k) {
    const int64_t base = seg_off[k] - n_table_pad;
    const float sc = s_scale[k];
    for (int64_t j = threadIdx.x; j < seg_len[k]; j += 256) dense_scale[base + j] = sc;
  }
}

// grad[i] += count * (l2 / ||p||) * p[i] over [i0, i1): ahead of an Adam step that divides by the (global) count of targets
__global__ void __launch_bounds__(256) l2_apply_kernel(float* __restrict__ grad, const float* __restrict__ param, int64_t i0,
                                                      int64_t i1, int64_t n_table_pad, const float* __restrict__ l2buf,
                                                      const float* __restrict__ dense_scale, const float* __restrict__ stats) {
  const float cnt = stats ? stats[2] : 1.0f, ts = l2buf[0] * cnt;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = i0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < i1; i += stride) {
    const float sc = i < n_table_pad ? ts : dense_scale[i - n_table_pad] * cnt;
    grad[i] += sc * param[i];
  }
}

__global__ void loss_finalize_kernel(const float* stats, float* loss_out) {
  if (threadIdx.x == 0 && blockIdx.x == 0) loss_out[0] = stats[0] / stats[2] + stats[1] / stats[2];
}

}  // namespace srfrd

using namespace srfrd;

extern "C" int srfrd_reduce_dense(const float* grad_slabs, int n_slabs, int64_t n_dense, float* grad_dense,
                                  const float* loss_part, int B, float* stats, float* loss_out, void* stream) {
  if (!grad_slabs || !grad_dense || n_slabs <= 0 || n_dense <= 0) return SRFRD_E_ARG;
  if ((loss_part != nullptr) != (stats != nullptr)) return SRFRD_E_ARG;
  const int grid = (int)((n_dense + 63) / 64);
  hipLaunchKernelGGL(reduce_dense_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, grad_slabs, n_slabs, n_dense,
                     grad_dense, loss_part, B, stats, loss_part ? loss_out : nullptr);
  return (int)hipGetLastError();
}

extern "C" int srfrd_table_reduce(const int64_t* sorted_keys, const int64_t* order, const float* table_contrib, int64_t n, int d_item,
                                  float* grad_table, void* stream) {
  if (!sorted_keys || !order || !table_contrib || !grad_table || n <= 0 || d_item <= 0 || d_item > 64) return SRFRD_E_ARG;
  hipLaunchKernelGGL(table_reduce_kernel, dim3((unsigned)((n + 3) / 4)), dim3(256), 0, (hipStream_t)stream, sorted_keys, order,
                     table_contrib, n, d_item, grad_table);
  return (int)hipGetLastError();
}

extern "C" int srfrd_loss_stats(const float* loss_part, int B, float* stats, float* loss_out, void* stream) {
  if (!loss_part || !stats || B <= 0) return SRFRD_E_ARG;
  hipLaunchKernelGGL(loss_stats_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, loss_part, B, stats, loss_out);
  return (int)hipGetLastError();
}

extern "C" int srfrd_step_begin(uint32_t* state, double lr, double beta1, double beta2, void* stream) {
  if (!state) return SRFRD_E_ARG;
  hipLaunchKernelGGL(step_begin_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, state, lr, beta1, beta2);
  return (int)hipGetLastError();
}

extern "C" int srfrd_table_to_bf16(const float* src, int64_t n, uint16_t* out, void* stream) {
  if (!src || !out || n <= 0) return SRFRD_E_ARG;
  int64_t grid = ((n >> 2) + 1 + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(table_to_bf16_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, src, n, out);
  return (int)hipGetLastError();
}

extern "C" int srfrd_adam_step(float* param, float* grad, float* m, float* v, int64_t n, int64_t i0, int64_t i1,
                               int64_t n_zero, double beta1, double beta2, double eps, const uint32_t* state,
                               const float* stats, uint16_t* table_bf16, int64_t n_table, void* stream) {
  if (!param || !grad || !m || !v || !state || n <= 0 || i0 < 0 || i1 > n || i0 > i1 || (i0 & 3)) return SRFRD_E_ARG;
  if (i0 == i1) return 0;
  const int64_t nvec = ((i1 - i0) >> 2) + 1;
  int64_t grid = (nvec + 255) / 256;
  if (grid > 4096) grid = 4096;
  hipLaunchKernelGGL(adam_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, param, grad, m, v, i0, i1, n_zero,
                     (float)beta1, (float)beta2, (float)eps, state, stats, table_bf16, table_bf16 ? n_table : 0);
  return (int)hipGetLastError();
}

extern "C" int srfrd_adam_pack_step(const srfrd_layout* lay, float* param, float* grad, float* m, float* v, int64_t n,
                                    int64_t n_table_pad, int64_t n_zero, double lr, double beta1, double beta2, double eps,
                                    uint32_t* state, const float* stats, float* packed, uint16_t* table_bf16, void* stream) {
  if (!lay || !param || !grad || !m || !v || !state || !packed || n <= 0 || n_table_pad < 0 || (n_table_pad & 3) ||
      n_table_pad + lay->n_dense > n)
    return SRFRD_E_ARG;
  if (lay->D > SRFRD_MAX_D) return SRFRD_E_UNSUPPORTED;
  Dims d = {};
  d.kind = lay->kind; d.d_item = lay->d_item; d.d_fake = lay->d_fake; d.D = lay->D; d.d_out = lay->d_out;
  d.n_labels = lay->n_labels; d.n_blocks = lay->n_blocks;
  d.off_pos = (int)lay->off_pos; d.off_side = (int)lay->off_side;
  d.blk0 = lay->n_blocks > 0 ? (int)lay->blk[0].ln1_w : 0;
  d.blk_stride = blk_stride_of(lay->D);
  d.off_lc_w = (int)lay->off_lc_w; d.off_lc_b = (int)lay->off_lc_b; d.off_ll_w = (int)lay->off_ll_w; d.off_ll_b = (int)lay->off_ll_b;
  d.n_dense = (int)lay->n_dense;
  // 768 = three 512-thread workgroups on each of the MI355X's 256 CUs, all resident at once, every CU with the same share
  // (measured at C2, kernel time: 256 / 384 / 512 / 640 / 768 / 896 / 1024 / 1536 / 2048 workgroups -> 21.1 / 20.7 / 21.4 /
  // 22.2 / 19.3 / 21.1 / 22.0 / 26.3 / 25.8 us; 1 M-item table 1.440 -> 1.431 ms per step)
  int64_t grid = ((n >> 2) + 1 + 511) / 512;
  if (grid > 768) grid = 768;
  hipLaunchKernelGGL(adam_pack_kernel, dim3((int)grid), dim3(512), 0, (hipStream_t)stream, param, grad, m, v, n, n_table_pad,
                     n_zero, (float)beta1, (float)beta2, (float)eps, state, stats, d, packed, lr, beta1, beta2, table_bf16,
                     table_bf16 ? lay->n_table : 0);
  return (int)hipGetLastError();
}

extern "C" int srfrd_l2_norms(const float* param, const int64_t* seg_off, const int64_t* seg_len, int n_seg, int64_t n_table_pad,
                              double l2_emb, float* partial, float* l2buf, float* dense_scale, void* stream) {
  if (!param || !seg_off || !seg_len || !partial || !l2buf || !dense_scale || n_seg < 1 || n_seg > 128 || n_table_pad < 0)
    return SRFRD_E_ARG;
  const int n_tw = 240;
  hipLaunchKernelGGL(l2_partial_kernel, dim3(n_tw + n_seg - 1), dim3(256), 0, (hipStream_t)stream, param, seg_off, seg_len, n_tw,
                     partial);
  hipLaunchKernelGGL(l2_finish_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partial, seg_off, seg_len, n_seg, n_tw,
                     n_table_pad, (float)l2_emb, l2buf, dense_scale);
  return (int)hipGetLastError();
}

extern "C" int srfrd_l2_apply(float* grad, const float* param, int64_t i0, int64_t i1, int64_t n_table_pad, const float* l2buf,
                              const float* dense_scale, const float* stats, void* stream) {
  if (!grad || !param || !l2buf || !dense_scale || i0 < 0 || i1 < i0) return SRFRD_E_ARG;
  if (i0 == i1) return 0;
  int64_t grid = (i1 - i0 + 255) / 256;
  if (grid > 2048) grid = 2048;
  hipLaunchKernelGGL(l2_apply_kernel, dim3((int)grid), dim3(256), 0, (hipStream_t)stream, grad, param, i0, i1, n_table_pad, l2buf,
                     dense_scale, stats);
  return (int)hipGetLastError();
}

extern "C" int srfrd_loss_finalize(const float* stats, float* loss_out, void* stream) {
  if (!stats || !loss_out) return SRFRD_E_ARG;
  hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, (hipStream_t)stream, stats, loss_out);
  return (int)hipGetLastError();
}
