// Ranking-side kernels: user labels, candidate logits (predict), full-catalog top-k, HR@10 / NDCG@10 ranks.
// Reference: SRFR_model.py:144-152 (+ :241-259, :532-540, :668-681), :546-570; utils.py:576-598.
#include <mutex>

#include "srfrd_dev.h"

namespace srfrd {

// ---------------------------------------------------------------------------------------------
// get_Labels / SRFRN predict label: one wave per sequence (integer-exact)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) user_labels_kernel(int kind, const int64_t* __restrict__ fake_ids, int B, int L,
                                                         int64_t* __restrict__ labels) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const int lab = user_label_wave(kind, fake_ids + (int64_t)b * L, L, 0);      // (0: the raw label, as get_Labels returns it)
  if (lane == 0) labels[b] = lab;
}

// ---------------------------------------------------------------------------------------------
// id validation: err[0] |= 1 if any item id lies outside [0, n_items], |= 2 if any fake / label id outside [0, fake_hi].
// The compute kernels clamp ids (memory safety); this is what turns a bad id into the IndexError the reference's
// nn.Embedding raises - lazily, when the host reads the word.
// ---------------------------------------------------------------------------------------------
struct IdSets {
  const int64_t* item[3];
  const int64_t* fake[3];
};
__global__ void __launch_bounds__(256) check_ids_kernel(IdSets s, int64_t n, int64_t n_items, int64_t fake_hi, uint32_t* err) {
  uint32_t bad = 0;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
      if (s.item[k]) { const int64_t v = s.item[k][i]; if (v < 0 || v > n_items) bad |= 1u; }
      if (s.fake[k]) { const int64_t v = s.fake[k][i]; if (v < 0 || v > fake_hi) bad |= 2u; }
    }
  }
  if (__builtin_amdgcn_ballot_w64(bad != 0) != 0) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) bad |= __shfl_xor(bad, o, 64);
    if ((threadIdx.x & 63) == 0) atomicOr(err, bad);
  }
}

// ---------------------------------------------------------------------------------------------
// predict: one wave per (user, candidate); lane = channel
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) predict_logits_kernel(srfrd_layout ly, const void* __restrict__ table_any,
                                                            const float* __restrict__ dense, const float* __restrict__ hidden,
                                                            int B, int L, const int64_t* __restrict__ cand, int n_cand,
                                                            int64_t cand_stride, const int64_t* __restrict__ user_label,
                                                            float* __restrict__ logits) {
  const int lane = threadIdx.x & 63;
  const int64_t w = (int64_t)blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= (int64_t)B * n_cand) return;
  const int b = (int)(w / n_cand), i = (int)(w - (int64_t)b * n_cand);
  const int dout = ly.d_out, di = ly.d_item;
  const float h = lane < dout ? hidden[((int64_t)b * L + (L - 1)) * dout + lane] : 0.f;
  const int64_t id = clamp_id(cand[(int64_t)b * cand_stride + i], ly.n_items);
  const ItemTable table{ly.table_bf16 ? nullptr : (const float*)table_any, ly.table_bf16 ? (const uint16_t*)table_any : nullptr};
  float e = 0.f;
  if (lane < di) e = table(id * di + lane);
  else if (ly.kind == SRFRD_SRFRN && lane < ly.D) e = dense[ly.off_side + clamp_id(user_label[b], 2) * ly.d_fake + (lane - di)];
  const float s = wave_sum(h * e);
  if (lane == 0) logits[w] = s;
}

// ---------------------------------------------------------------------------------------------
// full-catalog top-k, exact, logits never written to HBM.  Threshold scheme in four launches:
//   A  every (16-user tile, 512-item chunk) logits tile on the fp32 matrix cores -> per (user, chunk) MAXIMUM only
//   T  per user: tau = k-th largest chunk maximum.  Each chunk maximum is a real score, so at least k scores are
//      >= tau, i.e. tau is a lower bound on the k-th best score and every top-k item scores >= tau
//   B  the same tiles again; scores >= tau are appended to the user's short candidate list (atomic cursor)
//   S  per user: k rounds of (argmax, remove) over the candidates (value desc, item id asc => stable sort order)
// A workgroup stages its chunk of item rows in LDS once and walks a strided subset of the user tiles, so small
// catalogs still fill the chip.  If a user's candidates overflow the list (mass ties), the flag makes the host
// launcher fall back to the per-chunk selection kernels below, which are exact for any input.
// ---------------------------------------------------------------------------------------------
constexpr int kChunk = 256;
constexpr int kCandMax = 2048;

struct Cand {
  float v;
  int32_t i;
};

__device__ __forceinline__ bool better(float v, int i, float bv, int bi) { return v > bv || (v == bv && i < bi); }

struct TopkArgs {
  srfrd_layout ly;
  const void* table;            // fp32 item table, or its bf16 shadow when ly.table_bf16
  const float *dense, *hidden;
  const int64_t* user_label;
  int B, L, exclude_pad, k, n_chunks, user_splits;
  int crows, wg_per_group;      // bf16 streams: item rows per chunk (256 / 512), persistent workgroups per user group
  int64_t item_lo, item_hi;
  float* cmax;          // (B, n_chunks)
  float* tau;           // (B)
  Cand* cand;           // (B, kCandMax)
  int32_t* ccnt;        // (B) + 1 overflow flag at [B]
};

// contiguous global [rows][cols] -> LDS [rows][ld] with ITER loads in flight per thread (clamped indices, no predicate
// between the loads): a plain `for (i = tid; ...) lds[..] = g[i]` copy pays one L2 / HBM round trip per iteration -
// 108 of them for a 512-row chunk, which was 85 % of this kernel's time.
__device__ __forceinline__ float as_f32(float v) { return v; }
__device__ __forceinline__ float as_f32(uint16_t v) { return bf16_to_f32(v); }
template <int ITER, class T>
__device__ __forceinline__ void stage_rows(lds_f* dst, int ld, const T* src, int rows, int cols) {
  const int n = rows * cols, nthr = blockDim.x, tid = threadIdx.x;
  for (int base = 0; base < n; base += ITER * nthr) {
    T v[ITER];
#pragma unroll
    for (int u = 0; u < ITER; ++u) v[u] = src[min(base + u * nthr + tid, n - 1)];
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
      const int i = base + u * nthr + tid;
      if (i < n) {
        const int r = i / cols, c = i - r * cols;
        dst[r * ld + c] = as_f32(v[u]);
      }
    }
  }
}
// a chunk of item rows from the table as the launch sees it (fp32, or the bf16 shadow)
template <int ITER>
__device__ __forceinline__ void stage_item_rows(lds_f* dst, int ld, const srfrd_layout& ly, const void* table, int64_t row0, int rows) {
  if (ly.table_bf16) stage_rows<ITER>(dst, ld, (const uint16_t*)table + row0 * ly.d_item, rows, ly.d_item);
  else stage_rows<ITER>(dst, ld, (const float*)table + row0 * ly.d_item, rows, ly.d_item);
}

// stage a chunk of item rows and then, for this workgroup's user tiles, leave the 16 x kChunk logits tile in sS
template <class F>
__device__ __forceinline__ void topk_tiles(const TopkArgs& a, F&& per_tile) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nw = blockDim.x >> 6;
  const srfrd_layout& ly = a.ly;
  const int di = ly.d_item, dout = ly.d_out, D = ly.D;
  const bool srfrn = ly.kind == SRFRD_SRFRN;
  const int DKi = (di + 3) & ~3, DSi = DKi + 2;
  const int SLD = kChunk + 2;
  lds_f* sE = (lds_f*)smem;
  lds_f* sH = sE + kChunk * DSi;
  lds_f* sS = sH + 16 * DSi;
  lds_f* sF = sS + 16 * SLD;
  const int chunk = blockIdx.x / a.user_splits, split = blockIdx.x - chunk * a.user_splits;
  const int64_t i0 = a.item_lo + (int64_t)chunk * kChunk;
  const int n_here = (int)((a.item_hi - i0) < kChunk ? (a.item_hi - i0) : kChunk);
  for (int idx = tid; idx < kChunk * DSi; idx += blockDim.x) {          // k-padding columns and rows past the catalog
    const int r = idx / DSi, c = idx - r * DSi;
    if (r >= n_here || c >= di) sE[idx] = 0.f;
  }
  stage_item_rows<8>(sE, DSi, a.ly, a.table, i0, n_here);
  for (int idx = tid; idx < 16 * (DSi - di); idx += blockDim.x) {       // k-padding columns of the user rows
    const int r = idx / (DSi - di), c = di + idx - r * (DSi - di);
    sH[r * DSi + c] = 0.f;
  }
  for (int u0 = split * 16; u0 < a.B; u0 += a.user_splits * 16) {
    __syncthreads();
    {                                                                   // last hidden state of 16 users: one round trip
      float hv[4];
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = min(u * (int)blockDim.x + tid, 16 * di - 1);
        const int r = i / di, c = i - r * di;
        const int ub = min(u0 + r, a.B - 1);
        hv[u] = a.hidden[((int64_t)ub * a.L + (a.L - 1)) * dout + c];
      }
#pragma unroll
      for (int u = 0; u < 4; ++u) {
        const int i = u * (int)blockDim.x + tid;
        if (i < 16 * di) {
          const int r = i / di, c = i - r * di;
          sH[r * DSi + c] = u0 + r < a.B ? hv[u] : 0.f;
        }
      }
    }
    if (srfrn && tid < 16) {
      float s = 0.f;
      if (u0 + tid < a.B) {
        const int lab = clamp_id(a.user_label[u0 + tid], 2);
        for (int c = di; c < D; ++c)
          s += a.hidden[((int64_t)(u0 + tid) * a.L + (a.L - 1)) * dout + c] * a.dense[ly.off_side + lab * ly.d_fake + (c - di)];
      }
      sF[tid] = s;
    }
    __syncthreads();
    // items on the M side (32 row tiles: a wave walks them four at a time, one hidden-state fragment feeding four
    // independent accumulator chains), the 16 users as the single column strip; the tile lands user-major in sS
    gemm_tiles<0>(nw, kChunk / 16, 1, DKi, Mat{sE, DSi}, MatT{sH, DSi}, [&](int r, int c, float v) {
      if (srfrn) v += sF[c];
      const bool ok = r < n_here && !(a.exclude_pad && i0 + r == 0);
      sS[c * SLD + r] = ok ? v : -INFINITY;
    });
    __syncthreads();
    per_tile(u0, chunk, i0, sS, SLD);
  }
}

// The same walk for the two threshold passes, without the score tile: every logit goes from the accumulator straight
// into elem(user_in_tile, item, value) - a running maximum (pass A) or a compare against tau (pass B) - so the
// 16 x kChunk tile never makes the round trip through LDS.  Eight waves, the next user tile's hidden rows are
// requested before the current tile's GEMM and land in the other half of a two-deep LDS buffer.
template <class BEG, class ELEM, class END>
__device__ __forceinline__ void topk_stream(const TopkArgs& a, BEG&& begin, ELEM&& elem, END&& end) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int tid = threadIdx.x, nw = blockDim.x >> 6, nthr = blockDim.x;
  const srfrd_layout& ly = a.ly;
  const int di = ly.d_item, dout = ly.d_out, D = ly.D;
  const bool srfrn = ly.kind == SRFRD_SRFRN;
  const int DKi = (di + 3) & ~3, DSi = DKi + 2;
  lds_f* sE = (lds_f*)smem;
  lds_f* sH = sE + kChunk * DSi;          // [2][16][DSi]
  lds_f* sF = sH + 2 * 16 * DSi;          // [2][16]
  lds_f* sM = sF + 32;                    // scratch of the end() step: [nw][16]
  const int chunk = blockIdx.x / a.user_splits, split = blockIdx.x - chunk * a.user_splits;
  const int64_t i0 = a.item_lo + (int64_t)chunk * kChunk;
  const int n_here = (int)((a.item_hi - i0) < kChunk ? (a.item_hi - i0) : kChunk);
  for (int idx = tid; idx < kChunk * DSi; idx += nthr) {               // k-padding columns and rows past the catalog
    const int r = idx / DSi, c = idx - r * DSi;
    if (r >= n_here || c >= di) sE[idx] = 0.f;
  }
  stage_item_rows<8>(sE, DSi, a.ly, a.table, i0, n_here);
  for (int idx = tid; idx < 2 * 16 * (DSi - di); idx += nthr) {        // k-padding columns of the user rows
    const int r = idx / (DSi - di), c = di + idx - r * (DSi - di);
    sH[r * DSi + c] = 0.f;
  }
  // hidden rows of a user tile: element e of thread tid is (user e*nthr+tid / di, channel ...); two per thread at 512 threads
  constexpr int HIT = 4;
  float hv[HIT];
  float fside = 0.f;
  auto fetch = [&](int u0) {
#pragma unroll
    for (int u = 0; u < HIT; ++u) {
      const int i = min(u * nthr + tid, 16 * di - 1);
      const int r = i / di, c = i - r * di;
      hv[u] = a.hidden[((int64_t)min(u0 + r, a.B - 1) * a.L + (a.L - 1)) * dout + c];
    }
    if (srfrn && tid < 16) {
      float sacc = 0.f;
      if (u0 + tid < a.B) {
        const int lab = clamp_id(a.user_label[u0 + tid], 2);
        for (int c = di; c < D; ++c)
          sacc += a.hidden[((int64_t)(u0 + tid) * a.L + (a.L - 1)) * dout + c] * a.dense[ly.off_side + lab * ly.d_fake + (c - di)];
      }
      fside = sacc;
    }
  };
  auto put = [&](int u0, int buf) {
#pragma unroll
    for (int u = 0; u < HIT; ++u) {
      const int i = u * nthr + tid;
      if (i < 16 * di) {
        const int r = i / di, c = i - r * di;
        sH[(buf * 16 + r) * DSi + c] = u0 + r < a.B ? hv[u] : 0.f;
      }
    }
    if (srfrn && tid < 16) sF[buf * 16 + tid] = fside;
  };
  const int ustep = a.user_splits * 16;
  int u0 = split * 16, cur = 0;
  if (u0 < a.B) { fetch(u0); put(u0, 0); }
  __syncthreads();
  for (; u0 < a.B; u0 += ustep) {
    const bool more = u0 + ustep < a.B;
    if (more) fetch(u0 + ustep);
    begin(u0);
    const lds_f* hcur = sH + cur * 16 * DSi;
    const lds_f* fcur = sF + cur * 16;
    // items on the M side (a wave walks its row tiles four at a time: one hidden-state fragment, four independent
    // accumulator chains), the 16 users as the single column strip
    gemm_tiles<0>(nw, kChunk / 16, 1, DKi, Mat{sE, DSi}, MatT{hcur, DSi}, [&](int r, int c, float v) {
      if (srfrn) v += fcur[c];
      const bool ok = r < n_here && !(a.exclude_pad && i0 + r == 0);
      elem(u0, c, i0 + r, ok ? v : -INFINITY);
    });
    if (more) put(u0 + ustep, cur ^ 1);
    end(u0, chunk, sM);
    __syncthreads();
    cur ^= 1;
  }
}

// ---- bf16 item table (srfrd_layout::table_bf16): the threshold passes on the bf16 matrix cores --------------------------
// The table rows are exact bf16; a hidden state splits EXACTLY into three bf16 terms (h = h1 + h2 + h3: 3 x 8 significand
// bits), so logit = sum_k (h1 + h2 + h3)_k e_k is three v_mfma_f32_16x16x32_bf16 products per 32-deep k-step with exact
// products and fp32 accumulation - fp32-grade logits at 6 x 16 cycles per 16 x 16 tile instead of 13 x 32 on the fp32 form.
//
// Shape of the launch: the USERS live in registers, the ITEMS stream past them.  A 16-wave workgroup (one per CU,
// persistent) serves a group of 16 x 16 x NU users: wave w keeps the split hidden-state fragments of user tiles w, w + 16
// (NU x 24 registers) for the whole launch, so nothing about a user is ever staged again.  The workgroup walks its share
// of the item chunks (256 or 512 rows): a chunk is copied table -> registers -> LDS (rows padded to 144 B) while the
// previous one is being multiplied (two buffers, ONE barrier per chunk), and every wave reads every item tile of the chunk
// from LDS (two ds_read_b128 feed 6 NU MFMAs).  With users on the accumulator column, the per-(user, chunk) maximum and
// the candidate test are wave-local: no cross-wave reduction, no atomics in the maximum pass.
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __attribute__((address_space(3))) uint16_t lds_u16;
typedef __attribute__((address_space(3))) uint32_t lds_u32;
typedef __attribute__((address_space(3))) bf16x8 lds_bf16x8;
constexpr int kWaves16 = 16;                    // waves per workgroup of the bf16 streams
constexpr int kChunk16 = 512;                   // most item rows of a chunk
constexpr int kStageSlots16 = 13;               // staging registers per thread: a chunk's rows x copy elements per row <= 13 x 1024
constexpr int kHS = 72;                         // bf16 row stride of an item row in LDS (64 + 8: rows 144 B apart)
constexpr int kStream16Lds = 2 * kChunk16 * kHS * 2;

// begin(ut, user0) once per owned user tile; elem(ut, user0, item0, v) with the lane's four consecutive items of user user0 + li;
// end(ut, user0, chunk) once per (user tile, chunk); need(chunk, user0[]) -> wave-uniform bool: false lets the wave skip the
// chunk's tiles (it still helps staging the next chunk and meets the barriers).
// SPLIT_E = false: bf16 table (rows copied as they are; 512 / 256-row chunks, two LDS buffers, one barrier per chunk).
// SPLIT_E = true : fp32 table.  A chunk's rows are split into three exact bf16 planes e1 + e2 + e3 while they are staged
//   (256-row chunks, ONE buffer of three planes = 110 KB: the next chunk waits in registers and is split / written between
//   two barriers), and a logit takes the six products whose weight is above 2^-24 of the largest one - h1e1, h1e2, h2e1,
//   h1e3, h2e2, h3e1 (smallest first) - twelve bf16 MFMAs per 16 x 16 tile against thirteen fp32 ones at twice the
//   cycles each, with fp32-grade results (the dropped products are at the level of an fp32 rounding).
template <int NU, bool SPLIT_E, class BEG, class ELEM, class END, class NEED>
__device__ __forceinline__ void topk_stream16(const TopkArgs& a, BEG&& begin, ELEM&& elem, END&& end, NEED&& need) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  constexpr int NE = SPLIT_E ? 3 : 1;                  // bf16 planes of an item chunk
  constexpr int CH = SPLIT_E ? 256 : kChunk16;         // rows of a plane
  const int tid = threadIdx.x, nthr = kWaves16 * 64;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), li = lane & 15, lq = lane >> 4;
  const srfrd_layout& ly = a.ly;
  const int di = ly.d_item, dout = ly.d_out, D = ly.D;
  const bool srfrn = ly.kind == SRFRD_SRFRN;
  lds_u16* sE = (lds_u16*)smem;                               // [2][512][kHS]  or  [3][256][kHS]
  const int group = blockIdx.x / a.wg_per_group, pw = blockIdx.x - group * a.wg_per_group;
  for (int idx = tid; idx < kStream16Lds / 4; idx += nthr) ((lds_u32*)sE)[idx] = 0;      // k-padding columns stay zero
  // ---- this wave's users: split hidden fragments (lane (li, lq): user tile row li, k = 32 ks + 8 lq + 0..7) and side terms
  bf16x8 bfr[NU][3][2];
  float fs[NU];
  int user0[NU];
#pragma unroll
  for (int n = 0; n < NU; ++n) {
    user0[n] = (group * kWaves16 * NU + n * kWaves16 + wave) << 4;
    const int ub = min(user0[n] + li, a.B - 1);
    const float* hrow = a.hidden + ((int64_t)ub * a.L + (a.L - 1)) * dout;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      union { bf16x8 v; uint16_t h[8]; } t1, t2, t3;
      float xv[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) xv[e] = hrow[min(32 * ks + 8 * lq + e, di - 1)];      // (clamped, unconditional: one round trip)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int k = 32 * ks + 8 * lq + e;
        const float x = k < di ? xv[e] : 0.f;
        const uint16_t h1 = f32_to_bf16(x);
        const float r1 = x - bf16_to_f32(h1);
        const uint16_t h2 = f32_to_bf16(r1);
        const float r2 = r1 - bf16_to_f32(h2);
        t1.h[e] = h1; t2.h[e] = h2; t3.h[e] = f32_to_bf16(r2);
      }
      bfr[n][0][ks] = t1.v; bfr[n][1][ks] = t2.v; bfr[n][2][ks] = t3.v;
    }
    float sacc = 0.f;
    if (srfrn) {
      const int lab = clamp_id(a.user_label[ub], 2);
      for (int c = di; c < D; ++c) sacc += hrow[c] * a.dense[ly.off_side + lab * ly.d_fake + (c - di)];
    }
    fs[n] = sacc;
    begin(n, user0[n]);
  }
  // ---- item chunks: table -> registers -> LDS; thread t owns copy elements t, t + nthr, ... of the chunk
  const bool dw = !SPLIT_E && (di & 1) == 0;           // bf16 rows that are whole dwords (d_item even): 4-byte copies
  const int rw = SPLIT_E ? di : (dw ? di >> 1 : di);   // copy elements per row
  constexpr int SIT = kStageSlots16;                   // 512 x 25 dwords / 256 x 50 floats over 1024 threads
  uint32_t sv[SIT];
  // (the thread index is laundered at every use: its per-slot row / column split is cheap to redo and would otherwise
  // sit in 26 registers for the whole launch)
  auto fetch = [&](int chunk) {
    const int64_t i0 = a.item_lo + (int64_t)chunk * a.crows;
    const int nrow = (int)((a.item_hi - i0) < a.crows ? (a.item_hi - i0) : a.crows);
    const int n = nrow * rw;
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if (SPLIT_E) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>((const float*)a.table + i0 * di);
#pragma unroll
      for (int u = 0; u < SIT; ++u) sv[u] = src[min(u * nthr + tl, n - 1)];
    } else if (dw) {
      const uint32_t* src = reinterpret_cast<const uint32_t*>((const uint16_t*)a.table + i0 * di);
#pragma unroll
      for (int u = 0; u < SIT; ++u) sv[u] = src[min(u * nthr + tl, n - 1)];
    } else {
      const uint16_t* src = (const uint16_t*)a.table + i0 * di;
#pragma unroll
      for (int u = 0; u < SIT; ++u) sv[u] = (uint32_t)src[min(u * nthr + tl, n - 1)];
    }
  };
  auto put = [&](int chunk, int buf) {
    const int64_t i0 = a.item_lo + (int64_t)chunk * a.crows;
    const int nrow = (int)((a.item_hi - i0) < a.crows ? (a.item_hi - i0) : a.crows);
    const int n = nrow * rw;
    int tl = tid;
    asm volatile("" : "+v"(tl));
    if (SPLIT_E) {
#pragma unroll
      for (int u = 0; u < SIT; ++u) {
        const int i = u * nthr + tl;
        const int r = i / rw, c = i - r * rw;
        if (i < n) {
          const float x = __uint_as_float(sv[u]);
          const uint16_t e1 = f32_to_bf16(x);
          const float r1 = x - bf16_to_f32(e1);
          const uint16_t e2 = f32_to_bf16(r1);
          const float r2 = r1 - bf16_to_f32(e2);
          sE[(0 * CH + r) * kHS + c] = e1;
          sE[(1 * CH + r) * kHS + c] = e2;
          sE[(2 * CH + r) * kHS + c] = f32_to_bf16(r2);
        }
      }
    } else if (dw) {
      lds_u32* dst = (lds_u32*)sE + buf * CH * (kHS / 2);
#pragma unroll
      for (int u = 0; u < SIT; ++u) {
        const int i = u * nthr + tl;
        const int r = i / rw, c = i - r * rw;
        if (i < n) dst[r * (kHS / 2) + c] = sv[u];
      }
    } else {
      lds_u16* dst = sE + buf * CH * kHS;
#pragma unroll
      for (int u = 0; u < SIT; ++u) {
        const int i = u * nthr + tl;
        const int r = i / rw, c = i - r * rw;
        if (i < n) dst[r * kHS + c] = (uint16_t)sv[u];
      }
    }
  };
  // (d_item odd, bf16: 512 x d_item 2-byte elements need more than SIT slots per thread at 512 rows - the launcher then uses
  // 256-row chunks, which fit for d_item <= 51; fp32: 256 x d_item <= 13 x 1024 for d_item <= 52; wider tables take the
  // fp32-matrix path)
  int chunk = pw, cur = 0;
  __syncthreads();                       // (the zero fill above)
  if (chunk < a.n_chunks) { fetch(chunk); put(chunk, 0); }
  bool work = chunk < a.n_chunks ? need(chunk, user0) : false;
  __syncthreads();
  for (; chunk < a.n_chunks; chunk += a.wg_per_group) {
    const int nxt = chunk + a.wg_per_group;
    if (nxt < a.n_chunks) fetch(nxt);
    const bool work_next = nxt < a.n_chunks ? need(nxt, user0) : false;      // (its loads land under this chunk's tiles)
    const int64_t i0 = a.item_lo + (int64_t)chunk * a.crows;
    const int ntile = a.crows >> 4;
    // every row of the chunk is a rankable item (all chunks but the last one and, with exclude_pad, the one holding item 0):
    // the epilogue then has no per-element validity arithmetic - at 12 MFMAs per item tile it would cost more issue slots
    // than the MFMAs themselves
    const bool full = a.item_hi - i0 >= a.crows && !(a.exclude_pad && i0 == 0);
    const lds_u16* rowp = sE + ((SPLIT_E ? 0 : cur) * CH + li) * kHS + 8 * lq;
    struct AF { bf16x8 f[NE][2]; };
    auto tile_load = [&](int t) {
      AF r;
#pragma unroll
      for (int pl = 0; pl < NE; ++pl) {
        r.f[pl][0] = *reinterpret_cast<const lds_bf16x8*>(rowp + (pl * CH + t * 16) * kHS);
        r.f[pl][1] = *reinterpret_cast<const lds_bf16x8*>(rowp + (pl * CH + t * 16) * kHS + 32);
      }
      return r;
    };
    // the accumulation chains of the NU user tiles, interleaved (independent chains back to back), smallest products first
    auto tile_mma = [&](const AF& af, f32x4 (&acc)[NU]) {
#pragma unroll
      for (int n = 0; n < NU; ++n) acc[n] = f32x4{0.f, 0.f, 0.f, 0.f};
      if constexpr (SPLIT_E) {
        constexpr int TH[6] = {2, 1, 0, 1, 0, 0}, TE[6] = {0, 1, 2, 0, 1, 0};
#pragma unroll
        for (int q = 0; q < 6; ++q) {
#pragma unroll
          for (int n = 0; n < NU; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.f[TE[q]][0], bfr[n][TH[q]][0], acc[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < NU; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.f[TE[q]][1], bfr[n][TH[q]][1], acc[n], 0, 0, 0);
        }
      } else {
#pragma unroll
        for (int tm = 2; tm >= 0; --tm) {
#pragma unroll
          for (int n = 0; n < NU; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.f[0][0], bfr[n][tm][0], acc[n], 0, 0, 0);
#pragma unroll
          for (int n = 0; n < NU; ++n) acc[n] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af.f[0][1], bfr[n][tm][1], acc[n], 0, 0, 0);
        }
      }
    };
    auto tile_out = [&](int t, f32x4 (&acc)[NU], bool masked) {
      const int64_t item0 = i0 + (t << 4) + (lq << 2);
#pragma unroll
      for (int n = 0; n < NU; ++n) {
#pragma unroll
        for (int e = 0; e < 4; ++e) acc[n][e] += fs[n];          // (0 unless SRFRN: unconditional beats a per-element select)
        if (masked) {
#pragma unroll
          for (int e = 0; e < 4; ++e) {
            const bool ok = item0 + e < a.item_hi && !(a.exclude_pad && item0 + e == 0);
            acc[n][e] = ok ? acc[n][e] : -INFINITY;
          }
        }
        elem(n, user0[n], item0, acc[n]);
      }
    };
    if (!work) {
      // nothing in this chunk can matter to this wave's users
    } else if (full) {
      // software-pipelined over the item tiles: the MFMAs of tile t are in flight while tile t - 1 leaves its accumulators
      // and the fragments of tile t + 1 arrive from LDS (two tiles per trip: the accumulator sets alternate statically)
      f32x4 accA[NU], accB[NU];
      tile_mma(tile_load(0), accA);
      for (int t = 1; t + 1 < ntile; t += 2) {
        tile_mma(tile_load(t), accB);
        tile_out(t - 1, accA, false);
        tile_mma(tile_load(t + 1), accA);
        tile_out(t, accB, false);
      }
      // (ntile is even - 16 or 32: the loop leaves tile ntile - 2 in accA and tile ntile - 1 to do)
      tile_mma(tile_load(ntile - 1), accB);
      tile_out(ntile - 2, accA, false);
      tile_out(ntile - 1, accB, false);
    } else {
      for (int t = 0; t < ntile; ++t) {
        f32x4 acc[NU];
        tile_mma(tile_load(t), acc);
        tile_out(t, acc, true);
      }
    }
    if (work) {
#pragma unroll
      for (int n = 0; n < NU; ++n) end(n, user0[n], chunk);
    }
    work = work_next;
    if (SPLIT_E) {
      __syncthreads();                   // every wave is done reading the (single) buffer
      if (nxt < a.n_chunks) put(nxt, 0);
      __syncthreads();
    } else {
      if (nxt < a.n_chunks) put(nxt, cur ^ 1);
      __syncthreads();
      cur ^= 1;
    }
  }
}

// max(a, b, c) as ONE v_max3_f32 (fmaxf compiles to a canonicalising v_max per operand first: the logits are never NaN)
__device__ __forceinline__ float vmax3(float x, float y, float z) {
  float r;
  asm("v_max3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(x), "v"(y), "v"(z));
  return r;
}

template <int NU, bool SPLIT_E>
__global__ void __launch_bounds__(kWaves16 * 64) topk_max16_kernel(const TopkArgs a) {
  const int lane = threadIdx.x & 63;
  float m[NU];
  topk_stream16<NU, SPLIT_E>(a,
      [&](int n, int) { m[n] = -INFINITY; },
      [&](int n, int, int64_t, const f32x4& v) { m[n] = vmax3(vmax3(m[n], v[0], v[1]), v[2], v[3]); },
      [&](int n, int u0, int chunk) {
        float mm = fmaxf(m[n], __shfl_xor(m[n], 16, 64));
        mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
        if (lane < 16 && u0 + lane < a.B) a.cmax[(int64_t)(u0 + lane) * a.n_chunks + chunk] = mm;
        m[n] = -INFINITY;
      },
      [&](int, const int*) { return true; });
}

template <int NU, bool SPLIT_E>
__global__ void __launch_bounds__(kWaves16 * 64) topk_collect16_kernel(const TopkArgs a) {
  const int li = threadIdx.x & 15;
  float tau[NU];
  topk_stream16<NU, SPLIT_E>(a,
      [&](int n, int u0) { tau[n] = u0 + li < a.B ? a.tau[u0 + li] : INFINITY; },
      [&](int n, int u0, int64_t item0, const f32x4& v) {
        // hits are rare (about k per user over the whole catalog): one test of the four-item maximum, then a compact loop
        if (!(vmax3(vmax3(v[0], v[1], v[2]), v[3], v[3]) >= tau[n])) return;
        unsigned hm = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) hm |= (v[e] != -INFINITY && v[e] >= tau[n]) ? 1u << e : 0u;
        while (hm) {
          const int e = __ffs(hm) - 1;
          hm &= hm - 1;
          const float ve = e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3];
          const int b = u0 + li;
          const int slot = atomicAdd(&a.ccnt[b], 1);
          if (slot < kCandMax) a.cand[(int64_t)b * kCandMax + slot] = Cand{ve, (int32_t)(item0 + e)};
          else a.ccnt[a.B] = 1;
        }
      },
      [&](int, int, int) {},
      // a chunk can hold a candidate of user u only if its maximum (pass A) reaches tau_u: about k chunks per user do, so a
      // wave (32 users) multiplies ~1 chunk in 6 and skips the rest
      [&](int chunk, const int* u0s) {
        bool hit = false;
#pragma unroll
        for (int n = 0; n < NU; ++n) {
          const int u = u0s[n] + li;
          if (u < a.B) hit |= a.cmax[(int64_t)u * a.n_chunks + chunk] >= tau[n];
        }
        return __any(hit) != 0;
      });
}

__global__ void __launch_bounds__(512) topk_max_kernel(const TopkArgs a) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
  float m = -INFINITY;                     // this lane's user column (lane & 15) over the row tiles of its wave
  topk_stream(a,
      [&](int) { m = -INFINITY; },
      [&](int, int, int64_t, float v) { m = fmaxf(m, v); },
      [&](int u0, int chunk, lds_f* sM) {
        float mm = fmaxf(m, __shfl_xor(m, 16, 64));
        mm = fmaxf(mm, __shfl_xor(mm, 32, 64));
        if (lane < 16) sM[wave * 16 + lane] = mm;
        __syncthreads();
        if (threadIdx.x < 16 && u0 + (int)threadIdx.x < a.B) {
          float t = -INFINITY;
          for (int w = 0; w < nw; ++w) t = fmaxf(t, sM[w * 16 + threadIdx.x]);
          a.cmax[(int64_t)(u0 + threadIdx.x) * a.n_chunks + chunk] = t;
        }
      });
}

// tau[b] = k-th largest of cmax[b][:] (one wave per user; -inf if fewer than k finite maxima); also resets the cursor.
// The row is staged in LDS once (all loads in flight together) and the k argmax rounds walk LDS: k rounds over global
// memory were k x n_chunks / 64 dependent round trips (49 us at 1 M items).
__global__ void __launch_bounds__(256) topk_tau_kernel(const TopkArgs a) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int b = blockIdx.x * (blockDim.x >> 6) + wv;
  if (b >= a.B) return;
  const float* grow = a.cmax + (int64_t)b * a.n_chunks;
  lds_f* row = (lds_f*)smem + (int64_t)wv * a.n_chunks;
  for (int j0 = 0; j0 < a.n_chunks; j0 += 64 * 8) {
    float v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = grow[min(j0 + u * 64 + lane, a.n_chunks - 1)];
#pragma unroll
    for (int u = 0; u < 8; ++u)
      if (j0 + u * 64 + lane < a.n_chunks) row[j0 + u * 64 + lane] = v[u];
  }
  __builtin_amdgcn_wave_barrier();
  float tau = -INFINITY;
  for (int r = 0; r < a.k; ++r) {
    float bv = -INFINITY;
    int bp = -1;
    for (int j0 = 0; j0 < a.n_chunks; j0 += 64 * 8) {       // eight independent LDS reads in flight per trip
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const int j = j0 + u * 64 + lane;
        v[u] = j < a.n_chunks ? row[j] : -INFINITY;
      }
#pragma unroll
      for (int u = 0; u < 8; ++u)
        if (v[u] > bv) { bv = v[u]; bp = j0 + u * 64 + lane; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (ov > bv || (ov == bv && op >= 0 && (bp < 0 || op < bp))) { bv = ov; bp = op; }
    }
    if (bp < 0) { tau = -INFINITY; break; }       // fewer than k chunks hold a finite score: keep everything finite
    tau = bv;
    if (lane == 0) row[bp] = -INFINITY;
    __builtin_amdgcn_wave_barrier();
  }
  if (lane == 0) {
    a.tau[b] = tau;
    a.ccnt[b] = 0;
    if (b == 0) a.ccnt[a.B] = 0;
  }
}

__global__ void __launch_bounds__(512) topk_collect_kernel(const TopkArgs a) {
  const int li = threadIdx.x & 15;
  float tau = INFINITY;
  topk_stream(a,
      [&](int u0) { tau = u0 + li < a.B ? a.tau[u0 + li] : INFINITY; },
      [&](int u0, int c, int64_t item, float v) {
        if (v != -INFINITY && v >= tau) {
          const int b = u0 + c;
          const int slot = atomicAdd(&a.ccnt[b], 1);
          if (slot < kCandMax) a.cand[(int64_t)b * kCandMax + slot] = Cand{v, (int32_t)item};
          else a.ccnt[a.B] = 1;                     // overflow: the launcher re-runs the exhaustive path
        }
      },
      [&](int, int, lds_f*) {});
}

__global__ void __launch_bounds__(256) topk_select_kernel(const TopkArgs a, int64_t* __restrict__ topk_idx,
                                                         float* __restrict__ topk_val) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= a.B) return;
  Cand* c = a.cand + (int64_t)b * kCandMax;
  const int n = min(a.ccnt[b], kCandMax);
  for (int r = 0; r < a.k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF, bp = -1;
    for (int j = lane; j < n; j += 64) {
      const float v = c[j].v;
      const int id = c[j].i;
      if (id >= 0 && better(v, id, bv, bi)) { bv = v; bi = id; bp = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (op >= 0 && (bp < 0 || better(ov, oi, bv, bi))) { bv = ov; bi = oi; bp = op; }
    }
    if (lane == 0) {
      topk_idx[(int64_t)b * a.k + r] = bp >= 0 ? bi : -1;
      topk_val[(int64_t)b * a.k + r] = bp >= 0 ? bv : -INFINITY;
      if (bp >= 0) c[bp].i = -1;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
  }
}

// ---- exhaustive fallback (exact for any input, e.g. every score tied): per-chunk selection + merge ----------
// k rounds of (argmax, remove) over `n` scores in LDS by one wave; writes k candidates
__device__ __forceinline__ void wave_select_topk(lds_f* sc, const int* ids, int n, int k, Cand* out) {
  const int lane = threadIdx.x & 63;
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF, bp = -1;
    for (int j = lane; j < n; j += 64) {
      const float v = sc[j];
      const int id = ids ? ids[j] : j;
      if (v != -INFINITY && better(v, id, bv, bi)) { bv = v; bi = id; bp = j; }   // -inf marks removed / invalid
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; bp = op; }
    }
    if (lane == 0) {
      out[r].v = bv;
      out[r].i = bp >= 0 ? bi : -1;
      if (bp >= 0) sc[bp] = -INFINITY;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

__global__ void __launch_bounds__(256) topk_stage1_kernel(srfrd_layout ly, const void* __restrict__ table,
                                                         const float* __restrict__ dense, const float* __restrict__ hidden,
                                                         int B, int L, int64_t item_lo, int64_t item_hi, int exclude_pad,
                                                         const int64_t* __restrict__ user_label, int k, int n_chunks,
                                                         Cand* __restrict__ ws, const int32_t* __restrict__ overflow) {
  if (*overflow == 0) return;                      // the threshold scheme succeeded: nothing to do
  TopkArgs a = {};
  a.ly = ly; a.table = table; a.dense = dense; a.hidden = hidden; a.user_label = user_label;
  a.B = B; a.L = L; a.exclude_pad = exclude_pad; a.k = k; a.n_chunks = n_chunks; a.user_splits = 1;
  a.item_lo = item_lo; a.item_hi = item_hi;
  topk_tiles(a, [&](int u0, int chunk, int64_t i0, lds_f* sS, int SLD) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, nw = blockDim.x >> 6;
    for (int r = wave; r < 16; r += nw)
      if (u0 + r < B) {
        Cand* out = ws + ((int64_t)(u0 + r) * n_chunks + chunk) * k;
        wave_select_topk(sS + r * SLD, nullptr, kChunk, k, out);
        if (lane == 0)
          for (int q = 0; q < k; ++q)
            if (out[q].i >= 0) out[q].i += (int32_t)i0;       // chunk-local position -> item id
      }
  });
}

__global__ void __launch_bounds__(256) topk_stage2_kernel(const Cand* __restrict__ ws, int B, int k, int n_chunks,
                                                         int64_t* __restrict__ topk_idx, float* __restrict__ topk_val,
                                                         const int32_t* __restrict__ overflow) {
  if (*overflow == 0) return;
  // one wave per user: k rounds of argmax over its n_chunks*k candidates (kept in global; -inf marks removed)
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  Cand* c = const_cast<Cand*>(ws) + (int64_t)b * n_chunks * k;
  const int n = n_chunks * k;
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF, bp = -1;
    for (int j = lane; j < n; j += 64) {
      const float v = c[j].v;
      const int id = c[j].i;
      if (id >= 0 && v != -INFINITY && better(v, id, bv, bi)) { bv = v; bi = id; bp = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (better(ov, oi, bv, bi)) { bv = ov; bi = oi; bp = op; }
    }
    if (lane == 0) {
      topk_idx[(int64_t)b * k + r] = bp >= 0 ? bi : -1;
      topk_val[(int64_t)b * k + r] = bv;
      if (bp >= 0) c[bp].i = -1;
    }
    __threadfence_block();
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// merge of per-shard top-k lists (row-sharded catalog, BASELINE configs[4]): per user n_cand = shards x k candidates
// (idx < 0 = empty slot) -> the k best in stable descending order (value desc, item id asc) - the order one unsharded
// ranking returns, ties across shard boundaries included.  One wave per user, k rounds of (argmax, remove).
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) topk_merge_kernel(const int64_t* __restrict__ cand_idx, const float* __restrict__ cand_val,
                                                        int B, int n_cand, int k, int64_t* __restrict__ out_idx,
                                                        float* __restrict__ out_val) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int b = blockIdx.x * (blockDim.x >> 6) + wave;
  if (b >= B) return;
  float* sv = smem + (size_t)wave * 2 * n_cand;
  int* si = (int*)(sv + n_cand);
  for (int j = lane; j < n_cand; j += 64) {
    const int64_t id = cand_idx[(int64_t)b * n_cand + j];
    sv[j] = cand_val[(int64_t)b * n_cand + j];
    si[j] = id < 0 ? -1 : (int)id;
  }
  __builtin_amdgcn_wave_barrier();
  for (int r = 0; r < k; ++r) {
    float bv = -INFINITY;
    int bi = 0x7FFFFFFF, bp = -1;
    for (int j = lane; j < n_cand; j += 64) {
      const int id = si[j];
      if (id >= 0 && (bp < 0 || better(sv[j], id, bv, bi))) { bv = sv[j]; bi = id; bp = j; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      const float ov = __shfl_xor(bv, o, 64);
      const int oi = __shfl_xor(bi, o, 64);
      const int op = __shfl_xor(bp, o, 64);
      if (op >= 0 && (bp < 0 || better(ov, oi, bv, bi))) { bv = ov; bi = oi; bp = op; }
    }
    if (lane == 0) {
      out_idx[(int64_t)b * k + r] = bp >= 0 ? bi : -1;
      out_val[(int64_t)b * k + r] = bp >= 0 ? bv : -INFINITY;
      if (bp >= 0) si[bp] = -1;
    }
    __builtin_amdgcn_wave_barrier();
  }
}

// ---------------------------------------------------------------------------------------------
// rank of candidate 0 (strictly-greater count) + HR@10 / NDCG@10 accumulation (fp64, as the host loop does)
// ---------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) eval_rank_kernel(const float* __restrict__ logits, int B, int n_cand,
                                                       int32_t* __restrict__ rank, double* __restrict__ metric_acc) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  const float* row = logits + (int64_t)b * n_cand;
  const float x0 = row[0];
  int cnt = 0;
  for (int j = 1 + lane; j < n_cand; j += 64) cnt += row[j] > x0;
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) cnt += __shfl_xor(cnt, o, 64);
  if (lane == 0) {
    if (rank) rank[b] = cnt;
    if (metric_acc) {
      if (cnt < 10) {
        atomicAdd(&metric_acc[0], 1.0 / log2((double)cnt + 2.0));
        atomicAdd(&metric_acc[1], 1.0);
      }
      atomicAdd(&metric_acc[2], 1.0);
    }
  }
}

}  // namespace srfrd

using namespace srfrd;

extern "C" int srfrd_check_ids(const int64_t* item_a, const int64_t* item_b, const int64_t* item_c, const int64_t* fake_a,
                               const int64_t* fake_b, const int64_t* fake_c, int64_t n, int64_t n_items, int64_t fake_hi,
                               uint32_t* err_word, void* stream) {
  if (!err_word || n < 0 || n_items < 0) return SRFRD_E_ARG;
  if (n == 0 || !(item_a || item_b || item_c || fake_a || fake_b || fake_c)) return 0;
  IdSets s{{item_a, item_b, item_c}, {fake_a, fake_b, fake_c}};
  const int grid = (int)((n + 255) / 256 < 512 ? (n + 255) / 256 : 512);
  hipLaunchKernelGGL(check_ids_kernel, dim3(grid), dim3(256), 0, (hipStream_t)stream, s, n, n_items, fake_hi, err_word);
  return (int)hipGetLastError();
}

extern "C" int srfrd_user_labels(int kind, const int64_t* fake_ids, int B, int L, int64_t* labels, void* stream) {
  if (!fake_ids || !labels || B <= 0 || L <= 0) return SRFRD_E_ARG;
  if (!(kind == SRFRD_SRFU_B || kind == SRFRD_SRFU_F || kind == SRFRD_SRFU_R || kind == SRFRD_SRFRN)) return SRFRD_E_ARG;
  hipLaunchKernelGGL(user_labels_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, kind, fake_ids, B, L, labels);
  return (int)hipGetLastError();
}

extern "C" int srfrd_predict_logits(const srfrd_layout* lay, const void* item_table, const float* dense,
                                    const float* hidden, int B, int L, const int64_t* cand, int n_cand, int64_t cand_stride,
                                    const int64_t* user_label, float* logits, void* stream) {
  if (!lay || !item_table || !dense || !hidden || !cand || !logits || B <= 0 || L <= 0 || n_cand <= 0) return SRFRD_E_ARG;
  if (lay->D > SRFRD_MAX_D) return SRFRD_E_UNSUPPORTED;
  if (lay->kind == SRFRD_SRFRN && !user_label) return SRFRD_E_ARG;
  const int64_t waves = (int64_t)B * n_cand;
  hipLaunchKernelGGL(predict_logits_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, (hipStream_t)stream, *lay,
                     item_table, dense, hidden, B, L, cand, n_cand, cand_stride, user_label, logits);
  return (int)hipGetLastError();
}

// tau launch: the rows of 4 users per block while they fit the default 64 KiB of dynamic LDS, else one user per block
// (with the > 64 KiB opt-in up to the CU's 160 KiB: 40 k chunks)
static int launch_tau(const TopkArgs& a, hipStream_t st) {
  const size_t row = (size_t)a.n_chunks * sizeof(float);
  const int wpb = 4 * row <= 64 * 1024 ? 4 : 1;
  const size_t lds = wpb * row;
  if (lds > (size_t)kLdsLimit) return SRFRD_E_UNSUPPORTED;
  if (lds > 64 * 1024) {
    static std::mutex mu;
    static size_t opted[64] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SRFRD_E_DEVICE;
    std::lock_guard<std::mutex> lock(mu);
    if (lds > opted[dev]) {
      if (hipFuncSetAttribute((const void*)topk_tau_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SRFRD_E_DEVICE;
      opted[dev] = lds;
    }
  }
  hipLaunchKernelGGL(topk_tau_kernel, dim3((a.B + wpb - 1) / wpb), dim3(64 * wpb), lds, st, a);
  return 0;
}

// workspace layout: [cmax B*nc f32][tau B f32][cursor B+1 i32 (+pad)][candidates B*kCandMax][fallback B*nc*k]
static int64_t topk_off(int B, int k, int64_t nc, int which) {
  int64_t off = 0;
  const int64_t sizes[5] = {(int64_t)B * nc * 4, (int64_t)B * 4, ((int64_t)B + 4) * 4, (int64_t)B * kCandMax * (int64_t)sizeof(Cand),
                            (int64_t)B * nc * k * (int64_t)sizeof(Cand)};
  for (int i = 0; i < which; ++i) off += (sizes[i] + 255) & ~255ll;
  return off;
}

extern "C" int64_t srfrd_topk_workspace_bytes(int B, int k, int64_t n_rows) {
  if (B <= 0 || k <= 0 || n_rows <= 0) return 0;
  const int64_t n_chunks = (n_rows + kChunk - 1) / kChunk;
  return topk_off(B, k, n_chunks, 5);
}

extern "C" int srfrd_logits_topk(const srfrd_layout* lay, const void* item_table, const float* dense,
                                 const float* hidden, int B, int L, int64_t item_lo, int64_t item_hi, int exclude_pad,
                                 const int64_t* user_label, int k, int64_t* topk_idx, float* topk_val, void* workspace,
                                 void* stream) {
  if (!lay || !item_table || !dense || !hidden || !topk_idx || !topk_val || !workspace) return SRFRD_E_ARG;
  if (B <= 0 || L <= 0 || k <= 0 || k > 64 || item_lo < 0 || item_hi <= item_lo || item_hi > (int64_t)lay->n_items + 1) return SRFRD_E_ARG;
  if (lay->D > SRFRD_MAX_D) return SRFRD_E_UNSUPPORTED;
  if (lay->kind == SRFRD_SRFRN && !user_label) return SRFRD_E_ARG;
  const int n_chunks = (int)((item_hi - item_lo + kChunk - 1) / kChunk);
  const int DSi = ((lay->d_item + 3) & ~3) + 2;
  const size_t lds = ((size_t)kChunk * DSi + 16 * DSi + 16 * (kChunk + 2) + 16 + kSlack) * sizeof(float);
  if (lds > (size_t)kLdsLimit) return SRFRD_E_UNSUPPORTED;
  const size_t lds_stream = ((size_t)kChunk * DSi + 2 * 16 * DSi + 32 + 8 * 16 + kSlack) * sizeof(float);   // topk_stream
  static std::mutex attr_mu;
  static size_t attr_dev[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SRFRD_E_DEVICE;
  std::lock_guard<std::mutex> attr_lock(attr_mu);
  size_t& s_attr = attr_dev[dev];
  if (lds > s_attr) {
    if (hipFuncSetAttribute((const void*)topk_stage1_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess ||
        hipFuncSetAttribute((const void*)topk_max_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stream) != hipSuccess ||
        hipFuncSetAttribute((const void*)topk_collect_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_stream) != hipSuccess)
      return SRFRD_E_DEVICE;
    s_attr = lds;
  }
  char* ws = (char*)workspace;
  TopkArgs a = {};
  a.ly = *lay; a.table = item_table; a.dense = dense; a.hidden = hidden; a.user_label = user_label;
  a.B = B; a.L = L; a.exclude_pad = exclude_pad; a.k = k; a.n_chunks = n_chunks;
  a.item_lo = item_lo; a.item_hi = item_hi;
  a.cmax = (float*)(ws + topk_off(B, k, n_chunks, 0));
  a.tau = (float*)(ws + topk_off(B, k, n_chunks, 1));
  a.ccnt = (int32_t*)(ws + topk_off(B, k, n_chunks, 2));
  a.cand = (Cand*)(ws + topk_off(B, k, n_chunks, 3));
  Cand* fb = (Cand*)(ws + topk_off(B, k, n_chunks, 4));
  const int user_tiles = (B + 15) / 16;
  // user tiles are split over `splits` workgroups per chunk.  Each workgroup re-stages its chunk (~7 user tiles' worth of
  // time), and the launch runs in rounds of `slots` resident workgroups: pick the split with the smallest
  // rounds x (staging + tiles per workgroup).
  const int slots = 256 * (lds_stream * 2 <= (size_t)kLdsLimit ? 2 : 1);
  int splits = 1;
  double best = 1e30;
  for (int sp = 1; sp <= user_tiles && sp <= 64; ++sp) {
    const int64_t wgs = (int64_t)n_chunks * sp;
    const double rounds = (double)((wgs + slots - 1) / slots);
    const double cost = rounds * (7.0 + (double)((user_tiles + sp - 1) / sp));
    if (cost < best) { best = cost; splits = sp; }
  }
  a.user_splits = splits;
  hipStream_t st = (hipStream_t)stream;
  const bool bf16_tab = lay->table_bf16 != 0;
  const bool stream16 = getenv("SRFRD_TOPK_FP32") == nullptr &&
                        (bf16_tab ? (lay->d_item <= 64 && ((lay->d_item & 1) == 0 || lay->d_item <= 51)) : lay->d_item <= 52);
  if (stream16) {
    // the two threshold passes on the bf16 matrix cores (users in registers, item chunks streamed through LDS): a bf16
    // table as it is, an fp32 table split into three exact bf16 planes while it is staged.  The chunk-maxima array is walked
    // with this path's chunk count, everything else (tau, candidate lists, selection, the armed exhaustive path with its own
    // 256-item chunks) is shared
    static std::mutex mu16;
    static bool opted16[64] = {false};
    {
      std::lock_guard<std::mutex> lock(mu16);
      if (!opted16[dev]) {
        const void* fns[8] = {(const void*)topk_max16_kernel<1, false>, (const void*)topk_max16_kernel<2, false>,
                              (const void*)topk_collect16_kernel<1, false>, (const void*)topk_collect16_kernel<2, false>,
                              (const void*)topk_max16_kernel<1, true>, (const void*)topk_max16_kernel<2, true>,
                              (const void*)topk_collect16_kernel<1, true>, (const void*)topk_collect16_kernel<2, true>};
        for (const void* fn : fns)
          if (hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, kStream16Lds) != hipSuccess) return SRFRD_E_DEVICE;
        opted16[dev] = true;
      }
    }
    hipDeviceProp_t prop;
    static int cu_cached[64] = {0};
    if (cu_cached[dev] == 0) cu_cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
    const int cu = cu_cached[dev];
    const int64_t n_rows = item_hi - item_lo;
    int nu = user_tiles > kWaves16 ? 2 : 1;
    int groups = (user_tiles + kWaves16 * nu - 1) / (kWaves16 * nu);
    int crows = bf16_tab ? kChunk16 : 256;
    if ((lay->d_item & 1) != 0 || ((n_rows + crows - 1) / crows) * groups < 2 * (int64_t)cu) crows = 256;
    // a chunk is staged through kStageSlots16 registers per thread: rows x copy elements per row must fit them (a 512-row
    // chunk of a bf16 table does up to d_item 52; wider rows take 256-row chunks - rows beyond the slots would never be copied)
    const int64_t copy_w = bf16_tab ? ((lay->d_item & 1) == 0 ? lay->d_item >> 1 : lay->d_item) : lay->d_item;
    if ((int64_t)crows * copy_w > (int64_t)kStageSlots16 * kWaves16 * 64) crows = 256;
    if ((int64_t)crows * copy_w > (int64_t)kStageSlots16 * kWaves16 * 64) return SRFRD_E_UNSUPPORTED;
    int64_t nch = (n_rows + crows - 1) / crows;
    if (nch * groups < cu && nu == 2) { nu = 1; groups = (user_tiles + kWaves16 - 1) / kWaves16; }
    int per_group = cu / groups < 1 ? 1 : cu / groups;
    if (per_group > nch) per_group = (int)nch;
    TopkArgs h = a;
    h.n_chunks = (int)nch;
    h.crows = crows;
    h.wg_per_group = per_group;
    const dim3 grid16(groups * per_group), blk16(kWaves16 * 64);
#define SRFRD_L16(KERNEL) do { \
      if (bf16_tab) { if (nu == 2) hipLaunchKernelGGL((KERNEL<2, false>), grid16, blk16, kStream16Lds, st, h); \
                      else hipLaunchKernelGGL((KERNEL<1, false>), grid16, blk16, kStream16Lds, st, h); } \
      else { if (nu == 2) hipLaunchKernelGGL((KERNEL<2, true>), grid16, blk16, kStream16Lds, st, h); \
             else hipLaunchKernelGGL((KERNEL<1, true>), grid16, blk16, kStream16Lds, st, h); } } while (0)
    SRFRD_L16(topk_max16_kernel);
    if (int trc = launch_tau(h, st)) return trc;
    SRFRD_L16(topk_collect16_kernel);
#undef SRFRD_L16
  } else {
    hipLaunchKernelGGL(topk_max_kernel, dim3(n_chunks * splits), dim3(512), lds_stream, st, a);
    if (int trc = launch_tau(a, st)) return trc;
    hipLaunchKernelGGL(topk_collect_kernel, dim3(n_chunks * splits), dim3(512), lds_stream, st, a);
  }
  hipLaunchKernelGGL(topk_select_kernel, dim3((B + 3) / 4), dim3(256), 0, st, a, topk_idx, topk_val);
  // exhaustive path, armed only if a candidate list overflowed (device-side flag: no host synchronisation)
  hipLaunchKernelGGL(topk_stage1_kernel, dim3(n_chunks), dim3(256), lds, st, *lay, item_table, dense, hidden, B, L, item_lo,
                     item_hi, exclude_pad, user_label, k, n_chunks, fb, (const int32_t*)(a.ccnt + B));
  hipLaunchKernelGGL(topk_stage2_kernel, dim3((B + 3) / 4), dim3(256), 0, st, (const Cand*)fb, B, k, n_chunks, topk_idx,
                     topk_val, (const int32_t*)(a.ccnt + B));
  return (int)hipGetLastError();
}

extern "C" int srfrd_topk_merge(const int64_t* cand_idx, const float* cand_val, int B, int n_cand, int k, int64_t* topk_idx,
                                float* topk_val, void* stream) {
  if (!cand_idx || !cand_val || !topk_idx || !topk_val || B <= 0 || n_cand <= 0 || k <= 0 || n_cand > 4096) return SRFRD_E_ARG;
  const size_t lds = (size_t)4 * 2 * n_cand * sizeof(float);       // 4 waves per block, (value, id) per candidate: <= 128 KiB
  static std::mutex mu;
  static size_t opted[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) return SRFRD_E_DEVICE;
  {
    std::lock_guard<std::mutex> lock(mu);
    if (lds > 48 * 1024 && lds > opted[dev]) {
      if (hipFuncSetAttribute((const void*)topk_merge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SRFRD_E_DEVICE;
      opted[dev] = lds;
    }
  }
  hipLaunchKernelGGL(topk_merge_kernel, dim3((B + 3) / 4), dim3(256), lds, (hipStream_t)stream, cand_idx, cand_val, B, n_cand, k,
                     topk_idx, topk_val);
  return (int)hipGetLastError();
}

extern "C" int srfrd_eval_rank(const float* logits, int B, int n_cand, int32_t* rank, double* metric_acc, void* stream) {
  if (!logits || B <= 0 || n_cand <= 0) return SRFRD_E_ARG;
  hipLaunchKernelGGL(eval_rank_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, logits, B, n_cand, rank, metric_acc);
  return (int)hipGetLastError();
}

// ---------------------------------------------------------------------------------------------
// device-side batch sampler: the layout and semantics of reference utils.py:21-57 (sample_function_fr) with a
// counter-based RNG, one wave per sampled user.  Histories arrive as CSR over user ids 0..usernum.
// ---------------------------------------------------------------------------------------------
namespace srfrd {

__device__ __forceinline__ uint32_t samp_rnd(uint32_t seed, uint32_t batch, uint32_t b, uint32_t t, uint32_t k) {
  return fmix32(fmix32(fmix32(seed ^ (batch * 0x9E3779B9u)) + b) ^ (t * 4096u + k));
}
__device__ __forceinline__ int samp_range(uint32_t r, int n) {      // uniform integer in [0, n)
  return (int)(((unsigned long long)r * (unsigned long long)n) >> 32);
}

__global__ void __launch_bounds__(256) sample_batch_kernel(const int64_t* __restrict__ ptr, const int32_t* __restrict__ items,
                                                          const int32_t* __restrict__ reviews, int usernum, int itemnum,
                                                          int B, int L, uint32_t seed, uint32_t batch,
                                                          int64_t* __restrict__ out_user, int64_t* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
  if (b >= B) return;
  // user with more than one interaction (utils.py:24-25): rejection, every lane draws the same stream
  int u = 1, n = 0;
  int64_t p0 = 0;
  for (int k = 0; k < 4096; ++k) {
    u = 1 + samp_range(samp_rnd(seed, batch, (uint32_t)b, 0xFFFFFu, (uint32_t)k), usernum);
    p0 = ptr[u];
    n = (int)(ptr[u + 1] - p0);
    if (n > 1) break;
  }
  if (lane == 0) out_user[b] = u;
  const int32_t* it = items + p0;
  const int32_t* rv = reviews + p0;
  const int64_t BL = (int64_t)B * L;
  int64_t* seq = out + 0 * BL + (int64_t)b * L;
  int64_t* rsq = out + 1 * BL + (int64_t)b * L;
  int64_t* pos = out + 2 * BL + (int64_t)b * L;
  int64_t* prs = out + 3 * BL + (int64_t)b * L;
  int64_t* neg = out + 4 * BL + (int64_t)b * L;
  int64_t* nrs = out + 5 * BL + (int64_t)b * L;
  const int m = n > 1 ? min(n - 1, L) : 0;            // filled positions (most recent last)
  for (int j = lane; j < L; j += 64) {
    const int idx = L - 1 - j;
    if (j < m) {
      seq[idx] = it[n - 2 - j];
      pos[idx] = it[n - 1 - j];
      rsq[idx] = rv[n - 2 - j];
      prs[idx] = rv[n - 1 - j];
      nrs[idx] = 1;                                     // np.random.randint(1, 2) == 1 (utils.py:52)
      int cand = 1;
      for (int k = 0; k < 256; ++k) {                   // random_neq (utils.py:14-19): not among the user's items
        cand = 1 + samp_range(samp_rnd(seed, batch, (uint32_t)b, (uint32_t)idx, (uint32_t)k), itemnum);
        bool clash = false;
        for (int q = 0; q < n; ++q) clash |= (it[q] == cand);
        if (!clash) break;
      }
      neg[idx] = cand;
    } else {
      seq[idx] = 0; pos[idx] = 0; rsq[idx] = 0; prs[idx] = 0; neg[idx] = 0; nrs[idx] = 0;
    }
  }
}

}  // namespace srfrd

extern "C" int srfrd_sample_batch(const int64_t* user_ptr, const int32_t* items, const int32_t* reviews, int usernum,
                                  int itemnum, int B, int L, uint32_t seed, uint32_t batch_index, int64_t* out_user,
                                  int64_t* out_packed, void* stream) {
  if (!user_ptr || !items || !reviews || !out_user || !out_packed || usernum < 1 || itemnum < 1 || B <= 0 || L <= 0)
    return SRFRD_E_ARG;
  hipLaunchKernelGGL(srfrd::sample_batch_kernel, dim3((B + 3) / 4), dim3(256), 0, (hipStream_t)stream, user_ptr, items,
                     reviews, usernum, itemnum, B, L, seed, batch_index, out_user, out_packed);
  return (int)hipGetLastError();
}
