// Shared pieces of the fused encoder translation units (forward / backward): launch arguments, the LayerNorm
// parameter cache, debug taps, host-side argument marshalling.
#pragma once
#include <cstdlib>
#include <map>
#include <mutex>
#include <type_traits>
#include <utility>

#include "srfrd_dev.h"

namespace SRFRD_NS {

struct EncArgs {
  Dims dm;
  const float* table;       // fp32 item table, or NULL when the gathers read the bf16 shadow
  const uint16_t* table16;  // bf16 shadow of the item table (srfrd_layout::table_bf16), or NULL
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
  float *hidden, *pos_logits, *neg_logits, *save_x, *save_h1, *save_aux, *loss_part;
  // backward inputs / outputs
  const float *c_hidden, *c_pl, *c_nl, *c_save_x, *c_save_h1, *c_save_aux, *d_hidden, *d_pos, *d_neg;
  int fused_bce;
  int last_only;       // forward, inference: only the LAST position's hidden state is wanted (hidden is (B, d_out))
  float *grad_table, *grad_slabs;
  float* contrib;      // deterministic table scatter: (3, B, L, d_item) row contributions instead of float atomics (or NULL)
  // SRFRD_BUF_GLOBAL build only: per-workgroup working-set scratch (floats) in global memory
  float* scratch;
  int64_t scratch_stride;
  int lds_floats;      // dynamic LDS (floats) the long build may carve from before falling back to `scratch`
  int carve_mode;      // which buffers get the LDS share first (0: score matrices, 1: activation matrices)
  // debug taps
  float* dbg;
  int dbg_seq;
  int64_t dbg_slot;
  // ragged kernels (seq_len 50): sequence -> workgroup schedule from srfrd_seq_order's per-sequence lengths (NULL: b = blockIdx.x, += gridDim.x)
  const int* sched;    // int32 workspace, layout below (kSched*)
  int sched_mode;      // 0 none, 1 length order (workgroup x takes the sequence of rank perm(x): see rag_take)
  int long_prio;       // ragged kernels: wave priority of sequences with three or four row tiles (0: none; experiment)
  int ragged_off;      // diagnostic: the ragged kernels compute every row (t0 = 0), as the full kernels do
};

// Schedule workspace (int32), filled by srfrd_seq_order for ONE batch:
//   [kSchedG]        pair stride G (workgroups of the "first round": the CU count)
//   [kSchedT0 + b]   first non-pad position t0 of sequence b (L: all padding)
// The ragged kernels rank the sequences themselves (longest first = smallest t0 first, ties by index - a function of the
// batch, so the summation order of the dense-gradient slabs is too): every workgroup selects the ONE sequence of the rank
// it wants from the B lengths (a 64-bucket histogram in LDS + a ballot scan: rag_select) instead of a sort kernel ahead of
// the launch - a one-workgroup sort costs more dependent memory round trips than it saves.
constexpr int kSchedG = 0, kSchedT0 = 16;
__host__ __device__ __forceinline__ int64_t sched_ints(int B) { return kSchedT0 + (int64_t)B; }

// Checkpoint layout: SEQUENCE-major.  Everything the backward reads back for sequence b is contiguous per buffer -
// save_x: (nb + 1) blocks of [L][D] at b * (nb + 1) * L * D; save_h1: nb blocks at b * nb * L * D; save_aux
// (srfrd_aux_floats): nb blocks of aux_seq_floats at b * nb * aux_seq_floats, each the planes r, o, q, k, v [L][D] and
// Pm [H][L][LP] (one probability block per attention head).  (A block-major layout put the ten planes a workgroup touches per block 10 MB apart at BASELINE
// configs[3]: every first access of a phase was a TLB miss on the sequence's critical path.)
__host__ __device__ __forceinline__ int64_t aux_seq_floats(int L, int LP, int D, int H = 1) { return 5ll * L * D + (int64_t)H * L * LP; }
__host__ __device__ __forceinline__ int64_t x_off(int i, int b, int nb, int L, int D) { return ((int64_t)b * (nb + 1) + i) * L * D; }
__host__ __device__ __forceinline__ int64_t h1_off(int i, int b, int nb, int L, int D) { return ((int64_t)b * nb + i) * L * D; }
struct AuxOff {
  int64_t r, o, q, k, v, p;
};
__host__ __device__ __forceinline__ AuxOff aux_off(int i, int b, int nb, int L, int LP, int D, int H = 1) {
  const int64_t blk = ((int64_t)b * nb + i) * aux_seq_floats(L, LP, D, H), plane = (int64_t)L * D;
  AuxOff f;
  f.r = blk;
  f.o = blk + plane;
  f.q = blk + 2 * plane;
  f.k = blk + 3 * plane;
  f.v = blk + 4 * plane;
  f.p = blk + 5 * plane;
  return f;
}

// Optimisation barrier on a wave-uniform pointer: stops LLVM from hoisting the per-call-site address arithmetic of
// ~35 inlined GEMMs out of the sequence / block loops (which costs > 256 VGPRs and spills).
__device__ __forceinline__ void launder(lds_f*& p) {
  asm volatile("" : "+s"(p));
}
#ifndef SRFRD_BUF_GLOBAL
__device__ __forceinline__ void launder(const float*& p) {
  asm volatile("" : "+s"(p));
}
__device__ __forceinline__ void launder(float*& p) {
  asm volatile("" : "+s"(p));
}
#endif

// Diagnostic build only (tools/phase_profile.py compiles a copy with -DSRFRD_STAMPS and a STAMP(n) after every
// workgroup barrier): thread 0 adds the s_memtime delta of each phase into a per-workgroup table that aliases the
// debug-tap buffer.  The shipped library contains no stamp.
#ifdef SRFRD_STAMPS
// (accumulated in LDS and flushed once: a global read-modify-write per stamp would put a memory round trip of its own
// into every phase it measures)
#define STAMP_INIT __shared__ unsigned long long stamp_lds[128]; \
                   if (threadIdx.x < 128) stamp_lds[threadIdx.x] = 0; \
                   __syncthreads(); \
                   unsigned long long stamp_prev = __builtin_amdgcn_s_memtime();
#define STAMP(id) do { if (threadIdx.x == 0) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); \
                       stamp_lds[id] += t_ - stamp_prev; stamp_prev = t_; } } while (0)
#define STAMP_FLUSH do { __syncthreads(); if (threadIdx.x < 128 && a.dbg) \
                       ((unsigned long long*)a.dbg)[(int64_t)blockIdx.x * 128 + threadIdx.x] = stamp_lds[threadIdx.x]; } while (0)
#else
#define STAMP_INIT
#define STAMP(id) do {} while (0)
#define STAMP_FLUSH do {} while (0)
#endif

// A contiguous [rows][cols] fp32 block of global memory on its way into an LDS matrix [rows][ld].  load() puts every
// element the thread owns in flight at once: indices past the end are clamped instead of predicated, so the loads are
// unconditional, stay back to back, and the block costs ONE memory round trip (a `for (i = tid; ...) lds[..] = g[i]`
// loop that the compiler does not unroll pays one round trip - 1.5 to 2 k cycles from L2 / HBM here - per iteration).
// store() writes them to LDS; a caller may put independent work between the two.  V2: 8-byte accesses (cols even and
// the block 8-byte aligned).  One G2L covers ITER * blockDim elements (pairs); copy_g2l walks larger blocks in chunks.
typedef float f32x2 __attribute__((ext_vector_type(2)));
#ifdef SRFRD_BUF_GLOBAL
typedef f32x2 lds_f2;
#else
typedef __attribute__((address_space(3))) f32x2 lds_f2;
#endif
template <int ITER, bool V2>
struct G2L {
  float v[ITER][V2 ? 2 : 1];
  __device__ __forceinline__ void load(const float* src, int rows, int cols, int nthr, int base = 0) {
    const int n = V2 ? rows * (cols >> 1) : rows * cols;
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
      const int i = min(base + u * nthr + (int)threadIdx.x, n - 1);
      if constexpr (V2) {
        const f32x2 t = reinterpret_cast<const f32x2*>(src)[(unsigned)i];
        v[u][0] = t.x; v[u][1] = t.y;
      } else {
        v[u][0] = src[(unsigned)i];
      }
    }
  }
  // [rows][cols] window of a row-major global matrix whose rows are `sld` floats apart (V2: cols, sld even)
  __device__ __forceinline__ void load2d(const float* src, int rows, int cols, int sld, int nthr) {
    const int cw = V2 ? cols >> 1 : cols, n = rows * cw;
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
      const int i = min(u * nthr + (int)threadIdx.x, n - 1);
      const int t = i / cw, c = i - t * cw;
      if constexpr (V2) {
        const f32x2 w = *reinterpret_cast<const f32x2*>(src + (unsigned)(t * sld + 2 * c));
        v[u][0] = w.x; v[u][1] = w.y;
      } else {
        v[u][0] = src[(unsigned)(t * sld + c)];
      }
    }
  }
  __device__ __forceinline__ void store(lds_f* dst, int ld, int rows, int cols, int nthr, int base = 0) const {
    const int cw = V2 ? cols >> 1 : cols, n = rows * cw;
#pragma unroll
    for (int u = 0; u < ITER; ++u) {
      const int i = base + u * nthr + (int)threadIdx.x;
      if (i < n) {
        const int t = i / cw, c = i - t * cw;
        if constexpr (V2) *reinterpret_cast<lds_f2*>(dst + t * ld + 2 * c) = f32x2{v[u][0], v[u][1]};
        else dst[t * ld + c] = v[u][0];
      }
    }
  }
};
template <int ITER, bool V2>
__device__ __forceinline__ void copy_g2l(lds_f* dst, int ld, const float* src, int rows, int cols, int nthr) {
  const int n = V2 ? rows * (cols >> 1) : rows * cols;
  for (int base = 0; base < n; base += ITER * nthr) {
    G2L<ITER, V2> g;
    g.load(src, rows, cols, nthr, base);
    g.store(dst, ld, rows, cols, nthr, base);
  }
}

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

// ================================================================================================
// host side
// ================================================================================================
// CU count of the CURRENT device (cached per device: one process may drive several GPUs)
[[maybe_unused]] static int num_cu() {
  static std::mutex mu;
  static int cached[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(mu);
  if (cached[dev] == 0) {
    hipDeviceProp_t prop;
    cached[dev] = (hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0) ? prop.multiProcessorCount : 256;
  }
  return cached[dev];
}

// block size override for tuning runs (multiple of 64, <= 1024)
[[maybe_unused]] static int env_threads(const char* name, int dflt) {
  const char* e = getenv(name);
  if (!e) return dflt;
  const int v = atoi(e);
  return (v >= 64 && v <= 1024 && (v & 63) == 0) ? v : dflt;
}

[[maybe_unused]] static int fill_args(EncArgs& a, const srfrd_layout* lay, const void* item_table, const float* dense, const float* packed,
                     const int64_t* input_ids, const int64_t* fake_ids, const int64_t* pos_ids, const int64_t* pos_fake,
                     const int64_t* neg_ids, const int64_t* neg_fake, int B, int L, double dropout_p, uint32_t seed,
                     const uint32_t* seed_dev, int64_t seq_index0) {
  if (!lay || !item_table || !dense || !packed || !input_ids || B <= 0 || L <= 0) return SRFRD_E_ARG;
  if (lay->D > SRFRD_MAX_D || lay->n_heads < 1 || lay->D % lay->n_heads != 0 || lay->n_blocks > SRFRD_MAX_BLOCKS) return SRFRD_E_UNSUPPORTED;
  if (L > lay->max_len) return SRFRD_E_ARG;
  if (dropout_p < 0.0 || dropout_p >= 1.0) return SRFRD_E_ARG;
  if (lay->kind == SRFRD_SRFRN && ((pos_ids && !pos_fake) || (neg_ids && !neg_fake))) return SRFRD_E_ARG;
  Dims& d = a.dm;
  d.kind = lay->kind; d.d_item = lay->d_item; d.d_fake = lay->d_fake; d.D = lay->D; d.d_out = lay->d_out;
  d.n_labels = lay->n_labels; d.n_blocks = lay->n_blocks; d.n_items = lay->n_items; d.n_heads = lay->n_heads;
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
  a.table = lay->table_bf16 ? nullptr : (const float*)item_table;
  a.table16 = lay->table_bf16 ? (const uint16_t*)item_table : nullptr;
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

// The ragged seq_len-50 pair (srfrd_encoder_fwd_ragged_kernel.inc + srfrd_encoder_bwd_ragged_kernel.inc) exchanges
// checkpoints that hold only the rows of the computed tiles: a training forward may take the ragged kernel only when the
// backward of the same (layout, length) will be the ragged one, and vice versa - ONE predicate, asked by both launchers.
[[maybe_unused]] static bool ragged_pair(const srfrd_layout* lay, int L) {
  if (lay->D != 50 || lay->n_heads != 1 || L != 50 || lay->n_blocks > SRFRD_MAX_BLOCKS) return false;
  if (getenv("SRFRD_NO_RAGGED") || getenv("SRFRD_GENERIC") || getenv("SRFRD_NO_LSPEC") || getenv("SRFRD_NO_KSPEC") ||
      getenv("SRFRD_NO_SLOTS50") || getenv("SRFRD_ROWS_ALWAYS") || getenv("SRFRD_FWD_THREADS") || getenv("SRFRD_BWD_THREADS"))
    return false;
  if (lay->kind == SRFRD_SASREC) return true;
  if ((lay->kind == SRFRD_SRFR || lay->kind == SRFRD_SRFRN) && lay->d_item == 45) return true;
  return lay->kind >= SRFRD_SRFU_B && lay->d_item == 50;
}
// kind_variant of the ragged kernels: 0 SASRec 50 + 0, 1 SRFR 45 + 5, 2 SRFRN 45 + 5, 3 SRFU_* 50 + 0 (kind read at run time)
[[maybe_unused]] static int ragged_variant(const srfrd_layout* lay) {
  if (lay->kind == SRFRD_SASREC) return 0;
  if (lay->kind == SRFRD_SRFR) return 1;
  if (lay->kind == SRFRD_SRFRN) return 2;
  return 3;
}

// launch one instantiation.  The > 64 KiB dynamic-LDS opt-in (hipFuncSetAttribute) applies to one function on the CURRENT
// device: it is remembered per (device, function) under a mutex, so a second GPU driven from the same process, or two
// host threads launching concurrently, each get it set before their first launch.
template <class K>
static int launch_enc(K kernel, int grid, int threads, int64_t lds, void* stream, const EncArgs& a) {
  static std::mutex mu;
  static std::map<std::pair<int, const void*>, int64_t> opted;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return SRFRD_E_DEVICE;
  {
    std::lock_guard<std::mutex> lock(mu);
    int64_t& have = opted[{dev, (const void*)kernel}];
    if (lds > have) {
      if (hipFuncSetAttribute((const void*)kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess)
        return SRFRD_E_DEVICE;
      have = lds;
    }
  }
  hipLaunchKernelGGL(kernel, dim3(grid), dim3(threads), (size_t)lds, (hipStream_t)stream, a);
  return (int)hipGetLastError();
}

}  // namespace SRFRD_NS
