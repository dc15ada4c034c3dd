// Ragged fused encoder forward at the reference's default geometry (srfrd_encoder_fwd_ragged_kernel.inc): the forward of
// reference SRFR_model.py:92-142 / :192-239 / :473-530 / :621-666 computed on the rows a left-padded sequence really has.
// srfrd_encoder_fwd (srfrd_encoder_fwd.hip) dispatches here when shape and mode qualify.  Also home of srfrd_seq_order, the
// per-batch length order both ragged kernels schedule their sequences by.
#include "srfrd_enc_common.h"

#include "srfrd_encoder_fwd_ragged_kernel.inc"

#include <cstring>

namespace srfrd {

// ---- srfrd_seq_order: first non-pad position of every sequence, the sequences ranked longest first, counters zeroed ----
// One wave per sequence reads its ids (coalesced) and ballots the first non-zero; the LAST workgroup to finish (arrival
// ticket behind an agent-scope release) ranks the B lengths - rank = #{longer} + #{equally long with a smaller index}: a
// stable order, so the schedule (and with it the summation order of the dense-gradient slabs) is a function of the batch.
__global__ void __launch_bounds__(1024) seq_order_kernel(const int64_t* __restrict__ ids, int B, int L, int G, int* sched) {
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  int* len = sched + kSchedLen;
  for (int b = blockIdx.x * 16 + wave; b < B; b += gridDim.x * 16) {
    int t0 = L;
    for (int base = 0; base < L && t0 == L; base += 64) {
      const int t = base + lane;
      const bool nz = t < L && ids[(int64_t)b * L + t] != 0;
      const unsigned long long m = __ballot(nz);
      if (m) t0 = base + (int)__builtin_ctzll(m);
    }
    if (lane == 0) __hip_atomic_store(&len[b], t0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  }
  __shared__ int s_last;
  __shared__ int s_len[4096];
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const int k = atomicAdd(&sched[kSchedTicket], 1);
    s_last = k == (int)gridDim.x - 1;
    if (s_last) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (!s_last) return;
  int* sorted = sched + kSchedLen + B;
  for (int c0 = 0; c0 < B; c0 += 4096) {       // (ranks against chunks of 4096 lengths held in LDS)
    const int n = min(4096, B - c0);
    __syncthreads();
    for (int i = tid; i < n; i += 1024) s_len[i] = __hip_atomic_load(&len[c0 + i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    if (c0 == 0 && B <= 4096) {
      for (int b = tid; b < B; b += 1024) {
        const int mine = s_len[b];
        int rank = 0;
        for (int j = 0; j < B; ++j) {
          const int o = s_len[j];
          rank += (o < mine || (o == mine && j < b)) ? 1 : 0;
        }
        sorted[rank] = b;
      }
    }
  }
  if (B > 4096)                                  // very large batches: identity order (the schedule only balances, never decides results)
    for (int b = tid; b < B; b += 1024) sorted[b] = b;
  for (int i = tid; i < 32; i += 1024) sched[kSchedCnt + i] = 0;
  for (int i = tid; i < 2 * 2048; i += 1024) sched[kSchedArr + i] = 0;
  if (tid == 0) {
    sched[kSchedG] = G;
    sched[kSchedTicket] = 0;
  }
}

}  // namespace srfrd

using namespace srfrd;

extern "C" int64_t srfrd_sched_ints(int B) { return B > 0 ? sched_ints(B) : 0; }

extern "C" int srfrd_seq_order(const int64_t* input_ids, int B, int L, int pair_stride, int32_t* sched, void* stream) {
  if (!input_ids || !sched || B <= 0 || L <= 0 || pair_stride <= 0) return SRFRD_E_ARG;
  int grid = (B + 15) / 16;
  if (grid > 64) grid = 64;
  hipLaunchKernelGGL(seq_order_kernel, dim3(grid), dim3(1024), 0, (hipStream_t)stream, input_ids, B, L, pair_stride, sched);
  return (int)hipGetLastError();
}

// kind_variant: 0 SASRec (50 + 0), 1 SRFR (45 + 5), 2 SRFRN (45 + 5), 3 SRFU_* (50 + 0, kind read at run time); train: the
// fused-training instantiation (targets, checkpoints, loss sums, dropout all on).  Geometry is fixed: hidden 50, seq_len 50.
extern "C" int srfrd_fwd_ragged_launch(const void* args, int kind_variant, int train, int grid, void* stream) {
  EncArgs a;
  std::memcpy(&a, args, sizeof(a));
  if (a.dm.D != 50 || a.L != 50 || a.dm.n_heads != 1 || a.dm.n_blocks > SRFRD_MAX_BLOCKS || a.dbg) return SRFRD_E_UNSUPPORTED;
  const Geom g = make_geom(50, 50);
  const int64_t lds = fwd_lds_floats(g, a.dm.n_blocks) * 4;
  if (lds > kLdsLimit) return SRFRD_E_UNSUPPORTED;
#define SRFRD_RG(K, DI) (train ? launch_enc(encoder_fwd_ragged_kernel<K, 1, DI>, grid, 512, lds, stream, a) \
                               : launch_enc(encoder_fwd_ragged_kernel<K, 0, DI>, grid, 512, lds, stream, a))
  switch (kind_variant) {
    case 0: return SRFRD_RG(SRFRD_SASREC, 50);
    case 1: return SRFRD_RG(SRFRD_SRFR, 45);
    case 2: return SRFRD_RG(SRFRD_SRFRN, 45);
    case 3: return SRFRD_RG(-1, 50);
  }
#undef SRFRD_RG
  return SRFRD_E_UNSUPPORTED;
}
