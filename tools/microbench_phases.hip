// Diagnostic micro-benchmark (not part of the product): cycles per barrier-delimited phase of the building blocks
// in srfrd_dev.h, in isolation, one workgroup per CU.   hipcc --offload-arch=gfx950 -O3 -std=c++17 -o mb tools/microbench_phases.hip
#include <cstdio>
#include <vector>
#include "../srfrd_amd/csrc/srfrd_dev.h"
using namespace srfrd;

constexpr int LP = 64, D = 50, DS = 54, SLD = 66, NT = 4, MT = 4, DK = 52, L = 50;

template <int V, int NW>
__global__ void __launch_bounds__(NW * 64) mb(const float* packed, const float* bias, float* slab, unsigned long long* out, int iters) {
  extern __shared__ __attribute__((aligned(16))) float smem[];
  lds_f* A = (lds_f*)smem;
  lds_f* Bm = A + LP * DS;
  lds_f* Cm = Bm + LP * DS;
  lds_f* S = Cm + LP * DS;
  lds_f* lnw = S + LP * SLD;
  for (int i = threadIdx.x; i < 3 * LP * DS + LP * SLD + 128; i += blockDim.x) A[i] = 0.001f * (float)(i % 97);
  __syncthreads();
  PackedB pb{reinterpret_cast<const float4*>(packed)};
  WFrag w = load_wfrag(pb, bias, D, NT);
  DropSite ds = drop_site(1, 123u, 1, blockIdx.x, 0x80000000u, 2.0f);
  unsigned long long t0 = __builtin_amdgcn_s_memtime();
  for (int it = 0; it < iters; ++it) {
    if (V == 0) gemm_packed(NW, MT, NT, DK, Mat{A, DS}, w, [&](int r, int c, float v) { if (c < D) Cm[r * DS + c] = v; });
    if (V == 1) { WFrag w2 = load_wfrag(pb, bias, D, NT); gemm_packed(NW, MT, NT, DK, Mat{A, DS}, w2, [&](int r, int c, float v) { if (c < D) Cm[r * DS + c] = v; }); }
    if (V == 2) gemm_tiles<1>(NW, MT, MT, DK, Mat{A, DS}, MatT{Bm, DS}, [&](int r, int c, float v) { S[r * SLD + c] = v; });
    if (V == 3) gemm_slab(NW, NT, NT, LP, MatT{A, DS}, Mat{Bm, DS}, SlabWB{slab + blockIdx.x * 4096, 0, 3000, D, D, 1});
    if (V == 4) ln_rows(NW, A, Cm, L, DS, D, lnw, lnw + 64);
    if (V == 5) softmax_rows<true>(NW, S, L, SLD, LP, ds);
    if (V == 6) { }
    if (V == 7) gemm_tiles<2>(NW, MT, NT, LP, Mat{S, SLD}, Mat{Bm, DS}, [&](int r, int c, float v) { if (c < D) Cm[r * DS + c] = v; });
    if (V == 8) gemm_tiles<0>(NW, 1, NT, LP, OnesRow{}, Mat{A, DS}, [=](int r, int c, float v) { if (r == 0 && c < D) slab[blockIdx.x * 4096 + c] += v; });
    if (V == 10) { if ((threadIdx.x >> 6) >= NW / 2) __builtin_amdgcn_s_sleep(8); gemm_packed(NW, MT, NT, DK, Mat{A, DS}, w, [&](int r, int c, float v) { if (c < D) Cm[r * DS + c] = v; }); }
    if (V == 11) { if ((threadIdx.x >> 6) >= NW / 2) __builtin_amdgcn_s_sleep(16); gemm_packed(NW, MT, NT, DK, Mat{A, DS}, w, [&](int r, int c, float v) { if (c < D) Cm[r * DS + c] = v; }); }
    if (V == 12) { for (int rep = 0; rep < 3; ++rep) gemm_packed(NW, MT, NT, DK, Mat{A, DS}, w, [&](int r, int c, float v) { if (c < D) Cm[r * DS + c] = v + (float)rep; }); }
    if (V == 9) gemm_packed(NW, MT, NT, DK, Mat{A, DS}, w, [&](int r, int c, float v) { if (c < D) Cm[r * DS + c] = fmaxf(v * drop_mul(ds, r, c), 0.f); });
    __syncthreads();
  }
  unsigned long long t1 = __builtin_amdgcn_s_memtime();
  if (threadIdx.x == 0) out[blockIdx.x] = (t1 - t0);
}

template <int V, int NW>
void run1(const char* name, const float* packed, const float* bias, float* slab, unsigned long long* out);
template <int V>
void run(const char* name, int threads, const float* packed, const float* bias, float* slab, unsigned long long* out) {
  if (threads == 256) run1<V, 4>(name, packed, bias, slab, out);
  else if (threads == 512) run1<V, 8>(name, packed, bias, slab, out);
  else run1<V, 16>(name, packed, bias, slab, out);
}
template <int V, int NW>
void run1(const char* name, const float* packed, const float* bias, float* slab, unsigned long long* out) {
  const int threads = NW * 64;
  const int iters = 200, grid = 256;
  size_t lds = (3 * LP * DS + LP * SLD + 256) * 4;
  hipFuncSetAttribute((const void*)mb<V, NW>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  hipLaunchKernelGGL((mb<V, NW>), dim3(grid), dim3(threads), lds, 0, packed, bias, slab, out, iters);
  hipDeviceSynchronize();
  hipLaunchKernelGGL((mb<V, NW>), dim3(grid), dim3(threads), lds, 0, packed, bias, slab, out, iters);
  hipDeviceSynchronize();
  std::vector<unsigned long long> h(grid);
  hipMemcpy(h.data(), out, grid * 8, hipMemcpyDeviceToHost);
  double s = 0;
  for (auto v : h) s += (double)v;
  printf("%-44s threads=%d  %8.0f cycles/phase\n", name, threads, s / grid / iters);
}

int main() {
  float *packed, *bias, *slab;
  unsigned long long* out;
  hipMalloc(&packed, 4096 * 4 * 4);
  hipMalloc(&bias, 256);
  hipMalloc(&slab, 256 * 4096 * 4);
  hipMalloc(&out, 256 * 8);
  hipMemset(packed, 0, 4096 * 4 * 4);
  hipMemset(bias, 0, 256);
  hipMemset(slab, 0, 256 * 4096 * 4);
  for (int threads : {256, 512, 1024}) {
    run<6>("empty phase (barrier only)", threads, packed, bias, slab, out);
    run<0>("gemm_packed, fragments preloaded", threads, packed, bias, slab, out);
    run<1>("gemm_packed + load_wfrag each phase", threads, packed, bias, slab, out);
    run<10>("gemm_packed, waves >= nw/2 sleep(8)", threads, packed, bias, slab, out);
    run<11>("gemm_packed, waves >= nw/2 sleep(16)", threads, packed, bias, slab, out);
    run<12>("3 x gemm_packed in one phase", threads, packed, bias, slab, out);
    run<9>("gemm_packed + relu/dropout epilogue", threads, packed, bias, slab, out);
    run<2>("gemm_tiles<1> S = Q K^T", threads, packed, bias, slab, out);
    run<7>("gemm_tiles<2> O = P V", threads, packed, bias, slab, out);
    run<3>("gemm_slab dW (+db) rmw", threads, packed, bias, slab, out);
    run<8>("ones-row GEMM column sums -> slab", threads, packed, bias, slab, out);
    run<4>("ln_rows", threads, packed, bias, slab, out);
    run<5>("softmax_rows<masked>", threads, packed, bias, slab, out);
  }
  return 0;
}
