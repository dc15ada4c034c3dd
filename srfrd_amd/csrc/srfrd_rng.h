// Counter-based dropout hash shared by every kernel (and restated integer-exactly by the test oracle).
// keep(seed, site, seq, row, col) = fmix32( fmix32( fmix32(seed + site*GOLDEN) ^ seq ) ^ (row*4096 + col) ) >= threshold
// The reference draws torch Bernoulli masks (nn.Dropout / MHA dropout, reference SRFR_model.py:42,45,83,625);
// a device kernel cannot reproduce that stream, so masks are defined by coordinates instead: backward regenerates
// exactly what forward used, and results do not depend on how sequences are split over workgroups or ranks.
#pragma once
#include <stdint.h>

#if defined(__HIPCC__)
#define SRFRD_HD __host__ __device__ __forceinline__
#else
#define SRFRD_HD inline
#endif

namespace srfrd {

SRFRD_HD uint32_t fmix32(uint32_t h) {
  h ^= h >> 16; h *= 0x85EBCA6Bu; h ^= h >> 13; h *= 0xC2B2AE35u; h ^= h >> 16;
  return h;
}

enum { SITE_EMB = 0 };
SRFRD_HD int site_attn(int blk, int head = 0) { return 1 + 3 * blk + 1024 * head; }   // (one mask per attention head)
SRFRD_HD int site_ffn1(int blk) { return 2 + 3 * blk; }
SRFRD_HD int site_ffn2(int blk) { return 3 + 3 * blk; }

struct DropSite {
  uint32_t h2, thr;
  float scale;
  int on;
};

SRFRD_HD DropSite drop_site(int on, uint32_t seed, int site, uint32_t seq, uint32_t thr, float scale) {
  DropSite d;
  d.on = on; d.thr = thr; d.scale = scale;
  d.h2 = fmix32(fmix32(seed + (uint32_t)site * 0x9E3779B9u) ^ seq);
  return d;
}

// multiplier applied to element (row, col): 0 if dropped, 1/(1-p) if kept, 1 when dropout is off
SRFRD_HD float drop_mul(const DropSite& d, int row, int col) {
  if (!d.on) return 1.0f;
  const uint32_t h = fmix32(d.h2 ^ (uint32_t)(row * 4096 + col));
  return h >= d.thr ? d.scale : 0.0f;
}

SRFRD_HD uint32_t step_seed(uint32_t base_seed, uint32_t step) { return fmix32(base_seed ^ fmix32(step)); }

}  // namespace srfrd
