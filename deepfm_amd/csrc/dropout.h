// Counter-based dropout shared by the DNN-tower kernels: keep(i) = hash(seed, salt, i) >= p * 2^32,
// scale 1/(1-p).  The seed is read from device memory, so forward and backward rebuild the same
// mask and a replayed graph sees a fresh seed every step.
#pragma once

#include "common.h"

namespace dfm {

__device__ __forceinline__ uint32_t mix32(uint64_t x) {
  x ^= x >> 33; x *= 0xff51afd7ed558ccdull;
  x ^= x >> 33; x *= 0xc4ceb9fe1a85ec53ull;
  x ^= x >> 33;
  return static_cast<uint32_t>(x);
}
__device__ __forceinline__ float drop_scale(int64_t seed, int salt, int64_t idx, uint32_t thresh, float inv_keep) {
  if (thresh == 0) return 1.f;
  const uint32_t r = mix32(static_cast<uint64_t>(seed) * 0x9E3779B97F4A7C15ull + (static_cast<uint64_t>(salt) << 40) + static_cast<uint64_t>(idx));
  return r >= thresh ? inv_keep : 0.f;
}
static inline uint32_t dropout_thresh(float p) {
  if (p <= 0.f) return 0;
  const double t = static_cast<double>(p) * 4294967296.0;
  return t >= 4294967295.0 ? 4294967295u : static_cast<uint32_t>(t);
}

}  // namespace dfm
