// Internal helpers shared by the gfx950 kernels of libbreach_hip.so.  Not part of the C ABI.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "breach_hip.h"

namespace bh {

constexpr int kWave = 64;  // CDNA4 wavefront width; hard-coded on purpose (gfx950 only)
constexpr int kBlock = 256;
constexpr int kWavesPerBlock = kBlock / kWave;

inline int hip_status(hipError_t e) { return e == hipSuccess ? 0 : -(1000 + static_cast<int>(e)); }

inline int launch_status() { return hip_status(hipGetLastError()); }

inline hipStream_t as_stream(void* s) { return static_cast<hipStream_t>(s); }

// Sum over the 64 lanes of a wavefront; the total lands in lane 0 (other lanes hold partial garbage).
__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
  for (int off = kWave / 2; off > 0; off >>= 1) v += __shfl_down(v, off, kWave);
  return v;
}

// Block-wide sum of K doubles per thread for a 256-thread block.  Result valid in thread 0.
// `lds` must hold kWavesPerBlock * K doubles.  Fixed combine order => bitwise reproducible.
template <int K>
__device__ __forceinline__ void block_sum(double (&v)[K], double* lds) {
  const int lane = threadIdx.x & (kWave - 1);
  const int wave = threadIdx.x >> 6;
#pragma unroll
  for (int k = 0; k < K; ++k) v[k] = wave_sum(v[k]);
  if (lane == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) lds[wave * K + k] = v[k];
  }
  __syncthreads();
  if (threadIdx.x == 0) {
#pragma unroll
    for (int k = 0; k < K; ++k) {
      double s = lds[k];
      for (int w = 1; w < (int)(blockDim.x >> 6); ++w) s += lds[w * K + k];
      v[k] = s;
    }
  }
}

// Channel geometry shared by kernel D's forward (bn_sums) and kernel E (eval-mode BatchNorm): the B planes of a channel form one
// virtual array of B * HW elements; narrow (fewer than 2048 of them): one wavefront per channel, four channels per workgroup;
// otherwise the channel is cut into S slabs of about 8192 elements, one workgroup each, at most 64.  Both kernels use the SAME
// rule so that kernel E's forward can write the per-(channel, slab) sums kernel D's finalize reads.
constexpr int64_t kChannelSlabTarget = 8192;
constexpr int64_t kChannelNarrowLimit = 2048;
constexpr int kChannelMaxSlabs = 64;

inline void channel_geometry(int64_t per_channel, int32_t& S, int32_t& narrow) {
  narrow = per_channel < kChannelNarrowLimit ? 1 : 0;
  int64_t s = narrow ? 1 : (per_channel + kChannelSlabTarget / 2) / kChannelSlabTarget;
  S = (int32_t)(s < 1 ? 1 : (s > kChannelMaxSlabs ? kChannelMaxSlabs : s));
}

__device__ __forceinline__ float sgnf(float e) { return static_cast<float>((e > 0.f) - (e < 0.f)); }

}  // namespace bh
