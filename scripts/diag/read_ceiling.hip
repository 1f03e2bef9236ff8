// Diagnostic kernels for scripts/read_ceiling_probe.py -- NOT part of libbreach_hip.so.
//
// What is the fastest a launch shaped like kernel A's forward can pull two lists through the chip, with no arithmetic to speak
// of?  `diag_read` is that launch reduced to its memory side: a persistent grid, each workgroup walks chunks of 4096 floats of
// two buffers with eight staged 16-byte loads per lane (plain or non-temporal) and adds them up; `diag_fill` is the producer
// that leaves its output in L2 / the Infinity Cache the way autograd leaves `rec`.  Built by the probe with hipcc for gfx950.
#include <hip/hip_runtime.h>
#include <hip/hip_ext.h>
#include <stdint.h>

namespace {

constexpr int kBlock = 256;
constexpr int kChunk = 4096;                       // floats per chunk, as BH_GM_CHUNK
constexpr int kVec = kChunk / 4 / kBlock;          // 16-byte loads per lane, chunk and buffer (4)
typedef float v4f __attribute__((ext_vector_type(4)));

template <bool NT>
__device__ __forceinline__ v4f load16(const v4f* p) {
  if constexpr (NT) return __builtin_nontemporal_load(p);
  else return *p;
}

template <bool NT>
__global__ __launch_bounds__(kBlock) void diag_read_kernel(const float* __restrict__ a, const float* __restrict__ b, int64_t n_chunks,
                                                           float* __restrict__ out) {
  float acc = 0.f;
  for (int64_t c = blockIdx.x; c < n_chunks; c += gridDim.x) {
    const v4f* a4 = reinterpret_cast<const v4f*>(a + c * kChunk);
    const v4f* b4 = reinterpret_cast<const v4f*>(b + c * kChunk);
    v4f av[kVec], bv[kVec];
#pragma unroll
    for (int k = 0; k < kVec; ++k) av[k] = load16<NT>(a4 + threadIdx.x + k * kBlock);
#pragma unroll
    for (int k = 0; k < kVec; ++k) bv[k] = load16<NT>(b4 + threadIdx.x + k * kBlock);
#pragma unroll
    for (int k = 0; k < kVec; ++k) acc += (av[k].x * bv[k].x + av[k].y * bv[k].y) + (av[k].z * bv[k].z + av[k].w * bv[k].w);
  }
  if (acc == 123456.789f) out[blockIdx.x] = acc;  // keeps the loads alive; practically never taken
}

// The backward / multi-tensor shape reduced to its memory side: one workgroup per chunk (as gm_bwd_kernel and mt_kernel launch),
// four staged 16-byte loads per lane and buffer, out = a + 0.5 b, 16-byte stores (plain or non-temporal).
template <bool NTL, bool NTS>
__global__ __launch_bounds__(kBlock) void diag_rw_kernel(const float* __restrict__ a, const float* __restrict__ b, float* __restrict__ o) {
  const int64_t c = blockIdx.x;
  const v4f* a4 = reinterpret_cast<const v4f*>(a + c * kChunk);
  const v4f* b4 = reinterpret_cast<const v4f*>(b + c * kChunk);
  v4f* o4 = reinterpret_cast<v4f*>(o + c * kChunk);
  v4f av[kVec], bv[kVec];
#pragma unroll
  for (int k = 0; k < kVec; ++k) av[k] = load16<NTL>(a4 + threadIdx.x + k * kBlock);
#pragma unroll
  for (int k = 0; k < kVec; ++k) bv[k] = load16<NTL>(b4 + threadIdx.x + k * kBlock);
#pragma unroll
  for (int k = 0; k < kVec; ++k) {
    const v4f r = av[k] + 0.5f * bv[k];
    if constexpr (NTS) __builtin_nontemporal_store(r, o4 + threadIdx.x + k * kBlock);
    else o4[threadIdx.x + k * kBlock] = r;
  }
}

__global__ __launch_bounds__(kBlock) void diag_fill_kernel(float* __restrict__ p, int64_t n4, float value) {
  v4f v;
  v.x = v.y = v.z = v.w = value;
  for (int64_t i = (int64_t)blockIdx.x * kBlock + threadIdx.x; i < n4; i += (int64_t)gridDim.x * kBlock) reinterpret_cast<v4f*>(p)[i] = v;
}

}  // namespace

extern "C" {

// Reads n_chunks * 4096 floats of each buffer (16-byte aligned).  Returns 0 or the negated hipError_t.
int diag_read(const float* a, const float* b, int64_t n_chunks, int32_t grid, int32_t non_temporal, float* out, void* stream) {
  if (a == nullptr || b == nullptr || out == nullptr || n_chunks <= 0 || grid <= 0) return -1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  if (non_temporal) hipLaunchKernelGGL(diag_read_kernel<true>, dim3(grid), dim3(kBlock), 0, st, a, b, n_chunks, out);
  else hipLaunchKernelGGL(diag_read_kernel<false>, dim3(grid), dim3(kBlock), 0, st, a, b, n_chunks, out);
  return -(int)hipGetLastError();
}

// `reps` launches of diag_read, each bracketed by the dispatch's own start / stop events (hipExtLaunchKernelGGL -- the timestamps
// rocprofv3 reports, and the way libbreach_hip.so times kernel A), optionally each behind a diag_fill of `a`.  us_out[reps].
int diag_read_timed(const float* a, const float* b, int64_t n_chunks, int32_t grid, int32_t non_temporal, float* out, void* stream,
                    int32_t fill_first, int32_t reps, float* us_out) {
  if (a == nullptr || b == nullptr || out == nullptr || us_out == nullptr || n_chunks <= 0 || grid <= 0 || reps <= 0 || reps > 1024) return -1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev[2048];
  for (int i = 0; i < 2 * reps; ++i)
    if (hipEventCreate(&ev[i]) != hipSuccess) return -2;
  for (int r = 0; r < reps; ++r) {
    if (fill_first)
      hipLaunchKernelGGL(diag_fill_kernel, dim3(2048), dim3(kBlock), 0, st, const_cast<float*>(a), (n_chunks * kChunk) >> 2, 0.5f);
    if (non_temporal)
      hipExtLaunchKernelGGL(diag_read_kernel<true>, dim3(grid), dim3(kBlock), 0, st, ev[2 * r], ev[2 * r + 1], 0, a, b, n_chunks, out);
    else
      hipExtLaunchKernelGGL(diag_read_kernel<false>, dim3(grid), dim3(kBlock), 0, st, ev[2 * r], ev[2 * r + 1], 0, a, b, n_chunks, out);
  }
  int rc = -(int)hipGetLastError();
  if (hipStreamSynchronize(st) != hipSuccess) rc = rc ? rc : -3;
  for (int r = 0; r < reps; ++r) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]) != hipSuccess) rc = rc ? rc : -4;
    us_out[r] = ms * 1e3f;
  }
  for (int i = 0; i < 2 * reps; ++i) (void)hipEventDestroy(ev[i]);
  return rc;
}

// `reps` launches of diag_rw (read a, b; write o; n_chunks workgroups), dispatch start / stop events; policy 0 plain, 1 non-temporal
// loads, 2 non-temporal loads and stores (kernel A's BH_GM_CACHE_KEEP / _STREAM / _STREAM_ALL); optionally each behind a diag_fill of `a`.
int diag_rw_timed(const float* a, const float* b, float* o, int64_t n_chunks, int32_t policy, void* stream, int32_t fill_first, int32_t reps,
                  float* us_out) {
  if (a == nullptr || b == nullptr || o == nullptr || us_out == nullptr || n_chunks <= 0 || reps <= 0 || reps > 1024) return -1;
  hipStream_t st = static_cast<hipStream_t>(stream);
  hipEvent_t ev[2048];
  for (int i = 0; i < 2 * reps; ++i)
    if (hipEventCreate(&ev[i]) != hipSuccess) return -2;
  const dim3 grid((unsigned)n_chunks), block(kBlock);
  for (int r = 0; r < reps; ++r) {
    if (fill_first)
      hipLaunchKernelGGL(diag_fill_kernel, dim3(2048), dim3(kBlock), 0, st, const_cast<float*>(a), (n_chunks * kChunk) >> 2, 0.5f);
    if (policy == 2) hipExtLaunchKernelGGL((diag_rw_kernel<true, true>), grid, block, 0, st, ev[2 * r], ev[2 * r + 1], 0, a, b, o);
    else if (policy == 1) hipExtLaunchKernelGGL((diag_rw_kernel<true, false>), grid, block, 0, st, ev[2 * r], ev[2 * r + 1], 0, a, b, o);
    else hipExtLaunchKernelGGL((diag_rw_kernel<false, false>), grid, block, 0, st, ev[2 * r], ev[2 * r + 1], 0, a, b, o);
  }
  int rc = -(int)hipGetLastError();
  if (hipStreamSynchronize(st) != hipSuccess) rc = rc ? rc : -3;
  for (int r = 0; r < reps; ++r) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, ev[2 * r], ev[2 * r + 1]) != hipSuccess) rc = rc ? rc : -4;
    us_out[r] = ms * 1e3f;
  }
  for (int i = 0; i < 2 * reps; ++i) (void)hipEventDestroy(ev[i]);
  return rc;
}

int diag_fill(float* p, int64_t n_floats, float value, void* stream) {
  if (p == nullptr || n_floats <= 0 || (n_floats & 3)) return -1;
  hipLaunchKernelGGL(diag_fill_kernel, dim3(2048), dim3(kBlock), 0, static_cast<hipStream_t>(stream), p, n_floats >> 2, value);
  return -(int)hipGetLastError();
}

}  // extern "C"
