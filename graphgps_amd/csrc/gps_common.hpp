// Shared host/device helpers for libgps_hip.so (gfx950 only; no multi-backend paths).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdint>
#include <cstdio>

#include "gps_hip.h"

namespace gps {

void set_error(const char* fmt, ...);
int launch_status(const char* what);  // hipGetLastError -> GPS_OK / GPS_ELAUNCH (+ message)

static inline hipStream_t as_stream(gps_stream_t s) { return reinterpret_cast<hipStream_t>(s); }

// Device-resident dropout "salt" (one uint64, owned by the caller, NULL = none).  Every dropout
// kernel adds salt[0] * golden-ratio to its by-value seed, so a step captured in a hipGraph draws
// fresh masks on every replay when the captured step increments the salt (graphgps_amd/ops.py).
const uint64_t* dropout_salt();
__device__ __forceinline__ uint64_t salted_seed(uint64_t seed, const uint64_t* salt) {
  return salt ? seed + salt[0] * 0x9E3779B97F4A7C15ULL : seed;
}

// max|tensor| RECORDS of the fp16-form GEMMs (csrc/gemm_panel.hip, csrc/wgrad.hip): kAmaxWords uint32 words holding fp32
// bit patterns, kAmaxStride words (256 bytes) apart; the tensor's maximum is the (unsigned) maximum of the words.
// Producers raise word (blockIdx & 7) with one atomic per workgroup (or wavefront).  Why eight LINES: atomics on one
// cache line serialise at the L2 at ~6 ns each (measured twice: one atomic per wavefront on ONE word made a 12 MB max
// pass take 20 us whatever its size; the attention forward's 4,096 per-wavefront atomics on eight words of ONE line took
// it from 24 to 60 us).  Zero at allocation; only ever raised.
constexpr int kAmaxWords = 8;
constexpr int kAmaxStride = 64;                            // words between the words of a record
constexpr int kAmaxRecordWords = kAmaxWords * kAmaxStride;  // footprint of a record: 2 KB
#if defined(__HIPCC__)
// biased exponent of the record's maximum, floored at 16 (tensors below 2^-111 are scaled as if they were that large)
__device__ __forceinline__ unsigned amax_be(const uint32_t* rec) {
  unsigned m = 0;
#pragma unroll
  for (int i = 0; i < kAmaxWords; ++i) m = max(m, __builtin_nontemporal_load(rec + i * kAmaxStride));
  const unsigned be = (m >> 23) & 255u;
  return be < 16u ? 16u : be;
}
// `spread`: which of the eight words this caller raises (workgroups: blockIdx; wavefronts: their global index)
__device__ __forceinline__ void amax_raise(uint32_t* rec, uint32_t bits, unsigned spread) {
  if (bits) atomicMax(rec + (spread & (kAmaxWords - 1)) * kAmaxStride, bits);
}
__device__ __forceinline__ void amax_raise(uint32_t* rec, uint32_t bits) { amax_raise(rec, bits, blockIdx.x); }
#endif

// Fill `words` 32-bit words with `value` by a KERNEL launch instead of hipMemsetAsync: a captured step then holds kernel
// nodes only.  (Round 6 measured it: under rocprofv3 the memset nodes at the head of the replayed step sat behind 75 - 105 us
// idle gaps, ~260 us per step, and the fill kernels do not -- but the untraced step is the same 8.18 ms either way: the gaps
// are the tracer's, not the step's.  Kept because one node type is simpler to reason about.)
void fill_words(void* dst, uint32_t value, size_t words, hipStream_t stream);

static inline unsigned grid_for(int64_t work, int block) {
  return static_cast<unsigned>((work + block - 1) / block);
}

}  // namespace gps

#define GPS_REQUIRE(cond, ...)        \
  do {                                \
    if (!(cond)) {                    \
      gps::set_error(__VA_ARGS__);    \
      return GPS_EINVAL;              \
    }                                 \
  } while (0)

constexpr int kWave = 64;  // CDNA wavefront
