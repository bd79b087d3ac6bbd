// Helpers shared by the segment-attention kernels (seg_attention.hip: one wavefront per (16-row tile, head),
// operands straight from global; sattn.hip: one workgroup per (graph, 64-query block, head group), operands
// staged through LDS).  gfx950 only.
#pragma once
#include "gps_common.hpp"

namespace attn {

typedef float f32x4 __attribute__((ext_vector_type(4)));

// v_mfma_f32_16x16x4_f32: A[m = lane & 15][k = lane >> 4], B[k = lane >> 4][n = lane & 15],
// C[m = 4 * (lane >> 4) + r][n = lane & 15] in register r.
__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// --- counter-based attention-dropout mask -------------------------------------------------------
// keep(seed, row id = query*H + head, key index local to the graph).  ONE 32-bit hash decides TWO adjacent keys
// (2k, 2k+1): 16 bits each against thr16 = round(p * 65536) (the hash -- two quarter-rate 32-bit multiplies --
// was the largest VALU item of these kernels).  The drop probability is therefore thr16 / 65536 (p = 0.1 ->
// 0.100006) and the survivors are scaled by 65536 / (65536 - thr16), so the mask is exactly unbiased.
// Mirrored bit-for-bit by graphgps_amd/ops.py:attn_dropout_keep_mask(paired=True) (used by the parity tests).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t row_hash(uint32_t rowid, uint64_t seed) {
  return mix32(rowid ^ (uint32_t)seed) + (uint32_t)(seed >> 32);
}
__device__ __forceinline__ uint32_t pair_hash(uint32_t rh, uint32_t key_pair) {
  return mix32(rh + key_pair * 0x9E3779B9U);
}
__device__ __forceinline__ bool keep_lo(uint32_t h, uint32_t thr16) { return (h & 0xFFFFu) >= thr16; }
__device__ __forceinline__ bool keep_hi(uint32_t h, uint32_t thr16) { return (h >> 16) >= thr16; }
__device__ __forceinline__ bool keep_elem(uint32_t rh, uint32_t key_local, uint32_t thr16) {
  const uint32_t h = pair_hash(rh, key_local >> 1);
  return (key_local & 1u) ? keep_hi(h, thr16) : keep_lo(h, thr16);
}
__host__ __device__ __forceinline__ uint32_t drop_thr16(float p_drop) { return (uint32_t)(p_drop * 65536.0f + 0.5f); }
__host__ __device__ __forceinline__ float drop_inv_keep(uint32_t thr16) { return 65536.0f / (float)(65536u - thr16); }

// exp for the softmax numerators: exp2(x * log2 e) on the transcendental unit (v_exp_f32).  Arguments are
// <= 0 and rarely below -20, where the product's rounding costs <= 2e-6 relative -- inside the 1e-5 budget --
// against ~8 extra VALU instructions per element for the correctly rounded expf (the kernels are
// VALU-issue-bound).  GPS_ATTN_EXACT_EXP=1 at compile time restores expf.
__device__ __forceinline__ float sm_exp(float x) {
#ifdef GPS_ATTN_EXACT_EXP
  return expf(x);
#else
  return __builtin_amdgcn_exp2f(x * 1.44269504088896341f);
#endif
}

__device__ __forceinline__ float group_max(float v) {  // over the 4 lane groups (same lane & 15)
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// LDS-staged kernels (sattn.hip): taken by the plain (unbiased) entry points when the head shape has a compiled
// instantiation; the backward additionally needs every graph to fit one 64-row block.
bool sattn_applicable(const void* qkv, int64_t ld_qkv, const void* out, int H, int dh);
void sattn_fwd_launch(const float* qkv, int64_t ld_qkv, const int32_t* ptr, int64_t B, int64_t N, int H, int dh,
                      float scale, float p_drop, uint64_t seed, float* out, float* lse, uint32_t* amax, const int32_t* order,
                      hipStream_t s);
void sattn_bwd_launch(const float* d_out, const float* qkv, int64_t ld_qkv, const float* out, const float* lse,
                      const int32_t* ptr, int64_t B, int64_t N, int H, int dh, float scale, float p_drop,
                      uint64_t seed, float* d_qkv, int64_t ld_dqkv, uint32_t* amax, const int32_t* order, hipStream_t s);

}  // namespace attn
