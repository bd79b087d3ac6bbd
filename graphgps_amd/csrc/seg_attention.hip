// Segment (per-graph, variable-length) multi-head attention on fp32 MFMA.
//
// Replaces the dense path of the reference's Transformer branch
// (graphgps/layer/gps_layer.py:199-201,234-241): to_dense_batch padding -> torch.nn.MultiheadAttention
// core (softmax(q k^T / sqrt(dh) + key_padding_mask) -> dropout -> . v) -> boolean-mask un-pad.
// Here graphs are walked through `ptr`; nothing is padded and no mask tensor exists.
//
// Machine mapping (gfx950): one wavefront owns one (16-row tile, head) work item and keeps the
// whole online-softmax state in registers.  All contractions use v_mfma_f32_16x16x4_f32
// (exact fp32 FMA chain, 32 cycles/issue; 16x16 tiles waste the least on 4..64-node graphs).
// The transposed products are computed so that NO operand ever needs a cross-lane shuffle or
// an LDS round trip:
//     S^T = K . Q^T      (C layout: lane (i = l&15, g = l>>4) holds keys 4g+r, query i)
//     O^T = V^T . P^T    (the S^T accumulator registers ARE the B operand; the contraction
//                         index of an MFMA is order-free, so k-slot g of step r is key 4g+r)
// and the same trick gives dQ^T = K^T dS^T, dV^T = dO^T P, dK^T = Q^T dS in the backward.
// The head dimension is split over the 4 lane groups (KPL = ceil(dh/4) contiguous floats per
// lane), output tiles are DT = ceil(dh/16) accumulators of 4 registers.
//
// Lane algebra validated against a numpy model of the MFMA lane map before being written
// (see DESIGN.md "segment attention"); numerics validated against the CPU oracle in
// tests/test_hip_ops.py.
#include "gps_common.hpp"

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}

// --- counter-based dropout mask -------------------------------------------------------------
// keep(seed, row id = query*H + head, key index local to the graph).  Mirrored bit-for-bit by
// graphgps_amd/ops.py:attn_dropout_keep_mask (used by the parity tests).
__device__ __forceinline__ uint32_t mix32(uint32_t x) {
  x ^= x >> 16; x *= 0x7feb352dU; x ^= x >> 15; x *= 0x846ca68bU; x ^= x >> 16;
  return x;
}
__device__ __forceinline__ uint32_t row_hash(uint32_t rowid, uint64_t seed) {
  return mix32(rowid ^ (uint32_t)seed) + (uint32_t)(seed >> 32);
}
__device__ __forceinline__ bool keep_elem(uint32_t rh, uint32_t key_local, float p_drop) {
  const uint32_t r = mix32(rh + key_local * 0x9E3779B9U);
  return (float)(r >> 8) * (1.0f / 16777216.0f) >= p_drop;
}

template <int DH>
struct Geo {
  static constexpr int KPL = (DH + 3) / 4;    // contraction elements per lane group
  static constexpr int DT = (DH + 15) / 16;   // 16-wide output tiles along dh
};

// Load the KPL contiguous floats this lane contributes to a dh-contraction for `row`.
template <int DH>
__device__ __forceinline__ void load_kslice(const float* __restrict__ base, int64_t ld, int row,
                                            bool row_ok, int col0, int grp, float scale,
                                            float (&dst)[Geo<DH>::KPL]) {
  constexpr int KPL = Geo<DH>::KPL;
  const float* p = base + (int64_t)row * ld + col0 + grp * KPL;
#pragma unroll
  for (int s = 0; s < KPL; ++s) {
    const bool ok = row_ok && (grp * KPL + s < DH);
    dst[s] = ok ? p[s] * scale : 0.0f;
  }
}

__device__ __forceinline__ float group_max(float v) {  // over the 4 lane groups (same l&15)
  v = fmaxf(v, __shfl_xor(v, 16));
  v = fmaxf(v, __shfl_xor(v, 32));
  return v;
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16);
  v += __shfl_xor(v, 32);
  return v;
}

// =============================================================================================
// forward
// =============================================================================================
// One 64-key block with NT (1..4) live 16-key tiles.  Branch-free and fully unrolled on purpose:
// every K and V operand load of the block is issued before the first MFMA, so a wave pays ONE
// memory round trip per block instead of one per tile (the kernel is latency-bound at molecule
// sizes: ~30 x 30 x 24 per (graph, head)).  Rows past the graph end are clamped to its last row
// (finite data) and their scores masked to -inf, so they contribute exactly 0.
template <int DH, bool DROP, int NT>
__device__ __forceinline__ void attn_fwd_block(
    const float* __restrict__ qkv, int64_t ld, int kb, int n0, int n1, int h, int d, int i, int grp,
    const float (&qv)[Geo<DH>::KPL], uint32_t rh, float p_drop, float inv_keep, float& m, float& lsum,
    f32x4 (&oacc)[Geo<DH>::DT]) {
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT;
  float kv[NT][KPL];
  float vv[DT][NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int krow = min(kb + 16 * t + i, n1 - 1);
    load_kslice<DH>(qkv, ld, krow, true, d + h * DH, grp, 1.0f, kv[t]);
  }
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = min(kb + 16 * t + 4 * grp + r, n1 - 1);
        const int col = dt * 16 + i;
        vv[dt][t][r] = col < DH ? qkv[(int64_t)key * ld + 2 * d + h * DH + col] : 0.0f;
      }
  f32x4 s[NT];
  float mloc = -INFINITY;
#pragma unroll
  for (int t = 0; t < NT; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < KPL; ++c)          // tiles interleaved: NT independent accumulator chains
#pragma unroll
    for (int t = 0; t < NT; ++t) s[t] = mfma16(kv[t][c], qv[c], s[t]);
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 16 * t + 4 * grp + r;
      s[t][r] = key < n1 ? s[t][r] : -INFINITY;
      mloc = fmaxf(mloc, s[t][r]);
    }
  const float mnew = fmaxf(m, group_max(mloc));
  const float alpha = expf(m - mnew);  // m = -inf on the first block -> 0
  m = mnew;
  float psum = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = expf(s[t][r] - mnew);  // masked keys: exp(-inf) = 0
      psum += p;
      if (DROP) {
        const uint32_t key_local = (uint32_t)(kb + 16 * t + 4 * grp + r - n0);
        p = keep_elem(rh, key_local, p_drop) ? p * inv_keep : 0.0f;
      }
      s[t][r] = p;
    }
  lsum = lsum * alpha + psum;
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] *= alpha;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)    // DT independent accumulator chains interleaved
        oacc[dt] = mfma16(vv[dt][t][r], s[t][r], oacc[dt]);
}

template <int DH, bool DROP>
__global__ __launch_bounds__(256) void k_attn_fwd(
    const float* __restrict__ qkv, int64_t ld, const int32_t* __restrict__ ptr,
    const int32_t* __restrict__ tile_graph, const int32_t* __restrict__ tile_row0,
    int64_t n_work, int64_t N, int H, float scale, float p_drop, uint64_t seed, const uint64_t* __restrict__ salt,
    float* __restrict__ out, float* __restrict__ lse) {
  seed = gps::salted_seed(seed, salt);
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT, KT = 4;
  const int lane = threadIdx.x & 63;
  const int64_t w = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= n_work) return;
  const int64_t tile = w / H;
  const int h = (int)(w - tile * H);
  const int g = tile_graph[tile];
  if (g < 0) return;
  const int q0 = tile_row0[tile];
  const int n0 = ptr[g], n1 = ptr[g + 1];
  const int i = lane & 15, grp = lane >> 4;
  const int d = H * DH;
  const int qrow = q0 + i;
  const bool q_ok = qrow < n1;

  float qv[KPL];
  load_kslice<DH>(qkv, ld, qrow, q_ok, h * DH, grp, scale, qv);
  const uint32_t rh = DROP ? row_hash((uint32_t)qrow * (uint32_t)H + (uint32_t)h, seed) : 0u;
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;

  float m = -INFINITY, lsum = 0.0f;
  f32x4 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kb = n0; kb < n1; kb += 16 * KT) {
    const int nt = min(KT, (n1 - kb + 15) >> 4);  // wave-uniform
    switch (nt) {
      case 1: attn_fwd_block<DH, DROP, 1>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, rh, p_drop, inv_keep, m, lsum, oacc); break;
      case 2: attn_fwd_block<DH, DROP, 2>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, rh, p_drop, inv_keep, m, lsum, oacc); break;
      case 3: attn_fwd_block<DH, DROP, 3>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, rh, p_drop, inv_keep, m, lsum, oacc); break;
      default: attn_fwd_block<DH, DROP, 4>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, rh, p_drop, inv_keep, m, lsum, oacc); break;
    }
  }
  const float ltot = group_sum(lsum);
  const float inv_l = 1.0f / ltot;
  if (q_ok) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int col = dt * 16 + 4 * grp;
      float* o = out + (int64_t)qrow * d + h * DH + col;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < DH) o[r] = oacc[dt][r] * inv_l;
    }
    if (grp == 0) lse[(int64_t)h * N + qrow] = m + logf(ltot);
  }
}

// =============================================================================================
// backward, query-tile keyed: dQ (+ delta)
// =============================================================================================
// Same discipline as the forward block: one 64-key block with NT live tiles, branch-free, every K / V
// operand (row slices for S^T and dP^T, column form of K for dQ^T) loaded before the first MFMA.
// Key rows past the graph end are clamped to its last row (finite data); their P is forced to 0, so
// dS = 0 and they contribute nothing.
template <int DH, bool DROP, int NT>
__device__ __forceinline__ void attn_dq_block(
    const float* __restrict__ qkv, int64_t ld, int kb, int n0, int n1, int h, int d, int i, int grp,
    const float (&qv)[Geo<DH>::KPL], const float (&dov)[Geo<DH>::KPL], float lse_q, float dl_q,
    uint32_t rh, float p_drop, float inv_keep, f32x4 (&acc)[Geo<DH>::DT]) {
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT;
  float kv[NT][KPL], vv[NT][KPL];
  float kc[DT][NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int krow = min(kb + 16 * t + i, n1 - 1);
    load_kslice<DH>(qkv, ld, krow, true, d + h * DH, grp, 1.0f, kv[t]);
    load_kslice<DH>(qkv, ld, krow, true, 2 * d + h * DH, grp, 1.0f, vv[t]);
  }
#pragma unroll
  for (int dt = 0; dt < DT; ++dt)
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = min(kb + 16 * t + 4 * grp + r, n1 - 1);
        const int col = dt * 16 + i;
        kc[dt][t][r] = col < DH ? qkv[(int64_t)key * ld + d + h * DH + col] : 0.0f;
      }
  f32x4 s[NT], dp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int c = 0; c < KPL; ++c)          // 2*NT independent accumulator chains interleaved
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = mfma16(kv[t][c], qv[c], s[t]);      // S^T[key][query]
      dp[t] = mfma16(vv[t][c], dov[c], dp[t]);   // dP^T[key][query]
    }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 16 * t + 4 * grp + r;
      const float p = key < n1 ? expf(s[t][r] - lse_q) : 0.0f;
      float dpe = dp[t][r];
      if (DROP) dpe = keep_elem(rh, (uint32_t)(key - n0), p_drop) ? dpe * inv_keep : 0.0f;
      s[t][r] = p * (dpe - dl_q);                // dS^T, reused as the B operand below
    }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)            // dQ^T[dh][query] += K^T[dh][key] dS^T[key][query]
        acc[dt] = mfma16(kc[dt][t][r], s[t][r], acc[dt]);
}

template <int DH, bool DROP>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(
    const float* __restrict__ d_out, const float* __restrict__ qkv, int64_t ld,
    const float* __restrict__ out, const float* __restrict__ lse, float* __restrict__ delta,
    const int32_t* __restrict__ ptr, const int32_t* __restrict__ tile_graph,
    const int32_t* __restrict__ tile_row0, int64_t n_work, int64_t N, int H, float scale,
    float p_drop, uint64_t seed, const uint64_t* __restrict__ salt, float* __restrict__ d_qkv, int64_t ldg) {
  seed = gps::salted_seed(seed, salt);
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT, KT = 4;
  const int lane = threadIdx.x & 63;
  const int64_t w = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= n_work) return;
  const int64_t tile = w / H;
  const int h = (int)(w - tile * H);
  const int g = tile_graph[tile];
  if (g < 0) return;
  const int q0 = tile_row0[tile];
  const int n0 = ptr[g], n1 = ptr[g + 1];
  const int i = lane & 15, grp = lane >> 4;
  const int d = H * DH;
  const int qrow = q0 + i;
  const bool q_ok = qrow < n1;

  float qv[KPL], dov[KPL], ov[KPL];
  load_kslice<DH>(qkv, ld, qrow, q_ok, h * DH, grp, scale, qv);
  load_kslice<DH>(d_out, d, qrow, q_ok, h * DH, grp, 1.0f, dov);
  load_kslice<DH>(out, d, qrow, q_ok, h * DH, grp, 1.0f, ov);
  const float lse_q = q_ok ? lse[(int64_t)h * N + qrow] : 0.0f;
  // delta[h][q] = sum_c dO[q][h*DH+c] * O[q][h*DH+c]  (6 of the dh products per lane group)
  float dl_part = 0.0f;
#pragma unroll
  for (int c = 0; c < KPL; ++c) dl_part += dov[c] * ov[c];
  const float dl_q = group_sum(dl_part);
  if (q_ok && grp == 0) delta[(int64_t)h * N + qrow] = dl_q;   // read by k_attn_bwd_dkv (same stream)
  const uint32_t rh = DROP ? row_hash((uint32_t)qrow * (uint32_t)H + (uint32_t)h, seed) : 0u;
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;

  f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kb = n0; kb < n1; kb += 16 * KT) {
    const int nt = min(KT, (n1 - kb + 15) >> 4);  // wave-uniform
    switch (nt) {
      case 1: attn_dq_block<DH, DROP, 1>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, dov, lse_q, dl_q, rh, p_drop, inv_keep, acc); break;
      case 2: attn_dq_block<DH, DROP, 2>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, dov, lse_q, dl_q, rh, p_drop, inv_keep, acc); break;
      case 3: attn_dq_block<DH, DROP, 3>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, dov, lse_q, dl_q, rh, p_drop, inv_keep, acc); break;
      default: attn_dq_block<DH, DROP, 4>(qkv, ld, kb, n0, n1, h, d, i, grp, qv, dov, lse_q, dl_q, rh, p_drop, inv_keep, acc); break;
    }
  }
  if (q_ok) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int col = dt * 16 + 4 * grp;
      float* o = d_qkv + (int64_t)qrow * ldg + h * DH + col;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < DH) o[r] = acc[dt][r] * scale;
    }
  }
}

// =============================================================================================
// backward, key-tile keyed: dK, dV
// =============================================================================================
// One 32-query block with NT (1..2) live tiles, all Q / dO operands (row slices for S and dP, column
// forms for dK^T and dV^T) and the per-query lse / delta loaded before the first MFMA.
template <int DH, bool DROP, int NT>
__device__ __forceinline__ void attn_dkv_block(
    const float* __restrict__ d_out, const float* __restrict__ qkv, int64_t ld,
    const float* __restrict__ lse, const float* __restrict__ delta, int64_t N, int H, int qb, int n0,
    int n1, int h, int d, int i, int grp, int krow, const float (&kv)[Geo<DH>::KPL],
    const float (&vv)[Geo<DH>::KPL], float scale, uint64_t seed, float p_drop, float inv_keep,
    f32x4 (&dk)[Geo<DH>::DT], f32x4 (&dv)[Geo<DH>::DT]) {
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT;
  float qa[NT][KPL], da[NT][KPL];
  float qc[DT][NT][4], dc[DT][NT][4];
  float lq[NT][4], dq[NT][4];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int qrow = min(qb + 16 * t + i, n1 - 1);
    load_kslice<DH>(qkv, ld, qrow, true, h * DH, grp, scale, qa[t]);
    load_kslice<DH>(d_out, d, qrow, true, h * DH, grp, 1.0f, da[t]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = min(qb + 16 * t + 4 * grp + r, n1 - 1);
      lq[t][r] = lse[(int64_t)h * N + qq];
      dq[t][r] = delta[(int64_t)h * N + qq];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const int col = dt * 16 + i;
        const bool ok = col < DH;
        dc[dt][t][r] = ok ? d_out[(int64_t)qq * d + h * DH + col] : 0.0f;
        qc[dt][t][r] = ok ? qkv[(int64_t)qq * ld + h * DH + col] * scale : 0.0f;
      }
    }
  f32x4 s[NT], dp[NT];
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }
#pragma unroll
  for (int c = 0; c < KPL; ++c)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
      s[t] = mfma16(qa[t][c], kv[c], s[t]);      // S[query][key]   (C layout: query = 4*grp+r, key = i)
      dp[t] = mfma16(da[t][c], vv[c], dp[t]);    // dP[query][key]
    }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qb + 16 * t + 4 * grp + r;
      const float pr = qq < n1 ? expf(s[t][r] - lq[t][r]) : 0.0f;
      float dpe = dp[t][r];
      float pd = pr;
      if (DROP) {
        const uint32_t rh = row_hash((uint32_t)qq * (uint32_t)H + (uint32_t)h, seed);
        const bool keep = keep_elem(rh, (uint32_t)(krow - n0), p_drop);
        pd = keep ? pr * inv_keep : 0.0f;
        dpe = keep ? dpe * inv_keep : 0.0f;
      }
      s[t][r] = pd;                              // P_drop
      dp[t][r] = pr * (dpe - dq[t][r]);          // dS
    }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        dv[dt] = mfma16(dc[dt][t][r], s[t][r], dv[dt]);    // dV^T[dh][key] += dO^T[dh][q] P_drop[q][key]
        dk[dt] = mfma16(qc[dt][t][r], dp[t][r], dk[dt]);   // dK^T[dh][key] += (scale Q)^T[dh][q] dS[q][key]
      }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(
    const float* __restrict__ d_out, const float* __restrict__ qkv, int64_t ld,
    const float* __restrict__ lse, const float* __restrict__ delta,
    const int32_t* __restrict__ ptr, const int32_t* __restrict__ tile_graph,
    const int32_t* __restrict__ tile_row0, int64_t n_work, int64_t N, int H, float scale,
    float p_drop, uint64_t seed, const uint64_t* __restrict__ salt, float* __restrict__ d_qkv, int64_t ldg) {
  seed = gps::salted_seed(seed, salt);
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT, QT = 2;
  const int lane = threadIdx.x & 63;
  const int64_t w = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= n_work) return;
  const int64_t tile = w / H;
  const int h = (int)(w - tile * H);
  const int g = tile_graph[tile];
  if (g < 0) return;
  const int k0 = tile_row0[tile];
  const int n0 = ptr[g], n1 = ptr[g + 1];
  const int i = lane & 15, grp = lane >> 4;
  const int d = H * DH;
  const int krow = k0 + i;
  const bool k_ok = krow < n1;
  const float inv_keep = DROP ? 1.0f / (1.0f - p_drop) : 1.0f;

  float kv[KPL], vv[KPL];
  load_kslice<DH>(qkv, ld, krow, k_ok, d + h * DH, grp, 1.0f, kv);
  load_kslice<DH>(qkv, ld, krow, k_ok, 2 * d + h * DH, grp, 1.0f, vv);

  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  for (int qb = n0; qb < n1; qb += 16 * QT) {
    if (n1 - qb > 16)
      attn_dkv_block<DH, DROP, 2>(d_out, qkv, ld, lse, delta, N, H, qb, n0, n1, h, d, i, grp, krow, kv, vv,
                                  scale, seed, p_drop, inv_keep, dk, dv);
    else
      attn_dkv_block<DH, DROP, 1>(d_out, qkv, ld, lse, delta, N, H, qb, n0, n1, h, d, i, grp, krow, kv, vv,
                                  scale, seed, p_drop, inv_keep, dk, dv);
  }
  if (k_ok) {
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int col = dt * 16 + 4 * grp;
      float* ok_ = d_qkv + (int64_t)krow * ldg + d + h * DH + col;
      float* ov_ = d_qkv + (int64_t)krow * ldg + 2 * d + h * DH + col;
#pragma unroll
      for (int r = 0; r < 4; ++r)
        if (col + r < DH) {
          ok_[r] = dk[dt][r];
          ov_[r] = dv[dt][r];
        }
    }
  }
}

#define GPS_FOR_EACH_DH(X) X(4) X(6) X(8) X(12) X(13) X(16) X(18) X(24) X(32) X(48) X(64) X(76) X(96) X(128)

}  // namespace

extern "C" {

int gps_attn_supported_head_dim(int dh) {
  switch (dh) {
#define X(D) case D:
    GPS_FOR_EACH_DH(X)
#undef X
    return 1;
    default:
      return 0;
  }
}

int gps_seg_attn_fwd(const float* qkv, int64_t ld_qkv, const int32_t* ptr,
                     const int32_t* tile_graph, const int32_t* tile_row0, int64_t max_tiles,
                     int64_t N, int H, int dh, float scale, float p_drop, uint64_t seed, float* out,
                     float* lse, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && H > 0 && dh > 0 && max_tiles >= 0 && ld_qkv >= 3LL * H * dh,
              "gps_seg_attn_fwd: bad sizes N=%lld H=%d dh=%d ld=%lld", (long long)N, H, dh, (long long)ld_qkv);
  GPS_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "gps_seg_attn_fwd: p_drop=%f outside [0,1)", p_drop);
  GPS_REQUIRE(N * (int64_t)H < INT32_MAX, "gps_seg_attn_fwd: N*H exceeds the 32-bit dropout row id");
  if (N == 0 || max_tiles == 0) return GPS_OK;
  GPS_REQUIRE(qkv && ptr && tile_graph && tile_row0 && out && lse, "gps_seg_attn_fwd: null buffer");
  if (!gps_attn_supported_head_dim(dh)) {
    gps::set_error("gps_seg_attn_fwd: head dim %d has no compiled kernel", dh);
    return GPS_EUNSUPPORTED;
  }
  const int64_t n_work = max_tiles * H;
  const unsigned grid = gps::grid_for(n_work, 4);
  hipStream_t s = gps::as_stream(stream);
  switch (dh) {
#define X(D)                                                                                      \
  case D:                                                                                         \
    if (p_drop > 0.0f)                                                                            \
      k_attn_fwd<D, true><<<grid, 256, 0, s>>>(qkv, ld_qkv, ptr, tile_graph, tile_row0, n_work, N, \
                                               H, scale, p_drop, seed, gps::dropout_salt(), out, lse);                 \
    else                                                                                          \
      k_attn_fwd<D, false><<<grid, 256, 0, s>>>(qkv, ld_qkv, ptr, tile_graph, tile_row0, n_work,  \
                                                N, H, scale, p_drop, seed, gps::dropout_salt(), out, lse);             \
    break;
    GPS_FOR_EACH_DH(X)
#undef X
  }
  return gps::launch_status("gps_seg_attn_fwd");
}

int gps_seg_attn_bwd(const float* d_out, const float* qkv, int64_t ld_qkv, const float* out,
                     const float* lse, const int32_t* ptr, const int32_t* tile_graph,
                     const int32_t* tile_row0, int64_t max_tiles, int64_t N, int H, int dh,
                     float scale, float p_drop, uint64_t seed, float* delta, float* d_qkv,
                     int64_t ld_dqkv, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && H > 0 && dh > 0 && max_tiles >= 0 && ld_qkv >= 3LL * H * dh &&
                  ld_dqkv >= 3LL * H * dh,
              "gps_seg_attn_bwd: bad sizes");
  GPS_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "gps_seg_attn_bwd: p_drop=%f outside [0,1)", p_drop);
  GPS_REQUIRE(N * (int64_t)H < INT32_MAX, "gps_seg_attn_bwd: N*H exceeds the 32-bit dropout row id");
  if (N == 0 || max_tiles == 0) return GPS_OK;
  GPS_REQUIRE(d_out && qkv && out && lse && ptr && tile_graph && tile_row0 && delta && d_qkv,
              "gps_seg_attn_bwd: null buffer");
  if (!gps_attn_supported_head_dim(dh)) {
    gps::set_error("gps_seg_attn_bwd: head dim %d has no compiled kernel", dh);
    return GPS_EUNSUPPORTED;
  }
  const int64_t n_work = max_tiles * H;
  const unsigned grid = gps::grid_for(n_work, 4);
  hipStream_t s = gps::as_stream(stream);
  switch (dh) {
#define X(D)                                                                                       \
  case D:                                                                                          \
    if (p_drop > 0.0f) {                                                                           \
      k_attn_bwd_dq<D, true><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, out, lse, delta, ptr, tile_graph, \
                                                  tile_row0, n_work, N, H, scale, p_drop, seed, gps::dropout_salt(),    \
                                                  d_qkv, ld_dqkv);                                 \
      k_attn_bwd_dkv<D, true><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, lse, delta, ptr,            \
                                                   tile_graph, tile_row0, n_work, N, H, scale,     \
                                                   p_drop, seed, gps::dropout_salt(), d_qkv, ld_dqkv);                  \
    } else {                                                                                       \
      k_attn_bwd_dq<D, false><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, out, lse, delta, ptr,       \
                                                   tile_graph, tile_row0, n_work, N, H, scale,     \
                                                   p_drop, seed, gps::dropout_salt(), d_qkv, ld_dqkv);                  \
      k_attn_bwd_dkv<D, false><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, lse, delta, ptr,           \
                                                    tile_graph, tile_row0, n_work, N, H, scale,    \
                                                    p_drop, seed, gps::dropout_salt(), d_qkv, ld_dqkv);                 \
    }                                                                                              \
    break;
    GPS_FOR_EACH_DH(X)
#undef X
  }
  return gps::launch_status("gps_seg_attn_bwd");
}

}  // extern "C"
