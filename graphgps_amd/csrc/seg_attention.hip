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
#include "attn_common.hpp"
#include "gps_common.hpp"

#ifndef GPS_ATTN_KT
#define GPS_ATTN_KT 2   // 16-key tiles per block (forward, dQ): registers scale with it
#endif
#ifndef GPS_ATTN_QT
#define GPS_ATTN_QT 1   // 16-query tiles per block (dK/dV)
#endif

namespace {

using namespace attn;

template <int DH>
struct Geo {
  static constexpr int KPL = (DH + 3) / 4;    // contraction elements per lane group
  static constexpr int DT = (DH + 15) / 16;   // 16-wide output tiles along dh
  static constexpr bool EXACT = (KPL * 4 == DH);
};

// Addressing discipline (the kernels were VALU-bound on 64-bit index arithmetic: ~775 VALU
// instructions per wave around 56 MFMAs): every tensor is reached through a WAVE-UNIFORM base
// pointer (graph's first row, head's first column -> scalar registers) plus a 32-bit per-lane element
// offset `local_row * ld + col` (one v_mad_u32_u24).  Rows are clamped to the last row of the BUFFER
// (never read out of bounds) and rows past the end of the GRAPH are zeroed by a select, so no
// multiplication by zero ever has to cancel foreign (possibly non-finite) data.
__device__ __forceinline__ uint32_t row_off(uint32_t lrow, uint32_t rmax, uint32_t ld) {
  return __umul24(min(lrow, rmax), ld);
}

// The KPL contiguous floats lane group `grp` contributes to a dh-contraction, from `base[off ...]`
// (off already includes grp * KPL).  VEC: 8-byte loads (host guarantees alignment).
template <int DH, bool VEC>
__device__ __forceinline__ void load_slice(const float* __restrict__ base, uint32_t off, bool ok, int grp,
                                           float scale, float (&dst)[Geo<DH>::KPL]) {
  constexpr int KPL = Geo<DH>::KPL;
  if constexpr (VEC && Geo<DH>::EXACT && (KPL % 2 == 0)) {
    const float2* p = reinterpret_cast<const float2*>(base + off);
#pragma unroll
    for (int s = 0; s < KPL / 2; ++s) {
      const float2 v = p[s];
      dst[2 * s] = ok ? v.x * scale : 0.0f;
      dst[2 * s + 1] = ok ? v.y * scale : 0.0f;
    }
  } else {
    const float* p = base + off;
#pragma unroll
    for (int s = 0; s < KPL; ++s) {
      const bool in = Geo<DH>::EXACT || (grp * KPL + s < DH);
      const float v = in ? p[s] : 0.0f;
      dst[s] = (ok && in) ? v * scale : 0.0f;
    }
  }
}

// 4 consecutive output floats of one lane (columns col..col+3 of row `off`)
template <int DH, bool VEC>
__device__ __forceinline__ void store4(float* __restrict__ base, uint32_t off, int col, f32x4 v) {
  if constexpr (VEC && (DH % 4 == 0)) {
    if (col < DH) *reinterpret_cast<float4*>(base + off) = make_float4(v[0], v[1], v[2], v[3]);
  } else {
#pragma unroll
    for (int r = 0; r < 4; ++r)
      if (col + r < DH) base[off + r] = v[r];
  }
}

// What every (16-row tile, head) wave needs: graph extent, head, uniform base pointers.
struct Wave {
  int n0, n, h, l0, g;       // graph's first row, size, head, tile's first LOCAL row, graph id
  uint32_t rmax;             // last local row that is still inside the buffer
  bool live;
};
__device__ __forceinline__ Wave wave_setup(const int32_t* __restrict__ ptr,
                                           const int32_t* __restrict__ tile_graph,
                                           const int32_t* __restrict__ tile_row0, int64_t n_work,
                                           int64_t N, int H) {
  Wave w;
  w.live = false;
  const int64_t wi = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wi >= n_work) return w;
  const int64_t tile = wi / H;
  w.h = (int)(wi - tile * H);
  const int g = tile_graph[tile];
  if (g < 0) return w;
  w.g = g;
  w.n0 = ptr[g];
  w.n = ptr[g + 1] - w.n0;
  w.l0 = tile_row0[tile] - w.n0;
  w.rmax = (uint32_t)(N - 1 - w.n0);
  w.live = true;
  return w;
}

// Additive attention bias (the reference's `attn_mask=batch.attn_bias`, gps_layer.py:201-203 and
// graphormer_layer.py:43-44): a dense fp32 [B*H, nmax, nmax] tensor in the layout the reference's
// BiasEncoder emits (graphormer_encoder.py:148-183), element [(g*H + h), query, key] added to the
// scaled scores before the softmax.  The kernels read only the n x n corner of graph g and write the
// same corner of its gradient (dS); the padded remainder is never touched (the host zero-fills it).
struct Bias {
  const float* __restrict__ b;   // wave-uniform: first element of this (graph, head) plane
  float* __restrict__ g;         // same plane of the gradient (dQ kernel only), or nullptr
  uint32_t nmax, nlast;          // plane pitch; last local row of the graph (clamp for padding lanes)
};
__device__ __forceinline__ Bias bias_setup(const float* bias, float* g_bias, int64_t nmax, const Wave& w,
                                           int H) {
  Bias b;
  const int64_t plane = ((int64_t)w.g * H + w.h) * nmax * nmax;
  b.b = bias ? bias + plane : nullptr;
  b.g = g_bias ? g_bias + plane : nullptr;
  b.nmax = (uint32_t)nmax;
  b.nlast = (uint32_t)max(min(w.n, (int)nmax) - 1, 0);
  return b;
}

// =============================================================================================
// forward
// =============================================================================================
// One 64-key block with NT (1..4) live 16-key tiles.  Branch-free and fully unrolled on purpose:
// every K and V operand load of the block is issued before the first MFMA, so a wave pays ONE
// memory round trip per block instead of one per tile (the kernel is latency-bound at molecule
// sizes: ~30 x 30 x 24 per (graph, head)).
template <int DH, bool DROP, bool VEC, bool BIAS, int NT>
__device__ __forceinline__ void attn_fwd_block(
    const float* __restrict__ Kb, const float* __restrict__ Vb, uint32_t ld, int kb, const Wave& w, int i,
    int grp, const float (&qv)[Geo<DH>::KPL], uint32_t rh, uint32_t thr16, float inv_keep, const Bias& bs,
    uint32_t bq_off, float& m, float& lsum, f32x4 (&oacc)[Geo<DH>::DT]) {
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT;
  float kv[NT][KPL];
  float vv[DT][NT][4];
  float bv[NT][4];
  if constexpr (BIAS) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bv[t][r] = bs.b[bq_off + min((uint32_t)(kb + 16 * t + 4 * grp + r), bs.nlast)];
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kr = kb + 16 * t + i;
    load_slice<DH, VEC>(Kb, row_off(kr, w.rmax, ld) + grp * KPL, kr < w.n, grp, 1.0f, kv[t]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 16 * t + 4 * grp + r;
      const uint32_t off = row_off(key, w.rmax, ld) + i;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bool in = dt * 16 + i < DH;
        const float v = in ? Vb[off + dt * 16] : 0.0f;
        vv[dt][t][r] = (in && key < w.n) ? v : 0.0f;
      }
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
      if constexpr (BIAS) s[t][r] += bv[t][r];
      s[t][r] = key < w.n ? s[t][r] : -INFINITY;
      mloc = fmaxf(mloc, s[t][r]);
    }
  const float mnew = fmaxf(m, group_max(mloc));
  const float alpha = sm_exp(m - mnew);  // m = -inf on the first block -> 0
  m = mnew;
  float psum = 0.0f;
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      float p = sm_exp(s[t][r] - mnew);  // masked keys: exp(-inf) = 0
      psum += p;
      if (DROP) {
        const uint32_t key_local = (uint32_t)(kb + 16 * t + 4 * grp + r);
        p = keep_elem(rh, key_local, thr16) ? p * inv_keep : 0.0f;
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

template <int DH, bool DROP, bool VEC, bool BIAS>
__global__ __launch_bounds__(256) void k_attn_fwd(
    const float* __restrict__ qkv, int64_t ld64, const int32_t* __restrict__ ptr,
    const int32_t* __restrict__ tile_graph, const int32_t* __restrict__ tile_row0,
    int64_t n_work, int64_t N, int H, float scale, float p_drop, uint64_t seed, const uint64_t* __restrict__ salt,
    const float* __restrict__ bias, int64_t nmax, float* __restrict__ out, float* __restrict__ lse) {
  const Wave w = wave_setup(ptr, tile_graph, tile_row0, n_work, N, H);
  if (!w.live) return;
  seed = gps::salted_seed(seed, salt);
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT, KT = GPS_ATTN_KT;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, grp = lane >> 4;
  const int d = H * DH;
  const uint32_t ld = (uint32_t)ld64;
  const float* __restrict__ Qb = qkv + (int64_t)w.n0 * ld64 + w.h * DH;   // wave-uniform bases
  const float* __restrict__ Kb = Qb + d;
  const float* __restrict__ Vb = Qb + 2 * d;
  const int ql = w.l0 + i;                  // local query row
  const bool q_ok = ql < w.n;

  float qv[KPL];
  load_slice<DH, VEC>(Qb, row_off(ql, w.rmax, ld) + grp * KPL, q_ok, grp, scale, qv);
  const uint32_t rh = DROP ? row_hash((uint32_t)(w.n0 + ql) * (uint32_t)H + (uint32_t)w.h, seed) : 0u;
  const uint32_t thr16 = drop_thr16(p_drop);
  const float inv_keep = DROP ? drop_inv_keep(thr16) : 1.0f;
  const Bias bs = BIAS ? bias_setup(bias, nullptr, nmax, w, H) : Bias{nullptr, nullptr, 0u, 0u};
  const uint32_t bq_off = BIAS ? min((uint32_t)ql, bs.nlast) * bs.nmax : 0u;

  float m = -INFINITY, lsum = 0.0f;
  f32x4 oacc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kb = 0; kb < w.n; kb += 16 * KT) {
    const int nt = min(KT, (w.n - kb + 15) >> 4);  // wave-uniform
    switch (nt) {
      case 1: attn_fwd_block<DH, DROP, VEC, BIAS, 1>(Kb, Vb, ld, kb, w, i, grp, qv, rh, thr16, inv_keep, bs, bq_off, m, lsum, oacc); break;
      case 2: attn_fwd_block<DH, DROP, VEC, BIAS, (KT >= 2 ? 2 : KT)>(Kb, Vb, ld, kb, w, i, grp, qv, rh, thr16, inv_keep, bs, bq_off, m, lsum, oacc); break;
      case 3: attn_fwd_block<DH, DROP, VEC, BIAS, (KT >= 3 ? 3 : KT)>(Kb, Vb, ld, kb, w, i, grp, qv, rh, thr16, inv_keep, bs, bq_off, m, lsum, oacc); break;
      default: attn_fwd_block<DH, DROP, VEC, BIAS, KT>(Kb, Vb, ld, kb, w, i, grp, qv, rh, thr16, inv_keep, bs, bq_off, m, lsum, oacc); break;
    }
  }
  const float ltot = group_sum(lsum);
  const float inv_l = 1.0f / ltot;
  if (q_ok) {
    float* __restrict__ Ob = out + (int64_t)w.n0 * d + w.h * DH;
    const uint32_t orow = (uint32_t)ql * (uint32_t)d;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int col = dt * 16 + 4 * grp;
      store4<DH, VEC>(Ob, orow + col, col, oacc[dt] * inv_l);
    }
    if (grp == 0) lse[(int64_t)w.h * N + w.n0 + ql] = m + logf(ltot);
  }
}

// =============================================================================================
// backward, query-tile keyed: dQ (+ delta)
// =============================================================================================
// Same discipline as the forward block: one 64-key block with NT live tiles, branch-free, every K / V
// operand (row slices for S^T and dP^T, column form of K for dQ^T) loaded before the first MFMA.
template <int DH, bool DROP, bool VEC, bool BIAS, int NT>
__device__ __forceinline__ void attn_dq_block(
    const float* __restrict__ Kb, const float* __restrict__ Vb, uint32_t ld, int kb, const Wave& w, int i,
    int grp, const float (&qv)[Geo<DH>::KPL], const float (&dov)[Geo<DH>::KPL], float lse_q, float dl_q,
    uint32_t rh, uint32_t thr16, float inv_keep, const Bias& bs, uint32_t bq_off, bool q_ok,
    f32x4 (&acc)[Geo<DH>::DT]) {
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT;
  float kv[NT][KPL], vv[NT][KPL];
  float kc[DT][NT][4];
  float bv[NT][4];
  if constexpr (BIAS) {
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bv[t][r] = bs.b[bq_off + min((uint32_t)(kb + 16 * t + 4 * grp + r), bs.nlast)];
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int kr = kb + 16 * t + i;
    const uint32_t off = row_off(kr, w.rmax, ld) + grp * KPL;
    load_slice<DH, VEC>(Kb, off, kr < w.n, grp, 1.0f, kv[t]);
    load_slice<DH, VEC>(Vb, off, kr < w.n, grp, 1.0f, vv[t]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 16 * t + 4 * grp + r;
      const uint32_t off = row_off(key, w.rmax, ld) + i;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bool in = dt * 16 + i < DH;
        const float v = in ? Kb[off + dt * 16] : 0.0f;
        kc[dt][t][r] = (in && key < w.n) ? v : 0.0f;
      }
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
      if constexpr (BIAS) s[t][r] += bv[t][r];
      const float p = sm_exp(s[t][r] - lse_q);
      float dpe = dp[t][r];
      if (DROP) dpe = keep_elem(rh, (uint32_t)key, thr16) ? dpe * inv_keep : 0.0f;
      s[t][r] = key < w.n ? p * (dpe - dl_q) : 0.0f;   // dS^T, reused as the B operand below
      if constexpr (BIAS) {                            // d(bias)[query][key] = dS[query][key]
        if (q_ok && key < w.n) bs.g[bq_off + (uint32_t)key] = s[t][r];
      }
    }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int dt = 0; dt < DT; ++dt)            // dQ^T[dh][query] += K^T[dh][key] dS^T[key][query]
        acc[dt] = mfma16(kc[dt][t][r], s[t][r], acc[dt]);
}

template <int DH, bool DROP, bool VEC, bool BIAS>
__global__ __launch_bounds__(256) void k_attn_bwd_dq(
    const float* __restrict__ d_out, const float* __restrict__ qkv, int64_t ld64,
    const float* __restrict__ out, const float* __restrict__ lse, float* __restrict__ delta,
    const int32_t* __restrict__ ptr, const int32_t* __restrict__ tile_graph,
    const int32_t* __restrict__ tile_row0, int64_t n_work, int64_t N, int H, float scale,
    float p_drop, uint64_t seed, const uint64_t* __restrict__ salt, const float* __restrict__ bias,
    float* __restrict__ g_bias, int64_t nmax, float* __restrict__ d_qkv, int64_t ldg64) {
  const Wave w = wave_setup(ptr, tile_graph, tile_row0, n_work, N, H);
  if (!w.live) return;
  seed = gps::salted_seed(seed, salt);
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT, KT = GPS_ATTN_KT;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, grp = lane >> 4;
  const int d = H * DH;
  const uint32_t ld = (uint32_t)ld64;
  const float* __restrict__ Qb = qkv + (int64_t)w.n0 * ld64 + w.h * DH;
  const float* __restrict__ Kb = Qb + d;
  const float* __restrict__ Vb = Qb + 2 * d;
  const float* __restrict__ dOb = d_out + (int64_t)w.n0 * d + w.h * DH;
  const float* __restrict__ Ob = out + (int64_t)w.n0 * d + w.h * DH;
  const int ql = w.l0 + i;
  const bool q_ok = ql < w.n;

  float qv[KPL], dov[KPL], ov[KPL];
  load_slice<DH, VEC>(Qb, row_off(ql, w.rmax, ld) + grp * KPL, q_ok, grp, scale, qv);
  const uint32_t orow = row_off(ql, w.rmax, (uint32_t)d) + grp * KPL;
  load_slice<DH, VEC>(dOb, orow, q_ok, grp, 1.0f, dov);
  load_slice<DH, VEC>(Ob, orow, q_ok, grp, 1.0f, ov);
  const int64_t sidx = (int64_t)w.h * N + w.n0 + min((uint32_t)ql, w.rmax);
  const float lse_q = lse[sidx];
  // delta[h][q] = sum_c dO[q][h*DH+c] * O[q][h*DH+c]  (KPL of the dh products per lane group)
  float dl_part = 0.0f;
#pragma unroll
  for (int c = 0; c < KPL; ++c) dl_part += dov[c] * ov[c];
  const float dl_q = group_sum(dl_part);
  if (q_ok && grp == 0) delta[sidx] = dl_q;   // read by k_attn_bwd_dkv (same stream)
  const uint32_t rh = DROP ? row_hash((uint32_t)(w.n0 + ql) * (uint32_t)H + (uint32_t)w.h, seed) : 0u;
  const uint32_t thr16 = drop_thr16(p_drop);
  const float inv_keep = DROP ? drop_inv_keep(thr16) : 1.0f;
  const Bias bs = BIAS ? bias_setup(bias, g_bias, nmax, w, H) : Bias{nullptr, nullptr, 0u, 0u};
  const uint32_t bq_off = BIAS ? min((uint32_t)ql, bs.nlast) * bs.nmax : 0u;

  f32x4 acc[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};

  for (int kb = 0; kb < w.n; kb += 16 * KT) {
    const int nt = min(KT, (w.n - kb + 15) >> 4);  // wave-uniform
    switch (nt) {
      case 1: attn_dq_block<DH, DROP, VEC, BIAS, 1>(Kb, Vb, ld, kb, w, i, grp, qv, dov, lse_q, dl_q, rh, thr16, inv_keep, bs, bq_off, q_ok, acc); break;
      case 2: attn_dq_block<DH, DROP, VEC, BIAS, (KT >= 2 ? 2 : KT)>(Kb, Vb, ld, kb, w, i, grp, qv, dov, lse_q, dl_q, rh, thr16, inv_keep, bs, bq_off, q_ok, acc); break;
      case 3: attn_dq_block<DH, DROP, VEC, BIAS, (KT >= 3 ? 3 : KT)>(Kb, Vb, ld, kb, w, i, grp, qv, dov, lse_q, dl_q, rh, thr16, inv_keep, bs, bq_off, q_ok, acc); break;
      default: attn_dq_block<DH, DROP, VEC, BIAS, KT>(Kb, Vb, ld, kb, w, i, grp, qv, dov, lse_q, dl_q, rh, thr16, inv_keep, bs, bq_off, q_ok, acc); break;
    }
  }
  if (q_ok) {
    float* __restrict__ Gq = d_qkv + (int64_t)w.n0 * ldg64 + w.h * DH;
    const uint32_t grow = (uint32_t)ql * (uint32_t)ldg64;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int col = dt * 16 + 4 * grp;
      store4<DH, VEC>(Gq, grow + col, col, acc[dt] * scale);
    }
  }
}

// =============================================================================================
// backward, key-tile keyed: dK, dV
// =============================================================================================
// One 32-query block with NT (1..2) live tiles, all Q / dO operands (row slices for S and dP, column
// forms for dK^T and dV^T) and the per-query lse / delta loaded before the first MFMA.
template <int DH, bool DROP, bool VEC, bool BIAS, int NT>
__device__ __forceinline__ void attn_dkv_block(
    const float* __restrict__ Qb, const float* __restrict__ dOb, const float* __restrict__ lse_b,
    const float* __restrict__ delta_b, uint32_t ld, uint32_t d, int H, int qb, const Wave& w, int i, int grp,
    int kl, const float (&kv)[Geo<DH>::KPL], const float (&vv)[Geo<DH>::KPL], float scale, uint64_t seed,
    uint32_t thr16, float inv_keep, const Bias& bs, f32x4 (&dk)[Geo<DH>::DT], f32x4 (&dv)[Geo<DH>::DT]) {
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT;
  float qa[NT][KPL], da[NT][KPL];
  float qc[DT][NT][4], dc[DT][NT][4];
  float lq[NT][4], dq[NT][4];
  float bv[NT][4];
  if constexpr (BIAS) {
    const uint32_t bk = min((uint32_t)kl, bs.nlast);
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r)
        bv[t][r] = bs.b[min((uint32_t)(qb + 16 * t + 4 * grp + r), bs.nlast) * bs.nmax + bk];
  }
#pragma unroll
  for (int t = 0; t < NT; ++t) {
    const int qr = qb + 16 * t + i;
    load_slice<DH, VEC>(Qb, row_off(qr, w.rmax, ld) + grp * KPL, qr < w.n, grp, scale, qa[t]);
    load_slice<DH, VEC>(dOb, row_off(qr, w.rmax, d) + grp * KPL, qr < w.n, grp, 1.0f, da[t]);
  }
#pragma unroll
  for (int t = 0; t < NT; ++t)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qb + 16 * t + 4 * grp + r;
      const bool q_in = qq < w.n;
      const uint32_t qcl = min((uint32_t)qq, w.rmax);
      lq[t][r] = lse_b[qcl];
      dq[t][r] = delta_b[qcl];
      const uint32_t offq = __umul24(qcl, ld) + i, offd = __umul24(qcl, d) + i;
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) {
        const bool in = dt * 16 + i < DH;
        const float vq = in ? Qb[offq + dt * 16] : 0.0f;
        const float vd = in ? dOb[offd + dt * 16] : 0.0f;
        qc[dt][t][r] = (in && q_in) ? vq * scale : 0.0f;
        dc[dt][t][r] = (in && q_in) ? vd : 0.0f;
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
      const bool q_in = qq < w.n;
      if constexpr (BIAS) s[t][r] += bv[t][r];
      const float pr = q_in ? sm_exp(s[t][r] - lq[t][r]) : 0.0f;
      float dpe = dp[t][r];
      float pd = pr;
      if (DROP) {
        const uint32_t rh = row_hash((uint32_t)(w.n0 + qq) * (uint32_t)H + (uint32_t)w.h, seed);
        const bool keep = keep_elem(rh, (uint32_t)kl, thr16);
        pd = keep ? pr * inv_keep : 0.0f;
        dpe = keep ? dpe * inv_keep : 0.0f;
      }
      s[t][r] = pd;                                        // P_drop
      dp[t][r] = q_in ? pr * (dpe - dq[t][r]) : 0.0f;      // dS
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

template <int DH, bool DROP, bool VEC, bool BIAS>
__global__ __launch_bounds__(256) void k_attn_bwd_dkv(
    const float* __restrict__ d_out, const float* __restrict__ qkv, int64_t ld64,
    const float* __restrict__ lse, const float* __restrict__ delta,
    const int32_t* __restrict__ ptr, const int32_t* __restrict__ tile_graph,
    const int32_t* __restrict__ tile_row0, int64_t n_work, int64_t N, int H, float scale,
    float p_drop, uint64_t seed, const uint64_t* __restrict__ salt, const float* __restrict__ bias,
    int64_t nmax, float* __restrict__ d_qkv, int64_t ldg64) {
  const Wave w = wave_setup(ptr, tile_graph, tile_row0, n_work, N, H);
  if (!w.live) return;
  seed = gps::salted_seed(seed, salt);
  constexpr int KPL = Geo<DH>::KPL, DT = Geo<DH>::DT, QT = GPS_ATTN_QT;
  const int lane = threadIdx.x & 63;
  const int i = lane & 15, grp = lane >> 4;
  const int d = H * DH;
  const uint32_t ld = (uint32_t)ld64;
  const float* __restrict__ Qb = qkv + (int64_t)w.n0 * ld64 + w.h * DH;
  const float* __restrict__ Kb = Qb + d;
  const float* __restrict__ Vb = Qb + 2 * d;
  const float* __restrict__ dOb = d_out + (int64_t)w.n0 * d + w.h * DH;
  const float* __restrict__ lse_b = lse + (int64_t)w.h * N + w.n0;
  const float* __restrict__ delta_b = delta + (int64_t)w.h * N + w.n0;
  const int kl = w.l0 + i;                  // local key row
  const bool k_ok = kl < w.n;
  const uint32_t thr16 = drop_thr16(p_drop);
  const float inv_keep = DROP ? drop_inv_keep(thr16) : 1.0f;
  const Bias bs = BIAS ? bias_setup(bias, nullptr, nmax, w, H) : Bias{nullptr, nullptr, 0u, 0u};

  float kv[KPL], vv[KPL];
  const uint32_t koff = row_off(kl, w.rmax, ld) + grp * KPL;
  load_slice<DH, VEC>(Kb, koff, k_ok, grp, 1.0f, kv);
  load_slice<DH, VEC>(Vb, koff, k_ok, grp, 1.0f, vv);

  f32x4 dk[DT], dv[DT];
#pragma unroll
  for (int dt = 0; dt < DT; ++dt) {
    dk[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    dv[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
  }

  for (int qb = 0; qb < w.n; qb += 16 * QT) {
    if (QT > 1 && w.n - qb > 16)
      attn_dkv_block<DH, DROP, VEC, BIAS, QT>(Qb, dOb, lse_b, delta_b, ld, (uint32_t)d, H, qb, w, i, grp, kl, kv,
                                              vv, scale, seed, thr16, inv_keep, bs, dk, dv);
    else
      attn_dkv_block<DH, DROP, VEC, BIAS, 1>(Qb, dOb, lse_b, delta_b, ld, (uint32_t)d, H, qb, w, i, grp, kl, kv,
                                             vv, scale, seed, thr16, inv_keep, bs, dk, dv);
  }
  if (k_ok) {
    float* __restrict__ Gk = d_qkv + (int64_t)w.n0 * ldg64 + d + w.h * DH;
    float* __restrict__ Gv = Gk + d;
    const uint32_t grow = (uint32_t)kl * (uint32_t)ldg64;
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      const int col = dt * 16 + 4 * grp;
      store4<DH, VEC>(Gk, grow + col, col, dk[dt]);
      store4<DH, VEC>(Gv, grow + col, col, dv[dt]);
    }
  }
}

#define GPS_FOR_EACH_DH(X) \
  X(4) X(6) X(8) X(10) X(12) X(13) X(16) X(18) X(20) X(24) X(32) X(48) X(64) X(76) X(96) X(128)

}  // namespace

static inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

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

// Shared launcher of the plain and the biased forward (bias == nullptr: plain).
static int seg_attn_fwd_impl(const char* who, const float* qkv, int64_t ld_qkv, const float* bias,
                             int64_t nmax, const int32_t* ptr, const int32_t* tile_graph,
                             const int32_t* tile_row0, int64_t max_tiles, int64_t N, int H, int dh, float scale,
                             float p_drop, uint64_t seed, float* out, float* lse, int64_t num_graphs,
                             int64_t max_graph_nodes, uint32_t* amax, const int32_t* order, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && H > 0 && dh > 0 && max_tiles >= 0 && ld_qkv >= 3LL * H * dh,
              "%s: bad sizes N=%lld H=%d dh=%d ld=%lld", who, (long long)N, H, dh, (long long)ld_qkv);
  GPS_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "%s: p_drop=%f outside [0,1)", who, p_drop);
  GPS_REQUIRE(N * (int64_t)H < INT32_MAX, "%s: N*H exceeds the 32-bit dropout row id", who);
  if (N == 0 || max_tiles == 0) return GPS_OK;
  GPS_REQUIRE(qkv && ptr && tile_graph && tile_row0 && out && lse, "%s: null buffer", who);
  if (!gps_attn_supported_head_dim(dh)) {
    gps::set_error("%s: head dim %d has no compiled kernel", who, dh);
    return GPS_EUNSUPPORTED;
  }
  GPS_REQUIRE(N * ld_qkv < (int64_t(1) << 31), "%s: N * ld exceeds 32-bit element offsets", who);
  if (bias) GPS_REQUIRE(nmax > 0 && nmax * nmax < (int64_t(1) << 31), "%s: bad bias pitch nmax=%lld", who,
                        (long long)nmax);
  const int64_t n_work = max_tiles * H;
  const unsigned grid = gps::grid_for(n_work, 4);
  hipStream_t s = gps::as_stream(stream);
  if (!bias && num_graphs > 0 && max_graph_nodes > 0 && max_graph_nodes <= 64 &&
      attn::sattn_applicable(qkv, ld_qkv, out, H, dh)) {            // block form (sattn.hip)
    attn::sattn_fwd_launch(qkv, ld_qkv, ptr, num_graphs, N, H, dh, scale, p_drop, seed, out, lse, amax, order, s);
    return gps::launch_status(who);
  }
  const bool vec = ld_qkv % 4 == 0 && (H * dh) % 4 == 0 && al16(qkv) && al16(out);
#define LAUNCH_FWD(D, DROP, VEC, BIAS)                                                                  \
  k_attn_fwd<D, DROP, VEC, BIAS><<<grid, 256, 0, s>>>(qkv, ld_qkv, ptr, tile_graph, tile_row0, n_work, N, H, \
                                                      scale, p_drop, seed, gps::dropout_salt(), bias, nmax,  \
                                                      out, lse)
  switch (dh) {
#define X(D)                                                                              \
  case D:                                                                                 \
    if (bias) {                                                                           \
      if (p_drop > 0.0f) LAUNCH_FWD(D, true, false, true); else LAUNCH_FWD(D, false, false, true); \
    } else if (p_drop > 0.0f) {                                                           \
      if (vec) LAUNCH_FWD(D, true, true, false); else LAUNCH_FWD(D, true, false, false);   \
    } else {                                                                              \
      if (vec) LAUNCH_FWD(D, false, true, false); else LAUNCH_FWD(D, false, false, false); \
    }                                                                                     \
    break;
    GPS_FOR_EACH_DH(X)
#undef X
  }
#undef LAUNCH_FWD
  if (int rc = gps::launch_status(who)) return rc;
  if (amax) {        // the tile kernels do not track the maximum: one pre-pass over the output (gps_absmax)
    const gps_absmax_desc dsc{out, (int64_t)H * dh, N, H * dh, amax};
    return gps_absmax(1, &dsc, stream);
  }
  return GPS_OK;
}

static int seg_attn_bwd_impl(const char* who, const float* d_out, const float* qkv, int64_t ld_qkv,
                             const float* bias, int64_t nmax, const float* out, const float* lse,
                             const int32_t* ptr, const int32_t* tile_graph, const int32_t* tile_row0,
                             int64_t max_tiles, int64_t N, int H, int dh, float scale, float p_drop,
                             uint64_t seed, float* delta, float* d_qkv, int64_t ld_dqkv, float* d_bias,
                             int64_t num_graphs, int64_t max_graph_nodes, uint32_t* amax, const int32_t* order,
                             gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && H > 0 && dh > 0 && max_tiles >= 0 && ld_qkv >= 3LL * H * dh &&
                  ld_dqkv >= 3LL * H * dh,
              "%s: bad sizes", who);
  GPS_REQUIRE(p_drop >= 0.0f && p_drop < 1.0f, "%s: p_drop=%f outside [0,1)", who, p_drop);
  GPS_REQUIRE(N * (int64_t)H < INT32_MAX, "%s: N*H exceeds the 32-bit dropout row id", who);
  if (N == 0 || max_tiles == 0) return GPS_OK;
  GPS_REQUIRE(d_out && qkv && out && lse && ptr && tile_graph && tile_row0 && delta && d_qkv,
              "%s: null buffer", who);
  if (!gps_attn_supported_head_dim(dh)) {
    gps::set_error("%s: head dim %d has no compiled kernel", who, dh);
    return GPS_EUNSUPPORTED;
  }
  GPS_REQUIRE(N * ld_qkv < (int64_t(1) << 31) && N * ld_dqkv < (int64_t(1) << 31),
              "%s: N * ld exceeds 32-bit element offsets", who);
  if (bias) GPS_REQUIRE(d_bias && nmax > 0 && nmax * nmax < (int64_t(1) << 31),
                        "%s: bias needs its gradient buffer and a pitch < 46341 (nmax=%lld)", who,
                        (long long)nmax);
  const int64_t n_work = max_tiles * H;
  const unsigned grid = gps::grid_for(n_work, 4);
  hipStream_t s = gps::as_stream(stream);
  if (!bias && num_graphs > 0 && max_graph_nodes > 0 && max_graph_nodes <= 64 &&
      attn::sattn_applicable(qkv, ld_qkv, out, H, dh) && ld_dqkv % 4 == 0 && al16(d_out) && al16(d_qkv)) {
    attn::sattn_bwd_launch(d_out, qkv, ld_qkv, out, lse, ptr, num_graphs, N, H, dh, scale, p_drop, seed, d_qkv,
                           ld_dqkv, amax, order, s);             // one fused launch (sattn.hip)
    return gps::launch_status(who);
  }
  const bool vec = ld_qkv % 4 == 0 && ld_dqkv % 4 == 0 && (H * dh) % 4 == 0 && al16(qkv) && al16(out) &&
                   al16(d_out) && al16(d_qkv);
#define LAUNCH_BWD(D, DROP, VEC, BIAS)                                                                     \
  do {                                                                                                     \
    k_attn_bwd_dq<D, DROP, VEC, BIAS><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, out, lse, delta, ptr,       \
                                                           tile_graph, tile_row0, n_work, N, H, scale,     \
                                                           p_drop, seed, gps::dropout_salt(), bias, d_bias, \
                                                           nmax, d_qkv, ld_dqkv);                          \
    k_attn_bwd_dkv<D, DROP, VEC, BIAS><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, lse, delta, ptr, tile_graph, \
                                                            tile_row0, n_work, N, H, scale, p_drop, seed,  \
                                                            gps::dropout_salt(), bias, nmax, d_qkv,        \
                                                            ld_dqkv);                                      \
  } while (0)
  switch (dh) {
#define X(D)                                                                              \
  case D:                                                                                 \
    if (bias) {                                                                           \
      if (p_drop > 0.0f) LAUNCH_BWD(D, true, false, true); else LAUNCH_BWD(D, false, false, true); \
    } else if (p_drop > 0.0f) {                                                           \
      if (vec) LAUNCH_BWD(D, true, true, false); else LAUNCH_BWD(D, true, false, false);   \
    } else {                                                                              \
      if (vec) LAUNCH_BWD(D, false, true, false); else LAUNCH_BWD(D, false, false, false); \
    }                                                                                     \
    break;
    GPS_FOR_EACH_DH(X)
#undef X
  }
#undef LAUNCH_BWD
  if (int rc = gps::launch_status(who)) return rc;
  if (amax && (3LL * H * dh) % 4 == 0 && ld_dqkv % 4 == 0 && al16(d_qkv)) {
    const gps_absmax_desc dsc{d_qkv, ld_dqkv, N, 3 * H * dh, amax};
    return gps_absmax(1, &dsc, stream);
  }
  GPS_REQUIRE(!amax, "%s: the max|d_qkv| record needs 16-byte-aligned rows", who);
  return GPS_OK;
}

int gps_seg_attn_fwd(const float* qkv, int64_t ld_qkv, const int32_t* ptr,
                     const int32_t* tile_graph, const int32_t* tile_row0, int64_t max_tiles,
                     int64_t N, int H, int dh, float scale, float p_drop, uint64_t seed, float* out,
                     float* lse, int64_t num_graphs, int64_t max_graph_nodes, uint32_t* amax, const int32_t* graph_order,
                     gps_stream_t stream) {
  GPS_REQUIRE(!amax || ((H * dh) % 4 == 0 && al16(out)), "gps_seg_attn_fwd: the max|out| record needs 16-byte-aligned rows");
  return seg_attn_fwd_impl("gps_seg_attn_fwd", qkv, ld_qkv, nullptr, 0, ptr, tile_graph, tile_row0, max_tiles,
                           N, H, dh, scale, p_drop, seed, out, lse, num_graphs, max_graph_nodes, amax, graph_order, stream);
}

int gps_seg_attn_bwd(const float* d_out, const float* qkv, int64_t ld_qkv, const float* out,
                     const float* lse, const int32_t* ptr, const int32_t* tile_graph,
                     const int32_t* tile_row0, int64_t max_tiles, int64_t N, int H, int dh,
                     float scale, float p_drop, uint64_t seed, float* delta, float* d_qkv,
                     int64_t ld_dqkv, int64_t num_graphs, int64_t max_graph_nodes, uint32_t* amax,
                     const int32_t* graph_order, gps_stream_t stream) {
  return seg_attn_bwd_impl("gps_seg_attn_bwd", d_out, qkv, ld_qkv, nullptr, 0, out, lse, ptr, tile_graph,
                           tile_row0, max_tiles, N, H, dh, scale, p_drop, seed, delta, d_qkv, ld_dqkv, nullptr,
                           num_graphs, max_graph_nodes, amax, graph_order, stream);
}

int gps_seg_attn_bias_fwd(const float* qkv, int64_t ld_qkv, const float* bias, int64_t nmax,
                          const int32_t* ptr, const int32_t* tile_graph, const int32_t* tile_row0,
                          int64_t max_tiles, int64_t N, int H, int dh, float scale, float p_drop,
                          uint64_t seed, float* out, float* lse, gps_stream_t stream) {
  GPS_REQUIRE(bias != nullptr || N == 0, "gps_seg_attn_bias_fwd: null bias");
  return seg_attn_fwd_impl("gps_seg_attn_bias_fwd", qkv, ld_qkv, bias, nmax, ptr, tile_graph, tile_row0,
                           max_tiles, N, H, dh, scale, p_drop, seed, out, lse, 0, 0, nullptr, nullptr, stream);
}

int gps_seg_attn_bias_bwd(const float* d_out, const float* qkv, int64_t ld_qkv, const float* bias,
                          int64_t nmax, const float* out, const float* lse, const int32_t* ptr,
                          const int32_t* tile_graph, const int32_t* tile_row0, int64_t max_tiles, int64_t N,
                          int H, int dh, float scale, float p_drop, uint64_t seed, float* delta,
                          float* d_qkv, int64_t ld_dqkv, float* d_bias, gps_stream_t stream) {
  GPS_REQUIRE(bias != nullptr || N == 0, "gps_seg_attn_bias_bwd: null bias");
  return seg_attn_bwd_impl("gps_seg_attn_bias_bwd", d_out, qkv, ld_qkv, bias, nmax, out, lse, ptr, tile_graph,
                           tile_row0, max_tiles, N, H, dh, scale, p_drop, seed, delta, d_qkv, ld_dqkv, d_bias, 0, 0,
                           nullptr, nullptr, stream);
}

}  // extern "C"
