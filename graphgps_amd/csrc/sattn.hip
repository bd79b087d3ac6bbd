// Segment attention, block form: one wavefront per (graph, 64-query block, head), K-side operands resident in registers.
//
// Same arithmetic and lane algebra as seg_attention.hip (the reference's to_dense_batch -> nn.MultiheadAttention
// core -> un-pad path, graphgps/layer/gps_layer.py:199-201,234-241); what changes is how much work one wavefront
// owns.  At molecule sizes the core is latency- and VALU-issue-bound, not flop-bound (0.36 GFLOP against 46.5 MB per
// layer; ~775 VALU instructions around 56 MFMAs per wave in the per-(16-row tile, head) kernels, every query tile of
// a graph re-loading and re-addressing that graph's K / V).  Here a wavefront
//   * loads the K-side operands of a 64-key block ONCE -- row slices for S^T = K Q^T, column form of V for
//     O^T = V^T P^T -- with every load issued before the first use (one memory round trip), keeps them in registers
//     and serves ALL (up to 4) query tiles of its 64-query block from them: K / V loads, their addressing and the
//     masking selects are amortised over the query tiles, and a graph of <= 64 nodes reads each K / V element once
//     per head instead of once per (head, query tile);
//   * takes one dropout hash per two adjacent keys (attn_common.hpp);
//   * needs no LDS and no barrier in the forward, so every (graph, head) of a 256-molecule batch is resident at once
//     (4096 wavefronts, 16 per CU): nothing queues behind anything.  (A workgroup-per-graph variant that staged Q | K | V
//     rows through LDS -- fully coalesced, each line read exactly once -- was built and measured first: 43 us against
//     27 us for the round-1 kernel at P30.  76.8 KB of LDS per workgroup left 8 wavefronts per CU, the ~1900 tile
//     slots that start no 64-row block still had to wait for an LDS allocation to find that out, and the problem is
//     bound by latency and issue slots, not by the 1.5x re-read traffic the staging removed.)
// Both kernels cover batches whose longest graph has <= 64 nodes (the caller passes that bound, known to the host
// from `ptr` before the batch ever reaches the device); anything longer takes the per-tile kernels of
// seg_attention.hip, which walk the keys in blocks with the online softmax.
//
// Backward (k_sattn_bwd, graphs of <= 64 nodes): ONE launch per layer instead of two.  S^T, P and dS are computed once
// per (query tile, key tile) in the query-column orientation (what dQ^T = K^T dS^T needs), then P_drop and dS are
// transposed through a 2.5 KB per-wave LDS scratch into the key-column orientation that dV^T = dO^T P and
// dK^T = Q^T dS need -- the softmax / dropout VALU work is done once instead of twice.  delta = rowsum(dO o O) is
// formed on the fly.
#include <cstdlib>

#include "attn_common.hpp"

namespace {
using namespace attn;

// (graphs of up to 64 nodes = 4 key tiles x 4 query tiles per wavefront)

template <int DH>
struct SGeo {
  static constexpr int KPL = DH / 4;          // contraction elements per lane group
  static constexpr int DT = (DH + 15) / 16;   // 16-wide output tiles along dh
  static_assert(DH % 8 == 0, "row-slice loads are 8-byte");
};

// One atomic per wavefront on the tensor's max|.| record (gps_common.hpp: eight words on eight different lines, so the
// ~4,000 wavefronts of a launch queue ~500 deep per line, spread over the launch).  Every lane of the wave is live here.
__device__ __forceinline__ void wave_amax(float amx, uint32_t* __restrict__ rec) {
  if (!rec) return;                           // (kernel-uniform)
  uint32_t m = __float_as_uint(amx);
#pragma unroll
  for (int o = 32; o; o >>= 1) m = max(m, (uint32_t)__shfl_xor((int)m, o));
  if ((threadIdx.x & 63) == 0) gps::amax_raise(rec, m, blockIdx.x * 4u + (threadIdx.x >> 6));
}

// KPL contiguous floats of row `row` (< nrows, else zeros) for this lane group: 8-byte loads off a wave-uniform base
template <int KPL>
__device__ __forceinline__ void row_slice(const float* __restrict__ base, uint32_t ld, int row, int nrows, int col,
                                          float scale, float (&dst)[KPL]) {
  const bool ok = row < nrows;
  const float2* p = reinterpret_cast<const float2*>(base + __umul24((uint32_t)(ok ? row : 0), ld) + col);
#pragma unroll
  for (int c = 0; c < KPL / 2; ++c) {
    const float2 v = p[c];                 // unconditional (the address is clamped into the graph): a load under a
    dst[2 * c] = ok ? v.x * scale : 0.0f;  // condition becomes a branch, and the loads stop being issued together
    dst[2 * c + 1] = ok ? v.y * scale : 0.0f;
  }
}
// element [row][col] (zero outside the graph / the head)
__device__ __forceinline__ float col_elem(const float* __restrict__ base, uint32_t ld, int row, int nrows, int col,
                                          bool col_ok) {
  const bool ok = col_ok && row < nrows;
  const float v = base[__umul24((uint32_t)(ok ? row : 0), ld) + (ok ? col : 0)];
  return ok ? v : 0.0f;
}

// elements [row][col], [row][col + 1] in ONE 8-byte load (col even; zeros outside the graph / the head)
__device__ __forceinline__ float2 col_pair(const float* __restrict__ base, uint32_t ld, int row, int nrows, int col,
                                           bool col_ok) {
  const bool ok = col_ok && row < nrows;
  const float2 v = *reinterpret_cast<const float2*>(base + __umul24((uint32_t)(ok ? row : 0), ld) + (ok ? col : 0));
  return ok ? v : make_float2(0.0f, 0.0f);
}
// Round 5 -- column slots of the dh-side MFMA tiles when a head spans TWO 16-column tiles (dh = 24, 32).  Which head column a
// slot of an output tile stands for is free (the contraction runs over keys / queries, never over dh), so tile 0 takes the EVEN
// columns and tile 1 the ODD ones: slot i of tile dt is column 2 i + dt.  A lane's column-form operand elements of the two
// tiles are then ADJACENT in memory -- one 8-byte load instead of two 4-byte loads (Q and dO in the backward: 16 -> 8 load
// instructions per query tile; the forward's V loads likewise, measured slower there and left off) -- and the four slots a lane holds of each
// tile interleave into 8 consecutive columns, still two 16-byte stores.  Same sums in the same order: bit-identical
// results.  (Measured first, `k_sattn_fwd2`: prefetching a second item's operands per wavefront made the forward SLOWER,
// 29.0 vs 25.0 us -- its loads were sized for the batch's longest graph, 47 instead of ~28 load instructions per item -- so
// the kernel is bound by the NUMBER of vector-memory instructions, not by a load / compute lockstep: DESIGN section 4.4.)
template <int DT>
struct Slots {
  static constexpr bool PAIRED = DT == 2;
};

struct Item {
  int n0, n, h;
  bool live;
};
// wavefront -> (graph, head).  Indexed by GRAPH, not through the 16-row tile map: every wavefront launched is live
// (with the tile map two thirds of the slots start no block, and although such a wave exits at once, the live ones
// end up spread over three dispatch generations instead of one -- measured: 27 us against the ~8 us of one
// generation).
// `order` (or nullptr): slot -> graph, the balanced dispatch order of gps_attn_graph_order (csrc/graph_index.hip)
__device__ __forceinline__ Item item_setup(const int32_t* __restrict__ ptr, int64_t B, int H, const int32_t* __restrict__ order) {
  Item it;
  it.live = false;
  const int64_t wi = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (wi >= B * H) return it;
  const int64_t slot = wi / H;
  it.h = (int)(wi - slot * H);
  const int64_t g = order ? order[slot] : slot;
  it.n0 = ptr[g];
  it.n = ptr[g + 1] - it.n0;
  it.live = it.n > 0;
  return it;
}

// =============================================================================================================
// forward
// =============================================================================================================
// NT = live 16-row tiles of the graph (queries and keys alike), a template parameter: with run-time tile predicates the
// compiler splits the operand loads over basic blocks and waits after each (measured: 24 us; the wave spent its life
// in ~35 serial memory round trips).  Straight-line per NT, every K-side load is in flight before the first use.
template <int DH, bool DROP, int NT>
__device__ __forceinline__ void sattn_fwd_body(const Item& it, const float* __restrict__ qkv, int64_t ld64, int64_t N,
                                               int H, float scale, uint32_t thr16, float inv_keep, uint64_t seed,
                                               float* __restrict__ out, float* __restrict__ lse, float& amx) {
  using G = SGeo<DH>;
  constexpr int KPL = G::KPL, DT = G::DT;
  // paired column slots (Slots) are OFF in the forward: measured same box (tools/runs/gpu_r6k.sh), V as 8 eight-byte loads per
  // key tile instead of 16 four-byte ones made this kernel 1 us SLOWER (21.5 vs 20.4 us hot, 23.5 vs 22.5 rotating), while the
  // backward gained 5 us from the same change (44.2 vs 46.8 hot, 47.9 vs 53.0 rotating): the backward keeps it, this does not
  constexpr bool PAIRED_F = false && Slots<DT>::PAIRED;
  const int lane = threadIdx.x & 63, i = lane & 15, grp = lane >> 4;
  const int h = it.h, d = H * DH;
  const uint32_t ld = (uint32_t)ld64;
  const float* __restrict__ Qb = qkv + (int64_t)it.n0 * ld64 + h * DH;   // wave-uniform bases
  const float* __restrict__ Kb = Qb + d;
  const float* __restrict__ Vb = Qb + 2 * d;
  float* __restrict__ Ob = out + (int64_t)it.n0 * d + h * DH;

  // K-side operands of the graph: every load issued before the first use, resident for all query tiles
  float kv[4][KPL];
  float vv[DT][4][4];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < NT) {
      row_slice<KPL>(Kb, ld, 16 * t + i, it.n, grp * KPL, 1.0f, kv[t]);
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        if constexpr (PAIRED_F) {      // slot i of tile dt = column 2 i + dt: one 8-byte load feeds both tiles
          const float2 v2 = col_pair(Vb, ld, 16 * t + 4 * grp + r, it.n, 2 * i, 2 * i < DH);
          vv[0][t][r] = v2.x;
          vv[1][t][r] = v2.y;
        } else {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)
            vv[dt][t][r] = col_elem(Vb, ld, 16 * t + 4 * grp + r, it.n, dt * 16 + i, dt * 16 + i < DH);
        }
      }
    }
  float qv[KPL];
  row_slice<KPL>(Qb, ld, i, it.n, grp * KPL, scale, qv);
#pragma unroll 1
  for (int qt = 0; qt < NT; ++qt) {
    const int ql = 16 * qt + i;
    f32x4 s[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < KPL; ++c)            // key tiles interleaved: independent accumulator chains
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < NT) s[t] = mfma16(kv[t][c], qv[c], s[t]);            // S^T[key][query]
    if (qt + 1 < NT) row_slice<KPL>(Qb, ld, ql + 16, it.n, grp * KPL, scale, qv);   // next tile's Q, in flight
    float mloc = -INFINITY;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * t + 4 * grp + r;
          s[t][r] = key < it.n ? s[t][r] : -INFINITY;
          mloc = fmaxf(mloc, s[t][r]);
        }
      }
    const float m = group_max(mloc);
    const uint32_t rh = DROP ? row_hash((uint32_t)(it.n0 + ql) * (uint32_t)H + (uint32_t)h, seed) : 0u;
    float psum = 0.0f;
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < NT) {
        uint32_t h0 = 0, h1 = 0;
        if (DROP) {
          const uint32_t kp = (uint32_t)(16 * t + 4 * grp) >> 1;
          h0 = pair_hash(rh, kp);
          h1 = pair_hash(rh, kp + 1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          float p = sm_exp(s[t][r] - m);       // masked keys: exp(-inf) = 0
          psum += p;
          if (DROP) {
            const uint32_t hh = r < 2 ? h0 : h1;
            const bool keep = (r & 1) ? keep_hi(hh, thr16) : keep_lo(hh, thr16);
            p = keep ? p * inv_keep : 0.0f;
          }
          s[t][r] = p;
        }
      }
    f32x4 oacc[DT];
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) oacc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt)      // O^T[dh][query] += V^T[dh][key] P^T[key][query]
            oacc[dt] = mfma16(vv[dt][t][r], s[t][r], oacc[dt]);
      }
    const float ltot = group_sum(psum);
    const float inv_l = 1.0f / ltot;
    if (ql < it.n) {
      if constexpr (PAIRED_F) {        // slots 4 grp .. + 3 of the two tiles = columns 8 grp .. 8 grp + 7
        const int col = 8 * grp;
        if (col < DH) {
          const f32x4 o0 = oacc[0] * inv_l, o1 = oacc[1] * inv_l;
          float* dst = Ob + (int64_t)ql * d + col;
          *reinterpret_cast<float4*>(dst) = make_float4(o0[0], o1[0], o0[1], o1[1]);
          *reinterpret_cast<float4*>(dst + 4) = make_float4(o0[2], o1[2], o0[3], o1[3]);
#pragma unroll
          for (int r = 0; r < 4; ++r) amx = fmaxf(fmaxf(amx, fabsf(o0[r])), fabsf(o1[r]));
        }
      } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int col = dt * 16 + 4 * grp;
          if (col < DH) {
            const f32x4 o = oacc[dt] * inv_l;
            *reinterpret_cast<float4*>(Ob + (int64_t)ql * d + col) = make_float4(o[0], o[1], o[2], o[3]);
            amx = fmaxf(fmaxf(fmaxf(fmaxf(amx, fabsf(o[0])), fabsf(o[1])), fabsf(o[2])), fabsf(o[3]));
          }
        }
      }
      if (grp == 0) lse[(int64_t)h * N + it.n0 + ql] = m + logf(ltot);
    }
  }
}

// (dh = 32 needs 136 registers with the dropout hash and the max|out| word: three wavefronts per SIMD there, no spills)
template <int DH, bool DROP>
__global__ __launch_bounds__(256, DH >= 32 ? 3 : 4) void k_sattn_fwd(
    const float* __restrict__ qkv, int64_t ld64, const int32_t* __restrict__ ptr, int64_t B, int64_t N, int H,
    float scale, uint32_t thr16, float inv_keep, uint64_t seed, const uint64_t* __restrict__ salt,
    float* __restrict__ out, float* __restrict__ lse, uint32_t* __restrict__ amax, const int32_t* __restrict__ order) {
  const Item it = item_setup(ptr, B, H, order);      // host guarantees n <= 64: one block per graph
  if (!it.live) return;
  // the block form is selected by a HOST hint (the batch's longest graph); a stale or wrong hint must not produce a silently
  // truncated result (only the first 64 rows of the graph would be touched): abort the launch loudly instead
  if (it.n > 64) __builtin_trap();
  seed = gps::salted_seed(seed, salt);
  float amx = 0.0f;                           // max|out| of this (graph, head): the record of the out-projection GEMM
  switch ((it.n + 15) >> 4) {
    case 1: sattn_fwd_body<DH, DROP, 1>(it, qkv, ld64, N, H, scale, thr16, inv_keep, seed, out, lse, amx); break;
    case 2: sattn_fwd_body<DH, DROP, 2>(it, qkv, ld64, N, H, scale, thr16, inv_keep, seed, out, lse, amx); break;
    case 3: sattn_fwd_body<DH, DROP, 3>(it, qkv, ld64, N, H, scale, thr16, inv_keep, seed, out, lse, amx); break;
    default: sattn_fwd_body<DH, DROP, 4>(it, qkv, ld64, N, H, scale, thr16, inv_keep, seed, out, lse, amx); break;
  }
  wave_amax(amx, amax);
}

// =============================================================================================================
// backward, graphs of <= 64 nodes: dQ, dK, dV in one launch
// =============================================================================================================
constexpr int KTP = 68;      // K^T scratch pitch (floats): 64 keys + 4, rows stay 16-byte aligned
template <int DH, bool DROP, int NT>
__device__ __forceinline__ void sattn_bwd_body(const Item& it, float* __restrict__ tP, float* __restrict__ kT,
                                               const float* __restrict__ d_out,
                                               const float* __restrict__ qkv, int64_t ld64,
                                               const float* __restrict__ out, const float* __restrict__ lse,
                                               int64_t N, int H, float scale, uint32_t thr16, float inv_keep,
                                               uint64_t seed, float* __restrict__ d_qkv, int64_t ldg, float& amx) {
  using G = SGeo<DH>;
  constexpr int KPL = G::KPL, DT = G::DT;
  // paired column slots (Slots) in the backward for dh = 24 only: at dh = 32 the 8-byte operand pairs cost the no-dropout
  // instantiation its last free registers (256 + 4 spilled); the forward pairs at both widths
  constexpr bool PAIRED_B = Slots<DT>::PAIRED && DH != 32;
  constexpr int PT = 20;                      // transpose scratch pitch (floats): 16 + 4, rows stay 16-byte aligned
  const int lane = threadIdx.x & 63, i = lane & 15, grp = lane >> 4;
  const int h = it.h, d = H * DH;
  const uint32_t ld = (uint32_t)ld64, du = (uint32_t)d;
  const float* __restrict__ Qb = qkv + (int64_t)it.n0 * ld64 + h * DH;
  const float* __restrict__ Kb = Qb + d;
  const float* __restrict__ Vb = Qb + 2 * d;
  const float* __restrict__ dOb = d_out + (int64_t)it.n0 * d + h * DH;
  const float* __restrict__ Ob = out + (int64_t)it.n0 * d + h * DH;
  const float* __restrict__ lse_b = lse + (int64_t)h * N + it.n0;
  float* __restrict__ tS = tP + 16 * PT;

  // K-side operands, resident for all query tiles: row slices of K and V (S^T = K Q^T, dP^T = V dO^T).  The column
  // form of K (dQ^T = K^T dS^T) lives in the wave's LDS scratch as K^T[dh][key] (round 4): as 32 more resident registers
  // it put the dh = 24 instantiation at 256 VGPRs with 13 spilled (56 bytes of scratch per lane); each query tile now
  // fetches its K^T fragments with DT x NT 16-byte LDS reads, written once from the row slices already in registers.
  float kv[4][KPL], vk[4][KPL];
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < NT) {
      row_slice<KPL>(Kb, ld, 16 * t + i, it.n, grp * KPL, 1.0f, kv[t]);
      row_slice<KPL>(Vb, ld, 16 * t + i, it.n, grp * KPL, 1.0f, vk[t]);
#pragma unroll
      for (int c = 0; c < KPL; ++c)           // lane (i, grp) holds K[key 16t + i][dh grp * KPL + c]
        if (grp * KPL + c < DT * 16) kT[(grp * KPL + c) * KTP + 16 * t + i] = kv[t][c];
    }
  __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");     // same wave writes and reads: LDS is in order per wave
  __builtin_amdgcn_wave_barrier();
  f32x4 dk[4][DT], dv[4][DT];
#pragma unroll
  for (int t = 0; t < 4; ++t)
#pragma unroll
    for (int dt = 0; dt < DT; ++dt) {
      dk[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dv[t][dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }

#pragma unroll 1
  for (int qt = 0; qt < NT; ++qt) {
    const int ql = 16 * qt + i;               // this lane's query (column of the ^T tiles)
    const bool q_ok = ql < it.n;
    // everything this query tile needs from memory, requested up front
    float qv[KPL], dov[KPL], ov[KPL];
    row_slice<KPL>(Qb, ld, ql, it.n, grp * KPL, scale, qv);
    row_slice<KPL>(dOb, du, ql, it.n, grp * KPL, 1.0f, dov);
    row_slice<KPL>(Ob, du, ql, it.n, grp * KPL, 1.0f, ov);
    float qc[DT][4], dc[DT][4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (PAIRED_B) {        // (slot i of tile dt = column 2 i + dt, see Slots)
        const float2 q2 = col_pair(Qb, ld, 16 * qt + 4 * grp + r, it.n, 2 * i, 2 * i < DH);
        const float2 d2 = col_pair(dOb, du, 16 * qt + 4 * grp + r, it.n, 2 * i, 2 * i < DH);
        qc[0][r] = q2.x * scale; qc[1][r] = q2.y * scale;
        dc[0][r] = d2.x; dc[1][r] = d2.y;
      } else {
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const bool in = dt * 16 + i < DH;
          qc[dt][r] = col_elem(Qb, ld, 16 * qt + 4 * grp + r, it.n, dt * 16 + i, in) * scale;
          dc[dt][r] = col_elem(dOb, du, 16 * qt + 4 * grp + r, it.n, dt * 16 + i, in);
        }
      }
    }
    const float lse_q = lse_b[min(ql, it.n - 1)];
    // delta_q = sum_c dO[q][c] O[q][c] over the head's dh columns
    float dl_part = 0.0f;
#pragma unroll
    for (int c = 0; c < KPL; ++c) dl_part += dov[c] * ov[c];
    const float dl_q = group_sum(dl_part);
    const uint32_t rh = DROP ? row_hash((uint32_t)(it.n0 + ql) * (uint32_t)H + (uint32_t)h, seed) : 0u;

    f32x4 s[4], dp[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      s[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
      dp[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    }
#pragma unroll
    for (int c = 0; c < KPL; ++c)
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < NT) {
          s[t] = mfma16(kv[t][c], qv[c], s[t]);       // S^T[key][query]
          dp[t] = mfma16(vk[t][c], dov[c], dp[t]);    // dP^T[key][query]
        }
    // P, dropout, dS^T; s <- dS^T (B operand of dQ^T), dp <- P_drop^T
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < NT) {
        uint32_t h0 = 0, h1 = 0;
        if (DROP) {
          const uint32_t kp = (uint32_t)(16 * t + 4 * grp) >> 1;
          h0 = pair_hash(rh, kp);
          h1 = pair_hash(rh, kp + 1);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = 16 * t + 4 * grp + r;
          const bool live = key < it.n && q_ok;
          const float p = live ? sm_exp(s[t][r] - lse_q) : 0.0f;
          float dpe = dp[t][r], pd = p;
          if (DROP) {
            const uint32_t hh = r < 2 ? h0 : h1;
            const bool keep = (r & 1) ? keep_hi(hh, thr16) : keep_lo(hh, thr16);
            pd = keep ? p * inv_keep : 0.0f;
            dpe = keep ? dpe * inv_keep : 0.0f;
          }
          s[t][r] = p * (dpe - dl_q);                  // dS^T[key][query]
          dp[t][r] = pd;                               // P_drop^T[key][query]
        }
      }
    // dQ^T[dh][query] = sum_key K^T[dh][key] dS^T[key][query]
    {
      f32x4 acc[DT];
#pragma unroll
      for (int dt = 0; dt < DT; ++dt) acc[dt] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
      for (int t = 0; t < 4; ++t)
        if (t < NT) {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {   // K^T[dh 16dt + i][keys 16t + 4grp .. + 3]
            const bool in = dt * 16 + i < DH;
            const f32x4 k4 = *reinterpret_cast<const f32x4*>(kT + (dt * 16 + i) * KTP + 16 * t + 4 * grp);
#pragma unroll
            for (int r = 0; r < 4; ++r) acc[dt] = mfma16(in ? k4[r] : 0.0f, s[t][r], acc[dt]);
          }
        }
      if (q_ok) {
        float* __restrict__ Gq = d_qkv + (int64_t)(it.n0 + ql) * ldg + h * DH;
#pragma unroll
        for (int dt = 0; dt < DT; ++dt) {
          const int col = dt * 16 + 4 * grp;
          if (col < DH) {
            const f32x4 o = acc[dt] * scale;
            *reinterpret_cast<float4*>(Gq + col) = make_float4(o[0], o[1], o[2], o[3]);
            amx = fmaxf(fmaxf(fmaxf(fmaxf(amx, fabsf(o[0])), fabsf(o[1])), fabsf(o[2])), fabsf(o[3]));
          }
        }
      }
    }
    // per key tile: transpose (P_drop^T, dS^T)[key][query] -> [query][key] through the wave's scratch, then
    // dV^T[dh][key] += dO^T[dh][q] P_drop[q][key],  dK^T[dh][key] += (scale Q)^T[dh][q] dS[q][key]
#pragma unroll
    for (int t = 0; t < 4; ++t)
      if (t < NT) {
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          tP[(4 * grp + r) * PT + i] = dp[t][r];      // row = key (local to the tile), column = query
          tS[(4 * grp + r) * PT + i] = s[t][r];
        }
        // same wave wrote and reads: LDS operations of one wave complete in order (the fence only pins the
        // compiler's ordering of the stores above against the loads below)
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
        __builtin_amdgcn_wave_barrier();
        const float4 pT = *reinterpret_cast<const float4*>(tP + i * PT + 4 * grp);   // [key i][query 4grp..+3]
        const float4 sT_ = *reinterpret_cast<const float4*>(tS + i * PT + 4 * grp);
        const float pv[4] = {pT.x, pT.y, pT.z, pT.w}, sv[4] = {sT_.x, sT_.y, sT_.z, sT_.w};
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            dv[t][dt] = mfma16(dc[dt][r], pv[r], dv[t][dt]);
            dk[t][dt] = mfma16(qc[dt][r], sv[r], dk[t][dt]);
          }
        __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");   // the next tile overwrites the scratch
        __builtin_amdgcn_wave_barrier();
      }
  }
#pragma unroll
  for (int t = 0; t < 4; ++t)
    if (t < NT) {
      const int kl = 16 * t + i;
      if (kl < it.n) {
        float* __restrict__ Gk = d_qkv + (int64_t)(it.n0 + kl) * ldg + d + h * DH;
        float* __restrict__ Gv = Gk + d;
        if constexpr (PAIRED_B) {      // the dK^T / dV^T tiles carry the slot order of their Q^T / dO^T operands
          const int col = 8 * grp;
          if (col < DH) {
            *reinterpret_cast<float4*>(Gk + col) = make_float4(dk[t][0][0], dk[t][1][0], dk[t][0][1], dk[t][1][1]);
            *reinterpret_cast<float4*>(Gk + col + 4) = make_float4(dk[t][0][2], dk[t][1][2], dk[t][0][3], dk[t][1][3]);
            *reinterpret_cast<float4*>(Gv + col) = make_float4(dv[t][0][0], dv[t][1][0], dv[t][0][1], dv[t][1][1]);
            *reinterpret_cast<float4*>(Gv + col + 4) = make_float4(dv[t][0][2], dv[t][1][2], dv[t][0][3], dv[t][1][3]);
#pragma unroll
            for (int dt = 0; dt < 2; ++dt)
#pragma unroll
              for (int r = 0; r < 4; ++r) amx = fmaxf(fmaxf(amx, fabsf(dk[t][dt][r])), fabsf(dv[t][dt][r]));
          }
        } else {
#pragma unroll
          for (int dt = 0; dt < DT; ++dt) {
            const int col = dt * 16 + 4 * grp;
            if (col < DH) {
              *reinterpret_cast<float4*>(Gk + col) = make_float4(dk[t][dt][0], dk[t][dt][1], dk[t][dt][2], dk[t][dt][3]);
              *reinterpret_cast<float4*>(Gv + col) = make_float4(dv[t][dt][0], dv[t][dt][1], dv[t][dt][2], dv[t][dt][3]);
#pragma unroll
              for (int r = 0; r < 4; ++r) amx = fmaxf(fmaxf(amx, fabsf(dk[t][dt][r])), fabsf(dv[t][dt][r]));
            }
          }
        }
      }
    }
}

template <int DH, bool DROP>
__global__ __launch_bounds__(256, 2) void k_sattn_bwd(
    const float* __restrict__ d_out, const float* __restrict__ qkv, int64_t ld64, const float* __restrict__ out,
    const float* __restrict__ lse, const int32_t* __restrict__ ptr, int64_t B, int64_t N, int H, float scale,
    uint32_t thr16, float inv_keep, uint64_t seed, const uint64_t* __restrict__ salt, float* __restrict__ d_qkv,
    int64_t ldg, uint32_t* __restrict__ amax, const int32_t* __restrict__ order) {
  __shared__ __attribute__((aligned(16))) float sT[4][2 * 16 * 20];   // per-wave transpose scratch (P_drop | dS)
  __shared__ __attribute__((aligned(16))) float sK[4][SGeo<DH>::DT * 16 * KTP];   // per-wave K^T[dh][key]
  const Item it = item_setup(ptr, B, H, order);      // host guarantees n <= 64: one block per graph
  if (!it.live) return;
  // the block form is selected by a HOST hint (the batch's longest graph); a stale or wrong hint must not produce a silently
  // truncated result (only the first 64 rows of the graph would be touched): abort the launch loudly instead
  if (it.n > 64) __builtin_trap();
  seed = gps::salted_seed(seed, salt);
  float* tP = &sT[threadIdx.x >> 6][0];
  float* kT = &sK[threadIdx.x >> 6][0];
  float amx = 0.0f;                           // max over this item's dq | dk | dv: the record of the GEMMs that read d_qkv
#define SA_BODY(NTV) sattn_bwd_body<DH, DROP, NTV>(it, tP, kT, d_out, qkv, ld64, out, lse, N, H, scale, thr16, inv_keep, \
                                                  seed, d_qkv, ldg, amx)
  switch ((it.n + 15) >> 4) {
    case 1: SA_BODY(1); break;
    case 2: SA_BODY(2); break;
    case 3: SA_BODY(3); break;
    default: SA_BODY(4); break;
  }
#undef SA_BODY
  wave_amax(amx, amax);
}

inline bool al16(const void* p) { return (reinterpret_cast<uintptr_t>(p) % 16) == 0; }

}  // namespace

namespace attn {

static bool sattn_dh_ok(int dh) { return dh == 8 || dh == 16 || dh == 24 || dh == 32; }

bool sattn_applicable(const void* qkv, int64_t ld_qkv, const void* out, int H, int dh) {
  const char* sw = getenv("GPS_SATTN");
  if (sw && sw[0] == '0') return false;
  return sattn_dh_ok(dh) && ld_qkv % 4 == 0 && (H * dh) % 4 == 0 && al16(qkv) && al16(out);
}

// Launch the block-form forward.  Preconditions: sattn_applicable().
void sattn_fwd_launch(const float* qkv, int64_t ld_qkv, const int32_t* ptr, int64_t B, int64_t N, int H, int dh,
                      float scale, float p_drop, uint64_t seed, float* out, float* lse, uint32_t* amax, const int32_t* order,
                      hipStream_t s) {
  const uint32_t thr16 = drop_thr16(p_drop);
  const float inv_keep = p_drop > 0.0f ? drop_inv_keep(thr16) : 1.0f;
  const unsigned grid = gps::grid_for(B * H, 4);
#define SA_FWD(D)                                                                                               \
  do {                                                                                                          \
    if (p_drop > 0.0f)                                                                                          \
      k_sattn_fwd<D, true><<<grid, 256, 0, s>>>(qkv, ld_qkv, ptr, B, N, H, scale, thr16, inv_keep, seed,         \
                                                gps::dropout_salt(), out, lse, amax, order);                        \
    else                                                                                                        \
      k_sattn_fwd<D, false><<<grid, 256, 0, s>>>(qkv, ld_qkv, ptr, B, N, H, scale, thr16, inv_keep, seed,        \
                                                 gps::dropout_salt(), out, lse, amax, order);                       \
  } while (0)
  switch (dh) {
    case 8: SA_FWD(8); break;
    case 16: SA_FWD(16); break;
    case 24: SA_FWD(24); break;
    case 32: SA_FWD(32); break;
  }
#undef SA_FWD
}

// Launch the fused backward (every graph has <= 64 nodes).  Preconditions: sattn_applicable(), aligned d_out / d_qkv.
void sattn_bwd_launch(const float* d_out, const float* qkv, int64_t ld_qkv, const float* out, const float* lse,
                      const int32_t* ptr, int64_t B, int64_t N, int H, int dh, float scale, float p_drop,
                      uint64_t seed, float* d_qkv, int64_t ld_dqkv, uint32_t* amax, const int32_t* order, hipStream_t s) {
  const uint32_t thr16 = drop_thr16(p_drop);
  const float inv_keep = p_drop > 0.0f ? drop_inv_keep(thr16) : 1.0f;
  const unsigned grid = gps::grid_for(B * H, 4);
#define SA_BWD(D)                                                                                               \
  do {                                                                                                          \
    if (p_drop > 0.0f)                                                                                          \
      k_sattn_bwd<D, true><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, out, lse, ptr, B, N, H, scale, thr16,        \
                                                inv_keep, seed, gps::dropout_salt(), d_qkv, ld_dqkv, amax, order);      \
    else                                                                                                        \
      k_sattn_bwd<D, false><<<grid, 256, 0, s>>>(d_out, qkv, ld_qkv, out, lse, ptr, B, N, H, scale, thr16,       \
                                                 inv_keep, seed, gps::dropout_salt(), d_qkv, ld_dqkv, amax, order);     \
  } while (0)
  switch (dh) {
    case 8: SA_BWD(8); break;
    case 16: SA_BWD(16); break;
    case 24: SA_BWD(24); break;
    case 32: SA_BWD(32); break;
  }
#undef SA_BWD
}

}  // namespace attn
