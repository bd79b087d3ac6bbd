// In-launch column reductions: per-workgroup column partials -> two-level tree -> final [d] vectors, without a second
// launch.  Used by the producers of BatchNorm inputs (csrc/block_norm.hip row kernels, the GatedGCN forward, the ring
// GEMM epilogue) so that the batch statistics (graphgps/layer/gatedgcn_layer.py:72-73, gps_layer.py:191-194,212-229:
// every nn.BatchNorm1d of the block in training mode) and the BatchNorm-backward column sums are complete when the
// producing kernel retires: the 6 "finalize" launches per layer of round 2 are gone.
//
// Protocol (cdna_hip_programming.md section 6 Guideline 16, counter form; no dependence on dispatch order, timing or
// workgroup -> XCD placement, nobody ever waits for anybody):
//   * workgroup b stores its level-0 record (NV vectors of d floats + a row count) WRITE-THROUGH (relaxed agent-scope
//     atomic stores = `global_store ... sc1`), every wave drains its stores (`s_waitcnt vmcnt(0)`), the workgroup
//     barriers, ONE lane takes a ticket on the counter of the record's group (`fan` consecutive records);
//   * the workgroup that draws the last ticket of a group combines the group's records (sc1 loads: L1 is bypassed, the
//     records were written through, so no acquire fence is needed), stores the level-1 record the same way, resets the
//     group counter and takes a ticket on the root counter; the last of those combines the level-1 records into the
//     final vectors and resets the root counter.
//   * the counters are zero at entry and zero again at exit, so a hipGraph replay (or the next eager call) finds them
//     ready without a memset node; two launches that may run CONCURRENTLY (forked streams) must use different counters.
// Determinism: records are combined in index order, never in arrival order, so the result is bitwise reproducible.
//
// Memory model: the records travel as RELAXED agent-scope atomics (write-through `sc1` stores, L1-bypassing
// `sc1` loads) ordered against the ticket by `s_waitcnt vmcnt(0)` + the workgroup barrier, NOT by release / acquire on the
// ticket.  That is deliberate and specific to gfx942 / gfx950 (this header refuses other targets below): an agent-scope
// release there is a `buffer_wbl2` -- a write-back of the XCD's whole L2, i.e. of the very tensor the producing kernel has
// just stored (tens of MB) -- once per workgroup, which would cost more than the launch this protocol removes; the
// write-through stores make the records visible at the memory side without it, and every reader bypasses its own caches.
//
// Counters must be ZERO at entry.  A launch that died mid-tree (a fault elsewhere in the process, a killed graph replay)
// leaves them non-zero; the next launch on those counters would then elect the wrong "last" workgroup and publish
// statistics of incomplete records.  That cannot happen silently: whatever the stale value, at least one workgroup of the
// next launch draws a ticket >= the group size, and a ticket that large TRAPS (the launch -- and with it the HIP context --
// fails loudly; tests/test_hip_norm.py poisons a counter in a child process to show it).  gps_sync_reset (include/
// gps_hip.h) re-zeroes a counter buffer for a caller that survived the original failure and wants to go on.
//
// Record kinds: STATS = (mean_b, M2_b; n_b) combined exactly as csrc/bn_fused.hip's finalize (weighted mean, then
// sum of M2_b + n_b (mean_b - mean)^2: Welford/Chan-grade accuracy); SUMS = plain column sums.
#pragma once
#include <hip/hip_runtime.h>

#if defined(__HIP_DEVICE_COMPILE__) && !defined(__gfx950__) && !defined(__gfx942__)
#error "col_tree.hpp: the write-through / relaxed-atomic protocol is validated for gfx942 / gfx950 only (see the header comment)"
#endif

#include <cstddef>
#include <cstdint>

namespace gps {
namespace tree {

constexpr int kSyncWords = 32;                  // uint32 counters per tree: [0] root, [1 + g] group g
constexpr int kMaxFan = kSyncWords - 1;         // fan-in of either level (the group count is bounded by the counters)
constexpr int kMaxParts = kMaxFan * kMaxFan;    // level-0 records per tree
enum { STATS = 0, SUMS = 1 };

struct Tree {
  float* part;     // level 0: [P][NV][d]
  float* pcnt;     // level 0 row counts [P]
  float* grp;      // level 1: [NG][NV][d]
  float* gcnt;     // [NG]
  unsigned* tick;  // [1 + NG] arrival counters
  int P, fan, NG, NV, mode;
  // STATS: o0 = mean, o1 = rstd, o2 / o3 = running_mean / running_var (or null); SUMS: o0 = sum 0, o1 = sum 1,
  // (NV == 3) o2 = sum 0 again, o3 = sum 2
  float *o0, *o1, *o2, *o3;
  float eps, momentum;
  // STATS with NV == 4 (two (mean, M2) pairs per record, two counts): the second pair's outputs
  float *p0, *p1, *p2, *p3;
  float eps2, momentum2;
};

// ---- host side: layout of one tree inside a float workspace ----------------------------------------------------
__host__ __device__ static inline int fan_for(int P) {
  int f = 1;
  while (f * f < P) ++f;
  return f;
}
__host__ __device__ static inline size_t pad4(size_t x) { return (x + 3) & ~(size_t)3; }
// (count arrays are sized for two counts per record, the most any record kind carries)
static inline size_t floats_for(int P, int NV, int d) {
  const int fan = fan_for(P), NG = (P + fan - 1) / fan;
  return pad4((size_t)P * NV * d) + pad4((size_t)2 * P) + pad4((size_t)NG * NV * d) + pad4((size_t)2 * NG);
}
// Carves a tree over P level-0 records out of `ws` (advanced past it).  P <= kMaxParts.
static inline Tree carve(float*& ws, unsigned* tick, int P, int NV, int mode, int d) {
  Tree T{};
  T.P = P; T.NV = NV; T.mode = mode;
  T.fan = fan_for(P);
  T.NG = (P + T.fan - 1) / T.fan;
  T.part = ws; ws += pad4((size_t)P * NV * d);
  T.pcnt = ws; ws += pad4((size_t)2 * P);
  T.grp = ws; ws += pad4((size_t)T.NG * NV * d);
  T.gcnt = ws; ws += pad4((size_t)2 * T.NG);
  T.tick = tick;
  return T;
}

// ---- device side -------------------------------------------------------------------------------------------------
__device__ __forceinline__ void st_sc1(float* p, float v) {
  __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
__device__ __forceinline__ float ld_sc1(const float* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// final vectors of one training-mode BatchNorm1d from (mean, M2, n): mean, rstd and the running statistics
__device__ __forceinline__ void stats_out(float* mean_o, float* rstd_o, float* rmean, float* rvar, float eps, float mom,
                                          int c, float mean, float m2, float n) {
  mean_o[c] = mean;
  rstd_o[c] = 1.0f / sqrtf(m2 / n + eps);
  if (rmean) {
    rmean[c] = (1.0f - mom) * rmean[c] + mom * mean;
    rvar[c] = (1.0f - mom) * rvar[c] + mom * (m2 / fmaxf(n - 1.0f, 1.0f));
  }
}

// LDS scratch the caller must provide to arrive(): floats.
__host__ __device__ constexpr int scratch_floats(int NV, int threads) { return 4 + (NV + NV / 2 + 1) * threads; }

// Number of row counts a record carries: STATS records are NV / 2 independent (mean, M2) pairs, each with its own count
// (the GatedGCN forward emits the statistics of x~ and of e^ as ONE record: one ticket, one tree).
template <int NV, int MODE>
struct Counts {
  static constexpr int value = MODE == STATS ? NV / 2 : 0;
};

// (n, m, q) (+)= (n2, m2, q2): Chan's parallel combination of two (count, mean, M2) triples.
__device__ __forceinline__ void chan_merge(float& n, float& m, float& q, float n2, float m2, float q2) {
  const float nn = n + n2, dl = m2 - m, w = nn > 0.f ? n2 / nn : 0.f;
  q = q + q2 + dl * dl * n * w;
  m = m + dl * w;
  n = nn;
}

// Combines `cnt` records starting at i0 (record stride NV * d floats, counts in n_src) for ONE column c, taking the
// records sub, sub + nsub, ... ; returns the combined record in (acc, n).  All loads of up to MAXI records are issued
// before the first is used (one memory round trip); longer lists repeat in chunks merged by Chan's formula.
template <int NV, int MODE, int MAXI>
__device__ __forceinline__ void combine_column(const float* src, const float* n_src, int i0, int cnt, int sub, int nsub,
                                               int d, int c, float (&acc)[NV], float (&n)[NV / 2 + 1]) {
  constexpr int NS = Counts<NV, MODE>::value;
#pragma unroll
  for (int v = 0; v < NV; ++v) acc[v] = 0.f;
#pragma unroll
  for (int s = 0; s < NV / 2 + 1; ++s) n[s] = 0.f;
  const int mine = cnt > sub ? (cnt - sub + nsub - 1) / nsub : 0;
  for (int base = 0; base < mine; base += MAXI) {
    float val[NV][MAXI], nb[NS > 0 ? NS : 1][MAXI];
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
      const int jj = base + j < mine ? base + j : mine - 1;         // clamped: loads unconditional, masked below
      const int64_t rec = i0 + sub + (int64_t)jj * nsub;
#pragma unroll
      for (int v = 0; v < NV; ++v) val[v][j] = ld_sc1(src + (rec * NV + v) * d + c);
#pragma unroll
      for (int s = 0; s < NS; ++s) nb[s][j] = ld_sc1(n_src + rec * NS + s);
    }
#pragma unroll
    for (int j = 0; j < MAXI; ++j) {
      const bool ok = base + j < mine;
#pragma unroll
      for (int s = 0; s < NS; ++s) nb[s][j] = ok ? nb[s][j] : 0.f;
#pragma unroll
      for (int v = 0; v < NV; ++v) val[v][j] = ok ? val[v][j] : 0.f;
    }
    if constexpr (MODE == STATS) {
#pragma unroll
      for (int s = 0; s < NS; ++s) {
        float nc = 0.f, sm = 0.f;
#pragma unroll
        for (int j = 0; j < MAXI; ++j) { nc += nb[s][j]; sm += nb[s][j] * val[2 * s][j]; }
        const float mc = nc > 0.f ? sm / nc : 0.f;
        float qc = 0.f;
#pragma unroll
        for (int j = 0; j < MAXI; ++j) { const float dl = val[2 * s][j] - mc; qc += val[2 * s + 1][j] + nb[s][j] * dl * dl; }
        chan_merge(n[s], acc[2 * s], acc[2 * s + 1], nc, mc, qc);
      }
    } else {
#pragma unroll
      for (int v = 0; v < NV; ++v)
#pragma unroll
        for (int j = 0; j < MAXI; ++j) acc[v] += val[v][j];
    }
  }
}

// One level: every column of the d-wide records [i0, i0 + cnt) of `src` -> handler(c, acc, n) called by one thread
// per column.  Threads beyond d columns split the record list (nsub slices) and meet through LDS, merged in slice order.
// NTH: the threads that take part (0 = the whole workgroup; a kernel whose trailing wavefronts have already exited -- the
// loader wavefronts of k_gemm_ring16L -- names the count of the leading threads that are still there).
template <int NV, int MODE, int MAXI, int NTH = 0, typename F>
__device__ __forceinline__ void level(const float* src, const float* n_src, int i0, int cnt, int d, float* scr,
                                      F&& handler) {
  constexpr int NS = Counts<NV, MODE>::value;
  constexpr int RW = NV + NS;                            // floats per exchanged record
  const int T = NTH > 0 ? NTH : (int)blockDim.x, tid = threadIdx.x;
  const int nsub = T >= 2 * d ? T / d : 1;
  const int cpp = nsub > 1 ? d : (T < d ? T : d);      // columns per pass
  const int sub = tid / cpp, cl = tid - sub * cpp;
  for (int cb = 0; cb < d; cb += cpp) {
    const int c = cb + cl;
    const bool active = sub < nsub && c < d;
    float acc[NV], n[NV / 2 + 1];
#pragma unroll
    for (int v = 0; v < NV; ++v) acc[v] = 0.f;
#pragma unroll
    for (int s = 0; s < NV / 2 + 1; ++s) n[s] = 0.f;
    if (active) combine_column<NV, MODE, MAXI>(src, n_src, i0, cnt, sub, nsub, d, c, acc, n);
    if (nsub > 1) {
      if (active && sub > 0) {
#pragma unroll
        for (int v = 0; v < NV; ++v) scr[(sub * RW + v) * cpp + cl] = acc[v];
#pragma unroll
        for (int s = 0; s < NS; ++s) scr[(sub * RW + NV + s) * cpp + cl] = n[s];
      }
      __syncthreads();
      if (active && sub == 0) {
        for (int q = 1; q < nsub; ++q) {
          float a2[NV];
#pragma unroll
          for (int v = 0; v < NV; ++v) a2[v] = scr[(q * RW + v) * cpp + cl];
          if constexpr (MODE == STATS) {
#pragma unroll
            for (int s = 0; s < NS; ++s)
              chan_merge(n[s], acc[2 * s], acc[2 * s + 1], scr[(q * RW + NV + s) * cpp + cl], a2[2 * s], a2[2 * s + 1]);
          } else {
#pragma unroll
            for (int v = 0; v < NV; ++v) acc[v] += a2[v];
          }
        }
      }
      __syncthreads();
    }
    if (active && sub == 0) handler(c, acc, n);
  }
}

// Called by EVERY thread of workgroup-record b after the record's stores (st_sc1) were issued.  `lds`: scratch_floats(NV,
// blockDim.x) floats, 16-byte aligned, free to overwrite.  Returns only after this workgroup's share of the tree is done.
template <int NV, int MODE, int MAXI, int NTH = 0>
__device__ __forceinline__ void arrive(const Tree& T, int b, int d, float* lds) {
  constexpr int NS = Counts<NV, MODE>::value;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");     // every storing wave drains its write-through stores
  __syncthreads();
  int* flag = reinterpret_cast<int*>(lds);
  float* scr = lds + 4;
  const int g = b / T.fan;
  const int i0 = g * T.fan;
  const int cnt = T.P - i0 < T.fan ? T.P - i0 : T.fan;
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(T.tick + 1 + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t >= (unsigned)cnt) __builtin_trap();           // the counter was not zero at entry: fail loudly (header comment)
    *flag = t == (unsigned)(cnt - 1);
  }
  __syncthreads();
  if (!*flag) return;                                   // workgroup-uniform
  __syncthreads();                                      // the flag word is rewritten below
  float* gdst = T.grp + (int64_t)g * NV * d;
  float* gn = T.gcnt + (int64_t)g * NS;
  level<NV, MODE, MAXI, NTH>(T.part, T.pcnt, i0, cnt, d, scr, [&](int c, const float (&acc)[NV], const float (&n)[NV / 2 + 1]) {
#pragma unroll
    for (int v = 0; v < NV; ++v) st_sc1(gdst + (int64_t)v * d + c, acc[v]);
    if (c == 0) {
#pragma unroll
      for (int s = 0; s < NS; ++s) st_sc1(gn + s, n[s]);
    }
  });
  if (threadIdx.x == 0) __hip_atomic_store(T.tick + 1 + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (threadIdx.x == 0) {
    const unsigned t = __hip_atomic_fetch_add(T.tick, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t >= (unsigned)T.NG) __builtin_trap();
    *flag = t == (unsigned)(T.NG - 1);
  }
  __syncthreads();
  if (!*flag) return;
  __syncthreads();
  const Tree R = T;       // by value: the lambda below runs after T's storage may have been re-read a few times
  level<NV, MODE, MAXI, NTH>(T.grp, T.gcnt, 0, T.NG, d, scr, [&](int c, const float (&acc)[NV], const float (&n)[NV / 2 + 1]) {
    if constexpr (MODE == STATS) {
      stats_out(R.o0, R.o1, R.o2, R.o3, R.eps, R.momentum, c, acc[0], acc[1], n[0]);
      if constexpr (NV >= 4) stats_out(R.p0, R.p1, R.p2, R.p3, R.eps2, R.momentum2, c, acc[2], acc[3], n[1]);
    } else {
      R.o0[c] = acc[0];
      if constexpr (NV > 1) R.o1[c] = acc[1];
      if constexpr (NV > 2) { R.o2[c] = acc[0]; R.o3[c] = acc[2]; }
    }
  });
  if (threadIdx.x == 0) __hip_atomic_store(T.tick, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

}  // namespace tree
}  // namespace gps
