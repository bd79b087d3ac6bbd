// FAVOR+ (Performer softmax-kernel linear attention) over ptr segments, on fp32 MFMA.
//
// Reference arithmetic: graphgps/layer/performer_layer.py:119-144 (softmax_kernel),
// :200-205 (linear_attention), :467-503 (Attention.forward), as called through
// performer_pytorch.SelfAttention at graphgps/layer/gps_layer.py:111-114,206 on the
// to_dense_batch-padded tensor.  Per graph g (n rows; Nmax = longest graph of the batch) and head:
//     dd_q = (c q) P^T,  phi_q = r (exp(dd_q - |cq|^2/2 - rowmax(dd_q)) + 1e-4)
//     dd_k = (c k) P^T,  phi_k = r (exp(dd_k - |ck|^2/2 - M) + 1e-4),  M = max over ALL rows and features
//     ksum = sum_n phi_k (+ padded rows),  ctx = phi_k^T v,  out = (phi_q ctx) / (phi_q . ksum)
// with c = dh^-1/4, r = m^-1/2.  The reference masks only V, so each of the (Nmax - n) zero-padded
// key rows still contributes r (exp(-M) + 1e-4) to ksum and a 0 logit to M;
// this varlen kernel never builds padded rows and adds that closed form instead.
//
// Machine mapping: dim_head = 64, m <= 272 (17 feature tiles of 16; the package default
// m = int(64 ln 64) = 266).  All contractions are v_mfma_f32_16x16x4_f32 built from two operand
// patterns (same order-free-contraction trick as seg_attention.hip, no LDS, no cross-lane moves):
//   rows x rows^T : C[4g+r][l&15] = sum_k A_row(l&15)[k] * B_row(l&15)[k], each lane loading 16
//                   contiguous floats of its row                               (mm_rows)
//   chain         : acc[4g+r'][l&15] += sum_{4g+r} X[row 4g+r][col l&15] * Cprev[4g+r][l&15]
//                   i.e. a previous accumulator is consumed directly as the B operand  (mm_chain)
// Forward: k-max (atomicMax, order-free) -> ctx/ksum per (graph, head, feature tile) -> output per
// (16-query tile, head).  Backward: 4 passes (query side, context side, key side, key-max fix-up),
// every cross-row reduction keyed so that it is summed in a fixed order: deterministic.
// Round 5: key max, query side and key side also exist with the projection staged in LDS (LP = true, below) -- one workgroup
// per CU walking the work items; bit-identical to the one-wavefront-per-tile form, chosen by the launch size.
#include "gps_common.hpp"

#include <cstdlib>

namespace {

typedef float f32x4 __attribute__((ext_vector_type(4)));
constexpr int DH = 64;    // dim_head of performer_pytorch.SelfAttention (performer_layer.py:427)
constexpr int MT = 17;    // feature tiles of 16 -> m <= 272
constexpr int KPL = 16;   // contiguous floats per lane when contracting over a 64-wide dim
constexpr float FEPS = 1e-4f;

__device__ __forceinline__ f32x4 mfma16(float a, float b, f32x4 c) {
  return __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, c, 0, 0, 0);
}
__device__ __forceinline__ f32x4 zero4() { return (f32x4){0.f, 0.f, 0.f, 0.f}; }
__device__ __forceinline__ float group_max(float v) {
  v = fmaxf(v, __shfl_xor(v, 16));
  return fmaxf(v, __shfl_xor(v, 32));
}
__device__ __forceinline__ float group_sum(float v) {
  v += __shfl_xor(v, 16);
  return v + __shfl_xor(v, 32);
}
__device__ __forceinline__ float wave_sum(float v) {
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o));
  return v;
}

// 16 contiguous floats of a row starting at column 16*grp (zero if !ok), scaled.  `p` must be a VALID address whatever
// `ok` is (callers clamp the row index): the loads are unconditional and the zeros come from the scale -- a load under a
// condition compiles to a branch around it, which serialises the issue of the row's four loads and of whatever follows
// (round 4, GPS_FAVOR A/B in DESIGN 4.5).
__device__ __forceinline__ void load_row16(const float* __restrict__ p, bool ok, float scale,
                                           float (&dst)[KPL]) {
  const float4* q = reinterpret_cast<const float4*>(p);
  const float sc = ok ? scale : 0.0f;
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 v = q[s];
    dst[4 * s + 0] = v.x * sc; dst[4 * s + 1] = v.y * sc;
    dst[4 * s + 2] = v.z * sc; dst[4 * s + 3] = v.w * sc;
  }
}
// 16 contiguous floats of a row of the staged projection (csrc: stage_projection)
__device__ __forceinline__ void lds_row16(const float* p, float (&dst)[KPL]) {
  const float4* q = reinterpret_cast<const float4*>(p);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 v = q[s];
    dst[4 * s + 0] = v.x; dst[4 * s + 1] = v.y; dst[4 * s + 2] = v.z; dst[4 * s + 3] = v.w;
  }
}
__device__ __forceinline__ int clampi(int v, int hi) { return v < hi ? v : hi; }
__device__ __forceinline__ f32x4 mm_rows(const float (&a)[KPL], const float (&b)[KPL], f32x4 c) {
#pragma unroll
  for (int s = 0; s < KPL; ++s) c = mfma16(a[s], b[s], c);
  return c;
}
__device__ __forceinline__ float sumsq16(const float (&a)[KPL]) {
  float s = 0.0f;
#pragma unroll
  for (int i = 0; i < KPL; ++i) s += a[i] * a[i];
  return s;
}

// ordered encoding so that unsigned atomicMax == float max (memset 0 is below every float)
__device__ __forceinline__ uint32_t enc_f32(float f) {
  const uint32_t u = __float_as_uint(f);
  return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
__device__ __forceinline__ float dec_f32(uint32_t e) {
  return __uint_as_float((e & 0x80000000u) ? (e & 0x7FFFFFFFu) : ~e);
}

// ---- round 5: the projection staged in LDS (LP = true) ------------------------------------------------------------------
// The four per-tile kernels (key max, output, query side and key side of the backward) contract every 16-row tile with the
// whole projection P [m, 64] -- 70 KB that every wavefront pulled through L2 once or twice per tile (6,400 tiles at code2
// sizes: 0.9-1.3 GB per launch), in a chain load -> 16-step contraction -> exp -> contraction per feature tile.  With LP the
// launch is ONE workgroup per CU (12-16 wavefronts) that copies P into LDS once -- rows past m zero, row pitch 68 floats:
// conflict-free for the one-float-per-lane column reads (rows 4 grp + r: 68 x 4 = 16 banks apart); the 16-floats-of-a-row
// reads are NOT quite (measured, profiles/r05_pmc_favor.txt: SQ_LDS_BANK_CONFLICT ~1.5 cycles per LDS instruction in the
// query-side kernel = 3 % of its wave cycles, 0.7 % spent waiting on LDS): a ds_read_b128 is served in four NON-contiguous
// 16-lane groups ({0-3, 12-15, 20-27}, ...) that mix two column blocks, which at this pitch puts half of a group's lanes two
// to a slot.  The planes layout of the staged context kernels below ([column block][row][20 floats]) avoids that; P was
// left as it is for the 1-2 % it would buy -- and then walks the work items, wavefront w taking items
// w, w + W, w + 2 W, ...  Same loads of everything else, same arithmetic in the same order: bit-identical results.
constexpr int PP = 68;                          // LDS row pitch of the staged projection (floats)
constexpr int P_LDS_BYTES = 16 * MT * PP * 4;   // 73,984
// threads of the one-workgroup-per-CU launches: what each kernel's registers allow per SIMD (key max 4 wavefronts; query
// side 1 at 373 registers -- pinned to 2 per SIMD it spills and runs the same, 253 vs 259 us; key side 3 at 137).  The
// OUTPUT kernel is not launched in this form: its item loop needs 333 registers, one wavefront per SIMD instead of two,
// and it ran at 232 us against 149 (profiles/r05_favor_lds_projection.txt) -- the template parameter stays for the record.
constexpr int FM_THREADS = 1024, FO_THREADS = 256, FQ_THREADS = 256, FK_THREADS = 768;
__device__ __forceinline__ void stage_projection(float* __restrict__ lp, const float* __restrict__ P, int m) {
  for (int idx = threadIdx.x; idx < 16 * MT * (DH / 4); idx += blockDim.x) {
    const int row = idx / (DH / 4), c4 = idx % (DH / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < m) v = reinterpret_cast<const float4*>(P + (int64_t)row * DH)[c4];
    *reinterpret_cast<float4*>(lp + row * PP + 4 * c4) = v;
  }
  __syncthreads();
}

struct Seg {
  int g, h, row0, n0, n1, i, grp;
  bool valid;
};
__device__ __forceinline__ Seg tile_work(const int32_t* ptr, const int32_t* tile_graph,
                                         const int32_t* tile_row0, int64_t n_work, int H, int64_t w) {
  Seg s;
  s.valid = false;
  const int lane = threadIdx.x & 63;
  s.i = lane & 15;
  s.grp = lane >> 4;
  if (w >= n_work) return s;
  const int64_t tile = w / H;
  s.h = (int)(w - tile * H);
  s.g = tile_graph[tile];
  if (s.g < 0) return s;
  s.row0 = tile_row0[tile];
  s.n0 = ptr[s.g];
  s.n1 = ptr[s.g + 1];
  s.valid = true;
  return s;
}

// global key max incl. the zero logits of padded rows (only when the graph is shorter than Nmax)
__device__ __forceinline__ float key_max_M(const unsigned long long* kmax, int gh, int pad) {
  const float mr = dec_f32((uint32_t)(kmax[gh] >> 32));
  return pad > 0 ? fmaxf(mr, 0.0f) : mr;
}

// ---------------------------------------------------------------------------------------------
// F1: per (key tile, head): max over (key, feature) of dd_k, with its position, via atomicMax.
// ---------------------------------------------------------------------------------------------
template <bool LP>
__global__ __launch_bounds__(LP ? FM_THREADS : 256) void k_favor_kmax(
    const float* __restrict__ qkv, int64_t ld, const float* __restrict__ P, int m, float c,
    const int32_t* __restrict__ ptr, const int32_t* __restrict__ tile_graph,
    const int32_t* __restrict__ tile_row0, int64_t n_work, int H,
    unsigned long long* __restrict__ kmax) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  const float* Pp = P;
  if constexpr (LP) {
    stage_projection(lds_proj, P, m);
    Pp = lds_proj;
  }
  constexpr int pp = LP ? PP : DH;
  const int wpb = blockDim.x >> 6;
  for (int64_t w = blockIdx.x * (int64_t)wpb + (threadIdx.x >> 6); w < n_work; w = LP ? w + (int64_t)gridDim.x * wpb : n_work) {
  // (LP: the staged projection is the same for every work item -- without this the compiler hoists its reads out of the
  // loop, i.e. tries to keep all of P in registers.  !LP: the launch has one wavefront per item, the loop runs once.)
  if constexpr (LP) asm volatile("" ::: "memory");
  const Seg s = tile_work(ptr, tile_graph, tile_row0, n_work, H, w);
  if (!s.valid) continue;
  const int inner = H * DH;
  const int krow = s.row0 + s.i;
  float kv[KPL];
  load_row16(qkv + (int64_t)clampi(krow, s.n1 - 1) * ld + inner + s.h * DH + 16 * s.grp, krow < s.n1, c, kv);
  float best = -INFINITY;
  uint32_t best_idx = 0;
  for (int mt = 0; mt < MT; ++mt) {
    const int prow = mt * 16 + s.i;
    float pv[KPL];
    load_row16(Pp + clampi(prow, m - 1) * pp + 16 * s.grp, prow < m, 1.0f, pv);
    const f32x4 dd = mm_rows(pv, kv, zero4());  // [feature 4g+r][key l&15]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = mt * 16 + 4 * s.grp + r;
      if (f < m && krow < s.n1 && dd[r] > best) {
        best = dd[r];
        best_idx = (uint32_t)(krow - s.n0) * 272u + (uint32_t)f;
      }
    }
  }
  // wave arg-max (ties -> smallest index), then one 64-bit atomicMax per wave
  unsigned long long key = ((unsigned long long)enc_f32(best) << 32) | (0xFFFFFFFFu - best_idx);
  for (int o = 32; o > 0; o >>= 1) {
    const unsigned long long other = __shfl_xor(key, o);
    key = other > key ? other : key;
  }
  if ((threadIdx.x & 63) == 0) atomicMax(&kmax[s.g * H + s.h], key);
  }
}

// ---------------------------------------------------------------------------------------------
// F2: per (graph, head, feature tile): ctx[m][64] = phi_k^T v and ksum[m] over all keys.
// ---------------------------------------------------------------------------------------------
// Round 4: the rows of a graph are dealt to S wavefronts (slices of whole 16-row blocks).  One wavefront per (graph, head,
// feature tile) walked all ~800 rows of a code2 graph alone -- 2,176 wavefronts on 1,024 SIMDs, every iteration a chain of
// dependent loads: the kernel ran at 1/7 of its MFMA time.  With S slices there are enough wavefronts in flight to cover
// the loads; the partial sums go to a workspace and k_favor_sum_parts adds them in slice order (deterministic).
__global__ __launch_bounds__(256) void k_favor_ctx(
    const float* __restrict__ qkv, int64_t ld, const float* __restrict__ P, int m, float c,
    float ratio, const int32_t* __restrict__ ptr, const int32_t* __restrict__ nmax_dev, int64_t B,
    int H, const unsigned long long* __restrict__ kmax, float* __restrict__ ctx,
    float* __restrict__ ksum, int S) {
  const int lane = threadIdx.x & 63;
  int64_t w = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= B * H * MT * S) return;
  const int sl = (int)(w % S);
  w /= S;
  const int mt = (int)(w % MT);
  const int gh = (int)(w / MT);
  const int g = gh / H, h = gh - g * H;
  const int n0 = ptr[g], n1_all = ptr[g + 1];
  const int per = ((n1_all - n0 + 15) / 16 + S - 1) / S * 16;       // rows per slice (whole blocks)
  const int k0 = min(n1_all, n0 + sl * per), n1 = min(n1_all, k0 + per);
  if (S > 1) {                                                      // this slice's partial records
    ctx += (int64_t)sl * B * H * 272 * DH;
    ksum += (int64_t)sl * B * H * 272;
  }
  const int i = lane & 15, grp = lane >> 4;
  const int inner = H * DH;
  const int pad = max(nmax_dev[0] - (n1_all - n0), 0);     // (>= 0: a dead graph of a padded batch may be longer than Nmax)
  const float M = key_max_M(kmax, gh, pad);
  const int f = mt * 16 + i;  // this lane's feature column
  float pv[KPL];
  load_row16(P + (int64_t)clampi(f, m - 1) * DH + 16 * grp, f < m, 1.0f, pv);
  f32x4 acc[4];
#pragma unroll
  for (int et = 0; et < 4; ++et) acc[et] = zero4();
  float ks = 0.0f;
  for (int kb = k0; kb < n1; kb += 16) {
    const int krow = kb + i;
    float kv[KPL];
    load_row16(qkv + (int64_t)clampi(krow, n1 - 1) * ld + inner + h * DH + 16 * grp, krow < n1, c, kv);
    const float nrm = group_sum(sumsq16(kv));       // |c k|^2 of row (l&15)
    f32x4 dd = mm_rows(kv, pv, zero4());            // [key 4g+r][feature l&15]
    f32x4 phi;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int key = kb + 4 * grp + r;
      const float dg = 0.5f * __shfl(nrm, 4 * grp + r);
      const bool ok = key < n1 && f < m;
      phi[r] = ok ? ratio * (expf(dd[r] - dg - M) + FEPS) : 0.0f;
      ks += phi[r];
    }
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int key = kb + 4 * grp + r;
        // (clamped address; rows past the slice carry phi = 0, so their V value needs no select)
        const float vv = qkv[(int64_t)clampi(key, n1 - 1) * ld + 2 * inner + h * DH + et * 16 + i];
        acc[et] = mfma16(vv, phi[r], acc[et]);      // ctx^T[e 4g+r'][feature l&15]
      }
  }
  ks = group_sum(ks);
  if (f < m) {
    float* cp = ctx + ((int64_t)gh * 272 + f) * DH;
#pragma unroll
    for (int et = 0; et < 4; ++et)
      *reinterpret_cast<float4*>(cp + et * 16 + 4 * grp) =
          make_float4(acc[et][0], acc[et][1], acc[et][2], acc[et][3]);
    // (the padded rows' closed-form term: here when the graph is one slice, else by k_favor_sum_parts)
    if (grp == 0) ksum[(int64_t)gh * 272 + f] = S > 1 ? ks : fmaf((float)pad, ratio * (expf(-M) + FEPS), ks);
  }
}

// ---- round 5: the two context kernels with their rows staged in LDS (k_favor_ctx_st / k_favor_bwd_ctx_st) ---------------
// One wavefront per (graph, head, feature tile, slice) meant that the 17 feature-tile wavefronts of a (graph, head, slice)
// -- spread over 17 workgroups -- each pulled the same key and value rows: 420 MB fetched per launch for 52 MB of rows
// (rocprofv3 FETCH_SIZE, profiles/r05_pmc_favor.txt), ~3.3 TB/s at the fabric.  Here ONE workgroup of W wavefronts (12; 9 compiled for the A/B) owns a
// (graph, head, slice): every step its threads copy the next 16-row block of the operands into LDS (double-buffered, one
// barrier per step) and wavefront j runs feature tiles j and j + W against the staged block.  Layouts: the operand that
// is read 16-floats-of-a-row per lane (keys / queries) as four planes [column block][row][20 floats] -- the four 16-lane
// groups of a ds_read_b128 then each see 16 different rows at 5 slots apart: conflict-free --, the operand that is read
// one float per lane (values / g_out) as [row][68 floats].  Rows past the slice are copies of its last row, as the
// clamped loads of the per-wavefront kernels read them.  Per record the same arithmetic in the same row order: the
// partial records -- and everything downstream -- are bit-identical.
constexpr int CS_RP = 20;                       // row pitch inside a plane (floats)
constexpr int CS_PLANE = 16 * CS_RP;            // 320 floats per column block
constexpr int CS_A = 4 * CS_PLANE;              // 1280: the row16 operand of one block
constexpr int CS_B = 16 * PP;                   // 1088: the one-float-per-lane operand
constexpr int CS_X = 64;                        // per-row scalars (backward: mq, D, gD; 16 each)
constexpr int CS_BUF = CS_A + CS_B + CS_X;      // floats per stage buffer
constexpr int CS_LDS_BYTES = 2 * CS_BUF * 4;

__device__ __forceinline__ void cs_row16(const float* a_plane, int i, int grp, float sc, float (&dst)[KPL]) {
  const float4* q = reinterpret_cast<const float4*>(a_plane + grp * CS_PLANE + i * CS_RP);
#pragma unroll
  for (int s = 0; s < 4; ++s) {
    const float4 v = q[s];
    dst[4 * s + 0] = v.x * sc; dst[4 * s + 1] = v.y * sc;
    dst[4 * s + 2] = v.z * sc; dst[4 * s + 3] = v.w * sc;
  }
}
// thread t of the first 512 copies one float4: t < 256 -> operand A (row (t >> 4) & 15, 16-byte column t & 15),
// else operand B
// (threads past 511 load a valid address too and drop it: an unconditional load)
__device__ __forceinline__ float4 cs_load(const float* __restrict__ a_src, int64_t a_ld, const float* __restrict__ b_src,
                                          int64_t b_ld, int row0, int n1) {
  const int t = threadIdx.x;
  const int row = clampi(row0 + ((t >> 4) & 15), n1 - 1), c4 = t & 15;
  const float* src = (t & 256) == 0 ? a_src + (int64_t)row * a_ld : b_src + (int64_t)row * b_ld;
  return reinterpret_cast<const float4*>(src)[c4];
}
__device__ __forceinline__ void cs_store(float* buf, float4 v) {
  const int t = threadIdx.x;
  if (t < 512) {
    const int row = (t >> 4) & 15, c4 = t & 15;
    float* dst = t < 256 ? buf + (c4 >> 2) * CS_PLANE + row * CS_RP + 4 * (c4 & 3) : buf + CS_A + row * PP + 4 * c4;
    *reinterpret_cast<float4*>(dst) = v;
  }
}

template <int CS_WAVES>
__global__ __launch_bounds__(64 * CS_WAVES) void k_favor_ctx_st(
    const float* __restrict__ qkv, int64_t ld, const float* __restrict__ P, int m, float c,
    float ratio, const int32_t* __restrict__ ptr, const int32_t* __restrict__ nmax_dev, int64_t B,
    int H, const unsigned long long* __restrict__ kmax, float* __restrict__ ctx,
    float* __restrict__ ksum, int S) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x;                                   // (graph-head, slice), slice fastest
  const int sl = u % S, gh = u / S;
  const int g = gh / H, h = gh - g * H;
  const int n0 = ptr[g], n1_all = ptr[g + 1];
  const int per = ((n1_all - n0 + 15) / 16 + S - 1) / S * 16;       // rows per slice (whole blocks), as k_favor_ctx
  const int k0 = min(n1_all, n0 + sl * per), n1 = min(n1_all, k0 + per);
  if (S > 1) {
    ctx += (int64_t)sl * B * H * 272 * DH;
    ksum += (int64_t)sl * B * H * 272;
  }
  const int i = lane & 15, grp = lane >> 4;
  const int inner = H * DH;
  const int pad = max(nmax_dev[0] - (n1_all - n0), 0);
  const float M = key_max_M(kmax, gh, pad);
  const int nt = wave + CS_WAVES < MT ? 2 : 1;                // feature tiles of this wavefront: wave, wave + W
  float pv[2][KPL];
  f32x4 acc[2][4];
  float ks[2] = {0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int f = (wave + t * CS_WAVES) * 16 + i;
    load_row16(P + (int64_t)clampi(f, m - 1) * DH + 16 * grp, f < m && t < nt, 1.0f, pv[t]);
#pragma unroll
    for (int et = 0; et < 4; ++et) acc[t][et] = zero4();
  }
  const float* ksrc = qkv + inner + h * DH;
  const float* vsrc = qkv + 2 * inner + h * DH;
  // three row blocks in flight through registers (a block's loads are issued two and a half steps before it is written to
  // LDS).  Measured the same as one block in flight (119 us): a step's ~4.7 us are its arithmetic -- per SIMD 4-5 feature
  // tiles x (a 16-long dependent 16x16x4 chain + 16 more + ~200 vector instructions), serialised by the barrier per step
  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0, nx2 = nx0;
  if (k0 < n1) {
    nx0 = cs_load(ksrc, ld, vsrc, ld, k0, n1);
    cs_store(lds_proj, nx0);
  }
  if (k0 + 16 < n1) nx0 = cs_load(ksrc, ld, vsrc, ld, k0 + 16, n1);
  if (k0 + 32 < n1) nx1 = cs_load(ksrc, ld, vsrc, ld, k0 + 32, n1);
  __syncthreads();
  int step = 0;
  for (int kb = k0; kb < n1; kb += 16, ++step) {
    const float* buf = lds_proj + (step & 1) * CS_BUF;
    if (kb + 48 < n1) nx2 = cs_load(ksrc, ld, vsrc, ld, kb + 48, n1);
    float kv[KPL];
    cs_row16(buf, i, grp, kb + i < n1 ? c : 0.0f, kv);
    const float nrm = group_sum(sumsq16(kv));       // |c k|^2 of row (l&15)
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < nt) {
        const int f = (wave + t * CS_WAVES) * 16 + i;
        f32x4 dd = mm_rows(kv, pv[t], zero4());     // [key 4g+r][feature l&15]
        f32x4 phi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int key = kb + 4 * grp + r;
          const float dg = 0.5f * __shfl(nrm, 4 * grp + r);
          const bool ok = key < n1 && f < m;
          phi[r] = ok ? ratio * (expf(dd[r] - dg - M) + FEPS) : 0.0f;
          ks[t] += phi[r];
        }
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float vv = buf[CS_A + (4 * grp + r) * PP + et * 16 + i];
            acc[t][et] = mfma16(vv, phi[r], acc[t][et]);      // ctx^T[e 4g+r'][feature l&15]
          }
      }
    }
    if (kb + 16 < n1) cs_store(lds_proj + ((step + 1) & 1) * CS_BUF, nx0);
    __syncthreads();
    nx0 = nx1;
    nx1 = nx2;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < nt) {
      const int f = (wave + t * CS_WAVES) * 16 + i;
      const float kt = group_sum(ks[t]);
      if (f < m) {
        float* cp = ctx + ((int64_t)gh * 272 + f) * DH;
#pragma unroll
        for (int et = 0; et < 4; ++et)
          *reinterpret_cast<float4*>(cp + et * 16 + 4 * grp) =
              make_float4(acc[t][et][0], acc[t][et][1], acc[t][et][2], acc[t][et][3]);
        if (grp == 0) ksum[(int64_t)gh * 272 + f] = S > 1 ? kt : fmaf((float)pad, ratio * (expf(-M) + FEPS), kt);
      }
    }
  }
}

// out[j] = sum_s part[s][j] over the S row slices, in slice order; ksum additionally takes the padded rows' term
// pad * r (exp(-M) + 1e-4) (forward only: kmax != nullptr).
__global__ __launch_bounds__(256) void k_favor_sum_parts(const float* __restrict__ cpart, const float* __restrict__ kpart,
                                                         int S, int64_t BH, int m, float ratio,
                                                         const int32_t* __restrict__ ptr,
                                                         const int32_t* __restrict__ nmax_dev, int H,
                                                         const unsigned long long* __restrict__ kmax,
                                                         float* __restrict__ cout, float* __restrict__ kout) {
  const int64_t nc = BH * 272 * (DH / 4), nk = BH * 272;
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  if (t < nc) {
    f32x4 a = zero4();
    for (int sl = 0; sl < S; ++sl) a += *reinterpret_cast<const f32x4*>(cpart + (sl * nc + t) * 4);
    *reinterpret_cast<f32x4*>(cout + t * 4) = a;
  } else if (t < nc + nk) {
    const int64_t j = t - nc;
    float a = 0.0f;
    for (int sl = 0; sl < S; ++sl) a += kpart[sl * nk + j];
    if (kmax) {
      const int gh = (int)(j / 272), g = gh / H;
      const int pad = max(nmax_dev[0] - (ptr[g + 1] - ptr[g]), 0);
      a += (float)pad * (ratio * (expf(-key_max_M(kmax, gh, pad)) + FEPS));
    }
    kout[j] = a;
  }
}

// dd_q^T tiles [feature 4g+r + 16mt][query l&15] and phi_q for one 16-query tile
template <int PITCH, bool FENCE = true>
__device__ __forceinline__ void query_features(const float (&qv)[KPL], const float* __restrict__ P,
                                               int m, int i, int grp, float (&dd)[MT][4], float& mq,
                                               bool have_mq) {
  float mx = -INFINITY;
  if constexpr (PITCH == PP) {
    // staged projection: rows past m are zero (no clamp, no select); the reads of tile mt + 1 are issued in front of the
    // contraction of tile mt and nothing moves across the fence -- left alone the scheduler hoists all 68 reads to the top
    // (272 registers of projection rows)
    float pv[2][KPL];
    lds_row16(P + i * PP + 16 * grp, pv[0]);
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      if (mt + 1 < MT) lds_row16(P + ((mt + 1) * 16 + i) * PP + 16 * grp, pv[(mt + 1) & 1]);
      const f32x4 t = mm_rows(pv[mt & 1], qv, zero4());
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dd[mt][r] = t[r];
        if (mt * 16 + 4 * grp + r < m) mx = fmaxf(mx, t[r]);
      }
      if constexpr (FENCE) __builtin_amdgcn_sched_barrier(0);
    }
  } else {
#pragma unroll
    for (int mt = 0; mt < MT; ++mt) {
      const int prow = mt * 16 + i;
      float pv[KPL];
      load_row16(P + clampi(prow, m - 1) * PITCH + 16 * grp, prow < m, 1.0f, pv);
      const f32x4 t = mm_rows(pv, qv, zero4());
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        dd[mt][r] = t[r];
        if (mt * 16 + 4 * grp + r < m) mx = fmaxf(mx, t[r]);
      }
    }
  }
  if (!have_mq) mq = group_max(mx);
}

// ---------------------------------------------------------------------------------------------
// F3: per (query tile, head): out = (phi_q ctx) / (phi_q . ksum); saves mq (row max) and D.
// ---------------------------------------------------------------------------------------------
// One work item of the output kernel (see favor_bwd_q_item for the pitch parameters).
template <int PPITCH, int CPITCH, int KSTRIDE>
__device__ __forceinline__ void favor_out_item(const Seg& s, const float* __restrict__ qkv, int64_t ld, const float* Pp, int m,
                                               float c, float ratio, int64_t N, int H, const float* cbase, const float* kbase,
                                               float* __restrict__ out, float* __restrict__ mq_out, float* __restrict__ D_out) {
  const int inner = H * DH;
  const int qrow = s.row0 + s.i;
  const bool q_ok = qrow < s.n1;
  float qv[KPL];
  load_row16(qkv + (int64_t)clampi(qrow, s.n1 - 1) * ld + s.h * DH + 16 * s.grp, q_ok, c, qv);
  const float half_nrm = 0.5f * group_sum(sumsq16(qv));
  float dd[MT][4];
  float mq;
  query_features<PPITCH>(qv, Pp, m, s.i, s.grp, dd, mq, false);
  float dpart = 0.0f;
  f32x4 acc[4];
#pragma unroll
  for (int et = 0; et < 4; ++et) acc[et] = zero4();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = mt * 16 + 4 * s.grp + r;
      const float phi = f < m ? ratio * (expf(dd[mt][r] - half_nrm - mq) + FEPS) : 0.0f;
      dd[mt][r] = phi;
      dpart += phi * kbase[clampi(f, m - 1) * KSTRIDE];          // (phi = 0 past m)
    }
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = mt * 16 + 4 * s.grp + r;
        const float cv = cbase[clampi(f, m - 1) * CPITCH + et * 16 + s.i];      // (dd = phi = 0 past m)
        acc[et] = mfma16(cv, dd[mt][r], acc[et]);   // num^T[e 4g+r'][query l&15]
      }
  }
  const float D = group_sum(dpart);
  if (q_ok) {
    const float inv = 1.0f / D;
    float* o = out + (int64_t)qrow * inner + s.h * DH;
#pragma unroll
    for (int et = 0; et < 4; ++et)
      *reinterpret_cast<float4*>(o + et * 16 + 4 * s.grp) =
          make_float4(acc[et][0] * inv, acc[et][1] * inv, acc[et][2] * inv, acc[et][3] * inv);
    if (s.grp == 0) {
      mq_out[(int64_t)s.h * N + qrow] = mq;
      D_out[(int64_t)s.h * N + qrow] = D;
    }
  }
}

template <bool LP>
__global__ __launch_bounds__(LP ? FO_THREADS : 256) void k_favor_out(
    const float* __restrict__ qkv, int64_t ld, const float* __restrict__ P, int m, float c,
    float ratio, const int32_t* __restrict__ ptr, const int32_t* __restrict__ tile_graph,
    const int32_t* __restrict__ tile_row0, int64_t n_work, int64_t N, int H,
    const float* __restrict__ ctx, const float* __restrict__ ksum, float* __restrict__ out,
    float* __restrict__ mq_out, float* __restrict__ D_out) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  const float* Pp = P;
  if constexpr (LP) {
    stage_projection(lds_proj, P, m);
    Pp = lds_proj;
  }
  constexpr int pp = LP ? PP : DH;
  const int wpb = blockDim.x >> 6;
  for (int64_t w = blockIdx.x * (int64_t)wpb + (threadIdx.x >> 6); w < n_work; w = LP ? w + (int64_t)gridDim.x * wpb : n_work) {
    if constexpr (LP) asm volatile("" ::: "memory");        // (see k_favor_bwd_q)
    const Seg s = tile_work(ptr, tile_graph, tile_row0, n_work, H, w);
    if (!s.valid) continue;
    const int gh = s.g * H + s.h;
    favor_out_item<pp, DH, 1>(s, qkv, ld, Pp, m, c, ratio, N, H, ctx + (int64_t)gh * 272 * DH, ksum + (int64_t)gh * 272, out,
                              mq_out, D_out);
  }
}

// ---------------------------------------------------------------------------------------------
// B0: gD[h][q] = -sum_e g_out*out / D  (16 lanes per (q, h), float4 each)
// ---------------------------------------------------------------------------------------------
__global__ void k_favor_bwd_gd(const float* __restrict__ g_out, const float* __restrict__ out,
                               const float* __restrict__ D, int64_t N, int H,
                               float* __restrict__ gD) {
  const int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
  const int64_t item = t >> 4;
  const int sub = (int)(t & 15);
  float acc = 0.0f;
  const bool ok = item < N * H;
  int64_t q = 0;
  int h = 0;
  if (ok) {
    q = item / H;
    h = (int)(item - q * H);
    const float4 a = *reinterpret_cast<const float4*>(g_out + q * (int64_t)(H * DH) + h * DH + 4 * sub);
    const float4 b = *reinterpret_cast<const float4*>(out + q * (int64_t)(H * DH) + h * DH + 4 * sub);
    acc = a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w;
  }
  for (int o = 8; o > 0; o >>= 1) acc += __shfl_xor(acc, o);
  if (ok && sub == 0) gD[(int64_t)h * N + q] = -acc / D[(int64_t)h * N + q];
}

// ---------------------------------------------------------------------------------------------
// B1: per (query tile, head): g_q
// ---------------------------------------------------------------------------------------------
// One work item.  Pp: the projection with row pitch PPITCH (global: DH, staged: PP); cbase: the context record of the item's
// (graph, head) with row pitch CPITCH; ksum[f] = kbase[f * KSTRIDE].
template <int PPITCH, int CPITCH, int KSTRIDE, bool FENCE = true>
__device__ __forceinline__ void favor_bwd_q_item(
    const Seg& s, const float* __restrict__ g_out, const float* __restrict__ qkv, int64_t ld, const float* Pp, int m,
    float c, float ratio, int64_t N, int H, const float* cbase, const float* kbase, const float* __restrict__ mq_in,
    const float* __restrict__ D_in, const float* __restrict__ gD_in, float* __restrict__ d_qkv, int64_t ldg) {
  const int inner = H * DH;
  const int qrow = s.row0 + s.i;
  const bool q_ok = qrow < s.n1;
  float qv[KPL], gn[KPL];
  const int qrc = clampi(qrow, s.n1 - 1);             // (valid addresses; the row's results are dropped when !q_ok)
  load_row16(qkv + (int64_t)qrc * ld + s.h * DH + 16 * s.grp, q_ok, c, qv);
  const float half_nrm = 0.5f * group_sum(sumsq16(qv));
  const float mq_l = mq_in[(int64_t)s.h * N + qrc], D_l = D_in[(int64_t)s.h * N + qrc], gD_l = gD_in[(int64_t)s.h * N + qrc];
  float mq = q_ok ? mq_l : 0.0f;
  const float Dq = q_ok ? D_l : 1.0f;
  const float gDq = q_ok ? gD_l : 0.0f;
  load_row16(g_out + (int64_t)qrc * inner + s.h * DH + 16 * s.grp, q_ok, 1.0f / Dq, gn);  // g_num
  float dd[MT][4];
  query_features<PPITCH, FENCE>(qv, Pp, m, s.i, s.grp, dd, mq, true);
  float s1 = 0.0f;
  float gA[MT][4];
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
    const int crow = mt * 16 + s.i;
    float cv[KPL];
    load_row16(cbase + clampi(crow, m - 1) * CPITCH + 16 * s.grp, crow < m, 1.0f, cv);
    const f32x4 gphi = mm_rows(cv, gn, zero4());   // [feature 4g+r][query]: sum_e ctx[f][e] g_num[q][e]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = mt * 16 + 4 * s.grp + r;
      const float ex = ratio * expf(dd[mt][r] - half_nrm - mq);     // = phi - r*eps
      const float ga = f < m ? (gphi[r] + kbase[clampi(f, m - 1) * KSTRIDE] * gDq) * ex : 0.0f;
      gA[mt][r] = ga;
      s1 += ga;
    }
    if constexpr (CPITCH == PP && FENCE) __builtin_amdgcn_sched_barrier(0);      // (staged record: as in query_features)
  }
  s1 = group_sum(s1);                                 // sum_m g_A of this query
  f32x4 acc[4];
#pragma unroll
  for (int dt = 0; dt < 4; ++dt) acc[dt] = zero4();
#pragma unroll
  for (int mt = 0; mt < MT; ++mt) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = mt * 16 + 4 * s.grp + r;
      // the row max is subtracted inside the exponent: its gradient (-s1) lands on the arg-max
      if (f < m && dd[mt][r] == mq) gA[mt][r] -= s1;
    }
#pragma unroll
    for (int dt = 0; dt < 4; ++dt)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = mt * 16 + 4 * s.grp + r;
        const float pv = Pp[clampi(f, m - 1) * PPITCH + dt * 16 + s.i];      // (gA = 0 past m)
        acc[dt] = mfma16(pv, gA[mt][r], acc[dt]);   // g_q^T[dh 4g+r'][query]
      }
  }
  if (q_ok) {
    const float* qsrc = qkv + (int64_t)qrow * ld + s.h * DH;
    float* o = d_qkv + (int64_t)qrow * ldg + s.h * DH;
    const float k2 = -s1 * c * c;                     // d(diag)/dq = c^2 q, g_diag = -s1
#pragma unroll
    for (int dt = 0; dt < 4; ++dt) {
      const int col = dt * 16 + 4 * s.grp;
      const float4 qq = *reinterpret_cast<const float4*>(qsrc + col);
      *reinterpret_cast<float4*>(o + col) =
          make_float4(fmaf(k2, qq.x, c * acc[dt][0]), fmaf(k2, qq.y, c * acc[dt][1]),      // (explicit: left to the
                      fmaf(k2, qq.z, c * acc[dt][2]), fmaf(k2, qq.w, c * acc[dt][3]));     // compiler, LP / !LP contracted differently)
    }
  }
}

template <bool LP>
__global__ __launch_bounds__(LP ? FQ_THREADS : 256) void k_favor_bwd_q(
    const float* __restrict__ g_out, const float* __restrict__ qkv, int64_t ld,
    const float* __restrict__ P, int m, float c, float ratio, const int32_t* __restrict__ ptr,
    const int32_t* __restrict__ tile_graph, const int32_t* __restrict__ tile_row0, int64_t n_work,
    int64_t N, int H, const float* __restrict__ ctx, const float* __restrict__ ksum,
    const float* __restrict__ mq_in, const float* __restrict__ D_in, const float* __restrict__ gD_in,
    float* __restrict__ d_qkv, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  const float* Pp = P;
  if constexpr (LP) {
    stage_projection(lds_proj, P, m);
    Pp = lds_proj;
  }
  constexpr int pp = LP ? PP : DH;
  const int wpb = blockDim.x >> 6;
  for (int64_t w = blockIdx.x * (int64_t)wpb + (threadIdx.x >> 6); w < n_work; w = LP ? w + (int64_t)gridDim.x * wpb : n_work) {
    // (LP: the staged projection is the same for every work item -- without this the compiler hoists its reads out of the
    // loop, i.e. tries to keep all of P in registers.  !LP: the launch has one wavefront per item, the loop runs once.)
    if constexpr (LP) asm volatile("" ::: "memory");
    const Seg s = tile_work(ptr, tile_graph, tile_row0, n_work, H, w);
    if (!s.valid) continue;
    const int gh = s.g * H + s.h;
    favor_bwd_q_item<pp, DH, 1>(s, g_out, qkv, ld, Pp, m, c, ratio, N, H, ctx + (int64_t)gh * 272 * DH,
                                ksum + (int64_t)gh * 272, mq_in, D_in, gD_in, d_qkv, ldg);
  }
}

// ---- the context record staged as well (LC) -----------------------------------------------------------------------------
// What the LP form still waits for is the context record of the item's (graph, head): 70 KB read row by row from global
// memory in a chain load -> 16-step contraction per feature tile, at ONE wavefront per SIMD (the kernel holds ~370
// registers) -- ~17 exposed memory round trips per work item, 98k cycles per item against 26k of matrix pipe.  Here a
// workgroup (4 wavefronts, one per CU) takes CHUNKS = (graph, head, four consecutive 16-row tiles of that graph): it copies
// the record [m, 64] and ksum [m] of (graph, head) into LDS next to the projection (pitch 68: the record's row in columns
// 0..63, ksum in column 64) with all 256 threads, then every wavefront runs its tile against LDS only.  Chunks are numbered
// (graph, quad, head), head fastest, and dealt to the workgroups in contiguous ranges; the graph of a chunk comes from a
// prefix of quads per graph that every workgroup builds in LDS (B <= kLcMaxGraphs).  Same arithmetic per item, same order.
constexpr int kLcMaxGraphs = 1024;
constexpr int LC_LDS_BYTES = 2 * P_LDS_BYTES + (kLcMaxGraphs + 1) * 4;
__device__ __forceinline__ void stage_record(float* __restrict__ lc, const float* __restrict__ rec, const float* __restrict__ vec,
                                             int m) {
  for (int idx = threadIdx.x; idx < 16 * MT * (DH / 4); idx += blockDim.x) {
    const int row = idx / (DH / 4), c4 = idx % (DH / 4);
    float4 v = make_float4(0.f, 0.f, 0.f, 0.f);
    if (row < m) v = reinterpret_cast<const float4*>(rec + (int64_t)row * DH)[c4];
    *reinterpret_cast<float4*>(lc + row * PP + 4 * c4) = v;
  }
  for (int row = threadIdx.x; row < 16 * MT; row += blockDim.x) lc[row * PP + DH] = row < m ? vec[row] : 0.0f;
}
struct Chunk {
  int g, h, q4;
};
// chunk ch of the launch -> (graph, head, quad of the graph); pre[g] = quads of the graphs before g (LDS)
__device__ __forceinline__ Chunk chunk_of(const int* pre, int B, int H, int ch) {
  const int cq = ch / H;
  int lo = 0, hi = B;
  while (hi - lo > 1) {
    const int mid = (lo + hi) >> 1;
    if (pre[mid] <= cq) lo = mid; else hi = mid;
  }
  return Chunk{lo, ch - cq * H, cq - pre[lo]};          // (the LAST graph whose prefix is <= cq: empty graphs are skipped)
}
// pre[g] = chunks of the graphs before g, pre[B] = all of them: counts by all threads, the prefix by the first wavefront
// (each lane its own run of graphs, the runs joined by a wave scan) -- B up to kLcMaxGraphs costs a few hundred cycles
__device__ __forceinline__ int quad_prefix(int* pre, const int32_t* __restrict__ ptr, int B, int nw) {
  for (int g = threadIdx.x; g < B; g += blockDim.x) pre[g + 1] = (((ptr[g + 1] - ptr[g]) + 15) / 16 + nw - 1) / nw;
  if (threadIdx.x == 0) pre[0] = 0;
  __syncthreads();
  if (threadIdx.x < 64) {
    const int lane = threadIdx.x, per = (B + 63) / 64;
    const int lo = min(lane * per, B), hi = min(lo + per, B);
    int sum = 0;
    for (int g = lo; g < hi; ++g) sum += pre[g + 1];
    int incl = sum;
    for (int o = 1; o < 64; o <<= 1) {
      const int v = __shfl_up(incl, o);
      if (lane >= o) incl += v;
    }
    int run = incl - sum;
    for (int g = lo; g < hi; ++g) {
      run += pre[g + 1];
      pre[g + 1] = run;
    }
  }
  __syncthreads();
  return pre[B];
}

// the same copy through registers (256 threads): issued for chunk c + 1 in front of chunk c's arithmetic, written behind it
struct RecordRegs {
  float4 v[16 * MT * (DH / 4) / 256];     // 17
  float k0, k1;
};
__device__ __forceinline__ void load_record(RecordRegs& r, const float* __restrict__ rec, const float* __restrict__ vec, int m) {
#pragma unroll
  for (int j = 0; j < 16 * MT * (DH / 4) / 256; ++j) {
    const int idx = threadIdx.x + 256 * j, row = idx / (DH / 4), c4 = idx % (DH / 4);
    const float4 v = reinterpret_cast<const float4*>(rec + (int64_t)clampi(row, m - 1) * DH)[c4];      // (unconditional load)
    r.v[j] = row < m ? v : make_float4(0.f, 0.f, 0.f, 0.f);
  }
  const int t = threadIdx.x;
  const float a = vec[clampi(t, m - 1)], b = vec[clampi(256 + t, m - 1)];
  r.k0 = t < m ? a : 0.0f;
  r.k1 = 256 + t < m ? b : 0.0f;
}
__device__ __forceinline__ void write_record(float* __restrict__ lc, const RecordRegs& r) {
#pragma unroll
  for (int j = 0; j < 16 * MT * (DH / 4) / 256; ++j) {
    const int idx = threadIdx.x + 256 * j, row = idx / (DH / 4), c4 = idx % (DH / 4);
    *reinterpret_cast<float4*>(lc + row * PP + 4 * c4) = r.v[j];
  }
  const int t = threadIdx.x;
  lc[t * PP + DH] = r.k0;
  if (256 + t < 16 * MT) lc[(256 + t) * PP + DH] = r.k1;
}

// The chunk loop shared by the three kernels: `item(s, lp, lc)` runs one wavefront's tile against the staged projection
// `lp` and the staged record `lc` (record rows in columns 0..63, the vector in column 64) of (s.g, s.h).
template <bool PRE, int NW, typename F>
__device__ __forceinline__ void lc_drive(float* lds, const float* __restrict__ P, int m, const int32_t* __restrict__ ptr, int B,
                                         int H, const float* __restrict__ rec, const float* __restrict__ vec, F&& item) {
  float* const lp = lds;
  float* const lc = lds + 16 * MT * PP;
  int* const pre = reinterpret_cast<int*>(lc + 16 * MT * PP);
  stage_projection(lp, P, m);
  const int n_chunks = quad_prefix(pre, ptr, B, NW) * H;
  const int per = (n_chunks + (int)gridDim.x - 1) / (int)gridDim.x;
  const int c0 = (int)blockIdx.x * per, c1 = min(n_chunks, c0 + per);
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  static_assert(!PRE || NW == 4, "the register copy is laid out for 256 threads");
  RecordRegs nx;
  if constexpr (PRE) {
    if (c0 < c1) {
      const Chunk k = chunk_of(pre, B, H, c0);
      const int gh = k.g * H + k.h;
      load_record(nx, rec + (int64_t)gh * 272 * DH, vec + (int64_t)gh * 272, m);
    }
  }
  for (int ch = c0; ch < c1; ++ch) {
    const Chunk k = chunk_of(pre, B, H, ch);
    const int gh = k.g * H + k.h;
    __syncthreads();                                  // the previous chunk's readers are done with the record
    if constexpr (PRE) {
      write_record(lc, nx);
    } else {
      stage_record(lc, rec + (int64_t)gh * 272 * DH, vec + (int64_t)gh * 272, m);
    }
    __syncthreads();
    if constexpr (PRE) {
      if (ch + 1 < c1) {                              // the next record's loads fly during this chunk's arithmetic
        const Chunk kn = chunk_of(pre, B, H, ch + 1);
        const int ghn = kn.g * H + kn.h;
        load_record(nx, rec + (int64_t)ghn * 272 * DH, vec + (int64_t)ghn * 272, m);
      }
    }
    Seg s;
    s.g = k.g; s.h = k.h; s.n0 = ptr[k.g]; s.n1 = ptr[k.g + 1];
    s.row0 = s.n0 + 16 * (NW * k.q4 + wave);
    s.i = lane & 15; s.grp = lane >> 4;
    s.valid = s.row0 < s.n1;
    if (s.valid) item(s, lp, lc);
  }
}

template <bool PRE, int NW>
__global__ __launch_bounds__(64 * NW) void k_favor_bwd_q_lc(
    const float* __restrict__ g_out, const float* __restrict__ qkv, int64_t ld,
    const float* __restrict__ P, int m, float c, float ratio, const int32_t* __restrict__ ptr, int B,
    int64_t N, int H, const float* __restrict__ ctx, const float* __restrict__ ksum,
    const float* __restrict__ mq_in, const float* __restrict__ D_in, const float* __restrict__ gD_in,
    float* __restrict__ d_qkv, int64_t ldg) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  lc_drive<PRE, NW>(lds_proj, P, m, ptr, B, H, ctx, ksum, [&](const Seg& s, const float* lp, const float* lc) {
    favor_bwd_q_item<PP, PP, PP, true>(s, g_out, qkv, ld, lp, m, c, ratio, N, H, lc, lc + DH, mq_in, D_in, gD_in, d_qkv, ldg);
  });
}

template <bool PRE, int NW>
__global__ __launch_bounds__(64 * NW) void k_favor_out_lc(
    const float* __restrict__ qkv, int64_t ld, const float* __restrict__ P, int m, float c, float ratio,
    const int32_t* __restrict__ ptr, int B, int64_t N, int H, const float* __restrict__ ctx,
    const float* __restrict__ ksum, float* __restrict__ out, float* __restrict__ mq_out, float* __restrict__ D_out) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  lc_drive<PRE, NW>(lds_proj, P, m, ptr, B, H, ctx, ksum, [&](const Seg& s, const float* lp, const float* lc) {
    favor_out_item<PP, PP, PP>(s, qkv, ld, lp, m, c, ratio, N, H, lc, lc + DH, out, mq_out, D_out);
  });
}

// ---------------------------------------------------------------------------------------------
// B2: per (graph, head, feature tile): g_ctx[m][64] = sum_q phi_q g_num, g_ksum[m] = sum_q gD phi_q
// ---------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void k_favor_bwd_ctx(
    const float* __restrict__ g_out, const float* __restrict__ qkv, int64_t ld,
    const float* __restrict__ P, int m, float c, float ratio, const int32_t* __restrict__ ptr,
    int64_t B, int64_t N, int H, const float* __restrict__ mq_in, const float* __restrict__ D_in,
    const float* __restrict__ gD_in, float* __restrict__ g_ctx, float* __restrict__ g_ksum, int S) {
  const int lane = threadIdx.x & 63;
  int64_t w = blockIdx.x * (int64_t)(blockDim.x >> 6) + (threadIdx.x >> 6);
  if (w >= B * H * MT * S) return;
  const int sl = (int)(w % S);                 // row slice (see k_favor_ctx)
  w /= S;
  const int mt = (int)(w % MT);
  const int gh = (int)(w / MT);
  const int g = gh / H, h = gh - g * H;
  const int n_beg = ptr[g], n_end = ptr[g + 1];
  const int per = ((n_end - n_beg + 15) / 16 + S - 1) / S * 16;
  const int n0 = min(n_end, n_beg + sl * per), n1 = min(n_end, n0 + per);
  if (S > 1) {
    g_ctx += (int64_t)sl * B * H * 272 * DH;
    g_ksum += (int64_t)sl * B * H * 272;
  }
  const int i = lane & 15, grp = lane >> 4;
  const int inner = H * DH;
  const int f = mt * 16 + i;
  float pv[KPL];
  load_row16(P + (int64_t)clampi(f, m - 1) * DH + 16 * grp, f < m, 1.0f, pv);
  f32x4 acc[4];
#pragma unroll
  for (int et = 0; et < 4; ++et) acc[et] = zero4();
  float gks = 0.0f;
  for (int qb = n0; qb < n1; qb += 16) {
    const int qrow = qb + i;
    float qv[KPL];
    load_row16(qkv + (int64_t)clampi(qrow, n1 - 1) * ld + h * DH + 16 * grp, qrow < n1, c, qv);
    const float nrm = group_sum(sumsq16(qv));
    const f32x4 dd = mm_rows(qv, pv, zero4());     // [query 4g+r][feature l&15]
    f32x4 phi;
    float invD[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qb + 4 * grp + r;
      const bool ok = qq < n1 && f < m;
      const float dg = 0.5f * __shfl(nrm, 4 * grp + r);
      const int64_t qi = (int64_t)h * N + clampi(qq, n1 - 1);      // valid address; masked by `ok` / `qq < n1` below
      const float mqv = mq_in[qi], Dv = D_in[qi], gDv = gD_in[qi];
      phi[r] = ok ? ratio * (expf(dd[r] - dg - mqv) + FEPS) : 0.0f;
      invD[r] = qq < n1 ? 1.0f / Dv : 0.0f;
      gks = fmaf(gDv, phi[r], gks);                                // (phi = 0 past the slice; explicit: both forms of the kernel round alike)
    }
#pragma unroll
    for (int et = 0; et < 4; ++et)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int qq = qb + 4 * grp + r;
        const float gv = g_out[(int64_t)clampi(qq, n1 - 1) * inner + h * DH + et * 16 + i] * invD[r];   // (invD = 0 past the slice)
        acc[et] = mfma16(gv, phi[r], acc[et]);     // g_ctx^T[e 4g+r'][feature l&15]
      }
  }
  gks = group_sum(gks);
  if (f < m) {
    float* cp = g_ctx + ((int64_t)gh * 272 + f) * DH;
#pragma unroll
    for (int et = 0; et < 4; ++et)
      *reinterpret_cast<float4*>(cp + et * 16 + 4 * grp) =
          make_float4(acc[et][0], acc[et][1], acc[et][2], acc[et][3]);
    if (grp == 0) g_ksum[(int64_t)gh * 272 + f] = gks;
  }
}

// the backward context kernel in the staged form (see k_favor_ctx_st): operand A = the query rows, operand B = the g_out
// rows, and three scalars per row (mq, D, gD)
template <int CS_WAVES>
__global__ __launch_bounds__(64 * CS_WAVES) void k_favor_bwd_ctx_st(
    const float* __restrict__ g_out, const float* __restrict__ qkv, int64_t ld,
    const float* __restrict__ P, int m, float c, float ratio, const int32_t* __restrict__ ptr,
    int64_t B, int64_t N, int H, const float* __restrict__ mq_in, const float* __restrict__ D_in,
    const float* __restrict__ gD_in, float* __restrict__ g_ctx, float* __restrict__ g_ksum, int S) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int u = blockIdx.x;
  const int sl = u % S, gh = u / S;
  const int g = gh / H, h = gh - g * H;
  const int n_beg = ptr[g], n_end = ptr[g + 1];
  const int per = ((n_end - n_beg + 15) / 16 + S - 1) / S * 16;
  const int n0 = min(n_end, n_beg + sl * per), n1 = min(n_end, n0 + per);
  if (S > 1) {
    g_ctx += (int64_t)sl * B * H * 272 * DH;
    g_ksum += (int64_t)sl * B * H * 272;
  }
  const int i = lane & 15, grp = lane >> 4;
  const int inner = H * DH;
  const int nt = wave + CS_WAVES < MT ? 2 : 1;
  float pv[2][KPL];
  f32x4 acc[2][4];
  float gks[2] = {0.0f, 0.0f};
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const int f = (wave + t * CS_WAVES) * 16 + i;
    load_row16(P + (int64_t)clampi(f, m - 1) * DH + 16 * grp, f < m && t < nt, 1.0f, pv[t]);
#pragma unroll
    for (int et = 0; et < 4; ++et) acc[t][et] = zero4();
  }
  const float* qsrc = qkv + h * DH;
  const float* gsrc = g_out + h * DH;
  // threads 512..559 carry the per-row scalars of the block: mq | D | gD, 16 rows each
  const int xt = (int)threadIdx.x - 512;
  const bool has_x = xt >= 0 && xt < 48;
  const float* const xsrc = (xt < 16 ? mq_in : xt < 32 ? D_in : gD_in) + (int64_t)h * N;
  float4 nx0 = make_float4(0.f, 0.f, 0.f, 0.f), nx1 = nx0, nx2 = nx0;      // three row blocks in flight (see k_favor_ctx_st)
  float xs0 = 0.0f, xs1 = 0.0f, xs2 = 0.0f;
  if (n0 < n1) {
    nx0 = cs_load(qsrc, ld, gsrc, inner, n0, n1);
    xs0 = xsrc[clampi(n0 + (xt & 15), n1 - 1)];
    cs_store(lds_proj, nx0);
    if (has_x) lds_proj[CS_A + CS_B + xt] = xs0;
  }
  if (n0 + 16 < n1) {
    nx0 = cs_load(qsrc, ld, gsrc, inner, n0 + 16, n1);
    xs0 = xsrc[clampi(n0 + 16 + (xt & 15), n1 - 1)];
  }
  if (n0 + 32 < n1) {
    nx1 = cs_load(qsrc, ld, gsrc, inner, n0 + 32, n1);
    xs1 = xsrc[clampi(n0 + 32 + (xt & 15), n1 - 1)];
  }
  __syncthreads();
  int step = 0;
  for (int qb = n0; qb < n1; qb += 16, ++step) {
    const float* buf = lds_proj + (step & 1) * CS_BUF;
    if (qb + 48 < n1) {
      nx2 = cs_load(qsrc, ld, gsrc, inner, qb + 48, n1);
      xs2 = xsrc[clampi(qb + 48 + (xt & 15), n1 - 1)];
    }
    float qv[KPL];
    cs_row16(buf, i, grp, qb + i < n1 ? c : 0.0f, qv);
    const float nrm = group_sum(sumsq16(qv));
    float dg[4], mqv[4], invD[4], gDv[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int qq = qb + 4 * grp + r;
      dg[r] = 0.5f * __shfl(nrm, 4 * grp + r);
      const float* x = buf + CS_A + CS_B + 4 * grp + r;
      mqv[r] = x[0];
      invD[r] = qq < n1 ? 1.0f / x[16] : 0.0f;
      gDv[r] = x[32];
    }
#pragma unroll
    for (int t = 0; t < 2; ++t) {
      if (t < nt) {
        const int f = (wave + t * CS_WAVES) * 16 + i;
        const f32x4 dd = mm_rows(qv, pv[t], zero4());     // [query 4g+r][feature l&15]
        f32x4 phi;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          const int qq = qb + 4 * grp + r;
          const bool ok = qq < n1 && f < m;
          phi[r] = ok ? ratio * (expf(dd[r] - dg[r] - mqv[r]) + FEPS) : 0.0f;
          gks[t] = fmaf(gDv[r], phi[r], gks[t]);                     // (phi = 0 past the slice)
        }
#pragma unroll
        for (int et = 0; et < 4; ++et)
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float gv = buf[CS_A + (4 * grp + r) * PP + et * 16 + i] * invD[r];   // (invD = 0 past the slice)
            acc[t][et] = mfma16(gv, phi[r], acc[t][et]);     // g_ctx^T[e 4g+r'][feature l&15]
          }
      }
    }
    if (qb + 16 < n1) {
      float* nb = lds_proj + ((step + 1) & 1) * CS_BUF;
      cs_store(nb, nx0);
      if (has_x) nb[CS_A + CS_B + xt] = xs0;
    }
    __syncthreads();
    nx0 = nx1; xs0 = xs1;
    nx1 = nx2; xs1 = xs2;
  }
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    if (t < nt) {
      const int f = (wave + t * CS_WAVES) * 16 + i;
      const float gk = group_sum(gks[t]);
      if (f < m) {
        float* cp = g_ctx + ((int64_t)gh * 272 + f) * DH;
#pragma unroll
        for (int et = 0; et < 4; ++et)
          *reinterpret_cast<float4*>(cp + et * 16 + 4 * grp) =
              make_float4(acc[t][et][0], acc[t][et][1], acc[t][et][2], acc[t][et][3]);
        if (grp == 0) g_ksum[(int64_t)gh * 272 + f] = gk;
      }
    }
  }
}

// ---------------------------------------------------------------------------------------------
// B3: per (key tile, head): g_k (without the key-max term), g_v, and this tile's share of g_M
// ---------------------------------------------------------------------------------------------
// One work item of the key-side kernel: gcb / gkb = the g_ctx / g_ksum record of the item's (graph, head); returns the
// item's contribution to -g_M (valid on every lane).
template <int PPITCH, int CPITCH, int KSTRIDE>
__device__ __forceinline__ float favor_bwd_k_item(const Seg& s, const float* __restrict__ qkv, int64_t ld, const float* Pp, int m,
                                                  float c, float ratio, const int32_t* __restrict__ nmax_dev, int H,
                                                  const unsigned long long* __restrict__ kmax, const float* gcb,
                                                  const float* gkb, float* __restrict__ d_qkv, int64_t ldg) {
  const int inner = H * DH;
  const int gh = s.g * H + s.h;
  const int krow = s.row0 + s.i;
  const bool k_ok = krow < s.n1;
  const int pad = max(nmax_dev[0] - (s.n1 - s.n0), 0);
  const float M = key_max_M(kmax, gh, pad);
  float kv[KPL], vv[KPL];
  const int krc = clampi(krow, s.n1 - 1);
  load_row16(qkv + (int64_t)krc * ld + inner + s.h * DH + 16 * s.grp, k_ok, c, kv);
  load_row16(qkv + (int64_t)krc * ld + 2 * inner + s.h * DH + 16 * s.grp, k_ok, 1.0f, vv);
  const float half_nrm = 0.5f * group_sum(sumsq16(kv));
  float sk = 0.0f;
  f32x4 accK[4], accV[4];
#pragma unroll
  for (int t = 0; t < 4; ++t) { accK[t] = zero4(); accV[t] = zero4(); }
  for (int mt = 0; mt < MT; ++mt) {
    const int prow = mt * 16 + s.i;
    float pv[KPL], gc[KPL];
    load_row16(Pp + clampi(prow, m - 1) * PPITCH + 16 * s.grp, prow < m, 1.0f, pv);
    load_row16(gcb + clampi(prow, m - 1) * CPITCH + 16 * s.grp, prow < m, 1.0f, gc);
    const f32x4 dd = mm_rows(pv, kv, zero4());     // [feature 4g+r][key l&15]
    const f32x4 gphi = mm_rows(gc, vv, zero4());   // sum_e g_ctx[f][e] v[key][e]
    f32x4 phi, gB;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int f = mt * 16 + 4 * s.grp + r;
      const bool ok = f < m && k_ok;
      const float ex = ok ? ratio * expf(dd[r] - half_nrm - M) : 0.0f;
      phi[r] = ok ? ex + ratio * FEPS : 0.0f;
      gB[r] = (gphi[r] + gkb[clampi(f, m - 1) * KSTRIDE]) * ex;        // (ex = 0 when !ok)
      sk += gB[r];
    }
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
      for (int r = 0; r < 4; ++r) {
        const int f = mt * 16 + 4 * s.grp + r;
        const int fc = clampi(f, m - 1);               // (gB = phi = 0 past m)
        const float pe = Pp[fc * PPITCH + t * 16 + s.i];
        const float ge = gcb[fc * CPITCH + t * 16 + s.i];
        accK[t] = mfma16(pe, gB[r], accK[t]);      // g_k^T[dh][key] += P^T[dh][f] g_B[f][key]
        accV[t] = mfma16(ge, phi[r], accV[t]);     // g_v^T[e][key]  += g_ctx^T[e][f] phi_k^T[f][key]
      }
  }
  sk = group_sum(sk);                                // sum_f g_B for key (l&15)
  if (k_ok) {
    const float* ksrc = qkv + (int64_t)krow * ld + inner + s.h * DH;
    float* ok_ = d_qkv + (int64_t)krow * ldg + inner + s.h * DH;
    float* ov_ = d_qkv + (int64_t)krow * ldg + 2 * inner + s.h * DH;
    const float k2 = -sk * c * c;
#pragma unroll
    for (int t = 0; t < 4; ++t) {
      const int col = t * 16 + 4 * s.grp;
      const float4 kk = *reinterpret_cast<const float4*>(ksrc + col);
      *reinterpret_cast<float4*>(ok_ + col) =
          make_float4(fmaf(k2, kk.x, c * accK[t][0]), fmaf(k2, kk.y, c * accK[t][1]),
                      fmaf(k2, kk.z, c * accK[t][2]), fmaf(k2, kk.w, c * accK[t][3]));
      *reinterpret_cast<float4*>(ov_ + col) =
          make_float4(accV[t][0], accV[t][1], accV[t][2], accV[t][3]);
    }
  }
  // this tile's contribution to -g_M: sum over its valid keys of sk (each key counted once: group 0)
  return wave_sum((s.grp == 0 && k_ok) ? sk : 0.0f);
}

template <bool LP>
__global__ __launch_bounds__(LP ? FK_THREADS : 256) void k_favor_bwd_k(
    const float* __restrict__ qkv, int64_t ld, const float* __restrict__ P, int m, float c,
    float ratio, const int32_t* __restrict__ ptr, const int32_t* __restrict__ tile_graph,
    const int32_t* __restrict__ tile_row0, int64_t n_work, const int32_t* __restrict__ nmax_dev,
    int H, const unsigned long long* __restrict__ kmax, const float* __restrict__ g_ctx,
    const float* __restrict__ g_ksum, float* __restrict__ d_qkv, int64_t ldg,
    float* __restrict__ gM_part) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  const float* Pp = P;
  if constexpr (LP) {
    stage_projection(lds_proj, P, m);
    Pp = lds_proj;
  }
  constexpr int pp = LP ? PP : DH;
  const int wpb = blockDim.x >> 6;
  for (int64_t w = blockIdx.x * (int64_t)wpb + (threadIdx.x >> 6); w < n_work; w = LP ? w + (int64_t)gridDim.x * wpb : n_work) {
    if constexpr (LP) asm volatile("" ::: "memory");        // (see k_favor_bwd_q)
    const Seg s = tile_work(ptr, tile_graph, tile_row0, n_work, H, w);
    if (!s.valid) {
      if ((threadIdx.x & 63) == 0) gM_part[w] = 0.0f;
      continue;
    }
    const int gh = s.g * H + s.h;
    const float part = favor_bwd_k_item<pp, DH, 1>(s, qkv, ld, Pp, m, c, ratio, nmax_dev, H, kmax, g_ctx + (int64_t)gh * 272 * DH,
                                                   g_ksum + (int64_t)gh * 272, d_qkv, ldg);
    if ((threadIdx.x & 63) == 0) gM_part[w] = part;
  }
}

// (gM_part: the slot of tile t of graph g is (ptr[g] >> 4) + g + t -- the tile map of ops.build_graph_index, which
// k_favor_bwd_kmax_fix walks; slots of tiles that do not exist are never read)
template <bool PRE, int NW>
__global__ __launch_bounds__(64 * NW) void k_favor_bwd_k_lc(
    const float* __restrict__ qkv, int64_t ld, const float* __restrict__ P, int m, float c, float ratio,
    const int32_t* __restrict__ ptr, int B, const int32_t* __restrict__ nmax_dev, int H,
    const unsigned long long* __restrict__ kmax, const float* __restrict__ g_ctx, const float* __restrict__ g_ksum,
    float* __restrict__ d_qkv, int64_t ldg, float* __restrict__ gM_part) {
  extern __shared__ __attribute__((aligned(16))) float lds_proj[];
  lc_drive<PRE, NW>(lds_proj, P, m, ptr, B, H, g_ctx, g_ksum, [&](const Seg& s, const float* lp, const float* lc) {
    const float part = favor_bwd_k_item<PP, PP, PP>(s, qkv, ld, lp, m, c, ratio, nmax_dev, H, kmax, lc, lc + DH, d_qkv, ldg);
    const int64_t tile = (s.n0 >> 4) + s.g + ((s.row0 - s.n0) >> 4);
    if ((threadIdx.x & 63) == 0) gM_part[tile * H + s.h] = part;
  });
}

// ---------------------------------------------------------------------------------------------
// B4: per (graph, head): g_M = -(sum of tile parts) - pad * sum_f g_ksum[f] * r exp(-M); it lands on
// the arg-max element of dd_k when that element (not a padded row's 0 logit) is the global max.
// ---------------------------------------------------------------------------------------------
__global__ void k_favor_bwd_kmax_fix(const float* __restrict__ P, int m, float c, float ratio,
                                     const int32_t* __restrict__ ptr,
                                     const int32_t* __restrict__ nmax_dev, int64_t B, int H,
                                     const unsigned long long* __restrict__ kmax,
                                     const float* __restrict__ g_ksum,
                                     const float* __restrict__ gM_part, float* __restrict__ d_qkv,
                                     int64_t ldg) {
  const int gh = blockIdx.x;
  if (gh >= B * H) return;
  const int lane = threadIdx.x;  // 64 threads
  const int g = gh / H, h = gh - g * H;
  const int n0 = ptr[g], n1 = ptr[g + 1];
  if (n1 <= n0) return;
  const int pad = max(nmax_dev[0] - (n1 - n0), 0);
  const unsigned long long km = kmax[gh];
  const float m_real = dec_f32((uint32_t)(km >> 32));
  if (pad > 0 && !(m_real > 0.0f)) return;   // the max is a padded row's zero logit: constant
  const uint32_t idx = 0xFFFFFFFFu - (uint32_t)(km & 0xFFFFFFFFu);
  const int key = n0 + (int)(idx / 272u), f = (int)(idx % 272u);
  // tile slots of this graph: (n0>>4)+g .. , same map as gps_attn_tile_map; parts are indexed tile*H+h
  float acc = 0.0f;
  const int t0 = (n0 >> 4) + g;
  const int nt = (n1 - n0 + 15) >> 4;
  for (int t = lane; t < nt; t += 64) acc += gM_part[(int64_t)(t0 + t) * H + h];
  float gM = -wave_sum(acc);
  if (pad > 0) {
    float ks = 0.0f;
    for (int j = lane; j < m; j += 64) ks += g_ksum[(int64_t)gh * 272 + j];
    gM -= (float)pad * wave_sum(ks) * ratio * expf(-m_real);
  }
  d_qkv[(int64_t)key * ldg + H * DH + h * DH + lane] += c * gM * P[(int64_t)f * DH + lane];
}

// b_real (or null): device word with the number of REAL graphs of a padded batch (loader.BucketPadding appends dead graphs
// behind them) -- the reference's to_dense_batch Nmax is the longest graph the loader emitted, padding must not move it
__global__ void k_segment_max_len(const int32_t* __restrict__ ptr, int64_t B, const int32_t* __restrict__ b_real,
                                  int32_t* __restrict__ out) {
  int best = 0;
  if (b_real) B = min(B, (int64_t)max(b_real[0], 0));
  for (int64_t g = threadIdx.x; g < B; g += blockDim.x) best = max(best, ptr[g + 1] - ptr[g]);
  for (int o = 32; o > 0; o >>= 1) best = max(best, __shfl_xor(best, o));
  __shared__ int sm[16];
  if ((threadIdx.x & 63) == 0) sm[threadIdx.x >> 6] = best;
  __syncthreads();
  if (threadIdx.x == 0) {
    for (int w = 1; w < (int)(blockDim.x >> 6); ++w) best = max(best, sm[w]);
    out[0] = best;
  }
}

}  // namespace

extern "C" {

int gps_segment_max_len(const int32_t* ptr, int64_t B, int32_t* nmax, gps_stream_t stream) {
  GPS_REQUIRE(ptr && nmax && B >= 0, "gps_segment_max_len: bad arguments");
  k_segment_max_len<<<1, 1024, 0, gps::as_stream(stream)>>>(ptr, B, nullptr, nmax);
  return gps::launch_status("gps_segment_max_len");
}

int gps_segment_max_len_real(const int32_t* ptr, int64_t B, const int32_t* b_real, int32_t* nmax, gps_stream_t stream) {
  GPS_REQUIRE(ptr && nmax && B >= 0, "gps_segment_max_len_real: bad arguments");
  k_segment_max_len<<<1, 1024, 0, gps::as_stream(stream)>>>(ptr, B, b_real, nmax);
  return gps::launch_status("gps_segment_max_len_real");
}

}  // extern "C"
static bool favor_ctx_staged(int64_t N, int64_t B);
// Row slices per (graph, head, feature tile) of the two context kernels: enough wavefronts to put ~8 on every SIMD, at
// least 4 row blocks per slice on average, at most 8.  The STAGED context kernels launch one 12-wavefront workgroup per
// (graph, head, slice) and a CU holds one of them (registers), so there the slices are what fills ONE dispatch round:
// CUs / (graphs x heads), rounded down -- 32 AST graphs x 4 heads: 2 slices = 256 workgroups, where the wavefront rule's
// 4 made two rounds and 3 make one and a half (code2 step, same box: 9.54 | 9.73 | 9.59 ms for 2 | 3 | 4;
// profiles/r06_favor_slices.txt).
static int favor_slices(int64_t N, int64_t B, int H) {
  static const int forced = [] { const char* e = getenv("GPS_FAVOR_SLICES"); return e && *e ? atoi(e) : 0; }();
  if (forced >= 1) return forced > 8 ? 8 : forced;
  if (B <= 0 || H <= 0) return 1;
  const int64_t blocks = N / B / 16;
  int64_t S;
  if (favor_ctx_staged(N, B)) {
    static const int cus = [] {
      int dev = 0, n = 0;
      if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess ||
          n <= 0) n = 256;
      return n;
    }();
    S = cus / (B * H);
  } else {
    const int64_t waves = B * H * MT;
    S = (8192 + waves - 1) / waves;
  }
  if (S > blocks / 4) S = blocks / 4;
  return (int)(S < 1 ? 1 : (S > 8 ? 8 : S));
}
// The LDS-staged form of the per-tile kernels: one workgroup per CU.  GPS_FAVOR_LDS=0 never, =1 always (tests), default: when
// there are at least two work items per wavefront slot of the chip to spread the 74 KB copy over.
static bool favor_lds(int64_t n_work) {
  const char* e = getenv("GPS_FAVOR_LDS");
  if (e && *e) return atoi(e) != 0;
  return n_work >= 2048;
}
// The chunked form with the context record staged too (k_favor_*_lc): long graphs only -- a chunk is four 16-row tiles of
// ONE graph and costs a 70 KB copy.  GPS_FAVOR_LC=0 never, =1 whenever the shapes allow it.
static bool favor_lc(int64_t max_tiles, int64_t B) {
  if (B < 1 || B > kLcMaxGraphs) return false;
  const char* e = getenv("GPS_FAVOR_LC");
  if (e && *e) return atoi(e) != 0;
  return max_tiles >= 16 * B && max_tiles >= 512;
}
// Wavefronts per workgroup of the chunked kernels, as measured on the code2-long layer (profiles/r05_favor_lds_projection.txt:
// 4 wavefronts with the next record prefetched through registers | 8 wavefronts, two per SIMD, no prefetch): query side
// 208 | 217 us -> 4; key side 253 | 228 -> 8; output 141 | 128 -> 8.  Round 6 removed the forms that lost (among them the
// 8-wavefront query kernel, which spilled 62 registers); only the instantiations below are built.
// the context kernels with their rows staged in LDS (k_favor_*ctx_st): GPS_FAVOR_CTX_LDS=0 never, =1 always; default: LONG
// graphs only (>= 384 rows on average).  Measured (profiles/r05_favor_ctx_staged.txt): 600-1000-row graphs 129 -> 119 and
// 158 -> 132 us, but at the code2 dataset's own sizes (~125 rows per graph) the staged form LOSES -- 44 -> 55 / 53 -> 61 us
// at 32 graphs, 163 -> 204 / 199 -> 228 at 128: a workgroup per (graph, head, slice) has 17 times fewer units to spread
// over the chip than a wavefront per feature tile, and 8 row blocks do not amortise its start.
static bool favor_ctx_staged(int64_t N, int64_t B) {
  const char* e = getenv("GPS_FAVOR_CTX_LDS");
  if (e && *e) return atoi(e) != 0;
  return B > 0 && N / B >= 384;
}
// (staged context kernels: 12 wavefronts per workgroup -- the 17 feature tiles dealt 5 / 4 / 4 / 4 over the SIMDs; 9 measured
// the same, 16 needs a 128-register cap, spills and loses: profiles/r05_favor_ctx_staged.txt)
template <typename... A>
static void launch_ctx_st(unsigned grid, hipStream_t s, A... a) {
  k_favor_ctx_st<12><<<grid, 768, CS_LDS_BYTES, s>>>(a...);
}
template <typename... A>
static void launch_bwd_ctx_st(unsigned grid, hipStream_t s, A... a) {
  k_favor_bwd_ctx_st<12><<<grid, 768, CS_LDS_BYTES, s>>>(a...);
}
template <typename K>
static bool favor_lc_ready(K kernel) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             LC_LDS_BYTES) == hipSuccess;
}
static int cu_count() {
  static const int n = [] {
    int dev = 0, cus = 0;
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess)
      cus = 256;
    return cus > 0 ? cus : 256;
  }();
  return n;
}
template <typename K>
static bool favor_lds_ready(K kernel) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                             P_LDS_BYTES) == hipSuccess;
}
static int cu_count();
static unsigned favor_lc_grid(int64_t max_tiles, int64_t B, int H, int nw) {
  const int64_t chunks = (max_tiles / nw + B) * H;                      // upper bound (the exact count is on the device)
  return (unsigned)(chunks < cu_count() ? chunks : cu_count());
}
static unsigned favor_lds_grid(int64_t n_work, int threads) {
  const int64_t wpb = threads / 64, blocks = (n_work + wpb - 1) / wpb;
  return (unsigned)(blocks < cu_count() ? blocks : cu_count());
}
extern "C" {

// floats of workspace gps_favor_fwd / gps_favor_bwd take (partial context records of the row slices; 0: no slicing)
size_t gps_favor_workspace_floats(int64_t N, int64_t B, int H) {
  const int S = favor_slices(N, B, H);
  return S > 1 ? (size_t)S * B * H * 272 * (DH + 1) : 0;
}

int gps_favor_fwd(const float* qkv, int64_t ld_qkv, const float* proj, int m, const int32_t* ptr,
                  const int32_t* nmax, const int32_t* tile_graph, const int32_t* tile_row0,
                  int64_t max_tiles, int64_t N, int64_t B, int H, int dh, float* out, float* ctx,
                  float* ksum, uint64_t* kmax, float* mq, float* D, float* ws, size_t ws_floats, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && B >= 0 && H > 0 && max_tiles >= 0, "gps_favor_fwd: bad sizes");
  if (dh != DH || m <= 0 || m > 16 * MT) {
    gps::set_error("gps_favor_fwd: only dim_head=64 with nb_features<=272 is compiled (got dh=%d m=%d)", dh, m);
    return GPS_EUNSUPPORTED;
  }
  GPS_REQUIRE(ld_qkv >= 3LL * H * DH && ld_qkv % 4 == 0, "gps_favor_fwd: bad ld_qkv");
  if (N == 0 || B == 0) return GPS_OK;
  GPS_REQUIRE(qkv && proj && ptr && nmax && tile_graph && tile_row0 && out && ctx && ksum && kmax && mq && D,
              "gps_favor_fwd: null buffer");
  hipStream_t s = gps::as_stream(stream);
  const float c = powf((float)DH, -0.25f), ratio = 1.0f / sqrtf((float)m);
  gps::fill_words(kmax, 0u, (size_t)(2 * B * H), s);      // (a fill kernel, not a memset node: gps_common.hpp fill_words)
  const int64_t n_work = max_tiles * H;
  static const bool lds_ok = favor_lds_ready(&k_favor_kmax<true>);
  const bool lp = lds_ok && favor_lds(n_work);
  if (lp)
    k_favor_kmax<true><<<favor_lds_grid(n_work, FM_THREADS), FM_THREADS, P_LDS_BYTES, s>>>(qkv, ld_qkv, proj, m, c, ptr, tile_graph,
                                                                              tile_row0, n_work, H, (unsigned long long*)kmax);
  else
    k_favor_kmax<false><<<gps::grid_for(n_work, 4), 256, 0, s>>>(qkv, ld_qkv, proj, m, c, ptr, tile_graph, tile_row0,
                                                                 n_work, H, (unsigned long long*)kmax);
  int S = favor_slices(N, B, H);
  if (!ws || ws_floats < (size_t)S * B * H * 272 * (DH + 1)) S = 1;       // no workspace: one wavefront per record
  const bool staged = favor_ctx_staged(N, B);
  if (S > 1) {
    float* cpart = ws;
    float* kpart = ws + (size_t)S * B * H * 272 * DH;
    if (staged)
      launch_ctx_st((unsigned)(B * H * S), s, qkv, ld_qkv, proj, m, c, ratio, ptr, nmax, B, H,
                    (const unsigned long long*)kmax, cpart, kpart, S);
    else
      k_favor_ctx<<<gps::grid_for(B * H * MT * S, 4), 256, 0, s>>>(qkv, ld_qkv, proj, m, c, ratio, ptr, nmax, B, H,
                                                                   (const unsigned long long*)kmax, cpart, kpart, S);
    k_favor_sum_parts<<<gps::grid_for(B * H * 272 * (DH / 4 + 1), 256), 256, 0, s>>>(
        cpart, kpart, S, B * H, m, ratio, ptr, nmax, H, (const unsigned long long*)kmax, ctx, ksum);
  } else if (staged) {
    launch_ctx_st((unsigned)(B * H), s, qkv, ld_qkv, proj, m, c, ratio, ptr, nmax, B, H,
                  (const unsigned long long*)kmax, ctx, ksum, 1);
  } else {
    k_favor_ctx<<<gps::grid_for(B * H * MT, 4), 256, 0, s>>>(qkv, ld_qkv, proj, m, c, ratio, ptr, nmax, B, H,
                                                             (const unsigned long long*)kmax, ctx, ksum, 1);
  }
  // (LP form of the output kernel: not used -- with the loop over items it needs 333 registers, one wavefront per SIMD
  // instead of two, and ran at 232 us against 149, profiles/r05_favor_lds_projection.txt; the chunked form with the context
  // record staged as well is)
  static const bool lc_ok = favor_lc_ready(&k_favor_out_lc<false, 8>);
  if (lc_ok && favor_lc(max_tiles, B)) {
    k_favor_out_lc<false, 8><<<favor_lc_grid(max_tiles, B, H, 8), 512, LC_LDS_BYTES, s>>>(qkv, ld_qkv, proj, m, c, ratio, ptr, (int)B, N,
                                                                                         H, ctx, ksum, out, mq, D);
  } else
    k_favor_out<false><<<gps::grid_for(n_work, 4), 256, 0, s>>>(qkv, ld_qkv, proj, m, c, ratio, ptr, tile_graph,
                                                                tile_row0, n_work, N, H, ctx, ksum, out, mq, D);
  return gps::launch_status("gps_favor_fwd");
}

int gps_favor_bwd(const float* g_out, const float* qkv, int64_t ld_qkv, const float* proj, int m,
                  const float* out, const int32_t* ptr, const int32_t* nmax, const int32_t* tile_graph,
                  const int32_t* tile_row0, int64_t max_tiles, int64_t N, int64_t B, int H, int dh,
                  const float* ctx, const float* ksum, const uint64_t* kmax, const float* mq,
                  const float* D, float* gD, float* g_ctx, float* g_ksum, float* gM_part,
                  float* d_qkv, int64_t ld_dqkv, float* ws, size_t ws_floats, gps_stream_t stream) {
  GPS_REQUIRE(N >= 0 && B >= 0 && H > 0 && max_tiles >= 0, "gps_favor_bwd: bad sizes");
  if (dh != DH || m <= 0 || m > 16 * MT) {
    gps::set_error("gps_favor_bwd: only dim_head=64 with nb_features<=272 is compiled (got dh=%d m=%d)", dh, m);
    return GPS_EUNSUPPORTED;
  }
  GPS_REQUIRE(ld_qkv >= 3LL * H * DH && ld_dqkv >= 3LL * H * DH && ld_qkv % 4 == 0 && ld_dqkv % 4 == 0,
              "gps_favor_bwd: bad leading dims");
  if (N == 0 || B == 0) return GPS_OK;
  GPS_REQUIRE(g_out && qkv && proj && out && ptr && nmax && tile_graph && tile_row0 && ctx && ksum && kmax &&
                  mq && D && gD && g_ctx && g_ksum && gM_part && d_qkv,
              "gps_favor_bwd: null buffer");
  hipStream_t s = gps::as_stream(stream);
  const float c = powf((float)DH, -0.25f), ratio = 1.0f / sqrtf((float)m);
  const int64_t n_work = max_tiles * H;
  k_favor_bwd_gd<<<gps::grid_for(N * H * 16, 256), 256, 0, s>>>(g_out, out, D, N, H, gD);
  static const bool lds_ok = favor_lds_ready(&k_favor_bwd_q<true>) && favor_lds_ready(&k_favor_bwd_k<true>);
  const bool lp = lds_ok && favor_lds(n_work);
  static const bool lc_ok = favor_lc_ready(&k_favor_bwd_q_lc<true, 4>) && favor_lc_ready(&k_favor_bwd_k_lc<false, 8>);
  if (lc_ok && favor_lc(max_tiles, B)) {
    k_favor_bwd_q_lc<true, 4><<<favor_lc_grid(max_tiles, B, H, 4), 256, LC_LDS_BYTES, s>>>(g_out, qkv, ld_qkv, proj, m, c, ratio, ptr,
                                                                                          (int)B, N, H, ctx, ksum, mq, D, gD, d_qkv,
                                                                                          ld_dqkv);
  } else if (lp)
    k_favor_bwd_q<true><<<favor_lds_grid(n_work, FQ_THREADS), FQ_THREADS, P_LDS_BYTES, s>>>(g_out, qkv, ld_qkv, proj, m, c, ratio, ptr,
                                                                                          tile_graph, tile_row0, n_work, N, H, ctx,
                                                                                          ksum, mq, D, gD, d_qkv, ld_dqkv);
  else
    k_favor_bwd_q<false><<<gps::grid_for(n_work, 4), 256, 0, s>>>(g_out, qkv, ld_qkv, proj, m, c, ratio, ptr, tile_graph,
                                                                  tile_row0, n_work, N, H, ctx, ksum, mq, D, gD, d_qkv,
                                                                  ld_dqkv);
  int S = favor_slices(N, B, H);
  if (!ws || ws_floats < (size_t)S * B * H * 272 * (DH + 1)) S = 1;
  const bool staged = favor_ctx_staged(N, B);
  if (S > 1) {
    float* cpart = ws;
    float* kpart = ws + (size_t)S * B * H * 272 * DH;
    if (staged)
      launch_bwd_ctx_st((unsigned)(B * H * S), s, g_out, qkv, ld_qkv, proj, m, c, ratio, ptr, B, N, H, mq, D, gD,
                        cpart, kpart, S);
    else
      k_favor_bwd_ctx<<<gps::grid_for(B * H * MT * S, 4), 256, 0, s>>>(g_out, qkv, ld_qkv, proj, m, c, ratio, ptr, B, N,
                                                                       H, mq, D, gD, cpart, kpart, S);
    k_favor_sum_parts<<<gps::grid_for(B * H * 272 * (DH / 4 + 1), 256), 256, 0, s>>>(
        cpart, kpart, S, B * H, m, ratio, ptr, nmax, H, nullptr, g_ctx, g_ksum);
  } else if (staged) {
    launch_bwd_ctx_st((unsigned)(B * H), s, g_out, qkv, ld_qkv, proj, m, c, ratio, ptr, B, N, H, mq, D, gD,
                      g_ctx, g_ksum, 1);
  } else {
    k_favor_bwd_ctx<<<gps::grid_for(B * H * MT, 4), 256, 0, s>>>(g_out, qkv, ld_qkv, proj, m, c, ratio, ptr, B, N,
                                                                 H, mq, D, gD, g_ctx, g_ksum, 1);
  }
  if (lc_ok && favor_lc(max_tiles, B)) {
    k_favor_bwd_k_lc<false, 8><<<favor_lc_grid(max_tiles, B, H, 8), 512, LC_LDS_BYTES, s>>>(qkv, ld_qkv, proj, m, c, ratio, ptr, (int)B,
                                                                                           nmax, H, (const unsigned long long*)kmax,
                                                                                           g_ctx, g_ksum, d_qkv, ld_dqkv, gM_part);
  } else if (lp)
    k_favor_bwd_k<true><<<favor_lds_grid(n_work, FK_THREADS), FK_THREADS, P_LDS_BYTES, s>>>(qkv, ld_qkv, proj, m, c, ratio, ptr, tile_graph,
                                                                               tile_row0, n_work, nmax, H,
                                                                               (const unsigned long long*)kmax, g_ctx, g_ksum,
                                                                               d_qkv, ld_dqkv, gM_part);
  else
    k_favor_bwd_k<false><<<gps::grid_for(n_work, 4), 256, 0, s>>>(qkv, ld_qkv, proj, m, c, ratio, ptr, tile_graph,
                                                                  tile_row0, n_work, nmax, H,
                                                                  (const unsigned long long*)kmax, g_ctx, g_ksum, d_qkv,
                                                                  ld_dqkv, gM_part);
  k_favor_bwd_kmax_fix<<<(unsigned)(B * H), 64, 0, s>>>(proj, m, c, ratio, ptr, nmax, B, H,
                                                        (const unsigned long long*)kmax, g_ksum, gM_part,
                                                        d_qkv, ld_dqkv);
  return gps::launch_status("gps_favor_bwd");
}

}  // extern "C"
