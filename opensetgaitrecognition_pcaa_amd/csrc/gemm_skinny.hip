// Batch-skinny dense layers (M = batch <= 64 rows): the CGDecoder's Linear stack
// (reference models.py CGDecoder.forward: 32+K -> S/16 -> S/8 -> S/4 -> S/2 -> S, S = T*C*N).
//
// With 64 rows every product is a pass over the fp32 weight matrix W[N_out][K_in] (157 M
// parameters at T=30,C=4,N=128): forward and dgrad READ it once, wgrad WRITES a matrix of that
// size once.  The 256x256-tile kernel spent 75 % of its tile on padding and reached 1.4-1.8 TB/s
// on these; here the weights are streamed straight from HBM into MFMA fragments:
//
//   forward  y[m][n]  = sum_k x[m][k]  W[n][k]   contraction index contiguous in W: each wave
//            reads its 32 rows as 256-B runs (dwordx4), converts to bf16 and redistributes them
//            through a WAVE-PRIVATE LDS image (no workgroup barrier, LDS is in-order per wave).
//   dgrad    dx[m][k] = sum_n dz[m][n] W[n][k]   contraction index is the ROW of W: lane l of a
//            32-column fragment reads W[n0+8h+e][k0+l], e = 0..7 -- eight dword loads, each one
//            two whole 128-B lines per wave; no LDS for W at all.
//   wgrad    dW[n][k] = sum_m dz[m][n] x[m][k]   contraction over the <= 64 batch rows: both
//            operands load like dgrad's W (rows = contraction), the kernel is a pure fp32 store
//            stream (128 B per row per half-wave).
//
// The 64-row operand of forward/dgrad is shared by the 4 waves of a workgroup through a
// double-buffered bf16 LDS image ([64][64+8]); the workgroup barrier is a raw s_barrier after
// s_waitcnt lgkmcnt(0), NOT __syncthreads(): the latter drains vmcnt and with it the weight
// prefetch that is the whole point.  Split-K partials go to slabs (no atomics); the reduction
// kernel fuses bias + ELU (forward) or ELU'(previous activation) (dgrad).
#include <stdlib.h>

#include <algorithm>
#include <type_traits>

#include "gemm_common.h"

namespace {

constexpr int SP = 72;            // LDS pitch in bf16 elements of a 64-deep chunk row (144 B)
constexpr int CH = 64;            // contraction elements per chunk (4 MFMA steps)

__device__ __forceinline__ uint32_t pk2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 b;
  b.x = (bf16_t)lo;
  b.y = (bf16_t)hi;
  return __builtin_bit_cast(uint32_t, b);
}
__device__ __forceinline__ bf16x8 pack8(const float (&t)[8]) {
  uint4 u;
  u.x = pk2(t[0], t[1]); u.y = pk2(t[2], t[3]); u.z = pk2(t[4], t[5]); u.w = pk2(t[6], t[7]);
  return __builtin_bit_cast(bf16x8, u);
}
__device__ __forceinline__ void lds_barrier() {
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

// ---- shared 64-row operand: thread t stages row t>>2, floats [16*(t&3), +16) of the chunk
struct SmallStage { f32x4 v[4]; };
__device__ __forceinline__ void small_load(SmallStage& st, const float* __restrict__ S, long ld, int M, int k0,
                                           int tid) {
  const int m = tid >> 2, seg = (tid & 3) * 16;
  if (m < M) {
    const float* p = S + (long)m * ld + k0 + seg;
#pragma unroll
    for (int q = 0; q < 4; ++q) st.v[q] = load4(p + 4 * q);
  } else {
#pragma unroll
    for (int q = 0; q < 4; ++q) st.v[q] = f32x4{0.f, 0.f, 0.f, 0.f};
  }
}
__device__ __forceinline__ void small_store(const SmallStage& st, bf16_t* buf, int tid) {
  const int m = tid >> 2, seg = (tid & 3) * 16;
  uint4 a, b;
  a.x = pk2(st.v[0].x, st.v[0].y); a.y = pk2(st.v[0].z, st.v[0].w);
  a.z = pk2(st.v[1].x, st.v[1].y); a.w = pk2(st.v[1].z, st.v[1].w);
  b.x = pk2(st.v[2].x, st.v[2].y); b.y = pk2(st.v[2].z, st.v[2].w);
  b.z = pk2(st.v[3].x, st.v[3].y); b.w = pk2(st.v[3].z, st.v[3].w);
  *reinterpret_cast<uint4*>(&buf[m * SP + seg]) = a;
  *reinterpret_cast<uint4*>(&buf[m * SP + seg + 8]) = b;
}
__device__ __forceinline__ bf16x8 small_frag(const bf16_t* buf, int mf, int s, int l31, int h) {
  return *reinterpret_cast<const bf16x8*>(&buf[(mf * 32 + l31) * SP + s * 16 + h * 8]);
}

// ---- exact-fp32 variants (the parity modes: "fp32" and "fp16x3"): the same streams and the same lane maps, the operands
// stay fp32 and one v_mfma_f32_32x32x16_bf16 becomes eight v_mfma_f32_32x32x2_f32 (lane (c, h) holds k = 8h .. 8h+7 of
// its row / column; MFMA e contracts k = e and 8 + e): fp32 products, 1/16 of the bf16 rate -- 20 GFLOP per pass over the
// 157 M decoder weights, about what the fp32 weight stream itself takes.  Rounds 1-2 served these modes with the
// 128x128-tile fp32 GEMM: 64 of its 128 rows padding, and the weight gradient (contraction over the 64 batch rows) 2-3 ms.
constexpr int SPF = 68;           // LDS pitch in floats of a 64-deep fp32 chunk row (272 B: 16 rows x b128 = all 64 banks)
__device__ __forceinline__ void small_store_f32(const SmallStage& st, float* buf, int tid) {
  const int m = tid >> 2, seg = (tid & 3) * 16;
#pragma unroll
  for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(&buf[m * SPF + seg + 4 * q]) = st.v[q];
}
__device__ __forceinline__ void frag_f32(const float* buf, int row, int s, int h, float (&f)[8]) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(&buf[row * SPF + s * 16 + h * 8]);
  const f32x4 b = *reinterpret_cast<const f32x4*>(&buf[row * SPF + s * 16 + h * 8 + 4]);
  f[0] = a.x; f[1] = a.y; f[2] = a.z; f[3] = a.w; f[4] = b.x; f[5] = b.y; f[6] = b.z; f[7] = b.w;
}
__device__ __forceinline__ f32x16 mfma8_f32(const float (&a)[8], const float (&b)[8], f32x16 acc) {
#pragma unroll
  for (int e = 0; e < 8; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[e], b[e], acc, 0, 0, 0);
  return acc;
}

// slab store of a wave's 64(m) x 32(col) accumulators
__device__ __forceinline__ void store_acc(const f32x16 (&acc)[2], float* __restrict__ out, long ld, int M, int col,
                                          int ncols, int h) {
  if (col >= ncols) return;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < M) out[(long)m * ld + col] = acc[mf][r];
    }
}

// ------------------------------------------------------------------ dgrad: dx = dz . W
// grid (ceil(K/128), nsplit); wave w owns columns [128 bx + 32 w, +32); split by owns cps chunks of n
__global__ __launch_bounds__(256) void skinny_dgrad_kernel(const float* __restrict__ dz, long lddz,
                                                           const float* __restrict__ W, long ldw,
                                                           float* __restrict__ slabs, long slab_stride, int M,
                                                           int N, int K, int cps) {
  __shared__ __attribute__((aligned(16))) bf16_t sbuf[2][64 * SP];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int col = blockIdx.x * 128 + wave * 32 + l31;
  const int c0 = blockIdx.y * cps;
  const int nch = min(cps, N / CH - c0);
  if (nch <= 0) return;                                  // uniform; the host never launches such a split
  const int lane_off = 8 * h * (int)ldw + min(col, K - 1);   // < 2^31: checked on the host
  const float* Wc = W + (long)c0 * CH * ldw;             // uniform base of this split

  float wr[4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) wr[s][e] = (Wc + (long)(s * 16 + e) * ldw)[lane_off];
  SmallStage st;
  small_load(st, dz, lddz, M, c0 * CH, tid);
  small_store(st, sbuf[0], tid);
  f32x16 acc[2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;
  lds_barrier();

  for (int c = 0; c < nch; ++c) {
    const bool more = c + 1 < nch;
    const bf16_t* cur = sbuf[c & 1];
    if (more) small_load(st, dz, lddz, M, (c0 + c + 1) * CH, tid);
    // the last iteration re-reads its own chunk (L2 hit) instead of branching around the prefetch:
    // a conditional here made the compiler rotate the whole register ring with v_mov every step
    const float* Wn = Wc + (long)min(c + 1, nch - 1) * CH * ldw;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      const bf16x8 b = pack8(wr[s]);
#pragma unroll
      for (int e = 0; e < 8; ++e) wr[s][e] = (Wn + (long)(s * 16 + e) * ldw)[lane_off];
      const bf16x8 a0 = small_frag(cur, 0, s, l31, h), a1 = small_frag(cur, 1, s, l31, h);
      acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0], 0, 0, 0);
      acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1], 0, 0, 0);
    }
    if (more) small_store(st, sbuf[(c + 1) & 1], tid);
    lds_barrier();
  }
  store_acc(acc, slabs + (long)blockIdx.y * slab_stride, K, M, col, K, h);
}

// dgrad, two columns per lane: the same structure with 8-B loads -- lane l of a wave reads W[n][c0+2l, +1], the
// even columns feed one MFMA B-fragment and the odd ones a second (a column permutation the store undoes), so
// a wave covers 64 columns with HALF the vector-memory instructions per byte.  The one-column kernel was bound
// by those (7200 dword wave-loads per CU on the 7680x15360 layer: 3.2 TB/s against 4.5-5 of the other two).
// grid (ceil(K/256), nsplit)
// W16 (round 4): W is the bf16 image the fused update keeps beside the fp32 weights -- the same values the conversion
// below produces (round to nearest even), half the bytes of the one stream this kernel is made of
template <bool F32 = false, bool W16 = false>
__global__ __launch_bounds__(256) void skinny_dgrad2_kernel(const float* __restrict__ dz, long lddz,
                                                            const typename std::conditional<W16, bf16_t, float>::type* __restrict__ W, long ldw,
                                                            float* __restrict__ slabs, long slab_stride, int M,
                                                            int N, int K, int cps) {
  __shared__ __attribute__((aligned(16))) unsigned char sraw[2 * 64 * (F32 ? SPF * 4 : SP * 2)];
  bf16_t (*sbuf)[64 * SP] = reinterpret_cast<bf16_t (*)[64 * SP]>(sraw);
  float (*sbuf32)[64 * SPF] = reinterpret_cast<float (*)[64 * SPF]>(sraw);
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int col = blockIdx.x * 256 + wave * 64 + 2 * l31;          // this lane's column pair
  const int c0 = blockIdx.y * cps;
  const int nch = min(cps, N / CH - c0);
  if (nch <= 0) return;
  const int lane_off = 8 * h * (int)ldw + min(col, K - 2);
  const auto* Wc = W + (long)c0 * CH * ldw;
  typedef typename std::conditional<W16, uint32_t, f32x2>::type wpair_t;      // this lane's two columns of one row

  wpair_t wr[4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) wr[s][e] = *reinterpret_cast<const wpair_t*>(Wc + (long)(s * 16 + e) * ldw + lane_off);
  SmallStage st;
  small_load(st, dz, lddz, M, c0 * CH, tid);
  if constexpr (F32) small_store_f32(st, sbuf32[0], tid); else small_store(st, sbuf[0], tid);
  f32x16 acc[2][2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int f = 0; f < 2; ++f)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[mf][f][r] = 0.f;
  lds_barrier();

  for (int c = 0; c < nch; ++c) {
    const bool more = c + 1 < nch;
    const bf16_t* cur = sbuf[c & 1];
    const float* cur32 = sbuf32[c & 1];
    if (more) small_load(st, dz, lddz, M, (c0 + c + 1) * CH, tid);
    const auto* Wn = Wc + (long)min(c + 1, nch - 1) * CH * ldw;     // last trip: harmless re-read
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      float ev[8], od[8];
      bf16x8 b0w, b1w;
      if constexpr (W16) {
        // even columns = low halves, odd columns = high halves of the eight dwords: two byte permutes per pair
        uint4 lo, hi;
        uint32_t* lp = reinterpret_cast<uint32_t*>(&lo);
        uint32_t* hp = reinterpret_cast<uint32_t*>(&hi);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          lp[e] = __builtin_amdgcn_perm(wr[s][2 * e + 1], wr[s][2 * e], 0x05040100u);
          hp[e] = __builtin_amdgcn_perm(wr[s][2 * e + 1], wr[s][2 * e], 0x07060302u);
        }
        b0w = __builtin_bit_cast(bf16x8, lo);
        b1w = __builtin_bit_cast(bf16x8, hi);
      } else {
#pragma unroll
        for (int e = 0; e < 8; ++e) { ev[e] = wr[s][e].x; od[e] = wr[s][e].y; }
      }
#pragma unroll
      for (int e = 0; e < 8; ++e) wr[s][e] = *reinterpret_cast<const wpair_t*>(Wn + (long)(s * 16 + e) * ldw + lane_off);
      if constexpr (F32) {
        float a0[8], a1[8];
        frag_f32(cur32, l31, s, h, a0);
        frag_f32(cur32, 32 + l31, s, h, a1);
        acc[0][0] = mfma8_f32(a0, ev, acc[0][0]);
        acc[0][1] = mfma8_f32(a0, od, acc[0][1]);
        acc[1][0] = mfma8_f32(a1, ev, acc[1][0]);
        acc[1][1] = mfma8_f32(a1, od, acc[1][1]);
      } else {
        bf16x8 b0, b1;
        if constexpr (W16) { b0 = b0w; b1 = b1w; } else { b0 = pack8(ev); b1 = pack8(od); }
        const bf16x8 a0 = small_frag(cur, 0, s, l31, h), a1 = small_frag(cur, 1, s, l31, h);
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
      }
    }
    if (more) { if constexpr (F32) small_store_f32(st, sbuf32[(c + 1) & 1], tid); else small_store(st, sbuf[(c + 1) & 1], tid); }
    lds_barrier();
  }
  if (col >= K) return;
  float* out = slabs + (long)blockIdx.y * slab_stride;
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const int m = mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
      if (m < M) *reinterpret_cast<f32x2*>(out + (long)m * K + col) = f32x2{acc[mf][0][r], acc[mf][1][r]};
    }
}

// ------------------------------------------------------------------ forward: y = x . W^T
// grid (ceil(N/128), nsplit); wave w owns output columns (rows of W) [128 bx + 32 w, +32)
// What bounds it (round 6, the 472 MB layer alone; profiles/r06_skinny_probes.txt): NOT the depth of the weight stream -- a
// variant with two chunks of W per wave in flight (two register stages, x requested ahead of them so that its wait leaves
// the younger W loads in flight across the barrier, 156 VGPRs, three workgroups per CU) timed 105.5 us against 105.6 -- but
// the 64-row operand: at M = 64 / 32 / 8 / 1 rows the same weight stream takes 105.8 / 97.3 / 88.8 / 84.8 us (dgrad: 99.5 /
// 89.9 / 85.5 / 83.4).  Every workgroup re-reads its x chunk (16 KB per 32 KB of weights: 236 MB from the L2s per pass),
// writes a 32 KB slab, and the reduction reads them back.  Halving the FIRST of the three -- the operands as bf16 images
// written by the producing reduction / the Chamfer kernel, read instead of the fp32 operands, bit-identical results -- was
// built and changed nothing in the step (profiles/r06_ab_dec_operand_images.txt; removed): what is left is the split-K's
// own slab traffic and reduction launch.
template <bool F32 = false, bool W16 = false>
__global__ __launch_bounds__(256) void skinny_fwd_kernel(const float* __restrict__ x, long ldx,
                                                         const typename std::conditional<W16, bf16_t, float>::type* __restrict__ W, long ldw,
                                                         float* __restrict__ slabs, long slab_stride, int M,
                                                         int N, int K, int cps, const float* __restrict__ bias, int act) {
  __shared__ __attribute__((aligned(16))) unsigned char sraw[2 * 64 * (F32 ? SPF * 4 : SP * 2)];
  __shared__ __attribute__((aligned(16))) unsigned char wraw[4 * 32 * (F32 ? SPF * 4 : SP * 2)];
  bf16_t (*sbuf)[64 * SP] = reinterpret_cast<bf16_t (*)[64 * SP]>(sraw);
  float (*sbuf32)[64 * SPF] = reinterpret_cast<float (*)[64 * SPF]>(sraw);
  bf16_t (*wbuf)[32 * SP] = reinterpret_cast<bf16_t (*)[32 * SP]>(wraw);
  float (*wbuf32)[32 * SPF] = reinterpret_cast<float (*)[32 * SPF]>(wraw);
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.x * 128 + wave * 32;
  const int c0 = blockIdx.y * cps;
  const int nch = min(cps, K / CH - c0);
  if (nch <= 0) return;
  // load map: instruction i reads rows 4i .. 4i+3 of the wave's 32, 16 lanes x 16 B = one 256-B run each
  const int lr = lane >> 4, kseg = (lane & 15) * 4;
  int roff[8];
#pragma unroll
  for (int i = 0; i < 8; ++i) roff[i] = min(n0 + 4 * i + lr, N - 1) * (int)ldw + kseg;   // < 2^31: host-checked
  bf16_t* wl = wbuf[wave];
  float* wl32 = wbuf32[wave];

  typedef typename std::conditional<W16, uint2, f32x4>::type wquad_t;        // four consecutive k of one row
  wquad_t wr[8];
  const auto* Wk = W + (long)c0 * CH;
#pragma unroll
  for (int i = 0; i < 8; ++i) wr[i] = *reinterpret_cast<const wquad_t*>(Wk + roff[i]);
  SmallStage st;
  small_load(st, x, ldx, M, c0 * CH, tid);
  if constexpr (F32) small_store_f32(st, sbuf32[0], tid); else small_store(st, sbuf[0], tid);
  f32x16 acc[2];
#pragma unroll
  for (int mf = 0; mf < 2; ++mf)
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[mf][r] = 0.f;
  lds_barrier();

  for (int c = 0; c < nch; ++c) {
    const bool more = c + 1 < nch;
    const bf16_t* cur = sbuf[c & 1];
    const float* cur32 = sbuf32[c & 1];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      if constexpr (W16) {
        *reinterpret_cast<uint2*>(&wl[(4 * i + lr) * SP + kseg]) = wr[i];
      } else if constexpr (F32) {
        *reinterpret_cast<f32x4*>(&wl32[(4 * i + lr) * SPF + kseg]) = wr[i];
      } else {
        uint2 u;
        u.x = pk2(wr[i].x, wr[i].y);
        u.y = pk2(wr[i].z, wr[i].w);
        *reinterpret_cast<uint2*>(&wl[(4 * i + lr) * SP + kseg]) = u;
      }
    }
    {
      const auto* Wn = Wk + (long)min(c + 1, nch - 1) * CH;      // last iteration: harmless re-read
#pragma unroll
      for (int i = 0; i < 8; ++i) wr[i] = *reinterpret_cast<const wquad_t*>(Wn + roff[i]);
    }
    if (more) small_load(st, x, ldx, M, (c0 + c + 1) * CH, tid);
    // the wave reads back its own LDS image: LDS executes a wave's operations in order, the
    // compiler barrier keeps the reads below the writes
    asm volatile("" ::: "memory");
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (F32) {
        float bq[8], a0[8], a1[8];
        frag_f32(wl32, l31, s, h, bq);
        frag_f32(cur32, l31, s, h, a0);
        frag_f32(cur32, 32 + l31, s, h, a1);
        acc[0] = mfma8_f32(a0, bq, acc[0]);
        acc[1] = mfma8_f32(a1, bq, acc[1]);
      } else {
        const bf16x8 b = *reinterpret_cast<const bf16x8*>(&wl[l31 * SP + s * 16 + h * 8]);
        const bf16x8 a0 = small_frag(cur, 0, s, l31, h), a1 = small_frag(cur, 1, s, l31, h);
        acc[0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b, acc[0], 0, 0, 0);
        acc[1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b, acc[1], 0, 0, 0);
      }
    }
    if (more) { if constexpr (F32) small_store_f32(st, sbuf32[(c + 1) & 1], tid); else small_store(st, sbuf[(c + 1) & 1], tid); }
    lds_barrier();
  }
  if (gridDim.y == 1) {
    // no split (a contraction of one or two chunks, or more column groups than the chip holds workgroups): `slabs` is the
    // output itself and the reduction kernel's epilogue -- bias, ELU, in its order of operations -- runs here: one launch
    // less on the decoder's dependent chain (the 64 -> S/16 first layer)
    const int col = n0 + l31;
    if (col >= N) return;
    const float b = bias ? bias[col] : 0.f;
#pragma unroll
    for (int mf = 0; mf < 2; ++mf)
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int m = mf * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
        float v = acc[mf][r] + b;
        if (act == PCAA_ACT_ELU) v = elu_f(v);
        if (m < M) slabs[(long)m * N + col] = v;
      }
    return;
  }
  store_acc(acc, slabs + (long)blockIdx.y * slab_stride, N, M, n0 + l31, N, h);
}

// ------------------------------------------------------------------ wgrad: dW = dz^T . x
// grid (ceil(K / (4*32*JL)), ceil(N/128)); the workgroup owns dW rows [128 by, +128), wave w the
// columns [(4 bx + w) * 32 JL, + 32 JL).  The dz panel of those 128 rows is packed ONCE per
// workgroup into MFMA-fragment order in LDS (16 KB) and re-read per 32-column step -- holding it
// in registers (64 VGPRs + 128 loads in flight per wave) cost the kernel its occupancy.
// 32-bit element offsets from uniform base pointers (host-checked < 2^31) so loads and stores use
// the SGPR-base + VGPR-offset form.  Rows m >= M: dz is zeroed after a clamped load, x is only
// clamped (0 * finite = 0).
// TO = bf16 (the data-parallel step with bf16 gradient buckets: the gradient is produced in the form it crosses the wire
// in, no fp32 copy and no cast pass): neighbouring lanes exchange one packed pair (DPP) so that every lane stores two
// adjacent columns of one row as one dword -- 64-B runs per row and half-wave, the other half of the line follows with
// the wave's next 32 columns.
template <int JL, bool FULLN, typename TO = float, bool F32 = false>
__global__ __launch_bounds__(256) void skinny_wgrad_kernel(const float* __restrict__ dz, long lddz,
                                                           const float* __restrict__ x, long ldx,
                                                           TO* __restrict__ dW, long lddw, int M, int N,
                                                           int K) {
  __shared__ __attribute__((aligned(16))) bf16x8 apan[F32 ? 1 : 4][4][64];      // [row fragment i][k-step s][lane]
  __shared__ __attribute__((aligned(16))) f32x4 apan32[F32 ? 4 : 1][4][2][64];   // exact variant: the same, fp32 (2 x 16 B)
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const int n0 = blockIdx.y * 128;
  const int kb = (blockIdx.x * 4 + wave) * (32 * JL);
  const int jn = kb < K ? min(JL, (K - kb) / 32) : 0;

  {
    const int s = wave;                                   // wave w packs k-step w of all 4 fragments
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ncol = (unsigned)min(n0 + 32 * i + l31, N - 1);
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = 16 * s + 8 * h + e;
        const float v = dz[(unsigned)min(m, M - 1) * (unsigned)lddz + ncol];
        t[e] = m < M ? v : 0.f;
      }
      if constexpr (F32) {
        apan32[i][s][0][lane] = f32x4{t[0], t[1], t[2], t[3]};
        apan32[i][s][1][lane] = f32x4{t[4], t[5], t[6], t[7]};
      } else {
        apan[i][s][lane] = pack8(t);
      }
    }
  }
  unsigned xoff[4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e)
      xoff[s][e] = (unsigned)min(16 * s + 8 * h + e, M - 1) * (unsigned)ldx + min(kb, K - 32) + l31;
  float br[4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) br[s][e] = x[xoff[s][e]];
  __syncthreads();

  const unsigned row0 = (unsigned)(n0 + 4 * h);
  for (int j = 0; j < jn; ++j) {
    bf16x8 bf[4];
    float bq[F32 ? 4 : 1][8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (F32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bq[s][e] = br[s][e];
      } else {
        bf[s] = pack8(br[s]);
      }
    }
    {
      const float* xn = x + 32 * min(j + 1, jn - 1);      // last iteration: harmless re-read
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) br[s][e] = xn[xoff[s][e]];
    }
    asm volatile("" ::: "memory");
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (F32) {
          const f32x4 lo = apan32[i][s][0][lane], hi = apan32[i][s][1][lane];
          const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          acc[i] = mfma8_f32(a, bq[s], acc[i]);
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apan[i][s][lane], bf[s], acc[i], 0, 0, 0);
        }
      }
    TO* dj = dW + kb + 32 * j;                             // uniform
    unsigned o0 = row0 * (unsigned)lddw + l31;
    // opaque to the optimiser: LICM otherwise hoists all 64 store offsets (and the 16 LDS
    // fragment reads, hence the memory clobber) out of the j loop -- 360 VGPRs, one wave per SIMD
    asm volatile("" : "+v"(o0) : : "memory");
    if constexpr (sizeof(TO) == 4) {
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2);
          if (FULLN || row0 + rr < (unsigned)N) dj[o0 + rr * (unsigned)lddw] = acc[i][r];
        }
    } else {
      const bool odd = lane & 1;
      const uint32_t sel = odd ? 0x03020706u : 0x05040100u;      // bytes 0-3 = mine, 4-7 = the neighbouring column's
      uint32_t* dj32 = reinterpret_cast<uint32_t*>(dj);
      const unsigned w0 = (o0 & ~1u) >> 1;                        // dword index of my column pair in row row0
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          // rows rr, rr + 1 of my column as two bf16; even lane stores row rr: (mine.lo, theirs.lo), odd lane row rr + 1
          const uint32_t mine = pk2(acc[i][r], acc[i][r + 1]);
          const uint32_t theirs = (uint32_t)__builtin_amdgcn_update_dpp(0, (int)mine, 0xB1, 0xF, 0xF, true);
          const uint32_t packed = __builtin_amdgcn_perm(theirs, mine, sel);
          const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2) + (odd ? 1u : 0u);
          if (FULLN || row0 + rr < (unsigned)N) dj32[w0 + rr * (unsigned)(lddw >> 1)] = packed;
        }
    }
  }
}

// ------------------------------------------------------------------ wgrad + Adam: W -= step(dz^T . x)
// Same contraction as skinny_wgrad_kernel, but the 128 x 32 tile of dW a wave has just formed never leaves its
// registers: the wave reads W, exp_avg, exp_avg_sq at the addresses it would have stored dW to, applies Adam (the
// arithmetic of elementwise.hip's adam_kernel, expression for expression: fused and unfused updates are bit-identical)
// (common.h adam_update) and writes the three back -- 24 B per parameter instead of 4 (dW store) + 28 (Adam pass).  One 32 x 32 fragment's
// operands (48 dwords per lane) are in flight while the previous fragment is updated.  Single-process training only:
// a data-parallel step needs the reduced gradient before the update.
template <int JL, bool FULLN, int NB, bool F32 = false>
__global__ __launch_bounds__(256) void skinny_wgrad_adam_kernel(const float* __restrict__ dz, long lddz,
                                                                const float* __restrict__ x, long ldx,
                                                                float* __restrict__ W, float* __restrict__ mo,
                                                                float* __restrict__ vo, long ldw, int M, int N, int K,
                                                                float b1, float b2, float eps, float grad_scale,
                                                                const float* __restrict__ coef) {
  static_assert(NB == 2 || NB == 4, "fragment buffers: a ring of 2 or 4 (index = fragment number mod NB, static)");
  __shared__ __attribute__((aligned(16))) bf16x8 apan[F32 ? 1 : 4][4][64];      // [row fragment i][k-step s][lane]
  __shared__ __attribute__((aligned(16))) f32x4 apan32[F32 ? 4 : 1][4][2][64];   // exact variant: the same, fp32
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const float step_size = coef[0], inv_bc2_sqrt = coef[1];
  const int n0 = blockIdx.y * 128;
  const int kb = (blockIdx.x * 4 + wave) * (32 * JL);
  const int jn = kb < K ? min(JL, (K - kb) / 32) : 0;
  {
    const int s = wave;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const unsigned ncol = (unsigned)min(n0 + 32 * i + l31, N - 1);
      float t[8];
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const int m = 16 * s + 8 * h + e;
        const float v = dz[(unsigned)min(m, M - 1) * (unsigned)lddz + ncol];
        t[e] = m < M ? v : 0.f;
      }
      if constexpr (F32) {
        apan32[i][s][0][lane] = f32x4{t[0], t[1], t[2], t[3]};
        apan32[i][s][1][lane] = f32x4{t[4], t[5], t[6], t[7]};
      } else {
        apan[i][s][lane] = pack8(t);
      }
    }
  }
  unsigned xoff[4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e)
      xoff[s][e] = (unsigned)min(16 * s + 8 * h + e, M - 1) * (unsigned)ldx + min(kb, K - 32) + l31;
  float br[4][8];
#pragma unroll
  for (int s = 0; s < 4; ++s)
#pragma unroll
    for (int e = 0; e < 8; ++e) br[s][e] = x[xoff[s][e]];

  // the wave's fragments f = 4 j + i (32 rows x 32 columns each), NB - 1 of them in flight ahead of the update
  const unsigned row0 = (unsigned)(n0 + 4 * h);
  unsigned o0 = row0 * (unsigned)ldw + (unsigned)kb + l31;
  float pw[NB][16], pm[NB][16], pv[NB][16];
  const int nfrag = 4 * jn;
  auto fetch = [&](int f, int b) {                         // b = f % NB, static at every call site
    if (f >= nfrag) return;
    const unsigned base = o0 + 32u * (unsigned)(f >> 2);
    const int i = f & 3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2);
      if (FULLN || row0 + rr < (unsigned)N) {
        const unsigned o = base + rr * (unsigned)ldw;
        pw[b][r] = W[o]; pm[b][r] = mo[o]; pv[b][r] = vo[o];
      }
    }
  };
#pragma unroll
  for (int f = 0; f < NB - 1; ++f) fetch(f, f);
  __syncthreads();

  for (int j = 0; j < jn; ++j) {
    bf16x8 bf[4];
    float bq[F32 ? 4 : 1][8];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
      if constexpr (F32) {
#pragma unroll
        for (int e = 0; e < 8; ++e) bq[s][e] = br[s][e];
      } else {
        bf[s] = pack8(br[s]);
      }
    }
    {
      const float* xn = x + 32 * min(j + 1, jn - 1);
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int e = 0; e < 8; ++e) br[s][e] = xn[xoff[s][e]];
    }
    asm volatile("" : "+v"(o0) : : "memory");              // keeps the 64 row offsets out of registers (see wgrad)
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        if constexpr (F32) {
          const f32x4 lo = apan32[i][s][0][lane], hi = apan32[i][s][1][lane];
          const float a[8] = {lo.x, lo.y, lo.z, lo.w, hi.x, hi.y, hi.z, hi.w};
          acc[i] = mfma8_f32(a, bq[s], acc[i]);
        } else {
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apan[i][s][lane], bf[s], acc[i], 0, 0, 0);
        }
      }
    const unsigned base = o0 + 32u * (unsigned)j;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = i % NB;
      fetch(4 * j + i + NB - 1, (i + NB - 1) % NB);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2);
        if (FULLN || row0 + rr < (unsigned)N) {
          const unsigned o = base + rr * (unsigned)ldw;
          adam_update(pw[b][r], pm[b][r], pv[b][r], acc[i][r] * grad_scale, b1, b2, eps, step_size, inv_bc2_sqrt);
          W[o] = pw[b][r];
          mo[o] = pm[b][r];
          vo[o] = pv[b][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ round 5: the same update from GATHERED rows
// Data parallel, the weight gradient of a batch-skinny layer is dW = sum over ranks of dz_r^T . x_r = DZ^T . X with the
// ranks' rows stacked: DZ [world * B, N], X [world * B, K].  The gradient is 4 N K bytes per rank on the wire (2 N K as
// bf16); DZ and X together are 4 world B (N + K) -- for the decoder's 7680 -> 15360 layer at 8 x 64 rows 47 MB against
// 472 MB.  So the data-parallel step all-gathers the layer's two small operands instead of all-reducing its gradient,
// and every rank forms the GLOBAL gradient itself, in registers, inside this fused update: the exchange shrinks ~10x, the
// gradient still never reaches HBM, and the separate Adam pass over the decoder (28 B per parameter) that an all-reduce
// forces does not exist.  (The reference has no multi-GPU path; single process this is pcaa_skinny_linear_wgrad_adam.)
// MC chunks of 64 rows (M <= 64 MC): the dz fragments of ALL chunks stay in the LDS (16 KB per chunk), the contraction
// of a 128 x 32 fragment set runs over them with x requested one chunk ahead; everything else -- fragment ring, Adam
// arithmetic, store pattern -- is the single-process kernel's.  bf16 products (the throughput mode).
template <int JL, bool FULLN, int NB, int MC>
__global__ __launch_bounds__(256) void skinny_wgrad_adam_rows_kernel(const float* __restrict__ dz, long lddz,
                                                                     const float* __restrict__ x, long ldx,
                                                                     float* __restrict__ W, float* __restrict__ mo,
                                                                     float* __restrict__ vo, long ldw, int M, int N, int K,
                                                                     float b1, float b2, float eps, float grad_scale,
                                                                     const float* __restrict__ coef) {
  static_assert(NB == 2 || NB == 4, "fragment buffers: a ring of 2 or 4");
  extern __shared__ __attribute__((aligned(16))) bf16x8 apan_rows[];           // [MC][row fragment i][k-step s][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const float step_size = coef[0], inv_bc2_sqrt = coef[1];
  const int n0 = blockIdx.y * 128;
  const int kb = (blockIdx.x * 4 + wave) * (32 * JL);
  const int jn = kb < K ? min(JL, (K - kb) / 32) : 0;
  {
    const int s = wave;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned ncol = (unsigned)min(n0 + 32 * i + l31, N - 1);
        float t[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) {
          const int m = 64 * mc + 16 * s + 8 * h + e;
          const float v = dz[(unsigned)min(m, M - 1) * (unsigned)lddz + ncol];
          t[e] = m < M ? v : 0.f;                          // rows past M contribute nothing, whatever x holds there
        }
        apan_rows[((mc * 4 + i) * 4 + s) * 64 + lane] = pack8(t);
      }
  }
  // x[row][col]: row = 64 mc + 16 s + 8 h + e, col = kb + 32 j + l31 -- one vector offset per lane, everything else in
  // the (uniform) scalar offset of a buffer load.  The scalar offset is outside the buffer's range check: the caller
  // allocates x for 64 MC rows (the rows past M hold finite values -- zeros -- and meet zero dz fragments)
  const __amdgpu_buffer_rsrc_t rX = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)(unsigned)(64L * MC * ldx * 4), 0x00020000);
  const unsigned xv = (unsigned)(8 * h) * (unsigned)ldx * 4u + (unsigned)(min(kb, K - 32) + l31) * 4u;
  auto load_x = [&](float (&br)[4][8], int mc, int j) __attribute__((always_inline)) {
#pragma unroll
    for (int s = 0; s < 4; ++s)
#pragma unroll
      for (int e = 0; e < 8; ++e) {
        const unsigned so = (unsigned)(64 * mc + 16 * s + e) * (unsigned)ldx * 4u + 128u * (unsigned)j;
        br[s][e] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rX, xv, so, 0));
      }
  };
  float br[4][8];
  load_x(br, 0, 0);

  const unsigned row0 = (unsigned)(n0 + 4 * h);
  unsigned o0 = row0 * (unsigned)ldw + (unsigned)kb + l31;
  float pw[NB][16], pm[NB][16], pv[NB][16];
  const int nfrag = 4 * jn;
  auto fetch = [&](int f, int b) {
    if (f >= nfrag) return;
    const unsigned base = o0 + 32u * (unsigned)(f >> 2);
    const int i = f & 3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2);
      if (FULLN || row0 + rr < (unsigned)N) {
        const unsigned o = base + rr * (unsigned)ldw;
        pw[b][r] = W[o]; pm[b][r] = mo[o]; pv[b][r] = vo[o];
      }
    }
  };
#pragma unroll
  for (int f = 0; f < NB - 1; ++f) fetch(f, f);
  __syncthreads();

  for (int j = 0; j < jn; ++j) {
    asm volatile("" : "+v"(o0) : : "memory");
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
#pragma unroll
    for (int mc = 0; mc < MC; ++mc) {
      bf16x8 bf[4];
#pragma unroll
      for (int s = 0; s < 4; ++s) bf[s] = pack8(br[s]);
      // the next chunk's x (behind the last chunk: the first chunk of the next column step; last step: a harmless re-read)
      if (mc + 1 < MC) load_x(br, mc + 1, j);
      else load_x(br, 0, min(j + 1, jn - 1));
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(apan_rows[((mc * 4 + i) * 4 + s) * 64 + lane], bf[s], acc[i], 0, 0, 0);
    }
    const unsigned base = o0 + 32u * (unsigned)j;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = i % NB;
      fetch(4 * j + i + NB - 1, (i + NB - 1) % NB);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2);
        if (FULLN || row0 + rr < (unsigned)N) {
          const unsigned o = base + rr * (unsigned)ldw;
          adam_update(pw[b][r], pm[b][r], pv[b][r], acc[i][r] * grad_scale, b1, b2, eps, step_size, inv_bc2_sqrt);
          W[o] = pw[b][r];
          mo[o] = pm[b][r];
          vo[o] = pv[b][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ round 6: the gathered update from PACKED operands
// What round 5's rows kernel cost at the world sizes it exists for (tools/skinny_lab.py, the four wide layers of config[1],
// alone on the GPU): M = 64: 0.77 ms, 256: 0.87, 512: 1.14 -- the x operand is fetched with 32 dword loads per lane and
// 64-row chunk (8 batch rows x 4 k-steps, two 128-B lines per load) from a [512, K] fp32 matrix that no L2 holds
// (15.7 MB at K = 7680), re-read by every one of the 120 row panels: 1.8 GB through the L2s per update at M = 512,
// against 2.8 GB of HBM traffic for the weights themselves.
// Here every rank PACKS its two operands once, right where the backward produced them, into the form the MFMA wants:
// one 64-row chunk per rank, transposed and rounded to bf16 -- P_r[c][m], c = 0 .. N-1 the columns of dz (rows of W),
// c = N .. N+K-1 the columns of x, m = the rank's batch row (zero behind its B rows) -- 128 B per column, so that a
// lane's fragment of 8 contraction elements is ONE 16-B load instead of 8 dword loads, the bytes are half, and the
// ranks' chunks concatenate: ONE all-gather per layer (was two) moves 2 (N + K) 64 bytes per rank (was 4 (N + K) B).
// The contraction index of an MFMA is arbitrary as long as both operands agree: k-slot (s, h, e) of a chunk is batch
// row 32 h + 8 s + e, i.e. lane-half h reads the 64 contiguous bytes [64 h, 64 h + 64) of its column.
// The rounding is the rows kernel's own (fp32 -> bf16, nearest even, at the same point of the data flow), products and
// fp32 accumulation are the same MFMAs: the update differs from the rows kernel's only in summation order.
__global__ __launch_bounds__(256) void pack_rows_t16_kernel(const float* __restrict__ a, long lda, int wa,
                                                            const float* __restrict__ b, long ldb, int wb, int rows,
                                                            bf16_t* __restrict__ dst) {
  // thread = (8 batch rows q, column c), c fastest: 8 dword reads, coalesced across the threads of a row, one 16-B piece of
  // the column's 128-B row out.  (One thread per whole column -- 64 reads each, 90 workgroups at the widest layer -- took
  // 19-28 us per layer on the step's main stream: rocprofv3 trace of the first emulated 8-rank step.)
  const long t = (long)blockIdx.x * 256 + threadIdx.x;
  const int ncol = wa + wb;
  const int q = (int)(t / ncol), c = (int)(t - (long)q * ncol);
  if (q >= 8) return;
  const float* src = c < wa ? a + c : b + (c - wa);
  const long ld = c < wa ? lda : ldb;
  float v[8];
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    const int m = 8 * q + e;
    v[e] = m < rows ? src[(long)m * ld] : 0.f;
  }
  reinterpret_cast<uint4*>(dst + (long)c * 64)[q] = __builtin_bit_cast(uint4, pack8(v));
}

template <int JL, bool FULLN, int NB>
__global__ __launch_bounds__(256) void skinny_wgrad_adam_t16_kernel(const bf16_t* __restrict__ P, long chunk_stride, int MC,
                                                                    float* __restrict__ W, float* __restrict__ mo,
                                                                    float* __restrict__ vo, long ldw, int N, int K,
                                                                    float b1, float b2, float eps, float grad_scale,
                                                                    const float* __restrict__ coef) {
  static_assert(NB == 2 || NB == 4, "fragment buffers: a ring of 2 or 4");
  extern __shared__ __attribute__((aligned(16))) bf16x8 apan_t16[];            // [MC][row fragment i][k-step s][lane]
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, l31 = lane & 31, h = lane >> 5;
  const float step_size = coef[0], inv_bc2_sqrt = coef[1];
  const int n0 = blockIdx.y * 128;
  const int kb = (blockIdx.x * 4 + wave) * (32 * JL);
  const int jn = kb < K ? min(JL, (K - kb) / 32) : 0;
  // byte addressing through one buffer resource over all chunks (< 2^31 bytes: host-checked); an offset outside it
  // reads zeros instead of faulting
  const __amdgpu_buffer_rsrc_t rP = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(P), 0,
                                                                      (int)(unsigned)((long)MC * chunk_stride * 2), 0x00020000);
  typedef int i32x4 __attribute__((ext_vector_type(4)));
  {
    const int s = wave;                                    // wave w packs k-step w of all 4 row fragments of every chunk
    for (int mc = 0; mc < MC; ++mc)
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const unsigned ncol = (unsigned)min(n0 + 32 * i + l31, N - 1);
        const unsigned vo_ = ncol * 128u + (unsigned)h * 64u + (unsigned)s * 16u;
        const i32x4 v = __builtin_amdgcn_raw_buffer_load_b128(rP, vo_, (unsigned)((long)mc * chunk_stride * 2), 0);
        apan_t16[((mc * 4 + i) * 4 + s) * 64 + lane] = __builtin_bit_cast(bf16x8, v);
      }
  }
  // x fragments: column N + kb + 32 j + l31 of chunk mc, bytes [64 h + 16 s, + 16)
  const unsigned xv = (unsigned)(N + min(kb, K - 32) + l31) * 128u + (unsigned)h * 64u;
  auto load_x = [&](bf16x8 (&bf)[4], int mc, int j) __attribute__((always_inline)) {
    const unsigned so = (unsigned)((long)mc * chunk_stride * 2) + 4096u * (unsigned)j;       // 32 columns x 128 B per j step
#pragma unroll
    for (int s = 0; s < 4; ++s)
      bf[s] = __builtin_bit_cast(bf16x8, __builtin_amdgcn_raw_buffer_load_b128(rP, xv + 16u * (unsigned)s, so, 0));
  };
  bf16x8 bcur[4], bnxt[4];
  load_x(bcur, 0, 0);

  const unsigned row0 = (unsigned)(n0 + 4 * h);
  unsigned o0 = row0 * (unsigned)ldw + (unsigned)kb + l31;
  float pw[NB][16], pm[NB][16], pv[NB][16];
  const int nfrag = 4 * jn;
  auto fetch = [&](int f, int b) {
    if (f >= nfrag) return;
    const unsigned base = o0 + 32u * (unsigned)(f >> 2);
    const int i = f & 3;
#pragma unroll
    for (int r = 0; r < 16; ++r) {
      const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2);
      if (FULLN || row0 + rr < (unsigned)N) {
        const unsigned o = base + rr * (unsigned)ldw;
        pw[b][r] = W[o]; pm[b][r] = mo[o]; pv[b][r] = vo[o];
      }
    }
  };
#pragma unroll
  for (int f = 0; f < NB - 1; ++f) fetch(f, f);
  __syncthreads();

  for (int j = 0; j < jn; ++j) {
    asm volatile("" : "+v"(o0) : : "memory");
    f32x16 acc[4];
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    for (int mc = 0; mc < MC; ++mc) {
      // the next chunk's x (behind the last chunk: the first chunk of the next column step; last step: a harmless re-read)
      if (mc + 1 < MC) load_x(bnxt, mc + 1, j);
      else load_x(bnxt, 0, min(j + 1, jn - 1));
      const bf16x8* ap = apan_t16 + (mc * 16) * 64 + lane;
#pragma unroll
      for (int s = 0; s < 4; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i)
          acc[i] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ap[(i * 4 + s) * 64], bcur[s], acc[i], 0, 0, 0);
#pragma unroll
      for (int s = 0; s < 4; ++s) bcur[s] = bnxt[s];
    }
    const unsigned base = o0 + 32u * (unsigned)j;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const int b = i % NB;
      fetch(4 * j + i + NB - 1, (i + NB - 1) % NB);
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const unsigned rr = 32 * i + (r & 3) + 8 * (r >> 2);
        if (FULLN || row0 + rr < (unsigned)N) {
          const unsigned o = base + rr * (unsigned)ldw;
          adam_update(pw[b][r], pm[b][r], pv[b][r], acc[i][r] * grad_scale, b1, b2, eps, step_size, inv_bc2_sqrt);
          W[o] = pw[b][r];
          mo[o] = pm[b][r];
          vo[o] = pv[b][r];
        }
      }
    }
  }
}

// ------------------------------------------------------------------ slab reduction (+ bias/ELU, or * ELU'(a_prev))
__global__ __launch_bounds__(256) void skinny_reduce_kernel(const float* __restrict__ slabs, int ns, long stride,
                                                            float* __restrict__ out, const float* __restrict__ bias,
                                                            int act, const float* __restrict__ a_prev,
                                                            int accumulate, long nquads, int qpr) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long)gridDim.x * 256) {
    f32x4 v = load4(slabs + q * 4);
    for (int s = 1; s < ns; ++s) v += load4(slabs + s * stride + q * 4);
    if (bias) v += load4(bias + (int)((unsigned)q % (unsigned)qpr) * 4);
    if (act == PCAA_ACT_ELU) { v.x = elu_f(v.x); v.y = elu_f(v.y); v.z = elu_f(v.z); v.w = elu_f(v.w); }
    if (a_prev) {
      const f32x4 a = load4(a_prev + q * 4);
      v.x *= elu_grad_from_out(a.x); v.y *= elu_grad_from_out(a.y);
      v.z *= elu_grad_from_out(a.z); v.w *= elu_grad_from_out(a.w);
    }
    if (accumulate) v += load4(out + q * 4);
    store4(out + q * 4, v);
  }
}

// Workgroups a forward / dgrad launch may hold RESIDENT at once: 256 CUs x the workgroups per CU the kernel's registers
// and LDS admit (code-object metadata of this build: skinny_fwd_kernel<bf16> 140 VGPRs + 36 KB -> 3; <exact> 69.6 KB of LDS
// -> 2; skinny_dgrad2_kernel<bf16> 186 VGPRs -> 2; <exact> 166 -> 3; the bf16-image forms 124 / 150 -> 4 / 3, priced as
// their fp32-source siblings so that one split count serves both).  The split-K depth is the LARGEST that keeps the whole
// grid inside that -- one round of workgroups, every CU streaming to the end.  Through round 5 the depth was rounded UP
// from 768 for every kernel: the 7680 -> 15360 forward ran 840 workgroups on 768 slots and the dgrad 720 on 512, i.e. a
// second, nearly empty round each (4.3 / 4.5 TB/s; profiles/r06_skinny_splits_ab.txt).
int resident_blocks(int kind) {
  switch (kind) {
    case 0: return 768;      // forward, bf16 products
    case 1: return 512;      // dgrad, bf16 products
    case 2: return 512;      // forward, fp32 products
    case 3: return 768;      // dgrad, fp32 products
    default: return 512;
  }
}

bool aligned16(const void* p) { return ((uintptr_t)p % 16) == 0; }

int reduce_launch(const float* ws, int ns, long stride, float* out, const float* bias, int act, const float* a_prev,
                  int accumulate, int M, int ncols, hipStream_t s) {
  const long nq = (long)M * ncols / 4;
  const int grid = (int)std::min<long>(cdiv(nq, 256), 4096);
  hipLaunchKernelGGL(skinny_reduce_kernel, dim3(grid), dim3(256), 0, s, ws, ns, stride, out, bias, act, a_prev,
                     accumulate, nq, ncols / 4);
  return 0;
}

}  // namespace

// kind 0: forward (groups over N, contraction K); kind 1: dgrad (groups over K, contraction N)
extern "C" int pcaa_skinny_splits(int kind, int M, int N, int K) {
  (void)M;
  const bool fwd = (kind & 1) == 0;
  // column groups of one workgroup: 128 output columns forward; 256 input columns in the dgrad (two per lane)
  const int groups = (int)cdiv(fwd ? N : K, fwd ? 128 : 256);
  const int chunks = (fwd ? K : N) / CH;
  if (chunks < 1) return 1;
  static const bool legacy = getenv("PCAA_SKINNY_SPLITS_LEGACY") != nullptr;      // lab: the round 2-5 rule, for A/B runs
  int ns;
  if (legacy) ns = std::max(1, std::min(chunks, (768 + (int)cdiv(fwd ? N : K, 128) - 1) / (int)cdiv(fwd ? N : K, 128)));
  else ns = std::max(1, std::min(std::max(1, chunks / 2), resident_blocks(kind) / groups));   // (>= 2 chunks per workgroup: a
  // one-chunk workgroup is all prologue -- the 1920 -> 3840 dgrad went 21 -> 32 us with 480 of them)
  const int cps = (int)cdiv(chunks, ns);
  return (int)cdiv(chunks, cps);
}

extern "C" int pcaa_skinny_supported(int M, int N, int K) {
  return M >= 1 && M <= 64 && N >= 128 && K >= 64 && N % 64 == 0 && K % 64 == 0 && (long)N * K < (1L << 31) - (1L << 20);
}

static int skinny_fwd_impl(const float* x, long ldx, const void* Wv, int w16, long ldw, const float* bias, int act,
                           float* y, float* ws, long ws_floats, int M, int N, int K, int nsplit, int exact,
                           void* stream) {
  const float* W = static_cast<const float*>(Wv);
  PCAA_CHECK_ARG(x && W && y && ws, "pcaa_skinny_linear_fwd: null pointer");
  PCAA_CHECK_ARG(pcaa_skinny_supported(M, N, K), "pcaa_skinny_linear_fwd: unsupported shape M=%d N=%d K=%d", M, N, K);
  PCAA_CHECK_ARG(ldx >= K && ldx % 4 == 0 && ldw >= K && ldw % 4 == 0 && ldw * (long)N < (1L << 31),
                 "pcaa_skinny_linear_fwd: bad leading dimensions");
  PCAA_CHECK_ARG(aligned16(x) && aligned16(W) && aligned16(y) && aligned16(ws) && (!bias || aligned16(bias)),
                 "pcaa_skinny_linear_fwd: buffers must be 16-B aligned");
  PCAA_CHECK_ARG(act == PCAA_ACT_NONE || act == PCAA_ACT_ELU, "pcaa_skinny_linear_fwd: bad act");
  const int chunks = K / CH;
  PCAA_CHECK_ARG(nsplit >= 1 && nsplit <= chunks, "pcaa_skinny_linear_fwd: bad nsplit");
  const int cps = (int)cdiv(chunks, nsplit);
  PCAA_CHECK_ARG(cdiv(chunks, cps) == nsplit, "pcaa_skinny_linear_fwd: nsplit leaves empty splits (use pcaa_skinny_splits)");
  const long stride = (long)M * N;
  PCAA_CHECK_ARG(ws_floats >= stride * nsplit, "pcaa_skinny_linear_fwd: workspace too small");
  hipStream_t s = as_stream(stream);
  // one split: the kernel's own epilogue writes y (bias + activation); else slabs + the reduction launch
  float* dst = nsplit == 1 ? y : ws;
  const dim3 grid((unsigned)cdiv(N, 128), nsplit);
  if (w16)
    hipLaunchKernelGGL((skinny_fwd_kernel<false, true>), grid, dim3(256), 0, s, x, ldx, static_cast<const bf16_t*>(Wv), ldw,
                       dst, stride, M, N, K, cps, bias, act);
  else if (exact)
    hipLaunchKernelGGL(skinny_fwd_kernel<true>, grid, dim3(256), 0, s, x, ldx, W, ldw, dst, stride, M, N, K, cps, bias, act);
  else
    hipLaunchKernelGGL(skinny_fwd_kernel<false>, grid, dim3(256), 0, s, x, ldx, W, ldw, dst, stride, M, N, K, cps, bias, act);
  if (nsplit > 1) reduce_launch(ws, nsplit, stride, y, bias, act, nullptr, 0, M, N, s);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_skinny_linear_fwd");
}

extern "C" int pcaa_skinny_linear_fwd(const float* x, long ldx, const float* W, long ldw, const float* bias, int act,
                                      float* y, float* ws, long ws_floats, int M, int N, int K, int nsplit,
                                      void* stream) {
  return skinny_fwd_impl(x, ldx, W, 0, ldw, bias, act, y, ws, ws_floats, M, N, K, nsplit, 0, stream);
}
extern "C" int pcaa_skinny_linear_fwd_w16(const float* x, long ldx, const void* W16, long ldw, const float* bias, int act,
                                          float* y, float* ws, long ws_floats, int M, int N, int K, int nsplit,
                                          void* stream) {
  return skinny_fwd_impl(x, ldx, W16, 1, ldw, bias, act, y, ws, ws_floats, M, N, K, nsplit, 0, stream);
}
extern "C" int pcaa_skinny_linear_fwd_exact(const float* x, long ldx, const float* W, long ldw, const float* bias, int act,
                                            float* y, float* ws, long ws_floats, int M, int N, int K, int nsplit,
                                            void* stream) {
  return skinny_fwd_impl(x, ldx, W, 0, ldw, bias, act, y, ws, ws_floats, M, N, K, nsplit, 1, stream);
}

static int skinny_dgrad_impl(const float* dz, long lddz, const void* Wv, int w16, long ldw, float* dx,
                             const float* a_prev, int accumulate, float* ws, long ws_floats, int M, int N,
                             int K, int nsplit, int exact, void* stream) {
  const float* W = static_cast<const float*>(Wv);
  PCAA_CHECK_ARG(dz && W && dx && ws, "pcaa_skinny_linear_dgrad: null pointer");
  PCAA_CHECK_ARG(pcaa_skinny_supported(M, N, K), "pcaa_skinny_linear_dgrad: unsupported shape M=%d N=%d K=%d", M, N, K);
  PCAA_CHECK_ARG(lddz >= N && lddz % 4 == 0 && ldw >= K && ldw * (long)N < (1L << 31),
                 "pcaa_skinny_linear_dgrad: bad leading dimensions");
  PCAA_CHECK_ARG(aligned16(dz) && aligned16(dx) && aligned16(ws) && (!a_prev || aligned16(a_prev)),
                 "pcaa_skinny_linear_dgrad: buffers must be 16-B aligned");
  const int chunks = N / CH;
  PCAA_CHECK_ARG(nsplit >= 1 && nsplit <= chunks, "pcaa_skinny_linear_dgrad: bad nsplit");
  const int cps = (int)cdiv(chunks, nsplit);
  PCAA_CHECK_ARG(cdiv(chunks, cps) == nsplit, "pcaa_skinny_linear_dgrad: nsplit leaves empty splits (use pcaa_skinny_splits)");
  const long stride = (long)M * K;
  PCAA_CHECK_ARG(ws_floats >= stride * nsplit, "pcaa_skinny_linear_dgrad: workspace too small");
  hipStream_t s = as_stream(stream);
  // two columns per lane where the 8-B loads are aligned, else the one-column-per-lane kernel
  const bool pairs = (ldw % 2) == 0 && ((uintptr_t)W % 8) == 0;
  PCAA_CHECK_ARG(!exact || pairs, "pcaa_skinny_linear_dgrad_exact: W must be 8-B aligned with an even leading dimension");
  PCAA_CHECK_ARG(!w16 || ((ldw % 2) == 0 && ((uintptr_t)Wv % 4) == 0 && K % 2 == 0),
                 "pcaa_skinny_linear_dgrad_w16: the bf16 image must be 4-B aligned with an even leading dimension");
  if (w16)
    hipLaunchKernelGGL((skinny_dgrad2_kernel<false, true>), dim3((unsigned)cdiv(K, 256), nsplit), dim3(256), 0, s, dz, lddz,
                       static_cast<const bf16_t*>(Wv), ldw, ws, stride, M, N, K, cps);
  else if (exact)
    hipLaunchKernelGGL(skinny_dgrad2_kernel<true>, dim3((unsigned)cdiv(K, 256), nsplit), dim3(256), 0, s, dz, lddz, W, ldw,
                       ws, stride, M, N, K, cps);
  else if (pairs)
    hipLaunchKernelGGL(skinny_dgrad2_kernel<false>, dim3((unsigned)cdiv(K, 256), nsplit), dim3(256), 0, s, dz, lddz, W, ldw,
                       ws, stride, M, N, K, cps);
  else
    hipLaunchKernelGGL(skinny_dgrad_kernel, dim3((unsigned)cdiv(K, 128), nsplit), dim3(256), 0, s, dz, lddz, W, ldw,
                       ws, stride, M, N, K, cps);
  reduce_launch(ws, nsplit, stride, dx, nullptr, PCAA_ACT_NONE, a_prev, accumulate, M, K, s);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_skinny_linear_dgrad");
}

extern "C" int pcaa_skinny_linear_dgrad(const float* dz, long lddz, const float* W, long ldw, float* dx,
                                        const float* a_prev, int accumulate, float* ws, long ws_floats, int M, int N,
                                        int K, int nsplit, void* stream) {
  return skinny_dgrad_impl(dz, lddz, W, 0, ldw, dx, a_prev, accumulate, ws, ws_floats, M, N, K, nsplit, 0, stream);
}
extern "C" int pcaa_skinny_linear_dgrad_w16(const float* dz, long lddz, const void* W16, long ldw, float* dx,
                                            const float* a_prev, int accumulate, float* ws, long ws_floats, int M, int N,
                                            int K, int nsplit, void* stream) {
  return skinny_dgrad_impl(dz, lddz, W16, 1, ldw, dx, a_prev, accumulate, ws, ws_floats, M, N, K, nsplit, 0, stream);
}
extern "C" int pcaa_skinny_linear_dgrad_exact(const float* dz, long lddz, const float* W, long ldw, float* dx,
                                              const float* a_prev, int accumulate, float* ws, long ws_floats, int M,
                                              int N, int K, int nsplit, void* stream) {
  return skinny_dgrad_impl(dz, lddz, W, 0, ldw, dx, a_prev, accumulate, ws, ws_floats, M, N, K, nsplit, 1, stream);
}

static int skinny_wgrad_impl(const float* dz, long lddz, const float* x, long ldx, float* dW, long lddw,
                             int M, int N, int K, int exact, void* stream) {
  PCAA_CHECK_ARG(dz && x && dW, "pcaa_skinny_linear_wgrad: null pointer");
  PCAA_CHECK_ARG(M >= 1 && M <= 64 && N >= 1 && K >= 32 && K % 32 == 0,
                 "pcaa_skinny_linear_wgrad: unsupported shape M=%d N=%d K=%d", M, N, K);
  PCAA_CHECK_ARG(lddz >= N && ldx >= K && lddw >= K, "pcaa_skinny_linear_wgrad: bad leading dimensions");
  PCAA_CHECK_ARG((long)M * lddz < (1L << 31) && (long)M * ldx < (1L << 31) && (long)N * lddw < (1L << 31),
                 "pcaa_skinny_linear_wgrad: operands beyond 32-bit element offsets");
  constexpr int JL = 4;
  const dim3 grid((unsigned)cdiv(K, 4 * 32 * JL), (unsigned)cdiv(N, 128));
  if (exact) {
    if (N % 128 == 0)
      hipLaunchKernelGGL((skinny_wgrad_kernel<JL, true, float, true>), grid, dim3(256), 0, as_stream(stream), dz, lddz, x,
                         ldx, dW, lddw, M, N, K);
    else
      hipLaunchKernelGGL((skinny_wgrad_kernel<JL, false, float, true>), grid, dim3(256), 0, as_stream(stream), dz, lddz, x,
                         ldx, dW, lddw, M, N, K);
  } else if (N % 128 == 0) {
    hipLaunchKernelGGL((skinny_wgrad_kernel<JL, true>), grid, dim3(256), 0, as_stream(stream), dz, lddz, x, ldx, dW,
                       lddw, M, N, K);
  } else {
    hipLaunchKernelGGL((skinny_wgrad_kernel<JL, false>), grid, dim3(256), 0, as_stream(stream), dz, lddz, x, ldx, dW,
                       lddw, M, N, K);
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_skinny_linear_wgrad");
}

extern "C" int pcaa_skinny_linear_wgrad(const float* dz, long lddz, const float* x, long ldx, float* dW, long lddw,
                                        int M, int N, int K, void* stream) {
  return skinny_wgrad_impl(dz, lddz, x, ldx, dW, lddw, M, N, K, 0, stream);
}
extern "C" int pcaa_skinny_linear_wgrad_exact(const float* dz, long lddz, const float* x, long ldx, float* dW, long lddw,
                                              int M, int N, int K, void* stream) {
  return skinny_wgrad_impl(dz, lddz, x, ldx, dW, lddw, M, N, K, 1, stream);
}

extern "C" int pcaa_skinny_linear_wgrad_bf16(const float* dz, long lddz, const float* x, long ldx, void* dW_bf16, long lddw,
                                             int M, int N, int K, void* stream) {
  PCAA_CHECK_ARG(dz && x && dW_bf16, "pcaa_skinny_linear_wgrad_bf16: null pointer");
  PCAA_CHECK_ARG(M >= 1 && M <= 64 && N >= 1 && K >= 32 && K % 32 == 0,
                 "pcaa_skinny_linear_wgrad_bf16: unsupported shape M=%d N=%d K=%d", M, N, K);
  PCAA_CHECK_ARG(lddz >= N && ldx >= K && lddw >= K && lddw % 2 == 0 && ((uintptr_t)dW_bf16 % 4) == 0,
                 "pcaa_skinny_linear_wgrad_bf16: bad leading dimensions / alignment");
  PCAA_CHECK_ARG((long)M * lddz < (1L << 31) && (long)M * ldx < (1L << 31) && (long)N * lddw < (1L << 31),
                 "pcaa_skinny_linear_wgrad_bf16: operands beyond 32-bit element offsets");
  constexpr int JL = 4;
  const dim3 grid((unsigned)cdiv(K, 4 * 32 * JL), (unsigned)cdiv(N, 128));
  bf16_t* out = reinterpret_cast<bf16_t*>(dW_bf16);
  if (N % 128 == 0)
    hipLaunchKernelGGL((skinny_wgrad_kernel<JL, true, bf16_t>), grid, dim3(256), 0, as_stream(stream), dz, lddz, x, ldx, out,
                       lddw, M, N, K);
  else
    hipLaunchKernelGGL((skinny_wgrad_kernel<JL, false, bf16_t>), grid, dim3(256), 0, as_stream(stream), dz, lddz, x, ldx, out,
                       lddw, M, N, K);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_skinny_linear_wgrad_bf16");
}

static int skinny_wgrad_adam_impl(const float* dz, long lddz, const float* x, long ldx, float* W,
                                  float* exp_avg, float* exp_avg_sq, long ldw, int M, int N, int K,
                                  float beta1, float beta2, float eps, float grad_scale,
                                  const float* coef_dev, int exact, void* stream) {
  PCAA_CHECK_ARG(dz && x && W && exp_avg && exp_avg_sq && coef_dev, "pcaa_skinny_linear_wgrad_adam: null pointer");
  PCAA_CHECK_ARG(M >= 1 && M <= 64 && N >= 1 && K >= 32 && K % 32 == 0,
                 "pcaa_skinny_linear_wgrad_adam: unsupported shape M=%d N=%d K=%d", M, N, K);
  PCAA_CHECK_ARG(lddz >= N && ldx >= K && ldw >= K, "pcaa_skinny_linear_wgrad_adam: bad leading dimensions");
  PCAA_CHECK_ARG((long)M * lddz < (1L << 31) && (long)M * ldx < (1L << 31) && (long)N * ldw < (1L << 30),
                 "pcaa_skinny_linear_wgrad_adam: operands beyond 32-bit offsets");
  // (round 6, built and removed: the MFMA operands swapped -- the transposed tile, a lane holding four consecutive columns of
  // ONE row, so that every access to W / exp_avg / exp_avg_sq is 16 B wide, 4 + 4 instructions per fragment and array instead
  // of 16 + 16 -- bit-identical results, 0.99-1.03 ms against 0.79 on the same box: a wave instruction then touches a
  // 32-B piece of 32 different lines instead of two whole lines; profiles/r06_wgrad_adam_swap_lab.txt)
  // measured alone on the four wide layers of the bench shape (tools/skinny_lab.py): 0.81 ms fused against 0.97 ms
  // (weight gradient 0.17 + Adam 0.80), 5.0 TB/s on the 7680 -> 15360 layer; a ring of 4 fragment buffers (three
  // fragments in flight) or 8 column steps per wave: 0.83 / 0.88 / 0.86 ms -- not kept
  // columns per wave = 32 JL: the widest that still gives the chip >= 1024 workgroups (every fragment is a full
  // memory round trip for its wave: the 960 -> 1920 layer took 47 us in 30 workgroups of 4 x 4 fragments per wave)
  constexpr int NB = 2;
  auto ntile = [&](int jl) { return cdiv(K, 4 * 32 * jl) * cdiv(N, 128); };
  const int jl = ntile(4) >= 1024 ? 4 : (ntile(2) >= 1024 ? 2 : 1);
#define WA_LAUNCH(JL, FULL, EX)                                                                                   \
  hipLaunchKernelGGL((skinny_wgrad_adam_kernel<JL, FULL, NB, EX>), dim3((unsigned)cdiv(K, 4 * 32 * JL), (unsigned)cdiv(N, 128)), \
                     dim3(256), 0, as_stream(stream), dz, lddz, x, ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1,  \
                     beta2, eps, grad_scale, coef_dev)
#define WA_PICK(EX)                                                                                               \
  do {                                                                                                            \
    if (N % 128 == 0) {                                                                                           \
      if (jl == 4) WA_LAUNCH(4, true, EX); else if (jl == 2) WA_LAUNCH(2, true, EX); else WA_LAUNCH(1, true, EX); \
    } else {                                                                                                      \
      if (jl == 4) WA_LAUNCH(4, false, EX); else if (jl == 2) WA_LAUNCH(2, false, EX); else WA_LAUNCH(1, false, EX); \
    }                                                                                                             \
  } while (0)
  if (exact) WA_PICK(true); else WA_PICK(false);
#undef WA_PICK
#undef WA_LAUNCH
  PCAA_RETURN_LAUNCH_STATUS("pcaa_skinny_linear_wgrad_adam");
}

// gathered rows (data parallel): M = world * B <= 512 rows in chunks of 64
template <int MC>
static int launch_wgrad_adam_rows(const float* dz, long lddz, const float* x, long ldx, float* W, float* exp_avg,
                                  float* exp_avg_sq, long ldw, int M, int N, int K, float beta1, float beta2, float eps,
                                  float grad_scale, const float* coef_dev, hipStream_t st) {
  constexpr int NB = 2;
  constexpr size_t lds = (size_t)MC * 4 * 4 * 64 * sizeof(bf16x8);
  auto ntile = [&](int jl) { return cdiv(K, 4 * 32 * jl) * cdiv(N, 128); };
  const int jl = ntile(4) >= 1024 ? 4 : (ntile(2) >= 1024 ? 2 : 1);
  const bool full = N % 128 == 0;
#define WR_LAUNCH(JL, FULL)                                                                                            \
  do {                                                                                                                 \
    auto kern = skinny_wgrad_adam_rows_kernel<JL, FULL, NB, MC>;                                                       \
    static bool configured = false;                                                                                    \
    if (!configured) {                                                                                                 \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return 1;                                                                                                      \
      configured = true;                                                                                               \
    }                                                                                                                  \
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(K, 4 * 32 * JL), (unsigned)cdiv(N, 128)), dim3(256), lds, st, dz, lddz, x, \
                       ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1, beta2, eps, grad_scale, coef_dev);            \
  } while (0)
  if (full) { if (jl == 4) WR_LAUNCH(4, true); else if (jl == 2) WR_LAUNCH(2, true); else WR_LAUNCH(1, true); }
  else { if (jl == 4) WR_LAUNCH(4, false); else if (jl == 2) WR_LAUNCH(2, false); else WR_LAUNCH(1, false); }
#undef WR_LAUNCH
  return 0;
}

extern "C" int pcaa_skinny_linear_wgrad_adam_rows(const float* dz, long lddz, const float* x, long ldx, float* W,
                                                  float* exp_avg, float* exp_avg_sq, long ldw, int M, int N, int K,
                                                  float beta1, float beta2, float eps, float grad_scale,
                                                  const float* coef_dev, int rows_alloc, void* stream) {
  PCAA_CHECK_ARG(dz && x && W && exp_avg && exp_avg_sq && coef_dev, "pcaa_skinny_linear_wgrad_adam_rows: null pointer");
  PCAA_CHECK_ARG(M >= 1 && M <= 512 && N >= 1 && K >= 32 && K % 32 == 0,
                 "pcaa_skinny_linear_wgrad_adam_rows: unsupported shape M=%d N=%d K=%d (M <= 512, K %% 32 == 0)", M, N, K);
  {
    const int mc = M <= 64 ? 1 : (M <= 128 ? 2 : (M <= 256 ? 4 : 8));
    PCAA_CHECK_ARG(rows_alloc >= 64 * mc, "pcaa_skinny_linear_wgrad_adam_rows: x must be allocated (and finite) for %d rows, "
                   "got %d", 64 * mc, rows_alloc);
  }
  PCAA_CHECK_ARG(lddz >= N && ldx >= K && ldw >= K, "pcaa_skinny_linear_wgrad_adam_rows: bad leading dimensions");
  PCAA_CHECK_ARG((long)M * lddz < (1L << 31) && 512L * ldx < (1L << 30) && (long)N * ldw < (1L << 30) &&
                 ((uintptr_t)x % 4) == 0, "pcaa_skinny_linear_wgrad_adam_rows: operands beyond 32-bit offsets");
  hipStream_t st = as_stream(stream);
  int rc;
  if (M <= 64) rc = launch_wgrad_adam_rows<1>(dz, lddz, x, ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1, beta2, eps, grad_scale, coef_dev, st);
  else if (M <= 128) rc = launch_wgrad_adam_rows<2>(dz, lddz, x, ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1, beta2, eps, grad_scale, coef_dev, st);
  else if (M <= 256) rc = launch_wgrad_adam_rows<4>(dz, lddz, x, ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1, beta2, eps, grad_scale, coef_dev, st);
  else rc = launch_wgrad_adam_rows<8>(dz, lddz, x, ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1, beta2, eps, grad_scale, coef_dev, st);
  if (rc != 0) {
    pcaa_set_error("pcaa_skinny_linear_wgrad_adam_rows: cannot raise the dynamic LDS limit");
    return PCAA_ERR_LAUNCH;
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_skinny_linear_wgrad_adam_rows");
}

extern "C" int pcaa_skinny_linear_wgrad_adam(const float* dz, long lddz, const float* x, long ldx, float* W,
                                             float* exp_avg, float* exp_avg_sq, long ldw, int M, int N, int K,
                                             float beta1, float beta2, float eps, float grad_scale,
                                             const float* coef_dev, void* stream) {
  return skinny_wgrad_adam_impl(dz, lddz, x, ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1, beta2, eps, grad_scale,
                                coef_dev, 0, stream);
}
extern "C" int pcaa_skinny_linear_wgrad_adam_exact(const float* dz, long lddz, const float* x, long ldx, float* W,
                                                   float* exp_avg, float* exp_avg_sq, long ldw, int M, int N, int K,
                                                   float beta1, float beta2, float eps, float grad_scale,
                                                   const float* coef_dev, void* stream) {
  return skinny_wgrad_adam_impl(dz, lddz, x, ldx, W, exp_avg, exp_avg_sq, ldw, M, N, K, beta1, beta2, eps, grad_scale,
                                coef_dev, 1, stream);
}

// ------------------------------------------------------------------ packed (transposed bf16) gathered operands, round 6
extern "C" long pcaa_packed_chunk_elems(int N, int K) { return ((long)N + K) * 64; }

extern "C" int pcaa_pack_rows_t16(const float* dz, long lddz, int N, const float* x, long ldx, int K, int rows,
                                  void* chunk_bf16, void* stream) {
  PCAA_CHECK_ARG(dz && x && chunk_bf16, "pcaa_pack_rows_t16: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && rows <= 64 && N >= 1 && K >= 1, "pcaa_pack_rows_t16: unsupported shape rows=%d N=%d K=%d (rows <= 64)",
                 rows, N, K);
  PCAA_CHECK_ARG(lddz >= N && ldx >= K && ((uintptr_t)chunk_bf16 % 16) == 0, "pcaa_pack_rows_t16: bad leading dimensions / alignment");
  hipLaunchKernelGGL(pack_rows_t16_kernel, dim3((unsigned)cdiv(((long)N + K) * 8, 256)), dim3(256), 0, as_stream(stream), dz, lddz,
                     N, x, ldx, K, rows, reinterpret_cast<bf16_t*>(chunk_bf16));
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pack_rows_t16");
}

template <int NB>
static int launch_wgrad_adam_t16(const bf16_t* P, long chunk_stride, int MC, float* W, float* exp_avg, float* exp_avg_sq,
                                 long ldw, int N, int K, float beta1, float beta2, float eps, float grad_scale,
                                 const float* coef_dev, hipStream_t st) {
  const size_t lds = (size_t)MC * 4 * 4 * 64 * sizeof(bf16x8);
  auto ntile = [&](int jl) { return cdiv(K, 4 * 32 * jl) * cdiv(N, 128); };
  // columns per wave (32 JL).  Up to 2 chunks several workgroups share a CU and the single-process rule holds (>= 1024
  // workgroups); from 4 chunks on the dz panels leave room for ONE workgroup per CU (64-128 KB of LDS), so 256 run at a
  // time whatever the grid, and every workgroup pays a panel load (16 KB per chunk, from L2) + a barrier before its first
  // MFMA: the widest JL that still gives every CU >= 1.5 workgroups amortises that prologue over 4 column steps
  const int want = MC >= 4 ? 384 : 1024;
  const int jl = ntile(4) >= want ? 4 : (ntile(2) >= want ? 2 : 1);
  const bool full = N % 128 == 0;
#define WT_LAUNCH(JL, FULL)                                                                                            \
  do {                                                                                                                 \
    auto kern = skinny_wgrad_adam_t16_kernel<JL, FULL, NB>;                                                            \
    static size_t configured = 0;                                                                                      \
    if (lds > configured) {                                                                                            \
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds) != hipSuccess) \
        return 1;                                                                                                      \
      configured = lds;                                                                                                \
    }                                                                                                                  \
    hipLaunchKernelGGL(kern, dim3((unsigned)cdiv(K, 4 * 32 * JL), (unsigned)cdiv(N, 128)), dim3(256), lds, st, P, chunk_stride, \
                       MC, W, exp_avg, exp_avg_sq, ldw, N, K, beta1, beta2, eps, grad_scale, coef_dev);                \
  } while (0)
  if (full) { if (jl == 4) WT_LAUNCH(4, true); else if (jl == 2) WT_LAUNCH(2, true); else WT_LAUNCH(1, true); }
  else { if (jl == 4) WT_LAUNCH(4, false); else if (jl == 2) WT_LAUNCH(2, false); else WT_LAUNCH(1, false); }
#undef WT_LAUNCH
  return 0;
}

extern "C" int pcaa_skinny_linear_wgrad_adam_t16(const void* packed_bf16, long chunk_stride, int chunks, float* W,
                                                 float* exp_avg, float* exp_avg_sq, long ldw, int N, int K, float beta1,
                                                 float beta2, float eps, float grad_scale, const float* coef_dev,
                                                 void* stream) {
  PCAA_CHECK_ARG(packed_bf16 && W && exp_avg && exp_avg_sq && coef_dev, "pcaa_skinny_linear_wgrad_adam_t16: null pointer");
  PCAA_CHECK_ARG(chunks >= 1 && chunks <= 8 && N >= 1 && K >= 32 && K % 32 == 0,
                 "pcaa_skinny_linear_wgrad_adam_t16: unsupported shape chunks=%d N=%d K=%d (chunks <= 8, K %% 32 == 0)", chunks, N, K);
  PCAA_CHECK_ARG(chunk_stride >= pcaa_packed_chunk_elems(N, K) && chunk_stride % 8 == 0 && ((uintptr_t)packed_bf16 % 16) == 0,
                 "pcaa_skinny_linear_wgrad_adam_t16: chunk stride %ld below (N + K) * 64 or misaligned", chunk_stride);
  PCAA_CHECK_ARG(ldw >= K && (long)N * ldw < (1L << 30) && (long)chunks * chunk_stride * 2 < (1L << 31),
                 "pcaa_skinny_linear_wgrad_adam_t16: operands beyond 32-bit offsets");
  hipStream_t st = as_stream(stream);
  const bf16_t* P = reinterpret_cast<const bf16_t*>(packed_bf16);
  // one workgroup per CU once the dz panels of >= 4 chunks fill the LDS: more fragments of W / exp_avg / exp_avg_sq in
  // flight per wave then replace the second workgroup's (measured in tools/skinny_lab.py)
  const int rc = chunks >= 4
      ? launch_wgrad_adam_t16<4>(P, chunk_stride, chunks, W, exp_avg, exp_avg_sq, ldw, N, K, beta1, beta2, eps, grad_scale, coef_dev, st)
      : launch_wgrad_adam_t16<2>(P, chunk_stride, chunks, W, exp_avg, exp_avg_sq, ldw, N, K, beta1, beta2, eps, grad_scale, coef_dev, st);
  if (rc != 0) {
    pcaa_set_error("pcaa_skinny_linear_wgrad_adam_t16: cannot raise the dynamic LDS limit");
    return PCAA_ERR_LAUNCH;
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_skinny_linear_wgrad_adam_t16");
}
