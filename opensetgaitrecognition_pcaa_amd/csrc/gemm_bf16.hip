// 256x256-tile bf16 MFMA GEMM (gfx950) for the three big PointNet products:
//
//   forward   y[P,out]   = a[P,in]   . W[out,in]^T      (KC x KC, + BatchNorm column statistics)
//   dgrad     da[P,in]   = dy[P,out] . Wt[in,out]^T     (KC x KC)
//   wgrad     dW[out,in] = dy[P,out]^T . a[P,in]        (RC x RC, contraction over the P points,
//                                                        split-K with fp32 atomics)
//
// 512 threads = 8 waves as 2(M) x 4(N); each wave owns a 128x64 sub-tile = 4x2
// v_mfma_f32_32x32x16_bf16 accumulators (128 accumulator registers).  K advances
// 64 per step through a 2-stage LDS ring (one barrier per step): while the MFMAs
// of step t run from stage t&1, the global loads of step t+1 are in flight in
// registers and are written to the other stage before the barrier.
//
// LDS images (36,864 B per operand per stage, 147,456 B in all):
//   KC operand: [256 rows][64 k + 8 pad] bf16 (144-B rows) -- fragments by ds_read_b128,
//               conflict-free (9 r mod 16 distinct over a 16-lane group).
//   RC operand: [64 k][256 rows + 32 pad] bf16 (576-B rows = 144 dwords = 16 mod 64):
//               the tile is stored exactly as it lies in HBM (row-contiguous 16-B
//               chunks) and the MFMA fragment -- 8 consecutive k of one row -- is
//               gathered by two ds_read_b64_tr_b16 transpose reads; with that pitch
//               the 32 lanes of a half cover all 64 banks exactly once.
#include <hip/hip_ext.h>

#include <cstddef>
#include <mutex>
#include <type_traits>

#include "gemm_common.h"

namespace {

constexpr int KC = PCAA_LAYOUT_KC, RC = PCAA_LAYOUT_RC;
constexpr int BM = 256, BN = 256, BK = 64, NTHREADS = 512;
constexpr int FM = 4, FN = 2;                 // 32x32 fragments per wave in M, N
constexpr int P_KC = BK + 8;                  // 72 elements
constexpr int P_RC = 256 + 32;                // 288 elements
constexpr int TILE = 256 * P_KC;              // 18432 elements per operand per stage (== BK * P_RC)
static_assert(TILE == BK * P_RC, "both LDS images have the same size");
constexpr int LDS_BYTES = 2 /*stages*/ * 2 /*operands*/ * TILE * 2;

typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) s16x4 lds_s16x4;

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 b;
  b.x = (bf16_t)lo;
  b.y = (bf16_t)hi;
  return *reinterpret_cast<uint32_t*>(&b);
}
__device__ __forceinline__ uint4 load8(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 load8(const float* p) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  uint4 r;
  r.x = pack2(lo.x, lo.y); r.y = pack2(lo.z, lo.w); r.z = pack2(hi.x, hi.y); r.w = pack2(hi.z, hi.w);
  return r;
}

// 256 rows x 64 k = 2048 chunks of 8 elements, 4 per thread
template <typename T, int LAY>
__device__ __forceinline__ void load_tile(const T* __restrict__ base, long ld, int row0, int R, int k0,
                                          int kend, uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = tid + c * NTHREADS;
    uint4 v = {0u, 0u, 0u, 0u};
    if (LAY == KC) {
      const int r = q >> 3, kc = (q & 7) << 3;
      const int gr = row0 + r, gk = k0 + kc;
      if (gr < R && gk < kend) v = load8(base + (long)gr * ld + gk);
    } else {
      const int k = q >> 5, rc = (q & 31) << 3;
      const int gk = k0 + k, gr = row0 + rc;
      if (gk < kend && gr < R) v = load8(base + (long)gk * ld + gr);
    }
    reg[c] = v;
  }
}

template <int LAY>
__device__ __forceinline__ void store_tile(bf16_t* __restrict__ s, const uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = tid + c * NTHREADS;
    if (LAY == KC) {
      const int r = q >> 3, kc = (q & 7) << 3;
      *reinterpret_cast<uint4*>(&s[r * P_KC + kc]) = reg[c];
    } else {
      const int k = q >> 5, rc = (q & 31) << 3;
      *reinterpret_cast<uint4*>(&s[k * P_RC + rc]) = reg[c];
    }
  }
}

// per-lane element offset of the fragment of rows [row_base, row_base+32) at k-step 0
template <int LAY>
__device__ __forceinline__ int frag_offset(int row_base, int lane) {
  if (LAY == KC) return (row_base + (lane & 31)) * P_KC + 8 * (lane >> 5);
  const int j = lane & 15, mb = 16 * ((lane >> 4) & 1), h = lane >> 5;
  return (8 * h + (j >> 2)) * P_RC + row_base + mb + 4 * (j & 3);
}

template <int LAY>
__device__ __forceinline__ bf16x8 load_frag(const bf16_t* s, int off, int kk) {
  if (LAY == KC) return *reinterpret_cast<const bf16x8*>(s + off + kk);
  const bf16_t* p = s + off + kk * P_RC;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * P_RC));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

// Epilogue store.  The MFMA accumulator layout (column on the lane, rows in the
// registers) makes a direct bf16 store a 2-byte-per-lane, 64-B-per-row partial-line
// write -- measured ~30 us per 256x256 tile, more than the K=512 main loop.  bf16
// outputs are therefore transposed through the (now idle) LDS: each wave parks its
// 128x64 sub-tile as [row][col] bf16 and streams it out as 16 B per lane, 8 lanes =
// one whole 128-B line per row.  fp32 outputs / atomics already write 128 B per
// half-wave and go out directly.
template <typename TC, int PITCH>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, f32x16 (&acc)[FM][FN], bf16_t* smem,
                                               int tm, int tn, int tid, int split) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  const bool add_bias = p.bias != nullptr && (!p.atomic || split == 0);
  constexpr bool kBf16 = sizeof(TC) == 2;
  const bool wide = kBf16 && !p.atomic && (p.ldc % 8) == 0 && (p.N % 8) == 0 && ((uintptr_t)p.C % 16) == 0;
  if (wide) {
    bf16_t* w = smem + wave * 128 * PITCH;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const int gn = tn * BN + wn * 64 + j * 32 + l31;
      const float bv = (add_bias && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
          w[row * PITCH + j * 32 + l31] = (bf16_t)(acc[i][j][r] + bv);
        }
    }
    __syncthreads();
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C);
    const int cg = (lane & 7) * 8;
    const int gn = tn * BN + wn * 64 + cg;
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
      const int row = pass * 8 + (lane >> 3);
      const int gm = tm * BM + wm * 128 + row;
      const uint4 v = *reinterpret_cast<const uint4*>(&w[row * PITCH + cg]);
      if (gm < p.M && gn < p.N) *reinterpret_cast<uint4*>(&C[(long)gm * p.ldc + gn]) = v;
    }
    __syncthreads();   // the statistics reduction reuses this LDS
    return;
  }
  TC* C = reinterpret_cast<TC*>(p.C) + (long)split * p.c_split_stride;   // slab split-K (0 otherwise)
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int gn = tn * BN + wn * 64 + j * 32 + l31;
    const float bv = (add_bias && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
    for (int i = 0; i < FM; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = tm * BM + wm * 128 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (gm < p.M && gn < p.N) {
          const float v = acc[i][j][r] + bv;
          if (p.atomic) atomicAdd(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn, v);
          else store1(C + (long)gm * p.ldc + gn, v);
        }
      }
    }
  }
}

template <typename TA, typename TB, typename TC, int ALAY, int BLAY>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_big_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  int tm, tn;
  const int split = block_coords(p, (p.M + BM - 1) / BM, (p.N + BN - 1) / BN, tm, tn);

  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + BK - 1) / BK;
  const TA* A = reinterpret_cast<const TA*>(p.A);
  const TB* B = reinterpret_cast<const TB*>(p.B);

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int offA[FM], offB[FN];
#pragma unroll
  for (int i = 0; i < FM; ++i) offA[i] = frag_offset<ALAY>(wm * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < FN; ++j) offB[j] = frag_offset<BLAY>(wn * 64 + j * 32, lane);

  uint4 ra[4], rb[4];
  if (nt > 0) {
    load_tile<TA, ALAY>(A, p.lda, tm * BM, p.M, kbeg, kend, ra, tid);
    load_tile<TB, BLAY>(B, p.ldb, tn * BN, p.N, kbeg, kend, rb, tid);
    store_tile<ALAY>(smem, ra, tid);
    store_tile<BLAY>(smem + TILE, rb, tid);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const bf16_t* sA = smem + (t & 1) * 2 * TILE;
    const bf16_t* sB = sA + TILE;
    const bool more = (t + 1 < nt);
    if (more) {
      const int k0 = kbeg + (t + 1) * BK;
      load_tile<TA, ALAY>(A, p.lda, tm * BM, p.M, k0, kend, ra, tid);
      load_tile<TB, BLAY>(B, p.ldb, tn * BN, p.N, k0, kend, rb, tid);
    }
#pragma unroll
    for (int kk = 0; kk < BK; kk += 16) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = load_frag<ALAY>(sA, offA[i], kk);
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = load_frag<BLAY>(sB, offB[j], kk);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    if (more) {
      bf16_t* nA = smem + ((t + 1) & 1) * 2 * TILE;
      store_tile<ALAY>(nA, ra, tid);
      store_tile<BLAY>(nA + TILE, rb, tid);
    }
    __syncthreads();
  }

  // ---------------- epilogue: bias, store / atomic accumulate (the loop ended on a barrier)
  epilogue_store<TC, P_KC>(p, acc, smem, tm, tn, tid, split);
  // ---------------- BatchNorm column statistics of the bias-free accumulator
  if (p.colstats != nullptr) {
    float* red = reinterpret_cast<float*>(smem_raw);   // [2 stats][2 wm][256 cols]; loop ended on a barrier
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          s1 += v;
          s2 += v * v;
        }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (half == 0) {
        const int col = wn * 64 + j * 32 + l31;
        red[(0 * 2 + wm) * 256 + col] = s1;
        red[(1 * 2 + wm) * 256 + col] = s2;
      }
    }
    __syncthreads();
    const int stat = tid >> 8, col = tid & 255;
    const int gn = tn * BN + col;
    if (gn < p.N) {
      const double v = (double)red[(stat * 2 + 0) * 256 + col] + (double)red[(stat * 2 + 1) * 256 + col];
      unsafeAtomicAdd(&p.colstats[((long)(tm % p.nrep) * 2 + stat) * p.N + gn], v);
    }
  }
}

// ===========================================================================
// LDS-DMA kernel: both operands already bf16 in HBM, shapes whole tiles
// (M % 256 == N % 256 == 0, K-range % 64 == 0).  Tiles go HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write pass); the LDS images are
// unpadded, so the bank-conflict fix is an XOR swizzle applied to the per-lane
// SOURCE address (the DMA writes LDS linearly: base + lane*16) and again on the
// fragment reads:
//   KC image [256 rows][64 k] (128-B rows): 16-B granule g of row r holds global k-granule
//     g ^ ((r>>1)&7)  -> ds_read_b128 of 16 consecutive rows hits 16 distinct 16-B slots
//     (for the 32x32x16 fragment: 32 rows x one granule pair; for the 16x16x32 fragment:
//     16 rows x 4 granules -- both conflict-free with this swizzle).
//   RC image [64 k][256 rows] (512-B rows): 16-B granule c of k-row k holds global row-granule
//     c ^ 4*(k&3)     -> the 4 k-rows of a transpose read land on disjoint bank quarters.
// One barrier per 64-deep step.  Ring: THREE stages of the A operand, two of B (5 x 32 KB = all 160 KB of the LDS):
// between the MFMAs of step t a wave issues its pieces of B(t+1) and then of A(t+2) (one 1-KB piece in front of every
// 8th of the step's MFMAs); at the barrier it waits with a counted vmcnt(4) -- everything but its four youngest
// pieces, A(t+2).  See the kernel for why the depth goes to A, and tools/microbench/cu_load_bw.hip for the
// bytes-in-flight curve of a CU that motivates it.
//
// Measured and rejected (tools/gemm_lab.py history, docs/LAB_LOG.md section 4): touching the streamed operand's lines in L2
// a few K steps ahead with one plain global_load_dword per lane and step (a software L2 prefetch behind a counted
// vmcnt(1)) made every shape 3-10 % SLOWER (those loads queue in front of the pieces); round 1's five-slot ring gave
// the third stage to B -- the L2-resident weights -- and measured nothing.
// ===========================================================================
constexpr int D_TILE = 256 * 64;                       // elements per operand per stage (32 KB)
constexpr int D_LDS_BYTES = 5 * D_TILE * 2;            // 163840: A in a ring of three stages, B of two -- all of the LDS

// one of the 4 pieces a wave moves per tile
// PERM (KC, the B operand of the 16x16x32 kernels): LDS row l = 16 j + c of every 64-row group holds operand row
// 4 c + j of that group.  The fragment reads do not change (lane c of column block j still reads LDS row 16 j + c,
// conflict-free as before), but the accumulator a lane holds in block j is now output column 4 c + j: the four
// blocks of a lane are four ADJACENT columns of one row, and the epilogue stores them straight from registers
// (8 B per lane, 16 lanes = one 128-B line per row) -- no LDS transpose, no lane exchange (see epilogue_full_tile).
__device__ __forceinline__ int perm_row(int l) { return (l & ~63) + 4 * (l & 15) + ((l >> 4) & 3); }

template <int LAY, bool PERM = false>
__device__ __forceinline__ void dma_piece(const bf16_t* __restrict__ base, long ld, int row0, int R, int k0,
                                          bf16_t* s_tile, int wave, int lane, int j) {
  const int p = wave * 4 + j;       // 1-KB piece index, 32 per tile
  const bf16_t* src;
  if (LAY == KC) {
    const int r = 8 * p + (lane >> 3);
    const int g = (lane & 7) ^ ((r >> 1) & 7);
    src = base + (long)min(row0 + (PERM ? perm_row(r) : r), R - 1) * ld + k0 + 8 * g;
  } else {
    const int k = 2 * p + (lane >> 5);
    const int c = (lane & 31) ^ (4 * (k & 3));
    src = base + (long)(k0 + k) * ld + row0 + 8 * c;
  }
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                   (__attribute__((address_space(3))) void*)(s_tile + p * 512), 16, 0, 0);
}

template <int LAY>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ base, long ld, int row0, int R, int k0,
                                         bf16_t* s_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) dma_piece<LAY>(base, ld, row0, R, k0, s_tile, wave, lane, j);
}

// The same piece through a buffer resource (buffer_load_dwordx4 ... offen lds): the per-lane part of the source address
// is a 32-bit offset that never changes (row-in-piece, swizzled granule; it depends on the piece only through its
// parity), everything else -- tile origin, piece, K step -- is wave-uniform and rides in the scalar offset.  A piece
// then costs the vector ALU nothing (the flat form spent ~7 VALU instructions per piece on 64-bit address arithmetic,
// on the issue port the MFMAs share).  Whole-tile shapes only: no row clamp; operands below 4 GiB.
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
// per-lane byte offsets for pieces of even / odd index
template <int LAY, bool PERM = false>
__device__ __forceinline__ void piece_lane_offsets(long ld, int lane, unsigned (&voff)[2]) {
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    if (LAY == KC) {
      const int g = (lane & 7) ^ ((4 * e + (lane >> 4)) & 7);            // r >> 1 = 4 p + (lane >> 4)
      voff[e] = (unsigned)((PERM ? 4 : 1) * (lane >> 3) * ld * 2 + 16 * g);    // PERM: rows of a piece lie 4 apart
    } else {
      const int c = (lane & 31) ^ (4 * ((2 * e + (lane >> 5)) & 3));     // k & 3 = (2 p + (lane >> 5)) & 3
      voff[e] = (unsigned)((lane >> 5) * ld * 2 + 16 * c);
    }
  }
}
// wave (scalar) and j select the piece p = 4 wave + j; row0 / k0 as in dma_piece
template <int LAY, bool PERM = false>
__device__ __forceinline__ void dma_piece_buf(buf_rsrc_t rsrc, long ld, int row0, int k0, bf16_t* s_tile, int wave,
                                              const unsigned (&voff)[2], int j, long extra = 0) {
  const int p = wave * 4 + j;
  // PERM: piece p = LDS rows 8p .. 8p+7 = lanes c = 8 (p & 1) + 0..7 of column block (p >> 1) & 3 of group p >> 3
  const int prow = PERM ? 64 * (p >> 3) + 32 * (p & 1) + ((p >> 1) & 3) : 8 * p;
  const unsigned soff = LAY == KC ? (unsigned)(((long)(row0 + prow) * ld + k0 + extra) * 2)
                                  : (unsigned)(((long)(k0 + 2 * p) * ld + row0 + extra) * 2);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, (__attribute__((address_space(3))) void*)(s_tile + p * 512), 16,
                                           voff[j & 1], soff, 0, 0);
}

// Accumulator layouts of the two bf16 MFMA shapes (a wave owns 128 rows x 64 columns of the tile):
//   MF = 32: v_mfma_f32_32x32x16_bf16, 4 x 2 blocks of f32x16; MF = 16: v_mfma_f32_16x16x32_bf16, 8 x 4 blocks of f32x4.
// In both, registers r (even) and r+1 of a block are two consecutive rows of one column, neighbouring lanes
// hold neighbouring columns, and the lanes that share a column differ in the high lane bits only.
template <int MF> struct AccLayout;
template <> struct AccLayout<32> {
  typedef f32x16 vec;
  static constexpr int MB = 4, NB = 2, NR = 16;
  [[maybe_unused]] static constexpr int BR = 32;
  static __device__ __forceinline__ int row(int i, int r, int lane) { return i * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5); }
  static __device__ __forceinline__ int col(int j, int lane) { return j * 32 + (lane & 31); }
  static __device__ __forceinline__ float colreduce(float v) { return v + __shfl_xor(v, 32, 64); }
  static __device__ __forceinline__ bool col_leader(int lane) { return lane < 32; }
};
template <> struct AccLayout<16> {
  typedef f32x4 vec;
  static constexpr int MB = 8, NB = 4, NR = 4, BR = 16;
  static __device__ __forceinline__ int row(int i, int r, int lane) { return i * 16 + 4 * (lane >> 4) + r; }
  // (the B operand's rows are permuted on their way into the LDS, dma_piece<KC, PERM>: block j of lane c is column 4 c + j)
  static __device__ __forceinline__ int col(int j, int lane) { return 4 * (lane & 15) + j; }
  static __device__ __forceinline__ float colreduce(float v) {
    v += __shfl_xor(v, 16, 64);
    return v + __shfl_xor(v, 32, 64);
  }
  static __device__ __forceinline__ bool col_leader(int lane) { return lane < 16; }
};


// LDS scratch of the epilogues: the column statistics' [2 stats][2 wm][256 cols] floats, in the ring's last two slots (the
// ones the NEXT tile's first stages do not use, see the kernel).  (Rounds 1-2 also kept eight 4.5-KB wave images here:
// the accumulators went through the LDS to become row-contiguous 16-B stores.  With the B operand's rows permuted on
// their way in -- dma_piece<KC, PERM> -- a lane's four column blocks ARE four adjacent columns, and the tile leaves
// straight from the registers.)
constexpr int EP_RED_OFFSET = 0;
// workgroup barrier that leaves vector-memory operations (the next tile's LDS-DMA pieces, this tile's C stores) alone
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Full-tile epilogue (M, N multiples of 256: no bounds checks).
//   bf16 out : park_bf16, then 16-B-per-lane row-contiguous stores (8 lanes = one whole 128-B line per row).
//   fp32 out : plain stores (slab split-K: C is offset by split * c_split_stride) or atomics.
// AFFINE (bf16 out, eval-mode BatchNorm): the stored value is ELU(scale[col]*acc + shift[col]) -- with running
// statistics the BatchNorm is a per-channel affine map known BEFORE the product, so the activation leaves the
// GEMM directly and the separate BN+ELU pass over [P, ch] (read + write) of the train-mode path is gone.
template <typename TC, int MF, bool AFFINE>
__device__ __forceinline__ void epilogue_full_tile(const GemmParams& p,
                                                   typename AccLayout<MF>::vec (&acc)[AccLayout<MF>::MB][AccLayout<MF>::NB],
                                                   bf16_t* smem, int tm, int tn, int tid, int split) {
  typedef AccLayout<MF> L;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const bool add_bias = p.bias != nullptr && (!p.atomic || split == 0);
  if constexpr (sizeof(TC) == 2) {
    // 16x16x32 layout: lane (c = lane & 15, q = lane >> 4) holds rows 16 i + 4 q + r of columns 4 c .. 4 c + 3 (one per
    // column block j): two v_cvt_pk_bf16_f32 and one 8-B store per row; the 16 lanes of a q are one 128-B line
    static_assert(MF == 16, "bf16 output leaves the 16x16x32 kernels only");
    const int l15 = lane & 15, q = lane >> 4;
    float bv[4], esc[4], esh[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int gc = tn * BN + wn * 64 + 4 * l15 + j;
      bv[j] = add_bias ? p.bias[gc] : 0.f;
      esc[j] = AFFINE ? p.ep_scale[gc] : 1.f;
      esh[j] = AFFINE ? p.ep_shift[gc] : 0.f;
    }
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (long)(tm * BM + wm * 128 + 4 * q) * p.ldc + tn * BN + wn * 64 + 4 * l15;
    auto out = [&](float v, int j) {
      if constexpr (AFFINE) {
        v = fmaf(v + bv[j], esc[j], esh[j]);
        return v > 0.f ? v : __expf(v) - 1.f;
      } else {
        return add_bias ? v + bv[j] : v;       // (uniform) the BatchNorm layers pass no bias
      }
    };
#pragma unroll
    for (int i = 0; i < L::MB; ++i)
#pragma unroll
      for (int r = 0; r < L::NR; ++r) {
        uint2 o;
        o.x = pack2(out(acc[i][0][r], 0), out(acc[i][1][r], 1));
        o.y = pack2(out(acc[i][2][r], 2), out(acc[i][3][r], 3));
        *reinterpret_cast<uint2*>(C + (long)(i * 16 + r) * p.ldc) = o;
      }
  } else {
    float* C = reinterpret_cast<float*>(p.C) + (long)split * p.c_split_stride +
               (long)(tm * BM + wm * 128) * p.ldc + tn * BN + wn * 64;
    if constexpr (MF == 16) {
      // the lane's four column blocks are four adjacent floats of a row: one 16-B store (16 lanes = 256 B of the row)
      if (!p.atomic && (p.ldc & 3) == 0 && ((uintptr_t)p.C & 15) == 0 && (p.c_split_stride & 3) == 0) {
        const int l15 = lane & 15, q = lane >> 4;
        const f32x4 bv = add_bias ? load4(p.bias + tn * BN + wn * 64 + 4 * l15) : f32x4{0.f, 0.f, 0.f, 0.f};
        float* Cl = C + (long)(4 * q) * p.ldc + 4 * l15;
#pragma unroll
        for (int i = 0; i < L::MB; ++i)
#pragma unroll
          for (int r = 0; r < L::NR; ++r)
            *reinterpret_cast<f32x4*>(Cl + (long)(i * 16 + r) * p.ldc) =
                f32x4{acc[i][0][r], acc[i][1][r], acc[i][2][r], acc[i][3][r]} + bv;
        return;
      }
    }
#pragma unroll
    for (int j = 0; j < L::NB; ++j) {
      const int cl = L::col(j, lane);
      const float bv = add_bias ? p.bias[tn * BN + wn * 64 + cl] : 0.f;
#pragma unroll
      for (int i = 0; i < L::MB; ++i)
#pragma unroll
        for (int r = 0; r < L::NR; ++r) {
          float* dst = C + (long)L::row(i, r, lane) * p.ldc + cl;
          const float v = acc[i][j][r] + bv;
          if (p.atomic) atomicAdd(dst, v);
          else *dst = v;
        }
    }
  }
}

// BatchNorm column statistics (sum, sum of squares) of the bias-free accumulator; the caller's LDS is free
// (the epilogue above ended on a barrier).  fp64 atomics into replica tm % nrep.
template <int MF>
__device__ __forceinline__ void epilogue_colstats(const GemmParams& p,
                                                  typename AccLayout<MF>::vec (&acc)[AccLayout<MF>::MB][AccLayout<MF>::NB],
                                                  unsigned char* smem_raw, int tm, int tn, int tid) {
  typedef AccLayout<MF> L;
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  float* red = reinterpret_cast<float*>(smem_raw) + EP_RED_OFFSET / 2;   // [2 stats][2 wm][256 cols], behind the images
  typedef float f32x2 __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int j = 0; j < L::NB; ++j) {
    f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};      // two partial sums each: v_pk_add_f32 / v_pk_fma_f32
#pragma unroll
    for (int i = 0; i < L::MB; ++i)
#pragma unroll
      for (int r = 0; r < L::NR; r += 2) {
        const f32x2 v = {acc[i][j][r], acc[i][j][r + 1]};
        a1 += v;
        a2 = __builtin_elementwise_fma(v, v, a2);
      }
    float s1 = L::colreduce(a1.x + a1.y);
    float s2 = L::colreduce(a2.x + a2.y);
    if (L::col_leader(lane)) {
      const int col = wn * 64 + L::col(j, lane);
      red[(0 * 2 + wm) * 256 + col] = s1;
      red[(1 * 2 + wm) * 256 + col] = s2;
    }
  }
  lds_barrier();
  const int stat = tid >> 8, col = tid & 255;
  const double v = (double)red[(stat * 2 + 0) * 256 + col] + (double)red[(stat * 2 + 1) * 256 + col];
  unsafeAtomicAdd(&p.colstats[((long)(tm % p.nrep) * 2 + stat) * p.N + tn * BN + col], v);
}

// Eval-mode LAST PointNet layer: BatchNorm (affine) + ELU + the mean over the N points of a frame
// (AvgPool2d((1,N)), models.py:242-243, :282) straight from the accumulators: a wave holds 128 rows x 64
// columns of the tile = groups of 32*IPG consecutive rows (IPG = N/32 in {1,2,4}), so a group's column mean is
// a sum over the lane's registers plus the cross-lane fold of the lanes that share the column; the [P, ch]
// activation is never written or re-read (2 x 8 GB per 1024 sequences at N=128).  out fp32 [P/N, ch].
template <int MF, int IPG>
__device__ __forceinline__ void epilogue_affine_meanpool(const GemmParams& p,
                                                         typename AccLayout<MF>::vec (&acc)[AccLayout<MF>::MB][AccLayout<MF>::NB],
                                                         int tm, int tn, int tid) {
  typedef AccLayout<MF> L;
  constexpr int BPG = 32 * IPG / L::BR;          // accumulator row-blocks per group
  static_assert(L::MB % BPG == 0, "groups must not straddle waves");
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  float* out = reinterpret_cast<float*>(p.C);
  const float inv_n = 1.f / (32 * IPG);
#pragma unroll
  for (int j = 0; j < L::NB; ++j) {
    const int col = tn * BN + wn * 64 + L::col(j, lane);
    const float esc = p.ep_scale[col], esh = p.ep_shift[col];
#pragma unroll
    for (int g0 = 0; g0 < L::MB; g0 += BPG) {
      float sum = 0.f;
#pragma unroll
      for (int i = g0; i < g0 + BPG; ++i)
#pragma unroll
        for (int r = 0; r < L::NR; ++r) {
          const float z = fmaf(acc[i][j][r], esc, esh);
          sum += z > 0.f ? z : __expf(z) - 1.f;
        }
      sum = L::colreduce(sum);
      const long grp = ((long)tm * BM + wm * 128 + g0 * L::BR) / (32 * IPG);
      if (L::col_leader(lane)) out[grp * p.ldc + col] = sum * inv_n;
    }
  }
}

// dgrad epilogue fused with the BatchNorm+ELU backward of the layer BELOW (pcaa_gemm_dgrad_bn):
// the tile of da = dy.Wt never reaches HBM as such -- on its way out each lane loads the 8 B of that layer's stored
// pre-activation y that belong to its four adjacent columns of a row and writes
//   dz = da * ELU'(y*scale + shift)
// while accumulating the column sums {dz, dz * (y-mean)*rstd} the BatchNorm backward needs.  That
// replaces a separate pass that re-read da and y (0.31 ms per step for PointNet layers 2-3).
// POINTS: the layer below is the first PointNet layer on its recompute path -- its pre-activation was
// never stored, y[row][col] = sum_c x[row][c] * W1[col][c] (C <= 8 point features) is rebuilt here.
// TE = float (the split-fp16 parity mode, round 3): y and dz are fp32 (16 B per lane and row) and ELU' is the exact
// expression of the separate pass (bn_act_bwd_dz_kernel<float>), not the exp2 form of the bf16 mode.
template <int MF, bool POINTS, typename TE = bf16_t>
__device__ __forceinline__ void epilogue_dgrad_bn(const GemmParams& p,
                                                  typename AccLayout<MF>::vec (&acc)[AccLayout<MF>::MB][AccLayout<MF>::NB],
                                                  bf16_t* smem, int tm, int tn, int tid) {
  static_assert(MF == 16, "built for the 16x16x32 accumulator layout (a lane = 4 adjacent columns)");
  constexpr bool kF32 = sizeof(TE) == 4;
  static_assert(!(kF32 && POINTS), "the recompute form exists for the bf16 mode only");
  typedef AccLayout<MF> L;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 2, wn = wave & 3;
  const int l15 = lane & 15, q = lane >> 4;
  const int c0 = tn * BN + wn * 64 + 4 * l15;                  // the lane's first column
  const long row0 = (long)tm * BM + wm * 128 + 4 * q;          // its first row (rows 16 i + r further on)
  TE* C = reinterpret_cast<TE*>(p.C) + row0 * p.ldc + c0;
  const TE* Y = reinterpret_cast<const TE*>(p.ep_y) + row0 * p.ldc + c0;               // same shape and ld as C
  // The element-wise part runs on column PAIRS with packed fp32 instructions: ELU'(z) = exp(min(z, 0)) =
  // exp2(min(z log2e, 0)) with log2e folded into the affine coefficients (no compare / select), yhat = y rstd - mean rstd
  // as one fused multiply-add.
  f32x2 sc2[2], sh2[2], rs2[2], nm2[2], s1[2], s2[2];
  {
    constexpr float kLog2e = kF32 ? 1.f : 1.4426950408889634f;
    const f32x4 a = load4(p.ep_scale + c0), b = load4(p.ep_shift + c0), c = load4(p.ep_mean + c0), d = load4(p.ep_rstd + c0);
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      sc2[h] = f32x2{a[2 * h] * kLog2e, a[2 * h + 1] * kLog2e};
      sh2[h] = f32x2{b[2 * h] * kLog2e, b[2 * h + 1] * kLog2e};
      rs2[h] = f32x2{d[2 * h], d[2 * h + 1]};
      nm2[h] = f32x2{-c[2 * h] * d[2 * h], -c[2 * h + 1] * d[2 * h + 1]};
      s1[h] = f32x2{0.f, 0.f};
      s2[h] = f32x2{0.f, 0.f};
    }
  }
  const int xc = p.ep_xc;
  float w1[4][8];
  const float* X = nullptr;
  if (POINTS) {
    X = p.ep_x + row0 * xc;
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) w1[j][c] = c < xc ? p.ep_w1[(long)(c0 + j) * xc + c] : 0.f;
  }
  typedef typename std::conditional<kF32, f32x4, uint2>::type yraw_t;
  yraw_t yv[2][4];                       // the stored pre-activations of one 16-row block, requested one block ahead
  auto load_y = [&](int i, int b) {
#pragma unroll
    for (int r = 0; r < 4; ++r) yv[b][r] = *reinterpret_cast<const yraw_t*>(Y + (long)(i * 16 + r) * p.ldc);
  };
  if (!POINTS) load_y(0, 0);
#pragma unroll
  for (int i = 0; i < L::MB; ++i) {
    if (!POINTS && i + 1 < L::MB) load_y(i + 1, (i + 1) & 1);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x2 y2[2];
      if (POINTS) {
        const float* xr = X + (long)(i * 16 + r) * xc;
        float xv[8];
        if (xc == 4) {
          const f32x4 t = load4(xr);
          xv[0] = t.x; xv[1] = t.y; xv[2] = t.z; xv[3] = t.w; xv[4] = xv[5] = xv[6] = xv[7] = 0.f;
        } else {
#pragma unroll
          for (int c = 0; c < 8; ++c) xv[c] = c < xc ? xr[c] : 0.f;
        }
        float yy[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) {
          float a = 0.f;
#pragma unroll
          for (int c = 0; c < 8; ++c) a = fmaf(w1[j][c], xv[c], a);             // same order as pointnet_in.hip
          yy[j] = a;
        }
        y2[0] = f32x2{yy[0], yy[1]};
        y2[1] = f32x2{yy[2], yy[3]};
      } else if constexpr (kF32) {
        const f32x4 w = yv[i & 1][r];
        y2[0] = f32x2{w.x, w.y};
        y2[1] = f32x2{w.z, w.w};
      } else {
        const uint2 w = yv[i & 1][r];
        y2[0] = f32x2{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u)};
        y2[1] = f32x2{__uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
      }
      f32x2 dq[2];
#pragma unroll
      for (int h = 0; h < 2; ++h) {
        const f32x2 dav = {acc[i][2 * h][r], acc[i][2 * h + 1][r]};
        f32x2 g;
        if constexpr (kF32) {
          // the separate pass's arithmetic (elementwise.hip, bn_act_bwd_dz_kernel<float>): z = y*scale + shift, yhat = (y - mean)*rstd
          g = f32x2{elu_grad_from_pre(y2[h].x * sc2[h].x + sh2[h].x), elu_grad_from_pre(y2[h].y * sc2[h].y + sh2[h].y)};
        } else {
          f32x2 z2 = __builtin_elementwise_fma(y2[h], sc2[h], sh2[h]);
          z2 = __builtin_elementwise_min(z2, f32x2{0.f, 0.f});
          g = f32x2{__builtin_amdgcn_exp2f(z2.x), __builtin_amdgcn_exp2f(z2.y)};
        }
        const f32x2 d2 = dav * g;
        s1[h] += d2;
        s2[h] = __builtin_elementwise_fma(d2, __builtin_elementwise_fma(y2[h], rs2[h], nm2[h]), s2[h]);
        dq[h] = d2;
      }
      if constexpr (kF32) {
        *reinterpret_cast<f32x4*>(C + (long)(i * 16 + r) * p.ldc) = f32x4{dq[0].x, dq[0].y, dq[1].x, dq[1].y};
      } else {
        uint2 o;
        o.x = pack2(dq[0].x, dq[0].y);
        o.y = pack2(dq[1].x, dq[1].y);
        *reinterpret_cast<uint2*>(C + (long)(i * 16 + r) * p.ldc) = o;
      }
    }
  }
  // the four lanes q = 0..3 of a column quad hold partial sums over different rows: fold them (2 steps)
  float t1[4], t2[4];
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    t1[c] = L::colreduce(s1[c >> 1][c & 1]);
    t2[c] = L::colreduce(s2[c >> 1][c & 1]);
  }
  float* red = reinterpret_cast<float*>(smem) + EP_RED_OFFSET / 2;      // [2 stats][2 wm][256 cols]
  if (L::col_leader(lane)) {
    *reinterpret_cast<f32x4*>(&red[(0 * 2 + wm) * 256 + wn * 64 + 4 * l15]) = f32x4{t1[0], t1[1], t1[2], t1[3]};
    *reinterpret_cast<f32x4*>(&red[(1 * 2 + wm) * 256 + wn * 64 + 4 * l15]) = f32x4{t2[0], t2[1], t2[2], t2[3]};
  }
  lds_barrier();
  const int stat = tid >> 8, col = tid & 255;
  const double v = (double)red[(stat * 2 + 0) * 256 + col] + (double)red[(stat * 2 + 1) * 256 + col];
  unsafeAtomicAdd(&p.colstats[((long)(tm % p.nrep) * 2 + stat) * p.N + tn * BN + col], v);
}

// what leaves the kernel
enum { EPI_PLAIN = 0, EPI_DGRAD_BN = 1, EPI_DGRAD_BN_POINTS = 2, EPI_AFFINE = 3, EPI_POOL1 = 4, EPI_POOL2 = 5, EPI_POOL4 = 6 };

// per-lane element offset of fragment rows [row_base, row_base+32) at k-step 0 (row_base % 32 == 0), 32x32x16 shape
template <int LAY>
__device__ __forceinline__ int dma_frag_offset(int row_base, int lane) {
  if (LAY == KC) return (row_base + (lane & 31)) * 64;
  const int j = lane & 15, mb = 16 * ((lane >> 4) & 1), h = lane >> 5;
  const int q = j >> 2, ch = row_base + mb + 4 * (j & 3);
  return (8 * h + q) * 256 + ((((ch >> 3) ^ (4 * q)) << 3) | (ch & 7));
}

template <int LAY>
__device__ __forceinline__ bf16x8 dma_load_frag(const bf16_t* s, int off, int kstep, const int (&kofs)[4]) {
  if (LAY == KC) return *reinterpret_cast<const bf16x8*>(s + off + kofs[kstep]);
  const bf16_t* p = s + off + kstep * 16 * 256;
  const s16x4 lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p));
  const s16x4 hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_s16x4*)(p + 4 * 256));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  return u.v;
}

// the two MFMA shapes on bf16 operands, or (F16: the split-operand instantiations) on fp16 operands -- same operand
// bytes, same lane maps, same rate; only the element format differs
template <bool F16>
__device__ __forceinline__ f32x16 mfma_32x32x16(bf16x8 a, bf16x8 b, f32x16 c) {
  if constexpr (F16) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  }
}
template <bool F16>
__device__ __forceinline__ f32x4 mfma_16x16x32(bf16x8 a, bf16x8 b, f32x4 c) {
  if constexpr (F16) {
    typedef _Float16 h8 __attribute__((ext_vector_type(8)));
    return __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(h8, a), __builtin_bit_cast(h8, b), c, 0, 0, 0);
  } else {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  }
}

// end of a K step: the next stage's DMA pieces have landed (every wave waits for its own, then the barrier)
// every wave waits for its own pieces of the stage the next step reads, then the barrier; keep_far: the wave's four
// youngest pieces (the A stage two steps ahead) stay in flight
__device__ __forceinline__ void step_barrier(bool keep_far) {
  if (keep_far) asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)\n\ts_barrier" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

#include "gemm_v2.h"      // round 4: the 4-wave KC x KC tile loop (namespace v2)

template <typename TC, int ALAY, int BLAY, int EPI, int MF, bool BUF, bool SPLIT = false>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_dma_kernel(GemmParams p) {
  typedef AccLayout<MF> L;
  static_assert(MF == 32 || (ALAY == KC && BLAY == KC), "the 16x16x32 fragments are built for KC operands");
  constexpr bool PERMB = MF == 16;     // B's rows permuted into the LDS: a lane's four column blocks are adjacent columns
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);        // scalar: piece indices and LDS bases are SGPR math
  const int wm = wave >> 2, wn = wave & 3;
  // Tiles: a launch without K splits may start fewer workgroups than tiles (one per CU); workgroup b then walks the
  // tiles b, b + gridDim.x, ... of the XCD-aware order (gridDim.x is a multiple of 8: all of them on b's XCD, and
  // the workgroups of an XCD are on neighbouring tiles at any time, as with one workgroup per tile).
  constexpr bool PERSIST = ALAY == KC;             // the wgrad (RC x RC) launches split K: one tile per workgroup
  const int nbm = p.M / BM, nbn = p.N / BN;
  int vb = blockIdx.x;
  int tm, tn;
  const int split = block_coords(p, nbm, nbn, tm, tn);

  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg) / BK;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  // buffer resources over the whole operands (A: M x lda or K x lda elements, B likewise)
  buf_rsrc_t rA = make_rsrc(A, 0), rB = make_rsrc(B, 0);
  unsigned voA[2] = {0, 0}, voB[2] = {0, 0};
  if (BUF) {
    const int krows = SPLIT ? p.seg_len : p.K;           // rows of a row-contracted operand as it lies in memory
    rA = make_rsrc(A, (long)(ALAY == KC ? p.M : krows) * p.lda * 2);
    rB = make_rsrc(B, (long)(BLAY == KC ? p.N : krows) * p.ldb * 2);
    piece_lane_offsets<ALAY>(p.lda, lane, voA);
    piece_lane_offsets<BLAY, PERMB>(p.ldb, lane, voB);
  }
  // per-lane constants of the K loop are rebuilt at the top of every tile from an opaque copy of the lane id, so that
  // they do not stay in registers across the epilogue (which needs them all: accumulators + 64 of operands)
  int ln = lane;
  // one 1-KB piece of the A (which = 0) or B (1) tile of the K step starting at k0 into stage image s_tile
  auto piece_of = [&](int tmx, int tnx, int which, int k0, bf16_t* s_tile, int j) {
    long extra = 0;
    if constexpr (SPLIT) {
      // split-fp16 operands: K step k0 of the 3 * seg_len long contraction lies in segment seg (scalar arithmetic:
      // k0 is wave-uniform); that segment's hi / lo half of the operand starts `extra` elements further on
      const int seg = (k0 >= p.seg_len ? 1 : 0) + (k0 >= 2 * p.seg_len ? 1 : 0);
      k0 -= seg * p.seg_len;
      const long* so = which == 0 ? p.seg_off_a : p.seg_off_b;
      extra = seg == 0 ? so[0] : (seg == 1 ? so[1] : so[2]);
    }
    if (BUF) {
      if (which == 0) dma_piece_buf<ALAY>(rA, p.lda, tmx * BM, k0, s_tile, wave, voA, j, extra);
      else dma_piece_buf<BLAY, PERMB>(rB, p.ldb, tnx * BN, k0, s_tile, wave, voB, j, extra);
    } else {
      if (which == 0) dma_piece<ALAY>(A + extra, p.lda, tmx * BM, p.M, k0, s_tile, wave, lane, j);
      else dma_piece<BLAY, PERMB>(B + extra, p.ldb, tnx * BN, p.N, k0, s_tile, wave, lane, j);
    }
  };
  auto piece = [&](int which, int k0, bf16_t* s_tile, int j) { piece_of(tm, tn, which, k0, s_tile, j); };
  // LDS images of the operands' stages: slots A0 A1 A2 B0 B1 (5 x 32 KB = all of the LDS).  A -- the operand
  // streamed from HBM in the forward / dgrad products: a quarter of its lines miss the L2 and take ~2 us -- has THREE
  // stages and is requested TWO K steps ahead; B (there: the L2-resident weights) keeps two.  A stage lands as a
  // whole only when its slowest piece has: with one step of distance the step waited for that HBM round trip
  // (2 750 cycles for a stage against 2 048 of MFMA, round 1's stamps); same-box A/B of the two rings on the three
  // PointNet shapes: forward +9..12 %, fused dgrad +8..12 %, wgrad (both operands streamed) +3..4 %.
  // Order in memory: A0 A1 B0 | A2 B1.  A tile's FIRST stages -- A(0), B(0), A(1) -- go to the first three slots;
  // the epilogue works in the last two (64 KB: eight 4.5-KB wave images + the statistics scratch), so a workgroup
  // that has another tile to do requests that tile's first stages BEFORE its epilogue and they land beside it.
  auto slotA = [&](int i) { return smem + (i < 2 ? i : 3) * D_TILE; };
  auto slotB = [&](int i) { return smem + (i == 0 ? 2 : 4) * D_TILE; };
  bf16_t* epi = smem + 3 * D_TILE;
  // first stages of tile (tmx, tnx): the order the first barrier's counted wait relies on
  auto first_stages = [&](int tmx, int tnx) {
    if (nt > 0) {
#pragma unroll
      for (int j = 0; j < 4; ++j) piece_of(tmx, tnx, 0, kbeg, slotA(0), j);
#pragma unroll
      for (int j = 0; j < 4; ++j) piece_of(tmx, tnx, 1, kbeg, slotB(0), j);
      if (nt > 1) {
#pragma unroll
        for (int j = 0; j < 4; ++j) piece_of(tmx, tnx, 0, kbeg + BK, slotA(1), j);
      }
    }
  };
  typename L::vec acc[L::MB][L::NB];
  first_stages(tm, tn);
  bool primed = false;             // this tile's first stages were requested during the previous tile's epilogue
  // Which tile next?  With p.sched the workgroups of an XCD draw tickets from that XCD's counter (positions
  // gridDim.x / 8, ... of its range: the first gridDim.x / 8 are the workgroups' own first tiles), so a workgroup
  // that got its CU late -- the chip shared with another stream's kernels or a collective -- simply does fewer
  // tiles.  Thread 0 draws the ticket at the top of a tile (the round trip hides under the K loop) and hands it
  // to the other waves through LDS when the loop is over.  Every workgroup ends on a ticket beyond the range; the
  // last one to finish resets the counters for the next launch.  No waiting on other workgroups anywhere.
  int* const sched = PERSIST && !p.split_fast && (int)gridDim.x < nbm * nbn ? p.sched : nullptr;
  int* const ticket_lds = reinterpret_cast<int*>(epi + EP_RED_OFFSET) + 2 * 2 * 256;      // behind the statistics scratch
  for (;;) {
  int ticket = 0;
  if (sched != nullptr && tid == 0) ticket = atomicAdd(&sched[vb & 7], 1);
#pragma unroll
  for (int i = 0; i < L::MB; ++i)
#pragma unroll
    for (int j = 0; j < L::NB; ++j)
#pragma unroll
      for (int r = 0; r < L::NR; ++r) acc[i][j][r] = 0.f;
  // first tile: A(1) -- the four youngest pieces -- may stay in flight; later tiles: the stages have had the whole
  // epilogue to land, and the epilogue's C stores share the counter: everything
  step_barrier(!primed && nt > 1);
  if (PERSIST) {
    asm volatile("" : "+v"(ln));
    if (BUF) {
      piece_lane_offsets<ALAY>(p.lda, ln, voA);
      piece_lane_offsets<BLAY, PERMB>(p.ldb, ln, voB);
    }
  }
  int ia = 0;                                   // A slot of the current step (t mod 3)

  if constexpr (MF == 32) {
    const int l31 = ln & 31, half = ln >> 5;
    int offA[4], offB[2], kofs[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) offA[i] = dma_frag_offset<ALAY>(wm * 128 + i * 32, ln);
#pragma unroll
    for (int j = 0; j < 2; ++j) offB[j] = dma_frag_offset<BLAY>(wn * 64 + j * 32, ln);
    {
      const int swz = (l31 >> 1) & 7;
#pragma unroll
      for (int s = 0; s < 4; ++s) kofs[s] = ((2 * s + half) ^ swz) * 8;
    }
    // the fine interleave (one DMA piece in front of every 4 MFMAs) is the order of the KC x KC instantiations;
    // the RC x RC (wgrad) instantiation measured 0-8 % slower with it (transpose reads: two ds_read_b64_tr_b16
    // per fragment) and issues the whole next stage at the top of the step
    constexpr bool kFine = ALAY == KC;
    for (int t = 0; t < nt; ++t) {
      const bf16_t* sA = slotA(ia);
      const bf16_t* sB = slotB(t & 1);
      bf16_t* nA = slotA(ia == 0 ? 2 : ia - 1);            // stage t + 2
      bf16_t* nB = slotB((t + 1) & 1);                     // stage t + 1
      const int k0 = kbeg + (t + 1) * BK, k0A = k0 + BK;
      const bool more = t + 1 < nt, moreA = t + 2 < nt;
      if (!kFine && more) {
#pragma unroll
        for (int j = 0; j < 4; ++j) piece(1, k0, nB, j);
      }
      // software-pipelined fragment reads: the 6 ds_reads of k-step ks+1 are issued before the 8
      // MFMAs of k-step ks, so only the first read group of a stage exposes LDS latency
      bf16x8 af[2][4], bfr[2][2];
#pragma unroll
      for (int i = 0; i < 4; ++i) af[0][i] = dma_load_frag<ALAY>(sA, offA[i], 0, kofs);
#pragma unroll
      for (int j = 0; j < 2; ++j) bfr[0][j] = dma_load_frag<BLAY>(sB, offB[j], 0, kofs);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
#pragma unroll
          for (int i = 0; i < 4; ++i) af[nxt][i] = dma_load_frag<ALAY>(sA, offA[i], ks + 1, kofs);
#pragma unroll
          for (int j = 0; j < 2; ++j) bfr[nxt][j] = dma_load_frag<BLAY>(sB, offB[j], ks + 1, kofs);
        }
        // pin: next step's reads stay ABOVE this step's MFMAs
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (kFine) {
            const int pc = 2 * ks + g;                     // B first: the four youngest pieces are A's
            if (pc < 4) { if (more) piece(1, k0, nB, pc); }
            else if (moreA) piece(0, k0A, nA, pc - 4);
          } else if (moreA && ks >= 2) {
            piece(0, k0A, nA, 2 * (ks - 2) + g);           // the far stage: requested in the second half of the step
          }
#pragma unroll
          for (int i = 2 * g; i < 2 * g + 2; ++i)
#pragma unroll
            for (int j = 0; j < 2; ++j)
              acc[i][j] = mfma_32x32x16<SPLIT>(af[cur][i], bfr[cur][j], acc[i][j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      step_barrier(moreA);
      ia = ia == 2 ? 0 : ia + 1;
    }
  } else {
    // ---- 16x16x32 fragments: lane holds 8 consecutive k of row (lane & 15); k-granule 4*s2 + (lane >> 4)
    const int l15 = ln & 15, q = ln >> 4;
    int offA[8], offB[4], kofs[2];
#pragma unroll
    for (int i = 0; i < 8; ++i) offA[i] = (wm * 128 + i * 16 + l15) * 64;
#pragma unroll
    for (int j = 0; j < 4; ++j) offB[j] = (wn * 64 + j * 16 + l15) * 64;
#pragma unroll
    for (int s2 = 0; s2 < 2; ++s2) kofs[s2] = ((4 * s2 + q) ^ (l15 >> 1)) * 8;
    for (int t = 0; t < nt; ++t) {
      const bf16_t* sA = slotA(ia);
      const bf16_t* sB = slotB(t & 1);
      bf16_t* nA = slotA(ia == 0 ? 2 : ia - 1);            // stage t + 2
      bf16_t* nB = slotB((t + 1) & 1);                     // stage t + 1
      const int k0 = kbeg + (t + 1) * BK, k0A = k0 + BK;
      const bool more = t + 1 < nt, moreA = t + 2 < nt;
      // a step = 2 k-halves (32 deep) x 2 row-halves (64 rows): 4 blocks of 16 MFMAs; fragment reads of the next
      // block are issued before the MFMAs of the current one; two DMA pieces per block, one per 8 MFMAs
      bf16x8 af[2][4], bfr[2][4];
#pragma unroll
      for (int j = 0; j < 4; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(sB + offB[j] + kofs[0]);
#pragma unroll
      for (int i = 0; i < 4; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(sA + offA[i] + kofs[0]);
#pragma unroll
      for (int blk = 0; blk < 4; ++blk) {
        const int s2 = blk >> 1, h = blk & 1, cur = blk & 1, nxt = cur ^ 1;
        if (blk < 3) {
          const int ns2 = (blk + 1) >> 1, nh = (blk + 1) & 1;
          if (nh == 0) {
#pragma unroll
            for (int j = 0; j < 4; ++j) bfr[ns2 & 1][j] = *reinterpret_cast<const bf16x8*>(sB + offB[j] + kofs[ns2]);
          }
#pragma unroll
          for (int i = 0; i < 4; ++i) af[nxt][i] = *reinterpret_cast<const bf16x8*>(sA + offA[4 * nh + i] + kofs[ns2]);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          {
            const int pc = 2 * blk + g;                    // B first: the four youngest pieces are A's
            if (pc < 4) { if (more) piece(1, k0, nB, pc); }
            else if (moreA) piece(0, k0A, nA, pc - 4);
          }
#pragma unroll
          for (int i = 2 * g; i < 2 * g + 2; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
              acc[4 * h + i][j] = mfma_16x16x32<SPLIT>(af[cur][i], bfr[s2 & 1][j], acc[4 * h + i][j]);
          __builtin_amdgcn_sched_barrier(0);
        }
      }
      step_barrier(moreA);
      ia = ia == 2 ? 0 : ia + 1;
    }
  }
  // the next tile of this workgroup: its first stages go out now and land beside the epilogue
  int vb2 = vb + (int)gridDim.x;
  if (sched != nullptr) {
    if (tid == 0) *ticket_lds = ticket;
    lds_barrier();
    vb2 = (vb & 7) + 8 * ((int)(gridDim.x >> 3) + __builtin_amdgcn_readfirstlane(*ticket_lds));
  }
  const bool has_next = PERSIST && !p.split_fast && vb2 < nbm * nbn;
  int tm2 = 0, tn2 = 0;
  if (has_next) {
    xcd_tile_coords(nbm, nbn, vb2, tm2, tn2);
    first_stages(tm2, tn2);
  }
  if constexpr (SPLIT) {
    // the operand images hold value * 2^k: back to the value's scale (exact power of two)
    const float os = p.out_scale;
#pragma unroll
    for (int i = 0; i < L::MB; ++i)
#pragma unroll
      for (int j = 0; j < L::NB; ++j)
#pragma unroll
        for (int r = 0; r < L::NR; ++r) acc[i][j][r] *= os;
  }
  // the epilogue's per-lane address arithmetic must not be hoisted out of the tile loop (it would sit in ~40
  // registers through the K loop): it is derived from an opaque copy of the thread id
  int te = tid;
  if (PERSIST) asm volatile("" : "+v"(te));
  if constexpr (EPI == EPI_DGRAD_BN || EPI == EPI_DGRAD_BN_POINTS) {
    epilogue_dgrad_bn<MF, EPI == EPI_DGRAD_BN_POINTS, TC>(p, acc, epi, tm, tn, te);
  } else if constexpr (EPI == EPI_AFFINE) {
    epilogue_full_tile<TC, MF, true>(p, acc, epi, tm, tn, te, split);
  } else if constexpr (EPI == EPI_POOL1 || EPI == EPI_POOL2 || EPI == EPI_POOL4) {
    epilogue_affine_meanpool<MF, EPI == EPI_POOL1 ? 1 : (EPI == EPI_POOL2 ? 2 : 4)>(p, acc, tm, tn, te);
  } else {
    epilogue_full_tile<TC, MF, false>(p, acc, epi, tm, tn, te, split);
    if (p.colstats != nullptr) epilogue_colstats<MF>(p, acc, reinterpret_cast<unsigned char*>(epi), tm, tn, te);
  }
  if (!has_next) {
    if (sched != nullptr && tid == 0 && atomicAdd(&sched[8], 1) == (int)gridDim.x - 1) {
#pragma unroll
      for (int i = 0; i < 9; ++i) atomicExch(&sched[i], 0);
    }
    break;
  }
  vb = vb2;
  tm = tm2;
  tn = tn2;
  primed = true;
  }
  // this workgroup's tiles are done: if the launch carries the BatchNorm finalize of its statistics, the last
  // workgroup to get here runs it (bn_tail.h)
  if constexpr (PERSIST && (EPI == EPI_PLAIN || EPI == EPI_DGRAD_BN || EPI == EPI_DGRAD_BN_POINTS)) {
    // read from the kernel-argument segment HERE, through a pointer the compiler cannot see through: referenced as
    // p.tail its 20 fields are fetched at kernel entry and sit in (spilled) SGPRs through the whole tile loop
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const BnTail tail = *reinterpret_cast<const BnTail*>(ka + offsetof(GemmParams, tail));
    bn_tail_run(tail, tid, NTHREADS, gridDim.x, ticket_lds + 1);
  }
}

// MFMA shape per instantiation.  KC x KC (forward / dgrad): v_mfma_f32_16x16x32_bf16 -- same cycles per FLOP as
// 32x32x16, but the chip holds a higher clock on it (MI355X_MICROARCH.md, DVFS item 7); same-box A/B on the three
// PointNet shapes (median of 5 interleaved rounds): forward +9.2 / -0.7 / +2.6 %, fused dgrad +4.4 / +4.0 / +1.9 %.
// RC x RC (wgrad) keeps 32x32x16: its fragments come from ds_read_b64_tr_b16 pairs laid out for that shape.
template <typename TC, int ALAY, int BLAY, int EPI, bool BUF, bool SPLIT = false>
bool launch_dma_inst(const GemmParams& p, dim3 grid, hipStream_t s) {
  constexpr int MF = ALAY == KC ? 16 : 32;
  static bool configured = false;
  auto kern = gemm_bf16_dma_kernel<TC, ALAY, BLAY, EPI, MF, BUF, SPLIT>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            D_LDS_BYTES) != hipSuccess)
      return false;
    configured = true;
  }
  const PcaaLaunchEvents ev = pcaa_take_launch_events();
  if (ev.start != nullptr)
    hipExtLaunchKernelGGL(kern, grid, dim3(NTHREADS), D_LDS_BYTES, s, ev.start, ev.stop, 0, p);
  else
    hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), D_LDS_BYTES, s, p);
  return true;
}

// The 4-wave tile loop (gemm_v2.h) serves every KC x KC launch without K splits whose contraction is at least five
// 64-deep steps long (its ticket hand-off needs four of them); pcaa_gemm_v2_enable(0) routes them back to the 8-wave loop (A/B).
static int g_v2_enabled = -1;
bool pcaa_gemm_v2_is_enabled();
static bool v2_enabled() {
  if (g_v2_enabled < 0) {
    const char* e = getenv("PCAA_GEMM_V2");
    g_v2_enabled = (e != nullptr && e[0] == '0') ? 0 : 1;
  }
  return g_v2_enabled != 0;
}
template <typename TC, int EPI, bool SPLIT, bool RAG>
bool launch_v2_rag(const GemmParams& p, dim3 grid, hipStream_t s) {
  static bool configured = false;
  auto kern = v2::gemm_bf16_v2_kernel<TC, EPI, SPLIT, RAG>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            v2::LDS_BYTES) != hipSuccess)
      return false;
    configured = true;
  }
  const PcaaLaunchEvents ev = pcaa_take_launch_events();
  if (ev.start != nullptr)
    hipExtLaunchKernelGGL(kern, grid, dim3(v2::NT), v2::LDS_BYTES, s, ev.start, ev.stop, 0, p);
  else
    hipLaunchKernelGGL(kern, grid, dim3(v2::NT), v2::LDS_BYTES, s, p);
  return true;
}

template <bool SPLIT>
bool launch_v2rc(const GemmParams& p, dim3 grid, hipStream_t s) {
  static bool configured = false;
  auto kern = v2::gemm_bf16_v2rc_kernel<SPLIT>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            v2::LDS_BYTES) != hipSuccess)
      return false;
    configured = true;
  }
  const PcaaLaunchEvents ev = pcaa_take_launch_events();
  if (ev.start != nullptr)
    hipExtLaunchKernelGGL(kern, grid, dim3(v2::NT), v2::LDS_BYTES, s, ev.start, ev.stop, 0, p);
  else
    hipLaunchKernelGGL(kern, grid, dim3(v2::NT), v2::LDS_BYTES, s, p);
  return true;
}

template <typename TC, int EPI, bool SPLIT>
bool launch_v2_inst(const GemmParams& p, dim3 grid, hipStream_t s) {
  // a partial last row tile stores (and, fused dgrad, reads y) through 32-bit buffer offsets over the whole matrix; the
  // epilogues form offsets for every row of the last tile, also those past M, which the buffer's range check only
  // drops as long as they do not wrap: the bound is on the PADDED row count
  if ((p.M % BM) != 0 && (long)cdiv(p.M, BM) * BM * p.ldc * (long)sizeof(TC) >= (1L << 32)) return false;
  return (p.M % BM) != 0 ? launch_v2_rag<TC, EPI, SPLIT, true>(p, grid, s) : launch_v2_rag<TC, EPI, SPLIT, false>(p, grid, s);
}

// Launches without K splits start one workgroup per CU (a multiple of 8: the XCD-aware tile order) and let them draw
// their tiles: no dispatch gap between a CU's tiles, and the next tile's first stages are requested before the
// epilogue (see the kernel).  The draw is dynamic (ticket counters, sched_slot): with FIXED shares a launch that does
// not get every CU at once -- 8 CUs held by another stream's kernel for its duration, tools/gemm_contention.py -- took
// 0.83 ms instead of 0.49 (the late workgroups run their whole share afterwards); with tickets 0.50-0.61 ms for 8-64
// CUs held, i.e. the ideal 256 / (256 - H).  Price: one LDS hand-off + barrier per tile, 1-5 % on an idle chip.
static std::mutex g_sched_mutex;

static unsigned persistent_grid(long ntiles) {
  // per device (round-3 advisor finding: the count was cached for whichever device came first)
  static int ncu[64] = {0};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 64) dev = 0;
  std::lock_guard<std::mutex> lock(g_sched_mutex);
  if (ncu[dev] == 0) {
    int n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) n = 256;
    ncu[dev] = n >= 8 ? n / 8 * 8 : 8;
  }
  return (unsigned)(ntiles < ncu[dev] ? ntiles : ncu[dev]);
}

// Ticket counters of the tile loop: 16 ints per (device, stream) that launches these kernels (launches of one stream
// run in order, and the last workgroup of a launch leaves its counters at zero).  Allocated per device on first use,
// never during a stream capture (a captured first launch walks fixed shares instead).  128 slots per device: torch
// hands out its side streams from a pool of 32 + 32, so a process that builds trainer after trainer (bench.py's legs)
// keeps meeting new stream handles -- with 16 slots the fifth trainer's GEMMs silently fell back to fixed shares and
// ran 5 % slower beside the side-stream Adam (round 3).  Guarded by a mutex: the autograd engine's thread launches too.
static int* sched_slot(hipStream_t s) {
  constexpr int SLOTS = 128, DEVS = 64;
  struct Table { int* base; hipStream_t owner[SLOTS]; int used; bool failed; };
  static Table tables[DEVS] = {};
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= DEVS) return nullptr;
  std::lock_guard<std::mutex> lock(g_sched_mutex);
  Table& t = tables[dev];
  if (t.failed) return nullptr;
  if (t.base == nullptr) {
    hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &st) != hipSuccess || st != hipStreamCaptureStatusNone) return nullptr;
    if (hipMalloc(reinterpret_cast<void**>(&t.base), SLOTS * 16 * sizeof(int)) != hipSuccess ||
        hipMemset(t.base, 0, SLOTS * 16 * sizeof(int)) != hipSuccess) {
      t.base = nullptr;
      t.failed = true;
      return nullptr;
    }
  }
  for (int i = 0; i < t.used; ++i)
    if (t.owner[i] == s) return t.base + 16 * i;
  if (t.used == SLOTS) return nullptr;
  t.owner[t.used] = s;
  return t.base + 16 * t.used++;
}

template <typename TC, int ALAY, int BLAY, int EPI>
bool launch_dma(const GemmParams& p_in, dim3 grid_in, hipStream_t s) {
  dim3 grid = grid_in;
  GemmParams p = p_in;
  p.sched = nullptr;
  p.tail = BnTail{};
  if (ALAY == KC && !p.split_fast && grid.z == 1) {
    const unsigned g = persistent_grid((long)grid.x);
    if (g < grid.x) p.sched = sched_slot(s);
    grid.x = g;
    // every workgroup of such a launch owns at least one tile and adds to colstats: the launch can carry the finalize
    if ((EPI == EPI_PLAIN || EPI == EPI_DGRAD_BN || EPI == EPI_DGRAD_BN_POINTS) && p.colstats != nullptr && grid.y == 1)
      p.tail = pcaa_take_bn_tail(p.colstats);
  }
  // buffer addressing needs each operand below 4 GiB (32-bit offsets); the flat form serves anything larger.  Same-box
  // A/B (profiles/r02_gemm_lab2.txt): forward / fused dgrad +1..3 %, wgrad (whole stage issued at the top of the step:
  // 8 pieces x ~7 VALU each in front of the first MFMA) +2.6 / +7.3 / +10.1 % on the three PointNet shapes.
  const int krows = p.seg_len > 0 ? p.seg_len : p.K;
  const long a_bytes = (long)(ALAY == KC ? p.M : krows) * p.lda * 2, b_bytes = (long)(BLAY == KC ? p.N : krows) * p.ldb * 2;
  const bool buf = a_bytes < (1L << 32) && b_bytes < (1L << 32);
  // (advisor, round 3) every path that does not launch hands the tail back: resolve() then runs the stand-alone finalize
  bool ok = false;
  bool decided = false;
  if constexpr (ALAY == KC && BLAY == KC && EPI != EPI_DGRAD_BN_POINTS) {
    // PCAA_GEMM_TICKETS=0: fixed tile shares in the 4-wave loop (A/B of the ticket draw's once-per-tile drain)
    static const bool tickets = [] { const char* e = getenv("PCAA_GEMM_TICKETS"); return !(e != nullptr && e[0] == '0'); }();
    if (!tickets) p.sched = nullptr;
    const int steps = (p.seg_len > 0 ? 3 * p.seg_len : p.K) / BK;
    constexpr bool kSplitOk = sizeof(TC) == 4 && (EPI == EPI_PLAIN || EPI == EPI_DGRAD_BN);
    constexpr bool kPlainOk = !(EPI == EPI_DGRAD_BN && sizeof(TC) == 4);      // fp32 dz exists for the split operands only
    if (v2_enabled() && grid.z == 1 && grid.y == 1 && !p.split_fast && !p.atomic && p.nsplit <= 1 && p.c_split_stride == 0 &&
        steps >= 5 && (p.seg_len > 0 ? p.k_per_split == p.K : (p.k_per_split >= p.K)) && (p.ldc % 8) == 0 &&
        ((uintptr_t)p.C % 16) == 0) {
      if (p.seg_len > 0) {
        if constexpr (kSplitOk) {
          ok = launch_v2_inst<TC, EPI, true>(p, grid, s);
          decided = true;
        }
      } else {
        if constexpr (kPlainOk) {
          ok = launch_v2_inst<TC, EPI, false>(p, grid, s);
          decided = true;
        }
      }
    }
  }
  if constexpr (ALAY == RC && BLAY == RC && EPI == EPI_PLAIN && sizeof(TC) == 4) {
    // the weight gradients (slab split-K or a single K pass, no atomics): the 4-wave loop on the RC images
    static const bool rc_on = [] { const char* e = getenv("PCAA_GEMM_V2_RC"); return !(e != nullptr && e[0] == '0'); }();
    if (v2_enabled() && rc_on && !p.atomic && (p.M % BM) == 0 && (p.N % BN) == 0 && (p.k_per_split % BK) == 0 && buf &&
        (p.ldc % 1) == 0 && p.colstats == nullptr && p.bias == nullptr) {
      ok = p.seg_len > 0 ? launch_v2rc<true>(p, grid, s) : launch_v2rc<false>(p, grid, s);
      decided = true;
    }
  }
  if (!decided && (p.M % BM) != 0) decided = true;      // a partial last row tile: the 4-wave loop only (ok stays false)
  if (!decided) {
  if constexpr ((EPI == EPI_PLAIN || (EPI == EPI_DGRAD_BN && ALAY == KC)) && sizeof(TC) == 4) {
    // split-fp16 operands (pcaa_gemm_split3, pcaa_gemm_dgrad_bn_split3): fp32 result only
    if (p.seg_len > 0) {
      ok = buf ? launch_dma_inst<TC, ALAY, BLAY, EPI, true, true>(p, grid, s)
               : launch_dma_inst<TC, ALAY, BLAY, EPI, false, true>(p, grid, s);
      decided = true;
    } else if (EPI == EPI_DGRAD_BN) {
      decided = true;                                // fp32 dz exists for the split operands only
    }
  }
  if (!decided) {
    if (p.seg_len > 0) ok = false;
    else ok = buf ? launch_dma_inst<TC, ALAY, BLAY, EPI, true>(p, grid, s)
                  : launch_dma_inst<TC, ALAY, BLAY, EPI, false>(p, grid, s);
  }
  }
  if (!ok) pcaa_rearm_bn_tail(p.tail);
  return ok;
}

template <typename TA, typename TB, typename TC, int ALAY, int BLAY>
bool launch(const GemmParams& p, dim3 grid, hipStream_t s) {
  static bool configured = false;
  auto kern = gemm_bf16_big_kernel<TA, TB, TC, ALAY, BLAY>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            LDS_BYTES) != hipSuccess)
      return false;
    configured = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), LDS_BYTES, s, p);
  return true;
}

}  // namespace

// dgrad + BatchNorm/ELU backward of the layer below (pcaa_gemm_dgrad_bn): whole 256x256 tiles, bf16 KC x KC
bool pcaa_launch_gemm_dgrad_bn(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  if ((p.N % BN) || (p.K % BK)) return false;      // (M: a partial last row tile is the 4-wave loop's; launch_dma refuses it otherwise)
  p.nsplit = 1;
  p.split_fast = 0;
  p.k_per_split = p.K;
  p.atomic = 0;
  p.c_split_stride = 0;
  const long ntiles = cdiv(p.M, BM) * (p.N / BN);
  if (ntiles >= (1L << 31)) return false;
  if (p.seg_len > 0) {
    // split-fp16 operands: K is the contraction length of ONE pass (the kernel walks 3 K); y and dz fp32
    if (p.ep_y == nullptr) return false;
    p.k_per_split = p.K;
    return launch_dma<float, KC, KC, EPI_DGRAD_BN>(p, dim3((unsigned)ntiles, 1, 1), stream);
  }
  if (p.ep_y == nullptr) return launch_dma<bf16_t, KC, KC, EPI_DGRAD_BN_POINTS>(p, dim3((unsigned)ntiles, 1, 1), stream);
  return launch_dma<bf16_t, KC, KC, EPI_DGRAD_BN>(p, dim3((unsigned)ntiles, 1, 1), stream);
}

// product + eval-mode BatchNorm + ELU (pcaa_gemm_affine_elu): whole 256x256 tiles, bf16 KC x KC -> bf16
bool pcaa_launch_gemm_affine_elu(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  if ((p.N % BN) || (p.K % BK)) return false;
  p.nsplit = 1;
  p.split_fast = 0;
  p.k_per_split = p.K;
  p.atomic = 0;
  p.c_split_stride = 0;
  p.colstats = nullptr;
  const long ntiles = cdiv(p.M, BM) * (p.N / BN);
  if (ntiles >= (1L << 31)) return false;
  // p.ep_xc: rows per mean-pool group (0: plain activation output)
  switch (p.ep_xc) {
    case 0: return launch_dma<bf16_t, KC, KC, EPI_AFFINE>(p, dim3((unsigned)ntiles, 1, 1), stream);
    case 32: return launch_dma<float, KC, KC, EPI_POOL1>(p, dim3((unsigned)ntiles, 1, 1), stream);
    case 64: return launch_dma<float, KC, KC, EPI_POOL2>(p, dim3((unsigned)ntiles, 1, 1), stream);
    case 128: return launch_dma<float, KC, KC, EPI_POOL4>(p, dim3((unsigned)ntiles, 1, 1), stream);
    default: return false;
  }
}

bool pcaa_launch_gemm_bf16_big(const GemmParams& p_in, int a_dtype, int a_layout, int b_dtype, int b_layout,
                               int c_dtype, int nsplit, hipStream_t stream) {
  GemmParams p = p_in;
  if (p.N < 128) return false;
  if ((p.lda % 8) || (p.ldb % 8) || ((uintptr_t)p.A % 16) || ((uintptr_t)p.B % 16)) return false;
  // a KC operand needs whole 8-chunks along K, an RC operand whole 8-chunks along its rows
  if ((a_layout == KC || b_layout == KC) && (p.K % 8)) return false;
  if (a_layout == RC && (p.M % 8)) return false;
  if (b_layout == RC && (p.N % 8)) return false;
  const long ntiles = cdiv(p.M, BM) * cdiv(p.N, BN);
  if (ntiles * nsplit >= (1L << 31)) return false;
  p.nsplit = nsplit;
  // split_fast: all tiles of one K-range on one XCD, so the operand rows of that range cross the
  // fabric once instead of once per XCD that holds a tile.  Slab split-K (the wgrad products), >= 256
  // workgroups: fabric fetch per launch 3022 -> 1009 MB (= the algorithmic 1006 MB) on
  // dW[1024,1024], time -5..-8 % on the three wgrad shapes.  With the atomic
  // epilogue at 128 workgroups it had measured slower (0.53 -> 1.36 ms), so it stays off there.
  p.split_fast = (p.c_split_stride != 0 && nsplit >= 8 && nsplit % 8 == 0 && ntiles * nsplit >= 256) ? 1 : 0;
  dim3 grid((unsigned)ntiles, 1, (unsigned)nsplit);
  if (p.split_fast) grid = dim3((unsigned)(ntiles * nsplit), 1, 1);
  const bool af = a_dtype == PCAA_F32, bf = b_dtype == PCAA_F32, cf = c_dtype == PCAA_F32;
  // LDS-DMA kernel: bf16 x bf16, whole tiles only
  // (a partial last row tile: KC x KC without K splits through the 4-wave loop; launch_dma returns false otherwise and
  // the register-staged kernels below take the shape as before)
  const bool m_ok = (p.M % BM) == 0 || (a_layout == KC && nsplit == 1 && v2_enabled() && p.K / BK >= 5 && !p.atomic &&
                                        p.c_split_stride == 0);
  if (!af && !bf && m_ok && (p.N % BN) == 0 && (p.K % BK) == 0 && (p.k_per_split % BK) == 0 &&
      a_layout == b_layout) {
    if (a_layout == KC) {
      const bool ok = cf ? launch_dma<float, KC, KC, EPI_PLAIN>(p, grid, stream) : launch_dma<bf16_t, KC, KC, EPI_PLAIN>(p, grid, stream);
      // a partial last row tile the 4-wave loop declined (ldc, alignment, size): the register-staged kernel below
      if (ok || (p.M % BM) == 0) return ok;
    } else if (cf) {
      return launch_dma<float, RC, RC, EPI_PLAIN>(p, grid, stream);
    }
  }
  if (p.seg_len > 0) return false;       // split-fp16 operands are served by the LDS-DMA kernel only
  if (a_layout == KC && b_layout == KC) {
    // bf16 activations x fp32/bf16 weights (PointNet forward / dgrad), fp32 x fp32 (decoder forward)
    if (!af && bf && !cf) return launch<bf16_t, float, bf16_t, KC, KC>(p, grid, stream);
    if (!af && bf && cf) return launch<bf16_t, float, float, KC, KC>(p, grid, stream);
    if (!af && !bf && !cf) return launch<bf16_t, bf16_t, bf16_t, KC, KC>(p, grid, stream);
    if (!af && !bf && cf) return launch<bf16_t, bf16_t, float, KC, KC>(p, grid, stream);
    if (af && bf && cf) return launch<float, float, float, KC, KC>(p, grid, stream);
    return false;
  }
  if (a_layout == RC && b_layout == RC) {
    // wgrad: contraction over rows
    if (!af && !bf && cf) return launch<bf16_t, bf16_t, float, RC, RC>(p, grid, stream);
    if (af && bf && cf) return launch<float, float, float, RC, RC>(p, grid, stream);
    return false;
  }
  if (a_layout == KC && b_layout == RC) {
    // decoder dgrad: dX = dY . W with W stored [out, in]
    if (af && bf && cf) return launch<float, float, float, KC, RC>(p, grid, stream);
    return false;
  }
  return false;
}

bool pcaa_gemm_v2_is_enabled() { return v2_enabled(); }
extern "C" int pcaa_gemm_v2_enable(int on) {
  g_v2_enabled = on ? 1 : 0;
  return PCAA_OK;
}

// ------------------------------------------------------------------ kernel-exact launch timing
namespace {
thread_local PcaaLaunchEvents g_armed = {nullptr, nullptr};
}
PcaaLaunchEvents pcaa_take_launch_events() {
  const PcaaLaunchEvents e = g_armed;
  g_armed = {nullptr, nullptr};
  return e;
}
extern "C" int pcaa_timing_events_create(void** start, void** stop) {
  PCAA_CHECK_ARG(start && stop, "pcaa_timing_events_create: null");
  hipEvent_t a = nullptr, b = nullptr;
  if (hipEventCreate(&a) != hipSuccess || hipEventCreate(&b) != hipSuccess) {
    pcaa_set_error("pcaa_timing_events_create: hipEventCreate failed");
    return PCAA_ERR_LAUNCH;
  }
  *start = a;
  *stop = b;
  return PCAA_OK;
}
extern "C" int pcaa_timing_events_destroy(void* start, void* stop) {
  if (start) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(start));
  if (stop) (void)hipEventDestroy(reinterpret_cast<hipEvent_t>(stop));
  return PCAA_OK;
}
extern "C" int pcaa_time_next_gemm(void* start, void* stop) {
  PCAA_CHECK_ARG((start == nullptr) == (stop == nullptr), "pcaa_time_next_gemm: both events or none");
  g_armed = {reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop)};
  return PCAA_OK;
}
extern "C" int pcaa_timing_pending(void) { return g_armed.start != nullptr ? 1 : 0; }
extern "C" int pcaa_timing_elapsed_ms(void* start, void* stop, float* ms) {
  PCAA_CHECK_ARG(start && stop && ms, "pcaa_timing_elapsed_ms: null");
  const hipError_t e = hipEventElapsedTime(ms, reinterpret_cast<hipEvent_t>(start), reinterpret_cast<hipEvent_t>(stop));
  if (e != hipSuccess) {
    pcaa_set_error("pcaa_timing_elapsed_ms: %s", hipGetErrorString(e));
    return PCAA_ERR_LAUNCH;
  }
  return PCAA_OK;
}
