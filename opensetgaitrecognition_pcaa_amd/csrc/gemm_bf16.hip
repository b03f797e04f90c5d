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
  TC* C = reinterpret_cast<TC*>(p.C);
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
// LDS-DMA variant: both operands already bf16 in HBM, shapes whole tiles
// (M % 256 == N % 256 == 0, K-range % 64 == 0).  Tiles go HBM -> LDS with
// global_load_lds_dwordx4 (no VGPR staging, no ds_write pass); the LDS images are
// unpadded, so the bank-conflict fix is an XOR swizzle applied to the per-lane
// SOURCE address (the DMA writes LDS linearly: base + lane*16) and again on the
// fragment reads:
//   KC image [256 rows][64 k] (128-B rows): 16-B granule g of row r holds global k-granule
//     g ^ ((r>>1)&7)  -> ds_read_b128 of 16 consecutive rows hits 16 distinct 16-B slots.
//   RC image [64 k][256 rows] (512-B rows): 16-B granule c of k-row k holds global row-granule
//     c ^ 4*(k&3)     -> the 4 k-rows of a transpose read land on disjoint bank quarters.
// One barrier per 64-deep step: the DMA of step t+1 is issued before the MFMAs of
// step t and drained (vmcnt(0), emitted by __syncthreads) at the barrier.
// ===========================================================================
constexpr int D_TILE = 256 * 64;                       // elements per operand per stage (32 KB)
constexpr int D_LDS_BYTES = 2 * 2 * D_TILE * 2;        // 131072

template <int LAY>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ base, long ld, int row0, int R, int k0,
                                         bf16_t* s_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = wave * 4 + j;       // 1-KB piece index, 32 per tile
    const bf16_t* src;
    if (LAY == KC) {
      const int r = 8 * p + (lane >> 3);
      const int g = (lane & 7) ^ ((r >> 1) & 7);
      src = base + (long)min(row0 + r, R - 1) * ld + k0 + 8 * g;
    } else {
      const int k = 2 * p + (lane >> 5);
      const int c = (lane & 31) ^ (4 * (k & 3));
      src = base + (long)(k0 + k) * ld + row0 + 8 * c;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(s_tile + p * 512), 16, 0, 0);
  }
}

// per-lane element offset of fragment rows [row_base, row_base+32) at k-step 0 (row_base % 32 == 0)
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

template <typename TC, int ALAY, int BLAY>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_dma_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  int tm, tn;
  const int split = block_coords(p, p.M / BM, p.N / BN, tm, tn);

  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg) / BK;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int offA[FM], offB[FN], kofs[4];
#pragma unroll
  for (int i = 0; i < FM; ++i) offA[i] = dma_frag_offset<ALAY>(wm * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < FN; ++j) offB[j] = dma_frag_offset<BLAY>(wn * 64 + j * 32, lane);
  {
    const int swz = (l31 >> 1) & 7;
#pragma unroll
    for (int s = 0; s < 4; ++s) kofs[s] = ((2 * s + half) ^ swz) * 8;
  }

  if (nt > 0) {
    dma_tile<ALAY>(A, p.lda, tm * BM, p.M, kbeg, smem, wave, lane);
    dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, kbeg, smem + D_TILE, wave, lane);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const bf16_t* sA = smem + (t & 1) * 2 * D_TILE;
    const bf16_t* sB = sA + D_TILE;
    if (t + 1 < nt) {
      bf16_t* nA = smem + ((t + 1) & 1) * 2 * D_TILE;
      const int k0 = kbeg + (t + 1) * BK;
      dma_tile<ALAY>(A, p.lda, tm * BM, p.M, k0, nA, wave, lane);
      dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, k0, nA + D_TILE, wave, lane);
    }
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      bf16x8 af[FM], bfr[FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[i] = dma_load_frag<ALAY>(sA, offA[i], ks, kofs);
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[j] = dma_load_frag<BLAY>(sB, offB[j], ks, kofs);
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
    __syncthreads();
  }

  epilogue_store<TC, 64>(p, acc, smem, tm, tn, tid, split);
  if (p.colstats != nullptr) {
    float* red = reinterpret_cast<float*>(smem_raw);
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
    const double v = (double)red[(stat * 2 + 0) * 256 + col] + (double)red[(stat * 2 + 1) * 256 + col];
    unsafeAtomicAdd(&p.colstats[((long)(tm % p.nrep) * 2 + stat) * p.N + tn * BN + col], v);
  }
}

template <typename TC, int ALAY, int BLAY>
bool launch_dma(const GemmParams& p, dim3 grid, hipStream_t s) {
  static bool configured = false;
  auto kern = gemm_bf16_dma_kernel<TC, ALAY, BLAY>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            D_LDS_BYTES) != hipSuccess)
      return false;
    configured = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), D_LDS_BYTES, s, p);
  return true;
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
  // split_fast (all tiles of one K-range on one XCD) measured neutral-to-negative on MI355X
  // (wgrad 1024x512 K=245760: 0.53 -> 1.36 ms at 128 blocks; 1024^2: 0.61 -> 0.59): the re-reads
  // were already served by the Infinity Cache, and co-locating them adds same-line contention.
  p.split_fast = 0;
  dim3 grid((unsigned)ntiles, 1, (unsigned)nsplit);
  if (p.split_fast) grid = dim3((unsigned)(ntiles * nsplit), 1, 1);
  const bool af = a_dtype == PCAA_F32, bf = b_dtype == PCAA_F32, cf = c_dtype == PCAA_F32;
  // LDS-DMA kernel: bf16 x bf16, whole tiles only
  if (!af && !bf && (p.M % BM) == 0 && (p.N % BN) == 0 && (p.K % BK) == 0 && (p.k_per_split % BK) == 0 &&
      a_layout == b_layout) {
    if (a_layout == KC) return cf ? launch_dma<float, KC, KC>(p, grid, stream) : launch_dma<bf16_t, KC, KC>(p, grid, stream);
    if (cf) return launch_dma<float, RC, RC>(p, grid, stream);
  }
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
