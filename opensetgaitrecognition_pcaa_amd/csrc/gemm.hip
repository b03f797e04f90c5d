// MFMA GEMM for the PCAA path (gfx950).
//
//   C[M,N] (=|+=) A(M,K) . B(K,N) (+bias) , optional BatchNorm column statistics.
//
// One 128x128 output tile per 256-thread workgroup (4 waves as 2x2, each wave a
// 64x64 sub-tile = 2x2 MFMA 32x32 accumulators).  Operand tiles are staged
// global -> registers -> LDS with the next tile's global loads in flight while
// the current tile is multiplied.  Two math flavours:
//
//   F32 : v_mfma_f32_32x32x2_f32 -- exact fp32 (bit-for-bit an fmaf chain).
//         LDS holds [k][row] fp32 tiles; either operand may be stored
//         contraction-contiguous (KC) or row-contiguous (RC) in HBM, fp32 or
//         bf16.  Used for the fp32 parity mode and every small GEMM.
//   BF16: v_mfma_f32_32x32x16_bf16, fp32 accumulate.  LDS holds [row][k] bf16
//         tiles (144-B pitch: conflict-free ds_read_b128); KC operands only.
//
// Workgroup -> tile mapping is XCD-aware: the 8 XCDs each walk a contiguous
// range of tiles with the N-tiles of one M-panel adjacent, so the A panel is
// re-read from that XCD's L2 and not from HBM.
#include "gemm_common.h"

namespace {

constexpr int BM = 128, BN = 128;
constexpr int KC = PCAA_LAYOUT_KC, RC = PCAA_LAYOUT_RC;

__device__ __forceinline__ void tile_coords(int M, int N, int& tm, int& tn) {
  xcd_tile_coords((M + BM - 1) / BM, (N + BN - 1) / BN, blockIdx.x, tm, tn);
}

// ---------------------------------------------------------------------------
// F32 flavour
// ---------------------------------------------------------------------------
constexpr int F_BK = 32;
template <int LAY> struct FPitch { static constexpr int v = (LAY == KC) ? 129 : 132; };

// stage one 128 x 32 operand tile: 1024 chunks of 4 elements, 4 per thread
template <typename T, int LAY, bool VEC>
__device__ __forceinline__ void f_load_tile(const T* __restrict__ base, long ld, int row0, int R,
                                            int k0, int kend, f32x4 (&reg)[4], int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = tid + c * 256;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (LAY == KC) {
      const int r = q >> 3, kc = (q & 7) << 2;
      const int gr = row0 + r, gk = k0 + kc;
      if (gr < R) {
        const T* p = base + (long)gr * ld + gk;
        if (VEC) {
          if (gk < kend) v = load4(p);
        } else {
          if (gk + 0 < kend) v.x = load1(p + 0);
          if (gk + 1 < kend) v.y = load1(p + 1);
          if (gk + 2 < kend) v.z = load1(p + 2);
          if (gk + 3 < kend) v.w = load1(p + 3);
        }
      }
    } else {
      const int k = q >> 5, rc = (q & 31) << 2;
      const int gk = k0 + k, gr = row0 + rc;
      if (gk < kend) {
        const T* p = base + (long)gk * ld + gr;
        if (VEC) {
          if (gr < R) v = load4(p);
        } else {
          if (gr + 0 < R) v.x = load1(p + 0);
          if (gr + 1 < R) v.y = load1(p + 1);
          if (gr + 2 < R) v.z = load1(p + 2);
          if (gr + 3 < R) v.w = load1(p + 3);
        }
      }
    }
    reg[c] = v;
  }
}

template <int LAY>
__device__ __forceinline__ void f_store_tile(float* __restrict__ s, const f32x4 (&reg)[4], int tid) {
  constexpr int P = FPitch<LAY>::v;
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = tid + c * 256;
    if (LAY == KC) {
      const int r = q >> 3, kc = (q & 7) << 2;
      s[(kc + 0) * P + r] = reg[c].x;
      s[(kc + 1) * P + r] = reg[c].y;
      s[(kc + 2) * P + r] = reg[c].z;
      s[(kc + 3) * P + r] = reg[c].w;
    } else {
      const int k = q >> 5, rc = (q & 31) << 2;
      *reinterpret_cast<f32x4*>(&s[k * P + rc]) = reg[c];
    }
  }
}

// shared epilogue: bias, store / atomic, BatchNorm column statistics
template <typename TC>
__device__ __forceinline__ void epilogue(const GemmParams& p, f32x16 (&acc)[2][2], float* smem,
                                         int tm, int tn, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  TC* C = reinterpret_cast<TC*>(p.C) + (long)blockIdx.z * p.c_split_stride;   // slab split-K (0 otherwise)
  const bool add_bias = p.bias != nullptr && (!p.atomic || blockIdx.z == 0);
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int gn = tn * BN + wn * 64 + j * 32 + l31;
    const float bv = (add_bias && gn < p.N) ? p.bias[gn] : 0.f;
#pragma unroll
    for (int i = 0; i < 2; ++i) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        const int gm = tm * BM + wm * 64 + i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half;
        if (gm < p.M && gn < p.N) {
          const float v = acc[i][j][r] + bv;
          if (p.atomic) {
            atomicAdd(reinterpret_cast<float*>(p.C) + (long)gm * p.ldc + gn, v);
          } else {
            store1(C + (long)gm * p.ldc + gn, v);
          }
        }
      }
    }
  }
  if (p.colstats != nullptr) {
    // rows >= M were staged as zeros, so they add nothing to the bias-free sums
    __syncthreads();  // everyone is done reading the operand tiles in smem
    float* red = smem;  // [2 stats][2 wm][128 cols]
#pragma unroll
    for (int j = 0; j < 2; ++j) {
      float s1 = 0.f, s2 = 0.f;
#pragma unroll
      for (int i = 0; i < 2; ++i) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float v = acc[i][j][r];
          s1 += v;
          s2 += v * v;
        }
      }
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (half == 0) {
        const int col = wn * 64 + j * 32 + l31;
        red[(0 * 2 + wm) * 128 + col] = s1;
        red[(1 * 2 + wm) * 128 + col] = s2;
      }
    }
    __syncthreads();
    const int stat = tid >> 7, col = tid & 127;
    const int gn = tn * BN + col;
    if (gn < p.N) {
      const double v = (double)red[(stat * 2 + 0) * 128 + col] + (double)red[(stat * 2 + 1) * 128 + col];
      const int rep = tm % p.nrep;
      unsafeAtomicAdd(&p.colstats[((long)rep * 2 + stat) * p.N + gn], v);
    }
  }
}

// one 128 x 128 tile of one K range: ``bid`` = the tile's block number (XCD-aware map), ``split`` = the K range
template <typename TA, typename TB, typename TC, int ALAY, int BLAY, bool VEC>
__device__ __forceinline__ void gemm_f32_body(const GemmParams& p, int bid, int split, float* smem) {
  constexpr int PA = FPitch<ALAY>::v, PB = FPitch<BLAY>::v;
  float* sA = smem;
  float* sB = smem + F_BK * PA;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  int tm, tn;
  xcd_tile_coords((p.M + BM - 1) / BM, (p.N + BN - 1) / BN, bid, tm, tn);

  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + F_BK - 1) / F_BK;

  const TA* A = reinterpret_cast<const TA*>(p.A);
  const TB* B = reinterpret_cast<const TB*>(p.B);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  f32x4 ra[4], rb[4];
  if (nt > 0) {
    f_load_tile<TA, ALAY, VEC>(A, p.lda, tm * BM, p.M, kbeg, kend, ra, tid);
    f_load_tile<TB, BLAY, VEC>(B, p.ldb, tn * BN, p.N, kbeg, kend, rb, tid);
    f_store_tile<ALAY>(sA, ra, tid);
    f_store_tile<BLAY>(sB, rb, tid);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const bool more = (t + 1 < nt);
    if (more) {
      const int k0 = kbeg + (t + 1) * F_BK;
      f_load_tile<TA, ALAY, VEC>(A, p.lda, tm * BM, p.M, k0, kend, ra, tid);
      f_load_tile<TB, BLAY, VEC>(B, p.ldb, tn * BN, p.N, k0, kend, rb, tid);
    }
    const float* pa = sA + half * PA + wm * 64 + l31;
    const float* pb = sB + half * PB + wn * 64 + l31;
#pragma unroll
    for (int kk = 0; kk < F_BK; kk += 2) {
      const float a0 = pa[kk * PA], a1 = pa[kk * PA + 32];
      const float b0 = pb[kk * PB], b1 = pb[kk * PB + 32];
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      f_store_tile<ALAY>(sA, ra, tid);
      f_store_tile<BLAY>(sB, rb, tid);
      __syncthreads();
    }
  }
  epilogue<TC>(p, acc, smem, tm, tn, tid);
}

template <typename TA, typename TB, typename TC, int ALAY, int BLAY, bool VEC>
__global__ __launch_bounds__(256) void gemm_f32_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem[F_BK * FPitch<ALAY>::v + F_BK * FPitch<BLAY>::v];
  gemm_f32_body<TA, TB, TC, ALAY, BLAY, VEC>(p, blockIdx.x, blockIdx.z, smem);
}

// Several small products in ONE launch (round 4: the temporal block's six weight gradients dW_l = dy_l^T . col_l were six
// launches of 1-24 tiles x <= 15 K ranges each, 26-47 us apiece in a chain on the weight-gradient stream: 0.23 ms at
// N = 32 and, sharing hardware queues with it, in the way of the dgrad chain -- without them the N = 32 step is 0.17 ms
// shorter).  Block b of the launch belongs to product i with first[i] <= b < first[i + 1]; within it the tile number runs
// fastest, then the K range.  fp32 RC x RC operands, fp32 result accumulated atomically (the caller zeroes it).
constexpr int GROUP_MAX = 8;
struct GemmGroup {
  GemmParams p[GROUP_MAX];
  int first[GROUP_MAX + 1];
  int ntiles[GROUP_MAX];
  int n;
};
__global__ __launch_bounds__(256) void gemm_f32_group_kernel(GemmGroup g) {
  __shared__ __attribute__((aligned(16))) float smem[F_BK * FPitch<RC>::v + F_BK * FPitch<RC>::v];
  int i = 0;
#pragma unroll
  for (int j = 1; j < GROUP_MAX; ++j)
    if (j < g.n && (int)blockIdx.x >= g.first[j]) i = j;
  const int local = (int)blockIdx.x - g.first[i];
  const int split = local / g.ntiles[i], bid = local - split * g.ntiles[i];
  gemm_f32_body<float, float, float, RC, RC, true>(g.p[i], bid, split, smem);
}

// ---------------------------------------------------------------------------
// BF16 flavour (KC operands).  Tile 128 x 64 per operand per step.
// ---------------------------------------------------------------------------
constexpr int H_BK = 64;
constexpr int H_PITCH = H_BK + 8;  // bf16 elements; 144 B rows

struct Raw8 { uint4 v; };  // 8 packed bf16

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 b;
  b.x = (bf16_t)lo;
  b.y = (bf16_t)hi;
  return *reinterpret_cast<uint32_t*>(&b);
}

__device__ __forceinline__ uint4 load8_as_bf16(const bf16_t* p) { return *reinterpret_cast<const uint4*>(p); }
__device__ __forceinline__ uint4 load8_as_bf16(const float* p) {
  const f32x4 lo = *reinterpret_cast<const f32x4*>(p);
  const f32x4 hi = *reinterpret_cast<const f32x4*>(p + 4);
  uint4 r;
  r.x = pack_bf16x2(lo.x, lo.y);
  r.y = pack_bf16x2(lo.z, lo.w);
  r.z = pack_bf16x2(hi.x, hi.y);
  r.w = pack_bf16x2(hi.z, hi.w);
  return r;
}

// 128 rows x 64 k = 1024 chunks of 8 elements, 4 per thread
template <typename T>
__device__ __forceinline__ void h_load_tile(const T* __restrict__ base, long ld, int row0, int R,
                                            int k0, int kend, uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = tid + c * 256;
    const int r = q >> 3, kc = (q & 7) << 3;
    const int gr = row0 + r, gk = k0 + kc;
    uint4 v = {0u, 0u, 0u, 0u};
    if (gr < R && gk < kend) v = load8_as_bf16(base + (long)gr * ld + gk);
    reg[c] = v;
  }
}

__device__ __forceinline__ void h_store_tile(bf16_t* __restrict__ s, const uint4 (&reg)[4], int tid) {
#pragma unroll
  for (int c = 0; c < 4; ++c) {
    const int q = tid + c * 256;
    const int r = q >> 3, kc = (q & 7) << 3;
    *reinterpret_cast<uint4*>(&s[r * H_PITCH + kc]) = reg[c];
  }
}

template <typename TA, typename TB, typename TC>
__global__ __launch_bounds__(256) void gemm_bf16_kernel(GemmParams p) {
  __shared__ __attribute__((aligned(16))) float smem_f[(2 * BM * H_PITCH * 2) / 4];
  bf16_t* sA = reinterpret_cast<bf16_t*>(smem_f);
  bf16_t* sB = sA + BM * H_PITCH;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1, l31 = lane & 31, half = lane >> 5;
  int tm, tn;
  tile_coords(p.M, p.N, tm, tn);

  const int kbeg = blockIdx.z * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg + H_BK - 1) / H_BK;

  const TA* A = reinterpret_cast<const TA*>(p.A);
  const TB* B = reinterpret_cast<const TB*>(p.B);

  f32x16 acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  uint4 ra[4], rb[4];
  if (nt > 0) {
    h_load_tile<TA>(A, p.lda, tm * BM, p.M, kbeg, kend, ra, tid);
    h_load_tile<TB>(B, p.ldb, tn * BN, p.N, kbeg, kend, rb, tid);
    h_store_tile(sA, ra, tid);
    h_store_tile(sB, rb, tid);
  }
  __syncthreads();

  for (int t = 0; t < nt; ++t) {
    const bool more = (t + 1 < nt);
    if (more) {
      const int k0 = kbeg + (t + 1) * H_BK;
      h_load_tile<TA>(A, p.lda, tm * BM, p.M, k0, kend, ra, tid);
      h_load_tile<TB>(B, p.ldb, tn * BN, p.N, k0, kend, rb, tid);
    }
    const bf16_t* pa = sA + (wm * 64 + l31) * H_PITCH + half * 8;
    const bf16_t* pb = sB + (wn * 64 + l31) * H_PITCH + half * 8;
#pragma unroll
    for (int kk = 0; kk < H_BK; kk += 16) {
      const bf16x8 a0 = *reinterpret_cast<const bf16x8*>(pa + kk);
      const bf16x8 a1 = *reinterpret_cast<const bf16x8*>(pa + 32 * H_PITCH + kk);
      const bf16x8 b0 = *reinterpret_cast<const bf16x8*>(pb + kk);
      const bf16x8 b1 = *reinterpret_cast<const bf16x8*>(pb + 32 * H_PITCH + kk);
      acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b0, acc[0][0], 0, 0, 0);
      acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a0, b1, acc[0][1], 0, 0, 0);
      acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b0, acc[1][0], 0, 0, 0);
      acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a1, b1, acc[1][1], 0, 0, 0);
    }
    __syncthreads();
    if (more) {
      h_store_tile(sA, ra, tid);
      h_store_tile(sB, rb, tid);
      __syncthreads();
    }
  }
  epilogue<TC>(p, acc, smem_f, tm, tn, tid);
}

// ---------------------------------------------------------------------------
// dispatch
// ---------------------------------------------------------------------------
template <typename TA, typename TB, typename TC, int ALAY, int BLAY>
void launch_f32(const GemmParams& p, bool vec, dim3 grid, hipStream_t s) {
  if (vec)
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, TC, ALAY, BLAY, true>), grid, dim3(256), 0, s, p);
  else
    hipLaunchKernelGGL((gemm_f32_kernel<TA, TB, TC, ALAY, BLAY, false>), grid, dim3(256), 0, s, p);
}

template <typename TA, typename TB, typename TC>
void launch_f32_lay(const GemmParams& p, int al, int bl, bool vec, dim3 grid, hipStream_t s) {
  if (al == KC && bl == KC) launch_f32<TA, TB, TC, KC, KC>(p, vec, grid, s);
  else if (al == KC && bl == RC) launch_f32<TA, TB, TC, KC, RC>(p, vec, grid, s);
  else if (al == RC && bl == KC) launch_f32<TA, TB, TC, RC, KC>(p, vec, grid, s);
  else launch_f32<TA, TB, TC, RC, RC>(p, vec, grid, s);
}

inline size_t esize(int dt) { return dt == PCAA_BF16 ? 2 : 4; }

}  // namespace

static int gemm_num_splits(int math, int K, int split_k, int* kps_out) {
  const int bk = (math == PCAA_BF16) ? H_BK : F_BK;
  int kps = (int)cdiv(cdiv(K, split_k), bk) * bk;
  if (kps < bk) kps = bk;
  if (kps_out) *kps_out = kps;
  return (int)cdiv(K, kps);
}

extern "C" int pcaa_gemm_num_splits(int math, int K, int split_k) {
  if (K < 1 || split_k < 1) return 0;
  return gemm_num_splits(math, K, split_k, nullptr);
}

static int gemm_impl(int math,
                     const void* A, int a_dtype, int a_layout, long lda,
                     const void* B, int b_dtype, int b_layout, long ldb,
                     void* C, int c_dtype, long ldc,
                     int M, int N, int K,
                     const float* bias, double* colstats, int nrep,
                     int split_k, int accumulate, long c_split_stride, void* stream) {
  PCAA_CHECK_ARG(A && B && C, "pcaa_gemm: null operand");
  PCAA_CHECK_ARG(M > 0 && N > 0 && K > 0, "pcaa_gemm: bad shape M=%d N=%d K=%d", M, N, K);
  PCAA_CHECK_ARG(math == PCAA_F32 || math == PCAA_BF16, "pcaa_gemm: bad math %d", math);
  PCAA_CHECK_ARG((a_dtype | 1) == 1 && (b_dtype | 1) == 1 && (c_dtype | 1) == 1, "pcaa_gemm: bad dtype");
  PCAA_CHECK_ARG((a_layout | 1) == 1 && (b_layout | 1) == 1, "pcaa_gemm: bad layout");
  PCAA_CHECK_ARG(split_k >= 1, "pcaa_gemm: split_k must be >= 1");
  PCAA_CHECK_ARG(lda >= (a_layout == KC ? K : M) && ldb >= (b_layout == KC ? K : N) && ldc >= N,
                 "pcaa_gemm: leading dimension too small");
  const int atomic = (c_split_stride == 0 && (split_k > 1 || accumulate)) ? 1 : 0;
  PCAA_CHECK_ARG(!atomic || c_dtype == PCAA_F32, "pcaa_gemm: atomic accumulation needs fp32 C");
  PCAA_CHECK_ARG(!(atomic && colstats), "pcaa_gemm: column statistics need a single K pass");
  PCAA_CHECK_ARG(!colstats || nrep >= 1, "pcaa_gemm: nrep must be >= 1 with colstats");

  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.B = B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = K;
  p.bias = bias; p.colstats = colstats; p.nrep = nrep > 0 ? nrep : 1;
  p.atomic = atomic;
  p.nsplit = 1;
  p.split_fast = 0;

  p.c_split_stride = c_split_stride;
  int kps = 0;
  const int nsplit = gemm_num_splits(math, K, split_k, &kps);
  p.k_per_split = kps;
  const long ntiles = cdiv(M, BM) * cdiv(N, BN);
  PCAA_CHECK_ARG(ntiles < (1L << 31), "pcaa_gemm: too many tiles");
  dim3 grid((unsigned)ntiles, 1, (unsigned)nsplit);
  hipStream_t s = as_stream(stream);

  if (math == PCAA_BF16) {
    // big shapes: 256x256-tile kernel (KC x KC forward/dgrad, RC x RC wgrad)
    if (pcaa_launch_gemm_bf16_big(p, a_dtype, a_layout, b_dtype, b_layout, c_dtype, nsplit, s))
      PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm(bf16, 256x256)");
    PCAA_CHECK_ARG(a_dtype == PCAA_BF16, "pcaa_gemm: bf16 math on this shape needs a bf16 A operand");
    PCAA_CHECK_ARG(a_layout == KC && b_layout == KC, "pcaa_gemm: bf16 math on this shape needs KC operands");
    PCAA_CHECK_ARG(K % 8 == 0 && lda % 8 == 0 && ldb % 8 == 0, "pcaa_gemm: bf16 math needs K, lda, ldb %% 8 == 0");
    PCAA_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "pcaa_gemm: bf16 math needs 16-B aligned operands");
    if (b_dtype == PCAA_BF16 && c_dtype == PCAA_BF16)
      hipLaunchKernelGGL((gemm_bf16_kernel<bf16_t, bf16_t, bf16_t>), grid, dim3(256), 0, s, p);
    else if (b_dtype == PCAA_BF16 && c_dtype == PCAA_F32)
      hipLaunchKernelGGL((gemm_bf16_kernel<bf16_t, bf16_t, float>), grid, dim3(256), 0, s, p);
    else if (b_dtype == PCAA_F32 && c_dtype == PCAA_BF16)
      hipLaunchKernelGGL((gemm_bf16_kernel<bf16_t, float, bf16_t>), grid, dim3(256), 0, s, p);
    else
      hipLaunchKernelGGL((gemm_bf16_kernel<bf16_t, float, float>), grid, dim3(256), 0, s, p);
    PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm(bf16)");
  }

  // vector (4-element) staging is legal when every 4-chunk is aligned and
  // entirely inside or outside the operand
  auto vec_ok = [&](const void* ptr, int dt, int lay, long ld, int rows) {
    const size_t align = 4 * esize(dt);
    if (((uintptr_t)ptr % align) != 0 || (ld % 4) != 0) return false;
    if (lay == KC) return (K % 4) == 0;
    return (rows % 4) == 0;
  };
  const bool vec = vec_ok(A, a_dtype, a_layout, lda, M) && vec_ok(B, b_dtype, b_layout, ldb, N);

  if (a_dtype == PCAA_F32 && b_dtype == PCAA_F32 && c_dtype == PCAA_F32)
    launch_f32_lay<float, float, float>(p, a_layout, b_layout, vec, grid, s);
  else if (a_dtype == PCAA_BF16 && b_dtype == PCAA_BF16 && c_dtype == PCAA_F32)
    launch_f32_lay<bf16_t, bf16_t, float>(p, a_layout, b_layout, vec, grid, s);
  else if (a_dtype == PCAA_BF16 && b_dtype == PCAA_F32 && c_dtype == PCAA_F32)
    launch_f32_lay<bf16_t, float, float>(p, a_layout, b_layout, vec, grid, s);
  else if (a_dtype == PCAA_F32 && b_dtype == PCAA_F32 && c_dtype == PCAA_BF16)
    launch_f32_lay<float, float, bf16_t>(p, a_layout, b_layout, vec, grid, s);
  else {
    pcaa_set_error("pcaa_gemm: unsupported dtype combination a=%d b=%d c=%d for fp32 math", a_dtype, b_dtype, c_dtype);
    return PCAA_ERR_INVALID_ARG;
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm(f32)");
}

extern "C" int pcaa_gemm(int math,
                         const void* A, int a_dtype, int a_layout, long lda,
                         const void* B, int b_dtype, int b_layout, long ldb,
                         void* C, int c_dtype, long ldc,
                         int M, int N, int K,
                         const float* bias, double* colstats, int nrep,
                         int split_k, int accumulate, void* stream) {
  return gemm_impl(math, A, a_dtype, a_layout, lda, B, b_dtype, b_layout, ldb, C, c_dtype, ldc, M, N, K, bias,
                   colstats, nrep, split_k, accumulate, 0, stream);
}

/* n <= 8 products C_i[M_i, N_i] += A_i^T . B_i in one launch: A_i [K_i, M_i], B_i [K_i, N_i], C_i [M_i, N_i] fp32, contiguous
 * rows (lda = M_i, ldb = N_i, ldc = N_i), 16-B aligned, M_i and N_i multiples of 4; C_i is ACCUMULATED into (atomics over
 * the K ranges): the caller zeroes it.  split_k[i]: how many ranges product i's contraction is cut into. */
extern "C" int pcaa_gemm_group_rc_f32(int n, const void* const* A, const void* const* B, void* const* C, const int* M,
                                      const int* N, const int* K, const int* split_k, void* stream) {
  PCAA_CHECK_ARG(n >= 1 && n <= GROUP_MAX && A && B && C && M && N && K && split_k, "pcaa_gemm_group_rc_f32: bad args (n <= %d)",
                 GROUP_MAX);
  GemmGroup g;
  memset(&g, 0, sizeof(g));
  g.n = n;
  long total = 0;
  for (int i = 0; i < n; ++i) {
    PCAA_CHECK_ARG(A[i] && B[i] && C[i] && M[i] > 0 && N[i] > 0 && K[i] > 0 && split_k[i] >= 1,
                   "pcaa_gemm_group_rc_f32: product %d: bad shape", i);
    PCAA_CHECK_ARG((M[i] % 4) == 0 && (N[i] % 4) == 0 && ((uintptr_t)A[i] % 16) == 0 && ((uintptr_t)B[i] % 16) == 0 &&
                   ((uintptr_t)C[i] % 16) == 0, "pcaa_gemm_group_rc_f32: product %d: M, N multiples of 4, 16-B aligned operands", i);
    GemmParams& p = g.p[i];
    p.A = A[i]; p.B = B[i]; p.C = C[i];
    p.lda = M[i]; p.ldb = N[i]; p.ldc = N[i];
    p.M = M[i]; p.N = N[i]; p.K = K[i];
    p.nrep = 1;
    p.atomic = 1;
    p.nsplit = 1;
    int kps = 0;
    const int nsplit = gemm_num_splits(PCAA_F32, K[i], split_k[i], &kps);
    p.k_per_split = kps;
    g.ntiles[i] = (int)(cdiv(M[i], BM) * cdiv(N[i], BN));
    g.first[i] = (int)total;
    total += (long)g.ntiles[i] * nsplit;
  }
  for (int i = n; i <= GROUP_MAX; ++i) g.first[i] = (int)total;
  PCAA_CHECK_ARG(total < (1L << 31), "pcaa_gemm_group_rc_f32: too many blocks");
  hipLaunchKernelGGL(gemm_f32_group_kernel, dim3((unsigned)total), dim3(256), 0, as_stream(stream), g);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm_group_rc_f32");
}

// split-K without atomics: split s writes its partial product to slabs + s * slab_stride
extern "C" int pcaa_gemm_slabs(int math,
                               const void* A, int a_dtype, int a_layout, long lda,
                               const void* B, int b_dtype, int b_layout, long ldb,
                               float* slabs, long slab_stride, int M, int N, int K, int split_k, void* stream) {
  PCAA_CHECK_ARG(slabs && slab_stride >= (long)M * N, "pcaa_gemm_slabs: slab stride must cover an M x N tile");
  return gemm_impl(math, A, a_dtype, a_layout, lda, B, b_dtype, b_layout, ldb, slabs, PCAA_F32, N, M, N, K, nullptr,
                   nullptr, 0, split_k, 0, slab_stride, stream);
}

// ---------------------------------------------------------------------------------------------------------------
// Split product (round 3): both fp32 operands are given as [hi | lo] fp16 images (pcaa_split_f16:
// hi = fp16(s e), lo = fp16(s e - hi), s a power of two) and the product is hi.hi + lo.hi + hi.lo on the f16 MFMA
// pipe (the bf16 rate) with fp32 accumulation: three passes of the LDS-DMA kernel's K loop over the same tile (one
// launch, K' = 3K), relative error of a product ~2^-21 -- the parity-grade mode that does not need the 1/16-rate fp32
// MFMA.  (A bf16 pair carries 16 mantissa bits: 4.5e-6 per product, and the first PointNet layer's weight gradient
// missed the 5e-4 gate at 5.7e-4; the fp16 pair carries 22.)  out_scale = 1 / (s_A s_B).  KC operands: [rows, 2K] (lo at column K + k); RC operands: [K, 2 rows] (lo at column rows + r).
// ---------------------------------------------------------------------------------------------------------------
bool pcaa_gemm_v2_is_enabled();      // gemm_bf16.hip
// M need not be a multiple of 256 where the 4-wave tile loop serves the launch (round 4: it reads the rows past M as
// zeros and skips them on the way out): contraction >= 320 deep, KC operands
//  the predicates and the entry points' argument checks together state everything launch_dma asks of a
// ragged launch, so that a shape reported as supported cannot end in PCAA_ERR_LAUNCH: the result (and the fused dgrad's
// y) is addressed through 32-bit buffer offsets over ceil(M / 256) * 256 rows (the rows past M wrap otherwise), its
// leading dimension is a multiple of 8 elements and 16-B aligned, and the first-layer recompute variant
// (pcaa_gemm_dgrad_bn with x) has no ragged instantiation.
static bool ragged_m_ok(int M, int N, int K) {
  // N columns of at most 4 bytes: the predicate's bound; the entry points re-check with the real ld and element size
  return pcaa_gemm_v2_is_enabled() && K >= 320 && (long)cdiv(M, 256) * 256 * N * 4 < (1L << 32);
}
static bool ragged_out_ok(int M, const void* C, long ldc, int elem) {
  return (M % 256) == 0 || ((ldc % 8) == 0 && ((uintptr_t)C % 16) == 0 && (long)cdiv(M, 256) * 256 * ldc * elem < (1L << 32));
}
// (round 5: the 4-wave loops are the only LDS-DMA kernels left -- a contraction of at least five 64-deep steps, three
// K-long passes for the split operands)
extern "C" int pcaa_gemm_split3_supported(int M, int N, int K) {
  return M > 0 && N > 0 && K > 0 && pcaa_gemm_v2_is_enabled() && 3L * K >= 320 &&
         ((M % 256) == 0 || ragged_m_ok(M, N, K)) && (N % 256) == 0 && (K % 64) == 0;
}

static int gemm_split3_impl(const void* A, const void* B, int layout, long lda, long ldb, void* C, long ldc, int M, int N,
                            int K, double* colstats, int nrep, int split_k, long c_split_stride, float out_scale,
                            void* stream) {
  PCAA_CHECK_ARG(A && B && C, "pcaa_gemm_split3: null operand");
  PCAA_CHECK_ARG(layout == KC || layout == RC, "pcaa_gemm_split3: bad layout");
  PCAA_CHECK_ARG(pcaa_gemm_split3_supported(M, N, K) && (layout == KC || (M % 256) == 0),
                 "pcaa_gemm_split3: N must be a multiple of 256, K of 64, and M of 256 unless the 4-wave loop takes the "
                 "launch (KC operands, K >= 320, no K split) (M=%d N=%d K=%d)", M, N, K);
  PCAA_CHECK_ARG((M % 256) == 0 || (split_k <= 1 && c_split_stride == 0), "pcaa_gemm_split3: a partial last row tile needs a "
                 "single K pass");
  PCAA_CHECK_ARG(ragged_out_ok(M, C, ldc, 4), "pcaa_gemm_split3: a partial last row tile needs ldc %% 8 == 0, a 16-B aligned "
                 "C and ceil(M / 256) * 256 * ldc * 4 < 4 GiB (M=%d ldc=%ld)", M, ldc);
  PCAA_CHECK_ARG((long)K * 3 < (1L << 31), "pcaa_gemm_split3: K too large");
  const long a_cols = layout == KC ? 2L * K : 2L * M, b_cols = layout == KC ? 2L * K : 2L * N;
  PCAA_CHECK_ARG(lda >= a_cols && ldb >= b_cols && ldc >= N && (lda % 8) == 0 && (ldb % 8) == 0,
                 "pcaa_gemm_split3: leading dimensions must cover the [hi | lo] images");
  PCAA_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)B % 16) == 0, "pcaa_gemm_split3: 16-B aligned operands");
  PCAA_CHECK_ARG(!(colstats && (split_k > 1 || c_split_stride)), "pcaa_gemm_split3: column statistics need a single K pass");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.B = B; p.C = C;
  p.lda = lda; p.ldb = ldb; p.ldc = ldc;
  p.M = M; p.N = N; p.K = 3 * K;
  p.colstats = colstats; p.nrep = nrep > 0 ? nrep : 1;
  p.seg_len = K;
  p.out_scale = out_scale;
  const long half_a = layout == KC ? K : M, half_b = layout == KC ? K : N;
  p.seg_off_a[0] = 0; p.seg_off_a[1] = half_a; p.seg_off_a[2] = 0;        // hi, lo, hi
  p.seg_off_b[0] = 0; p.seg_off_b[1] = 0;      p.seg_off_b[2] = half_b;   // hi, hi, lo
  p.c_split_stride = c_split_stride;
  p.atomic = 0;
  int kps = 0;
  const int nsplit = gemm_num_splits(PCAA_BF16, 3 * K, split_k, &kps);
  p.k_per_split = kps;
  const long ntiles = cdiv(M, 256) * cdiv(N, 256);
  if (!pcaa_launch_gemm_bf16_big(p, PCAA_BF16, layout, PCAA_BF16, layout, PCAA_F32, nsplit, as_stream(stream))) {
    pcaa_set_error("pcaa_gemm_split3: shape not served by the LDS-DMA kernel (ntiles %ld, nsplit %d)", ntiles, nsplit);
    return PCAA_ERR_INVALID_ARG;
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm_split3");
}

extern "C" int pcaa_gemm_split3(const void* A, const void* B, int layout, long lda, long ldb, float* C, long ldc, int M,
                                int N, int K, double* colstats, int nrep, float out_scale, void* stream) {
  return gemm_split3_impl(A, B, layout, lda, ldb, C, ldc, M, N, K, colstats, nrep, 1, 0, out_scale, stream);
}

extern "C" int pcaa_gemm_split3_num_splits(int K, int split_k) {
  int kps = 0;
  return gemm_num_splits(PCAA_BF16, 3 * K, split_k, &kps);
}

extern "C" int pcaa_gemm_slabs_split3(const void* A, const void* B, int layout, long lda, long ldb, float* slabs,
                                      long slab_stride, int M, int N, int K, int split_k, float out_scale, void* stream) {
  PCAA_CHECK_ARG(slabs && slab_stride >= (long)M * N, "pcaa_gemm_slabs_split3: slab stride must cover an M x N tile");
  return gemm_split3_impl(A, B, layout, lda, ldb, slabs, N, M, N, K, nullptr, 0, split_k, slab_stride, out_scale, stream);
}

extern "C" int pcaa_gemm_dgrad_bn_supported(int M, int N, int K) {
  return M > 0 && N > 0 && K >= 320 && pcaa_gemm_v2_is_enabled() && ((M % 256) == 0 || ragged_m_ok(M, N, K)) &&
         (N % 256) == 0 && (K % 64) == 0;
}

extern "C" int pcaa_gemm_dgrad_bn(const void* dy, long lddy, const void* Wt, long ldw, const void* y, void* dz,
                                  long ld, const float* scale, const float* shift, const float* mean,
                                  const float* rstd, double* stats, int nrep, int M, int N, int K,
                                  const float* x, int xc, const float* W1, void* stream) {
  PCAA_CHECK_ARG(dy && Wt && dz && scale && shift && mean && rstd && stats, "pcaa_gemm_dgrad_bn: null pointer");
  // (x / xc / W1: the first-layer recompute variant of rounds 1-4 -- y rebuilt from the points in the epilogue; it lived in
  // the 8-wave kernel, was never faster than the separate statistics pass and left with that kernel in round 5.  The
  // arguments stay in the signature; passing them is an argument error.)
  (void)xc; (void)W1;
  PCAA_CHECK_ARG(y != nullptr && x == nullptr, "pcaa_gemm_dgrad_bn: y is required (the recompute variant -- x, xc, W1 -- was "
                 "removed in round 5)");
  PCAA_CHECK_ARG(pcaa_gemm_dgrad_bn_supported(M, N, K), "pcaa_gemm_dgrad_bn: M, N must be multiples of 256 and K of 64 "
                 "(M=%d N=%d K=%d)", M, N, K);
  PCAA_CHECK_ARG(lddy >= K && ldw >= K && ld >= N && (lddy % 8) == 0 && (ldw % 8) == 0 && (ld % 8) == 0 && nrep >= 1,
                 "pcaa_gemm_dgrad_bn: bad leading dimension / nrep");
  PCAA_CHECK_ARG(ragged_out_ok(M, dz, ld, 2), "pcaa_gemm_dgrad_bn: a partial last row tile needs ceil(M / 256) * 256 * ld * 2 "
                 "< 4 GiB (M=%d ld=%ld)", M, ld);
  PCAA_CHECK_ARG(((uintptr_t)dy % 16) == 0 && ((uintptr_t)Wt % 16) == 0 && (!y || ((uintptr_t)y % 16) == 0) &&
                 ((uintptr_t)dz % 16) == 0 && ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0 &&
                 ((uintptr_t)mean % 16) == 0 && ((uintptr_t)rstd % 16) == 0, "pcaa_gemm_dgrad_bn: 16-B alignment");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = dy; p.B = Wt; p.C = dz;
  p.lda = lddy; p.ldb = ldw; p.ldc = ld;
  p.M = M; p.N = N; p.K = K;
  p.colstats = stats; p.nrep = nrep;
  p.ep_y = y; p.ep_scale = scale; p.ep_shift = shift; p.ep_mean = mean; p.ep_rstd = rstd;
  if (!pcaa_launch_gemm_dgrad_bn(p, as_stream(stream))) {
    pcaa_set_error("pcaa_gemm_dgrad_bn: launch configuration failed");
    return PCAA_ERR_LAUNCH;
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm_dgrad_bn");
}

/* The same fusion for the split-fp16 parity mode: dy and Wt are [hi | lo] fp16 images (pcaa_split_f16: dy [M, 2K],
 * Wt [N, 2K]), y and dz fp32 [M, ld]; ELU' and the statistics in the arithmetic of pcaa_bn_act_bwd_dz on fp32 tensors. */
extern "C" int pcaa_gemm_dgrad_bn_split3(const void* dy_img, long lddy, const void* Wt_img, long ldw, const float* y,
                                         float* dz, long ld, const float* scale, const float* shift, const float* mean,
                                         const float* rstd, double* stats, int nrep, int M, int N, int K,
                                         float out_scale, void* stream) {
  PCAA_CHECK_ARG(dy_img && Wt_img && y && dz && scale && shift && mean && rstd && stats,
                 "pcaa_gemm_dgrad_bn_split3: null pointer");
  // (the split operands walk three K-long passes: the same shape rule as pcaa_gemm_split3)
  PCAA_CHECK_ARG(pcaa_gemm_split3_supported(M, N, K), "pcaa_gemm_dgrad_bn_split3: M, N must be multiples of 256 and K "
                 "of 64, 3 K >= 320 (M=%d N=%d K=%d)", M, N, K);
  PCAA_CHECK_ARG((long)K * 3 < (1L << 31) && lddy >= 2L * K && ldw >= 2L * K && ld >= N && (lddy % 8) == 0 &&
                 (ldw % 8) == 0 && (ld % 4) == 0 && nrep >= 1, "pcaa_gemm_dgrad_bn_split3: bad leading dimension / nrep");
  PCAA_CHECK_ARG(ragged_out_ok(M, dz, ld, 4), "pcaa_gemm_dgrad_bn_split3: a partial last row tile needs ld %% 8 == 0 and "
                 "ceil(M / 256) * 256 * ld * 4 < 4 GiB (M=%d ld=%ld)", M, ld);
  PCAA_CHECK_ARG(((uintptr_t)dy_img % 16) == 0 && ((uintptr_t)Wt_img % 16) == 0 && ((uintptr_t)y % 16) == 0 &&
                 ((uintptr_t)dz % 16) == 0 && ((uintptr_t)scale % 16) == 0 && ((uintptr_t)shift % 16) == 0 &&
                 ((uintptr_t)mean % 16) == 0 && ((uintptr_t)rstd % 16) == 0, "pcaa_gemm_dgrad_bn_split3: 16-B alignment");
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = dy_img; p.B = Wt_img; p.C = dz;
  p.lda = lddy; p.ldb = ldw; p.ldc = ld;
  p.M = M; p.N = N; p.K = 3 * K;
  p.colstats = stats; p.nrep = nrep;
  p.seg_len = K;
  p.out_scale = out_scale;
  p.seg_off_a[0] = 0; p.seg_off_a[1] = K; p.seg_off_a[2] = 0;        // hi, lo, hi
  p.seg_off_b[0] = 0; p.seg_off_b[1] = 0; p.seg_off_b[2] = K;        // hi, hi, lo
  p.ep_y = y; p.ep_scale = scale; p.ep_shift = shift; p.ep_mean = mean; p.ep_rstd = rstd;
  if (!pcaa_launch_gemm_dgrad_bn(p, as_stream(stream))) {
    pcaa_set_error("pcaa_gemm_dgrad_bn_split3: launch configuration failed");
    return PCAA_ERR_LAUNCH;
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm_dgrad_bn_split3");
}

extern "C" int pcaa_gemm_affine_elu(const void* A, long lda, const void* W, long ldw, void* out, long ldo,
                                    const float* scale, const float* shift, int M, int N, int K, int pool_rows,
                                    void* stream) {
  PCAA_CHECK_ARG(A && W && out && scale && shift, "pcaa_gemm_affine_elu: null pointer");
  PCAA_CHECK_ARG(pool_rows == 0 || pool_rows == 32 || pool_rows == 64 || pool_rows == 128,
                 "pcaa_gemm_affine_elu: pool_rows must be 0, 32, 64 or 128");
  PCAA_CHECK_ARG(pcaa_gemm_dgrad_bn_supported(M, N, K), "pcaa_gemm_affine_elu: M, N must be multiples of 256 and K of 64 "
                 "(M=%d N=%d K=%d)", M, N, K);
  PCAA_CHECK_ARG(lda >= K && ldw >= K && ldo >= N && (lda % 8) == 0 && (ldw % 8) == 0 && (pool_rows != 0 || (ldo % 8) == 0),
                 "pcaa_gemm_affine_elu: bad leading dimension");
  PCAA_CHECK_ARG(((uintptr_t)A % 16) == 0 && ((uintptr_t)W % 16) == 0 && ((uintptr_t)out % 16) == 0,
                 "pcaa_gemm_affine_elu: 16-B alignment");
  PCAA_CHECK_ARG(ragged_out_ok(M, out, ldo, pool_rows ? 4 : 2), "pcaa_gemm_affine_elu: a partial last row tile needs ldo %% 8 "
                 "== 0 and ceil(M / 256) * 256 * ldo * size < 4 GiB (M=%d ldo=%ld)", M, ldo);
  GemmParams p;
  memset(&p, 0, sizeof(p));
  p.A = A; p.B = W; p.C = out;
  p.lda = lda; p.ldb = ldw; p.ldc = ldo;
  p.M = M; p.N = N; p.K = K;
  p.nrep = 1;
  p.ep_scale = scale; p.ep_shift = shift;
  p.ep_xc = pool_rows;
  if (!pcaa_launch_gemm_affine_elu(p, as_stream(stream))) {
    pcaa_set_error("pcaa_gemm_affine_elu: launch configuration failed");
    return PCAA_ERR_LAUNCH;
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_gemm_affine_elu");
}

