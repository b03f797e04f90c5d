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
// The LDS-DMA tile loops (csrc/gemm_v2.h, namespace v2): both operands already bf16 (or [hi | lo] fp16 images) in HBM;
// tiles go HBM -> LDS with buffer_load_dwordx4 ... lds (no VGPR staging, no ds_write pass); the LDS images are unpadded,
// so the bank-conflict fix is an XOR swizzle applied to the per-lane SOURCE address (the DMA writes LDS linearly: base +
// lane * 16) and again on the fragment reads:
//   KC image [256 rows][64 k] (128-B rows): 16-B granule g of row r holds global k-granule g ^ ((r >> 1) & 7);
//   RC image [64 k][256 rows] (512-B rows): 16-B granule c of k-row k holds global row-granule c ^ 4 (k & 3) -- the 4
//     k-rows of a transpose read land on disjoint bank quarters.
// Rounds 1-3 ran an 8-wave kernel here (A in a ring of three stages, B of two, 128 x 64 wave tiles); round 4's 4-wave
// loops replaced it on every shape the product launches and round 5 removed it (profiles/r05_bench_kernel_names.txt,
// r05_tests_kernel_names.txt: no BASELINE config reached it; the few test shapes that did -- K < 320, K-split atomics --
// take the register-staged kernel above).  docs/LAB_LOG.md sections 4, 7, 9 keep its history.
// ===========================================================================
typedef __amdgpu_buffer_rsrc_t buf_rsrc_t;
__device__ __forceinline__ buf_rsrc_t make_rsrc(const void* base, long bytes) {
  return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)(unsigned)bytes, 0x00020000);
}
// per-lane byte offsets for pieces of even / odd index (a piece = 1 KB of a stage image; the per-lane part of its source
// address never changes, everything else -- tile origin, piece, K step -- is wave-uniform and rides in the scalar offset)
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
// workgroup barrier that leaves vector-memory operations (the next tile's LDS-DMA pieces, this tile's C stores) alone
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// what leaves the kernel
enum { EPI_PLAIN = 0, EPI_DGRAD_BN = 1, EPI_AFFINE = 3, EPI_POOL1 = 4, EPI_POOL2 = 5, EPI_POOL4 = 6 };

// per-lane element offset of fragment rows [row_base, row_base+32) at k-step 0 (row_base % 32 == 0), 32x32x16 shape
template <int LAY>
__device__ __forceinline__ int dma_frag_offset(int row_base, int lane) {
  if (LAY == KC) return (row_base + (lane & 31)) * 64;
  const int j = lane & 15, mb = 16 * ((lane >> 4) & 1), h = lane >> 5;
  const int q = j >> 2, ch = row_base + mb + 4 * (j & 3);
  return (8 * h + q) * 256 + ((((ch >> 3) ^ (4 * q)) << 3) | (ch & 7));
}

#include "gemm_v2.h"      // the 4-wave tile loops (namespace v2)


// The 4-wave tile loop (gemm_v2.h) serves every KC x KC launch without K splits whose contraction is at least five
// 64-deep steps long (its ticket hand-off needs four of them); pcaa_gemm_v2_enable(0) declines them all (lab A/B against
// the register-staged kernel; the fused entry points then report their shapes unsupported).
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
    if ((EPI == EPI_PLAIN || EPI == EPI_DGRAD_BN) && p.colstats != nullptr && grid.y == 1)
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
  if constexpr (ALAY == KC && BLAY == KC) {
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
  // (anything else -- a contraction shorter than five steps, K splits with atomics, a partial tile the loop cannot take --
  // is declined: the caller falls back to the register-staged kernel, the fused entry points report the shape unsupported)
  (void)decided;
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
  if (p.ep_y == nullptr) return false;
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
      // (declined -- a contraction shorter than five steps, K splits, a partial last row tile it cannot address: the
      // register-staged kernel below)
      if (cf ? launch_dma<float, KC, KC, EPI_PLAIN>(p, grid, stream) : launch_dma<bf16_t, KC, KC, EPI_PLAIN>(p, grid, stream))
        return true;
    } else if (cf) {
      // (declined -- atomics, a partial tile, an operand beyond 4 GiB: the register-staged kernel below)
      if (launch_dma<float, RC, RC, EPI_PLAIN>(p, grid, stream)) return true;
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
