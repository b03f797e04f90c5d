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

// Full-tile epilogue of the LDS-DMA kernels (M, N multiples of 256: no bounds checks, 32-bit
// offsets).  PMC/diagnostic builds put 40 % of the forward GEMM's time in the generic epilogue
// (4.4 K instructions, per-element exec-mask branches, 64-bit address math, 240 spill
// instructions), so this one is specialised at compile time:
//   bf16 out : neighbouring lanes exchange one value (DPP quad_perm [1,0,3,2]) so that every lane
//              owns two adjacent columns of one row -> v_cvt_pk_bf16_f32 + ds_write_b32 (64 per
//              lane instead of 128 ds_write_b16), then 16-B-per-lane row-contiguous stores.
//   fp32 out : plain stores (slab split-K: C is offset by split * c_split_stride) or atomics.
__device__ __forceinline__ float dpp_swap_neighbour(float v) {
  return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

// AFFINE (bf16 out, eval-mode BatchNorm): the stored value is ELU(scale[col]*acc + shift[col]) -- with running
// statistics the BatchNorm is a per-channel affine map known BEFORE the product, so the activation leaves the
// GEMM directly and the separate BN+ELU pass over [P, ch] (read + write) of the train-mode path is gone.
template <typename TC, bool AFFINE = false>
__device__ __forceinline__ void epilogue_full_tile(const GemmParams& p, f32x16 (&acc)[FM][FN], bf16_t* smem,
                                                   int tm, int tn, int tid, int split) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  const bool add_bias = p.bias != nullptr && (!p.atomic || split == 0);
  if constexpr (sizeof(TC) == 2) {
    constexpr int PITCH = 64;
    uint32_t* w32 = reinterpret_cast<uint32_t*>(smem + wave * 128 * PITCH);
    const bool odd = lane & 1;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const float bv = add_bias ? p.bias[tn * BN + wn * 64 + j * 32 + l31] : 0.f;
      const float esc = AFFINE ? p.ep_scale[tn * BN + wn * 64 + j * 32 + l31] : 1.f;
      const float esh = AFFINE ? p.ep_shift[tn * BN + wn * 64 + j * 32 + l31] : 0.f;
      const int colw = (j * 32 + (l31 & ~1)) >> 1;            // 32-bit word index of the column pair
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
          float va = acc[i][j][r] + bv, vb = acc[i][j][r + 1] + bv;   // rows R and R+1, my column
          if constexpr (AFFINE) {
            va = fmaf(va, esc, esh); vb = fmaf(vb, esc, esh);
            va = va > 0.f ? va : __expf(va) - 1.f;
            vb = vb > 0.f ? vb : __expf(vb) - 1.f;
          }
          const float got = dpp_swap_neighbour(odd ? va : vb);
          // even lane: row R, (mine, right neighbour's) ; odd lane: row R+1, (left neighbour's, mine)
          const uint32_t packed = odd ? pack2(got, vb) : pack2(va, got);
          const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half + (odd ? 1 : 0);
          w32[row * (PITCH / 2) + colw] = packed;
        }
    }
    __syncthreads();
    bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + (long)(tm * BM + wm * 128) * p.ldc + tn * BN + wn * 64;
    const bf16_t* w = smem + wave * 128 * PITCH;
    const int cg = (lane & 7) * 8, r0 = lane >> 3;
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
      const int row = pass * 8 + r0;
      *reinterpret_cast<uint4*>(C + (long)row * p.ldc + cg) = *reinterpret_cast<const uint4*>(&w[row * PITCH + cg]);
    }
    __syncthreads();   // the statistics reduction reuses this LDS
  } else {
    float* C = reinterpret_cast<float*>(p.C) + (long)split * p.c_split_stride +
               (long)(tm * BM + wm * 128 + 4 * half) * p.ldc + tn * BN + wn * 64 + l31;
#pragma unroll
    for (int j = 0; j < FN; ++j) {
      const float bv = add_bias ? p.bias[tn * BN + wn * 64 + j * 32 + l31] : 0.f;
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          float* dst = C + (long)(i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc + j * 32;
          const float v = acc[i][j][r] + bv;
          if (p.atomic) atomicAdd(dst, v);
          else *dst = v;
        }
    }
  }
}

// Eval-mode LAST PointNet layer: BatchNorm (affine) + ELU + the mean over the N points of a frame
// (AvgPool2d((1,N)), models.py:242-243, :282) straight from the accumulators: a wave holds 128 rows x 64
// columns of the tile = IPG-block groups of 32*IPG consecutive rows (IPG = N/32 in {1,2,4}), so a group's
// column mean is a sum over the lane's registers plus one cross-half shuffle; the [P, ch] activation is never
// written or re-read (2 x 8 GB per 1024 sequences at N=128).  out fp32 [P/N, ch].
template <int IPG>
__device__ __forceinline__ void epilogue_affine_meanpool(const GemmParams& p, f32x16 (&acc)[FM][FN], int tm, int tn,
                                                         int tid) {
  static_assert(FM % IPG == 0, "groups must not straddle waves");
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  float* out = reinterpret_cast<float*>(p.C);
  const float inv_n = 1.f / (32 * IPG);
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int col = tn * BN + wn * 64 + j * 32 + l31;
    const float esc = p.ep_scale[col], esh = p.ep_shift[col];
#pragma unroll
    for (int g0 = 0; g0 < FM; g0 += IPG) {
      float sum = 0.f;
#pragma unroll
      for (int i = g0; i < g0 + IPG; ++i)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
          const float z = fmaf(acc[i][j][r], esc, esh);
          sum += z > 0.f ? z : __expf(z) - 1.f;
        }
      sum += __shfl_xor(sum, 32, 64);
      const long grp = ((long)tm * BM + wm * 128 + g0 * 32) / (32 * IPG);
      if (half == 0) out[grp * p.ldc + col] = sum * inv_n;
    }
  }
}

// dgrad epilogue fused with the BatchNorm+ELU backward of the layer BELOW (pcaa_gemm_dgrad_bn):
// the tile of da = dy.Wt never reaches HBM as such -- on its way out (row-contiguous, after the LDS
// transpose) each lane loads the same 16 B of that layer's stored pre-activation y and writes
//   dz = da * ELU'(y*scale + shift)
// while accumulating the column sums {dz, dz * (y-mean)*rstd} the BatchNorm backward needs.  That
// replaces a separate pass that re-read da and y (0.31 ms per step for PointNet layers 2-3).
__device__ __forceinline__ void unpack8(const uint4 v, float (&f)[8]) {
  f[0] = __uint_as_float(v.x << 16); f[1] = __uint_as_float(v.x & 0xffff0000u);
  f[2] = __uint_as_float(v.y << 16); f[3] = __uint_as_float(v.y & 0xffff0000u);
  f[4] = __uint_as_float(v.z << 16); f[5] = __uint_as_float(v.z & 0xffff0000u);
  f[6] = __uint_as_float(v.w << 16); f[7] = __uint_as_float(v.w & 0xffff0000u);
}

// POINTS: the layer below is the first PointNet layer on its recompute path -- its pre-activation was
// never stored, y[row][col] = sum_c x[row][c] * W1[col][c] (C <= 8 point features) is rebuilt here.
template <bool POINTS>
__device__ __forceinline__ void epilogue_dgrad_bn(const GemmParams& p, f32x16 (&acc)[FM][FN], bf16_t* smem,
                                                  int tm, int tn, int tid) {
  const int lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  constexpr int PITCH = 64;
  uint32_t* w32 = reinterpret_cast<uint32_t*>(smem + wave * 128 * PITCH);
  const bool odd = lane & 1;
#pragma unroll
  for (int j = 0; j < FN; ++j) {
    const int colw = (j * 32 + (l31 & ~1)) >> 1;
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int r = 0; r < 16; r += 2) {
        const float va = acc[i][j][r], vb = acc[i][j][r + 1];
        const float got = dpp_swap_neighbour(odd ? va : vb);
        const uint32_t packed = odd ? pack2(got, vb) : pack2(va, got);
        const int row = i * 32 + (r & 3) + 8 * (r >> 2) + 4 * half + (odd ? 1 : 0);
        w32[row * (PITCH / 2) + colw] = packed;
      }
  }
  __syncthreads();
  const int cg = (lane & 7) * 8, r0 = lane >> 3;
  const long tile_off = (long)(tm * BM + wm * 128) * p.ldc + tn * BN + wn * 64;
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C) + tile_off;
  const bf16_t* Y = reinterpret_cast<const bf16_t*>(p.ep_y) + tile_off;       // same shape and ld as C
  const bf16_t* w = smem + wave * 128 * PITCH;
  const int gcol = tn * BN + wn * 64 + cg;
  float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
  for (int q = 0; q < 2; ++q) {
    const f32x4 a = load4(p.ep_scale + gcol + 4 * q), b = load4(p.ep_shift + gcol + 4 * q);
    const f32x4 c = load4(p.ep_mean + gcol + 4 * q), d = load4(p.ep_rstd + gcol + 4 * q);
#pragma unroll
    for (int e = 0; e < 4; ++e) { sc[4 * q + e] = a[e]; sh[4 * q + e] = b[e]; mu[4 * q + e] = c[e]; rs[4 * q + e] = d[e]; }
  }
  uint4 yv[16];
  f32x4 xv[16][2];
  float w1[8][8];
  const int xc = p.ep_xc;
  if (POINTS) {
    const float* X = p.ep_x + (long)(tm * BM + wm * 128) * xc;
#pragma unroll
    for (int pass = 0; pass < 16; ++pass) {
      const float* xr = X + (long)(pass * 8 + r0) * xc;
      if (xc == 4) { xv[pass][0] = load4(xr); xv[pass][1] = f32x4{0.f, 0.f, 0.f, 0.f}; }
      else {
#pragma unroll
        for (int c = 0; c < 8; ++c) xv[pass][c >> 2][c & 3] = c < xc ? xr[c] : 0.f;
      }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j)
#pragma unroll
      for (int c = 0; c < 8; ++c) w1[j][c] = c < xc ? p.ep_w1[(long)(gcol + j) * xc + c] : 0.f;
  } else {
#pragma unroll
    for (int pass = 0; pass < 16; ++pass)
      yv[pass] = *reinterpret_cast<const uint4*>(Y + (long)(pass * 8 + r0) * p.ldc + cg);
  }
  float s1[8], s2[8];
#pragma unroll
  for (int c = 0; c < 8; ++c) { s1[c] = 0.f; s2[c] = 0.f; }
#pragma unroll
  for (int pass = 0; pass < 16; ++pass) {
    const int row = pass * 8 + r0;
    float da[8], yy[8], dz[8];
    unpack8(*reinterpret_cast<const uint4*>(&w[row * PITCH + cg]), da);
    if (POINTS) {
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        float a = 0.f;
#pragma unroll
        for (int c = 0; c < 8; ++c) a = fmaf(w1[j][c], xv[pass][c >> 2][c & 3], a);   // same order as pointnet_in.hip
        yy[j] = a;
      }
    } else {
      unpack8(yv[pass], yy);
    }
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      const float z = yy[c] * sc[c] + sh[c];
      dz[c] = da[c] * (z > 0.f ? 1.f : __expf(z));
      s1[c] += dz[c];
      s2[c] += dz[c] * ((yy[c] - mu[c]) * rs[c]);
    }
    uint4 o;
    o.x = pack2(dz[0], dz[1]); o.y = pack2(dz[2], dz[3]); o.z = pack2(dz[4], dz[5]); o.w = pack2(dz[6], dz[7]);
    *reinterpret_cast<uint4*>(C + (long)row * p.ldc + cg) = o;
  }
  // lanes with equal (lane & 7) hold the same 8 columns: fold the 8 row lanes
#pragma unroll
  for (int c = 0; c < 8; ++c) {
#pragma unroll
    for (int o = 8; o < 64; o <<= 1) {
      s1[c] += __shfl_xor(s1[c], o, 64);
      s2[c] += __shfl_xor(s2[c], o, 64);
    }
  }
  __syncthreads();                       // every wave is done with its LDS image
  float* red = reinterpret_cast<float*>(smem);      // [2 stats][2 wm][256 cols]
  if (lane < 8) {
#pragma unroll
    for (int c = 0; c < 8; ++c) {
      red[(0 * 2 + wm) * 256 + wn * 64 + cg + c] = s1[c];
      red[(1 * 2 + wm) * 256 + wn * 64 + cg + c] = s2[c];
    }
  }
  __syncthreads();
  const int stat = tid >> 8, col = tid & 255;
  const double v = (double)red[(stat * 2 + 0) * 256 + col] + (double)red[(stat * 2 + 1) * 256 + col];
  unsafeAtomicAdd(&p.colstats[((long)(tm % p.nrep) * 2 + stat) * p.N + tn * BN + col], v);
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

// MODE 1 / 2: timing-only address patterns (WRONG data): 1 = no source swizzle, 2 = each piece one
// contiguous 1-KB run
template <int LAY, int MODE = 0>
__device__ __forceinline__ void dma_tile(const bf16_t* __restrict__ base, long ld, int row0, int R, int k0,
                                         bf16_t* s_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int p = wave * 4 + j;       // 1-KB piece index, 32 per tile
    const bf16_t* src;
    if (LAY == KC) {
      const int r = 8 * p + (lane >> 3);
      const int g = MODE == 1 ? (lane & 7) : ((lane & 7) ^ ((r >> 1) & 7));
      if (MODE == 2) src = base + (long)min(row0 + p, R - 2) * ld + (k0 & 511) + 8 * lane;
      else src = base + (long)min(row0 + r, R - 1) * ld + k0 + 8 * g;
    } else {
      const int k = 2 * p + (lane >> 5);
      const int c = (lane & 31) ^ (4 * (k & 3));
      src = base + (long)(k0 + k) * ld + row0 + 8 * c;
    }
    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)src,
                                     (__attribute__((address_space(3))) void*)(s_tile + p * 512), 16, 0, 0);
  }
}

// one of the 4 pieces a wave moves per tile (dma_tile = pieces 0..3)
template <int LAY>
__device__ __forceinline__ void dma_piece(const bf16_t* __restrict__ base, long ld, int row0, int R, int k0,
                                          bf16_t* s_tile, int wave, int lane, int j) {
  const int p = wave * 4 + j;
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

// DIAG 20: correct results + s_memtime / s_memrealtime stamps of the workgroup's phases (wave 0
// only) into a buffer of its own, read back with pcaa_debug_gemm_stamps (tools/gemm_l2.py --stamps)
constexpr int STAMP_SLOTS = 12, STAMP_WGS = 8192;   // 8..11: per-step sums (issue, MFMA, vmcnt wait, barrier)
__device__ unsigned long long g_gemm_stamps[STAMP_WGS * STAMP_SLOTS];
#define GEMM_STAMP(slot)                                                                         \
  do {                                                                                           \
    if (DIAG == 20 && tid == 0 && blockIdx.x < STAMP_WGS)                                        \
      g_gemm_stamps[blockIdx.x * STAMP_SLOTS + (slot)] =                                         \
          (slot) == 7 ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime();         \
  } while (0)

// DIAG != 0: timing-only builds (WRONG results) used to attribute the loop time:
//   1 no DMA inside the loop, 2 no MFMA, 3 no epilogue.  Selected with PCAA_GEMM_DIAG.
template <typename TC, int ALAY, int BLAY, int DIAG = 0>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_dma_kernel(GemmParams p) {
  // DIAG 23: front-loaded DMA issue; it stays the order of the RC x RC (wgrad) instantiation, where the
  // interleaved order measured 0-8 % slower (transpose reads: two ds_read_b64_tr_b16 per fragment)
  constexpr bool kFine = ALAY == KC && (DIAG == 0 || DIAG == 20 || DIAG == 21 || DIAG == 30 || DIAG == 31 || DIAG == 40 || DIAG == 41 || DIAG == 42 || DIAG == 44);
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  GEMM_STAMP(0);      // workgroup start
  GEMM_STAMP(7);      // wall clock (100 MHz) of the same instant
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
  GEMM_STAMP(1);      // first stage issued
  __syncthreads();
  GEMM_STAMP(4);      // first stage landed (prologue over); slot order: 0 1 4 2 3 5 6
  unsigned long long loop_sum[4] = {0, 0, 0, 0};

  if (DIAG >= 11 && DIAG <= 13) {
    // timing only: the DMA stream alone (no MFMA, no LDS reads), three source-address patterns
    constexpr int MODE = DIAG - 11;
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt) {
        bf16_t* nA = smem + ((t + 1) & 1) * 2 * D_TILE;
        const int k0 = kbeg + (t + 1) * BK;
        dma_tile<ALAY, MODE>(A, p.lda, tm * BM, p.M, k0, nA, wave, lane);
        dma_tile<BLAY, MODE>(B, p.ldb, tn * BN, p.N, k0, nA + D_TILE, wave, lane);
      }
      __syncthreads();
    }
  } else if (DIAG == 9 || DIAG == 10) {
    // timing only: the MFMAs run on fragments read ONCE (no LDS reads in the loop); 9 keeps the DMA
    bf16x8 af0[FM], bf0[FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) af0[i] = dma_load_frag<ALAY>(smem, offA[i], 0, kofs);
#pragma unroll
    for (int j = 0; j < FN; ++j) bf0[j] = dma_load_frag<BLAY>(smem + D_TILE, offB[j], 0, kofs);
    for (int t = 0; t < nt; ++t) {
      if (t + 1 < nt && DIAG == 9) {
        bf16_t* nA = smem + ((t + 1) & 1) * 2 * D_TILE;
        const int k0 = kbeg + (t + 1) * BK;
        dma_tile<ALAY>(A, p.lda, tm * BM, p.M, k0, nA, wave, lane);
        dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, k0, nA + D_TILE, wave, lane);
      }
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af0[i], bf0[j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      __syncthreads();
    }
  } else
  for (int t = 0; t < nt; ++t) {
    const bf16_t* sA = smem + (t & 1) * 2 * D_TILE;
    const bf16_t* sB = sA + D_TILE;
    unsigned long long ts0 = 0, ts1 = 0, ts2 = 0, ts3 = 0;
    if (DIAG == 20) ts0 = __builtin_amdgcn_s_memtime();
    if (t + 1 < nt && DIAG != 1 && !kFine && DIAG != 22) {
      bf16_t* nA = smem + ((t + 1) & 1) * 2 * D_TILE;
      const int k0 = kbeg + (t + 1) * BK;
      dma_tile<ALAY>(A, p.lda, tm * BM, p.M, k0, nA, wave, lane);
      dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, k0, nA + D_TILE, wave, lane);
    }
    if (DIAG == 20) ts1 = __builtin_amdgcn_s_memtime();
    // software-pipelined fragment reads: the 6 ds_reads of k-step ks+1 are issued before the 8
    // MFMAs of k-step ks, so only the first read group of a stage exposes LDS latency (the
    // compiler's own schedule was "read; s_waitcnt lgkmcnt(0); mfma" per group)
    bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) af[0][i] = dma_load_frag<ALAY>(sA, offA[i], 0, kofs);
#pragma unroll
    for (int j = 0; j < FN; ++j) bfr[0][j] = dma_load_frag<BLAY>(sB, offB[j], 0, kofs);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[nxt][i] = dma_load_frag<ALAY>(sA, offA[i], ks + 1, kofs);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[nxt][j] = dma_load_frag<BLAY>(sB, offB[j], ks + 1, kofs);
      }
      // pin: next step's reads stay ABOVE this step's MFMAs (the machine scheduler otherwise sinks
      // them to just before their use and waits lgkmcnt(0))
      __builtin_amdgcn_sched_barrier(0);
      if (DIAG == 22) {
        // loader waves: waves 0-3 (one per SIMD) issue ALL 64 pieces of the step, two in front of every
        // 4 of their MFMAs; waves 4-7 (their SIMD partners) only compute and keep the MFMA pipe busy
        // while the loaders sit in the vector-memory issue queue
        bf16_t* nA = smem + ((t + 1) & 1) * 2 * D_TILE;
        const int k0 = kbeg + (t + 1) * BK;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (t + 1 < nt && wave < 4) {
            const int slot = 2 * ks + g;                  // 0..7
            const int vw = wave + 4 * (slot >> 2);        // slots 0-3: own pieces, 4-7: the partner's
            const int j = slot & 3;
            dma_piece<ALAY>(A, p.lda, tm * BM, p.M, k0, nA, vw, lane, j);
            dma_piece<BLAY>(B, p.ldb, tn * BN, p.N, k0, nA + D_TILE, vw, lane, j);
          }
#pragma unroll
          for (int i = 2 * g; i < 2 * g + 2; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (kFine) {
        // fine interleave (default): one DMA piece in front of every 4 MFMAs, 8 pieces over the 32 MFMAs of
        // a step.  Per-step stamps of the front-loaded order (DIAG 23): the wave spent ~700 cycles issuing
        // its 8 pieces, ~1370 issuing MFMAs, ~100 waiting for vmcnt(0) and ~1400 at the barrier waiting for
        // the waves that got through the CU's vector-memory queue last.  Interleaved, the queueing overlaps
        // the other waves' MFMAs: +4-5 % on all three PointNet shapes; dedicated loader waves (DIAG 22) -5 %.
        bf16_t* nA = smem + ((t + 1) & 1) * 2 * D_TILE;
        const int k0 = kbeg + (t + 1) * BK;
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          if (t + 1 < nt) {
            const int pc = 2 * ks + g;
            if (pc < 4) dma_piece<ALAY>(A, p.lda, tm * BM, p.M, k0, nA, wave, lane, pc);
            else dma_piece<BLAY>(B, p.ldb, tn * BN, p.N, k0, nA + D_TILE, wave, lane, pc - 4);
          }
#pragma unroll
          for (int i = 2 * g; i < 2 * g + 2; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else if (DIAG != 2) {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i) asm volatile("" ::"v"(af[cur][i]));
#pragma unroll
        for (int j = 0; j < FN; ++j) asm volatile("" ::"v"(bfr[cur][j]));
      }
      __builtin_amdgcn_sched_barrier(0);
    }
    if (DIAG == 20) {
      // MFMA results are not waited for here: ts2 is "all MFMAs ISSUED"; the s_nop keeps the stamp after them
      ts2 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      ts3 = __builtin_amdgcn_s_memtime();
      asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
      const unsigned long long ts4 = __builtin_amdgcn_s_memtime();
      loop_sum[0] += ts1 - ts0; loop_sum[1] += ts2 - ts1; loop_sum[2] += ts3 - ts2; loop_sum[3] += ts4 - ts3;
    } else {
      __syncthreads();
    }
  }
  if (DIAG == 20 && tid == 0 && blockIdx.x < STAMP_WGS) {
#pragma unroll
    for (int i = 0; i < 4; ++i) g_gemm_stamps[blockIdx.x * STAMP_SLOTS + 8 + i] = loop_sum[i];
  }

  if (DIAG == 3) {
    if (acc[0][0][0] == 12345.678f) reinterpret_cast<float*>(p.C)[0] = 1.f;   // keep the accumulators live
    return;
  }
  GEMM_STAMP(2);      // K loop done
  if constexpr (DIAG == 30 || DIAG == 31) {       // dgrad fused with the BatchNorm+ELU backward of the layer below
    epilogue_dgrad_bn<DIAG == 31>(p, acc, smem, tm, tn, tid);
    return;
  }
  if constexpr (DIAG == 40) {
    epilogue_full_tile<TC, true>(p, acc, smem, tm, tn, tid, split);
    return;
  }
  if constexpr (DIAG == 41 || DIAG == 42 || DIAG == 44) {
    epilogue_affine_meanpool<DIAG - 40>(p, acc, tm, tn, tid);
    return;
  }
  epilogue_full_tile<TC>(p, acc, smem, tm, tn, tid, split);
  GEMM_STAMP(3);      // C stores issued
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
  GEMM_STAMP(5);      // statistics atomics issued
  if (DIAG == 20) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    GEMM_STAMP(6);    // all stores / atomics of this wave retired
  }
}

}  // namespace

// host side of the DIAG 20 stamps: n <= STAMP_WGS * STAMP_SLOTS values, [workgroup][slot]
extern "C" int pcaa_debug_gemm_stamps(unsigned long long* host_out, int n) {
  if (!host_out || n < 1 || n > STAMP_WGS * STAMP_SLOTS) return PCAA_ERR_INVALID_ARG;
  const hipError_t e = hipMemcpyFromSymbol(host_out, HIP_SYMBOL(g_gemm_stamps), sizeof(unsigned long long) * n);
  return e == hipSuccess ? PCAA_OK : PCAA_ERR_LAUNCH;
}
namespace {

// ===========================================================================
// Persistent LDS-DMA kernel (forward / dgrad: thousands of 256x256 tiles, K only 512-1024).
// Timing-only builds of the kernel above put ~40 % of its time OUTSIDE the K loop: workgroup
// launch, the exposed latency of the first DMA of every tile, and the drain of the C stores
// before the workgroup can retire (one workgroup per CU, nothing to overlap with).  Here each
// workgroup walks a list of tiles with ONE continuous DMA stream:
//   * the first stage of tile i+1 is requested during the last K step of tile i, into the
//     stage that step is not reading;
//   * the epilogue of tile i runs out of the stage that step just consumed (64 KB, two
//     half-tile passes), entirely wave-local in LDS (a wave reads back only what it wrote, and
//     a wave's LDS operations execute in order), while that DMA and the C stores are in flight;
//   * the wait before tile i+1's first MFMA is a COUNTED vmcnt: only the 16 C stores (+1
//     statistics atomic) issued after the DMA may still be pending, so the stores keep draining
//     under the next tile's MFMAs.
// Tile order: every XCD (workgroups b, b+8, ...) walks its own contiguous range of the XCD-aware
// tile list, so an A panel is still shared through that XCD's L2.
// ===========================================================================
__device__ __forceinline__ bool persistent_tile(int nb, int nbn, int it, int& tm, int& tn) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = blockIdx.x & 7, per_xcd = gridDim.x >> 3;       // gridDim.x is a multiple of 8
  const int idx = (blockIdx.x >> 3) + it * per_xcd;
  const int count = q + (xcd < r ? 1 : 0);
  if (idx >= count) return false;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tm = v / nbn;
  tn = v - tm * nbn;
  return true;
}

template <int ALAY, int BLAY>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_dmap_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  const int nbm = p.M / BM, nbn = p.N / BN, nb = nbm * nbn;
  const int nt = p.K / BK;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  bf16_t* C = reinterpret_cast<bf16_t*>(p.C);

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

  int tm, tn, ntm = 0, ntn = 0;
  bool have = persistent_tile(nb, nbn, 0, tm, tn);
  if (!have) return;                       // (uniform: whole workgroup)
  dma_tile<ALAY>(A, p.lda, tm * BM, p.M, 0, smem, wave, lane);
  dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, 0, smem + D_TILE, wave, lane);
  __syncthreads();

  int g = 0;                               // global K-step counter: stage = g & 1
  for (int it = 0; have; ++it) {
    const bool have_next = persistent_tile(nb, nbn, it + 1, ntm, ntn);
    f32x16 acc[FM][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i)
#pragma unroll
      for (int j = 0; j < FN; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    for (int t = 0; t < nt; ++t, ++g) {
      const bf16_t* sA = smem + (g & 1) * 2 * D_TILE;
      const bf16_t* sB = sA + D_TILE;
      bf16_t* nA = smem + ((g + 1) & 1) * 2 * D_TILE;
      const bool last = (t + 1 == nt);
      if (!last) {
        dma_tile<ALAY>(A, p.lda, tm * BM, p.M, (t + 1) * BK, nA, wave, lane);
        dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, (t + 1) * BK, nA + D_TILE, wave, lane);
      } else if (have_next) {
        dma_tile<ALAY>(A, p.lda, ntm * BM, p.M, 0, nA, wave, lane);
        dma_tile<BLAY>(B, p.ldb, ntn * BN, p.N, 0, nA + D_TILE, wave, lane);
      }
      bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
      for (int i = 0; i < FM; ++i) af[0][i] = dma_load_frag<ALAY>(sA, offA[i], 0, kofs);
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[0][j] = dma_load_frag<BLAY>(sB, offB[j], 0, kofs);
#pragma unroll
      for (int ks = 0; ks < 4; ++ks) {
        const int cur = ks & 1, nxt = cur ^ 1;
        if (ks < 3) {
#pragma unroll
          for (int i = 0; i < FM; ++i) af[nxt][i] = dma_load_frag<ALAY>(sA, offA[i], ks + 1, kofs);
#pragma unroll
          for (int j = 0; j < FN; ++j) bfr[nxt][j] = dma_load_frag<BLAY>(sB, offB[j], ks + 1, kofs);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
      if (!last) {
        __syncthreads();                   // vmcnt(0) + barrier: step t+1 has landed everywhere
      } else {
        // every wave is done READING this stage; the next tile's DMA stays in flight
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
      }
    }

    // ---- epilogue out of the stage the last step consumed (g was already advanced past it)
    bf16_t* scratch = smem + ((g - 1) & 1) * 2 * D_TILE;          // 64 KB
    uint32_t* w32 = reinterpret_cast<uint32_t*>(scratch + wave * 64 * 64);
    const bf16_t* wr = scratch + wave * 64 * 64;
    const bool odd = lane & 1;
    bf16_t* Ct = C + (long)(tm * BM + wm * 128) * p.ldc + tn * BN + wn * 64;
    const int cg = (lane & 7) * 8, r0 = lane >> 3;
#pragma unroll
    for (int h2 = 0; h2 < 2; ++h2) {
      asm volatile("" ::: "memory");       // keep the previous half's LDS reads above these writes
#pragma unroll
      for (int j = 0; j < FN; ++j) {
        const int colw = (j * 32 + (l31 & ~1)) >> 1;
#pragma unroll
        for (int ii = 0; ii < 2; ++ii)
#pragma unroll
          for (int r = 0; r < 16; r += 2) {
            const int i = 2 * h2 + ii;
            // no bias here (an ordinary global load next to an in-flight LDS-DMA makes hipcc
            // drain the DMA with vmcnt(0)); the launcher only takes this kernel with bias == NULL
            const float va = acc[i][j][r], vb = acc[i][j][r + 1];
            const float got = dpp_swap_neighbour(odd ? va : vb);
            const uint32_t packed = odd ? pack2(got, vb) : pack2(va, got);
            const int row = ii * 32 + (r & 3) + 8 * (r >> 2) + 4 * half + (odd ? 1 : 0);
            w32[row * 32 + colw] = packed;
          }
      }
      asm volatile("" ::: "memory");       // LDS is in-order per wave; this pins the compiler's order too
#pragma unroll
      for (int pass = 0; pass < 8; ++pass) {
        const int row = pass * 8 + r0;
        *reinterpret_cast<uint4*>(Ct + (long)(h2 * 64 + row) * p.ldc + cg) =
            *reinterpret_cast<const uint4*>(&wr[row * 64 + cg]);
      }
    }
    // ---- BatchNorm column statistics (bias-free accumulator)
    if (p.colstats != nullptr) {
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();        // every wave has finished with its scratch area
      float* red = reinterpret_cast<float*>(scratch);
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
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
      const int stat = tid >> 8, col = tid & 255;
      const double v = (double)red[(stat * 2 + 0) * 256 + col] + (double)red[(stat * 2 + 1) * 256 + col];
      unsafeAtomicAdd(&p.colstats[((long)(tm % p.nrep) * 2 + stat) * p.N + tn * BN + col], v);
    }
    // ---- next tile: its first stage (8 DMA instructions per wave, issued before the 16 stores and
    // the atomic above) must have landed; the stores may still be pending
    if (have_next) {
      asm volatile("s_waitcnt vmcnt(16) lgkmcnt(0)" ::: "memory");
      __builtin_amdgcn_s_barrier();
    }
    tm = ntm;
    tn = ntn;
    have = have_next;
  }
}

template <int ALAY, int BLAY>
bool launch_dmap(const GemmParams& p, hipStream_t s) {
  static bool configured = false;
  static int ncu = 0;
  auto kern = gemm_bf16_dmap_kernel<ALAY, BLAY>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            D_LDS_BYTES) != hipSuccess)
      return false;
    int dev = 0;
    hipDeviceProp_t prop;
    if (hipGetDevice(&dev) != hipSuccess || hipGetDeviceProperties(&prop, dev) != hipSuccess) return false;
    ncu = prop.multiProcessorCount;
    configured = true;
  }
  const long ntiles = (long)(p.M / BM) * (p.N / BN);
  long grid = ncu - (ncu % 8);
  if (grid < 8) grid = 8;
  if (grid > ntiles) grid = ((ntiles + 7) / 8) * 8;      // surplus workgroups find no tile and exit
  hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(NTHREADS), D_LDS_BYTES, s, p);
  return true;
}

// ===========================================================================
// 4-stage LDS-DMA variant (BK = 32 per stage, same 128 KB of LDS): the DMA of
// step t+3 is issued at the top of step t, so up to three 32-KB stages (96 KB per
// CU, vs 64 KB in the 2-stage kernel) are in flight behind the MFMAs.  PMC on the
// 2-stage kernel showed 40 % (KC) / 59 % (RC) of wave time parked in s_waitcnt /
// s_barrier with zero LDS bank conflicts: the fix is prefetch depth, and that needs
// COUNTED waits -- __syncthreads() would drain every DMA (vmcnt(0)) each step.  Each
// wave issues exactly 4 DMA instructions per step (2 pieces of A, 2 of B), so
// "all but the youngest g groups have landed" is s_waitcnt vmcnt(4*g); the wait sits
// immediately before the raw s_barrier that precedes the first read of that stage
// (RAW), and a stage is refilled only after the barrier that follows its last read
// (WAR).  No ordinary global load lives in the loop (it would force vmcnt(0)).
//   KC image [256 rows][32 k] (64-B rows): granule g of row r holds k-granule g ^ ((r>>2)&3)
//   RC image [32 k][256 rows]: as in the 2-stage kernel.
// ===========================================================================
constexpr int Q_BK = 32;
constexpr int Q_STAGES = 4;
constexpr int Q_TILE = 256 * Q_BK;                          // elements per operand per stage (16 KB)
constexpr int Q_LDS_BYTES = Q_STAGES * 2 * Q_TILE * 2;      // 131072

template <int LAY>
__device__ __forceinline__ void q_dma_tile(const bf16_t* __restrict__ base, long ld, int row0, int R, int k0,
                                           bf16_t* s_tile, int wave, int lane) {
#pragma unroll
  for (int j = 0; j < 2; ++j) {
    const int p = wave * 2 + j;       // 1-KB piece index, 16 per tile
    const bf16_t* src;
    if (LAY == KC) {
      const int r = 16 * p + (lane >> 2);
      const int g = (lane & 3) ^ ((r >> 2) & 3);
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

template <int LAY>
__device__ __forceinline__ int q_frag_offset(int row_base, int lane) {
  if (LAY == KC) return (row_base + (lane & 31)) * Q_BK;
  const int j = lane & 15, mb = 16 * ((lane >> 4) & 1), h = lane >> 5;
  const int q = j >> 2, ch = row_base + mb + 4 * (j & 3);
  return (8 * h + q) * 256 + ((((ch >> 3) ^ (4 * q)) << 3) | (ch & 7));
}

template <int LAY>
__device__ __forceinline__ bf16x8 q_load_frag(const bf16_t* s, int off, int kstep, const int (&kofs)[2]) {
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
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_dma4_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 2, wn = wave & 3, l31 = lane & 31, half = lane >> 5;
  int tm, tn;
  const int split = block_coords(p, p.M / BM, p.N / BN, tm, tn);

  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = (kend - kbeg) / Q_BK;
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);

  f32x16 acc[FM][FN];
#pragma unroll
  for (int i = 0; i < FM; ++i)
#pragma unroll
    for (int j = 0; j < FN; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

  int offA[FM], offB[FN], kofs[2];
#pragma unroll
  for (int i = 0; i < FM; ++i) offA[i] = q_frag_offset<ALAY>(wm * 128 + i * 32, lane);
#pragma unroll
  for (int j = 0; j < FN; ++j) offB[j] = q_frag_offset<BLAY>(wn * 64 + j * 32, lane);
  {
    const int swz = (l31 >> 2) & 3;
#pragma unroll
    for (int s = 0; s < 2; ++s) kofs[s] = ((2 * s + half) ^ swz) * 8;
  }

  // prologue: stages 0..2 <- steps 0..2 (as many as exist); step 0 must have landed before the loop
  const int pre = min(nt, Q_STAGES - 1);
  for (int t = 0; t < pre; ++t) {
    bf16_t* st = smem + t * 2 * Q_TILE;
    q_dma_tile<ALAY>(A, p.lda, tm * BM, p.M, kbeg + t * Q_BK, st, wave, lane);
    q_dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, kbeg + t * Q_BK, st + Q_TILE, wave, lane);
  }
  if (pre >= 3) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
  else if (pre == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __builtin_amdgcn_s_barrier();

  // Ping-pong: the DMA instructions are expensive to ISSUE (~100 cycles each while the wave can do
  // nothing else), and with all 8 waves in lockstep both waves of a SIMD issued them at the same
  // time while the MFMA pipe idled.  The two wave groups (waves 0-3 / 4-7: wave w and w+4 share a
  // SIMD) now run half a step apart: in every phase one group issues its 4 DMA instructions for
  // step t+3 while the other runs its 16 MFMAs alone on the pipe; a raw barrier ends each phase.
  //   phase 2t  : group 1 computes step t      | group 0 issues DMA(t+3)
  //   phase 2t+1: group 1 issues DMA(t+3), then both wait "step t+1 landed" | group 0 computes step t
  // WAR: stage (t+3)&3 was last read in step t-1 (group 0: phase 2t-1, group 1: phase 2t-2).
  // RAW: every wave's counted vmcnt + the barrier ending phase 2t+1 precede the first read of
  // step t+1 (group 1, phase 2t+2).
  const int grp = __builtin_amdgcn_readfirstlane(wave >> 2);
  auto compute = [&](int t) {
    const bf16_t* sA = smem + (t & (Q_STAGES - 1)) * 2 * Q_TILE;
    const bf16_t* sB = sA + Q_TILE;
    bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
#pragma unroll
      for (int i = 0; i < FM; ++i) af[ks][i] = q_load_frag<ALAY>(sA, offA[i], ks, kofs);
#pragma unroll
      for (int j = 0; j < FN; ++j) bfr[ks][j] = q_load_frag<BLAY>(sB, offB[j], ks, kofs);
    }
    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int i = 0; i < FM; ++i)
#pragma unroll
        for (int j = 0; j < FN; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
  };
  auto refill = [&](int t) {
    if (t + 3 < nt) {
      bf16_t* st = smem + ((t + 3) & (Q_STAGES - 1)) * 2 * Q_TILE;
      const int k0 = kbeg + (t + 3) * Q_BK;
      q_dma_tile<ALAY>(A, p.lda, tm * BM, p.M, k0, st, wave, lane);
      q_dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, k0, st + Q_TILE, wave, lane);
    }
  };
  auto wait_next = [&](int t) {
    // this wave's groups younger than step t+1's: steps t+2, t+3 where they exist
    const int ahead = min(nt - 1, t + 3) - (t + 1);
    if (ahead >= 2) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
    else if (ahead == 1) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  };
  if (grp == 1) {
    for (int t = 0; t < nt; ++t) {
      compute(t);
      __builtin_amdgcn_s_barrier();      // end of phase 2t
      refill(t);
      wait_next(t);
      __builtin_amdgcn_s_barrier();      // end of phase 2t+1
    }
  } else {
    for (int t = 0; t < nt; ++t) {
      refill(t);
      __builtin_amdgcn_s_barrier();      // end of phase 2t
      compute(t);
      wait_next(t);
      __builtin_amdgcn_s_barrier();      // end of phase 2t+1
    }
  }
  // the fragment reads of the last step were consumed by its MFMAs before the final barrier;
  // make the epilogue's LDS reuse safe against any straggling LDS traffic
  __syncthreads();

  epilogue_full_tile<TC>(p, acc, smem, tm, tn, tid, split);
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

// ===========================================================================
// 5-tile ring (all 160 KB of LDS): A double-buffered, B TRIPLE-buffered.
// Phase stamps of the 2-stage kernel (PCAA_GEMM_DIAG=20) put a 64-deep step at ~3730 shader cycles
// against 2048 of MFMA: the CU's vector-memory path moves the step's 64 KB of LDS-DMA in ~2750
// cycles (~24 B/clk, DMA-only build), and in the 2-stage ring it sits idle from "stage landed" to
// the next issue after the barrier.  Here step t issues A(t+1) and then B(t+2): the wait at the
// end of the step is s_waitcnt vmcnt(4) -- everything but the four youngest pieces, B(t+2) -- so
// there is always half a stage queued behind the data the next step needs and the memory path never
// drains.  Same fragment reads, MFMA schedule and epilogue as the 2-stage kernel; raw s_barrier
// (__syncthreads would wait vmcnt(0)).
// ===========================================================================
constexpr int R5_LDS_BYTES = 5 * D_TILE * 2;           // 163840

__device__ __forceinline__ void r5_wait_barrier(bool keep_b_in_flight) {
  if (keep_b_in_flight) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
  else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

template <typename TC, int ALAY, int BLAY>
__global__ __launch_bounds__(NTHREADS) void gemm_bf16_r5_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  bf16_t* sAbuf = smem;                      // 2 tiles
  bf16_t* sBbuf = smem + 2 * D_TILE;         // 3 tiles

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

  // prologue: A(0), B(0), then B(1) which may stay in flight
  if (nt > 0) {
    dma_tile<ALAY>(A, p.lda, tm * BM, p.M, kbeg, sAbuf, wave, lane);
    dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, kbeg, sBbuf, wave, lane);
    if (nt > 1) dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, kbeg + BK, sBbuf + D_TILE, wave, lane);
  }
  r5_wait_barrier(nt > 1);

  int bcur = 0;                                // B buffer of step t (t % 3 without the division)
  for (int t = 0; t < nt; ++t) {
    const bf16_t* sA = sAbuf + (t & 1) * D_TILE;
    const bf16_t* sB = sBbuf + bcur * D_TILE;
    const int bnext2 = bcur == 0 ? 2 : bcur - 1;            // (t + 2) % 3
    constexpr bool kFine = ALAY == KC;          // as in the 2-stage kernel: interleave the pieces with the MFMAs
    if (!kFine) {
      if (t + 1 < nt) dma_tile<ALAY>(A, p.lda, tm * BM, p.M, kbeg + (t + 1) * BK, sAbuf + ((t + 1) & 1) * D_TILE, wave, lane);
      if (t + 2 < nt) dma_tile<BLAY>(B, p.ldb, tn * BN, p.N, kbeg + (t + 2) * BK, sBbuf + bnext2 * D_TILE, wave, lane);
    }
    bf16x8 af[2][FM], bfr[2][FN];
#pragma unroll
    for (int i = 0; i < FM; ++i) af[0][i] = dma_load_frag<ALAY>(sA, offA[i], 0, kofs);
#pragma unroll
    for (int j = 0; j < FN; ++j) bfr[0][j] = dma_load_frag<BLAY>(sB, offB[j], 0, kofs);
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
      const int cur = ks & 1, nxt = cur ^ 1;
      if (ks < 3) {
#pragma unroll
        for (int i = 0; i < FM; ++i) af[nxt][i] = dma_load_frag<ALAY>(sA, offA[i], ks + 1, kofs);
#pragma unroll
        for (int j = 0; j < FN; ++j) bfr[nxt][j] = dma_load_frag<BLAY>(sB, offB[j], ks + 1, kofs);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (kFine) {
#pragma unroll
        for (int g = 0; g < 2; ++g) {
          const int pc = 2 * ks + g;                    // pieces 0-3: A(t+1), 4-7: B(t+2) -- B stays youngest
          if (pc < 4) {
            if (t + 1 < nt) dma_piece<ALAY>(A, p.lda, tm * BM, p.M, kbeg + (t + 1) * BK, sAbuf + ((t + 1) & 1) * D_TILE, wave, lane, pc);
          } else {
            if (t + 2 < nt) dma_piece<BLAY>(B, p.ldb, tn * BN, p.N, kbeg + (t + 2) * BK, sBbuf + bnext2 * D_TILE, wave, lane, pc - 4);
          }
#pragma unroll
          for (int i = 2 * g; i < 2 * g + 2; ++i)
#pragma unroll
            for (int j = 0; j < FN; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
          __builtin_amdgcn_sched_barrier(0);
        }
      } else {
#pragma unroll
        for (int i = 0; i < FM; ++i)
#pragma unroll
          for (int j = 0; j < FN; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(af[cur][i], bfr[cur][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_sched_barrier(0);
      }
    }
    // A(t+1) and B(t+1) must have landed; B(t+2) (the 4 youngest pieces of this wave) may not
    r5_wait_barrier(t + 2 < nt);
    bcur = bcur == 2 ? 0 : bcur + 1;
  }

  epilogue_full_tile<TC>(p, acc, smem, tm, tn, tid, split);
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
bool launch_r5(const GemmParams& p, dim3 grid, hipStream_t s) {
  static bool configured = false;
  auto kern = gemm_bf16_r5_kernel<TC, ALAY, BLAY>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            R5_LDS_BYTES) != hipSuccess)
      return false;
    configured = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), R5_LDS_BYTES, s, p);
  return true;
}

template <typename TC, int ALAY, int BLAY>
bool launch_dma4(const GemmParams& p, dim3 grid, hipStream_t s) {
  static bool configured = false;
  auto kern = gemm_bf16_dma4_kernel<TC, ALAY, BLAY>;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                            Q_LDS_BYTES) != hipSuccess)
      return false;
    configured = true;
  }
  hipLaunchKernelGGL(kern, grid, dim3(NTHREADS), Q_LDS_BYTES, s, p);
  return true;
}

template <typename TC, int ALAY, int BLAY, int DIAG = 0>
bool launch_dma(const GemmParams& p, dim3 grid, hipStream_t s) {
  static bool configured = false;
  auto kern = gemm_bf16_dma_kernel<TC, ALAY, BLAY, DIAG>;
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
  if ((p.M % BM) || (p.N % BN) || (p.K % BK)) return false;
  p.nsplit = 1;
  p.split_fast = 0;
  p.diag = 0;
  p.k_per_split = p.K;
  p.atomic = 0;
  p.c_split_stride = 0;
  const long ntiles = (long)(p.M / BM) * (p.N / BN);
  if (ntiles >= (1L << 31)) return false;
  if (p.ep_y == nullptr) return launch_dma<bf16_t, KC, KC, 31>(p, dim3((unsigned)ntiles, 1, 1), stream);
  return launch_dma<bf16_t, KC, KC, 30>(p, dim3((unsigned)ntiles, 1, 1), stream);
}

// product + eval-mode BatchNorm + ELU (pcaa_gemm_affine_elu): whole 256x256 tiles, bf16 KC x KC -> bf16
bool pcaa_launch_gemm_affine_elu(const GemmParams& p_in, hipStream_t stream) {
  GemmParams p = p_in;
  if ((p.M % BM) || (p.N % BN) || (p.K % BK)) return false;
  p.nsplit = 1;
  p.split_fast = 0;
  p.diag = 0;
  p.k_per_split = p.K;
  p.atomic = 0;
  p.c_split_stride = 0;
  p.colstats = nullptr;
  const long ntiles = (long)(p.M / BM) * (p.N / BN);
  if (ntiles >= (1L << 31)) return false;
  // p.ep_xc: rows per mean-pool group (0: plain activation output)
  switch (p.ep_xc) {
    case 0: return launch_dma<bf16_t, KC, KC, 40>(p, dim3((unsigned)ntiles, 1, 1), stream);
    case 32: return launch_dma<float, KC, KC, 41>(p, dim3((unsigned)ntiles, 1, 1), stream);
    case 64: return launch_dma<float, KC, KC, 42>(p, dim3((unsigned)ntiles, 1, 1), stream);
    case 128: return launch_dma<float, KC, KC, 44>(p, dim3((unsigned)ntiles, 1, 1), stream);
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
  // dW[1024,1024], time -5..-8 % on the three wgrad shapes (tools/wgrad_exp.sh).  With the atomic
  // epilogue at 128 workgroups it had measured slower (0.53 -> 1.36 ms), so it stays off there.
  static const bool split_fast_off = getenv("PCAA_GEMM_SPLIT_FAST") && atoi(getenv("PCAA_GEMM_SPLIT_FAST")) == 0;
  p.split_fast = (!split_fast_off && p.c_split_stride != 0 && nsplit >= 8 && nsplit % 8 == 0 &&
                  ntiles * nsplit >= 256) ? 1 : 0;
  { const char* d = getenv("PCAA_GEMM_DIAG"); p.diag = d ? atoi(d) : 0; }
  dim3 grid((unsigned)ntiles, 1, (unsigned)nsplit);
  if (p.split_fast) grid = dim3((unsigned)(ntiles * nsplit), 1, 1);
  const bool af = a_dtype == PCAA_F32, bf = b_dtype == PCAA_F32, cf = c_dtype == PCAA_F32;
  // LDS-DMA kernel: bf16 x bf16, whole tiles only
  if (!af && !bf && (p.M % BM) == 0 && (p.N % BN) == 0 && (p.K % BK) == 0 && (p.k_per_split % BK) == 0 &&
      a_layout == b_layout) {
    // the 4-stage / BK=32 / counted-vmcnt kernel measured 5-17 % SLOWER than the 2-stage one
    // (twice the barriers per K outweigh the deeper prefetch); kept behind PCAA_GEMM_DMA4 for A/B
    static const bool two_stage = getenv("PCAA_GEMM_DMA4") == nullptr;   // default: 2-stage kernel
    if (two_stage && p.diag) {
      if (a_layout == KC && !cf) {
        if (p.diag == 1) return launch_dma<bf16_t, KC, KC, 1>(p, grid, stream);
        if (p.diag == 2) return launch_dma<bf16_t, KC, KC, 2>(p, grid, stream);
        if (p.diag == 9) return launch_dma<bf16_t, KC, KC, 9>(p, grid, stream);
        if (p.diag == 10) return launch_dma<bf16_t, KC, KC, 10>(p, grid, stream);
        if (p.diag == 11) return launch_dma<bf16_t, KC, KC, 11>(p, grid, stream);
        if (p.diag == 12) return launch_dma<bf16_t, KC, KC, 12>(p, grid, stream);
        if (p.diag == 13) return launch_dma<bf16_t, KC, KC, 13>(p, grid, stream);
        if (p.diag == 20) return launch_dma<bf16_t, KC, KC, 20>(p, grid, stream);   // correct results + phase stamps
        if (p.diag == 21) return launch_dma<bf16_t, KC, KC, 21>(p, grid, stream);   // correct results, fine DMA interleave
        if (p.diag == 22) return launch_dma<bf16_t, KC, KC, 22>(p, grid, stream);   // correct results, loader waves
        if (p.diag == 23) return launch_dma<bf16_t, KC, KC, 23>(p, grid, stream);   // correct results, front-loaded issue
        return launch_dma<bf16_t, KC, KC, 3>(p, grid, stream);
      }
      if (a_layout == RC && cf) {
        if (p.diag == 1) return launch_dma<float, RC, RC, 1>(p, grid, stream);
        if (p.diag == 2) return launch_dma<float, RC, RC, 2>(p, grid, stream);
        if (p.diag == 23) return launch_dma<float, RC, RC, 23>(p, grid, stream);  // correct results, front-loaded issue
        if (p.diag >= 20) return launch_dma<float, RC, RC>(p, grid, stream);      // KC-only diagnostics: default kernel
        return launch_dma<float, RC, RC, 3>(p, grid, stream);
      }
    }
    // forward / dgrad (no split, bf16 out, no bias): persistent kernel
    // measured equal to the one-tile-per-workgroup kernel (0.61 vs 0.59 ms on [245760,1024]x[1024,1024]):
    // a no-MFMA timing build shows the loop is bound by L2->LDS delivery of the operand tiles
    // (~36 GB/s per CU), which persistence cannot change.  Opt-in: PCAA_GEMM_PERSISTENT=1.
    static const bool persistent = getenv("PCAA_GEMM_PERSISTENT") != nullptr;
    if (persistent && nsplit == 1 && !cf && p.bias == nullptr && !p.atomic && (p.ldc % 8) == 0 &&
        ((uintptr_t)p.C % 16) == 0) {
      if (a_layout == KC) return launch_dmap<KC, KC>(p, stream);
      return launch_dmap<RC, RC>(p, stream);
    }
    static const bool r5 = getenv("PCAA_GEMM_R5") != nullptr && atoi(getenv("PCAA_GEMM_R5")) != 0;
    if (r5) {
      bool ok = false;
      if (a_layout == KC) ok = cf ? launch_r5<float, KC, KC>(p, grid, stream) : launch_r5<bf16_t, KC, KC>(p, grid, stream);
      else if (cf) ok = launch_r5<float, RC, RC>(p, grid, stream);
      if (ok) return true;
    }
    if (two_stage) {
      if (a_layout == KC) return cf ? launch_dma<float, KC, KC>(p, grid, stream) : launch_dma<bf16_t, KC, KC>(p, grid, stream);
      if (cf) return launch_dma<float, RC, RC>(p, grid, stream);
    } else {
      if (a_layout == KC) return cf ? launch_dma4<float, KC, KC>(p, grid, stream) : launch_dma4<bf16_t, KC, KC>(p, grid, stream);
      if (cf) return launch_dma4<float, RC, RC>(p, grid, stream);
    }
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
