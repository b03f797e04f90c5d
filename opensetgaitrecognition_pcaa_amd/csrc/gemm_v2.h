// The KC x KC bf16 (and split-fp16) tile loop of round 4 -- included by gemm_bf16.hip inside its anonymous namespace.
//
// Restructured after hipBLASLt's hand-written gfx950 kernel (Custom_Cijk_Alik_Bljk_BBS_BH_..._MT256x256x64_MI16x16x1,
// disassembled from torch's TensileLibrary_BB_BB_HA_Bias_SAV_UA_..._gfx950.co; docs/LAB_LOG.md section 9 has the side-by-side):
//   * FOUR waves (2 x 2), one per SIMD, 128 x 128 accumulators each (256 accumulator registers, pinned to the
//     accumulator file): every fragment read from the LDS feeds 8 MFMAs -- 128 KB of fragment reads per K step
//     instead of the 8-wave kernel's 192 KB;
//   * the fragments of the NEXT k-half are read into a second register set under the MFMAs of the current one, so an
//     operand's stage image is dead as soon as its second half has been read: its refill for K step t + 2 is issued in
//     the MIDDLE of step t, behind a barrier per operand.  Two LDS stages (128 KB) then carry a prefetch distance of
//     TWO steps for BOTH operands (the 8-wave ring: A two steps ahead in three stages, B one step in two);
//   * the request stream does not stop at tile boundaries: the cursor runs two steps ahead of the MFMAs straight into
//     the workgroup's next tile (the 8-wave kernel requests the next tile's first stages in a burst before its epilogue);
//   * a third barrier per step (counted vmcnt: only this step's own 16 requests may still be in flight) publishes
//     stage t + 1, whose first-half fragments are then read under the second half's MFMAs;
//   * the B operand's rows are permuted on their way into the LDS so that a lane's EIGHT column blocks are eight
//     adjacent output columns: a row leaves as one 16-B store per lane (fp32: two), 16 lanes = 256 contiguous bytes --
//     half the store instructions of the 8-wave kernel's 8-B stores.
// Lab history (tools/microbench/gemm_v2.hip, profiles/r04_gemm_v2_lab.txt), same box, [245760, K] x [N, K]^T with bf16
// C stores: 512->512 0.139 ms (8-wave kernel with statistics 0.168), 512->1024 0.276 (0.315), 1024->1024 0.440 (0.490;
// hipBLASLt 0.437).
//
// MFMAs and fragment reads are inline asm: (1) left to the register allocator, a wave with 256 accumulator + 128
// fragment registers got accumulators in both register files, ~1 200 v_accvgpr copies in the loop and 165 spills;
// the "a" constraint pins them; (2) a plain LDS load whose value is consumed in the NEXT loop iteration is sunk to the
// loop latch, behind the MFMAs it was meant to hide under.  The compiler therefore does not know the fragment
// registers are pending: every use sits behind an explicit s_waitcnt lgkmcnt(0) in this file.
#pragma once

#ifndef PCAA_V2_YRING
#define PCAA_V2_YRING 2   // 16-row blocks of y kept in flight by the fused-dgrad epilogue.  4 measured the same in isolation and
                          // +0.23 ms per STEP (the extra registers spill: 228 B of scratch per lane; docs/LAB_LOG.md section 9)
#endif
namespace v2 {

// C stores with the non-temporal hint (PCAA_V2_NT_STORE=1, round 4 lab): alone the K = 512 layers gain 4-8 % (the result no
// longer evicts the operands the K loop re-reads: 512 -> 1024 0.300 -> 0.282 ms) -- and the STEP loses 0.17 ms
// (5.32 -> 5.49, same box): written the normal way, the tail of C is still in the L2 / the 256 MB memory-side cache when
// the BatchNorm pass that follows reads it.  Off; profiles/r04_ab_nt_store.txt.
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));
#ifndef PCAA_V2_NT_STORE
#define PCAA_V2_NT_STORE 0
#endif
__device__ __forceinline__ void store_nt(void* dst, const uint4& o) {
  u32x4_t v = {o.x, o.y, o.z, o.w};
  if (PCAA_V2_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<u32x4_t*>(dst));
  else *reinterpret_cast<u32x4_t*>(dst) = v;
}
__device__ __forceinline__ void store_nt(float* dst, const f32x4& v) {
  if (PCAA_V2_NT_STORE) __builtin_nontemporal_store(v, reinterpret_cast<f32x4*>(dst));
  else *reinterpret_cast<f32x4*>(dst) = v;
}

constexpr int NT = 256;                       // threads: 4 waves
constexpr int OP_TILE = 256 * 64;             // elements of one operand's stage image (32 KB)
constexpr int STAGE = 2 * OP_TILE;            // A image, then B image (64 KB)
constexpr int SCRATCH_OFF = 2 * STAGE * 2;    // bytes: behind the two stages
constexpr int LDS_BYTES = SCRATCH_OFF + 8192; // statistics scratch [2][2][256] floats + ticket / flag words

template <bool F16>
__device__ __forceinline__ void mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// the first product of a tile: C = 0 as the inline constant instead of 256 v_accvgpr_write per wave and tile (PCAA_V2_ZERO_BY_MFMA)
template <bool F16>
__device__ __forceinline__ void mfma0(f32x4& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (F16) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}
#ifndef PCAA_V2_BIAS_BRANCH
#define PCAA_V2_BIAS_BRANCH 1      // lab builds: 0 = the bias add folded into the one store expression (rounds 1-4a)
#endif
#ifndef PCAA_V2_ZERO_BY_MFMA
#define PCAA_V2_ZERO_BY_MFMA 1
#endif
#ifndef PCAA_V2_FUSED_STATS
#define PCAA_V2_FUSED_STATS 1      // lab builds: 0 = C stores, then the column statistics in a second walk (rounds 4-5a)
#endif
__device__ __forceinline__ void lds_read(bf16x8& d, unsigned addr, int off) {
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(off));
}

// one 1-KB piece (8 LDS rows x 128 B) of an operand's stage image, p = 0..31.  PERM (the B operand): LDS row 16 j + c
// of a 128-row group holds operand row 8 c + j -- the fragment reads do not change (lane c of column block j reads LDS
// row 16 j + c), but block j of lane c is then output column 8 c + j: eight adjacent columns per lane.
// The A operand (PERM false) carries the piece's row offset in the per-lane (vector) offset: the buffer range check
// covers vector + immediate offsets only -- a row offset in the SCALAR offset is not checked, and the rows past M of a
// partial last row tile would be read for real (garbage in the column statistics) instead of as zeros.
template <bool PERM>
__device__ __forceinline__ void piece(buf_rsrc_t r, long ld, int row0, int koff, bf16_t* s_img, int p, const unsigned (&vo)[2]) {
  if constexpr (!PERM) {
    const unsigned voff = vo[p & 1] + (unsigned)((long)(row0 + 8 * p) * ld * 2);
    const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane(koff * 2);
    __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s_img + p * 512), 16, voff, soff, 0, 0);
    return;
  }
  const int prow = PERM ? 128 * (p >> 4) + 64 * (p & 1) + ((p >> 1) & 7) : 8 * p;
  // (wave-uniform by construction; the readfirstlane keeps the compiler from wrapping the request in a waterfall loop
  // when its divergence analysis cannot see that -- it could not for the split operands' segment offsets)
  const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(((long)(row0 + prow) * ld + koff) * 2));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s_img + p * 512), 16, vo[p & 1], soff, 0, 0);
}

typedef float f32x2 __attribute__((ext_vector_type(2)));

// the lane id, re-read from the hardware where it is needed (volatile: not hoisted).  Round 5: the tile loop keeps all 256
// vector registers busy, so a lane id kept live across it -- for the per-tile lane constants, the epilogues, the
// "thread 0" tests -- was spilled at kernel entry and reloaded from scratch once per tile (8-28 B of scratch in every
// instantiation); two mbcnt instead.
// value of lane (le ^ mask): __shfl_xor derives its index from the compiler's own lane id, which is loop-invariant, kept
// live across the tile loop and spilled there; here the index comes from the caller's freshly read lane id
__device__ __forceinline__ float xor_lane(float v, int le, int mask) {
  return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute((le ^ mask) << 2, __builtin_bit_cast(int, v)));
}
__device__ __forceinline__ int lane_id_now() {
  int l;
  asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(l));
  return l;
}

// ELU of z given z2 = z * log2(e) (the caller folds log2e into the affine): max(z2, 0) * ln2 + (clamp(exp2(z2)) - 1) --
// the [0, 1] clamp is the exp instruction's own output modifier, so exp2 of a positive z2 contributes 1 - 1 = 0 and of a
// negative one exp(z) - 1; no compare / select pair and no separate multiply by log2e per element (the eval epilogues
// spent ~1 000 of their 1 800 vector instructions per wave and tile on those)
__device__ __forceinline__ float elu_from_z2(float z2) {
  const float e = __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(z2), 0.f), 1.f);
  return fmaf(__builtin_fmaxf(z2, 0.f), 0.6931471805599453f, e - 1.f);
}

// ---------------------------------------------------------------------------------------------------- epilogues
// accumulator geometry: wave (wm, wn) owns rows wm * 128 .. + 127, columns wn * 128 .. + 127 of the tile; block (i, j),
// register r of lane (c = lane & 15, q = lane >> 4) is row 16 i + 4 q + r, column 8 c + j.

// C (=) acc (+ bias) [AFFINE: ELU(scale * (acc + bias) + shift), eval-mode BatchNorm + ELU]; bf16 or fp32 rows
// RAG (every epilogue): the tile hangs over the last row of a matrix whose row count is not a multiple of 256 -- the
// rows past M were requested out of the operand's buffer range (they read as zeros, so their accumulators and their
// share of the column statistics are zero) and are neither loaded from y nor stored.
// HASB: a bias is added (no BatchNorm layer of the train step has one -- BatchNorm cancels it -- and with the add folded
// into the one store expression every launch paid 256 v_add_f32 of 0.0 per wave and tile: the caller branches once)
template <typename TC, bool AFFINE, bool SC, bool RAG, bool HASB = true>
__device__ __forceinline__ void epilogue_store(const GemmParams& p, f32x4 (&acc)[8][8], int tm, int tn, int wm, int wn, int le) {
  const float os = SC ? p.out_scale : 1.f;
  const int l15 = le & 15, q = le >> 4;
  const int c0 = tn * BN + wn * 128 + 8 * l15;
  const long row0 = (long)tm * BM + wm * 128 + 4 * q;
  const int mrows = (int)min((long)BM, (long)p.M - row0);          // valid rows from this lane's first one on
  float bv[8], esc[8], esh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    bv[j] = (HASB && p.bias != nullptr) ? p.bias[c0 + j] : 0.f;
    esc[j] = AFFINE ? p.ep_scale[c0 + j] * 1.4426950408889634f : 1.f;      // (AFFINE: times log2e, see elu_from_z2)
    esh[j] = AFFINE ? p.ep_shift[c0 + j] * 1.4426950408889634f : 0.f;
  }
  auto out = [&](float v, int j) __attribute__((always_inline)) {
    if constexpr (AFFINE) {
      return elu_from_z2(fmaf(HASB ? v + bv[j] : v, esc[j], esh[j]));
    } else {
      return HASB ? (SC ? v * os : v) + bv[j] : (SC ? v * os : v);
    }
  };
  static_assert(!(AFFINE && SC), "the eval epilogues take bf16 operands");
  TC* C = reinterpret_cast<TC*>(p.C) + row0 * p.ldc + c0;
  // RAG: stores through a buffer resource that ends with row M - 1 (the hardware drops a store past it: no per-row
  // predicates; see epilogue_dgrad_bn).  The range check covers the vector offset only: the row offset goes there.
  buf_rsrc_t rC = make_rsrc(p.C, RAG ? (long)p.M * p.ldc * (long)sizeof(TC) : 0);
  const unsigned voff0 = RAG ? (unsigned)((row0 * p.ldc + c0) * (long)sizeof(TC)) : 0u;
  const unsigned rstep = RAG ? (unsigned)(p.ldc * (long)sizeof(TC)) : 0u;
  (void)mrows;
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      if constexpr (RAG) {
        const unsigned vo = voff0 + (unsigned)(i * 16 + r) * rstep;
        if constexpr (sizeof(TC) == 2) {
          const u32x4_t o = {pack2(out(acc[i][0][r], 0), out(acc[i][1][r], 1)), pack2(out(acc[i][2][r], 2), out(acc[i][3][r], 3)),
                             pack2(out(acc[i][4][r], 4), out(acc[i][5][r], 5)), pack2(out(acc[i][6][r], 6), out(acc[i][7][r], 7))};
          __builtin_amdgcn_raw_buffer_store_b128(o, rC, vo, 0, 0);
        } else {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{out(acc[i][0][r], 0), out(acc[i][1][r], 1),
                                                                                    out(acc[i][2][r], 2), out(acc[i][3][r], 3)}), rC, vo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{out(acc[i][4][r], 4), out(acc[i][5][r], 5),
                                                                                    out(acc[i][6][r], 6), out(acc[i][7][r], 7)}), rC, vo + 16u, 0, 0);
        }
      } else if constexpr (sizeof(TC) == 2) {
        uint4 o;
        o.x = pack2(out(acc[i][0][r], 0), out(acc[i][1][r], 1));
        o.y = pack2(out(acc[i][2][r], 2), out(acc[i][3][r], 3));
        o.z = pack2(out(acc[i][4][r], 4), out(acc[i][5][r], 5));
        o.w = pack2(out(acc[i][6][r], 6), out(acc[i][7][r], 7));
        store_nt(C + (long)(i * 16 + r) * p.ldc, o);
      } else {
        float* d = reinterpret_cast<float*>(C) + (long)(i * 16 + r) * p.ldc;
        store_nt(d, f32x4{out(acc[i][0][r], 0), out(acc[i][1][r], 1), out(acc[i][2][r], 2), out(acc[i][3][r], 3)});
        store_nt(d + 4, f32x4{out(acc[i][4][r], 4), out(acc[i][5][r], 5), out(acc[i][6][r], 6), out(acc[i][7][r], 7)});
      }
    }
}

// fold the four lanes (q = 0..3) that share a column, then hand the wave's 128 column sums of both statistics to the
// workgroup's scratch; after the barrier thread t adds column t of both statistics to the fp64 replicas
__device__ __forceinline__ void colstats_finish(const GemmParams& p, float (&t1)[8], float (&t2)[8], float* red, int tm, int tn,
                                                int wm, int wn, int le, int tid) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    t1[j] += xor_lane(t1[j], le, 16);
    t1[j] += xor_lane(t1[j], le, 32);
    t2[j] += xor_lane(t2[j], le, 16);
    t2[j] += xor_lane(t2[j], le, 32);
  }
  if (le < 16) {
    float* d1 = &red[(0 * 2 + wm) * 256 + wn * 128 + 8 * le];
    float* d2 = &red[(1 * 2 + wm) * 256 + wn * 128 + 8 * le];
    *reinterpret_cast<f32x4*>(d1) = f32x4{t1[0], t1[1], t1[2], t1[3]};
    *reinterpret_cast<f32x4*>(d1 + 4) = f32x4{t1[4], t1[5], t1[6], t1[7]};
    *reinterpret_cast<f32x4*>(d2) = f32x4{t2[0], t2[1], t2[2], t2[3]};
    *reinterpret_cast<f32x4*>(d2 + 4) = f32x4{t2[4], t2[5], t2[6], t2[7]};
  }
  lds_barrier();
  const long base = ((long)(tm % p.nrep) * 2) * p.N + tn * BN + tid;
  unsafeAtomicAdd(&p.colstats[base], (double)red[(0 * 2 + 0) * 256 + tid] + (double)red[(0 * 2 + 1) * 256 + tid]);
  unsafeAtomicAdd(&p.colstats[base + p.N], (double)red[(1 * 2 + 0) * 256 + tid] + (double)red[(1 * 2 + 1) * 256 + tid]);
  lds_barrier();      // the scratch is free again (the next tile's epilogue, or the finalize's flag word)
}

// Whole tiles, no bias (every BatchNorm'd product of the train step): C = acc AND the column statistics in ONE walk over
// the accumulators -- rows r, r + 1 of a row block together: the pair feeds the packed statistics (v_pk_add / v_pk_fma on
// the register pair) and two stores.  Separately (epilogue_store, then epilogue_colstats) every accumulator crosses from
// the accumulator file to the vector registers twice: 512 v_accvgpr_read per wave and tile instead of 256.
template <typename TC, bool SC>
__device__ __forceinline__ void epilogue_store_colstats(const GemmParams& p, f32x4 (&acc)[8][8], float* red, int tm, int tn, int wm,
                                                        int wn, int le, int tid) {
  const float os = SC ? p.out_scale : 1.f;
  const int l15 = le & 15, q = le >> 4;
  TC* C = reinterpret_cast<TC*>(p.C) + ((long)tm * BM + wm * 128 + 4 * q) * p.ldc + tn * BN + wn * 128 + 8 * l15;
  f32x2 a1[8], a2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) a1[j] = a2[j] = f32x2{0.f, 0.f};
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; r += 2) {
      f32x2 v[8];
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        v[j] = f32x2{acc[i][j][r], acc[i][j][r + 1]};
        a1[j] += v[j];
        a2[j] = __builtin_elementwise_fma(v[j], v[j], a2[j]);
        if constexpr (SC) v[j] *= f32x2{os, os};
      }
#pragma unroll
      for (int e = 0; e < 2; ++e) {
        TC* d = C + (long)(i * 16 + r + e) * p.ldc;
        if constexpr (sizeof(TC) == 2) {
          uint4 o;
          o.x = pack2(v[0][e], v[1][e]);
          o.y = pack2(v[2][e], v[3][e]);
          o.z = pack2(v[4][e], v[5][e]);
          o.w = pack2(v[6][e], v[7][e]);
          store_nt(d, o);
        } else {
          float* df = reinterpret_cast<float*>(d);
          store_nt(df, f32x4{v[0][e], v[1][e], v[2][e], v[3][e]});
          store_nt(df + 4, f32x4{v[4][e], v[5][e], v[6][e], v[7][e]});
        }
      }
      __builtin_amdgcn_sched_barrier(0);      // (one row pair at a time: hoisted accumulator reads of later pairs spilled the next tile's fragments)
    }
  float t1[8], t2[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    t1[j] = (a1[j].x + a1[j].y) * os;
    t2[j] = ((a2[j].x + a2[j].y) * os) * os;
  }
  colstats_finish(p, t1, t2, red, tm, tn, wm, wn, le, tid);
}

// BatchNorm column statistics (sum, sum of squares) of the bias-free accumulator
template <bool SC, bool RAG>
__device__ __forceinline__ void epilogue_colstats(const GemmParams& p, f32x4 (&acc)[8][8], float* red, int tm, int tn, int wm,
                                                  int wn, int le, int tid) {
  const float os = SC ? p.out_scale : 1.f;
  // RAG: rows past M are left out by a select -- an out-of-range LDS-DMA request writes NOTHING to the LDS (it does
  // not zero-fill), so those rows of the A image hold whatever the stage held before, possibly NaN bit patterns
  const int mrows = (int)min((long)BM, (long)p.M - ((long)tm * BM + wm * 128 + 4 * (le >> 4)));
  float t1[8], t2[8];
  if constexpr (RAG) {
    // rows outermost: one validity test per row serves the lane's eight columns (columns outermost, the 32 row
    // predicates stayed live across the whole loop nest: 360 B of scratch in the fp32-output instantiations)
    f32x2 a1[8], a2[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) a1[j] = a2[j] = f32x2{0.f, 0.f};
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int r = 0; r < 4; r += 2) {
        const bool ok0 = i * 16 + r < mrows, ok1 = i * 16 + r + 1 < mrows;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          const f32x2 v = {ok0 ? acc[i][j][r] : 0.f, ok1 ? acc[i][j][r + 1] : 0.f};
          a1[j] += v;
          a2[j] = __builtin_elementwise_fma(v, v, a2[j]);
        }
      }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      t1[j] = (a1[j].x + a1[j].y) * os;
      t2[j] = ((a2[j].x + a2[j].y) * os) * os;      // (two scalar-operand multiplies: the splat of os^2 was hoisted out of the tile loop and spilled)
    }
  } else {
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      f32x2 a1 = {0.f, 0.f}, a2 = {0.f, 0.f};
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
          const f32x2 v = {acc[i][j][r], acc[i][j][r + 1]};
          a1 += v;
          a2 = __builtin_elementwise_fma(v, v, a2);
        }
      t1[j] = (a1.x + a1.y) * os;                  // sums of (acc * os), (acc * os)^2: os is a power of two, exact
      t2[j] = ((a2.x + a2.y) * os) * os;
    }
  }
  colstats_finish(p, t1, t2, red, tm, tn, wm, wn, le, tid);
}

// dgrad fused with the first half of the BatchNorm + ELU backward of the layer below (gemm_bf16.hip, epilogue_dgrad_bn):
// dz = da * ELU'(y * scale + shift) leaves instead of da, with the column sums {dz, dz * yhat}.  TE = bf16 (bf16 mode:
// ELU' = exp2(min(z log2e, 0))) or float (split-fp16 parity mode: the separate pass's exact expression).
template <typename TE, bool SC, bool RAG>
__device__ __forceinline__ void epilogue_dgrad_bn(const GemmParams& p, f32x4 (&acc)[8][8], float* red, int tm, int tn, int wm,
                                                  int wn, int le, int tid) {
  constexpr bool kF32 = sizeof(TE) == 4;
  const float os = SC ? p.out_scale : 1.f;
  const int l15 = le & 15, q = le >> 4;
  const int c0 = tn * BN + wn * 128 + 8 * l15;
  const long row0 = (long)tm * BM + wm * 128 + 4 * q;
  TE* C = reinterpret_cast<TE*>(p.C) + row0 * p.ldc + c0;
  const TE* Y = reinterpret_cast<const TE*>(p.ep_y) + row0 * p.ldc + c0;
  const int mrows = (int)min((long)BM, (long)p.M - row0);
  f32x2 sc2[4], sh2[4], rs2[4], nm2[4], s1[4], s2[4];
  {
    constexpr float kLog2e = kF32 ? 1.f : 1.4426950408889634f;
#pragma unroll
    for (int h = 0; h < 4; ++h) {
      const float a0 = p.ep_scale[c0 + 2 * h], a1 = p.ep_scale[c0 + 2 * h + 1];
      const float b0 = p.ep_shift[c0 + 2 * h], b1 = p.ep_shift[c0 + 2 * h + 1];
      const float m0 = p.ep_mean[c0 + 2 * h], m1 = p.ep_mean[c0 + 2 * h + 1];
      const float d0 = p.ep_rstd[c0 + 2 * h], d1 = p.ep_rstd[c0 + 2 * h + 1];
      sc2[h] = f32x2{a0 * kLog2e, a1 * kLog2e};
      sh2[h] = f32x2{b0 * kLog2e, b1 * kLog2e};
      rs2[h] = f32x2{d0, d1};
      nm2[h] = f32x2{-m0 * d0, -m1 * d1};
      s1[h] = f32x2{0.f, 0.f};
      s2[h] = f32x2{0.f, 0.f};
    }
  }
  // the stored pre-activations of one 16-row block (4 rows per lane), requested one block ahead
  typedef typename std::conditional<kF32, f32x4, uint4>::type yraw_t;
  constexpr int YW = kF32 ? 2 : 1;                 // raw words of that type per row (8 columns)
  // (round 4: with one 16-row block requested ahead the epilogue paid eight HBM round trips in sequence -- 7 us of a 35 us
  // tile; now the whole tile's y is in flight at once (bf16: 32 x 16 B per lane; fp32: a ring of four blocks, three ahead))
  constexpr int RING = PCAA_V2_YRING, AHEAD = RING - 1;
  yraw_t yv[RING][4][YW];
  // Partial last row tile (RAG): y is read and dz written through buffer resources that end with row M - 1 -- the
  // hardware returns zeros for a row past M (finite: 0 * ELU'(NaN) would still be NaN in the column sums) and drops a
  // store to one; no per-row pointer selects or store predicates (which, hoisted out of the unrolled loops, cost the
  // bf16 instantiation 688 B of scratch per lane and 2.2x its time at the reference's default shape, 72 000 rows).
  // The range check covers the VECTOR offset only, so the row offset goes there.
  buf_rsrc_t rY = make_rsrc(p.ep_y, RAG ? (long)p.M * p.ldc * (long)sizeof(TE) : 0);
  buf_rsrc_t rC = make_rsrc(p.C, RAG ? (long)p.M * p.ldc * (long)sizeof(TE) : 0);
  const unsigned voff0 = RAG ? (unsigned)((row0 * p.ldc + c0) * (long)sizeof(TE)) : 0u;
  const unsigned rstep = RAG ? (unsigned)(p.ldc * (long)sizeof(TE)) : 0u;
  auto load_y = [&](int i, int b) __attribute__((always_inline)) {
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
      for (int w = 0; w < YW; ++w) {
        if constexpr (RAG) {
          const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(rY, voff0 + (unsigned)(i * 16 + r) * rstep + 16u * w, 0, 0);
          yv[b][r][w] = __builtin_bit_cast(yraw_t, raw);
        } else {
          yv[b][r][w] = *reinterpret_cast<const yraw_t*>(Y + (long)(i * 16 + r) * p.ldc + (kF32 ? 4 * w : 0));
        }
      }
  };
#pragma unroll
  for (int i = 0; i < AHEAD; ++i) load_y(i, i);
#pragma unroll
  for (int i = 0; i < 8; ++i) {
    if (i + AHEAD < 8) load_y(i + AHEAD, (i + AHEAD) % RING);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      f32x2 y2[4];
      if constexpr (kF32) {
        const f32x4 w0 = yv[i % RING][r][0], w1 = yv[i % RING][r][YW - 1];
        y2[0] = f32x2{w0.x, w0.y}; y2[1] = f32x2{w0.z, w0.w}; y2[2] = f32x2{w1.x, w1.y}; y2[3] = f32x2{w1.z, w1.w};
      } else {
        const uint4 w = yv[i % RING][r][0];
        y2[0] = f32x2{__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u)};
        y2[1] = f32x2{__uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
        y2[2] = f32x2{__uint_as_float(w.z << 16), __uint_as_float(w.z & 0xffff0000u)};
        y2[3] = f32x2{__uint_as_float(w.w << 16), __uint_as_float(w.w & 0xffff0000u)};
      }
      f32x2 dq[4];
#pragma unroll
      for (int h = 0; h < 4; ++h) {
        f32x2 dav = {acc[i][2 * h][r], acc[i][2 * h + 1][r]};
        if (RAG && i * 16 + r >= mrows) dav = f32x2{0.f, 0.f};      // (a select: the stale LDS rows may hold NaN patterns)
        if (SC) dav *= os;
        f32x2 g;
        if constexpr (kF32) {
          g = f32x2{elu_grad_from_pre(y2[h].x * sc2[h].x + sh2[h].x), elu_grad_from_pre(y2[h].y * sc2[h].y + sh2[h].y)};
        } else {
          // ELU'(z) = min(exp2(z log2e), 1): the [0, 1] clamp is the exp instruction's own output modifier (was: a v_min
          // of z against 0 in front of every exp -- 256 per wave and tile)
          const f32x2 z2 = __builtin_elementwise_fma(y2[h], sc2[h], sh2[h]);
          g = f32x2{__builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(z2.x), 0.f), 1.f),
                    __builtin_fminf(__builtin_fmaxf(__builtin_amdgcn_exp2f(z2.y), 0.f), 1.f)};
        }
        const f32x2 d2 = dav * g;
        s1[h] += d2;
        s2[h] = __builtin_elementwise_fma(d2, __builtin_elementwise_fma(y2[h], rs2[h], nm2[h]), s2[h]);
        dq[h] = d2;
      }
      if constexpr (RAG) {
        const unsigned vo = voff0 + (unsigned)(i * 16 + r) * rstep;
        if constexpr (kF32) {
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{dq[0].x, dq[0].y, dq[1].x, dq[1].y}), rC, vo, 0, 0);
          __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4_t, f32x4{dq[2].x, dq[2].y, dq[3].x, dq[3].y}), rC, vo + 16u, 0, 0);
        } else {
          const u32x4_t o = {pack2(dq[0].x, dq[0].y), pack2(dq[1].x, dq[1].y), pack2(dq[2].x, dq[2].y), pack2(dq[3].x, dq[3].y)};
          __builtin_amdgcn_raw_buffer_store_b128(o, rC, vo, 0, 0);
        }
      } else if constexpr (kF32) {
        float* d = reinterpret_cast<float*>(C) + (long)(i * 16 + r) * p.ldc;
        store_nt(d, f32x4{dq[0].x, dq[0].y, dq[1].x, dq[1].y});
        store_nt(d + 4, f32x4{dq[2].x, dq[2].y, dq[3].x, dq[3].y});
      } else {
        uint4 o;
        o.x = pack2(dq[0].x, dq[0].y);
        o.y = pack2(dq[1].x, dq[1].y);
        o.z = pack2(dq[2].x, dq[2].y);
        o.w = pack2(dq[3].x, dq[3].y);
        store_nt(C + (long)(i * 16 + r) * p.ldc, o);
      }
    }
  }
  float t1[8], t2[8];
#pragma unroll
  for (int h = 0; h < 4; ++h) {
    t1[2 * h] = s1[h].x; t1[2 * h + 1] = s1[h].y;
    t2[2 * h] = s2[h].x; t2[2 * h + 1] = s2[h].y;
  }
  colstats_finish(p, t1, t2, red, tm, tn, wm, wn, le, tid);
}

// Eval-mode LAST PointNet layer: BatchNorm (affine) + ELU + the mean over the N points of a frame, N = 32 IPG
// (gemm_bf16.hip, epilogue_affine_meanpool): a wave's 128 rows are whole groups; out fp32 [P / N, ch]
template <int IPG, bool RAG>
__device__ __forceinline__ void epilogue_affine_meanpool(const GemmParams& p, f32x4 (&acc)[8][8], int tm, int tn, int wm, int wn,
                                                         int le) {
  constexpr int BPG = 2 * IPG;                    // 16-row accumulator blocks per group
  const int l15 = le & 15;
  float* out = reinterpret_cast<float*>(p.C);
  const float inv_n = 1.f / (32 * IPG);
  const int c0 = tn * BN + wn * 128 + 8 * l15;
#pragma unroll
  for (int g0 = 0; g0 < 8; g0 += BPG) {
    float sum[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float esc = p.ep_scale[c0 + j] * 1.4426950408889634f, esh = p.ep_shift[c0 + j] * 1.4426950408889634f;
      float s = 0.f;
#pragma unroll
      for (int i = g0; i < g0 + BPG; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) s += elu_from_z2(fmaf(acc[i][j][r], esc, esh));
      s += xor_lane(s, le, 16);
      s += xor_lane(s, le, 32);
      sum[j] = s * inv_n;
    }
    const long grp = ((long)tm * BM + wm * 128 + g0 * 16) / (32 * IPG);
    if (RAG && (long)tm * BM + wm * 128 + g0 * 16 >= p.M) continue;      // (M is a whole number of groups)
    if (le < 16) {
      float* d = out + grp * p.ldc + c0;
      *reinterpret_cast<f32x4*>(d) = f32x4{sum[0], sum[1], sum[2], sum[3]};
      *reinterpret_cast<f32x4*>(d + 4) = f32x4{sum[4], sum[5], sum[6], sum[7]};
    }
  }
}

// ---------------------------------------------------------------------------------------------------- the kernel
// TC: bf16 or float output (EPI_DGRAD_BN: the type of y and dz).  SPLIT: operands are [hi | lo] fp16 images
// (pcaa_gemm_split3): the contraction walks three segments of p.seg_len at the offsets p.seg_off_a / _b.
// Tiles: p.sched == NULL: workgroup b walks the tiles b, b + gridDim.x, ... of the XCD-aware order; else the
// workgroups of an XCD draw the tiles behind their first one as tickets from that XCD's counter (see gemm_bf16.hip:
// a workgroup that gets its CU late simply does fewer tiles) -- here one tile AHEAD, because the request cursor enters
// the next tile three K steps before the MFMAs do.
// RAG: M is not a multiple of 256 -- its own instantiation (guarded epilogues on every tile; a runtime choice per tile
// inside one kernel doubled the epilogue code and pushed the fused dgrad's epilogue into 600-1000 B of scratch).
template <typename TC, int EPI, bool SPLIT, bool RAG>
__global__ __launch_bounds__(NT) void gemm_bf16_v2_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  float* red = reinterpret_cast<float*>(smem_raw + SCRATCH_OFF);            // [2 stats][2 wm][256 cols]
  int* words = reinterpret_cast<int*>(smem_raw + SCRATCH_OFF + 4096);      // [0]: next-tile hand-off, [1]: finalize flag
  const int wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbm = (p.M + BM - 1) / BM, nbn = p.N / BN, ntiles = nbm * nbn;      // (the last row tile may be partial)
  const int seg_steps = (SPLIT ? p.seg_len : p.K) / BK;                  // K steps per segment
  const int nt = SPLIT ? 3 * seg_steps : seg_steps;                       // K steps per tile
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  const long lda = p.lda, ldb = p.ldb;
  const int gstride = gridDim.x;
  int* const sched = p.sched;

  // per-lane constants, rebuilt from an opaque copy of the lane id at the top of every tile (kept live across the
  // epilogue they were spilled, and their reload in front of the loop drew a drain-everything s_waitcnt vmcnt into it)
  unsigned voA[2], voB[2];
  int kof0, kof1, fA, fB;
  auto lane_consts = [&](int ln) __attribute__((always_inline)) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int g = (ln & 7) ^ ((4 * e + (ln >> 4)) & 7);
      voA[e] = (unsigned)((ln >> 3) * lda * 2 + 16 * g);
      voB[e] = (unsigned)(8 * (ln >> 3) * ldb * 2 + 16 * g);            // PERM: the rows of a piece lie 8 apart
    }
    const int l15 = ln & 15, q = ln >> 4;
    kof0 = ((q) ^ (l15 >> 1)) * 8;                                       // k-granule (4 half + q) ^ row swizzle
    kof1 = ((4 + q) ^ (l15 >> 1)) * 8;
    fA = (wm * 128 + l15) * 64;                                          // + 1024 i per row block
    fB = OP_TILE + (wn * 128 + l15) * 64;                                // + 1024 j
  };
  lane_consts(lane_id_now());

  // request cursor: (tile, segment, k within the segment); two K steps ahead of the MFMAs, across tile boundaries.
  // The buffer resources are re-based on the cursor's tile rows, so any operand size is addressable.
  int vb = blockIdx.x;                 // tile of the MFMAs
  if (vb >= ntiles) return;
  int nvb = vb + gstride;              // the tile after it (fixed shares; tickets: drawn below)
  // (kernel arguments read once: an s_load inside the loop would share lgkmcnt with the fragment reads; the element
  // offsets of the hi / lo halves inside an image row are < 2^31)
  const int seg_len = SPLIT ? p.seg_len : p.K;
  const int soA0 = SPLIT ? (int)p.seg_off_a[0] : 0, soA1 = SPLIT ? (int)p.seg_off_a[1] : 0, soA2 = SPLIT ? (int)p.seg_off_a[2] : 0;
  const int soB0 = SPLIT ? (int)p.seg_off_b[0] : 0, soB1 = SPLIT ? (int)p.seg_off_b[1] : 0, soB2 = SPLIT ? (int)p.seg_off_b[2] : 0;
  int lvb = vb, lseg = 0, lk = 0;      // cursor
  int lkA = soA0, lkB = soB0;          // its element offset inside a row of A / B: segment offset + k
  buf_rsrc_t rA, rB;
  {
    int ltm, ltn;
    xcd_tile_coords(nbm, nbn, lvb, ltm, ltn);
    rA = make_rsrc(A + (long)ltm * BM * lda, (long)min(BM, p.M - ltm * BM) * lda * 2);      // rows past M read as zeros
    rB = make_rsrc(B + (long)ltn * BN * ldb, (long)BN * ldb * 2);
  }
  const int Mrows = p.M;
  // (macros, not lambdas: with nested by-reference closures the split instantiation kept the captured variables in a
  // stack frame and reached them through flat pointers -- 400 B of scratch traffic inside the loop)
#define V2_REQ_A(st, jj) piece<false>(rA, lda, 0, lkA, (st), wave * 8 + (jj), voA)
#define V2_REQ_B(st, jj) piece<true>(rB, ldb, 0, lkB, (st) + OP_TILE, wave * 8 + (jj), voB)
  // into the next K step; behind a tile's last step into the next tile -- past the last one the cursor wraps to this
  // workgroup's first tile: the loop stays branch-free, the two surplus stages land in dead LDS and are drained
  // before the kernel ends
#define V2_ADVANCE()                                                                       \
  do {                                                                                     \
    lk += BK;                                                                              \
    lkA += BK;                                                                             \
    lkB += BK;                                                                             \
    if (lk == seg_len) {                                                                   \
      lk = 0;                                                                              \
      if (SPLIT) lseg = lseg == 2 ? 0 : lseg + 1;                                          \
      lkA = lseg == 0 ? soA0 : (lseg == 1 ? soA1 : soA2);                                  \
      lkB = lseg == 0 ? soB0 : (lseg == 1 ? soB1 : soB2);                                  \
      if (lseg == 0) {                                                                     \
        lvb = nvb < ntiles ? nvb : (int)blockIdx.x;                                        \
        int ltm_, ltn_;                                                                    \
        xcd_tile_coords(nbm, nbn, lvb, ltm_, ltn_);                                        \
        rA = make_rsrc(A + (long)ltm_ * BM * lda, (long)min(BM, Mrows - ltm_ * BM) * lda * 2); \
        rB = make_rsrc(B + (long)ltn_ * BN * ldb, (long)BN * ldb * 2);                     \
      }                                                                                    \
    }                                                                                      \
  } while (0)

  // One K step as a macro over the MFMA form of its first half (the tile's first step starts the accumulators from the
  // constant 0: mfma0).  k-half 0: MFMAs on af[0] / bfr[0]; second-half fragments of B then A from the current stage;
  // m = 21: B image of this stage dead (barrier); m = 22 of step 1: thread 0 draws this workgroup's NEXT tile (one tile
  // ahead: the request cursor enters it three K steps before the MFMAs do) and hands it over through LDS -- request, wait
  // and hand-off in ONE asm statement, so that the compiler never sees a register whose value is still on its way (a first
  // version let the returned value "rest" in an asm output for two K steps: an instantiation that spilled that register
  // right behind the asm handed out garbage tiles -- a memory access fault in the bf16 N = 32 step; a compiler-visible
  // atomic instead dragged a spill slot and an s_waitcnt vmcnt(0) into the loop); the wait drains this wave's 16
  // requests of the previous step, at least half a step old: ~0.5 us once per tile; m = 23..37: the 8 pieces of
  // B(t + 2); m = 46: A image dead (barrier); m = 48..60: pieces 0..3 of A(t + 2).  k-half 1: MFMAs on af[1] / bfr[1];
  // m = 0..12: pieces 4..7 of A(t + 2); m = 20: stage t + 1 has landed -- everything but this step's own 16 requests
  // (loads retire in order, so "at most 16 outstanding" means the older stage is complete whatever the previous tile's
  // stores, which share the counter, are doing) -- barrier; then the next step's first-half fragments of B and A.  At
  // the end lgkmcnt(0): those fragments are in their registers; the ticket handed over in step 1 is read by every wave.
#define V2_STEP(MFMA0_FN) \
      bf16_t* cur = smem + s * STAGE; \
      const unsigned cb = lds0 + (unsigned)s * (STAGE * 2), nb = lds0 + (unsigned)(s ^ 1) * (STAGE * 2); \
      const unsigned aB1 = cb + (unsigned)(fB + kof1) * 2, aA1 = cb + (unsigned)(fA + kof1) * 2; \
      const unsigned aB0 = nb + (unsigned)(fB + kof0) * 2, aA0 = nb + (unsigned)(fA + kof0) * 2; \
_Pragma("unroll") \
      for (int m = 0; m < 64; ++m) { \
        const int i = m >> 3, j = m & 7; \
        if ((m & 1) == 1 && m < 16) lds_read(bfr[1][m >> 1], aB1, (m >> 1) * 2048); \
        if (m == 21) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        if (m == 22 && sched != nullptr && kt == 1 && wave == 0 && lane_id_now() == 0) { \
          int tk; \
          asm volatile("global_atomic_add %0, %1, %2, off sc0\n\ts_waitcnt vmcnt(0)\n\tds_write_b32 %3, %0\n\ts_waitcnt lgkmcnt(0)" \
                       : "=&v"(tk) : "v"(sched + (vb & 7)), "v"(1), "v"(lds0 + SCRATCH_OFF + 4096) : "memory"); \
        } \
        if (m >= 24 && m < 40 && (m & 1) == 0) lds_read(af[1][(m - 24) >> 1], aA1, ((m - 24) >> 1) * 2048); \
        if (m >= 23 && m < 39 && (m & 1) == 1) V2_REQ_B(cur, (m - 23) >> 1); \
        if (m == 46) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); \
        if (m >= 48 && (m & 3) == 0) V2_REQ_A(cur, (m - 48) >> 2); \
        MFMA0_FN<SPLIT>(acc[i][j], af[0][i], bfr[0][j]); \
        __builtin_amdgcn_sched_barrier(0); \
      } \
_Pragma("unroll") \
      for (int m = 0; m < 64; ++m) { \
        const int i = m >> 3, j = m & 7; \
        if (m < 16 && (m & 3) == 0) V2_REQ_A(cur, 4 + (m >> 2)); \
        if (m == 20) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory"); \
        if (m >= 22 && m < 38 && (m & 1) == 0) lds_read(bfr[0][(m - 22) >> 1], aB0, ((m - 22) >> 1) * 2048); \
        if (m >= 38 && m < 54 && (m & 1) == 0) lds_read(af[0][(m - 38) >> 1], aA0, ((m - 38) >> 1) * 2048); \
        mfma<SPLIT>(acc[i][j], af[1][i], bfr[1][j]); \
        __builtin_amdgcn_sched_barrier(0); \
      } \
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
      if (sched != nullptr && kt == 1) nvb = (vb & 7) + 8 * ((gstride >> 3) + __builtin_amdgcn_readfirstlane(words[0])); \
      V2_ADVANCE(); \
      s ^= 1;
  f32x4 acc[8][8];
  bf16x8 af[2][8], bfr[2][8];
  // prologue: stages 0 and 1 of the first tile
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) V2_REQ_B(smem, jj);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) V2_REQ_A(smem, jj);
  V2_ADVANCE();
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) V2_REQ_B(smem + STAGE, jj);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) V2_REQ_A(smem + STAGE, jj);
  V2_ADVANCE();
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int j = 0; j < 8; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(smem + fB + j * 1024 + kof0);
#pragma unroll
  for (int i = 0; i < 8; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(smem + fA + i * 1024 + kof0);

  int s = 0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  for (;;) {
    int tm, tn;
    xcd_tile_coords(nbm, nbn, vb, tm, tn);
    lane_consts(lane_id_now());
    if (!PCAA_V2_ZERO_BY_MFMA) {
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
      asm volatile("s_nop 4" ::: "memory");
    }
    if (PCAA_V2_ZERO_BY_MFMA) {
      { const int kt = 0; V2_STEP(mfma0) }
      for (int kt = 1; kt < nt; ++kt) { V2_STEP(mfma) }
    } else {
      for (int kt = 0; kt < nt; ++kt) { V2_STEP(mfma) }
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (the asm hides the hazard from the compiler)
    // (split operands: the images hold value * 2^k; the epilogues multiply by p.out_scale -- an exact power of two -- as
    // they read the accumulators: rescaling them in place would move all 256 through the vector registers and back)
    const int le = lane_id_now(), te = le + 64 * wave;
    if constexpr (EPI == EPI_DGRAD_BN) {
      epilogue_dgrad_bn<TC, SPLIT, RAG>(p, acc, red, tm, tn, wm, wn, le, te);
    } else if constexpr (EPI == EPI_AFFINE) {
      epilogue_store<TC, true, false, RAG, false>(p, acc, tm, tn, wm, wn, le);      // (pcaa_gemm_affine_elu takes no bias: the caller folds it into the shift)
    } else if constexpr (EPI == EPI_POOL1 || EPI == EPI_POOL2 || EPI == EPI_POOL4) {
      epilogue_affine_meanpool<EPI == EPI_POOL1 ? 1 : (EPI == EPI_POOL2 ? 2 : 4), RAG>(p, acc, tm, tn, wm, wn, le);
    } else {
      if constexpr (PCAA_V2_FUSED_STATS && !RAG && sizeof(TC) == 2 && !SPLIT) {
        if (p.bias == nullptr && p.colstats != nullptr) {
          epilogue_store_colstats<TC, SPLIT>(p, acc, red, tm, tn, wm, wn, le, te);
          if (nvb >= ntiles) break;
          vb = nvb;
          nvb = vb + gstride;
          continue;
        }
      }
      if constexpr (PCAA_V2_BIAS_BRANCH && (sizeof(TC) == 2 || SPLIT)) {
        if (p.bias != nullptr) epilogue_store<TC, false, SPLIT, RAG, true>(p, acc, tm, tn, wm, wn, le);
        else epilogue_store<TC, false, SPLIT, RAG, false>(p, acc, tm, tn, wm, wn, le);
      } else {
        // (fp32 results of bf16 operands: two copies of this epilogue cost the instantiation 500-700 B of scratch)
        epilogue_store<TC, false, SPLIT, RAG, true>(p, acc, tm, tn, wm, wn, le);
      }
      if (p.colstats != nullptr) epilogue_colstats<SPLIT, RAG>(p, acc, red, tm, tn, wm, wn, le, te);
    }
    if (nvb >= ntiles) break;
    vb = nvb;
    nvb = vb + gstride;                  // (tickets: replaced at kt == 2 of the tile that starts now)
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus stages have landed before the LDS is released
  const int tid = lane_id_now() + 64 * wave;
  if (sched != nullptr && tid == 0 && atomicAdd(&sched[8], 1) == gstride - 1) {
#pragma unroll
    for (int i = 0; i < 9; ++i) atomicExch(&sched[i], 0);
  }
  if constexpr (EPI == EPI_PLAIN || EPI == EPI_DGRAD_BN) {
    // (read from the kernel-argument segment here, through a pointer the compiler cannot see through: referenced as
    // p.tail its 20 fields sit in SGPRs through the whole tile loop)
    const char* ka = (const char*)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(ka));
    const BnTail tail = *reinterpret_cast<const BnTail*>(ka + offsetof(GemmParams, tail));
    bn_tail_run(tail, tid, NT, gridDim.x, words + 1);
  }
}

#undef V2_STEP
#undef V2_REQ_A
#undef V2_REQ_B
#undef V2_ADVANCE

// ---------------------------------------------------------------------------------------------------- RC x RC (wgrad)
// The weight gradients dW[cout, cin] = dy[P, cout]^T . a[P, cin]: both operands row-contiguous, the contraction runs
// over the P rows (split into K ranges, one 256 x 256 tile of one range per workgroup, fp32 slabs out).  Same pipeline
// as the KC x KC loop above -- 4 waves x 128 x 128, both operands requested two K steps ahead into two LDS stages, the
// fragments of the next half-step in a second register set -- on the 8-wave kernel's RC images: a stage is [64 k][256
// rows] as it lies in HBM (source-side XOR swizzle of the 16-B granules, dma_frag_offset<RC>), an MFMA fragment (8
// consecutive k of one row) is two ds_read_b64_tr_b16 transpose reads, the MFMA shape is 32x32x16 (4 x 4 blocks per wave).
// Counters of the 8-wave kernel on dW[1024,1024] (profiles/r04_gemm_counters.json): waves parked at s_waitcnt / the
// step barrier 47 % of their cycles (the 4-wave KC loop: 16 %), 17 % more GPU cycles than the KC product of the same size.
template <bool F16>
__device__ __forceinline__ void mfma32(f32x16& c, const bf16x8& a, const bf16x8& b) {
  if constexpr (F16) asm volatile("v_mfma_f32_32x32x16_f16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
  else asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}
// one fragment = two transpose reads 4 k-rows (2 KB) apart; both halves land in one 4-register tuple
__device__ __forceinline__ void lds_read_tr(bf16x8& d, unsigned addr, int off) {
  s16x4 lo, hi;
  asm volatile("ds_read_b64_tr_b16 %0, %2 offset:%3\n\tds_read_b64_tr_b16 %1, %2 offset:%4"
               : "=&v"(lo), "=&v"(hi) : "v"(addr), "n"(off), "n"(off + 2048));
  union { struct { s16x4 a, b; } s; bf16x8 v; } u;
  u.s.a = lo;
  u.s.b = hi;
  d = u.v;
}
// one 1-KB piece (2 k-rows x 512 B) of an RC stage image, p = 0..31
__device__ __forceinline__ void piece_rc(buf_rsrc_t r, long ld, long krow, int col0, bf16_t* s_img, int p, const unsigned (&vo)[2]) {
  const unsigned soff = (unsigned)__builtin_amdgcn_readfirstlane((int)(((krow + 2 * p) * ld + col0) * 2));
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s_img + p * 512), 16, vo[p & 1], soff, 0, 0);
}

template <bool SPLIT>
__global__ __launch_bounds__(NT) void gemm_bf16_v2rc_kernel(GemmParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbm = p.M / BM, nbn = p.N / BN;
  int tm, tn;
  const int split = block_coords(p, nbm, nbn, tm, tn);
  const int kbeg = split * p.k_per_split;
  const int kend = min(p.K, kbeg + p.k_per_split);
  const int nt = max(0, (kend - kbeg) / BK);
  const bf16_t* A = reinterpret_cast<const bf16_t*>(p.A);
  const bf16_t* B = reinterpret_cast<const bf16_t*>(p.B);
  const long lda = p.lda, ldb = p.ldb;
  const int seg_len = SPLIT ? p.seg_len : p.K;                       // rows of the operands as they lie in memory
  const buf_rsrc_t rA = make_rsrc(A, (long)seg_len * lda * 2), rB = make_rsrc(B, (long)seg_len * ldb * 2);
  const int soA0 = SPLIT ? (int)p.seg_off_a[0] : 0, soA1 = SPLIT ? (int)p.seg_off_a[1] : 0, soA2 = SPLIT ? (int)p.seg_off_a[2] : 0;
  const int soB0 = SPLIT ? (int)p.seg_off_b[0] : 0, soB1 = SPLIT ? (int)p.seg_off_b[1] : 0, soB2 = SPLIT ? (int)p.seg_off_b[2] : 0;
  unsigned voA[2], voB[2];
  piece_lane_offsets<RC>(lda, lane, voA);
  piece_lane_offsets<RC>(ldb, lane, voB);
  // fragment offsets (elements inside an operand image) of the wave's four 32-row blocks per operand
  int offA[4], offB[4];
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    offA[i] = dma_frag_offset<RC>(wm * 128 + i * 32, lane);
    offB[i] = dma_frag_offset<RC>(wn * 128 + i * 32, lane);
  }
  // request cursor over this workgroup's K range [kbeg, kend) of the (SPLIT: three-segment) contraction; behind the
  // last step it wraps to the first one: the loop stays branch-free, the two surplus stages land in dead LDS
  const int seg0 = SPLIT ? kbeg / seg_len : 0, lk0 = SPLIT ? kbeg - seg0 * seg_len : kbeg;
  int lseg = seg0, lk = lk0, lcount = 0;
#define V2RC_COLA() (tm * BM + (SPLIT ? (lseg == 0 ? soA0 : (lseg == 1 ? soA1 : soA2)) : 0))
#define V2RC_COLB() (tn * BN + (SPLIT ? (lseg == 0 ? soB0 : (lseg == 1 ? soB1 : soB2)) : 0))
#define V2RC_REQ_A(st, jj) piece_rc(rA, lda, lk, V2RC_COLA(), (st), wave * 8 + (jj), voA)
#define V2RC_REQ_B(st, jj) piece_rc(rB, ldb, lk, V2RC_COLB(), (st) + OP_TILE, wave * 8 + (jj), voB)
#define V2RC_ADVANCE()                                 \
  do {                                                 \
    lk += BK;                                          \
    if (SPLIT && lk == seg_len) { lk = 0; ++lseg; }    \
    if (++lcount >= nt) { lcount = 0; lseg = seg0; lk = lk0; } \
  } while (0)

  f32x16 acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
  bf16x8 af[2][2][4], bfr[2][2][4];            // [register set][k-step inside the half][block]
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  if (nt > 0) {
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) V2RC_REQ_B(smem, jj);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) V2RC_REQ_A(smem, jj);
    V2RC_ADVANCE();
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) V2RC_REQ_B(smem + STAGE, jj);
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) V2RC_REQ_A(smem + STAGE, jj);
    V2RC_ADVANCE();
    asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        lds_read_tr(bfr[0][ks][j], lds0 + (unsigned)offB[j] * 2, 32768 + ks * 8192);
        lds_read_tr(af[0][ks][j], lds0 + (unsigned)offA[j] * 2, ks * 8192);
      }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  int s = 0;
  for (int kt = 0; kt < nt; ++kt) {
    bf16_t* cur = smem + s * STAGE;
    const unsigned cb = lds0 + (unsigned)s * (STAGE * 2), nb = lds0 + (unsigned)(s ^ 1) * (STAGE * 2);
    unsigned aA[4], aB[4], nA[4], nB[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      aA[i] = cb + (unsigned)offA[i] * 2;
      aB[i] = cb + (unsigned)offB[i] * 2;
      nA[i] = nb + (unsigned)offA[i] * 2;
      nB[i] = nb + (unsigned)offB[i] * 2;
    }
    // ---------------- first half (k-steps 0, 1 of the stage): MFMAs on register set 0; k-steps 2, 3 are read into set 1
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const int ks = m >> 4, i = (m >> 2) & 3, j = m & 3;
      if (m < 8) lds_read_tr(bfr[1][m >> 2][m & 3], aB[m & 3], 32768 + (2 + (m >> 2)) * 8192);
      if (m == 10) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // B image of this stage: dead
      if (m >= 11 && m < 19) V2RC_REQ_B(cur, m - 11);
      if (m >= 12 && m < 20) lds_read_tr(af[1][(m - 12) >> 2][(m - 12) & 3], aA[(m - 12) & 3], (2 + ((m - 12) >> 2)) * 8192);
      if (m == 23) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");       // A image: dead
      if (m >= 24 && (m & 1) == 0) V2RC_REQ_A(cur, (m - 24) >> 1);
      mfma32<SPLIT>(acc[i][j], af[0][ks][i], bfr[0][ks][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    // ---------------- second half: MFMAs on set 1; the next stage's k-steps 0, 1 are read into set 0
#pragma unroll
    for (int m = 0; m < 32; ++m) {
      const int ks = m >> 4, i = (m >> 2) & 3, j = m & 3;
      if (m < 8 && (m & 1) == 0) V2RC_REQ_A(cur, 4 + (m >> 1));
      if (m == 10) asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");        // stage t + 1 has landed
      if (m >= 11 && m < 19) lds_read_tr(bfr[0][(m - 11) >> 2][(m - 11) & 3], nB[(m - 11) & 3], 32768 + ((m - 11) >> 2) * 8192);
      if (m >= 19 && m < 27) lds_read_tr(af[0][(m - 19) >> 2][(m - 19) & 3], nA[(m - 19) & 3], ((m - 19) >> 2) * 8192);
      mfma32<SPLIT>(acc[i][j], af[1][ks][i], bfr[1][ks][j]);
      __builtin_amdgcn_sched_barrier(0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    V2RC_ADVANCE();
    s ^= 1;
  }
  asm volatile("s_nop 15\n\ts_nop 15\n\ts_nop 15" ::: "memory");
  // fp32 slab out: accumulator r of lane (c = lane & 31, half = lane >> 5) of block (i, j) is row 32 i + (r & 3) + 8 (r >> 2)
  // + 4 half, column 32 j + c (32 lanes = 128 contiguous bytes of a row)
  {
    const float os = SPLIT ? p.out_scale : 1.f;
    const int l31 = lane & 31, half = lane >> 5;
    float* C = reinterpret_cast<float*>(p.C) + (long)split * p.c_split_stride + (long)(tm * BM + wm * 128 + 4 * half) * p.ldc +
               tn * BN + wn * 128 + l31;
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int r = 0; r < 16; ++r)
          C[(long)(i * 32 + (r & 3) + 8 * (r >> 2)) * p.ldc + j * 32] = SPLIT ? acc[i][j][r] * os : acc[i][j][r];
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef V2RC_COLA
#undef V2RC_COLB
#undef V2RC_REQ_A
#undef V2RC_REQ_B
#undef V2RC_ADVANCE
}

}  // namespace v2
