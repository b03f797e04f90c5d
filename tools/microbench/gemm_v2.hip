// LAB: the 256x256x64 bf16 tile loop restructured after hipBLASLt's hand-written gfx950 kernel
// (Custom_Cijk_Alik_Bljk_BBS_BH_..._MT256x256x64_MI16x16x1; disassembled in round 4, docs/LAB_LOG.md section 7):
//   * FOUR waves (2 x 2), one per SIMD, 128 x 128 accumulators each (256 accumulator registers): every fragment read
//     from the LDS feeds 8 MFMAs (the product kernel's 128 x 64 wave tiles: 4 or 8) -- 128 KB of fragment reads per
//     K step instead of 192 KB;
//   * the fragments of the NEXT k-half are read into a second register set while the MFMAs of the current one run,
//     so an LDS stage is dead as soon as its second half has been read -- its refill for K step t + 2 is issued in
//     the MIDDLE of step t (a barrier per operand releases the region).  Two LDS stages (128 KB) then carry a
//     prefetch distance of TWO steps for BOTH operands: a piece has a whole step (>= 1.2 us) to land, up to
//     ~116 KB in flight per CU (the product ring: A two steps ahead, B one; <= 96 KB);
//   * the load stream does not stop at tile boundaries: steps t + 1, t + 2 of the last steps of a tile are the next
//     tile's first stages (the product kernel requests them in a burst before its epilogue);
//   * a third barrier per step (counted vmcnt: only this step's own requests may still be in flight) publishes
//     stage t + 1, whose first-half fragments are then read under the second half's MFMAs.
// Output: bf16 C straight from the accumulators (B rows permuted into the LDS, as the product's 16x16x32 kernels).
//   hipcc --offload-arch=gfx950 -O3 -o gemm_v2 gemm_v2.hip && ./gemm_v2 [check]
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int BM = 256, BN = 256, BK = 64;
constexpr int OP_TILE = 256 * 64;            // elements of one operand's stage image (32 KB)
constexpr int STAGE = 2 * OP_TILE;           // A image then B image (64 KB)
constexpr int LDS_BYTES = 2 * STAGE * 2;     // two stages: 128 KB

#ifndef SCHED
#define SCHED 1
#endif
#ifndef EXTRA
#define EXTRA 0
#endif
#ifndef VARIANT
#define VARIANT 0
#endif
#ifndef NT_STORE
#define NT_STORE 0
#endif
// LAB knobs (timing only, results are garbage): drop one ingredient of the K loop to see what the step time is made of
#ifndef NO_DMA
#define NO_DMA 0
#endif
#ifndef NO_LDS
#define NO_LDS 0
#endif
#ifndef NO_BAR
#define NO_BAR 0
#endif
#if NO_BAR
#define BAR_LGKM() asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory")
#define BAR_VM() asm volatile("s_waitcnt vmcnt(16)" ::: "memory")
#else
#define BAR_LGKM() asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory")
#define BAR_VM() asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory")
#endif
#define SB() __builtin_amdgcn_sched_barrier(0)
// The accumulators are pinned to the ACCUMULATOR register file ("a" constraint): left to the register allocator (the
// builtin), a wave with 256 accumulator + 128 fragment registers got accumulators in both files, ~1 200
// v_accvgpr_* copies in the loop and 165 spilled registers.  The asm statements keep their order; hazards: an
// accumulator is next touched 64 MFMAs later, fragments come from ds_reads (the compiler's own s_waitcnt).
// fragment reads as asm too: a plain LDS load whose value is only consumed in the NEXT loop iteration gets sunk to the
// loop latch (behind all the MFMAs it was meant to hide under).  The compiler then does not know these registers are
// pending: every use is behind an explicit s_waitcnt lgkmcnt(0) placed in this file (before the two release barriers
// and at the very end of the loop body -- ahead of any copy the back edge may need).
__device__ __forceinline__ void lds_read(bf16x8& d, unsigned addr, int off) {
#if NO_LDS
  asm volatile("" : "+v"(d) : "v"(addr));
#else
  asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(off));
#endif
}
__device__ __forceinline__ void mfma(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// LAB (round 5, MFMA32=1, TIMING ONLY -- the products are garbage): the same K loop issuing HALF as many MFMAs of twice the
// length (v_mfma_f32_32x32x16_bf16: 32 cycles of matrix pipe per 4-cycle issue instead of 16).  Question: is the 20-35 % that
// the 16 requests per wave and step cost issue time that a longer MFMA would cover (one wave per SIMD: while a request sits
// at the head of the wave's instruction stream nothing else issues, and a 16-cycle MFMA drains before the next one starts)?
#ifndef MFMA32
#define MFMA32 0
#endif
typedef float f32x16 __attribute__((ext_vector_type(16)));
__device__ __forceinline__ void mfma32(f32x16& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_32x32x16_bf16 %0, %1, %2, %0" : "+a"(c) : "v"(a), "v"(b));
}

// the first product of a tile: C = 0 as the inline constant (the product kernel's mfma0)
__device__ __forceinline__ void mfma0(f32x4& c, const bf16x8& a, const bf16x8& b) {
  asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, 0" : "=a"(c) : "v"(a), "v"(b));
}

__device__ __forceinline__ uint32_t pack2(float lo, float hi) {
  typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
  bf16x2 b;
  b.x = (bf16_t)lo;
  b.y = (bf16_t)hi;
  return *reinterpret_cast<uint32_t*>(&b);
}

// one 1-KB piece (8 LDS rows x 128 B) of an operand's stage image: p = 0..31
template <bool PERM>
__device__ __forceinline__ void piece(rsrc_t r, long ld, int row0, int k0, bf16_t* s_img, int p, const unsigned (&vo)[2]) {
  const int prow = PERM ? 128 * (p >> 4) + 64 * (p & 1) + ((p >> 1) & 7) : 8 * p;      // PERM: LDS row 16 j + c of a 128-row group <- operand row 8 c + j
  const unsigned soff = (unsigned)(((long)(row0 + prow) * ld + k0) * 2);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s_img + p * 512), 16, vo[p & 1], soff, 0, 0);
}

// XCD-aware, bijective block -> tile map (the product's gemm_common.h): the 8 XCDs (blocks b, b + 8, ... share one) each
// walk a contiguous range of tiles with the N tiles of one M panel adjacent: an A panel is re-read from that XCD's L2
__device__ __forceinline__ void tile_coords(int nb, int nbn, int bid, int& tm, int& tn) {
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tm = v / nbn;
  tn = v - tm * nbn;
}

template <bool STORE>
__global__ __launch_bounds__(256) void gemm_v2(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                               bf16_t* __restrict__ C, int M, int N, int K, float* sink) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = N / BN, ntiles = (M / BM) * nbn, nt = K / BK;
  rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)(unsigned)((long)M * K * 2), 0x00020000);
  rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)(unsigned)((long)N * K * 2), 0x00020000);
  // per-lane constants are rebuilt from an opaque copy of the lane id at the top of every tile: kept live across the
  // epilogue (which wants every VGPR for the accumulators on their way out) they were spilled, and their reload in
  // front of the loop made the compiler put a drain-everything s_waitcnt vmcnt in the loop header
  unsigned voA[2], voB[2];
  int l15, q, kof0, kof1, fA, fB;
  auto lane_consts = [&](int ln) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int g = (ln & 7) ^ ((4 * e + (ln >> 4)) & 7);
      voA[e] = (unsigned)((ln >> 3) * (long)K * 2 + 16 * g);
      voB[e] = (unsigned)(8 * (ln >> 3) * (long)K * 2 + 16 * g);      // PERM: the rows of a piece lie 8 apart
    }
    l15 = ln & 15;
    q = ln >> 4;
    // fragment addresses (elements): row (block * 16 + l15), k-granule (4 * half + q) ^ swizzle
    kof0 = ((q) ^ (l15 >> 1)) * 8;
    kof1 = ((4 + q) ^ (l15 >> 1)) * 8;
    fA = (wm * 128 + l15) * 64;                 // + i * 1024 per row block
    fB = OP_TILE + (wn * 128 + l15) * 64;       // + j * 1024
  };
  int ln = lane;
  lane_consts(ln);

  // load cursor: (tile, k step) of the next stage to request -- runs two steps ahead of the compute cursor and
  // straight across tile boundaries
  // past the last tile it wraps to this workgroup's first one: the loop stays branch-free (the two surplus stages per
  // launch land in dead LDS; the kernel drains them before it ends)
  const int gstride_l = gridDim.x;
  int lvb = blockIdx.x, lkt = 0;
  int ltm, ltn;
  tile_coords(ntiles, nbn, lvb, ltm, ltn);
  auto advance = [&]() {
    if (++lkt == nt) {
      lkt = 0;
      lvb += gstride_l;
      if (lvb >= ntiles) lvb = blockIdx.x;
      tile_coords(ntiles, nbn, lvb, ltm, ltn);
    }
  };
  auto reqA = [&](bf16_t* st, int jj) { if (!NO_DMA) piece<false>(rA, K, ltm * BM, lkt * BK, st, wave * 8 + jj, voA); };
  auto reqB = [&](bf16_t* st, int jj) { if (!NO_DMA) piece<true>(rB, K, ltn * BN, lkt * BK, st + OP_TILE, wave * 8 + jj, voB); };

  f32x4 acc[8][8];
#if MFMA32
  f32x16 acc32[4][4];
#endif
  bf16x8 af[2][8], bfr[2][8];

  int vb = blockIdx.x;
  if (vb >= ntiles) return;
  // prologue: stages 0 and 1
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) reqB(smem, jj);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) reqA(smem, jj);
  advance();
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) reqB(smem + STAGE, jj);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) reqA(smem + STAGE, jj);
  advance();
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int j = 0; j < 8; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(smem + fB + j * 1024 + kof0);
#pragma unroll
  for (int i = 0; i < 8; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(smem + fA + i * 1024 + kof0);

  int s = 0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const int gstride = gridDim.x;
  for (;;) {
    int tm, tn;
    tile_coords(ntiles, nbn, vb, tm, tn);
    asm volatile("" : "+v"(ln));
    lane_consts(ln);
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < 8; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
#if MFMA32
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
      for (int j = 0; j < 4; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) acc32[i][j][e] = 0.f;
#endif
    for (int kt = 0; kt < nt; ++kt) {
      bf16_t* cur = smem + s * STAGE;              // stage of this step (and destination of the requests for step t + 2)
      // LDS byte addresses of the fragment reads: second half of this stage, first half of the next
      const unsigned cb = lds0 + (unsigned)s * (STAGE * 2), nb = lds0 + (unsigned)(s ^ 1) * (STAGE * 2);
      const unsigned aB1 = cb + (unsigned)(fB + kof1) * 2, aA1 = cb + (unsigned)(fA + kof1) * 2;
      const unsigned aB0 = nb + (unsigned)(fB + kof0) * 2, aA0 = nb + (unsigned)(fA + kof0) * 2;
      // ---------------- k-half 0: MFMAs on af[0] / bfr[0]
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        const int i = m >> 3, j = m & 7;
        // second-half fragments of B, then of A, from the current stage
        if ((m & 1) == 1 && m < 16) lds_read(bfr[1][m >> 1], aB1, (m >> 1) * 2048);
#if VARIANT == 0
        if (m == 21) BAR_LGKM();      // B image of this stage: dead
        if (m >= 24 && m < 40 && (m & 1) == 0) lds_read(af[1][(m - 24) >> 1], aA1, ((m - 24) >> 1) * 2048);
        if (m >= 23 && m < 39 && (m & 1) == 1) reqB(cur, (m - 23) >> 1);                  // 8 pieces of B(t + 2)
        if (m == 46) BAR_LGKM();      // A image: dead
        if (m >= 48 && (m & 3) == 0) reqA(cur, (m - 48) >> 2);                             // pieces 0..3 of A(t + 2)
#elif VARIANT == 1
        // B's requests one per FOUR MFMAs instead of every other one
        if (m == 21) BAR_LGKM();
        if (m >= 24 && m < 40 && (m & 1) == 0) lds_read(af[1][(m - 24) >> 1], aA1, ((m - 24) >> 1) * 2048);
        if (m >= 23 && m < 55 && ((m - 23) & 3) == 0) reqB(cur, (m - 23) >> 2);
        if (m == 46) BAR_LGKM();
        if (m >= 49 && ((m - 49) & 3) == 0) reqA(cur, (m - 49) >> 2);
#endif
#if MFMA32
        (void)i; (void)j;
        if (m & 1) mfma32(acc32[m >> 4][(m >> 2) & 3], af[0][2 * (m >> 4) + ((m >> 1) & 1)], bfr[0][2 * ((m >> 2) & 3) + ((m >> 1) & 1)]);
#else
        mfma(acc[i][j], af[0][i], bfr[0][j]);
#endif
#if SCHED
        SB();
#endif
      }
      // ---------------- k-half 1: MFMAs on af[1] / bfr[1]
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        const int i = m >> 3, j = m & 7;
#if VARIANT == 0
        if (m < 16 && (m & 3) == 0) reqA(cur, 4 + (m >> 2));                               // pieces 4..7 of A(t + 2)
#elif VARIANT == 1
        if (m < 16 && (m & 3) == 1) reqA(cur, 4 + (m >> 2));
#endif
        // stage t + 1 has landed: everything but this step's own 16 requests (loads retire in order, so "at most 16
        // outstanding" means the older stage is complete whatever the previous tile's C stores -- which share the
        // counter -- are doing; they can only make the wait longer)
        if (m == 20) BAR_VM();
        if (m >= 22 && m < 38 && (m & 1) == 0) lds_read(bfr[0][(m - 22) >> 1], aB0, ((m - 22) >> 1) * 2048);
        if (m >= 38 && m < 54 && (m & 1) == 0) lds_read(af[0][(m - 38) >> 1], aA0, ((m - 38) >> 1) * 2048);
#if MFMA32
        (void)i; (void)j;
        if (m & 1) mfma32(acc32[m >> 4][(m >> 2) & 3], af[1][2 * (m >> 4) + ((m >> 1) & 1)], bfr[1][2 * ((m >> 2) & 3) + ((m >> 1) & 1)]);
#else
        mfma(acc[i][j], af[1][i], bfr[1][j]);
#endif
#if SCHED
        SB();
#endif
      }
#if EXTRA
      // LAB: a third product per step on fragments already in registers (the split-fp16 mode's hi*lo term would be one):
      // does the delivery-bound loop absorb 50 % more MFMAs?
#pragma unroll
      for (int m = 0; m < 64; ++m) {
        mfma(acc[m >> 3][m & 7], af[1][m >> 3], bfr[1][m & 7]);
#if SCHED
        SB();
#endif
      }
#endif
      asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // the next step's first-half fragments are in their registers
      advance();
      s ^= 1;
    }
    asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");      // the last MFMAs' results (the asm hides the hazard from the compiler)
    // ---------------- tile done: C leaves from the registers (block j of lane c = column 64 (j >> 2) + 4 c + (j & 3))
    if (STORE) {
      int le = lane;
      asm volatile("" : "+v"(le));
      bf16_t* c0 = C + (long)(tm * BM + wm * 128 + 4 * (le >> 4)) * N + tn * BN + wn * 128 + 8 * (le & 15);
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          uint4 o;
          o.x = pack2(acc[i][0][r], acc[i][1][r]);
          o.y = pack2(acc[i][2][r], acc[i][3][r]);
          o.z = pack2(acc[i][4][r], acc[i][5][r]);
          o.w = pack2(acc[i][6][r], acc[i][7][r]);
#if NT_STORE
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
          u32x4 ov = {o.x, o.y, o.z, o.w};
          __builtin_nontemporal_store(ov, reinterpret_cast<u32x4*>(c0 + (long)(i * 16 + r) * N));
#else
          *reinterpret_cast<uint4*>(c0 + (long)(i * 16 + r) * N) = o;      // 16 B per lane, 16 lanes = 256 contiguous bytes
#endif
        }
    } else {
      float t = 0.f;
#pragma unroll
      for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) t += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
#if MFMA32
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j)
#pragma unroll
          for (int e = 0; e < 16; ++e) t += acc32[i][j][e];
#endif
      if (t == 12345.f) sink[tid] = t;
    }
    vb += gstride;
    if (vb >= ntiles) break;
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // the surplus stages must have landed before the LDS is released
}


// =====================================================================================================================
// ROUND 5 LAB: the tile epilogue UNDER the next tile's first K step.
// One wave per SIMD and all 256 accumulator registers in use: there is no second accumulator set, and a second workgroup
// per CU would need 256 x 128 tiles (1.5x the operand requests per MFMA, the very ingredient that bounds the loop).  But a
// tile's FIRST K step starts every accumulator from the MFMA's zero constant (mfma0), in row-block order: right before the
// eight mfma0 of row block i overwrite acc[i][0..7], those 32 registers still hold the PREVIOUS tile's results.  So the
// previous tile's epilogue is cut into eight slices, one per row block: read the 32 accumulators (v_accvgpr_read), issue
// the eight mfma0, and convert + store the 4 rows x 8 columns between them -- the stores' issue, their latency and (with
// STATS) the column statistics run in the shadow of MFMAs instead of in front of them.  The first-half loop of that step
// becomes VALU-bound (~48-80 vector instructions per eight MFMAs), ~+0.5 ... 1.1 us per tile against the ~3.8 us the
// stores cost behind the loop.
// Stores share vmcnt with the LDS-DMA requests, and the loop's "stage t + 1 has landed" test is a COUNTED wait
// (vmcnt(16): only this step's own requests may be outstanding): 32 stores issued in between would have to complete
// first.  The step that carries an epilogue therefore waits for stage t + 1 with vmcnt(0) BEFORE its first store (the
// requests are a whole step old by then) and only needs the barrier at the usual place; by the counted wait of the
// step after it the stores are more than a step old.
// C is addressed through a buffer resource re-based per tile: row offsets live in the scalar offset, no vector address
// arithmetic per store.
template <bool STATS>
__global__ __launch_bounds__(256) void gemm_v2o(const bf16_t* __restrict__ A, const bf16_t* __restrict__ B,
                                                bf16_t* __restrict__ C, int M, int N, int K, float* colsum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave >> 1, wn = wave & 1;
  const int nbn = N / BN, ntiles = (M / BM) * nbn, nt = K / BK;
  rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)(unsigned)((long)M * K * 2), 0x00020000);
  rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)(unsigned)((long)N * K * 2), 0x00020000);
  unsigned voA[2], voB[2];
  int l15, q, kof0, kof1, fA, fB;
  auto lane_consts = [&](int ln) {
#pragma unroll
    for (int e = 0; e < 2; ++e) {
      const int g = (ln & 7) ^ ((4 * e + (ln >> 4)) & 7);
      voA[e] = (unsigned)((ln >> 3) * (long)K * 2 + 16 * g);
      voB[e] = (unsigned)(8 * (ln >> 3) * (long)K * 2 + 16 * g);
    }
    l15 = ln & 15;
    q = ln >> 4;
    kof0 = ((q) ^ (l15 >> 1)) * 8;
    kof1 = ((4 + q) ^ (l15 >> 1)) * 8;
    fA = (wm * 128 + l15) * 64;
    fB = OP_TILE + (wn * 128 + l15) * 64;
  };
  int ln = lane;
  lane_consts(ln);
  const int gstride_l = gridDim.x;
  int lvb = blockIdx.x, lkt = 0;
  int ltm, ltn;
  tile_coords(ntiles, nbn, lvb, ltm, ltn);
#define O_ADVANCE()                                   \
  do {                                                \
    if (++lkt == nt) {                                \
      lkt = 0;                                        \
      lvb += gstride_l;                               \
      if (lvb >= ntiles) lvb = blockIdx.x;            \
      tile_coords(ntiles, nbn, lvb, ltm, ltn);        \
    }                                                 \
  } while (0)
#define O_REQ_A(st, jj) piece<false>(rA, K, ltm * BM, lkt * BK, (st), wave * 8 + (jj), voA)
#define O_REQ_B(st, jj) piece<true>(rB, K, ltn * BN, lkt * BK, (st) + OP_TILE, wave * 8 + (jj), voB)

  f32x4 acc[8][8];
  bf16x8 af[2][8], bfr[2][8];
  int vb = blockIdx.x;
  if (vb >= ntiles) return;
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) O_REQ_B(smem, jj);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) O_REQ_A(smem, jj);
  O_ADVANCE();
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) O_REQ_B(smem + STAGE, jj);
#pragma unroll
  for (int jj = 0; jj < 8; ++jj) O_REQ_A(smem + STAGE, jj);
  O_ADVANCE();
  asm volatile("s_waitcnt vmcnt(16)\n\ts_barrier" ::: "memory");
#pragma unroll
  for (int j = 0; j < 8; ++j) bfr[0][j] = *reinterpret_cast<const bf16x8*>(smem + fB + j * 1024 + kof0);
#pragma unroll
  for (int i = 0; i < 8; ++i) af[0][i] = *reinterpret_cast<const bf16x8*>(smem + fA + i * 1024 + kof0);

  int s = 0;
  const unsigned lds0 = (unsigned)(size_t)(__attribute__((address_space(3))) unsigned char*)smem_raw;
  const int gstride = gridDim.x;
  // the finished tile whose results are still in the accumulators: its C window as a buffer resource + this lane's offset
  rsrc_t rC = __builtin_amdgcn_make_buffer_rsrc(C, 0, 0, 0x00020000);
  unsigned voC = 0;
  const unsigned rowb = (unsigned)N * 2u;
  float cs[8], cs2[8];                                // STATS: this lane's running column sums / sums of squares of the tile being stored
  // E: 0 = first tile of the workgroup (nothing to store), 1 = the previous tile's results leave under this step
#define O_STEP(MF0, E)                                                                                                  \
  {                                                                                                                     \
    bf16_t* cur = smem + s * STAGE;                                                                                     \
    const unsigned cb = lds0 + (unsigned)s * (STAGE * 2), nb = lds0 + (unsigned)(s ^ 1) * (STAGE * 2);                  \
    const unsigned aB1 = cb + (unsigned)(fB + kof1) * 2, aA1 = cb + (unsigned)(fA + kof1) * 2;                          \
    const unsigned aB0 = nb + (unsigned)(fB + kof0) * 2, aA0 = nb + (unsigned)(fA + kof0) * 2;                          \
    f32x4 old[8];                                                                                                       \
    _Pragma("unroll") for (int m = 0; m < 64; ++m) {                                                                    \
      const int i = m >> 3, j = m & 7;                                                                                  \
      if ((m & 1) == 1 && m < 16) lds_read(bfr[1][m >> 1], aB1, (m >> 1) * 2048);                                       \
      if (m == 21) BAR_LGKM();                                                                                          \
      if (m >= 24 && m < 40 && (m & 1) == 0) lds_read(af[1][(m - 24) >> 1], aA1, ((m - 24) >> 1) * 2048);               \
      if (m >= 23 && m < 39 && (m & 1) == 1) O_REQ_B(cur, (m - 23) >> 1);                                               \
      if (m == 46) BAR_LGKM();                                                                                          \
      if (m >= 48 && (m & 3) == 0) O_REQ_A(cur, (m - 48) >> 2);                                                         \
      if (E) {                                                                                                          \
        if (j == 0) {                                                                                                   \
          /* explicit reads: an SSA alias would keep the old accumulator value alive BESIDE the one mfma0 defines */    \
          _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                            \
            _Pragma("unroll") for (int rr = 0; rr < 4; ++rr) {                                                          \
              float t_;                                                                                                 \
              asm volatile("v_accvgpr_read_b32 %0, %1" : "=v"(t_) : "a"(acc[i][jj][rr]));                               \
              old[jj][rr] = t_;                                                                                         \
            }                                                                                                           \
          }                                                                                                             \
        }                                                                                                               \
        if (m == 3) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      /* stage t + 1 landed, before the first store */ \
        if (j >= 4) {                                                                                                   \
          const int r = j - 4;                                                                                          \
          typedef unsigned u32x4 __attribute__((ext_vector_type(4)));                                                   \
          const u32x4 o = {pack2(old[0][r], old[1][r]), pack2(old[2][r], old[3][r]), pack2(old[4][r], old[5][r]),        \
                           pack2(old[6][r], old[7][r])};                                                                \
          __builtin_amdgcn_raw_buffer_store_b128(o, rC, voC, (i * 16 + r) * rowb, 0);                                   \
          if (STATS) {                                                                                                  \
            /* the empty asm pins each add to its slot: left alone, the adds are SUNK to their only use behind the    \
               step, and the 256 old values they need stay live until then (126 spilled registers) */                  \
            _Pragma("unroll") for (int jj = 0; jj < 8; ++jj) {                                                          \
              cs[jj] += old[jj][r];                                                                                     \
              cs2[jj] = fmaf(old[jj][r], old[jj][r], cs2[jj]);                                                          \
              asm volatile("" : "+v"(cs[jj]), "+v"(cs2[jj]));                                                           \
            }                                                                                                           \
          }                                                                                                             \
        }                                                                                                               \
      }                                                                                                                 \
      MF0(acc[i][j], af[0][i], bfr[0][j]);                                                                              \
      SB();                                                                                                             \
    }                                                                                                                   \
    _Pragma("unroll") for (int m = 0; m < 64; ++m) {                                                                    \
      const int i = m >> 3, j = m & 7;                                                                                  \
      if (m < 16 && (m & 3) == 0) O_REQ_A(cur, 4 + (m >> 2));                                                           \
      if (m == 20) {                                                                                                    \
        if (E) asm volatile("s_barrier" ::: "memory");                                                                  \
        else BAR_VM();                                                                                                  \
      }                                                                                                                 \
      if (m >= 22 && m < 38 && (m & 1) == 0) lds_read(bfr[0][(m - 22) >> 1], aB0, ((m - 22) >> 1) * 2048);              \
      if (m >= 38 && m < 54 && (m & 1) == 0) lds_read(af[0][(m - 38) >> 1], aA0, ((m - 38) >> 1) * 2048);               \
      mfma(acc[i][j], af[1][i], bfr[1][j]);                                                                             \
      SB();                                                                                                             \
    }                                                                                                                   \
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");                                                                  \
    O_ADVANCE();                                                                                                        \
    s ^= 1;                                                                                                             \
  }

  // The workgroup's FIRST tile is peeled out of the tile loop: one body whose first step sometimes carries an epilogue
  // and sometimes not defines every accumulator in two branches, and the merge cost 750-1200 B of scratch (the round-4
  // finding about `if (kt == 0)` inside the K loop, again)
#define O_TILE_DONE()                                                                                                   \
  {                                                                                                                     \
    int le = lane;                                                                                                      \
    asm volatile("" : "+v"(le));                                                                                        \
    rC = __builtin_amdgcn_make_buffer_rsrc(C + (long)(tm * BM + wm * 128) * N + tn * BN + wn * 128, 0,                  \
                                           (int)(128u * rowb), 0x00020000);                                             \
    voC = (unsigned)(4 * (le >> 4)) * rowb + 16u * (unsigned)(le & 15);                                                 \
  }
  int tm, tn;
  tile_coords(ntiles, nbn, vb, tm, tn);
  O_STEP(mfma0, 0)
  for (int kt = 1; kt < nt; ++kt) O_STEP(mfma, 0)
  O_TILE_DONE()
  vb += gstride;
  while (vb < ntiles) {
    tile_coords(ntiles, nbn, vb, tm, tn);
    asm volatile("" : "+v"(ln));
    lane_consts(ln);
    if (STATS) {
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) cs[jj] = cs2[jj] = 0.f;
    }
    O_STEP(mfma0, 1)
    if (STATS) {
      float t = 0.f;
#pragma unroll
      for (int jj = 0; jj < 8; ++jj) t += cs[jj] + cs2[jj];
      if (t == 12345.f) colsum[tid] = t;               // (lab: the sums are formed, their hand-off is not timed here)
    }
    for (int kt = 1; kt < nt; ++kt) O_STEP(mfma, 0)
    O_TILE_DONE()
    vb += gstride;
  }
  // the last tile of this workgroup: the stand-alone epilogue
  asm volatile("s_nop 15\n\ts_nop 15" ::: "memory");
#pragma unroll
  for (int i = 0; i < 8; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
      const u32x4 o = {pack2(acc[i][0][r], acc[i][1][r]), pack2(acc[i][2][r], acc[i][3][r]), pack2(acc[i][4][r], acc[i][5][r]),
                       pack2(acc[i][6][r], acc[i][7][r])};
      __builtin_amdgcn_raw_buffer_store_b128(o, rC, voC, (i * 16 + r) * rowb, 0);
    }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#undef O_STEP
#undef O_TILE_DONE
#undef O_ADVANCE
#undef O_REQ_A
#undef O_REQ_B
}

template <bool STATS>
static float run_o(const bf16_t* A, const bf16_t* B, bf16_t* C, float* sink, int M, int N, int K, int reps = 8) {
  auto kern = gemm_v2o<STATS>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < reps; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), LDS_BYTES, 0, A, B, C, M, N, K, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double fl = 2.0 * M * N * K, tiles = (double)(M / 256) * (N / 256) / 256;
  printf("gemm_v2o %s [%d,%d]x[%d,%d]^T  %.3f ms  %.0f TF  %.1f us per tile\n",
         STATS ? "stores under step 0 + sums" : "stores under step 0       ", M, K, N, K, best, fl / best / 1e9, best * 1e3 / tiles);
  return best;
}

template <bool STORE>
static float run(const bf16_t* A, const bf16_t* B, bf16_t* C, float* sink, int M, int N, int K, int reps = 8) {
  auto kern = gemm_v2<STORE>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < reps; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(256), LDS_BYTES, 0, A, B, C, M, N, K, sink);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double fl = 2.0 * M * N * K, tiles = (double)(M / 256) * (N / 256) / 256;
  printf("gemm_v2 %s [%d,%d]x[%d,%d]^T  %.3f ms  %.0f TF  %.1f us per tile\n", STORE ? "with bf16 C stores" : "K loop only     ", M, K,
         N, K, best, fl / best / 1e9, best * 1e3 / tiles);
  return best;
}

static float bf2f(unsigned short h) { unsigned u = (unsigned)h << 16; float f; memcpy(&f, &u, 4); return f; }

int main(int argc, char** argv) {
  const bool check = argc > 1 && !strcmp(argv[1], "check");
  const int M = check ? 1024 : 245760, NMAX = 1024, KMAX = 1024;
  bf16_t *A, *B, *C; float* sink;
  if (hipMalloc(&A, (size_t)M * KMAX * 2) != hipSuccess || hipMalloc(&B, (size_t)NMAX * KMAX * 2) != hipSuccess ||
      hipMalloc(&C, (size_t)M * NMAX * 2) != hipSuccess) return 1;
  (void)hipMalloc(&sink, 4096);
  const size_t na = (size_t)M * KMAX, nbw = (size_t)NMAX * KMAX;
  unsigned short* h = (unsigned short*)malloc((na + nbw) * 2);
  unsigned long long st = 88172645463325252ULL;
  auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
  for (size_t i = 0; i < na + nbw; ++i) {
    const float f = (float)((rnd() >> 40) & 0xffff) / 32768.f - 1.f;
    unsigned u; memcpy(&u, &f, 4);
    h[i] = (unsigned short)(u >> 16);
  }
  (void)hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice);
  (void)hipMemcpy(B, h + na, nbw * 2, hipMemcpyHostToDevice);
  if (check) {
    int bad = 0;
    for (int K : {128, 512, 1024})
      for (int N : {256, 512}) {
      for (int variant = 0; variant < 3; ++variant) {
        (void)hipMemset(C, 0, (size_t)M * NMAX * 2);
        (void)hipFuncSetAttribute((const void*)gemm_v2<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_v2o<false>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        (void)hipFuncSetAttribute((const void*)gemm_v2o<true>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS_BYTES);
        if (variant == 0) hipLaunchKernelGGL(gemm_v2<true>, dim3(3), dim3(256), LDS_BYTES, 0, A, B, C, M, N, K, sink);
        if (variant == 1) hipLaunchKernelGGL(gemm_v2o<false>, dim3(3), dim3(256), LDS_BYTES, 0, A, B, C, M, N, K, sink);
        if (variant == 2) hipLaunchKernelGGL(gemm_v2o<true>, dim3(1), dim3(256), LDS_BYTES, 0, A, B, C, M, N, K, sink);
        (void)hipDeviceSynchronize();
        unsigned short* hc = (unsigned short*)malloc((size_t)M * N * 2);
        (void)hipMemcpy(hc, C, (size_t)M * N * 2, hipMemcpyDeviceToHost);
        double worst = 0;
        for (int t = 0; t < 4000; ++t) {
          const int r = (int)(rnd() % M), c = (int)(rnd() % N);
          double ref = 0;
          for (int k = 0; k < K; ++k) ref += (double)bf2f(h[(size_t)r * K + k]) * (double)bf2f(h[na + (size_t)c * K + k]);
          const double err = fabs(ref - (double)bf2f(hc[(size_t)r * N + c])) / (fabs(ref) + 1.0);
          if (err > worst) worst = err;
        }
        printf("check variant %d K=%d N=%d: worst relative error of 4000 samples %.3e %s\n", variant, K, N, worst, worst < 1e-2 ? "ok" : "BAD");
        if (!(worst < 1e-2)) bad = 1;
        free(hc);
      }
      }
    printf("%s\n", hipGetErrorString(hipGetLastError()));
    return bad;
  }
  const bool lab = NO_DMA || NO_LDS || NO_BAR || MFMA32;
  if (MFMA32) printf("MFMA32=1 (timing only)\n");
  if (VARIANT) printf("VARIANT=%d\n", VARIANT);
  if (lab) printf("NO_DMA=%d NO_LDS=%d NO_BAR=%d EXTRA=%d\n", NO_DMA, NO_LDS, NO_BAR, EXTRA);
  for (int K : {512, 1024})
    for (int N : {512, 1024}) {
      if (lab && !MFMA32 && !(K == 1024 && N == 1024)) continue;
      run<false>(A, B, C, sink, M, N, K);
      if (!lab) run<true>(A, B, C, sink, M, N, K);
      if (!lab) run_o<false>(A, B, C, sink, M, N, K);
      if (!lab) run_o<true>(A, B, C, sink, M, N, K);
    }
  (void)hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
