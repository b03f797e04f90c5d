// LAB (timing only, no result written): the K loop of the 256x256 bf16 tile with FOUR waves of 128x128 accumulators
// (one wave per SIMD, 256 accumulator registers) instead of eight of 128x64 -- a third fewer LDS fragment reads per
// SIMD and no second wave competing for the MFMA pipe -- on the product kernel's LDS-DMA ring (A x3, B x2) and tile
// walk.  Compare with the product kernel built without its epilogue.
//   hipcc --offload-arch=gfx950 -O3 -o gemm4w gemm4w.hip && ./gemm4w
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
typedef __bf16 bf16_t;
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef __amdgpu_buffer_rsrc_t rsrc_t;
constexpr int BM = 256, BN = 256, BK = 64, D_TILE = 256 * 64;

__device__ __forceinline__ void piece(rsrc_t r, long ld, int row0, int k0, bf16_t* s_tile, int p, const unsigned (&vo)[2]) {
  const unsigned soff = (unsigned)(((long)(row0 + 8 * p) * ld + k0) * 2);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(r, (__attribute__((address_space(3))) void*)(s_tile + p * 512), 16, vo[p & 1], soff, 0, 0);
}

template <int NW>      // 4: wave tile 128x128; 8: 128x64 (the product's split) -- same loop structure for both
__global__ __launch_bounds__(NW * 64) void k_loop(const bf16_t* A, const bf16_t* B, float* out, int M, int N, int K) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  bf16_t* smem = reinterpret_cast<bf16_t*>(smem_raw);
  constexpr int WN = NW / 2;                 // waves along N
  constexpr int NB = 256 / WN / 16;          // 16-column blocks per wave: 8 (4 waves) or 4 (8 waves)
  constexpr int PPW = 32 / NW;               // pieces per wave and operand
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;
  const int nbn = N / BN, nb = (M / BM) * nbn, nt = K / BK;
  rsrc_t rA = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(A), 0, (int)(unsigned)((long)M * K * 2), 0x00020000);
  rsrc_t rB = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(B), 0, (int)(unsigned)((long)N * K * 2), 0x00020000);
  unsigned vo[2];
#pragma unroll
  for (int e = 0; e < 2; ++e) {
    const int g = (lane & 7) ^ ((4 * e + (lane >> 4)) & 7);
    vo[e] = (unsigned)((lane >> 3) * (long)K * 2 + 16 * g);
  }
  auto slotA = [&](int i) { return smem + (i < 2 ? i : 3) * D_TILE; };
  auto slotB = [&](int i) { return smem + (i == 0 ? 2 : 4) * D_TILE; };
  const int l15 = lane & 15, q = lane >> 4;
  int kofs[2];
#pragma unroll
  for (int s2 = 0; s2 < 2; ++s2) kofs[s2] = ((4 * s2 + q) ^ (l15 >> 1)) * 8;
  const int offA0 = (wm * 128 + l15) * 64, offB0 = (wn * (256 / WN) + l15) * 64;
  float total = 0.f;
  for (int vb = blockIdx.x; vb < nb; vb += gridDim.x) {
    const int tm = vb / nbn, tn = vb % nbn;
    f32x4 acc[8][NB];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) acc[i][j] = f32x4{0.f, 0.f, 0.f, 0.f};
    auto pA = [&](int k0, bf16_t* s, int j) { piece(rA, K, tm * BM, k0, s, wave * PPW + j, vo); };
    auto pB = [&](int k0, bf16_t* s, int j) { piece(rB, K, tn * BN, k0, s, wave * PPW + j, vo); };
#pragma unroll
    for (int j = 0; j < PPW; ++j) pA(0, slotA(0), j);
#pragma unroll
    for (int j = 0; j < PPW; ++j) pB(0, slotB(0), j);
#pragma unroll
    for (int j = 0; j < PPW; ++j) pA(BK, slotA(1), j);
    asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW) : "memory");
    int ia = 0;
    for (int t = 0; t < nt; ++t) {
      const bf16_t* sA = slotA(ia);
      const bf16_t* sB = slotB(t & 1);
      bf16_t* nA = slotA(ia == 0 ? 2 : ia - 1);
      bf16_t* nB = slotB((t + 1) & 1);
      const int k0 = (t + 1) * BK, k0A = k0 + BK;
      const bool more = t + 1 < nt, moreA = t + 2 < nt;
      // per k-half: all B fragments of the wave, then the A row blocks one by one (a block's NB MFMAs per A fragment);
      // the requests for the next stages go out one piece per A row block
      int pc = 0;
#pragma unroll
      for (int s2 = 0; s2 < 2; ++s2) {
        bf16x8 bfr[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) bfr[j] = *reinterpret_cast<const bf16x8*>(sB + offB0 + j * 16 * 64 + kofs[s2]);
        bf16x8 af = *reinterpret_cast<const bf16x8*>(sA + offA0 + kofs[s2]);
#pragma unroll
        for (int i = 0; i < 8; ++i) {
          bf16x8 afn = af;
          if (i < 7) afn = *reinterpret_cast<const bf16x8*>(sA + offA0 + (i + 1) * 16 * 64 + kofs[s2]);
          // 16 request slots per step (8 per k-half): B(t+1) first, then A(t+2); PPW pieces each per wave
          {
            const int slot = s2 * 8 + i;                       // 0..15
            constexpr int per = (2 * PPW + 15) / 16;          // pieces per slot: 1 (4 waves: 16 pieces) or <1 (8 waves)
            for (int z = 0; z < (per > 0 ? per : 1); ++z) {
              const int pz = (2 * PPW >= 16) ? slot * per + z : (slot % 2 == 0 ? slot / 2 : -1);
              if (pz >= 0 && pz < 2 * PPW) {
                if (pz < PPW) { if (more) pB(k0, nB, pz); }
                else if (moreA) pA(k0A, nA, pz - PPW);
              }
            }
          }
          (void)pc;
#pragma unroll
          for (int j = 0; j < NB; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af, bfr[j], acc[i][j], 0, 0, 0);
          af = afn;
        }
      }
      if (moreA) asm volatile("s_waitcnt vmcnt(%0) lgkmcnt(0)\n\ts_barrier" ::"n"(PPW) : "memory");
      else asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)\n\ts_barrier" ::: "memory");
      ia = ia == 2 ? 0 : ia + 1;
    }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
      for (int j = 0; j < NB; ++j) total += acc[i][j].x + acc[i][j].y + acc[i][j].z + acc[i][j].w;
  }
  if (total == 12345.f) out[tid] = total;
}

template <int NW>
static void run(const bf16_t* A, const bf16_t* B, float* out, int M, int N, int K) {
  auto kern = k_loop<NW>;
  (void)hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, 163840);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 6; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL(kern, dim3(256), dim3(NW * 64), 163840, 0, A, B, out, M, N, K);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms; (void)hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double fl = 2.0 * M * N * K, tiles = (double)(M / 256) * (N / 256) / 256;
  printf("%d waves  [%d,%d]x[%d,%d]^T  %.3f ms  %.0f TF-equivalent  %.1f us per tile (K loop only, no epilogue)\n", NW, M, K, N, K,
         best, fl / best / 1e9, best * 1e3 / tiles);
}

int main() {
  const int M = 245760;
  bf16_t *A, *B; float* out;
  if (hipMalloc(&A, (size_t)M * 1024 * 2) != hipSuccess || hipMalloc(&B, (size_t)1024 * 1024 * 2) != hipSuccess) return 1;
  (void)hipMalloc(&out, 4096);
  {  // uniform random bf16 in [-1, 1): constant operands draw less power and read high
    const size_t na = (size_t)M * 1024, nbw = (size_t)1024 * 1024;
    unsigned short* h = (unsigned short*)malloc(na * 2);
    unsigned long long st = 88172645463325252ULL;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (size_t i = 0; i < na; ++i) {
      const float f = (float)((rnd() >> 40) & 0xffff) / 32768.f - 1.f;
      unsigned u; memcpy(&u, &f, 4);
      h[i] = (unsigned short)(u >> 16);
    }
    (void)hipMemcpy(A, h, na * 2, hipMemcpyHostToDevice);
    (void)hipMemcpy(B, h + 12345, nbw * 2, hipMemcpyHostToDevice);
    free(h);
  }
  for (int K : {512, 1024}) {
    run<8>(A, B, out, M, 1024, K);
    run<4>(A, B, out, M, 1024, K);
  }
  (void)hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
