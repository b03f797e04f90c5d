// Per-CU load bandwidth on gfx950 as a function of the bytes a CU keeps in flight: the LDS-DMA path
// (buffer_load_dwordx4 ... lds, what the GEMM's operand ring uses) against plain 16-B loads into registers.
// One 512-thread workgroup per CU (128 KB of dynamic LDS keeps it alone), 256 workgroups, every wave streams
// 1-KB pieces with H..2H of them outstanding.  Regimes: "hbm" = every workgroup its own 32-MB region (no reuse),
// "l2" = the 32 workgroups of an XCD-sized group re-read one 2-MB region (L2 hits after the first pass).
//   hipcc --offload-arch=gfx950 -O3 -o cu_load_bw cu_load_bw.hip && ./cu_load_bw
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>

typedef __amdgpu_buffer_rsrc_t rsrc_t;
typedef unsigned int u32x4 __attribute__((ext_vector_type(4)));

// MODE 3: LDS-DMA as the GEMM issues it -- a wave instruction = 8 rows x one 128-B segment (8 lanes x 16 B) of a
// row-major matrix whose rows lie `stride` bytes apart -- against MODE 0's 1 KB of consecutive bytes.
template <int H, int MODE>
__global__ __launch_bounds__(512) void stream_kernel(const unsigned char* __restrict__ src, long region_bytes,
                                                     int share, int iters, unsigned* __restrict__ sink, int stride = 0) {
  extern __shared__ __attribute__((aligned(16))) unsigned char lds[];
  const int lane = threadIdx.x & 63;
  const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
  const long region = (long)(blockIdx.x / share) * region_bytes;
  const unsigned char* base = src + region;
  rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(base), 0, (int)(unsigned)region_bytes, 0x00020000);
  const unsigned voff = lane * 16;
  // the workgroup walks its region in 8-KB rows (one 1-KB piece per wave), workgroups that share a region start apart
  unsigned pos = (unsigned)(((blockIdx.x % share) * 65536u + wave * 1024u) % (unsigned)region_bytes);
  u32x4 acc = {0, 0, 0, 0};
  if (MODE == 0 || MODE == 3) {
    unsigned char* slot = lds + wave * (2 * H) * 1024;
    // MODE 3: the region is a matrix of region_bytes / stride rows; piece n of the workgroup's walk = rows 8n .. 8n+7
    // of 128-B column (8n / rows); the per-lane part of the address never changes
    const unsigned voffs = MODE == 3 ? (unsigned)((lane >> 3) * stride + (lane & 7) * 16) : voff;
    const unsigned rows = MODE == 3 ? (unsigned)(region_bytes / stride) : 1u;
    unsigned piece = (blockIdx.x % share) * 64u + wave;
    auto issue = [&](int s) {
      unsigned soff = pos;
      if (MODE == 3) {
        const unsigned r0 = (piece * 8u) % rows, kc = ((piece * 8u) / rows) % ((unsigned)stride / 128u);
        soff = r0 * (unsigned)stride + kc * 128u;
        piece += 8;
      }
      __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(slot + s * 1024), 16, voffs, soff, 0, 0);
      pos += 8192; if (pos >= (unsigned)region_bytes) pos -= (unsigned)region_bytes;
    };
#pragma unroll
    for (int s = 0; s < H; ++s) issue(s);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < H; ++s) issue(H + s);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H) : "memory");
#pragma unroll
      for (int s = 0; s < H; ++s) issue(s);
      asm volatile("s_waitcnt vmcnt(%0)" ::"n"(H) : "memory");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    acc.x = *(volatile unsigned*)(slot + lane * 4);
  } else if (MODE == 2) {
    // 8-B loads per lane: a wave instruction moves 512 B; the workgroup's 8 waves x 2 instructions cover the same 8-KB row
    typedef unsigned int u32x2 __attribute__((ext_vector_type(2)));
    u32x2 buf[2][2 * H];
    const unsigned voff8 = lane * 8;
    auto issue = [&](int b, int s) {
      buf[b][2 * s] = *reinterpret_cast<const u32x2*>(base + pos + voff8);
      buf[b][2 * s + 1] = *reinterpret_cast<const u32x2*>(base + pos + 512 + voff8);
      pos += 8192; if (pos >= (unsigned)region_bytes) pos -= (unsigned)region_bytes;
    };
#pragma unroll
    for (int s = 0; s < H; ++s) issue(0, s);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < H; ++s) issue(1, s);
#pragma unroll
      for (int s = 0; s < 2 * H; ++s) { acc.x ^= buf[0][s].x; acc.y ^= buf[0][s].y; }
#pragma unroll
      for (int s = 0; s < H; ++s) issue(0, s);
#pragma unroll
      for (int s = 0; s < 2 * H; ++s) { acc.x ^= buf[1][s].x; acc.y ^= buf[1][s].y; }
    }
#pragma unroll
    for (int s = 0; s < 2 * H; ++s) { acc.x ^= buf[0][s].x; acc.y ^= buf[0][s].y; }
  } else {
    u32x4 buf[2][H];
    auto issue = [&](int b, int s) {
      buf[b][s] = *reinterpret_cast<const u32x4*>(base + pos + voff);
      pos += 8192; if (pos >= (unsigned)region_bytes) pos -= (unsigned)region_bytes;
    };
#pragma unroll
    for (int s = 0; s < H; ++s) issue(0, s);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
      for (int s = 0; s < H; ++s) issue(1, s);
#pragma unroll
      for (int s = 0; s < H; ++s) acc ^= buf[0][s];
#pragma unroll
      for (int s = 0; s < H; ++s) issue(0, s);
#pragma unroll
      for (int s = 0; s < H; ++s) acc ^= buf[1][s];
    }
#pragma unroll
    for (int s = 0; s < H; ++s) acc ^= buf[0][s];
  }
  if ((acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345678u) sink[0] = 1;
}

template <int H, int MODE>
static void run(const unsigned char* src, long region_bytes, int share, const char* regime, unsigned* sink, int stride = 0) {
  const int wgs = 256, iters = 2048 / H;       // ~4096 pieces per wave
  hipFuncSetAttribute((const void*)stream_kernel<H, MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 131072);
  hipEvent_t e0, e1;
  hipEventCreate(&e0); hipEventCreate(&e1);
  float best = 1e9f;
  for (int rep = 0; rep < 5; ++rep) {
    hipEventRecord(e0);
    hipLaunchKernelGGL((stream_kernel<H, MODE>), dim3(wgs), dim3(512), 131072, 0, src, region_bytes, share, iters, sink, stride);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    if (rep > 0 && ms < best) best = ms;
  }
  const double bytes = (double)wgs * 8 * 1024.0 * (H + 2.0 * H * iters);
  const double gbs = bytes / best / 1e6;
  char name[32];
  if (MODE == 3) snprintf(name, sizeof(name), "dma-%dB", stride);
  else snprintf(name, sizeof(name), "%s", MODE == 0 ? "lds-dma" : (MODE == 1 ? "vgpr" : "vgpr-8B"));
  printf("%-9s %-4s in flight per CU %3d-%3d KB: %8.1f GB/s total, %6.2f GB/s per CU = %5.1f B/clk at 2.1 GHz (%.3f ms)\n",
         name, regime, 8 * H, 16 * H, gbs, gbs / wgs, gbs / wgs / 2.1, best);
}

int main() {
  const long total = 8L << 30;
  unsigned char* src; unsigned* sink;
  if (hipMalloc(&src, total) != hipSuccess) { printf("alloc failed\n"); return 1; }
  hipMalloc(&sink, 4);
  hipMemset(src, 1, total);
  hipMemset(sink, 0, 4);
  // hbm: 256 regions of 32 MB; l2: 8 regions of 2 MB shared by 32 workgroups each
#define BOTH(H)                                              \
  run<H, 2>(src, 32L << 20, 1, "hbm", sink);                 \
  run<H, 0>(src, 32L << 20, 1, "hbm", sink);                 \
  run<H, 1>(src, 32L << 20, 1, "hbm", sink);                 \
  run<H, 0>(src, 2L << 20, 32, "l2", sink);                  \
  run<H, 3>(src, 2L << 20, 32, "l2", sink, 1024);            \
  run<H, 3>(src, 2L << 20, 32, "l2", sink, 2048);            \
  run<H, 3>(src, 32L << 20, 1, "hbm", sink, 1024);           \
  run<H, 3>(src, 32L << 20, 1, "hbm", sink, 2048);           \
  run<H, 1>(src, 2L << 20, 32, "l2", sink);
  BOTH(1) BOTH(2) BOTH(4) BOTH(8)
  hipDeviceSynchronize();
  printf("%s\n", hipGetErrorString(hipGetLastError()));
  return 0;
}
