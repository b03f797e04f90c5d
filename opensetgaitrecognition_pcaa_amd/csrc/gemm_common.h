// Shared between gemm.hip (fp32-MFMA and small-tile bf16 kernels, C-ABI entry)
// and gemm_bf16.hip (256x256-tile bf16 kernel).
#pragma once
#include "common.h"
#include "bn_tail.h"

struct GemmParams {
  const void* A; const void* B; void* C;
  long lda, ldb, ldc;
  int M, N, K;
  const float* bias;
  double* colstats;
  int nrep;
  int k_per_split;
  int atomic;
  int nsplit;       // number of K splits
  int split_fast;   // 1: 1-D grid, block b -> split b % nsplit, tile b / nsplit (see block_coords)
  long c_split_stride;   // slab split-K: split s stores its partial tile at C + s * c_split_stride (no atomics)
  // dgrad fused with the BatchNorm+ELU backward of the layer below (pcaa_gemm_dgrad_bn)
  const void* ep_y; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
  int ep_xc;   // pcaa_gemm_affine_elu: rows per mean-pool group (0: the activation itself leaves)
  // LDS-DMA kernel, one workgroup per CU: 9 zero-initialised ints (8 per-XCD tile tickets + a count of finished
  // workgroups, reset by the last one); NULL: every workgroup walks a fixed share of the tiles
  int* sched;
  // split-fp16 operands (pcaa_gemm_split3, round 3): the contraction runs over three segments of seg_len elements;
  // in segment s operand A is read at element offset seg_off_a[s] and B at seg_off_b[s] (the hi / lo halves of the
  // operands' [hi | lo] fp16 images: hi.hi + lo.hi + hi.lo on the f16 MFMA), K = 3 * seg_len.  seg_len == 0: plain
  // bf16 operands.
  int seg_len;
  long seg_off_a[3], seg_off_b[3];
  float out_scale;     // split operands are fp16 images of (value * 2^k): the accumulators are multiplied by this
                       // (the exact power of two 2^-(ka + kb)) before the epilogue
  // the BatchNorm finalize of the statistics this launch accumulates (colstats), run by its last workgroup
  // (bn_tail.h); kind 0: none.  Only the LDS-DMA kernel's KC x KC instantiations carry one.
  BnTail tail{};
};

// XCD-aware, bijective block -> (tile_m, tile_n) map: the 8 XCDs (blocks b, b+8,
// ... share one) each walk a contiguous range of tiles with the N-tiles of one
// M-panel adjacent, so an A panel is re-read from that XCD's L2.
__device__ __forceinline__ void xcd_tile_coords(int nbm, int nbn, int bid, int& tm, int& tn) {
  const int nb = nbm * nbn;
  const int q = nb >> 3, r = nb & 7;
  const int xcd = bid & 7, idx = bid >> 3;
  const int v = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
  tm = v / nbn;
  tn = v - tm * nbn;
}

// Block -> (split, tile).  With split_fast (wgrad: long K, few tiles, nsplit % 8 == 0) all tiles
// of one K-range land on one XCD (blocks b and b+8 share an XCD), so the operand rows of that
// range are fetched from HBM once and re-read by the other tiles from that XCD's L2.
__device__ __forceinline__ int block_coords(const GemmParams& p, int nbm, int nbn, int& tm, int& tn) {
  if (p.split_fast) {
    const int s = blockIdx.x % p.nsplit, t = blockIdx.x / p.nsplit;
    tm = t / nbn;
    tn = t - tm * nbn;
    return s;
  }
  xcd_tile_coords(nbm, nbn, blockIdx.x, tm, tn);
  return blockIdx.z;
}

// Kernel-exact timing of ONE launch (bench.py's roofline figure): start/stop events armed by
// pcaa_time_next_gemm are attached to the next LDS-DMA GEMM launch of this thread through
// hipExtLaunchKernelGGL, i.e. they carry the timestamps of the kernel's own dispatch packet (what rocprofv3
// reports) instead of bracketing it with two marker packets that each drain the queue (+25 us per launch).
struct PcaaLaunchEvents { hipEvent_t start, stop; };
PcaaLaunchEvents pcaa_take_launch_events();      // returns {nullptr, nullptr} when nothing is armed; disarms

// 256x256-tile bf16 kernel (gemm_bf16.hip).  Returns false if the shape/dtype
// combination is not served by it (caller falls back to the small-tile kernel).
bool pcaa_launch_gemm_bf16_big(const GemmParams& p, int a_dtype, int a_layout, int b_dtype, int b_layout,
                               int c_dtype, int nsplit, hipStream_t stream);
bool pcaa_launch_gemm_dgrad_bn(const GemmParams& p, hipStream_t stream);
bool pcaa_launch_gemm_affine_elu(const GemmParams& p, hipStream_t stream);
