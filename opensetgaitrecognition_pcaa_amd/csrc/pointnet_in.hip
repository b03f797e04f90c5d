// First PointNet layer (reference models.py:87-89: Conv2d(C -> 512, 1x1) on the raw
// points) and its weight gradient.  The contraction is only C = 4/5 wide, so
// this is HBM-bound streaming of the [P, 512] output (forward) or of dy
// (wgrad), not MFMA work: FMAs from registers, the point tile of each
// workgroup staged once in LDS, BatchNorm statistics as in the GEMM epilogue.
#include "common.h"

namespace {

constexpr int MAXC = 8;
constexpr int FWD_ROWS = 128;    // points per workgroup (forward)
constexpr int WG_ROWS = 256;     // points per workgroup (wgrad): 960 workgroups at P = 245760

template <typename T>
__global__ __launch_bounds__(256) void pointnet_in_fwd_kernel(const float* __restrict__ x, int C,
                                                              const float* __restrict__ W,   // [cout, C]
                                                              const float* __restrict__ bias,
                                                              T* __restrict__ y, long P, int cout,
                                                              double* __restrict__ stats, int nrep) {
  __shared__ float xs[FWD_ROWS * MAXC];
  __shared__ f32x4 red[2][256];
  const int qpr = cout >> 2, rl = 256 / qpr;
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const long r0 = (long)blockIdx.x * FWD_ROWS;
  const int nrows = (int)min((long)FWD_ROWS, P - r0);
  for (int e = threadIdx.x; e < FWD_ROWS * MAXC; e += 256) {
    const int r = e / MAXC, c = e - r * MAXC;
    xs[e] = (r < nrows && c < C) ? x[(r0 + r) * C + c] : 0.f;
  }
  float w[4][MAXC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) w[j][c] = (c < C) ? W[(cq * 4 + j) * C + c] : 0.f;
  const f32x4 b = bias ? load4(bias + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (int r = rlane; r < nrows; r += rl) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const float xv = xs[r * MAXC + c];
      acc.x = fmaf(w[0][c], xv, acc.x);
      acc.y = fmaf(w[1][c], xv, acc.y);
      acc.z = fmaf(w[2][c], xv, acc.z);
      acc.w = fmaf(w[3][c], xv, acc.w);
    }
    s1 += acc;
    s2 += acc * acc;
    store4(y + (r0 + r) * cout + cq * 4, acc + b);
  }
  if (stats) {
    red[0][threadIdx.x] = s1;
    red[1][threadIdx.x] = s2;
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * cout; o += 256) {
      const int stat = o / cout, cc = o - stat * cout;
      double v = 0.0;
      for (int l = 0; l < rl; ++l) v += (double)red[stat][l * qpr + (cc >> 2)][cc & 3];
      unsafeAtomicAdd(&stats[((long)(blockIdx.x % nrep) * 2 + stat) * cout + cc], v);
    }
  }
}

// dW[o][c] += sum_p dy[p][o] * x[p][c]
template <typename T>
__global__ __launch_bounds__(256) void pointnet_in_wgrad_kernel(const T* __restrict__ dy,
                                                                const float* __restrict__ x, int C,
                                                                float* __restrict__ dW, long P, int cout) {
  // also reused for the [rl][cout][MAXC] row-lane combine below: rl*cout = 1024 -> 8192 floats
  __shared__ float xs[(WG_ROWS * MAXC > 8192) ? WG_ROWS * MAXC : 8192];
  const int qpr = cout >> 2, rl = 256 / qpr;
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const long r0 = (long)blockIdx.x * WG_ROWS;
  const int nrows = (int)min((long)WG_ROWS, P - r0);
  for (int e = threadIdx.x; e < WG_ROWS * MAXC; e += 256) {
    const int r = e / MAXC, c = e - r * MAXC;
    xs[e] = (r < nrows && c < C) ? x[(r0 + r) * C + c] : 0.f;
  }
  __syncthreads();
  float acc[4][MAXC];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) acc[j][c] = 0.f;
  // 4 rows per trip: four independent 8/16-B loads in flight per lane (the loop is latency-bound)
  int r = rlane;
  for (; r + 3 * rl < nrows; r += 4 * rl) {
    f32x4 d[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) d[u] = load4(dy + (r0 + r + u * rl) * cout + cq * 4);
#pragma unroll
    for (int u = 0; u < 4; ++u) {
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const float xv = xs[(r + u * rl) * MAXC + c];
        acc[0][c] = fmaf(d[u].x, xv, acc[0][c]);
        acc[1][c] = fmaf(d[u].y, xv, acc[1][c]);
        acc[2][c] = fmaf(d[u].z, xv, acc[2][c]);
        acc[3][c] = fmaf(d[u].w, xv, acc[3][c]);
      }
    }
  }
  for (; r < nrows; r += rl) {
    const f32x4 d = load4(dy + (r0 + r) * cout + cq * 4);
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      const float xv = xs[r * MAXC + c];
      acc[0][c] = fmaf(d.x, xv, acc[0][c]);
      acc[1][c] = fmaf(d.y, xv, acc[1][c]);
      acc[2][c] = fmaf(d.z, xv, acc[2][c]);
      acc[3][c] = fmaf(d.w, xv, acc[3][c]);
    }
  }
  // combine the row lanes through LDS (xs is free now), then ONE atomic per (channel, feature)
  // per workgroup: all workgroups add into the same cout*C words, so the count matters
  __syncthreads();
  float* red = xs;   // [rl][cout][MAXC]  (rl * cout * 8 floats <= 256 * 4 * 8 = 8192 <= WG_ROWS * MAXC)
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < MAXC; ++c) red[(rlane * cout + cq * 4 + j) * MAXC + c] = acc[j][c];
  __syncthreads();
  for (int o = threadIdx.x; o < cout * C; o += 256) {
    const int ch = o / C, c = o - ch * C;
    float v = 0.f;
    for (int l = 0; l < rl; ++l) v += red[(l * cout + ch) * MAXC + c];
    atomicAdd(&dW[o], v);
  }
}

// dst[r][c] = bf16(src[r][c]); dst_t[c][r] = bf16(src[r][c])  (weights: a few MB at most)
__global__ void cast_bf16_kernel(const float* __restrict__ src, bf16_t* __restrict__ dst,
                                 bf16_t* __restrict__ dst_t, int R, int C) {
  const long n = (long)R * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const bf16_t v = (bf16_t)src[i];
    if (dst) dst[i] = v;
    if (dst_t) {
      const int r = (int)(i / C), c = (int)(i - (long)r * C);
      dst_t[(long)c * R + r] = v;
    }
  }
}

inline bool shape_ok(int C, int cout) {
  if (C < 1 || C > MAXC || cout < 4 || cout > 1024 || (cout & 3)) return false;
  return (256 % (cout >> 2)) == 0;
}

}  // namespace

extern "C" int pcaa_pointnet_in_fwd(const float* x, int C, const float* W, const float* bias, void* y,
                                    int y_dtype, long P, int cout, double* stats, int nrep, void* stream) {
  PCAA_CHECK_ARG(x && W && y && P >= 1, "pcaa_pointnet_in_fwd: bad args");
  PCAA_CHECK_ARG(shape_ok(C, cout), "pcaa_pointnet_in_fwd: need C<=%d and cout/4 dividing 256 (C=%d cout=%d)", MAXC, C, cout);
  PCAA_CHECK_ARG(!stats || nrep >= 1, "pcaa_pointnet_in_fwd: bad nrep");
  const unsigned grid = (unsigned)cdiv(P, FWD_ROWS);
  if (y_dtype == PCAA_F32)
    hipLaunchKernelGGL(pointnet_in_fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), x, C, W, bias,
                       (float*)y, P, cout, stats, nrep);
  else if (y_dtype == PCAA_BF16)
    hipLaunchKernelGGL(pointnet_in_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), x, C, W, bias,
                       (bf16_t*)y, P, cout, stats, nrep);
  else { pcaa_set_error("pcaa_pointnet_in_fwd: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_fwd");
}

extern "C" int pcaa_pointnet_in_wgrad(const void* dy, int dy_dtype, const float* x, int C, float* dW, long P,
                                      int cout, void* stream) {
  PCAA_CHECK_ARG(dy && x && dW && P >= 1, "pcaa_pointnet_in_wgrad: bad args");
  PCAA_CHECK_ARG(shape_ok(C, cout), "pcaa_pointnet_in_wgrad: need C<=%d and cout/4 dividing 256 (C=%d cout=%d)", MAXC, C, cout);
  const unsigned grid = (unsigned)cdiv(P, WG_ROWS);
  if (dy_dtype == PCAA_F32)
    hipLaunchKernelGGL(pointnet_in_wgrad_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream),
                       (const float*)dy, x, C, dW, P, cout);
  else if (dy_dtype == PCAA_BF16)
    hipLaunchKernelGGL(pointnet_in_wgrad_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)dy, x, C, dW, P, cout);
  else { pcaa_set_error("pcaa_pointnet_in_wgrad: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_wgrad");
}

extern "C" int pcaa_cast_bf16(const float* src, void* dst, void* dst_t, int R, int C, void* stream) {
  PCAA_CHECK_ARG(src && (dst || dst_t) && R >= 1 && C >= 1, "pcaa_cast_bf16: bad args");
  const long n = (long)R * C;
  long g = cdiv(n, 256);
  if (g > 2048) g = 2048;
  hipLaunchKernelGGL(cast_bf16_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), src, (bf16_t*)dst,
                     (bf16_t*)dst_t, R, C);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_cast_bf16");
}
