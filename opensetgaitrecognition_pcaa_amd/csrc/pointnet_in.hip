// First PointNet layer (reference models.py:87-89: Conv2d(C -> 512, 1x1) on the raw
// points) and its weight gradient.  The contraction is only C = 4/5 wide, so
// this is HBM-bound streaming of the [P, 512] output (forward) or of dy
// (wgrad), not MFMA work: FMAs from registers, the point tile of each
// workgroup staged once in LDS, BatchNorm statistics as in the GEMM epilogue.
#include "common.h"
#include "bn_tail.h"

namespace {

constexpr int MAXC = 8;
constexpr int FWD_ROWS = 128;    // points per workgroup (forward)
constexpr int WG_ROWS = 256;     // points per workgroup (wgrad): 960 workgroups at P = 245760
constexpr int BWD_ROWS = 256;   // points per workgroup of the recompute backward passes (960 workgroups at config[1]; 128 rows: +10 % time, twice the atomics)

// CP: point features padded to 4 or 8 (C <= 4 is the common case: half the multiply-adds of the padded-to-8 loop);
// the product runs on column pairs with packed fp32 fused multiply-adds -- per element the same operations in the same
// order as the scalar form (and as the backward's recomputation).
template <typename T, int CP>
__global__ __launch_bounds__(256) void pointnet_in_fwd_kernel(const float* __restrict__ x, int C,
                                                              const float* __restrict__ W,   // [cout, C]
                                                              const float* __restrict__ bias,
                                                              T* __restrict__ y, long P, int cout,
                                                              double* __restrict__ stats, int nrep, BnTail tail) {
  __shared__ float xs[FWD_ROWS * CP];
  __shared__ f32x4 red[2][256];
  __shared__ int tail_flag;
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  const int qpr = cout >> 2, rl = 256 / qpr;
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const long r0 = (long)blockIdx.x * FWD_ROWS;
  const int nrows = (int)min((long)FWD_ROWS, P - r0);
  for (int e = threadIdx.x; e < FWD_ROWS * CP; e += 256) {
    const int r = e / CP, c = e - r * CP;
    xs[e] = (r < nrows && c < C) ? x[(r0 + r) * C + c] : 0.f;
  }
  f32x2 wlo[CP], whi[CP];
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    wlo[c] = f32x2{(c < C) ? W[(cq * 4 + 0) * C + c] : 0.f, (c < C) ? W[(cq * 4 + 1) * C + c] : 0.f};
    whi[c] = f32x2{(c < C) ? W[(cq * 4 + 2) * C + c] : 0.f, (c < C) ? W[(cq * 4 + 3) * C + c] : 0.f};
  }
  const f32x4 b = bias ? load4(bias + cq * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
  __syncthreads();
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (int r = rlane; r < nrows; r += rl) {
    f32x2 alo = {0.f, 0.f}, ahi = {0.f, 0.f};
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      const f32x2 xv = {xs[r * CP + c], xs[r * CP + c]};
      alo = __builtin_elementwise_fma(wlo[c], xv, alo);
      ahi = __builtin_elementwise_fma(whi[c], xv, ahi);
    }
    const f32x4 acc = {alo.x, alo.y, ahi.x, ahi.y};
    s1 += acc;
    s2 += acc * acc;
    if (y) store4(y + (r0 + r) * cout + cq * 4, acc + b);     // y == NULL: statistics only (recompute path)
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
  bn_tail_run(tail, threadIdx.x, 256, gridDim.x, &tail_flag);
}

// ---------------------------------------------------------------------------------------------
// Recompute path: the layer's pre-activation y = x . W^T costs C (4-5) FMAs per element, less than
// reading it back, so it is never stored.  Forward = statistics pass (kernel above, y == NULL) +
// apply pass (a = ELU(BN(y)) straight from x); backward = two passes over the incoming gradient
// da only: statistics (dz = da * ELU'(z) and y-hat recomputed), then dy = c0*dz + c1*y + c2 formed
// in registers and contracted with x into dW -- neither dz nor dy nor y touch HBM.
// ---------------------------------------------------------------------------------------------
template <int CP>
__device__ __forceinline__ void stage_points(float* xs, const float* __restrict__ x, int C, long r0, int nrows,
                                             int rows_cap) {
  for (int e = threadIdx.x; e < rows_cap * CP; e += 256) {
    const int r = e / CP, c = e - r * CP;
    xs[e] = (r < nrows && c < C) ? x[(r0 + r) * C + c] : 0.f;
  }
}
template <int CP>
__device__ __forceinline__ void load_weights(float (&w)[4][CP], const float* __restrict__ W, int C, int cq) {
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < CP; ++c) w[j][c] = (c < C) ? W[(cq * 4 + j) * C + c] : 0.f;
}
template <int CP>
__device__ __forceinline__ f32x4 point_dot(const float (&w)[4][CP], const float* xr) {
  f32x4 acc = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
  for (int c = 0; c < CP; ++c) {
    const float xv = xr[c];
    acc.x = fmaf(w[0][c], xv, acc.x);
    acc.y = fmaf(w[1][c], xv, acc.y);
    acc.z = fmaf(w[2][c], xv, acc.z);
    acc.w = fmaf(w[3][c], xv, acc.w);
  }
  return acc;
}

template <typename T, int CP, bool SPLIT = false>
__global__ __launch_bounds__(256) void pointnet_in_apply_kernel(const float* __restrict__ x, int C,
                                                                const float* __restrict__ W,
                                                                const float* __restrict__ scale,
                                                                const float* __restrict__ shift, T* __restrict__ a,
                                                                long P, int cout, int* oflow = nullptr) {
  __shared__ float xs[FWD_ROWS * CP];
  const int qpr = cout >> 2, rl = 256 / qpr;
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const long r0 = (long)blockIdx.x * FWD_ROWS;
  const int nrows = (int)min((long)FWD_ROWS, P - r0);
  stage_points<CP>(xs, x, C, r0, nrows, FWD_ROWS);
  float w[4][CP];
  load_weights<CP>(w, W, C, cq);
  const f32x4 sc = load4(scale + cq * 4), sh = load4(shift + cq * 4);
  __syncthreads();
  for (int r = rlane; r < nrows; r += rl) {
    const f32x4 acc = point_dot<CP>(w, xs + r * CP);
    f32x4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = elu_t<T>(acc[e] * sc[e] + sh[e]);
    if constexpr (SPLIT) store4_split(reinterpret_cast<split_t*>(a), (size_t)(r0 + r), cout, cq * 4, o, 1.f, oflow);   // [hi | lo] image
    else store4(a + (r0 + r) * cout + cq * 4, o);
  }
}

// MODE 0: statistics {sum dz, sum dz*yhat}; MODE 1: dW += dy^T . x with dy = c0*dz + c1*y + c2;
// MODE 2: as 1 with the incoming tensor already dz (ELU' applied by the fused dgrad epilogue);
// MODE 3 (round 3): ONE pass instead of MODE 0 + MODE 1 -- the statistics AND G += dz^T . x.  The weight gradient is
// linear in dy, and y = x . W^T, so
//     dW = dy^T x = c0 (.) (dz^T x) + c1 (.) (W . x^T x) + c2 (x) sum_p x
// needs, beside G = dz^T x, only the C x C second moments and the C sums of the points (points_moments_kernel); the
// combination is pointnet_in_bwd_combine_kernel.  The second read of the [P, cout] gradient (252 MB, 0.1 ms) is gone.
template <typename T, int MODE, int CP>
__global__ __launch_bounds__(256) void pointnet_in_bwd_kernel(const T* __restrict__ da, const float* __restrict__ x,
                                                              int C, const float* __restrict__ W,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ mean,     // MODE 0
                                                              const float* __restrict__ rstd,     // MODE 0
                                                              const float* __restrict__ coef,     // MODE 1: [3][cout]
                                                              double* __restrict__ stats, int nrep,
                                                              float* __restrict__ dW, long P, int cout, BnTail tail,
                                                              const double* __restrict__ pivot_mom = nullptr,
                                                              double pivot_inv_count = 0.0) {
  __shared__ int tail_flag;
  __shared__ float xs[1024 * CP];     // points [WG_ROWS][CP], later the [rl][cout][CP] row-lane combine (rl*cout = 1024)
  const int qpr = cout >> 2, rl = 256 / qpr;
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const long r0 = (long)blockIdx.x * BWD_ROWS;
  const int nrows = (int)min((long)BWD_ROWS, P - r0);
  stage_points<CP>(xs, x, C, r0, nrows, BWD_ROWS);
  // MODE 3 (round 4, advisor finding): G is accumulated against the points CENTRED on a pivot (their mean, from the
  // moments): dW = sum dy x has the three terms c0 G, c1 W.x^T x, c2 sum x cancel to the part that comes from the
  // points' spread, and for un-centred features (range, power in dB: mean >> deviation) fp32 partial sums of dz.x lose
  // exactly those digits.  y itself (ELU', y-hat) is still formed from the raw points.  xc: [BWD_ROWS][CP] behind xs.
  float* xc = xs + BWD_ROWS * CP;
  if constexpr (MODE == 3) {
    for (int e = threadIdx.x; e < BWD_ROWS * CP; e += 256) {
      const int r = e / CP, c = e - r * CP;
      const float piv = (pivot_mom != nullptr && c < C) ? (float)(pivot_mom[MAXC * MAXC + c] * pivot_inv_count) : 0.f;
      xc[e] = (r < nrows && c < C) ? x[(r0 + r) * C + c] - piv : 0.f;
    }
  }
  float w[4][CP];
  load_weights<CP>(w, W, C, cq);
  const f32x4 sc = load4(scale + cq * 4), sh = load4(shift + cq * 4);
  f32x4 p0, p1, p2 = {0.f, 0.f, 0.f, 0.f};
  if (MODE == 0 || MODE == 3) { p0 = load4(mean + cq * 4); p1 = load4(rstd + cq * 4); }
  else { p0 = load4(coef + cq * 4); p1 = load4(coef + cout + cq * 4); p2 = load4(coef + 2 * cout + cq * 4); }
  __syncthreads();
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = s1;
  float acc[4][CP];
#pragma unroll
  for (int j = 0; j < 4; ++j)
#pragma unroll
    for (int c = 0; c < CP; ++c) acc[j][c] = 0.f;
  // 4 rows per trip; MODE 0-2 issue the NEXT trip's four gradient loads before this trip's arithmetic
  auto load_trip = [&](int r, f32x4 (&g)[4]) {
#pragma unroll
    for (int u = 0; u < 4; ++u)
      g[u] = load4(da + (r0 + min(r + u * rl, nrows - 1)) * cout + cq * 4);     // unconditional (clamped): a load under
                                                                                // a branch waits for itself at once
  };
  if constexpr (MODE == 3 && sizeof(T) == 2) {
    // Packed-fp32 form (v_pk_fma_f32: two channels per instruction): ~76 VALU instructions per 4 channels of one row in
    // the scalar form, 37 here.  ELU' = e^min(z,0) is one v_exp_f32 of the pre-scaled z (scale and shift carry log2 e),
    // no compare / select.  Measured at config[1] (252 MB read): scalar form 0.119 ms, packed 0.113, packed with the
    // loads un-branched (below) 0.078.
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    constexpr float LOG2E = 1.4426950408889634f;
    f32x2 wlo[CP], whi[CP], alo[CP], ahi[CP];
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      wlo[c] = f32x2{w[0][c], w[1][c]};
      whi[c] = f32x2{w[2][c], w[3][c]};
      alo[c] = f32x2{0.f, 0.f};
      ahi[c] = f32x2{0.f, 0.f};
    }
    const f32x2 sclo = f32x2{sc.x, sc.y} * LOG2E, schi = f32x2{sc.z, sc.w} * LOG2E;
    const f32x2 shlo = f32x2{sh.x, sh.y} * LOG2E, shhi = f32x2{sh.z, sh.w} * LOG2E;
    const f32x2 rslo = {p1.x, p1.y}, rshi = {p1.z, p1.w};                       // y-hat = y * rstd - mean * rstd
    const f32x2 nmlo = f32x2{-p0.x, -p0.y} * rslo, nmhi = f32x2{-p0.z, -p0.w} * rshi;
    f32x2 s1lo = {0.f, 0.f}, s1hi = s1lo, s2lo = s1lo, s2hi = s1lo;
    const f32x2 zero2 = {0.f, 0.f};
    // TRIP rows per trip, the next trip's loads (raw, unconverted) issued before this trip's arithmetic.  Every load is
    // unconditional -- rows past the end are clamped to the last one and their arithmetic skipped: a load under a
    // branch is followed by its own s_waitcnt, which serialised the four loads of a trip (0.12 ms for 252 MB).
    constexpr int TRIP = 8;
    const T* base = da + r0 * cout + cq * 4;                  // workgroup-uniform part + lane column
    const int last = nrows - 1;
    auto ld_raw = [&](int row) {
      return *reinterpret_cast<const uint2*>(base + (unsigned)(min(row, last) * cout));
    };
    auto cvt = [&](uint2 raw) {
      return f32x4{__uint_as_float(raw.x << 16), __uint_as_float(raw.x & 0xffff0000u), __uint_as_float(raw.y << 16),
                   __uint_as_float(raw.y & 0xffff0000u)};
    };
    uint2 graw[TRIP], gnext[TRIP];
#pragma unroll
    for (int u = 0; u < TRIP; ++u) graw[u] = ld_raw(rlane + u * rl);
    for (int r = rlane; r < nrows; r += TRIP * rl) {
#pragma unroll
      for (int u = 0; u < TRIP; ++u) gnext[u] = ld_raw(r + (TRIP + u) * rl);
#pragma unroll
      for (int u = 0; u < TRIP; ++u) {
        if (r + u * rl >= nrows) break;
        const f32x4 gu = cvt(graw[u]);
        const f32x4 xq = *reinterpret_cast<const f32x4*>(xs + (r + u * rl) * CP);
        float xr[CP];
        xr[0] = xq.x; xr[1] = xq.y; xr[2] = xq.z; xr[3] = xq.w;
        if constexpr (CP > 4) {
          const f32x4 xq2 = *reinterpret_cast<const f32x4*>(xs + (r + u * rl) * CP + 4);
          xr[4] = xq2.x; xr[5] = xq2.y; xr[6] = xq2.z; xr[7] = xq2.w;
        }
        f32x2 ylo = zero2, yhi = zero2;
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const f32x2 xv = {xr[c], xr[c]};
          ylo = __builtin_elementwise_fma(wlo[c], xv, ylo);
          yhi = __builtin_elementwise_fma(whi[c], xv, yhi);
        }
        const f32x2 zlo = __builtin_elementwise_min(__builtin_elementwise_fma(ylo, sclo, shlo), zero2);
        const f32x2 zhi = __builtin_elementwise_min(__builtin_elementwise_fma(yhi, schi, shhi), zero2);
        const f32x2 elo = {__builtin_amdgcn_exp2f(zlo.x), __builtin_amdgcn_exp2f(zlo.y)};     // e^min(z, 0) == ELU'(z)
        const f32x2 ehi = {__builtin_amdgcn_exp2f(zhi.x), __builtin_amdgcn_exp2f(zhi.y)};
        const f32x2 dlo = f32x2{gu.x, gu.y} * elo;
        const f32x2 dhi = f32x2{gu.z, gu.w} * ehi;
        s1lo += dlo;
        s1hi += dhi;
        s2lo = __builtin_elementwise_fma(dlo, __builtin_elementwise_fma(ylo, rslo, nmlo), s2lo);
        s2hi = __builtin_elementwise_fma(dhi, __builtin_elementwise_fma(yhi, rshi, nmhi), s2hi);
        const f32x4 cq0 = *reinterpret_cast<const f32x4*>(xc + (r + u * rl) * CP);      // centred points for G
        xr[0] = cq0.x; xr[1] = cq0.y; xr[2] = cq0.z; xr[3] = cq0.w;
        if constexpr (CP > 4) {
          const f32x4 cq1 = *reinterpret_cast<const f32x4*>(xc + (r + u * rl) * CP + 4);
          xr[4] = cq1.x; xr[5] = cq1.y; xr[6] = cq1.z; xr[7] = cq1.w;
        }
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const f32x2 xv = {xr[c], xr[c]};
          alo[c] = __builtin_elementwise_fma(dlo, xv, alo[c]);
          ahi[c] = __builtin_elementwise_fma(dhi, xv, ahi[c]);
        }
      }
#pragma unroll
      for (int u = 0; u < TRIP; ++u) graw[u] = gnext[u];
    }
    s1 = f32x4{s1lo.x, s1lo.y, s1hi.x, s1hi.y};
    s2 = f32x4{s2lo.x, s2lo.y, s2hi.x, s2hi.y};
#pragma unroll
    for (int c = 0; c < CP; ++c) {
      acc[0][c] = alo[c].x; acc[1][c] = alo[c].y; acc[2][c] = ahi[c].x; acc[3][c] = ahi[c].y;
    }
  } else {
  f32x4 g[4], gn[4];
  load_trip(rlane, g);
  for (int r = rlane; r < nrows; r += 4 * rl) {
    load_trip(r + 4 * rl, gn);           // clamped to the last row past the end
#pragma unroll
    for (int u = 0; u < 4; ++u) {
      if (r + u * rl >= nrows) break;
      const float* xr = xs + (r + u * rl) * CP;
      const f32x4 yv = point_dot<CP>(w, xr);
      f32x4 d;
#pragma unroll
      for (int e = 0; e < 4; ++e)
        d[e] = MODE == 2 ? g[u][e] : g[u][e] * elu_grad_from_pre_t<T>(yv[e] * sc[e] + sh[e]);   // MODE 2: da is already dz
      if (MODE == 0 || MODE == 3) {
        s1 += d;
        s2 += d * ((yv - p0) * p1);
      }
      if (MODE != 0) {
        const f32x4 dy = MODE == 3 ? d : p0 * d + p1 * yv + p2;
        const float* xg = MODE == 3 ? xc + (r + u * rl) * CP : xr;      // MODE 3: the centred points (see the top)
#pragma unroll
        for (int c = 0; c < CP; ++c) {
          const float xv = xg[c];
          acc[0][c] = fmaf(dy.x, xv, acc[0][c]);
          acc[1][c] = fmaf(dy.y, xv, acc[1][c]);
          acc[2][c] = fmaf(dy.z, xv, acc[2][c]);
          acc[3][c] = fmaf(dy.w, xv, acc[3][c]);
        }
      }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) g[u] = gn[u];
  }
  }
  __syncthreads();                       // xs is reused for the row-lane combine
  if (MODE == 0 || MODE == 3) {
    f32x4* red = reinterpret_cast<f32x4*>(xs);      // [2][256]
    red[threadIdx.x] = s1;
    red[256 + threadIdx.x] = s2;
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * cout; o += 256) {
      const int stat = o / cout, cc = o - stat * cout;
      double v = 0.0;
      for (int l = 0; l < rl; ++l) v += (double)red[stat * 256 + l * qpr + (cc >> 2)][cc & 3];
      unsafeAtomicAdd(&stats[((long)(blockIdx.x % nrep) * 2 + stat) * cout + cc], v);
    }
    if (MODE == 3) __syncthreads();      // the combine below reuses the same LDS
  }
  if (MODE != 0) {
    float* red = xs;                     // [rl][cout][CP]
#pragma unroll
    for (int j = 0; j < 4; ++j)
#pragma unroll
      for (int c = 0; c < CP; ++c) red[(rlane * cout + cq * 4 + j) * CP + c] = acc[j][c];
    __syncthreads();
    for (int o = threadIdx.x; o < cout * C; o += 256) {
      const int ch = o / C, c = o - ch * C;
      float v = 0.f;
      for (int l = 0; l < rl; ++l) v += red[(l * cout + ch) * CP + c];
      // MODE 3 spreads the 1 920 workgroups' adds over nrep replicas of G (all of them into ONE [cout, C] image ran at the
      // contended atomic rate: the pass took 0.14 ms for a 252 MB read); the combine kernel sums the replicas
      atomicAdd(&dW[(MODE == 3 ? (long)(blockIdx.x % nrep) * cout * C : 0L) + o], v);
    }
  }
  if (MODE == 0 || MODE == 3) bn_tail_run(tail, threadIdx.x, 256, gridDim.x, &tail_flag);
}

// Second moments of the points: mom[k*MAXC + c] += sum_p x[p][k] x[p][c] (k, c < C), mom[MAXC*MAXC + c] += sum_p x[p][c];
// fp64 atomics, one set per workgroup.  x is 16 B per point: a 4 MB read.  They serve the forward (the first layer's
// BatchNorm statistics follow from them: y = x.W^T is linear) and the one-pass backward.
template <int CP>
__global__ __launch_bounds__(256) void points_moments_kernel(const float* __restrict__ x, int C, long P,
                                                             double* __restrict__ mom) {
  constexpr int NV = CP * (CP + 1) / 2 + CP;           // upper triangle + sums
  // fp64 partial sums (round 4, advisor finding: they were fp32): a product of two fp32 values is exact in fp64, so
  // the moments carry ~1e-16 relative error and var = w^T (x^T x / P) w - (w.m)^2 survives features whose mean is
  // 1e3..1e5 times their deviation.  14 (44) fp64 FMAs per point on a 4 MB read: still a few microseconds.
  __shared__ double red[4][NV];
  double a[NV];
#pragma unroll
  for (int i = 0; i < NV; ++i) a[i] = 0.0;
  for (long r = (long)blockIdx.x * 256 + threadIdx.x; r < P; r += (long)gridDim.x * 256) {
    double xv[CP];
    if (CP == 4 && C == 4) {
      const f32x4 v = load4(x + r * 4);
      xv[0] = v.x; xv[1] = v.y; xv[2] = v.z; xv[3] = v.w;
    } else {
#pragma unroll
      for (int c = 0; c < CP; ++c) xv[c] = c < C ? (double)x[r * C + c] : 0.0;
    }
    int i = 0;
#pragma unroll
    for (int k = 0; k < CP; ++k)
#pragma unroll
      for (int c = k; c < CP; ++c) { a[i] = fma(xv[k], xv[c], a[i]); ++i; }
#pragma unroll
    for (int c = 0; c < CP; ++c) a[i + c] += xv[c];
  }
#pragma unroll
  for (int i = 0; i < NV; ++i) {
    const double v = wave_sum_d(a[i]);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6][i] = v;
  }
  __syncthreads();
  if (threadIdx.x < NV) {
    const double v = red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
    // scatter the triangle entry to both symmetric positions of the MAXC x MAXC layout
    int i = threadIdx.x;
    if (i < CP * (CP + 1) / 2) {
      int k = 0;
      while (i >= CP - k) { i -= CP - k; ++k; }
      const int c = k + i;
      unsafeAtomicAdd(&mom[k * MAXC + c], v);
      if (c != k) unsafeAtomicAdd(&mom[c * MAXC + k], v);
    } else {
      unsafeAtomicAdd(&mom[MAXC * MAXC + (i - CP * (CP + 1) / 2)], v);
    }
  }
}

// The first layer's BatchNorm coefficients straight from the moments (round 3): y = x.W^T, so per output channel
//   sum_p y = W[o] . sum_p x,    sum_p y^2 = W[o]^T (x^T x) W[o]
// -- no pass over the points at all.  One thread per channel, fp64; `t` carries the finalize's arguments (bn_tail.h;
// no arrival counter is involved: this launch is the only writer).
__global__ void pointnet_in_moment_stats_kernel(const double* __restrict__ mom, const float* __restrict__ W, int C,
                                                BnTail t) {
  const int o = blockIdx.x * blockDim.x + threadIdx.x;
  if (o == 0 && t.kind == 1 && t.nbt != nullptr) *t.nbt += 1;
  if (o >= t.ch) return;
  double w[MAXC];
  for (int k = 0; k < MAXC; ++k) w[k] = k < C ? (double)W[o * C + k] : 0.0;
  double s1 = 0.0, s2 = 0.0;
  for (int k = 0; k < C; ++k) {
    s1 += w[k] * mom[MAXC * MAXC + k];
    double row = 0.0;
    for (int c = 0; c < C; ++c) row += mom[k * MAXC + c] * w[c];
    s2 += w[k] * row;
  }
  bn_tail_channel(t, o, s1, s2);
}

// dW[o][c] = c0[o] G'[o][c] + c1[o] sum_k W[o][k] XtX'[k][c] + c2[o] sum_p x'[p][c]     (fp64 combination)
// with x' = x - pivot (the pivot the one-pass kernel centred G on: pivot_mom's sums * pivot_inv_count; null: 0):
//   XtX'[k][c] = sum_p x_k x'_c = XtX[k][c] - pivot_c sum_p x_k,      sum_p x'_c = sum_p x_c - P pivot_c.
// The term pivot_c * sum_p dy[p][o] that completes sum dy x is dropped: sum_p dy = 0 for every channel (BatchNorm's
// backward removes the mean; under SyncBN the pivot is the GLOBAL mean, the same on every rank, so the ranks' terms
// cancel in the gradient all-reduce) -- it only ever held rounding noise times the pivot.
__global__ void pointnet_in_bwd_combine_kernel(const float* __restrict__ G, int nrep, const float* __restrict__ W,
                                               const double* __restrict__ mom, const float* __restrict__ coef,
                                               float* __restrict__ dW, int cout, int C, double P,
                                               const double* __restrict__ pivot_mom, double pivot_inv_count) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= cout * C) return;
  const int o = i / C, c = i - o * C;
  const double piv = pivot_mom ? (double)(float)(pivot_mom[MAXC * MAXC + c] * pivot_inv_count) : 0.0;
  double yx = 0.0, g = 0.0;
  for (int k = 0; k < C; ++k) yx += (double)W[o * C + k] * (mom[k * MAXC + c] - piv * mom[MAXC * MAXC + k]);
  for (int r = 0; r < nrep; ++r) g += (double)G[(long)r * cout * C + i];
  dW[i] = (float)((double)coef[o] * g + (double)coef[cout + o] * yx +
                  (double)coef[2 * cout + o] * (mom[MAXC * MAXC + c] - P * piv));
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
  PCAA_CHECK_ARG(x && W && (y || stats) && P >= 1, "pcaa_pointnet_in_fwd: bad args");
  PCAA_CHECK_ARG(shape_ok(C, cout), "pcaa_pointnet_in_fwd: need C<=%d and cout/4 dividing 256 (C=%d cout=%d)", MAXC, C, cout);
  PCAA_CHECK_ARG(!stats || nrep >= 1, "pcaa_pointnet_in_fwd: bad nrep");
  const unsigned grid = (unsigned)cdiv(P, FWD_ROWS);
  const BnTail tail = stats ? pcaa_take_bn_tail(stats) : BnTail{};
#define LAUNCH_FWD(T, CP)                                                                                          \
  hipLaunchKernelGGL((pointnet_in_fwd_kernel<T, CP>), dim3(grid), dim3(256), 0, as_stream(stream), x, C, W, bias, \
                     (T*)y, P, cout, stats, nrep, tail)
  if (y_dtype == PCAA_F32) { if (C <= 4) LAUNCH_FWD(float, 4); else LAUNCH_FWD(float, 8); }
  else if (y_dtype == PCAA_BF16) { if (C <= 4) LAUNCH_FWD(bf16_t, 4); else LAUNCH_FWD(bf16_t, 8); }
  else { pcaa_set_error("pcaa_pointnet_in_fwd: bad dtype"); return PCAA_ERR_INVALID_ARG; }
#undef LAUNCH_FWD
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

// the point-feature loops are unrolled to CP = 4 (C <= 4: xyz+v, the reference's default) or 8
#define LAUNCH_CP(kern, T, ...)                                                                              \
  do {                                                                                                       \
    if (C <= 4) hipLaunchKernelGGL((kern<T, 4>), dim3(grid), dim3(256), 0, as_stream(stream), __VA_ARGS__);  \
    else hipLaunchKernelGGL((kern<T, 8>), dim3(grid), dim3(256), 0, as_stream(stream), __VA_ARGS__);         \
  } while (0)
#define LAUNCH_BWD(T, MODE, ...)                                                                                          \
  do {                                                                                                                    \
    if (C <= 4) hipLaunchKernelGGL((pointnet_in_bwd_kernel<T, MODE, 4>), dim3(grid), dim3(256), 0, as_stream(stream), __VA_ARGS__); \
    else hipLaunchKernelGGL((pointnet_in_bwd_kernel<T, MODE, 8>), dim3(grid), dim3(256), 0, as_stream(stream), __VA_ARGS__);        \
  } while (0)

extern "C" int pcaa_pointnet_in_apply(const float* x, int C, const float* W, const float* scale, const float* shift,
                                      void* a, int a_dtype, long P, int cout, void* stream) {
  PCAA_CHECK_ARG(x && W && scale && shift && a && P >= 1, "pcaa_pointnet_in_apply: bad args");
  PCAA_CHECK_ARG(shape_ok(C, cout), "pcaa_pointnet_in_apply: need C<=%d and cout/4 dividing 256 (C=%d cout=%d)", MAXC, C, cout);
  const unsigned grid = (unsigned)cdiv(P, FWD_ROWS);
  if (a_dtype == PCAA_F32)
    LAUNCH_CP(pointnet_in_apply_kernel, float, x, C, W, scale, shift, (float*)a, P, cout);
  else if (a_dtype == PCAA_BF16)
    LAUNCH_CP(pointnet_in_apply_kernel, bf16_t, x, C, W, scale, shift, (bf16_t*)a, P, cout);
  else if (a_dtype == PCAA_SPLIT_F16) {
    // fp32 arithmetic, the activation written as its [hi | lo] bf16 image [P, 2 cout]
    if (C <= 4)
      hipLaunchKernelGGL((pointnet_in_apply_kernel<float, 4, true>), dim3(grid), dim3(256), 0, as_stream(stream), x, C, W,
                         scale, shift, (float*)a, P, cout, pcaa_range_flag_ptr());
    else
      hipLaunchKernelGGL((pointnet_in_apply_kernel<float, 8, true>), dim3(grid), dim3(256), 0, as_stream(stream), x, C, W,
                         scale, shift, (float*)a, P, cout, pcaa_range_flag_ptr());
  }
  else { pcaa_set_error("pcaa_pointnet_in_apply: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_apply");
}

extern "C" int pcaa_pointnet_in_bwd_stats(const void* da, int dtype, const float* x, int C, const float* W,
                                          const float* scale, const float* shift, const float* mean,
                                          const float* rstd, double* stats, int nrep, long P, int cout,
                                          void* stream) {
  PCAA_CHECK_ARG(da && x && W && scale && shift && mean && rstd && stats && P >= 1 && nrep >= 1,
                 "pcaa_pointnet_in_bwd_stats: bad args");
  PCAA_CHECK_ARG(shape_ok(C, cout), "pcaa_pointnet_in_bwd_stats: need C<=%d and cout/4 dividing 256 (C=%d cout=%d)", MAXC, C, cout);
  const unsigned grid = (unsigned)cdiv(P, BWD_ROWS);
  const BnTail tail = pcaa_take_bn_tail(stats);
  if (dtype == PCAA_F32)
    LAUNCH_BWD(float, 0, (const float*)da, x, C, W, scale, shift, mean, rstd, nullptr, stats, nrep, nullptr, P, cout, tail);
  else if (dtype == PCAA_BF16)
    LAUNCH_BWD(bf16_t, 0, (const bf16_t*)da, x, C, W, scale, shift, mean, rstd, nullptr, stats, nrep, nullptr, P, cout, tail);
  else { pcaa_set_error("pcaa_pointnet_in_bwd_stats: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_bwd_stats");
}

extern "C" int pcaa_pointnet_in_bwd_wgrad(const void* da, int dtype, const float* x, int C, const float* W,
                                          const float* scale, const float* shift, const float* coef, float* dW,
                                          long P, int cout, int dz_is_pre, void* stream) {
  PCAA_CHECK_ARG(da && x && W && scale && shift && coef && dW && P >= 1, "pcaa_pointnet_in_bwd_wgrad: bad args");
  PCAA_CHECK_ARG(shape_ok(C, cout), "pcaa_pointnet_in_bwd_wgrad: need C<=%d and cout/4 dividing 256 (C=%d cout=%d)", MAXC, C, cout);
  const unsigned grid = (unsigned)cdiv(P, BWD_ROWS);
  if (dtype == PCAA_F32 && !dz_is_pre)
    LAUNCH_BWD(float, 1, (const float*)da, x, C, W, scale, shift, nullptr, nullptr, coef, nullptr, 1, dW, P, cout, BnTail{});
  else if (dtype == PCAA_F32)
    LAUNCH_BWD(float, 2, (const float*)da, x, C, W, scale, shift, nullptr, nullptr, coef, nullptr, 1, dW, P, cout, BnTail{});
  else if (dtype == PCAA_BF16 && !dz_is_pre)
    LAUNCH_BWD(bf16_t, 1, (const bf16_t*)da, x, C, W, scale, shift, nullptr, nullptr, coef, nullptr, 1, dW, P, cout, BnTail{});
  else if (dtype == PCAA_BF16)
    LAUNCH_BWD(bf16_t, 2, (const bf16_t*)da, x, C, W, scale, shift, nullptr, nullptr, coef, nullptr, 1, dW, P, cout, BnTail{});
  else { pcaa_set_error("pcaa_pointnet_in_bwd_wgrad: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_bwd_wgrad");
}

/* One-pass backward of the recompute layer (round 3): statistics AND G = dz^T . x from one read of da
 * (pointnet_in_bwd_kernel MODE 3), then pcaa_pointnet_in_bwd_combine with the points' moments. */
extern "C" int pcaa_pointnet_in_bwd_onepass(const void* da, int dtype, const float* x, int C, const float* W,
                                            const float* scale, const float* shift, const float* mean,
                                            const float* rstd, double* stats, int nrep, float* G, long P, int cout,
                                            const double* pivot_mom, double pivot_inv_count, void* stream) {
  PCAA_CHECK_ARG(da && x && W && scale && shift && mean && rstd && stats && G && P >= 1 && nrep >= 1,
                 "pcaa_pointnet_in_bwd_onepass: bad args");
  PCAA_CHECK_ARG(shape_ok(C, cout), "pcaa_pointnet_in_bwd_onepass: need C<=%d and cout/4 dividing 256 (C=%d cout=%d)", MAXC, C, cout);
  const unsigned grid = (unsigned)cdiv(P, BWD_ROWS);
  const BnTail tail = pcaa_take_bn_tail(stats);
  if (dtype == PCAA_F32)
    LAUNCH_BWD(float, 3, (const float*)da, x, C, W, scale, shift, mean, rstd, nullptr, stats, nrep, G, P, cout, tail,
               pivot_mom, pivot_inv_count);
  else if (dtype == PCAA_BF16)
    LAUNCH_BWD(bf16_t, 3, (const bf16_t*)da, x, C, W, scale, shift, mean, rstd, nullptr, stats, nrep, G, P, cout, tail,
               pivot_mom, pivot_inv_count);
  else { pcaa_set_error("pcaa_pointnet_in_bwd_onepass: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_bwd_onepass");
}

extern "C" int pcaa_points_moments_size(void) { return MAXC * MAXC + MAXC; }

extern "C" int pcaa_points_moments(const float* x, int C, long P, double* mom, void* stream) {
  PCAA_CHECK_ARG(x && mom && C >= 1 && C <= MAXC && P >= 1, "pcaa_points_moments: bad args");
  long g = cdiv(P, 256 * 4);
  if (g > 512) g = 512;
  if (C <= 4) hipLaunchKernelGGL(points_moments_kernel<4>, dim3((unsigned)g), dim3(256), 0, as_stream(stream), x, C, P, mom);
  else hipLaunchKernelGGL(points_moments_kernel<8>, dim3((unsigned)g), dim3(256), 0, as_stream(stream), x, C, P, mom);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_points_moments");
}

/* BatchNorm coefficients of the first layer from the points' moments: takes the finalize armed by pcaa_bn_tail_arm_fwd
 * (its `stats` argument must be `mom`: the armed tail is matched by that pointer) and runs it on sums derived from
 * the moments instead of a statistics pass over the points. */
extern "C" int pcaa_pointnet_in_moment_stats(const double* mom, const float* W, int C, int cout, void* stream) {
  PCAA_CHECK_ARG(mom && W && C >= 1 && C <= MAXC && cout >= 1, "pcaa_pointnet_in_moment_stats: bad args");
  const BnTail t = pcaa_take_bn_tail(mom);
  PCAA_CHECK_ARG(t.kind == 1 && t.ch == cout, "pcaa_pointnet_in_moment_stats: arm the forward finalize on `mom` first "
                 "(pcaa_bn_tail_arm_fwd)");
  hipLaunchKernelGGL(pointnet_in_moment_stats_kernel, dim3((unsigned)cdiv(cout, 256)), dim3(256), 0, as_stream(stream), mom,
                     W, C, t);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_moment_stats");
}

extern "C" int pcaa_pointnet_in_bwd_combine(const float* G, int nrep, const float* W, const double* mom,
                                            const float* coef, float* dW, int cout, int C, long P,
                                            const double* pivot_mom, double pivot_inv_count, void* stream) {
  PCAA_CHECK_ARG(G && W && mom && coef && dW && cout >= 1 && C >= 1 && C <= MAXC && nrep >= 1,
                 "pcaa_pointnet_in_bwd_combine: bad args");
  hipLaunchKernelGGL(pointnet_in_bwd_combine_kernel, dim3((unsigned)cdiv((long)cout * C, 256)), dim3(256), 0,
                     as_stream(stream), G, nrep, W, mom, coef, dW, cout, C, (double)P, pivot_mom, pivot_inv_count);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pointnet_in_bwd_combine");
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
