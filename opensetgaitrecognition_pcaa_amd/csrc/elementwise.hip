// BatchNorm(+ELU) forward/backward pieces, pooling, bias/activation helpers,
// im2col for the causal dilated convolutions, Adam.  All HBM-bound: every
// thread moves 4 consecutive channels (16 B fp32 / 8 B bf16) of one row, rows
// of a tile are walked by the workgroup so per-channel partial sums stay in
// registers and leave through one fp64 atomic per channel per workgroup.
#include <stdarg.h>

#include "common.h"
#include "bn_tail.h"

// ---------------------------------------------------------------- error string
static thread_local char g_err[512] = "";
void pcaa_set_error(const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}
extern "C" const char* pcaa_last_error(void) { return g_err; }
// the device flag the split-image producers raise when a value leaves fp16's range (common.h, split_guard)
static thread_local int* g_range_flag = nullptr;
int* pcaa_range_flag_ptr() { return g_range_flag; }
extern "C" int pcaa_set_range_flag(int* dev_flag) { g_range_flag = dev_flag; return PCAA_OK; }
extern "C" int pcaa_abi_version(void) { return PCAA_ABI_VERSION; }

namespace {

constexpr int ROWS_PER_BLOCK = 128;  // rows of a [rows, ch] tensor one workgroup reduces

inline bool ch_ok(int ch) {
  // ch/4 lanes per row must tile a 256-thread workgroup
  if (ch < 4 || ch > 1024 || (ch & 3)) return false;
  const int q = ch >> 2;
  return (256 % q) == 0;
}

// sum of the nrep replicas of one channel's two statistics; 8 replicas (16 loads) in flight per trip --
// the one-at-a-time loop made these single-workgroup kernels 8-10 us of pure load latency
__device__ __forceinline__ void replica_sums(const double* __restrict__ stats, int nrep, int ch, int c, double& s1,
                                             double& s2) {
  int r = 0;
  for (; r + 8 <= nrep; r += 8) {
    double a[8], b[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) {
      a[u] = stats[((long)(r + u) * 2 + 0) * ch + c];
      b[u] = stats[((long)(r + u) * 2 + 1) * ch + c];
    }
#pragma unroll
    for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += b[u]; }
  }
  for (; r < nrep; ++r) {
    s1 += stats[((long)r * 2 + 0) * ch + c];
    s2 += stats[((long)r * 2 + 1) * ch + c];
  }
}

// ---------------------------------------------------------------- BN finalize (forward)
__global__ void bn_finalize_kernel(const double* __restrict__ stats, int nrep, double inv_count,
                                   double unbias, const float* __restrict__ lin_bias,
                                   const float* __restrict__ gamma, const float* __restrict__ beta,
                                   float* running_mean, float* running_var, long long* nbt,
                                   float momentum, float eps, float* scale, float* shift,
                                   float* mean_out, float* rstd_out, int ch) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c == 0 && nbt) *nbt += 1;
  if (c >= ch) return;
  double s1 = 0.0, s2 = 0.0;
  replica_sums(stats, nrep, ch, c, s1, s2);
  const double m0 = s1 * inv_count;                 // mean of the bias-free linear output
  double var = s2 * inv_count - m0 * m0;            // biased variance
  if (var < 0.0) var = 0.0;
  // The linear layer's bias is NOT in the stored pre-BN tensor y (= the bias-free accumulator):
  // BatchNorm subtracts the batch mean, so in train mode the bias cancels exactly --
  // BN(acc + b) = gamma*(acc - mean(acc))*rstd + beta.  It only enters the running mean.
  const double mean = m0 + (lin_bias ? (double)lin_bias[c] : 0.0);
  const double rstd = 1.0 / sqrt(var + (double)eps);
  const float sc = (float)((double)gamma[c] * rstd);
  scale[c] = sc;
  shift[c] = (float)((double)beta[c] - m0 * (double)gamma[c] * rstd);
  mean_out[c] = (float)m0;
  rstd_out[c] = (float)rstd;
  if (running_mean) {
    running_mean[c] = (1.f - momentum) * running_mean[c] + momentum * (float)mean;
    running_var[c] = (1.f - momentum) * running_var[c] + momentum * (float)(var * unbias);
  }
}

__global__ void bn_eval_coeffs_kernel(const float* gamma, const float* beta, const float* rm,
                                      const float* rv, const float* lin_bias, float eps, float* scale,
                                      float* shift, int ch) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ch) return;
  const float rstd = 1.f / sqrtf(rv[c] + eps);
  const float sc = gamma[c] * rstd;
  scale[c] = sc;
  // y is stored bias-free: z = (acc + b - running_mean) * sc + beta
  shift[c] = beta[c] + ((lin_bias ? lin_bias[c] : 0.f) - rm[c]) * sc;
}

// ---------------------------------------------------------------- a = ELU(y*scale+shift)
// Flat quad index, grid-stride loop.  The host rounds the grid so that the stride (gridDim*256
// quads) is a multiple of the quads per row: a thread then stays on ONE channel quad for its whole
// life -- its coefficients live in registers and there is no per-element index division (the
// 64-bit q % qpr, q / qpr of the first version cost more VALU time than the HBM stream).
template <typename T>
__global__ __launch_bounds__(256) void bn_act_fwd_kernel(const T* __restrict__ y, T* __restrict__ a,
                                                         const float* __restrict__ scale,
                                                         const float* __restrict__ shift,
                                                         unsigned nquads, unsigned qpr) {
  const unsigned q0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  const unsigned c = (q0 % qpr) << 2;
  const f32x4 sc = load4(scale + c), sh = load4(shift + c);
  for (unsigned q = q0; q < nquads; q += stride) {
    const f32x4 v = load4(y + (size_t)q * 4);
    f32x4 o;
    o.x = elu_t<T>(v.x * sc.x + sh.x);
    o.y = elu_t<T>(v.y * sc.y + sh.y);
    o.z = elu_t<T>(v.z * sc.z + sh.z);
    o.w = elu_t<T>(v.w * sc.w + sh.w);
    store4(a + (size_t)q * 4, o);
  }
}

// fp32 y -> the [hi | lo] fp16 image of a = ELU(y*scale+shift) (split parity mode: a only feeds GEMMs)
__global__ __launch_bounds__(256) void bn_act_fwd_split_kernel(const float* __restrict__ y, split_t* __restrict__ a,
                                                               const float* __restrict__ scale,
                                                               const float* __restrict__ shift, unsigned nquads,
                                                               unsigned qpr, float img_scale, int* oflow) {
  const unsigned q0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  const unsigned c = (q0 % qpr) << 2, ch = qpr << 2, rstep = stride / qpr;
  unsigned r = q0 / qpr;
  const f32x4 sc = load4(scale + c), sh = load4(shift + c);
  for (unsigned q = q0; q < nquads; q += stride, r += rstep) {
    const f32x4 v = load4(y + (size_t)q * 4);
    store4_split(a, r, ch, c, f32x4{elu_f(v.x * sc.x + sh.x), elu_f(v.y * sc.y + sh.y), elu_f(v.z * sc.z + sh.z),
                                     elu_f(v.w * sc.w + sh.w)}, img_scale, oflow);
  }
}

// fp32 [rows, ch] -> its [hi | lo] fp16 image [rows, 2 ch] (of value * img_scale); transpose != 0: the image of the TRANSPOSED matrix, [ch, 2 rows]
// (weights: a few MB)
__global__ void split_f16_kernel(const float* __restrict__ src, split_t* __restrict__ dst, long rows, int ch,
                                  int transpose, float img_scale, int* oflow) {
  const long n = rows * ch;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    const long r = i / ch;
    const int c = (int)(i - r * ch);
    const float v = split_guard(src[i] * img_scale, oflow);
    const split_t hi = (split_t)v, lo = (split_t)(v - (float)hi);
    if (transpose) {
      dst[(long)c * 2 * rows + r] = hi;
      dst[(long)c * 2 * rows + rows + r] = lo;
    } else {
      dst[r * 2 * ch + c] = hi;
      dst[r * 2 * ch + ch + c] = lo;
    }
  }
}

// ---------------------------------------------------------------- mean-pool of ELU(BN(y)) over group_rows
// one workgroup per (group, 1024-channel slab): thread = channel quad x row lane
// TRAIN: also emits, per (group, channel), E1 = sum_r ELU'(z) and E2 = sum_r ELU'(z) * yhat.  The
// gradient that comes back is one value per (group, channel), so the BatchNorm-backward statistics
// of this layer are sum_g dpool * E1 and sum_g dpool * E2 (pcaa_bn_pool_bwd_stats): a pass over
// [groups, ch] instead of a second read of the whole pre-activation tensor.
template <typename T, bool TRAIN>
__global__ __launch_bounds__(256) void bn_act_meanpool_kernel(const T* __restrict__ y,
                                                              const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ mean,
                                                              const float* __restrict__ rstd,
                                                              float* __restrict__ pooled, float* __restrict__ e1,
                                                              float* __restrict__ e2, int group_rows, int ch) {
  __shared__ f32x4 red[TRAIN ? 3 : 1][256];
  const int qpr = ch >> 2;              // quads per row
  const int rl = 256 / qpr;             // row lanes
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const long g = blockIdx.x;
  const T* base = y + g * (long)group_rows * ch + cq * 4;
  const f32x4 sc = load4(scale + cq * 4), sh = load4(shift + cq * 4);
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = mu;
  if (TRAIN) { mu = load4(mean + cq * 4); rs = load4(rstd + cq * 4); }
  f32x4 acc = {0.f, 0.f, 0.f, 0.f}, a1 = acc, a2 = acc;
  // (one 8-B load per row and thread; batching 8 rows per trip measured SLOWER -- 184 vs 140 us -- the 32 extra
  // live registers cost more occupancy than the deeper queue gains)
  for (int r = rlane; r < group_rows; r += rl) {
    const f32x4 v = load4(base + (long)r * ch);
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      const float z = v[e] * sc[e] + sh[e];
      const float a = elu_t<T>(z);
      acc[e] += a;
      if (TRAIN) {
        const float d = z > 0.f ? 1.f : a + 1.f;      // ELU'(z) = e^z = ELU(z) + 1 for z <= 0
        a1[e] += d;
        a2[e] += d * ((v[e] - mu[e]) * rs[e]);
      }
    }
  }
  red[0][threadIdx.x] = acc;
  if (TRAIN) { red[1][threadIdx.x] = a1; red[2][threadIdx.x] = a2; }
  __syncthreads();
  if (rlane == 0) {
    for (int l = 1; l < rl; ++l) {
      acc += red[0][l * qpr + cq];
      if (TRAIN) { a1 += red[1][l * qpr + cq]; a2 += red[2][l * qpr + cq]; }
    }
    const float inv = 1.f / (float)group_rows;
    acc *= inv;
    store4(pooled + g * (long)ch + cq * 4, acc);
    if (TRAIN) {
      store4(e1 + g * (long)ch + cq * 4, a1);
      store4(e2 + g * (long)ch + cq * 4, a2);
    }
  }
}

// bf16 storage, whole rows per workgroup (ch = 1024: 256 threads x one channel quad): the streaming form of the kernel
// above.  One workgroup per group means ~1900 short concurrent streams and reaches 3.3 TB/s (tools/elementwise_lab.py);
// here ~4 workgroups per CU walk the groups (measured: 512 / 1024 / 1920 workgroups 0.136 / 0.118 / 0.123 ms against 0.157), every lane keeps two batches of UB rows in flight (the next batch is
// requested before the current one is consumed), and the arithmetic runs on column pairs with packed fp32
// instructions so that it fits beside the stream: with e = exp(min(z, 0)), ELU(z) = max(z, 0) + e - 1 and ELU'(z) = e --
// one exponential (exp2, log2e folded into a second affine pair), no compare / select; the -1 leaves the loop.
template <bool TRAIN>
__global__ __launch_bounds__(256) void bn_act_meanpool_stream_kernel(const bf16_t* __restrict__ y,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     const float* __restrict__ mean,
                                                                     const float* __restrict__ rstd,
                                                                     float* __restrict__ pooled, float* __restrict__ e1,
                                                                     float* __restrict__ e2, long ngroups, int group_rows,
                                                                     int ch) {
  typedef float f32x2 __attribute__((ext_vector_type(2)));
  constexpr int UB = 4;
  constexpr float kLog2e = 1.4426950408889634f;
  const int cq = threadIdx.x;                         // ch == 1024: one channel quad per thread, every row
  const f32x4 sc = load4(scale + cq * 4), sh = load4(shift + cq * 4);
  f32x4 mu = {0.f, 0.f, 0.f, 0.f}, rs = mu;
  if (TRAIN) { mu = load4(mean + cq * 4); rs = load4(rstd + cq * 4); }
  const f32x2 sclo = {sc.x, sc.y}, schi = {sc.z, sc.w}, shlo = {sh.x, sh.y}, shhi = {sh.z, sh.w};
  const f32x2 sl2 = sclo * kLog2e, sh2 = schi * kLog2e, tl2 = shlo * kLog2e, th2 = shhi * kLog2e;
  const f32x2 rslo = {rs.x, rs.y}, rshi = {rs.z, rs.w};
  const f32x2 nmlo = {-mu.x * rs.x, -mu.y * rs.y}, nmhi = {-mu.z * rs.z, -mu.w * rs.w};
  const f32x2 zero = {0.f, 0.f};
  const int nb = group_rows / UB;                     // host-checked: group_rows % UB == 0
  for (long g = blockIdx.x; g < ngroups; g += gridDim.x) {
    const bf16_t* base = y + g * (long)group_rows * ch + cq * 4;
    f32x2 plo = zero, phi = zero, elo = zero, ehi = zero, qlo = zero, qhi = zero;
    uint2 buf[2][UB];
    auto fetch = [&](int b, int batch) {
#pragma unroll
      for (int u = 0; u < UB; ++u) buf[b][u] = *reinterpret_cast<const uint2*>(base + (long)(batch * UB + u) * ch);
    };
    auto consume = [&](int b) {
#pragma unroll
      for (int u = 0; u < UB; ++u) {
        const uint2 w = buf[b][u];
        const f32x2 vlo = {__uint_as_float(w.x << 16), __uint_as_float(w.x & 0xffff0000u)};
        const f32x2 vhi = {__uint_as_float(w.y << 16), __uint_as_float(w.y & 0xffff0000u)};
        const f32x2 zlo = __builtin_elementwise_fma(vlo, sclo, shlo), zhi = __builtin_elementwise_fma(vhi, schi, shhi);
        const f32x2 mlo = __builtin_elementwise_min(__builtin_elementwise_fma(vlo, sl2, tl2), zero);
        const f32x2 mhi = __builtin_elementwise_min(__builtin_elementwise_fma(vhi, sh2, th2), zero);
        const f32x2 xlo = {__builtin_amdgcn_exp2f(mlo.x), __builtin_amdgcn_exp2f(mlo.y)};
        const f32x2 xhi = {__builtin_amdgcn_exp2f(mhi.x), __builtin_amdgcn_exp2f(mhi.y)};
        plo += __builtin_elementwise_max(zlo, zero) + xlo;
        phi += __builtin_elementwise_max(zhi, zero) + xhi;
        if (TRAIN) {
          elo += xlo;
          ehi += xhi;
          qlo = __builtin_elementwise_fma(xlo, __builtin_elementwise_fma(vlo, rslo, nmlo), qlo);
          qhi = __builtin_elementwise_fma(xhi, __builtin_elementwise_fma(vhi, rshi, nmhi), qhi);
        }
      }
    };
    fetch(0, 0);
    for (int batch = 0; batch < nb; batch += 2) {
      if (batch + 1 < nb) fetch(1, batch + 1);
      consume(0);
      if (batch + 2 < nb) fetch(0, batch + 2);
      if (batch + 1 < nb) consume(1);
    }
    const float n = (float)group_rows, inv = 1.f / n;
    store4(pooled + g * (long)ch + cq * 4, f32x4{(plo.x - n) * inv, (plo.y - n) * inv, (phi.x - n) * inv, (phi.y - n) * inv});
    if (TRAIN) {
      store4(e1 + g * (long)ch + cq * 4, f32x4{elo.x, elo.y, ehi.x, ehi.y});
      store4(e2 + g * (long)ch + cq * 4, f32x4{qlo.x, qlo.y, qhi.x, qhi.y});
    }
  }
}

// stats[0][c] += sum_g dpool[g][c]*pool_scale*E1[g][c];  stats[1][c] += sum_g dpool[g][c]*pool_scale*E2[g][c]
// grid: one block per 32 groups, thread = channel (strided over ch)
__global__ __launch_bounds__(256) void bn_pool_bwd_stats_kernel(const float* __restrict__ dpool,
                                                                const float* __restrict__ e1,
                                                                const float* __restrict__ e2, float pool_scale,
                                                                double* __restrict__ stats, int nrep, long groups,
                                                                int ch, int gpb, BnTail tail) {
  __shared__ int tail_flag;
  // thread = channel quad; the block's gpb groups in trips of 8 with all 24 loads in flight (the first version
  // walked them one dependent load at a time: 150 us beside the side-stream Adam; 32 groups per block left the
  // 1920-group PointNet case with 60 workgroups on 256 CUs: 50 us on the critical path)
  const long g0 = (long)blockIdx.x * gpb, g1 = min(groups, g0 + gpb);
  for (int c = threadIdx.x * 4; c < ch; c += 1024) {
    double s1[4] = {0.0, 0.0, 0.0, 0.0}, s2[4] = {0.0, 0.0, 0.0, 0.0};
    for (long g = g0; g < g1; g += 8) {
      f32x4 d[8], a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        const long gg = min(g + u, g1 - 1);
        d[u] = load4(dpool + gg * ch + c);
        a[u] = load4(e1 + gg * ch + c);
        b[u] = load4(e2 + gg * ch + c);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        if (g + u >= g1) break;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          const float dv = d[u][e] * pool_scale;
          s1[e] += (double)(dv * a[u][e]);
          s2[e] += (double)(dv * b[u][e]);
        }
      }
    }
#pragma unroll
    for (int e = 0; e < 4; ++e) {
      unsafeAtomicAdd(&stats[((long)(blockIdx.x % nrep) * 2 + 0) * ch + c + e], s1[e]);
      unsafeAtomicAdd(&stats[((long)(blockIdx.x % nrep) * 2 + 1) * ch + c + e], s2[e]);
    }
  }
  bn_tail_run(tail, threadIdx.x, 256, gridDim.x, &tail_flag);
}

// ---------------------------------------------------------------- dz = da * ELU'(z) + BN-backward statistics
template <typename T, bool POOL>
__global__ __launch_bounds__(256) void bn_act_bwd_dz_kernel(const T* __restrict__ da,
                                                            const float* __restrict__ dpool,
                                                            int group_rows, float pool_scale,
                                                            const T* __restrict__ y, T* __restrict__ dz,
                                                            const float* __restrict__ scale,
                                                            const float* __restrict__ shift,
                                                            const float* __restrict__ mean,
                                                            const float* __restrict__ rstd,
                                                            double* __restrict__ stats, int nrep,
                                                            long rows, int ch) {
  __shared__ f32x4 red[2][256];
  const int qpr = ch >> 2;
  const int rl = 256 / qpr;
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const int c = cq * 4;
  const f32x4 sc = load4(scale + c), sh = load4(shift + c), mu = load4(mean + c), rs = load4(rstd + c);
  const long r0 = (long)blockIdx.x * ROWS_PER_BLOCK;
  const long r1 = min(rows, r0 + ROWS_PER_BLOCK);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (long r = r0 + rlane; r < r1; r += rl) {
    const f32x4 yv = load4(y + r * ch + c);
    f32x4 g;
    if (POOL) {
      g = load4(dpool + (size_t)((unsigned)r / (unsigned)group_rows) * ch + c);   // rows < 2^31 (host-checked)
      g.x *= pool_scale; g.y *= pool_scale; g.z *= pool_scale; g.w *= pool_scale;
    } else {
      g = load4(da + r * ch + c);
    }
    f32x4 d;
    d.x = g.x * elu_grad_from_pre_t<T>(yv.x * sc.x + sh.x);
    d.y = g.y * elu_grad_from_pre_t<T>(yv.y * sc.y + sh.y);
    d.z = g.z * elu_grad_from_pre_t<T>(yv.z * sc.z + sh.z);
    d.w = g.w * elu_grad_from_pre_t<T>(yv.w * sc.w + sh.w);
    if (dz) store4(dz + r * ch + c, d);       // dz == NULL: statistics-only pass
    s1.x += d.x; s1.y += d.y; s1.z += d.z; s1.w += d.w;
    s2.x += d.x * ((yv.x - mu.x) * rs.x);
    s2.y += d.y * ((yv.y - mu.y) * rs.y);
    s2.z += d.z * ((yv.z - mu.z) * rs.z);
    s2.w += d.w * ((yv.w - mu.w) * rs.w);
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  // 2*ch outputs; thread t handles stat = t / (ch/... ) -- walk all (stat, channel) pairs
  for (int o = threadIdx.x; o < 2 * ch; o += 256) {
    const int stat = o / ch, cc = o - stat * ch;
    const int q = cc >> 2, e = cc & 3;
    double v = 0.0;
    for (int l = 0; l < rl; ++l) v += (double)red[stat][l * qpr + q][e];
    unsafeAtomicAdd(&stats[((long)(blockIdx.x % nrep) * 2 + stat) * ch + cc], v);
  }
}

__global__ void bn_bwd_finalize_kernel(const double* __restrict__ stats, int nrep, double inv_count,
                                       const float* __restrict__ gamma, const float* __restrict__ mean,
                                       const float* __restrict__ rstd, float* coef, float* dgamma,
                                       float* dbeta, int ch) {
  const int c = blockIdx.x * blockDim.x + threadIdx.x;
  if (c >= ch) return;
  double s1 = 0.0, s2 = 0.0;
  replica_sums(stats, nrep, ch, c, s1, s2);
  if (dbeta) dbeta[c] = (float)s1;
  if (dgamma) dgamma[c] = (float)s2;
  const double c1 = s1 * inv_count, c2 = s2 * inv_count;
  const double rs = rstd[c], mu = mean[c];
  const double g = (double)gamma[c] * rs;
  coef[0 * ch + c] = (float)g;
  coef[1 * ch + c] = (float)(-g * c2 * rs);
  coef[2 * ch + c] = (float)(-g * c1 + g * c2 * rs * mu);
}

template <typename T>
__global__ __launch_bounds__(256) void bn_bwd_dy_kernel(const T* __restrict__ dz, const T* __restrict__ y,
                                                        T* __restrict__ dy, const float* __restrict__ coef,
                                                        unsigned nquads, unsigned qpr, unsigned ch) {
  // column-invariant grid (see bn_act_fwd_kernel): the thread's coefficients live in registers
  const unsigned q0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  const unsigned c = (q0 % qpr) << 2;
  const f32x4 k0 = load4(coef + c), k1 = load4(coef + ch + c), k2 = load4(coef + 2 * ch + c);
  for (unsigned q = q0; q < nquads; q += stride) {
    const f32x4 d = load4(dz + (size_t)q * 4), yv = load4(y + (size_t)q * 4);
    store4(dy + (size_t)q * 4, k0 * d + k1 * yv + k2);
  }
}

// dy = c0*dz + c1*y + c2 written as its [hi | lo] fp16 image (split parity mode: the dgrad above already formed dz
// and its statistics in its epilogue, pcaa_gemm_dgrad_bn_split3; dy only feeds the two products below)
__global__ __launch_bounds__(256) void bn_bwd_dy_split_kernel(const float* __restrict__ dz, const float* __restrict__ y,
                                                              split_t* __restrict__ dy_img,
                                                              const float* __restrict__ coef, unsigned nquads,
                                                              unsigned qpr, unsigned ch, float img_scale, int* oflow) {
  const unsigned q0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  const unsigned c = (q0 % qpr) << 2;
  const unsigned rstep = stride / qpr;
  unsigned rr = q0 / qpr;
  const f32x4 k0 = load4(coef + c), k1 = load4(coef + ch + c), k2 = load4(coef + 2 * ch + c);
  for (unsigned q = q0; q < nquads; q += stride) {
    const f32x4 d = load4(dz + (size_t)q * 4), yv = load4(y + (size_t)q * 4);
    store4_split(dy_img, rr, ch, c, k0 * d + k1 * yv + k2, img_scale, oflow);
    rr += rstep;
  }
}

// ---------------------------------------------------------------- small fp32 helpers
__global__ void bias_act_kernel(float* y, const float* __restrict__ bias, int act, long n, int cols) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x) {
    float v = y[i] + (bias ? bias[i % cols] : 0.f);
    y[i] = act == PCAA_ACT_ELU ? elu_f(v) : v;
  }
}

__global__ void elu_bwd_from_out_kernel(const float* __restrict__ da, const float* __restrict__ a,
                                        float* dz, long n) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    dz[i] = da[i] * elu_grad_from_out(a[i]);
}

// out[c] = sum_r x[r][c]; block = 256 threads = 64 columns x 4 row lanes
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, float* out, long rows, int cols) {
  __shared__ float red[256];
  const int cl = threadIdx.x & 63, rlane = threadIdx.x >> 6;
  const int c = blockIdx.x * 64 + cl;
  float acc = 0.f;
  if (c < cols) {
    long r = rlane;
    for (; r + 28 < rows; r += 32) {      // 8 rows in flight per trip (these launches are pure load latency)
      float v[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) v[u] = x[(r + 4 * u) * cols + c];
#pragma unroll
      for (int u = 0; u < 8; ++u) acc += v[u];
    }
    for (; r < rows; r += 4) acc += x[r * cols + c];
  }
  red[threadIdx.x] = acc;
  __syncthreads();
  if (rlane == 0 && c < cols) out[c] = red[cl] + red[64 + cl] + red[128 + cl] + red[192 + cl];
}

__global__ __launch_bounds__(1024) void sum_kernel(const float* __restrict__ x, long n, float scale, float* out) {
  __shared__ double red[16];
  double acc = 0.0;
  for (long i = threadIdx.x; i < n; i += 1024) acc += (double)x[i];
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0) {
    double t = 0.0;
    for (int i = 0; i < 16; ++i) t += red[i];
    out[0] = (float)(t * (double)scale);
  }
}

__global__ __launch_bounds__(64) void rowsum_kernel(const float* __restrict__ x, float* out, int cols, float scale) {
  const long r = blockIdx.x;
  double acc = 0.0;
  for (int c = threadIdx.x; c < cols; c += 64) acc += (double)x[r * cols + c];
  acc = wave_sum_d(acc);
  if (threadIdx.x == 0) out[r] = (float)(acc * (double)scale);
}

__global__ void scale_dev_kernel(const float* __restrict__ x, const float* __restrict__ s, float* out, long n) {
  const float f = s[0];
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = x[i] * f;
}

__global__ void scale_rows_kernel(const float* __restrict__ x, const float* __restrict__ s, float* out,
                                  long n, int cols) {
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (long)gridDim.x * blockDim.x)
    out[i] = x[i] * s[i / cols];
}

__global__ void prior_sample_kernel(const float* __restrict__ z0, const float* __restrict__ means,
                                    const long long* __restrict__ gt, int B, int K, int D, float* z,
                                    float* onehot) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < B * D) {
    const int b = i / D, d = i - b * D;
    z[i] = z0[i] + means[gt[b] * D + d];
  }
  if (i < B * K) {
    const int b = i / K, k = i - b * K;
    onehot[i] = (gt[b] == k) ? 1.f : 0.f;
  }
}

__global__ void pack_points_kernel(const float* __restrict__ src, long sb, long sc, long st, long sn,
                                   float* __restrict__ dst, int B, int C, int T, int N) {
  const long total = (long)B * T * N * C;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int c = (int)(i % C);
    long r = i / C;
    const int n = (int)(r % N); r /= N;
    const int t = (int)(r % T);
    const long b = r / T;
    dst[i] = src[b * sb + c * sc + t * st + n * sn];
  }
}

// col[(b,t)][ci*3+tap] = a[b][t-(2-tap)*d][ci]  (zero for negative time)
__global__ void dtc_im2col_kernel(const float* __restrict__ a, float* __restrict__ col, int B, int T,
                                  int Cin, int d) {
  const long total = (long)B * T * Cin * 3;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int tap = (int)(i % 3);
    long r = i / 3;
    const int ci = (int)(r % Cin); r /= Cin;
    const int t = (int)(r % T);
    const long b = r / T;
    const int ts = t - (2 - tap) * d;
    col[i] = ts >= 0 ? a[(b * T + ts) * Cin + ci] : 0.f;
  }
}

// da[b][t][ci] = sum_tap dcol[b][t+(2-tap)*d][ci*3+tap]  (while t+(2-tap)*d < T)
__global__ void dtc_col2im_kernel(const float* __restrict__ dcol, float* __restrict__ da, int B, int T,
                                  int Cin, int d) {
  const long total = (long)B * T * Cin;
  for (long i = (long)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (long)gridDim.x * blockDim.x) {
    const int ci = (int)(i % Cin);
    long r = i / Cin;
    const int t = (int)(r % T);
    const long b = r / T;
    float acc = 0.f;
#pragma unroll
    for (int tap = 0; tap < 3; ++tap) {
      const int tt = t + (2 - tap) * d;
      if (tt < T) acc += dcol[((b * T + tt) * Cin + ci) * 3 + tap];
    }
    da[i] = acc;
  }
}

// ---------------------------------------------------------------- Adam
// HBM-bound: 4 reads + 3 writes of 16 B per quad.  U quads per thread per trip keep U x 64 B per
// lane in flight, so a SMALL grid (1-2 workgroups per CU) still saturates HBM -- which is what
// lets the decoder's update run on a side stream beside the latency-bound temporal-conv / head
// kernels without taking their wave slots (a 4096-block grid made those 2-7x slower).
template <int U, typename TG = float>
__global__ __launch_bounds__(256) void adam_kernel(float* __restrict__ p, const TG* __restrict__ g,
                                                   float* __restrict__ m, float* __restrict__ v, long n,
                                                   float b1, float b2, float eps, float step_size,
                                                   float inv_bc2_sqrt, float grad_scale,
                                                   const float* __restrict__ coef) {
  if (coef) {            // step-dependent scalars kept on the device (hipGraph replay: no host argument changes)
    step_size = coef[0];
    inv_bc2_sqrt = coef[1];
  }
  const long nq = n >> 2;
  const long stride = (long)gridDim.x * 256;
  for (long q0 = (long)blockIdx.x * 256 + threadIdx.x; q0 < nq; q0 += stride * U) {
    f32x4 pv[U], gv[U], mv[U], vv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long q = q0 + u * stride;
      if (q < nq) { pv[u] = load4(p + q * 4); gv[u] = load4(g + q * 4); mv[u] = load4(m + q * 4); vv[u] = load4(v + q * 4); }   // g: fp32 or bf16 (load4 widens)
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long q = q0 + u * stride;
      if (q >= nq) break;
#pragma unroll
      for (int e = 0; e < 4; ++e) {
        float pe = pv[u][e], me = mv[u][e], ve = vv[u][e];
        adam_update(pe, me, ve, gv[u][e] * grad_scale, b1, b2, eps, step_size, inv_bc2_sqrt);
        pv[u][e] = pe; mv[u][e] = me; vv[u][e] = ve;
      }
      store4(p + q * 4, pv[u]);
      store4(m + q * 4, mv[u]);
      store4(v + q * 4, vv[u]);
    }
  }
  // tail (n % 4)
  if (blockIdx.x == 0 && threadIdx.x < (n & 3)) {
    const long i = (nq << 2) + threadIdx.x;
    float pp = p[i], mm = m[i], vv = v[i];
    adam_update(pp, mm, vv, (float)g[i] * grad_scale, b1, b2, eps, step_size, inv_bc2_sqrt);
    m[i] = mm;
    v[i] = vv;
    p[i] = pp;
  }
}

// one thread: advance the device-side step count and derive the two step-dependent Adam scalars in fp64,
// exactly as the host launcher does
__global__ void adam_coef_kernel(int* __restrict__ step, float* __restrict__ coef, float lr, float b1, float b2) {
  const int t = *step + 1;
  *step = t;
  const double bc1 = 1.0 - pow((double)b1, (double)t);
  const double bc2 = 1.0 - pow((double)b2, (double)t);
  coef[0] = (float)((double)lr / bc1);
  coef[1] = (float)(1.0 / sqrt(bc2));
}

inline int grid_for(long work_items, int per_block = 256, int cap = 256 * 8) {
  long g = cdiv(work_items, per_block);
  if (g < 1) g = 1;
  if (g > cap) g = cap;
  return (int)g;
}

// grid whose stride (grid*256 quads) is a multiple of the quads per row: every thread keeps one channel quad
inline int col_invariant_grid(long nquads, int qpr) {
  int a = qpr, b = 256;
  while (b) { const int t = a % b; a = b; b = t; }
  const int unit = qpr / a;                           // grid must be a multiple of qpr / gcd(qpr, 256)
  long g = cdiv(nquads, 256);
  if (g > 2048) g = 2048;
  g = cdiv(g, unit) * unit;
  return (int)g;
}

}  // namespace

// ---------------------------------------------------------------- C ABI
extern "C" int pcaa_bn_finalize(const double* stats, int nrep, long count, const float* lin_bias,
                                const float* gamma, const float* beta, float* running_mean,
                                float* running_var, long long* num_batches_tracked, float momentum,
                                float eps, float* scale, float* shift, float* mean, float* rstd,
                                int ch, void* stream) {
  PCAA_CHECK_ARG(stats && gamma && beta && scale && shift && mean && rstd, "pcaa_bn_finalize: null pointer");
  PCAA_CHECK_ARG(nrep >= 1 && count >= 1 && ch >= 1, "pcaa_bn_finalize: bad sizes");
  PCAA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "pcaa_bn_finalize: running stats must come together");
  const double unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  hipLaunchKernelGGL(bn_finalize_kernel, dim3((unsigned)cdiv(ch, 256)), dim3(256), 0, as_stream(stream),
                     stats, nrep, 1.0 / (double)count, unbias, lin_bias, gamma, beta, running_mean,
                     running_var, num_batches_tracked, momentum, eps, scale, shift, mean, rstd, ch);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_finalize");
}

extern "C" int pcaa_bn_eval_coeffs(const float* gamma, const float* beta, const float* running_mean,
                                   const float* running_var, const float* lin_bias, float eps, float* scale,
                                   float* shift, int ch, void* stream) {
  PCAA_CHECK_ARG(gamma && beta && running_mean && running_var && scale && shift && ch >= 1, "pcaa_bn_eval_coeffs: bad args");
  hipLaunchKernelGGL(bn_eval_coeffs_kernel, dim3((unsigned)cdiv(ch, 256)), dim3(256), 0, as_stream(stream),
                     gamma, beta, running_mean, running_var, lin_bias, eps, scale, shift, ch);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_eval_coeffs");
}

extern "C" int pcaa_bn_act_fwd(const void* y, void* a, int dtype, const float* scale, const float* shift,
                               long rows, int ch, void* stream) {
  PCAA_CHECK_ARG(y && a && scale && shift, "pcaa_bn_act_fwd: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && ch >= 4 && (ch & 3) == 0, "pcaa_bn_act_fwd: ch must be a multiple of 4");
  const long nq = rows * (ch >> 2);
  PCAA_CHECK_ARG(nq < (1L << 31), "pcaa_bn_act_fwd: tensor too large for 32-bit quad indices");
  const int grid = col_invariant_grid(nq, ch >> 2);
  if (dtype == PCAA_F32)
    hipLaunchKernelGGL(bn_act_fwd_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream),
                       (const float*)y, (float*)a, scale, shift, (unsigned)nq, (unsigned)(ch >> 2));
  else if (dtype == PCAA_BF16)
    hipLaunchKernelGGL(bn_act_fwd_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream),
                       (const bf16_t*)y, (bf16_t*)a, scale, shift, (unsigned)nq, (unsigned)(ch >> 2));
  else { pcaa_set_error("pcaa_bn_act_fwd: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_act_fwd");
}

extern "C" int pcaa_bn_act_fwd_split(const float* y, void* a_img, const float* scale, const float* shift, long rows,
                                     int ch, float img_scale, void* stream) {
  PCAA_CHECK_ARG(y && a_img && scale && shift, "pcaa_bn_act_fwd_split: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && ch >= 4 && (ch & 3) == 0, "pcaa_bn_act_fwd_split: ch must be a multiple of 4");
  const long nq = rows * (ch >> 2);
  PCAA_CHECK_ARG(nq < (1L << 31), "pcaa_bn_act_fwd_split: tensor too large for 32-bit quad indices");
  const int grid = col_invariant_grid(nq, ch >> 2);
  hipLaunchKernelGGL(bn_act_fwd_split_kernel, dim3(grid), dim3(256), 0, as_stream(stream), y, (split_t*)a_img, scale, shift,
                     (unsigned)nq, (unsigned)(ch >> 2), img_scale, pcaa_range_flag_ptr());
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_act_fwd_split");
}

extern "C" int pcaa_split_f16(const float* src, void* dst_img, long rows, int ch, int transpose, float img_scale,
                               void* stream) {
  PCAA_CHECK_ARG(src && dst_img && rows >= 1 && ch >= 1, "pcaa_split_f16: bad args");
  long g = cdiv(rows * ch, 256);
  if (g > 4096) g = 4096;
  hipLaunchKernelGGL(split_f16_kernel, dim3((unsigned)g), dim3(256), 0, as_stream(stream), src, (split_t*)dst_img, rows, ch,
                     transpose, img_scale, pcaa_range_flag_ptr());
  PCAA_RETURN_LAUNCH_STATUS("pcaa_split_f16");
}

extern "C" int pcaa_bn_act_meanpool_fwd(const void* y, int dtype, const float* scale, const float* shift,
                                        const float* mean, const float* rstd, float* pooled, float* e1, float* e2,
                                        long groups, int group_rows, int ch, void* stream) {
  PCAA_CHECK_ARG(y && scale && shift && pooled, "pcaa_bn_act_meanpool_fwd: null pointer");
  PCAA_CHECK_ARG(groups >= 1 && group_rows >= 1 && ch_ok(ch), "pcaa_bn_act_meanpool_fwd: ch/4 must divide 256 (ch=%d)", ch);
  const bool train = e1 != nullptr;
  PCAA_CHECK_ARG((e2 != nullptr) == train && (!train || (mean && rstd)),
                 "pcaa_bn_act_meanpool_fwd: e1, e2, mean, rstd come together");
  hipStream_t s = as_stream(stream);
#define LAUNCH_MP(T, TRAIN)                                                                                   \
  hipLaunchKernelGGL((bn_act_meanpool_kernel<T, TRAIN>), dim3((unsigned)groups), dim3(256), 0, s, (const T*)y, \
                     scale, shift, mean, rstd, pooled, e1, e2, group_rows, ch)
  if (dtype == PCAA_BF16 && ch == 1024 && group_rows % 4 == 0 && groups >= 512) {
    // the PointNet block's last layer at the shapes that matter: the streaming form
    const unsigned grid = (unsigned)std::min<long>(groups, 1024);
    if (train)
      hipLaunchKernelGGL(bn_act_meanpool_stream_kernel<true>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift,
                         mean, rstd, pooled, e1, e2, groups, group_rows, ch);
    else
      hipLaunchKernelGGL(bn_act_meanpool_stream_kernel<false>, dim3(grid), dim3(256), 0, s, (const bf16_t*)y, scale, shift,
                         mean, rstd, pooled, e1, e2, groups, group_rows, ch);
  }
  else if (dtype == PCAA_F32) { if (train) LAUNCH_MP(float, true); else LAUNCH_MP(float, false); }
  else if (dtype == PCAA_BF16) { if (train) LAUNCH_MP(bf16_t, true); else LAUNCH_MP(bf16_t, false); }
  else { pcaa_set_error("pcaa_bn_act_meanpool_fwd: bad dtype"); return PCAA_ERR_INVALID_ARG; }
#undef LAUNCH_MP
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_act_meanpool_fwd");
}

extern "C" int pcaa_bn_pool_bwd_stats(const float* dpool, const float* e1, const float* e2, float pool_scale,
                                      double* stats, int nrep, long groups, int ch, void* stream) {
  PCAA_CHECK_ARG(dpool && e1 && e2 && stats, "pcaa_bn_pool_bwd_stats: null pointer");
  PCAA_CHECK_ARG(groups >= 1 && ch >= 4 && (ch & 3) == 0 && nrep >= 1, "pcaa_bn_pool_bwd_stats: ch must be a multiple of 4");
  const int gpb = (int)std::max<long>(1, std::min<long>(32, groups / 256));      // >= 256 workgroups where there are groups for them
  hipLaunchKernelGGL(bn_pool_bwd_stats_kernel, dim3((unsigned)cdiv(groups, gpb)), dim3(256), 0, as_stream(stream),
                     dpool, e1, e2, pool_scale, stats, nrep, groups, ch, gpb, pcaa_take_bn_tail(stats));
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_pool_bwd_stats");
}

extern "C" int pcaa_bn_act_bwd_dz(const void* da, const float* dpool, int group_rows, float pool_scale,
                                  const void* y, void* dz, int dtype, const float* scale,
                                  const float* shift, const float* mean, const float* rstd,
                                  double* stats, int nrep, long rows, int ch, void* stream) {
  PCAA_CHECK_ARG((da != nullptr) != (dpool != nullptr), "pcaa_bn_act_bwd_dz: exactly one of da / dpool");
  PCAA_CHECK_ARG(y && scale && shift && mean && rstd && stats, "pcaa_bn_act_bwd_dz: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && rows < (1L << 31) && nrep >= 1 && ch_ok(ch),
                 "pcaa_bn_act_bwd_dz: ch/4 must divide 256 (ch=%d), rows < 2^31", ch);
  PCAA_CHECK_ARG(!dpool || group_rows >= 1, "pcaa_bn_act_bwd_dz: bad group_rows");
  const unsigned grid = (unsigned)cdiv(rows, ROWS_PER_BLOCK);
  hipStream_t s = as_stream(stream);
#define LAUNCH_DZ(T, POOL)                                                                       \
  hipLaunchKernelGGL((bn_act_bwd_dz_kernel<T, POOL>), dim3(grid), dim3(256), 0, s, (const T*)da, \
                     dpool, group_rows, pool_scale, (const T*)y, (T*)dz, scale, shift, mean,     \
                     rstd, stats, nrep, rows, ch)
  if (dtype == PCAA_F32) { if (dpool) LAUNCH_DZ(float, true); else LAUNCH_DZ(float, false); }
  else if (dtype == PCAA_BF16) { if (dpool) LAUNCH_DZ(bf16_t, true); else LAUNCH_DZ(bf16_t, false); }
  else { pcaa_set_error("pcaa_bn_act_bwd_dz: bad dtype"); return PCAA_ERR_INVALID_ARG; }
#undef LAUNCH_DZ
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_act_bwd_dz");
}

namespace {
// dy = c0 * (da * ELU'(y*scale+shift)) + c1 * y + c2 in ONE pass: dz is never materialised
// (the statistics pass before it reads da and y but writes nothing)
template <typename T, bool POOL, bool SPLIT = false>
__global__ __launch_bounds__(256) void bn_bwd_dy_fused_kernel(const T* __restrict__ da,
                                                              const float* __restrict__ dpool, unsigned group_rows,
                                                              float pool_scale, const T* __restrict__ y,
                                                              T* __restrict__ dy, const float* __restrict__ scale,
                                                              const float* __restrict__ shift,
                                                              const float* __restrict__ coef, unsigned nquads,
                                                              unsigned qpr, unsigned ch, float img_scale = 1.f,
                                                              int* oflow = nullptr) {
  // column-invariant grid (see bn_act_fwd_kernel): coefficients in registers, rows advance by a
  // constant; the pooled variant's group index is one 32-bit division per quad
  const unsigned q0 = blockIdx.x * 256u + threadIdx.x, stride = gridDim.x * 256u;
  const unsigned c = (q0 % qpr) << 2;
  const unsigned rstep = stride / qpr;
  const unsigned r = q0 / qpr;
  f32x4 sc = load4(scale + c), sh = load4(shift + c);
  const f32x4 k0 = load4(coef + c), k1 = load4(coef + ch + c), k2 = load4(coef + 2 * ch + c);
  constexpr bool kFast = sizeof(T) == 2;      // bf16 storage: ELU'(z) = exp2(min(z log2e, 0)), log2e folded into the affine
  if (kFast) { sc *= 1.4426950408889634f; sh *= 1.4426950408889634f; }
  // the pooled gradient is constant over a group's rows: the thread's (group, row in group) advance by constants --
  // no per-quad division (the first version spent more vector-ALU time on r / group_rows than the pass spends on HBM:
  // 237 us for 1 GB at config[1], round 2's trace)
  unsigned grp = POOL ? r / group_rows : 0u, rem = POOL ? r - grp * group_rows : 0u;
  const unsigned gstep = POOL ? rstep / group_rows : 0u, rrem = POOL ? rstep - gstep * group_rows : 0u;
  unsigned cur_group = 0xffffffffu;
  unsigned rr = r;
  f32x4 gpool = {0.f, 0.f, 0.f, 0.f};
  for (unsigned q = q0; q < nquads; q += stride) {
    const f32x4 yv = load4(y + (size_t)q * 4);
    f32x4 g;
    if (POOL) {
      if (grp != cur_group) {
        cur_group = grp;
        gpool = load4(dpool + (size_t)grp * ch + c);
        gpool *= pool_scale;
      }
      g = gpool;
      rem += rrem;
      grp += gstep;
      if (rem >= group_rows) { rem -= group_rows; ++grp; }
    } else {
      g = load4(da + (size_t)q * 4);
    }
    f32x4 e;
    if (kFast) {
      const f32x4 z = __builtin_elementwise_min(yv * sc + sh, f32x4{0.f, 0.f, 0.f, 0.f});
      e = f32x4{__builtin_amdgcn_exp2f(z.x), __builtin_amdgcn_exp2f(z.y), __builtin_amdgcn_exp2f(z.z),
                __builtin_amdgcn_exp2f(z.w)};
    } else {
      e = f32x4{elu_grad_from_pre(yv.x * sc.x + sh.x), elu_grad_from_pre(yv.y * sc.y + sh.y),
                elu_grad_from_pre(yv.z * sc.z + sh.z), elu_grad_from_pre(yv.w * sc.w + sh.w)};
    }
    if constexpr (SPLIT) {
      // dy as the [hi | lo] bf16 image (its only readers are the split-fp16 GEMMs); dy points at the image
      store4_split(reinterpret_cast<split_t*>(dy), rr, ch, c, k0 * (g * e) + k1 * yv + k2, img_scale, oflow);
      rr += rstep;
    } else {
      store4(dy + (size_t)q * 4, k0 * (g * e) + k1 * yv + k2);
    }
  }
}
}  // namespace

extern "C" int pcaa_bn_bwd_dy_fused(const void* da, const float* dpool, int group_rows, float pool_scale,
                                    const void* y, void* dy, int dtype, const float* scale, const float* shift,
                                    const float* coef, long rows, int ch, void* stream) {
  PCAA_CHECK_ARG((da != nullptr) != (dpool != nullptr), "pcaa_bn_bwd_dy_fused: exactly one of da / dpool");
  PCAA_CHECK_ARG(y && dy && scale && shift && coef, "pcaa_bn_bwd_dy_fused: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && ch >= 4 && (ch & 3) == 0, "pcaa_bn_bwd_dy_fused: ch must be a multiple of 4");
  PCAA_CHECK_ARG(!dpool || group_rows >= 1, "pcaa_bn_bwd_dy_fused: bad group_rows");
  const long nq = rows * (ch >> 2);
  PCAA_CHECK_ARG(nq < (1L << 31), "pcaa_bn_bwd_dy_fused: tensor too large for 32-bit quad indices");
  const int grid = col_invariant_grid(nq, ch >> 2);
  hipStream_t s = as_stream(stream);
#define LAUNCH_DYF(T, POOL)                                                                             \
  hipLaunchKernelGGL((bn_bwd_dy_fused_kernel<T, POOL>), dim3(grid), dim3(256), 0, s, (const T*)da, dpool, \
                     (unsigned)group_rows, pool_scale, (const T*)y, (T*)dy, scale, shift, coef, (unsigned)nq, \
                     (unsigned)(ch >> 2), (unsigned)ch)
  if (dtype == PCAA_F32) { if (dpool) LAUNCH_DYF(float, true); else LAUNCH_DYF(float, false); }
  else if (dtype == PCAA_BF16) { if (dpool) LAUNCH_DYF(bf16_t, true); else LAUNCH_DYF(bf16_t, false); }
  else { pcaa_set_error("pcaa_bn_bwd_dy_fused: bad dtype"); return PCAA_ERR_INVALID_ARG; }
#undef LAUNCH_DYF
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_bwd_dy_fused");
}

/* fp32 in, dy written as its [hi | lo] bf16 image [rows, 2 ch] (split-fp16 parity mode) */
extern "C" int pcaa_bn_bwd_dy_fused_split(const float* da, const float* dpool, int group_rows, float pool_scale,
                                          const float* y, void* dy_img, const float* scale, const float* shift,
                                          const float* coef, long rows, int ch, float img_scale, void* stream) {
  PCAA_CHECK_ARG((da != nullptr) != (dpool != nullptr), "pcaa_bn_bwd_dy_fused_split: exactly one of da / dpool");
  PCAA_CHECK_ARG(y && dy_img && scale && shift && coef, "pcaa_bn_bwd_dy_fused_split: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && ch >= 4 && (ch & 3) == 0, "pcaa_bn_bwd_dy_fused_split: ch must be a multiple of 4");
  PCAA_CHECK_ARG(!dpool || group_rows >= 1, "pcaa_bn_bwd_dy_fused_split: bad group_rows");
  PCAA_CHECK_ARG((const void*)da != dy_img, "pcaa_bn_bwd_dy_fused_split: the image cannot alias da");
  const long nq = rows * (ch >> 2);
  PCAA_CHECK_ARG(nq < (1L << 31), "pcaa_bn_bwd_dy_fused_split: tensor too large for 32-bit quad indices");
  const int grid = col_invariant_grid(nq, ch >> 2);
  hipStream_t s = as_stream(stream);
  if (dpool)
    hipLaunchKernelGGL((bn_bwd_dy_fused_kernel<float, true, true>), dim3(grid), dim3(256), 0, s, da, dpool,
                       (unsigned)group_rows, pool_scale, y, (float*)dy_img, scale, shift, coef, (unsigned)nq,
                       (unsigned)(ch >> 2), (unsigned)ch, img_scale, pcaa_range_flag_ptr());
  else
    hipLaunchKernelGGL((bn_bwd_dy_fused_kernel<float, false, true>), dim3(grid), dim3(256), 0, s, da, dpool,
                       (unsigned)group_rows, pool_scale, y, (float*)dy_img, scale, shift, coef, (unsigned)nq,
                       (unsigned)(ch >> 2), (unsigned)ch, img_scale, pcaa_range_flag_ptr());
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_bwd_dy_fused_split");
}

extern "C" int pcaa_bn_bwd_finalize(const double* stats, int nrep, long count, const float* gamma,
                                    const float* mean, const float* rstd, float* coef, float* dgamma,
                                    float* dbeta, int ch, void* stream) {
  PCAA_CHECK_ARG(stats && gamma && mean && rstd && coef, "pcaa_bn_bwd_finalize: null pointer");
  PCAA_CHECK_ARG(nrep >= 1 && count >= 1 && ch >= 1, "pcaa_bn_bwd_finalize: bad sizes");
  hipLaunchKernelGGL(bn_bwd_finalize_kernel, dim3((unsigned)cdiv(ch, 256)), dim3(256), 0, as_stream(stream),
                     stats, nrep, 1.0 / (double)count, gamma, mean, rstd, coef, dgamma, dbeta, ch);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_bwd_finalize");
}

// ---------------------------------------------------------------- finalize carried by the producer (bn_tail.h)
namespace {
thread_local BnTail g_tail = {};
}
BnTail pcaa_take_bn_tail(const double* stats) {
  BnTail t = {};
  if (g_tail.kind != 0 && g_tail.stats == stats) {
    t = g_tail;
    g_tail.kind = 0;
  }
  return t;
}
extern "C" int pcaa_bn_tail_arm_fwd(const double* stats, int nrep, long count, const float* lin_bias,
                                    const float* gamma, const float* beta, float* running_mean, float* running_var,
                                    long long* num_batches_tracked, float momentum, float eps, float* scale,
                                    float* shift, float* mean, float* rstd, int ch, unsigned* counter) {
  PCAA_CHECK_ARG(stats && gamma && beta && scale && shift && mean && rstd && counter, "pcaa_bn_tail_arm_fwd: null pointer");
  PCAA_CHECK_ARG(nrep >= 1 && count >= 1 && ch >= 1, "pcaa_bn_tail_arm_fwd: bad sizes");
  PCAA_CHECK_ARG((running_mean == nullptr) == (running_var == nullptr), "pcaa_bn_tail_arm_fwd: running stats must come together");
  BnTail t = {};
  t.kind = 1; t.nrep = nrep; t.ch = ch; t.counter = counter; t.stats = stats;
  t.inv_count = 1.0 / (double)count;
  t.unbias = count > 1 ? (double)count / (double)(count - 1) : 1.0;
  t.lin_bias = lin_bias; t.gamma = gamma; t.beta = beta;
  t.running_mean = running_mean; t.running_var = running_var; t.nbt = num_batches_tracked;
  t.momentum = momentum; t.eps = eps;
  t.scale = scale; t.shift = shift; t.mean = mean; t.rstd = rstd;
  g_tail = t;
  return PCAA_OK;
}
extern "C" int pcaa_bn_tail_arm_bwd(const double* stats, int nrep, long count, const float* gamma, const float* mean,
                                    const float* rstd, float* coef, float* dgamma, float* dbeta, int ch,
                                    unsigned* counter) {
  PCAA_CHECK_ARG(stats && gamma && mean && rstd && coef && counter, "pcaa_bn_tail_arm_bwd: null pointer");
  PCAA_CHECK_ARG(nrep >= 1 && count >= 1 && ch >= 1, "pcaa_bn_tail_arm_bwd: bad sizes");
  BnTail t = {};
  t.kind = 2; t.nrep = nrep; t.ch = ch; t.counter = counter; t.stats = stats;
  t.inv_count = 1.0 / (double)count;
  t.gamma = gamma; t.mean = const_cast<float*>(mean); t.rstd = const_cast<float*>(rstd);
  t.coef = coef; t.dgamma = dgamma; t.dbeta = dbeta;
  g_tail = t;
  return PCAA_OK;
}
// a launcher that took the tail and then could NOT launch hands it back: the stand-alone finalize then runs
// (ops.BnTailFwd/Bwd.resolve sees it pending) instead of coefficient tensors nobody ever wrote
void pcaa_rearm_bn_tail(const BnTail& t) {
  if (t.kind != 0) g_tail = t;
}
extern "C" int pcaa_bn_tail_pending(void) { return g_tail.kind != 0 ? 1 : 0; }
extern "C" int pcaa_bn_tail_disarm(void) {
  g_tail.kind = 0;
  return PCAA_OK;
}

extern "C" int pcaa_bn_bwd_dy(const void* dz, const void* y, void* dy, int dtype, const float* coef,
                              long rows, int ch, void* stream) {
  PCAA_CHECK_ARG(dz && y && dy && coef, "pcaa_bn_bwd_dy: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && ch >= 4 && (ch & 3) == 0, "pcaa_bn_bwd_dy: ch must be a multiple of 4");
  const long nq = rows * (ch >> 2);
  PCAA_CHECK_ARG(nq < (1L << 31), "pcaa_bn_bwd_dy: tensor too large for 32-bit quad indices");
  const int grid = col_invariant_grid(nq, ch >> 2);
  if (dtype == PCAA_F32)
    hipLaunchKernelGGL(bn_bwd_dy_kernel<float>, dim3(grid), dim3(256), 0, as_stream(stream), (const float*)dz,
                       (const float*)y, (float*)dy, coef, (unsigned)nq, (unsigned)(ch >> 2), (unsigned)ch);
  else if (dtype == PCAA_BF16)
    hipLaunchKernelGGL(bn_bwd_dy_kernel<bf16_t>, dim3(grid), dim3(256), 0, as_stream(stream), (const bf16_t*)dz,
                       (const bf16_t*)y, (bf16_t*)dy, coef, (unsigned)nq, (unsigned)(ch >> 2), (unsigned)ch);
  else { pcaa_set_error("pcaa_bn_bwd_dy: bad dtype"); return PCAA_ERR_INVALID_ARG; }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_bwd_dy");
}

extern "C" int pcaa_bn_bwd_dy_split(const float* dz, const float* y, void* dy_img, const float* coef, long rows, int ch,
                                    float img_scale, void* stream) {
  PCAA_CHECK_ARG(dz && y && dy_img && coef, "pcaa_bn_bwd_dy_split: null pointer");
  PCAA_CHECK_ARG(rows >= 1 && ch >= 4 && (ch & 3) == 0, "pcaa_bn_bwd_dy_split: ch must be a multiple of 4");
  PCAA_CHECK_ARG((const void*)dz != dy_img && (const void*)y != dy_img, "pcaa_bn_bwd_dy_split: the image cannot alias an input");
  const long nq = rows * (ch >> 2);
  PCAA_CHECK_ARG(nq < (1L << 31), "pcaa_bn_bwd_dy_split: tensor too large for 32-bit quad indices");
  const int grid = col_invariant_grid(nq, ch >> 2);
  hipLaunchKernelGGL(bn_bwd_dy_split_kernel, dim3(grid), dim3(256), 0, as_stream(stream), dz, y, (split_t*)dy_img, coef,
                     (unsigned)nq, (unsigned)(ch >> 2), (unsigned)ch, img_scale, pcaa_range_flag_ptr());
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bn_bwd_dy_split");
}

extern "C" int pcaa_bias_act(float* y, const float* bias, int act, long rows, int cols, void* stream) {
  PCAA_CHECK_ARG(y && rows >= 1 && cols >= 1, "pcaa_bias_act: bad args");
  PCAA_CHECK_ARG(act == PCAA_ACT_NONE || act == PCAA_ACT_ELU, "pcaa_bias_act: bad activation");
  const long n = rows * cols;
  hipLaunchKernelGGL(bias_act_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), y, bias, act, n, cols);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_bias_act");
}

extern "C" int pcaa_elu_bwd_from_out(const float* da, const float* a, float* dz, long n, void* stream) {
  PCAA_CHECK_ARG(da && a && dz && n >= 1, "pcaa_elu_bwd_from_out: bad args");
  hipLaunchKernelGGL(elu_bwd_from_out_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), da, a, dz, n);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_elu_bwd_from_out");
}

extern "C" int pcaa_colsum(const float* x, float* out, long rows, int cols, void* stream) {
  PCAA_CHECK_ARG(x && out && rows >= 1 && cols >= 1, "pcaa_colsum: bad args");
  hipLaunchKernelGGL(colsum_kernel, dim3((unsigned)cdiv(cols, 64)), dim3(256), 0, as_stream(stream), x, out, rows, cols);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_colsum");
}

extern "C" int pcaa_sum(const float* x, long n, float scale, float* out, void* stream) {
  PCAA_CHECK_ARG(x && out && n >= 1, "pcaa_sum: bad args");
  hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, as_stream(stream), x, n, scale, out);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_sum");
}

extern "C" int pcaa_rowsum(const float* x, float* out, long rows, int cols, float scale, void* stream) {
  PCAA_CHECK_ARG(x && out && rows >= 1 && cols >= 1, "pcaa_rowsum: bad args");
  hipLaunchKernelGGL(rowsum_kernel, dim3((unsigned)rows), dim3(64), 0, as_stream(stream), x, out, cols, scale);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_rowsum");
}

extern "C" int pcaa_scale_by_device_scalar(const float* x, const float* s, float* out, long n, void* stream) {
  PCAA_CHECK_ARG(x && s && out && n >= 1, "pcaa_scale_by_device_scalar: bad args");
  hipLaunchKernelGGL(scale_dev_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, s, out, n);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_scale_by_device_scalar");
}

extern "C" int pcaa_scale_rows(const float* x, const float* s, float* out, long rows, int cols, void* stream) {
  PCAA_CHECK_ARG(x && s && out && rows >= 1 && cols >= 1, "pcaa_scale_rows: bad args");
  const long n = rows * cols;
  hipLaunchKernelGGL(scale_rows_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), x, s, out, n, cols);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_scale_rows");
}

extern "C" int pcaa_prior_sample(const float* z0, const float* means, const long long* gt, int B, int K, int D,
                                 float* z, float* onehot, void* stream) {
  PCAA_CHECK_ARG(z0 && means && gt && z && onehot && B >= 1 && K >= 1 && D >= 1, "pcaa_prior_sample: bad args");
  const int n = B * (D > K ? D : K);
  hipLaunchKernelGGL(prior_sample_kernel, dim3((unsigned)cdiv(n, 256)), dim3(256), 0, as_stream(stream), z0, means, gt, B, K, D, z, onehot);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_prior_sample");
}

extern "C" int pcaa_pack_points(const float* src, long sb, long sc, long st, long sn, float* dst,
                                int B, int C, int T, int N, void* stream) {
  PCAA_CHECK_ARG(src && dst && B >= 1 && C >= 1 && T >= 1 && N >= 1, "pcaa_pack_points: bad args");
  const long n = (long)B * C * T * N;
  hipLaunchKernelGGL(pack_points_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), src, sb, sc, st, sn, dst, B, C, T, N);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_pack_points");
}

namespace {
// dst[r] = src[idx[r]] for rows of row_vec 16-B vectors: the device-side batch assembly of the packed crop
// store (batcher.py).  Pure byte movement, coalesced 16-B copies; an index outside [0, n_src) zero-fills
// its row and raises the error flag instead of reading out of bounds.
__global__ __launch_bounds__(256) void gather_rows_kernel(const uint4* __restrict__ src, const long long* __restrict__ idx,
                                                          long n_src, uint4* __restrict__ dst, long n_rows,
                                                          long row_vec, int* __restrict__ err) {
  const long total = n_rows * row_vec;
  for (long v = (long)blockIdx.x * 256 + threadIdx.x; v < total; v += (long)gridDim.x * 256) {
    const long r = v / row_vec, off = v - r * row_vec;
    const long long i = idx[r];
    uint4 val = {0u, 0u, 0u, 0u};
    if (i >= 0 && i < n_src) val = src[i * row_vec + off];
    else if (err != nullptr && off == 0) atomicOr(err, 1);
    dst[v] = val;
  }
}
}  // namespace

extern "C" int pcaa_gather_rows(const void* src, long n_src_rows, long row_bytes, const long long* idx, void* dst,
                                long n_rows, int* err_flag, void* stream) {
  PCAA_CHECK_ARG(src && idx && dst && n_src_rows >= 1 && n_rows >= 1 && row_bytes >= 16 && row_bytes % 16 == 0,
                 "pcaa_gather_rows: bad args (rows are multiples of 16 bytes)");
  PCAA_CHECK_ARG(((uintptr_t)src % 16) == 0 && ((uintptr_t)dst % 16) == 0, "pcaa_gather_rows: 16-B alignment");
  const long row_vec = row_bytes / 16;
  hipLaunchKernelGGL(gather_rows_kernel, dim3(grid_for(n_rows * row_vec, 256, 256 * 16)), dim3(256), 0, as_stream(stream),
                     reinterpret_cast<const uint4*>(src), idx, n_src_rows, reinterpret_cast<uint4*>(dst), n_rows,
                     row_vec, err_flag);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_gather_rows");
}

extern "C" int pcaa_dtc_im2col(const float* a, float* col, int B, int T, int Cin, int dilation, void* stream) {
  PCAA_CHECK_ARG(a && col && B >= 1 && T >= 1 && Cin >= 1 && dilation >= 1, "pcaa_dtc_im2col: bad args");
  const long n = (long)B * T * Cin * 3;
  hipLaunchKernelGGL(dtc_im2col_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), a, col, B, T, Cin, dilation);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_dtc_im2col");
}

extern "C" int pcaa_dtc_col2im(const float* dcol, float* da, int B, int T, int Cin, int dilation, void* stream) {
  PCAA_CHECK_ARG(dcol && da && B >= 1 && T >= 1 && Cin >= 1 && dilation >= 1, "pcaa_dtc_col2im: bad args");
  const long n = (long)B * T * Cin;
  hipLaunchKernelGGL(dtc_col2im_kernel, dim3(grid_for(n)), dim3(256), 0, as_stream(stream), dcol, da, B, T, Cin, dilation);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_dtc_col2im");
}

namespace {
// out[i] (=|+=) sum_s slabs[s*stride + i]   -- second half of slab split-K (no atomics, deterministic)
// acc += sum over the slabs of one quad, 8 slabs in flight per trip (nsplit reaches 64: walked one load at
// a time the reduction was a 20 us latency chain)
__device__ __forceinline__ void slab_sum(const float* __restrict__ base, int nsplit, long stride, f32x4& acc) {
  int s = 0;
  for (; s + 8 <= nsplit; s += 8) {
    f32x4 v[8];
#pragma unroll
    for (int u = 0; u < 8; ++u) v[u] = load4(base + (s + u) * stride);
#pragma unroll
    for (int u = 0; u < 8; ++u) acc += v[u];
  }
  for (; s < nsplit; ++s) acc += load4(base + s * stride);
}

__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ slabs, int nsplit,
                                                            long stride, long nquads, float* __restrict__ out,
                                                            int accumulate) {
  for (long q = (long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long)gridDim.x * 256) {
    f32x4 acc = accumulate ? load4(out + q * 4) : f32x4{0.f, 0.f, 0.f, 0.f};
    slab_sum(slabs + q * 4, nsplit, stride, acc);
    store4(out + q * 4, acc);
  }
}
}  // namespace

namespace {
// out[r][c] = sum_s slabs[s][r][c], plus the BatchNorm column statistics (sum, sum of squares
// over the rows) of the result: lets a statistics-producing GEMM with a tiny tile grid (the
// temporal block: 15-60 tiles on 256 CUs) still be split over K.
__global__ __launch_bounds__(256) void splitk_reduce_stats_kernel(const float* __restrict__ slabs, int nsplit,
                                                                  long stride, float* __restrict__ out,
                                                                  double* __restrict__ stats, int nrep,
                                                                  long rows, int ch) {
  __shared__ f32x4 red[2][256];
  const int qpr = ch >> 2, rl = 256 / qpr;
  const int cq = threadIdx.x % qpr, rlane = threadIdx.x / qpr;
  const int c = cq * 4;
  constexpr int RPB = 32;
  const long r0 = (long)blockIdx.x * RPB, r1 = min(rows, r0 + RPB);
  f32x4 s1 = {0.f, 0.f, 0.f, 0.f}, s2 = {0.f, 0.f, 0.f, 0.f};
  for (long r = r0 + rlane; r < r1; r += rl) {
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    slab_sum(slabs + r * ch + c, nsplit, stride, acc);
    store4(out + r * ch + c, acc);
    s1 += acc;
    s2 += acc * acc;
  }
  red[0][threadIdx.x] = s1;
  red[1][threadIdx.x] = s2;
  __syncthreads();
  for (int o = threadIdx.x; o < 2 * ch; o += 256) {
    const int stat = o / ch, cc = o - stat * ch;
    double v = 0.0;
    for (int l = 0; l < rl; ++l) v += (double)red[stat][l * qpr + (cc >> 2)][cc & 3];
    unsafeAtomicAdd(&stats[((long)(blockIdx.x % nrep) * 2 + stat) * ch + cc], v);
  }
}
}  // namespace

extern "C" int pcaa_splitk_reduce_stats(const float* slabs, int nsplit, long slab_stride, float* out,
                                        double* stats, int nrep, long rows, int ch, void* stream) {
  PCAA_CHECK_ARG(slabs && out && stats && nsplit >= 1 && nrep >= 1 && rows >= 1, "pcaa_splitk_reduce_stats: bad args");
  PCAA_CHECK_ARG(ch_ok(ch) && (slab_stride % 4) == 0, "pcaa_splitk_reduce_stats: ch/4 must divide 256 (ch=%d)", ch);
  hipLaunchKernelGGL(splitk_reduce_stats_kernel, dim3((unsigned)cdiv(rows, 32)), dim3(256), 0, as_stream(stream),
                     slabs, nsplit, slab_stride, out, stats, nrep, rows, ch);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_splitk_reduce_stats");
}

extern "C" int pcaa_splitk_reduce(const float* slabs, int nsplit, long slab_stride, long n, float* out,
                                  int accumulate, void* stream) {
  PCAA_CHECK_ARG(slabs && out && nsplit >= 1 && n >= 4 && (n % 4) == 0 && (slab_stride % 4) == 0,
                 "pcaa_splitk_reduce: n and slab_stride must be multiples of 4");
  PCAA_CHECK_ARG(((uintptr_t)slabs % 16) == 0 && ((uintptr_t)out % 16) == 0, "pcaa_splitk_reduce: 16-B alignment");
  hipLaunchKernelGGL(splitk_reduce_kernel, dim3(grid_for(n >> 2)), dim3(256), 0, as_stream(stream), slabs, nsplit,
                     slab_stride, n >> 2, out, accumulate);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_splitk_reduce");
}

static int launch_adam(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n, float beta1,
                       float beta2, float eps, float step_size, float inv_bc2_sqrt, const float* coef,
                       float grad_scale, int max_blocks, void* stream, const char* what) {
  PCAA_CHECK_ARG(((uintptr_t)param % 16) == 0 && ((uintptr_t)grad % 16) == 0 && ((uintptr_t)exp_avg % 16) == 0 &&
                 ((uintptr_t)exp_avg_sq % 16) == 0, "pcaa_adam_step: buffers must be 16-B aligned");
  const int cap = max_blocks > 0 ? max_blocks : 256 * 16;
  if (cap <= 1024)
    hipLaunchKernelGGL(adam_kernel<4>, dim3(grid_for(n >> 4, 256, cap)), dim3(256), 0, as_stream(stream), param, grad,
                       exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size, inv_bc2_sqrt, grad_scale, coef);
  else
    hipLaunchKernelGGL(adam_kernel<1>, dim3(grid_for(n >> 2, 256, cap)), dim3(256), 0, as_stream(stream), param, grad,
                       exp_avg, exp_avg_sq, n, beta1, beta2, eps, step_size, inv_bc2_sqrt, grad_scale, coef);
  PCAA_RETURN_LAUNCH_STATUS(what);
}

extern "C" int pcaa_adam_step_dev_g16(float* param, const void* grad_bf16, float* exp_avg, float* exp_avg_sq, long n,
                                      float beta1, float beta2, float eps, const float* coef_dev, float grad_scale,
                                      int max_blocks, void* stream) {
  PCAA_CHECK_ARG(param && grad_bf16 && exp_avg && exp_avg_sq && coef_dev && n >= 1 && max_blocks >= 0,
                 "pcaa_adam_step_dev_g16: bad args");
  PCAA_CHECK_ARG(((uintptr_t)param % 16) == 0 && ((uintptr_t)grad_bf16 % 8) == 0 && ((uintptr_t)exp_avg % 16) == 0 &&
                 ((uintptr_t)exp_avg_sq % 16) == 0, "pcaa_adam_step_dev_g16: buffers must be 16-B (gradient: 8-B) aligned");
  const bf16_t* g = reinterpret_cast<const bf16_t*>(grad_bf16);
  const int cap = max_blocks > 0 ? max_blocks : 256 * 16;
  if (cap <= 1024)
    hipLaunchKernelGGL((adam_kernel<4, bf16_t>), dim3(grid_for(n >> 4, 256, cap)), dim3(256), 0, as_stream(stream), param, g,
                       exp_avg, exp_avg_sq, n, beta1, beta2, eps, 0.f, 0.f, grad_scale, coef_dev);
  else
    hipLaunchKernelGGL((adam_kernel<1, bf16_t>), dim3(grid_for(n >> 2, 256, cap)), dim3(256), 0, as_stream(stream), param, g,
                       exp_avg, exp_avg_sq, n, beta1, beta2, eps, 0.f, 0.f, grad_scale, coef_dev);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_adam_step_dev_g16");
}

extern "C" int pcaa_adam_step(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                              float lr, float beta1, float beta2, float eps, int step, float grad_scale,
                              int max_blocks, void* stream) {
  PCAA_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && n >= 1 && step >= 1 && max_blocks >= 0,
                 "pcaa_adam_step: bad args");
  const double bc1 = 1.0 - pow((double)beta1, (double)step);
  const double bc2 = 1.0 - pow((double)beta2, (double)step);
  return launch_adam(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, (float)((double)lr / bc1),
                     (float)(1.0 / sqrt(bc2)), nullptr, grad_scale, max_blocks, stream, "pcaa_adam_step");
}

extern "C" int pcaa_adam_advance(int* step_dev, float* coef_dev, float lr, float beta1, float beta2, void* stream) {
  PCAA_CHECK_ARG(step_dev && coef_dev, "pcaa_adam_advance: bad args");
  hipLaunchKernelGGL(adam_coef_kernel, dim3(1), dim3(1), 0, as_stream(stream), step_dev, coef_dev, lr, beta1, beta2);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_adam_advance");
}

extern "C" int pcaa_adam_step_dev(float* param, const float* grad, float* exp_avg, float* exp_avg_sq, long n,
                                  float beta1, float beta2, float eps, const float* coef_dev, float grad_scale,
                                  int max_blocks, void* stream) {
  PCAA_CHECK_ARG(param && grad && exp_avg && exp_avg_sq && coef_dev && n >= 1 && max_blocks >= 0,
                 "pcaa_adam_step_dev: bad args");
  return launch_adam(param, grad, exp_avg, exp_avg_sq, n, beta1, beta2, eps, 0.f, 0.f, coef_dev, grad_scale,
                     max_blocks, stream, "pcaa_adam_step_dev");
}
