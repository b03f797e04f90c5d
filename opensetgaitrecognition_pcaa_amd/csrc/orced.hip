// OR-CED baseline heads (reference models.py:446-505, train_ORCED.py:143-176, utils.py:72-85) on the device:
//
//   vae_mu     = MLP_mu(x4)      = x4 . Wmu^T + bmu            [B, L]     (L = SUP_LATENT_DIM = 32, x4 [B, 512])
//   vae_logvar = MLP_logvar(x4)  = x4 . Wlv^T + blv            [B, L]
//   sup_fv     = vae_mu + eps * exp(0.5 * vae_logvar)           (eps: the caller's randn draw, [B, L])
//   logits     = MLP_classification(sup_fv) = sup_fv . Wc^T + bc   [B, K]   (no activation anywhere)
//
// and  KL( N(mu, exp(logvar)) || N(mu_k, I) )  averaged over the batch.  Everything here is [B, 32]-sized: one
// workgroup per batch row, fp32 FMAs, the row of x4 staged in LDS.  Round 2 ran these as stock torch ops.
#include "common.h"

namespace {

constexpr int MAX_IN = 1024, MAX_LAT = 128, MAX_K = 64;

// one wave: dot(w[0..n), x[0..n)) with x in LDS
__device__ __forceinline__ float wave_dot(const float* __restrict__ w, const float* xs, int n, int lane) {
  float a = 0.f;
  for (int k = lane; k < n; k += 64) a = fmaf(w[k], xs[k], a);
  return wave_sum(a);
}

__global__ __launch_bounds__(256) void orced_heads_fwd_kernel(const float* __restrict__ x4, const float* __restrict__ Wmu,
                                                              const float* __restrict__ bmu, const float* __restrict__ Wlv,
                                                              const float* __restrict__ blv, const float* __restrict__ eps,
                                                              const float* __restrict__ Wc, const float* __restrict__ bc,
                                                              float* mu, float* logvar, float* sup_fv, float* logits,
                                                              int K, int d_in, int L) {
  __shared__ float xs[MAX_IN];
  __shared__ float ml[2 * MAX_LAT];
  __shared__ float ss[MAX_LAT];
  const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int k = tid; k < d_in; k += 256) xs[k] = x4[(long)b * d_in + k];
  __syncthreads();
  for (int o = wave; o < 2 * L; o += 4) {
    const bool is_mu = o < L;
    const int j = is_mu ? o : o - L;
    const float v = wave_dot((is_mu ? Wmu : Wlv) + (long)j * d_in, xs, d_in, lane) + (is_mu ? bmu[j] : blv[j]);
    if (lane == 0) ml[o] = v;
  }
  __syncthreads();
  if (tid < L) {
    const float m = ml[tid], lv = ml[L + tid];
    const float s = m + eps[(long)b * L + tid] * expf(0.5f * lv);
    mu[(long)b * L + tid] = m;
    logvar[(long)b * L + tid] = lv;
    sup_fv[(long)b * L + tid] = s;
    ss[tid] = s;
  }
  __syncthreads();
  if (tid < K) {
    float a = bc[tid];
    for (int j = 0; j < L; ++j) a = fmaf(Wc[tid * L + j], ss[j], a);
    logits[(long)b * K + tid] = a;
  }
}

// per row: the gradients that reach mu and logvar (into ws[0][B][L], ws[1][B][L]) and dx4
__global__ __launch_bounds__(256) void orced_heads_bwd_rows_kernel(const float* __restrict__ eps,
                                                                   const float* __restrict__ logvar,
                                                                   const float* __restrict__ Wmu,
                                                                   const float* __restrict__ Wlv,
                                                                   const float* __restrict__ Wc,
                                                                   const float* __restrict__ d_logits,
                                                                   const float* __restrict__ d_sup,
                                                                   const float* __restrict__ d_mu,
                                                                   const float* __restrict__ d_logvar, float* ws,
                                                                   float* dx4, int B, int K, int d_in, int L) {
  __shared__ float dm[MAX_LAT], dl[MAX_LAT];
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid < L) {
    float ds = d_sup ? d_sup[(long)b * L + tid] : 0.f;
    if (d_logits)
      for (int k = 0; k < K; ++k) ds = fmaf(Wc[k * L + tid], d_logits[(long)b * K + k], ds);
    const float lv = logvar[(long)b * L + tid];
    const float gm = (d_mu ? d_mu[(long)b * L + tid] : 0.f) + ds;
    const float gl = (d_logvar ? d_logvar[(long)b * L + tid] : 0.f) + ds * eps[(long)b * L + tid] * 0.5f * expf(0.5f * lv);
    dm[tid] = gm;
    dl[tid] = gl;
    ws[(long)b * L + tid] = gm;
    ws[(long)(B + b) * L + tid] = gl;
  }
  __syncthreads();
  if (dx4 != nullptr) {
    for (int k = tid; k < d_in; k += 256) {
      float a = 0.f;
      for (int j = 0; j < L; ++j) a = fmaf(Wmu[(long)j * d_in + k], dm[j], fmaf(Wlv[(long)j * d_in + k], dl[j], a));
      dx4[(long)b * d_in + k] = a;
    }
  }
}

// weight / bias gradients: block o < 2L: row o of dWmu (o < L) or dWlv; block 2L: dWc and dbc
__global__ __launch_bounds__(256) void orced_heads_bwd_params_kernel(const float* __restrict__ x4,
                                                                     const float* __restrict__ sup_fv,
                                                                     const float* __restrict__ d_logits,
                                                                     const float* __restrict__ ws, float* dWmu, float* dbmu,
                                                                     float* dWlv, float* dblv, float* dWc, float* dbc, int B,
                                                                     int K, int d_in, int L) {
  const int o = blockIdx.x, tid = threadIdx.x;
  if (o < 2 * L) {
    const bool is_mu = o < L;
    const int j = is_mu ? o : o - L;
    const float* g = ws + (is_mu ? 0 : (long)B * L) + j;      // g[b * L]
    float* dW = (is_mu ? dWmu : dWlv) + (long)j * d_in;
    for (int k = tid; k < d_in; k += 256) {
      float a = 0.f;
      for (int b = 0; b < B; ++b) a = fmaf(g[(long)b * L], x4[(long)b * d_in + k], a);
      dW[k] = a;
    }
    if (tid == 0) {
      float a = 0.f;
      for (int b = 0; b < B; ++b) a += g[(long)b * L];
      (is_mu ? dbmu : dblv)[j] = a;
    }
  } else {
    for (int i = tid; i < K * L; i += 256) {
      const int k = i / L, j = i - k * L;
      float a = 0.f;
      if (d_logits)
        for (int b = 0; b < B; ++b) a = fmaf(d_logits[(long)b * K + k], sup_fv[(long)b * L + j], a);
      dWc[i] = a;
    }
    if (tid < K) {
      float a = 0.f;
      if (d_logits)
        for (int b = 0; b < B; ++b) a += d_logits[(long)b * K + tid];
      dbc[tid] = a;
    }
  }
}

// loss = mean_b( -0.5 sum_j (1 + lv - (mu - mk)^2 - exp(lv)) ); gradients scaled by gscale / B
__global__ __launch_bounds__(256) void orced_kl_kernel(const float* __restrict__ mu, const float* __restrict__ logvar,
                                                       const float* __restrict__ mu_k, float* loss, float* d_mu,
                                                       float* d_logvar, float* d_muk, float gscale, int B, int L) {
  __shared__ double red[4];
  const int tid = threadIdx.x, n = B * L;
  double acc = 0.0;
  const float gs = gscale / (float)B;
  for (int i = tid; i < n; i += 256) {
    const float m = mu[i], lv = logvar[i], d = m - mu_k[i], e = expf(lv);
    acc += (double)(1.f + lv - d * d - e);
    if (d_mu) d_mu[i] = gs * d;
    if (d_muk) d_muk[i] = -gs * d;
    if (d_logvar) d_logvar[i] = gs * (-0.5f) * (1.f - e);
  }
  acc = wave_sum_d(acc);
  if ((tid & 63) == 0) red[tid >> 6] = acc;
  __syncthreads();
  if (tid == 0 && loss) *loss = (float)(-0.5 * ((red[0] + red[1]) + (red[2] + red[3])) / (double)B);
}

}  // namespace

extern "C" int pcaa_orced_heads_supported(int B, int K, int d_in, int d_lat) {
  return (B >= 1 && K >= 1 && K <= MAX_K && d_in >= 1 && d_in <= MAX_IN && d_lat >= 1 && d_lat <= MAX_LAT) ? 1 : 0;
}

extern "C" int pcaa_orced_heads_fwd(const float* x4, const float* Wmu, const float* bmu, const float* Wlv,
                                    const float* blv, const float* eps, const float* Wc, const float* bc, float* mu,
                                    float* logvar, float* sup_fv, float* logits, int B, int K, int d_in, int d_lat,
                                    void* stream) {
  PCAA_CHECK_ARG(x4 && Wmu && bmu && Wlv && blv && eps && Wc && bc && mu && logvar && sup_fv && logits,
                 "pcaa_orced_heads_fwd: null pointer");
  PCAA_CHECK_ARG(pcaa_orced_heads_supported(B, K, d_in, d_lat), "pcaa_orced_heads_fwd: need K <= %d, d_in <= %d, d_lat <= %d",
                 MAX_K, MAX_IN, MAX_LAT);
  hipLaunchKernelGGL(orced_heads_fwd_kernel, dim3(B), dim3(256), 0, as_stream(stream), x4, Wmu, bmu, Wlv, blv, eps, Wc, bc,
                     mu, logvar, sup_fv, logits, K, d_in, d_lat);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_orced_heads_fwd");
}

extern "C" int pcaa_orced_heads_bwd(const float* x4, const float* eps, const float* logvar, const float* sup_fv,
                                    const float* Wmu, const float* Wlv, const float* Wc, const float* d_logits,
                                    const float* d_sup, const float* d_mu, const float* d_logvar, float* ws, float* dWmu,
                                    float* dbmu, float* dWlv, float* dblv, float* dWc, float* dbc, float* dx4, int B, int K,
                                    int d_in, int d_lat, void* stream) {
  PCAA_CHECK_ARG(x4 && eps && logvar && sup_fv && Wmu && Wlv && Wc && ws && dWmu && dbmu && dWlv && dblv && dWc && dbc,
                 "pcaa_orced_heads_bwd: null pointer");
  PCAA_CHECK_ARG(pcaa_orced_heads_supported(B, K, d_in, d_lat), "pcaa_orced_heads_bwd: need K <= %d, d_in <= %d, d_lat <= %d",
                 MAX_K, MAX_IN, MAX_LAT);
  hipLaunchKernelGGL(orced_heads_bwd_rows_kernel, dim3(B), dim3(256), 0, as_stream(stream), eps, logvar, Wmu, Wlv, Wc,
                     d_logits, d_sup, d_mu, d_logvar, ws, dx4, B, K, d_in, d_lat);
  hipLaunchKernelGGL(orced_heads_bwd_params_kernel, dim3(2 * d_lat + 1), dim3(256), 0, as_stream(stream), x4, sup_fv,
                     d_logits, ws, dWmu, dbmu, dWlv, dblv, dWc, dbc, B, K, d_in, d_lat);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_orced_heads_bwd");
}

extern "C" int pcaa_orced_kl(const float* mu, const float* logvar, const float* mu_k, float* loss, float* d_mu,
                             float* d_logvar, float* d_muk, float gscale, int B, int d_lat, void* stream) {
  PCAA_CHECK_ARG(mu && logvar && mu_k && B >= 1 && d_lat >= 1, "pcaa_orced_kl: bad args");
  hipLaunchKernelGGL(orced_kl_kernel, dim3(1), dim3(256), 0, as_stream(stream), mu, logvar, mu_k, loss, d_mu, d_logvar, d_muk,
                     gscale, B, d_lat);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_orced_kl");
}
