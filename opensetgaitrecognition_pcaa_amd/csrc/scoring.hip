// Open-set scoring of the PCAA inference path (reference inference_PCAA.py:129-136,
// 251-271): mixture likelihood of an embedding under K unit-covariance Gaussians in
// float64, and the k-window majority vote.
#include "common.h"

namespace {

// lik[b] = (1/K) sum_k exp(-0.5 * (D*log(2*pi) + |x_b - mu_k|^2))   -- the way scipy's
// multivariate_normal(mean, eye).pdf evaluates it (exp of the log-pdf), in fp64.
__global__ void joint_likelihood_kernel(const float* __restrict__ x, const float* __restrict__ means,
                                        int B, int K, int D, double* __restrict__ lik) {
  const int b = blockIdx.x * blockDim.x + threadIdx.x;
  if (b >= B) return;
  const double log2pi = 1.8378770664093453;
  double acc = 0.0;
  for (int k = 0; k < K; ++k) {
    double maha = 0.0;
    for (int d = 0; d < D; ++d) {
      const double diff = (double)x[(long)b * D + d] - (double)means[(long)k * D + d];
      maha += diff * diff;
    }
    acc += exp(-0.5 * ((double)D * log2pi + maha));
  }
  lik[b] = acc / (double)K;
}

// window w = crops [w*k, (w+1)*k): known iff #(lik > thr) > k/2, then the most frequent
// predicted label over ALL n_classes encoder outputs (lowest label on ties, like argmax(bincount)), else
// n_labels (= unknown; the number of labels present in the known test split)
__global__ void kvote_kernel(const double* __restrict__ lik, const long long* __restrict__ preds, double thr,
                             int k, int n_labels, int n_classes, int nwin, long long* __restrict__ out) {
  const int w = blockIdx.x * blockDim.x + threadIdx.x;
  if (w >= nwin) return;
  int above = 0;
  for (int i = 0; i < k; ++i) above += lik[(long)w * k + i] > thr ? 1 : 0;
  if (2 * above <= k) {
    out[w] = n_labels;
    return;
  }
  int best = 0, best_count = -1;
  for (int c = 0; c < n_classes; ++c) {
    int cnt = 0;
    for (int i = 0; i < k; ++i) cnt += preds[(long)w * k + i] == c ? 1 : 0;
    if (cnt > best_count) { best_count = cnt; best = c; }
  }
  out[w] = best;
}

}  // namespace

extern "C" int pcaa_joint_likelihood(const float* x, const float* means, int B, int K, int D, double* lik,
                                     void* stream) {
  PCAA_CHECK_ARG(x && means && lik && B >= 1 && K >= 1 && D >= 1, "pcaa_joint_likelihood: bad args");
  hipLaunchKernelGGL(joint_likelihood_kernel, dim3((unsigned)cdiv(B, 128)), dim3(128), 0, as_stream(stream), x,
                     means, B, K, D, lik);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_joint_likelihood");
}

extern "C" int pcaa_kvote(const double* lik, const long long* preds, double threshold, int k, int n_labels,
                          int n_classes, int n_windows, long long* out, void* stream) {
  PCAA_CHECK_ARG(lik && preds && out && k >= 1 && n_labels >= 1 && n_classes >= 1 && n_windows >= 1,
                 "pcaa_kvote: bad args");
  hipLaunchKernelGGL(kvote_kernel, dim3((unsigned)cdiv(n_windows, 128)), dim3(128), 0, as_stream(stream), lik, preds,
                     threshold, k, n_labels, n_classes, n_windows, out);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_kvote");
}
