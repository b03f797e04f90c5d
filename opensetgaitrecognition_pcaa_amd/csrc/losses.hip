// SeqChamferLoss (reference utils.py:88-132) fused forward + analytic backward,
// and CrossEntropyLoss + softmax-argmax (PCAA_ablation.py:891-893, 1008).
//
// Chamfer: one workgroup per (batch, time) frame.  Both point sets of the frame
// (2 * N * C floats) sit in LDS; work item w < N handles prediction j = w
// (min over ground-truth points, loss_1), work item N + i handles ground-truth
// point i (min over predictions, loss_2, gradient scattered with LDS atomics).
// The N x N distance matrix is never materialised.  Distances use the
// reference's expanded form |g|^2 + |p|^2 - 2 g.p so values and argmins follow
// the same rounding behaviour.
#include "common.h"

namespace {

constexpr int MAXC = 8;
constexpr int CH_THREADS = 256;

__global__ __launch_bounds__(CH_THREADS) void chamfer_kernel(
    const float* __restrict__ preds, long p_sb, long p_sc, long p_st, long p_sn,
    const float* __restrict__ gts, long g_sb, long g_sc, long g_st, long g_sn,
    int T, int N, int C, float* __restrict__ frame_loss,
    float* __restrict__ dpreds, long d_sb, long d_sc, long d_st, long d_sn,
    float grad_scale, const float* __restrict__ grad_per_b) {
  extern __shared__ __attribute__((aligned(16))) float sm[];
  float* sg = sm;                    // [N][MAXC] ground truth
  float* sp = sg + (long)N * MAXC;   // [N][MAXC] predictions
  float* rg = sp + (long)N * MAXC;   // [N] |g|^2
  float* rp = rg + N;                // [N] |p|^2
  float* sgrad = rp + N;             // [N][MAXC] gradient w.r.t. predictions
  __shared__ float red[CH_THREADS / 64];

  const int frame = blockIdx.x;
  const int b = frame / T, t = frame - b * T;
  const float* pb = preds + b * p_sb + t * p_st;
  const float* gb = gts + b * g_sb + t * g_st;

  for (int e = threadIdx.x; e < N * MAXC; e += CH_THREADS) {
    const int n = e / MAXC, c = e - n * MAXC;
    sg[e] = (c < C) ? gb[c * g_sc + n * g_sn] : 0.f;
    sp[e] = (c < C) ? pb[c * p_sc + n * p_sn] : 0.f;
    sgrad[e] = 0.f;
  }
  __syncthreads();
  for (int n = threadIdx.x; n < N; n += CH_THREADS) {
    float a = 0.f, q = 0.f;
#pragma unroll
    for (int c = 0; c < MAXC; ++c) {
      a = fmaf(sg[n * MAXC + c], sg[n * MAXC + c], a);
      q = fmaf(sp[n * MAXC + c], sp[n * MAXC + c], q);
    }
    rg[n] = a;
    rp[n] = q;
  }
  __syncthreads();

  const float w = grad_scale * (grad_per_b ? grad_per_b[b] : 1.f);
  float loss = 0.f;
  for (int item = threadIdx.x; item < 2 * N; item += CH_THREADS) {
    const bool dir1 = item < N;     // dir1: fixed prediction j, scan ground truth
    const int me = dir1 ? item : item - N;
    const float* mine = dir1 ? sp : sg;
    const float* other = dir1 ? sg : sp;
    const float* rother = dir1 ? rg : rp;
    float v[MAXC];
#pragma unroll
    for (int c = 0; c < MAXC; ++c) v[c] = mine[me * MAXC + c];
    const float rme = dir1 ? rp[me] : rg[me];
    float best = 3.4e38f;
    int bi = 0;
    for (int o = 0; o < N; ++o) {
      float dot = 0.f;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) dot = fmaf(other[o * MAXC + c], v[c], dot);
      // reference: rx (gt) + ry (pred) - 2 zz
      const float P = (dir1 ? (rother[o] + rme) : (rme + rother[o])) - 2.f * dot;
      if (P < best) { best = P; bi = o; }
    }
    loss += best;
    if (dpreds) {
      // d/dpred_j (|g_i|^2 + |p_j|^2 - 2 g_i.p_j) = 2 (p_j - g_i)
      const int j = dir1 ? me : bi;
      const int i = dir1 ? bi : me;
#pragma unroll
      for (int c = 0; c < MAXC; ++c) {
        const float gval = 2.f * w * (sp[j * MAXC + c] - sg[i * MAXC + c]);
        atomicAdd(&sgrad[j * MAXC + c], gval);
      }
    }
  }
  loss = wave_sum(loss);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = loss;
  __syncthreads();
  if (threadIdx.x == 0) {
    float tot = 0.f;
    for (int i = 0; i < CH_THREADS / 64; ++i) tot += red[i];
    frame_loss[frame] = tot;
  }
  if (dpreds) {
    float* db = dpreds + b * d_sb + t * d_st;
    // walk (c, n) with n fastest: contiguous in the decoder's [B,C,T,N] layout
    for (int e = threadIdx.x; e < N * C; e += CH_THREADS) {
      const int c = e / N, n = e - c * N;
      db[c * d_sc + n * d_sn] = sgrad[n * MAXC + c];
    }
  }
}

// one thread per row; single workgroup
__global__ __launch_bounds__(256) void cross_entropy_kernel(const float* __restrict__ logits,
                                                            const long long* __restrict__ target, int B,
                                                            int K, float* loss, float* dlogits,
                                                            float grad_scale, long long* preds, int* err_flag) {
  __shared__ double red[4];
  double acc = 0.0;
  for (int r = threadIdx.x; r < B; r += 256) {
    const float* x = logits + (long)r * K;
    float mx = x[0];
    for (int k = 1; k < K; ++k) mx = fmaxf(mx, x[k]);
    float se = 0.f;
    for (int k = 0; k < K; ++k) se += expf(x[k] - mx);
    const float inv = 1.f / se;
    int am = 0;
    float pbest = -1.f;
    long long tg = target ? target[r] : 0;
    if (tg < 0 || tg >= K) {       // torch raises here; flag it and stay inside the row
      if (err_flag) *err_flag = 1;
      tg = 0;
    }
    for (int k = 0; k < K; ++k) {
      const float pk = expf(x[k] - mx) / se;   // softmax exactly as exp / sum
      if (pk > pbest) { pbest = pk; am = k; }  // first index on ties
      if (dlogits) dlogits[(long)r * K + k] = grad_scale * (expf(x[k] - mx) * inv - (k == tg ? 1.f : 0.f)) / (float)B;
    }
    if (preds) preds[r] = am;
    if (target) acc += (double)(logf(se) + mx - x[tg]);
  }
  acc = wave_sum_d(acc);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
  __syncthreads();
  if (threadIdx.x == 0 && loss) loss[0] = (float)((red[0] + red[1] + red[2] + red[3]) / B);
}

}  // namespace

extern "C" int pcaa_chamfer_fwd_bwd(const float* preds, long p_sb, long p_sc, long p_st, long p_sn,
                                    const float* gts, long g_sb, long g_sc, long g_st, long g_sn,
                                    int B, int T, int N, int C, float* frame_loss, float* dpreds,
                                    long d_sb, long d_sc, long d_st, long d_sn, float grad_scale,
                                    const float* grad_per_b, void* stream) {
  PCAA_CHECK_ARG(preds && gts && frame_loss, "pcaa_chamfer_fwd_bwd: null pointer");
  PCAA_CHECK_ARG(B >= 1 && T >= 1 && N >= 1 && C >= 1 && C <= MAXC, "pcaa_chamfer_fwd_bwd: need 1<=C<=%d (C=%d)", MAXC, C);
  const size_t lds = ((size_t)3 * N * MAXC + 2 * (size_t)N) * sizeof(float);
  PCAA_CHECK_ARG(lds <= 150 * 1024, "pcaa_chamfer_fwd_bwd: N=%d too large for one LDS tile", N);
  if (lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(chamfer_kernel),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) { pcaa_set_error("pcaa_chamfer_fwd_bwd: cannot raise LDS limit: %s", hipGetErrorString(e)); return PCAA_ERR_LAUNCH; }
  }
  hipLaunchKernelGGL(chamfer_kernel, dim3((unsigned)(B * T)), dim3(CH_THREADS), lds, as_stream(stream),
                     preds, p_sb, p_sc, p_st, p_sn, gts, g_sb, g_sc, g_st, g_sn, T, N, C, frame_loss,
                     dpreds, d_sb, d_sc, d_st, d_sn, grad_scale, grad_per_b);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_chamfer_fwd_bwd");
}

extern "C" int pcaa_cross_entropy(const float* logits, const long long* target, int B, int K, float* loss,
                                  float* dlogits, float grad_scale, long long* preds, int* err_flag,
                                  void* stream) {
  PCAA_CHECK_ARG(logits && B >= 1 && K >= 1, "pcaa_cross_entropy: bad args");
  PCAA_CHECK_ARG(target || (!loss && !dlogits), "pcaa_cross_entropy: loss/grad need targets");
  hipLaunchKernelGGL(cross_entropy_kernel, dim3(1), dim3(256), 0, as_stream(stream), logits, target, B, K,
                     loss, dlogits, grad_scale, preds, err_flag);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_cross_entropy");
}
