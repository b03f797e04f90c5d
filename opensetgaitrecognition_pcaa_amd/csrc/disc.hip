// CGDiscriminator (reference models.py:405-421) and the WGAN-GP critic step
// (PCAA_ablation.py:939-976) with the closed-form second-order gradient of
// SURVEY.md Appendix A.
//
// One wavefront per batch row: lane t owns hidden unit t of layer 1 (64 units)
// and, for t < 32, unit t of layer 2, input column t of W1 and w3[t].  Weights
// live in LDS with odd pitches so both row-wise and column-wise walks are
// bank-conflict free; the per-row vectors that the parameter gradients are
// outer products of are written to a workspace and summed over rows by a
// second, parameter-parallel kernel (no atomics, deterministic).
#include "common.h"

namespace {

constexpr int XD = 32;    // latent width (SUP_LATENT_DIM)
constexpr int H1 = 64, H2 = 32;
constexpr int MAXK = 32;  // label width limit
// per-(pass,row) record layout (floats)
constexpr int R_U = 0, R_D1 = 64, R_H1 = 128, R_D2 = 192, R_H2C = 224, R_C = 256, REC = 272;
// GP extras per row
constexpr int G_S1 = 0, G_GB = 64, G_S2 = 96, G_RB1 = 128, GPX = 192;

struct DiscParams {
  const float *W1, *b1, *W2, *b2, *W3, *b3;
};

struct DiscLds {
  float W1[H1 * 65];   // pitch P1 = IN|1 <= 65
  float W2[H2 * 65];
  float b1[H1], b2[H2], w3[H2];
  float u[64], h1[64], d1[64], h2[32], d2[32], va[64], vb[64];
  float b3;
};

__device__ __forceinline__ void load_weights(DiscLds& s, const DiscParams& p, int IN, int P1, int t) {
  for (int i = t; i < H1 * IN; i += 64) s.W1[(i / IN) * P1 + (i % IN)] = p.W1[i];
  for (int i = t; i < H2 * H1; i += 64) s.W2[(i >> 6) * 65 + (i & 63)] = p.W2[i];
  s.b1[t] = p.b1[t];
  if (t < H2) { s.b2[t] = p.b2[t]; s.w3[t] = p.W3[t]; }
  if (t == 0) s.b3 = p.b3[0];
}

// state a lane keeps after a forward pass
struct Fwd {
  float h1, e1, e1pp;   // unit t of layer 1: ELU, ELU', ELU''
  float h2, e2, e2pp;   // unit t of layer 2 (t < 32)
  float D;
};

__device__ __forceinline__ Fwd forward_pass(DiscLds& s, int IN, int P1, int t) {
  Fwd f;
  float a1 = s.b1[t];
  for (int i = 0; i < IN; ++i) a1 = fmaf(s.W1[t * P1 + i], s.u[i], a1);
  f.h1 = elu_f(a1);
  f.e1 = a1 > 0.f ? 1.f : expf(a1);
  f.e1pp = a1 > 0.f ? 0.f : f.e1;
  s.h1[t] = f.h1;
  __syncthreads();
  float a2 = 0.f;
  f.h2 = f.e2 = f.e2pp = 0.f;
  if (t < H2) {
    a2 = s.b2[t];
#pragma unroll 8
    for (int o = 0; o < H1; ++o) a2 = fmaf(s.W2[t * 65 + o], s.h1[o], a2);
    f.h2 = elu_f(a2);
    f.e2 = a2 > 0.f ? 1.f : expf(a2);
    f.e2pp = a2 > 0.f ? 0.f : f.e2;
  }
  const float part = (t < H2) ? s.w3[t] * f.h2 : 0.f;
  f.D = wave_sum(part) + s.b3;
  return f;
}

// first-order backward of c * D w.r.t. everything; fills the record, returns du_t
__device__ __forceinline__ float first_order(DiscLds& s, const Fwd& f, float c, int IN, int P1, int t,
                                             float* rec) {
  if (t < H2) {
    const float d2 = c * s.w3[t] * f.e2;
    s.d2[t] = d2;
    if (rec) { rec[R_D2 + t] = d2; rec[R_H2C + t] = c * f.h2; }
  }
  __syncthreads();
  float acc = 0.f;
#pragma unroll 8
  for (int p = 0; p < H2; ++p) acc = fmaf(s.W2[p * 65 + t], s.d2[p], acc);
  const float d1 = f.e1 * acc;
  s.d1[t] = d1;
  if (rec) {
    rec[R_D1 + t] = d1;
    rec[R_H1 + t] = f.h1;
    rec[R_U + t] = (t < IN) ? s.u[t] : 0.f;
    if (t == 0) rec[R_C] = c;
  }
  __syncthreads();
  float du = 0.f;
  if (t < IN) {
#pragma unroll 8
    for (int o = 0; o < H1; ++o) du = fmaf(s.W1[o * P1 + t], s.d1[o], du);
  }
  return du;
}

// MODE 0: forward, 1: first-order backward with gout, 2: WGAN-GP critic step,
// 3: backward OF the first-order input gradient (double backward): with g = d(sum_b c_b D_b)/dx the input
//    gradient MODE 1 returns and gbar the incoming gradient w.r.t. g, the gradients of <gbar, g> w.r.t. the
//    parameters, x, label and c (SURVEY.md Appendix A with c general) -- what autograd needs to differentiate
//    through torch.autograd.grad(D(interp), interp, create_graph=True) (PCAA_ablation.py:955-963)
template <int MODE>
__global__ __launch_bounds__(64) void disc_rows_kernel(const float* __restrict__ xa,   // x (MODE 0/1) or z (MODE 2)
                                                       const float* __restrict__ xb,   // fv (MODE 2)
                                                       const float* __restrict__ label,
                                                       const float* __restrict__ aux,  // gout (MODE 1) or alphas (MODE 2)
                                                       int B, int K, DiscParams prm, float gp_weight,
                                                       float* __restrict__ out,        // D (MODE 0) / rowvals [B][3] (MODE 2)
                                                       float* __restrict__ dx, float* __restrict__ dlabel,
                                                       float* __restrict__ recs, float* __restrict__ gpx) {
  __shared__ DiscLds s;
  const int t = threadIdx.x, row = blockIdx.x;
  const int IN = XD + K, P1 = IN | 1;
  load_weights(s, prm, IN, P1, t);
  const float lab = (t >= XD && t < IN) ? label[(long)row * K + (t - XD)] : 0.f;
  __syncthreads();

  if (MODE == 0 || MODE == 1) {
    s.u[t] = (t < XD) ? xa[(long)row * XD + t] : lab;
    __syncthreads();
    Fwd f = forward_pass(s, IN, P1, t);
    if (MODE == 0) {
      if (t == 0) out[row] = f.D;
      return;
    }
    const float c = aux[row];
    const float du = first_order(s, f, c, IN, P1, t, recs ? recs + (long)row * REC : nullptr);
    if (dx && t < XD) dx[(long)row * XD + t] = du;
    if (dlabel && t >= XD && t < IN) dlabel[(long)row * K + (t - XD)] = du;
    return;
  }

  if (MODE == 3) {
    s.u[t] = (t < XD) ? xa[(long)row * XD + t] : lab;
    __syncthreads();
    Fwd fi = forward_pass(s, IN, P1, t);
    const float c = aux[row];
    float* rec = recs + (long)row * REC;
    float* gx = gpx + (long)row * GPX;
    if (t < H2) {
      const float s2 = c * fi.e2 * s.w3[t];
      s.d2[t] = s2;
      gx[G_S2 + t] = s2;
    }
    __syncthreads();
    float r1 = 0.f;
#pragma unroll 8
    for (int p = 0; p < H2; ++p) r1 = fmaf(s.W2[p * 65 + t], s.d2[p], r1);
    gx[G_S1 + t] = fi.e1 * r1;
    const float gbar = (t < XD) ? xb[(long)row * XD + t] : 0.f;
    s.va[t] = gbar;
    if (t < XD) gx[G_GB + t] = gbar;
    __syncthreads();
    float sb1 = 0.f;
#pragma unroll 8
    for (int i = 0; i < XD; ++i) sb1 = fmaf(s.W1[t * P1 + i], s.va[i], sb1);
    const float rb1 = fi.e1 * sb1;
    float ab1 = fi.e1pp * r1 * sb1;
    s.vb[t] = rb1;
    gx[G_RB1 + t] = rb1;
    __syncthreads();
    float dc_part = 0.f;
    if (t < H2) {
      float sb2 = 0.f;
#pragma unroll 8
      for (int o = 0; o < H1; ++o) sb2 = fmaf(s.W2[t * 65 + o], s.vb[o], sb2);
      const float ab2 = fi.e2pp * c * s.w3[t] * sb2;
      s.d2[t] = ab2;
      rec[R_D2 + t] = ab2;
      rec[R_H2C + t] = c * fi.e2 * sb2;
      dc_part = fi.e2 * s.w3[t] * sb2;
    }
    const float dc = wave_sum(dc_part);
    __syncthreads();
    float hb1 = 0.f;
#pragma unroll 8
    for (int p = 0; p < H2; ++p) hb1 = fmaf(s.W2[p * 65 + t], s.d2[p], hb1);
    ab1 = fmaf(fi.e1, hb1, ab1);
    rec[R_D1 + t] = ab1;
    rec[R_H1 + t] = fi.h1;
    rec[R_U + t] = (t < IN) ? s.u[t] : 0.f;
    s.d1[t] = ab1;
    __syncthreads();
    if (t < IN && (dx || dlabel)) {
      float du = 0.f;
#pragma unroll 8
      for (int o = 0; o < H1; ++o) du = fmaf(s.W1[o * P1 + t], s.d1[o], du);
      if (dx && t < XD) dx[(long)row * XD + t] = du;
      if (dlabel && t >= XD) dlabel[(long)row * K + (t - XD)] = du;
    }
    if (t == 0) {
      rec[R_C] = 0.f;                       // b3 is not reached by the input gradient
      if (out) out[row] = dc;
    }
    return;
  }

  // ---------------- MODE 2
  const float invB = 1.f / (float)B;
  const float zv = (t < XD) ? xa[(long)row * XD + t] : 0.f;
  const float fvv = (t < XD) ? xb[(long)row * XD + t] : 0.f;
  // real pass: c = -1/B
  s.u[t] = (t < XD) ? zv : lab;
  __syncthreads();
  Fwd fr = forward_pass(s, IN, P1, t);
  const float du_real = first_order(s, fr, -invB, IN, P1, t, recs + ((long)0 * B + row) * REC);
  __syncthreads();
  // fake pass: c = +1/B
  s.u[t] = (t < XD) ? fvv : lab;
  __syncthreads();
  Fwd ff = forward_pass(s, IN, P1, t);
  (void)first_order(s, ff, invB, IN, P1, t, recs + ((long)1 * B + row) * REC);
  __syncthreads();
  // interpolate pass: gradient penalty
  const float alpha = aux[row];
  s.u[t] = (t < XD) ? (zv + alpha * (fvv - zv)) : lab;
  __syncthreads();
  Fwd fi = forward_pass(s, IN, P1, t);
  float* rec = recs + ((long)2 * B + row) * REC;
  float* gx = gpx + (long)row * GPX;
  if (t < H2) {
    const float s2 = fi.e2 * s.w3[t];
    s.d2[t] = s2;
    gx[G_S2 + t] = s2;
  }
  __syncthreads();
  float r1 = 0.f;
#pragma unroll 8
  for (int p = 0; p < H2; ++p) r1 = fmaf(s.W2[p * 65 + t], s.d2[p], r1);
  const float s1 = fi.e1 * r1;
  s.d1[t] = s1;
  gx[G_S1 + t] = s1;
  __syncthreads();
  float g = 0.f;
  if (t < XD) {
#pragma unroll 8
    for (int o = 0; o < H1; ++o) g = fmaf(s.W1[o * P1 + t], s.d1[o], g);
  }
  const float gg = wave_sum(g * g);
  const float nrm = sqrtf(gg + 1e-12f);
  const float gprow = (nrm - 1.f) * (nrm - 1.f);
  const float gbar = gp_weight * 2.f * invB * (nrm - 1.f) / nrm * g;   // zero for t >= 32
  s.va[t] = gbar;
  if (t < XD) gx[G_GB + t] = gbar;
  __syncthreads();
  float sb1 = 0.f;
#pragma unroll 8
  for (int i = 0; i < XD; ++i) sb1 = fmaf(s.W1[t * P1 + i], s.va[i], sb1);
  const float rb1 = fi.e1 * sb1;
  float ab1 = fi.e1pp * r1 * sb1;
  s.vb[t] = rb1;
  gx[G_RB1 + t] = rb1;
  __syncthreads();
  if (t < H2) {
    float sb2 = 0.f;
#pragma unroll 8
    for (int o = 0; o < H1; ++o) sb2 = fmaf(s.W2[t * 65 + o], s.vb[o], sb2);
    const float ab2 = fi.e2pp * s.w3[t] * sb2;
    s.d2[t] = ab2;
    rec[R_D2 + t] = ab2;
    rec[R_H2C + t] = fi.e2 * sb2;
  }
  __syncthreads();
  float hb1 = 0.f;
#pragma unroll 8
  for (int p = 0; p < H2; ++p) hb1 = fmaf(s.W2[p * 65 + t], s.d2[p], hb1);
  ab1 = fmaf(fi.e1, hb1, ab1);
  rec[R_D1 + t] = ab1;
  rec[R_H1 + t] = fi.h1;
  rec[R_U + t] = (t < IN) ? s.u[t] : 0.f;
  if (dx != nullptr) {
    // d(d_loss)/dz (variant 1: z = z0 + GaussianMeanLearner(onehot) carries a gradient): the real pass's input
    // gradient plus the penalty's, which reaches z through interp = z + alpha (fv - z), i.e. times (1 - alpha);
    // ab1 is d(gp_weight * gp)/d(layer-1 pre-activation) of the interpolate pass
    s.d1[t] = ab1;
    __syncthreads();
    if (t < XD) {
      float dgp = 0.f;
#pragma unroll 8
      for (int o = 0; o < H1; ++o) dgp = fmaf(s.W1[o * P1 + t], s.d1[o], dgp);
      dx[(long)row * XD + t] = fmaf(1.f - alpha, dgp, du_real);
    }
  }
  if (t == 0) {
    rec[R_C] = 0.f;
    out[(long)row * 3 + 0] = fr.D;
    out[(long)row * 3 + 1] = ff.D;
    out[(long)row * 3 + 2] = gprow;
  }
}

// parameter-parallel reduction of the per-row outer products
__global__ __launch_bounds__(256) void disc_param_grad_kernel(const float* __restrict__ recs,
                                                              const float* __restrict__ gpx, int npass,
                                                              int B, int K, float* dW1, float* db1,
                                                              float* dW2, float* db2, float* dW3,
                                                              float* db3) {
  const int IN = XD + K;
  const int nW1 = H1 * IN, nW2 = H2 * H1;
  const int total = nW1 + H1 + nW2 + H2 + H2 + 1;
  const int idx = blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const long nrec = (long)npass * B;
  // fp64 accumulation: analytically-zero sums (db3 = sum(+1/B) + sum(-1/B)) must come out
  // EXACTLY zero, as they do in the reference -- Adam turns any rounding residue into a +-lr step
  double acc = 0.0;
  if (idx < nW1) {
    const int o = idx / IN, i = idx - o * IN;
    // (unrolled: the loads of 8 records are in flight together -- one at a time this 19-workgroup kernel was a chain of
    // ~200 dependent L2 round trips, 217 us on the critic stream)
#pragma unroll 8
    for (long r = 0; r < nrec; ++r) acc += (double)recs[r * REC + R_D1 + o] * (double)recs[r * REC + R_U + i];
    if (gpx && i < XD)
#pragma unroll 8
      for (int r = 0; r < B; ++r) acc += (double)gpx[(long)r * GPX + G_S1 + o] * (double)gpx[(long)r * GPX + G_GB + i];
    if (dW1) dW1[idx] = (float)acc;
    return;
  }
  int j = idx - nW1;
  if (j < H1) {
#pragma unroll 8
    for (long r = 0; r < nrec; ++r) acc += recs[r * REC + R_D1 + j];
    if (db1) db1[j] = (float)acc;
    return;
  }
  j -= H1;
  if (j < nW2) {
    const int p = j >> 6, o = j & 63;
#pragma unroll 8
    for (long r = 0; r < nrec; ++r) acc += (double)recs[r * REC + R_D2 + p] * (double)recs[r * REC + R_H1 + o];
    if (gpx)
#pragma unroll 8
      for (int r = 0; r < B; ++r) acc += (double)gpx[(long)r * GPX + G_S2 + p] * (double)gpx[(long)r * GPX + G_RB1 + o];
    if (dW2) dW2[j] = (float)acc;
    return;
  }
  j -= nW2;
  if (j < H2) {
#pragma unroll 8
    for (long r = 0; r < nrec; ++r) acc += recs[r * REC + R_D2 + j];
    if (db2) db2[j] = (float)acc;
    return;
  }
  j -= H2;
  if (j < H2) {
#pragma unroll 8
    for (long r = 0; r < nrec; ++r) acc += recs[r * REC + R_H2C + j];
    if (dW3) dW3[j] = (float)acc;
    return;
  }
  for (long r = 0; r < nrec; ++r) acc += recs[r * REC + R_C];
  if (db3) db3[0] = (float)acc;
}

__global__ __launch_bounds__(256) void disc_loss_kernel(const float* __restrict__ rowvals, int B,
                                                        float gp_weight, float* losses) {
  __shared__ double red[3][4];
  double a = 0.0, b = 0.0, c = 0.0;
  for (int r = threadIdx.x; r < B; r += 256) {
    a += rowvals[r * 3 + 0];
    b += rowvals[r * 3 + 1];
    c += rowvals[r * 3 + 2];
  }
  a = wave_sum_d(a); b = wave_sum_d(b); c = wave_sum_d(c);
  if ((threadIdx.x & 63) == 0) {
    red[0][threadIdx.x >> 6] = a;
    red[1][threadIdx.x >> 6] = b;
    red[2][threadIdx.x >> 6] = c;
  }
  __syncthreads();
  if (threadIdx.x == 0) {
    const double real = (red[0][0] + red[0][1] + red[0][2] + red[0][3]) / B;
    const double fake = (red[1][0] + red[1][1] + red[1][2] + red[1][3]) / B;
    const double gp = (red[2][0] + red[2][1] + red[2][2] + red[2][3]) / B;
    losses[0] = (float)(fake - real + (double)gp_weight * gp);
    losses[1] = (float)gp;
  }
}

inline bool disc_args_ok(int B, int K) { return B >= 1 && K >= 0 && K <= MAXK; }

}  // namespace

extern "C" size_t pcaa_disc_workspace_bytes(int B, int K) {
  (void)K;
  if (B < 1) return 0;
  return ((size_t)3 * B * REC + (size_t)B * GPX + (size_t)B * 3) * sizeof(float);
}

extern "C" int pcaa_disc_forward(const float* x, const float* label, int B, int K, const float* W1,
                                 const float* b1, const float* W2, const float* b2, const float* W3,
                                 const float* b3, float* out, void* stream) {
  PCAA_CHECK_ARG(disc_args_ok(B, K), "pcaa_disc_forward: bad B=%d K=%d (K<=%d)", B, K, MAXK);
  PCAA_CHECK_ARG(x && (label || K == 0) && W1 && b1 && W2 && b2 && W3 && b3 && out, "pcaa_disc_forward: null pointer");
  DiscParams p{W1, b1, W2, b2, W3, b3};
  hipLaunchKernelGGL(disc_rows_kernel<0>, dim3(B), dim3(64), 0, as_stream(stream), x, (const float*)nullptr,
                     label, (const float*)nullptr, B, K, p, 0.f, out, (float*)nullptr, (float*)nullptr,
                     (float*)nullptr, (float*)nullptr);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_disc_forward");
}

extern "C" int pcaa_disc_backward(const float* x, const float* label, int B, int K, const float* W1,
                                  const float* b1, const float* W2, const float* b2, const float* W3,
                                  const float* b3, const float* gout, float* dx, float* dlabel,
                                  float* dW1, float* db1, float* dW2, float* db2, float* dW3, float* db3,
                                  float* workspace, size_t workspace_bytes, void* stream) {
  PCAA_CHECK_ARG(disc_args_ok(B, K), "pcaa_disc_backward: bad B=%d K=%d", B, K);
  PCAA_CHECK_ARG(x && (label || K == 0) && W1 && b1 && W2 && b2 && W3 && b3 && gout, "pcaa_disc_backward: null pointer");
  const bool want_params = dW1 || db1 || dW2 || db2 || dW3 || db3;
  PCAA_CHECK_ARG(!want_params || (workspace && workspace_bytes >= (size_t)B * REC * sizeof(float)),
                 "pcaa_disc_backward: workspace too small");
  DiscParams p{W1, b1, W2, b2, W3, b3};
  hipStream_t s = as_stream(stream);
  hipLaunchKernelGGL(disc_rows_kernel<1>, dim3(B), dim3(64), 0, s, x, (const float*)nullptr, label, gout, B, K,
                     p, 0.f, (float*)nullptr, dx, dlabel, want_params ? workspace : (float*)nullptr,
                     (float*)nullptr);
  if (want_params) {
    const int total = H1 * (XD + K) + H1 + H2 * H1 + H2 + H2 + 1;
    hipLaunchKernelGGL(disc_param_grad_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, workspace,
                       (const float*)nullptr, 1, B, K, dW1, db1, dW2, db2, dW3, db3);
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_disc_backward");
}

// double backward: gradients of <gbar, g> with g = d(sum_b gout_b D(x_b, label_b)) / dx (the dx of
// pcaa_disc_backward).  d_b3 is identically zero and is written as such.  dx2 / dlabel2 / dgout nullable.
extern "C" int pcaa_disc_backward_backward(const float* x, const float* label, int B, int K, const float* W1,
                                           const float* b1, const float* W2, const float* b2, const float* W3,
                                           const float* b3, const float* gout, const float* gbar, float* dx2,
                                           float* dlabel2, float* dgout, float* dW1, float* db1, float* dW2,
                                           float* db2, float* dW3, float* db3, float* workspace,
                                           size_t workspace_bytes, void* stream) {
  PCAA_CHECK_ARG(disc_args_ok(B, K), "pcaa_disc_backward_backward: bad B=%d K=%d", B, K);
  PCAA_CHECK_ARG(x && (label || K == 0) && W1 && b1 && W2 && b2 && W3 && b3 && gout && gbar,
                 "pcaa_disc_backward_backward: null pointer");
  PCAA_CHECK_ARG(workspace && workspace_bytes >= pcaa_disc_workspace_bytes(B, K),
                 "pcaa_disc_backward_backward: workspace too small");
  DiscParams p{W1, b1, W2, b2, W3, b3};
  hipStream_t s = as_stream(stream);
  float* recs = workspace;
  float* gpx = recs + (size_t)3 * B * REC;
  hipLaunchKernelGGL(disc_rows_kernel<3>, dim3(B), dim3(64), 0, s, x, gbar, label, gout, B, K, p, 0.f, dgout, dx2,
                     dlabel2, recs, gpx);
  if (dW1 || db1 || dW2 || db2 || dW3 || db3) {
    const int total = H1 * (XD + K) + H1 + H2 * H1 + H2 + H2 + 1;
    hipLaunchKernelGGL(disc_param_grad_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, recs, gpx, 1, B,
                       K, dW1, db1, dW2, db2, dW3, db3);
  }
  PCAA_RETURN_LAUNCH_STATUS("pcaa_disc_backward_backward");
}

extern "C" int pcaa_disc_wgan_gp(const float* z, const float* fv, const float* label, const float* alphas,
                                 int B, int K, const float* W1, const float* b1, const float* W2,
                                 const float* b2, const float* W3, const float* b3, float gp_weight,
                                 float* losses, float* dW1, float* db1, float* dW2, float* db2, float* dW3,
                                 float* db3, float* dz, float* workspace, size_t workspace_bytes, void* stream) {
  PCAA_CHECK_ARG(disc_args_ok(B, K), "pcaa_disc_wgan_gp: bad B=%d K=%d", B, K);
  PCAA_CHECK_ARG(z && fv && (label || K == 0) && alphas && W1 && b1 && W2 && b2 && W3 && b3 && losses,
                 "pcaa_disc_wgan_gp: null pointer");
  PCAA_CHECK_ARG(workspace && workspace_bytes >= pcaa_disc_workspace_bytes(B, K), "pcaa_disc_wgan_gp: workspace too small");
  DiscParams p{W1, b1, W2, b2, W3, b3};
  hipStream_t s = as_stream(stream);
  float* recs = workspace;
  float* gpx = recs + (size_t)3 * B * REC;
  float* rowvals = gpx + (size_t)B * GPX;
  hipLaunchKernelGGL(disc_rows_kernel<2>, dim3(B), dim3(64), 0, s, z, fv, label, alphas, B, K, p, gp_weight,
                     rowvals, dz, (float*)nullptr, recs, gpx);
  const int total = H1 * (XD + K) + H1 + H2 * H1 + H2 + H2 + 1;
  hipLaunchKernelGGL(disc_param_grad_kernel, dim3((unsigned)cdiv(total, 256)), dim3(256), 0, s, recs, gpx, 3, B,
                     K, dW1, db1, dW2, db2, dW3, db3);
  hipLaunchKernelGGL(disc_loss_kernel, dim3(1), dim3(256), 0, s, rowvals, B, gp_weight, losses);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_disc_wgan_gp");
}
