// The MLP heads of CGEncoder (reference models.py:252-277, 285-292) and the decoder projection head
// (PCAA_ablation.py:778-781) as ONE forward and ONE backward kernel.
//
//   x4 [B,512] -> sup_fv = ELU(W1 x4 + b1) [B,32] -> h = ELU(Wh sup_fv + bh) [B,16] (projection head, optional)
//              -> logits = ELU(W2 h + b2) [B,K];   hproj = ELU(Wg sup_fv + bg) [B,64] (decoder head, optional)
//
// These layers hold 0.002 % of the step's FLOPs but were ~30 launches of 5-70 us each on the critical
// path of the step (a 128x128-tile MFMA GEMM, a bias/ELU pass, a column sum and an ELU' pass per layer
// and direction): latency, not work.  Here the forward is one workgroup per 4 batch rows, the backward
// one workgroup per 32-column slice of the 512-wide input (every workgroup recomputes the tiny
// upstream chain for all rows, then forms its slice of dx4 and dW1; the small weight gradients are
// spread over the workgroups).  Plain fp32 FMAs: exact-fp32 in both precision modes.
#include "common.h"

namespace {

constexpr int HB_MAX_ROWS = 64;    // backward: all batch rows of the chain live in LDS (74 KB)
constexpr int D_IN = 512, D_SUP = 32, D_HEAD = 16, D_PROJ = 64;

struct HeadsParams {
  const float* x4;       // [B, 512]
  const float* W1; const float* b1;     // [32, 512], [32]
  const float* Wh; const float* bh;     // [16, 32], [16]   (null: no projection head, W2 is [K, 32])
  const float* W2; const float* b2;     // [K, 16 | 32], [K]
  const float* Wg; const float* bg;     // [64, 32], [64]   (null: no decoder projection head)
  float* sup_fv; float* h; float* logits; float* hproj;
  int B, K;
};

// ------------------------------------------------------------------ forward: 4 rows per workgroup
__global__ __launch_bounds__(256) void heads_fwd_kernel(HeadsParams p) {
  __shared__ __attribute__((aligned(16))) float xs[4][D_IN];
  __shared__ float sup[4][D_SUP];
  __shared__ float hh[4][D_HEAD];
  const int tid = threadIdx.x;
  const int row0 = blockIdx.x * 4;
  // stage the 4 input rows (2048 floats, 2 float4 per thread)
#pragma unroll
  for (int c = 0; c < 2; ++c) {
    const int q = tid + c * 256;            // float4 index 0..511
    const int r = q >> 7, k4 = (q & 127) << 2;
    f32x4 v = {0.f, 0.f, 0.f, 0.f};
    if (row0 + r < p.B) v = load4(p.x4 + (long)(row0 + r) * D_IN + k4);
    *reinterpret_cast<f32x4*>(&xs[r][k4]) = v;
  }
  __syncthreads();
  {
    // sup1: thread = (row r, output o, k-half): 256-long dot product, pairs combined by a shuffle
    const int r = tid >> 6, o = (tid & 63) >> 1, kh = tid & 1;
    const float* w = p.W1 + (long)o * D_IN + kh * 256;
    const float* x = &xs[r][kh * 256];
    float acc = 0.f;
#pragma unroll 8
    for (int k = 0; k < 256; k += 4) {
      const f32x4 wv = load4(w + k);
      const f32x4 xv = *reinterpret_cast<const f32x4*>(x + k);
      acc = fmaf(wv.x, xv.x, acc);
      acc = fmaf(wv.y, xv.y, acc);
      acc = fmaf(wv.z, xv.z, acc);
      acc = fmaf(wv.w, xv.w, acc);
    }
    acc += __shfl_xor(acc, 1, 64);
    if (kh == 0) {
      const float v = elu_f(acc + p.b1[o]);
      sup[r][o] = v;
      if (row0 + r < p.B) p.sup_fv[(long)(row0 + r) * D_SUP + o] = v;
    }
  }
  __syncthreads();
  // decoder projection head: 4 rows x 64 outputs = 256 threads
  if (p.Wg != nullptr) {
    const int r = tid >> 6, o = tid & 63;
    float acc = p.bg[o];
#pragma unroll
    for (int k = 0; k < D_SUP; ++k) acc = fmaf(p.Wg[o * D_SUP + k], sup[r][k], acc);
    if (row0 + r < p.B) p.hproj[(long)(row0 + r) * D_PROJ + o] = elu_f(acc);
  }
  const int din2 = p.Wh != nullptr ? D_HEAD : D_SUP;
  if (p.Wh != nullptr) {
    if (tid < 4 * D_HEAD) {
      const int r = tid >> 4, o = tid & 15;
      float acc = p.bh[o];
#pragma unroll
      for (int k = 0; k < D_SUP; ++k) acc = fmaf(p.Wh[o * D_SUP + k], sup[r][k], acc);
      const float v = elu_f(acc);
      hh[r][o] = v;
      if (row0 + r < p.B) p.h[(long)(row0 + r) * D_HEAD + o] = v;
    }
    __syncthreads();
  }
  for (int q = tid; q < 4 * p.K; q += 256) {
    const int r = q / p.K, o = q - r * p.K;
    float acc = p.b2[o];
    for (int k = 0; k < din2; ++k) acc = fmaf(p.W2[o * din2 + k], p.Wh != nullptr ? hh[r][k] : sup[r][k], acc);
    if (row0 + r < p.B) p.logits[(long)(row0 + r) * p.K + o] = elu_f(acc);
  }
}

// ------------------------------------------------------------------ backward
struct HeadsBwdParams {
  const float* x4; const float* sup_fv; const float* h; const float* logits; const float* hproj;
  const float* W1; const float* Wh; const float* W2; const float* Wg;
  const float* d_logits;   // [B,K] gradient w.r.t. the ELU'd logits, or null
  const float* d_sup;      // [B,32] gradient arriving at sup_fv from elsewhere (critic, decoder), or null
  const float* d_hproj;    // [B,64] gradient w.r.t. hproj (decoder side), or null
  float* dW1; float* db1; float* dWh; float* dbh; float* dW2; float* db2; float* dWg; float* dbg;
  float* dx4;              // [B,512]
  int B, K;
};

constexpr int HB_SLICE = 32;      // input columns per workgroup -> 16 workgroups

// LDS plan of the backward kernel (floats): everything the chain reads from global memory is staged in ONE burst
// at the top of the kernel.  The kernel runs beside the decoder's side-stream Adam (4.4 GB at ~5.5 TB/s): with its
// weights and activations read from global memory inside each of its five barrier-separated phases, every phase
// paid a loaded-memory round trip of tens of microseconds (233 us on the step's critical path, 45 us alone); staged
// up front it pays one.
struct HbLds {
  static constexpr int R = HB_MAX_ROWS;
  static constexpr int DL = 0;                         // [R][9]   d(pre-activation of the logits)
  static constexpr int DH = DL + R * 9;                // [R][17]  d(pre-activation of h)
  static constexpr int DG = DH + R * 17;               // [R][65]  d(pre-activation of hproj)
  static constexpr int DS = DG + R * 65;               // [R][33]  d(pre-activation of sup_fv)
  static constexpr int XS = DS + R * 33;               // [R][33]  this workgroup's slice of x4
  static constexpr int SUP = XS + R * 33;              // [R][33]  sup_fv
  static constexpr int HH = SUP + R * 33;              // [R][17]  h
  static constexpr int DSI = HH + R * 17;              // [R][33]  d_sup (incoming)
  static constexpr int W2 = DSI + R * 33;              // [8][33]
  static constexpr int WH = W2 + 8 * 33;               // [16][33]
  static constexpr int WG = WH + 16 * 33;              // [64][33]
  static constexpr int W1 = WG + 64 * 33;              // [32][33] rows of W1, this workgroup's columns
  static constexpr int TOTAL = W1 + 32 * 33;
};
constexpr int HB_LDS_BYTES = HbLds::TOTAL * 4;

__global__ __launch_bounds__(256) void heads_bwd_kernel(HeadsBwdParams p) {
  extern __shared__ __attribute__((aligned(16))) float hb[];
  float* dl = hb + HbLds::DL;
  float* dh = hb + HbLds::DH;
  float* dg = hb + HbLds::DG;
  float* ds = hb + HbLds::DS;
  float* xs = hb + HbLds::XS;
  float* sups = hb + HbLds::SUP;
  float* hs = hb + HbLds::HH;
  float* dsi = hb + HbLds::DSI;
  float* W2s = hb + HbLds::W2;
  float* Whs = hb + HbLds::WH;
  float* Wgs = hb + HbLds::WG;
  float* W1s = hb + HbLds::W1;
  const int tid = threadIdx.x, B = p.B, K = p.K;
  const bool head = p.Wh != nullptr, proj = p.Wg != nullptr && p.d_hproj != nullptr;
  const int din2 = head ? D_HEAD : D_SUP;
  const int c0 = blockIdx.x * HB_SLICE;

  // ---- (0) one burst of global reads: activations, incoming gradients, weights
  for (int q = tid; q < B * (HB_SLICE / 4); q += 256) {
    const int r = q / (HB_SLICE / 4), c4 = (q - r * (HB_SLICE / 4)) << 2;
    const f32x4 v = load4(p.x4 + (long)r * D_IN + c0 + c4);
    xs[r * 33 + c4 + 0] = v.x; xs[r * 33 + c4 + 1] = v.y; xs[r * 33 + c4 + 2] = v.z; xs[r * 33 + c4 + 3] = v.w;
  }
  for (int q = tid; q < B * K; q += 256) {
    const int r = q / K, o = q - r * K;
    dl[r * 9 + o] = p.d_logits != nullptr ? p.d_logits[q] * elu_grad_from_out(p.logits[q]) : 0.f;
  }
  if (proj)
    for (int q = tid; q < B * D_PROJ; q += 256)
      dg[(q >> 6) * 65 + (q & 63)] = p.d_hproj[q] * elu_grad_from_out(p.hproj[q]);
  for (int q = tid; q < B * D_SUP; q += 256) {
    sups[(q >> 5) * 33 + (q & 31)] = p.sup_fv[q];
    dsi[(q >> 5) * 33 + (q & 31)] = p.d_sup != nullptr ? p.d_sup[q] : 0.f;
  }
  if (head) {
    for (int q = tid; q < B * D_HEAD; q += 256) hs[(q >> 4) * 17 + (q & 15)] = p.h[q];
    for (int q = tid; q < D_HEAD * D_SUP; q += 256) Whs[(q >> 5) * 33 + (q & 31)] = p.Wh[q];
  }
  for (int q = tid; q < K * din2; q += 256) W2s[(q / din2) * 33 + (q % din2)] = p.W2[q];
  if (proj)
    for (int q = tid; q < D_PROJ * D_SUP; q += 256) Wgs[(q >> 5) * 33 + (q & 31)] = p.Wg[q];
  for (int q = tid; q < D_SUP * HB_SLICE; q += 256) W1s[(q >> 5) * 33 + (q & 31)] = p.W1[(long)(q >> 5) * D_IN + c0 + (q & 31)];
  __syncthreads();

  // ---- (2) projection head
  if (head) {
    for (int q = tid; q < B * D_HEAD; q += 256) {
      const int r = q >> 4, j = q & 15;
      float acc = 0.f;
      for (int o = 0; o < K; ++o) acc = fmaf(dl[r * 9 + o], W2s[o * 33 + j], acc);
      dh[r * 17 + j] = acc * elu_grad_from_out(hs[r * 17 + j]);
    }
    __syncthreads();
  }
  // ---- (3) sup_fv: everything that arrives there, then through its ELU
  for (int q = tid; q < B * D_SUP; q += 256) {
    const int r = q >> 5, i = q & 31;
    float acc = dsi[r * 33 + i];
    if (head) {
#pragma unroll
      for (int j = 0; j < D_HEAD; ++j) acc = fmaf(dh[r * 17 + j], Whs[j * 33 + i], acc);
    } else {
      for (int o = 0; o < K; ++o) acc = fmaf(dl[r * 9 + o], W2s[o * 33 + i], acc);
    }
    if (proj) {
#pragma unroll 8
      for (int g = 0; g < D_PROJ; ++g) acc = fmaf(dg[r * 65 + g], Wgs[g * 33 + i], acc);
    }
    ds[r * 33 + i] = acc * elu_grad_from_out(sups[r * 33 + i]);
  }
  __syncthreads();

  // ---- (4) this workgroup's slice of the wide layer: dx4[:, c0:c0+32] and dW1[:, c0:c0+32]
  for (int q = tid; q < B * HB_SLICE; q += 256) {
    const int r = q >> 5, c = q & 31;
    float acc = 0.f;
#pragma unroll
    for (int i = 0; i < D_SUP; ++i) acc = fmaf(ds[r * 33 + i], W1s[i * 33 + c], acc);
    p.dx4[(long)r * D_IN + c0 + c] = acc;
  }
  for (int q = tid; q < D_SUP * HB_SLICE; q += 256) {
    const int i = q >> 5, c = q & 31;
    float acc = 0.f;
    for (int r = 0; r < B; ++r) acc = fmaf(ds[r * 33 + i], xs[r * 33 + c], acc);
    p.dW1[(long)i * D_IN + c0 + c] = acc;
  }

  // ---- (5) the small gradients, spread over the workgroups by a running job index
  const int nwg = gridDim.x;
  int job = 0;
  auto mine = [&](int j) { return (j % nwg) == (int)blockIdx.x; };
  if (mine(job++)) {            // db1, db2, dbh, dbg: column sums
    for (int q = tid; q < D_SUP + K + (head ? D_HEAD : 0) + (proj ? D_PROJ : 0); q += 256) {
      float acc = 0.f;
      if (q < D_SUP) {
        for (int r = 0; r < B; ++r) acc += ds[r * 33 + q];
        p.db1[q] = acc;
      } else if (q < D_SUP + K) {
        const int o = q - D_SUP;
        for (int r = 0; r < B; ++r) acc += dl[r * 9 + o];
        p.db2[o] = acc;
      } else if (head && q < D_SUP + K + D_HEAD) {
        const int j = q - D_SUP - K;
        for (int r = 0; r < B; ++r) acc += dh[r * 17 + j];
        p.dbh[j] = acc;
      } else {
        const int g = q - D_SUP - K - (head ? D_HEAD : 0);
        for (int r = 0; r < B; ++r) acc += dg[r * 65 + g];
        p.dbg[g] = acc;
      }
    }
  }
  if (mine(job++)) {            // dW2 [K, din2] = dl^T . (h | sup_fv)
    for (int q = tid; q < K * din2; q += 256) {
      const int o = q / din2, j = q - o * din2;
      float acc = 0.f;
      for (int r = 0; r < B; ++r) acc = fmaf(dl[r * 9 + o], head ? hs[r * 17 + j] : sups[r * 33 + j], acc);
      p.dW2[q] = acc;
    }
  }
  if (head && mine(job++)) {    // dWh [16, 32] = dh^T . sup_fv
    for (int q = tid; q < D_HEAD * D_SUP; q += 256) {
      const int j = q >> 5, i = q & 31;
      float acc = 0.f;
      for (int r = 0; r < B; ++r) acc = fmaf(dh[r * 17 + j], sups[r * 33 + i], acc);
      p.dWh[q] = acc;
    }
  }
  if (proj) {                   // dWg [64, 32] = dg^T . sup_fv, in 4 jobs of 16 rows
    for (int part = 0; part < 4; ++part) {
      if (!mine(job++)) continue;
      for (int q = tid; q < 16 * D_SUP; q += 256) {
        const int g = part * 16 + (q >> 5), i = q & 31;
        float acc = 0.f;
        for (int r = 0; r < B; ++r) acc = fmaf(dg[r * 65 + g], sups[r * 33 + i], acc);
        p.dWg[g * D_SUP + i] = acc;
      }
    }
  }
}

}  // namespace

extern "C" int pcaa_heads_supported(int B, int K, int d_in, int d_sup, int d_head, int d_proj, int backward) {
  if (d_in != D_IN || d_sup != D_SUP || (d_head != 0 && d_head != D_HEAD) || (d_proj != 0 && d_proj != D_PROJ))
    return 0;
  if (B < 1 || K < 1) return 0;
  if (backward && (B > HB_MAX_ROWS || K > 8)) return 0;
  return 1;
}

extern "C" int pcaa_heads_fwd(const float* x4, const float* W1, const float* b1, const float* Wh, const float* bh,
                              const float* W2, const float* b2, const float* Wg, const float* bg, float* sup_fv,
                              float* h, float* logits, float* hproj, int B, int K, void* stream) {
  PCAA_CHECK_ARG(x4 && W1 && b1 && W2 && b2 && sup_fv && logits && B >= 1 && K >= 1, "pcaa_heads_fwd: bad args");
  PCAA_CHECK_ARG((Wh == nullptr) == (bh == nullptr) && (Wh == nullptr) == (h == nullptr),
                 "pcaa_heads_fwd: Wh, bh and h go together");
  PCAA_CHECK_ARG((Wg == nullptr) == (bg == nullptr) && (Wg == nullptr) == (hproj == nullptr),
                 "pcaa_heads_fwd: Wg, bg and hproj go together");
  PCAA_CHECK_ARG(((uintptr_t)x4 % 16) == 0 && ((uintptr_t)W1 % 16) == 0, "pcaa_heads_fwd: x4 / W1 must be 16-B aligned");
  HeadsParams p{x4, W1, b1, Wh, bh, W2, b2, Wg, bg, sup_fv, h, logits, hproj, B, K};
  hipLaunchKernelGGL(heads_fwd_kernel, dim3((unsigned)cdiv(B, 4)), dim3(256), 0, as_stream(stream), p);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_heads_fwd");
}

extern "C" int pcaa_heads_bwd(const float* x4, const float* sup_fv, const float* h, const float* logits,
                              const float* hproj, const float* W1, const float* Wh, const float* W2,
                              const float* Wg, const float* d_logits, const float* d_sup, const float* d_hproj,
                              float* dW1, float* db1, float* dWh, float* dbh, float* dW2, float* db2, float* dWg,
                              float* dbg, float* dx4, int B, int K, void* stream) {
  PCAA_CHECK_ARG(x4 && sup_fv && logits && W1 && W2 && dW1 && db1 && dW2 && db2 && dx4, "pcaa_heads_bwd: bad args");
  PCAA_CHECK_ARG(B >= 1 && B <= HB_MAX_ROWS && K >= 1 && K <= 8, "pcaa_heads_bwd: B <= %d and K <= 8", HB_MAX_ROWS);
  PCAA_CHECK_ARG((Wh == nullptr) == (h == nullptr) && (Wh == nullptr) == (dWh == nullptr) &&
                 (Wh == nullptr) == (dbh == nullptr), "pcaa_heads_bwd: Wh, h, dWh and dbh go together");
  PCAA_CHECK_ARG((d_hproj == nullptr) || (Wg && hproj && dWg && dbg), "pcaa_heads_bwd: d_hproj needs Wg, hproj, dWg, dbg");
  PCAA_CHECK_ARG(((uintptr_t)x4 % 16) == 0, "pcaa_heads_bwd: x4 must be 16-B aligned");
  HeadsBwdParams p{x4, sup_fv, h, logits, hproj, W1, Wh, W2, Wg, d_logits, d_sup, d_hproj,
                   dW1, db1, dWh, dbh, dW2, db2, dWg, dbg, dx4, B, K};
  static bool configured = false;
  if (!configured) {
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(heads_bwd_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            HB_LDS_BYTES) != hipSuccess) {
      pcaa_set_error("pcaa_heads_bwd: cannot raise the LDS limit to %d bytes", HB_LDS_BYTES);
      return PCAA_ERR_LAUNCH;
    }
    configured = true;
  }
  hipLaunchKernelGGL(heads_bwd_kernel, dim3(D_IN / HB_SLICE), dim3(256), HB_LDS_BYTES, as_stream(stream), p);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_heads_bwd");
}
