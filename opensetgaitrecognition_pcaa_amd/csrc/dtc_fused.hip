// TemporalConvolutionBlock forward (reference models.py:37-79, 108-160), one launch per layer:
//
//   y_l[(b,t)][co] = sum_{ci,tap} W_l[co][ci][tap] * a_{l-1}[b][t-(2-tap)*d][ci]      (causal, zero for t<0)
//   a_{l-1} = ELU(scale_{l-1} * y_{l-1} + shift_{l-1})   applied once, while the sequence is staged (layer 1: the input as is)
//
// The block holds 0.08 % of the step's FLOPs (2.2 GFLOP forward at B=64) in six layers that each need
// the BatchNorm statistics of ALL B*T rows of the layer before: the unfused path spent 310 us here in 29
// launches (im2col, 128x128-tile MFMA GEMM, split-K reduction, finalize, BN+ELU pass per layer).  This
// kernel does BN/ELU-on-load + implicit im2col + the contraction + the BatchNorm statistics of its own
// output in one launch on the exact-fp32 MFMA (v_mfma_f32_32x32x2_f32: an fmaf chain, so fp32 mode keeps
// its parity and bf16 mode shares the path).  One workgroup = one sequence (T <= 32 rows) x 32 output
// channels:
//   * the activated sequence tile a[T][channels] is staged in LDS ONCE (coalesced 16-B loads, one ELU per
//     element); the causal shifts of the three taps are row offsets of the MFMA's A-fragment reads into
//     that tile (a dedicated zero row serves t < 0), so im2col never exists as data on the forward path;
//   * the contraction runs tap-major inside 32-channel chunks (k' = tap*32 + ci: any order is valid as
//     long as both operands share it), so a lane's four consecutive k' are four consecutive channels:
//     one 16-B LDS read per operand feeds four MFMAs;
//   * per chunk only the weights move: 32 x 96 floats, contiguous in HBM, scattered tap-major into LDS;
//   * the four waves split each chunk's contraction on the same 32x32 tile and are combined through
//     LDS at the end, where the tile is stored and its column sums go to the fp64 statistics.
// Earlier forms, measured at B=64 (us for the 256->512 layer): gathering im2col elements one by one
// with BN+ELU per gathered element 50-80 (24 scalar loads per thread and chunk: load-issue bound, not
// the expm1f), plain-FMA register tiles 80 (the non-packed fp32 VALU peaks at half the fp32-MFMA rate).
// The first column tile also writes the im2col matrix the BACKWARD's weight gradient contracts with.
#include "common.h"
#include "bn_tail.h"

// LAB builds only (PCAA_HIPCC_EXTRA=-DPCAA_DTC_TRACE, tools/dtc_lab.py --trace): wave 0 of every workgroup adds the clock
// ticks it spent in each phase of the kernel to a device array
#ifdef PCAA_DTC_TRACE
__device__ unsigned long long g_dtc_trace[16];
#define DTC_T0() unsigned long long t_prev = __builtin_readcyclecounter()
#define DTC_MARK(i)                                                              \
  do {                                                                           \
    const unsigned long long t_now = __builtin_readcyclecounter();               \
    if (threadIdx.x == 0) atomicAdd(&g_dtc_trace[i], t_now - t_prev);            \
    t_prev = t_now;                                                              \
  } while (0)
extern "C" int pcaa_lab_dtc_trace(unsigned long long* out16, int reset) {
  if (out16 != nullptr && hipMemcpyFromSymbol(out16, HIP_SYMBOL(g_dtc_trace), sizeof(g_dtc_trace)) != hipSuccess) return 1;
  if (reset) {
    unsigned long long z[16] = {0};
    if (hipMemcpyToSymbol(HIP_SYMBOL(g_dtc_trace), z, sizeof(z)) != hipSuccess) return 1;
  }
  return 0;
}
#else
#define DTC_T0()
#define DTC_MARK(i)
#endif

namespace {

constexpr int ROWS = 32;   // rows (time steps) per workgroup, T <= ROWS

struct DtcFwdParams {
  const float* src;     // [B*T, cin]: block input (layer 1) or the previous layer's bias-free pre-BN output
  const float* scale;   // [cin] BatchNorm+ELU of the previous layer applied on load; null: src used as is
  const float* shift;
  const float* W;       // [cout, cin*3], k = ci*3 + tap
  float* y;             // [B*T, cout]
  float* col;           // [B*T, cin*3] (im2col of the activated input) or null
  double* stats;        // [nrep][2][cout] fp64 sums (sum, sum of squares) or null
  int B, T, cin, cout, dil, nrep;
  long slab_stride;     // gridDim.z > 1: split z writes its partial product to y + z*slab_stride
  BnTail tail;          // the BatchNorm finalize of ``stats``, run by the last workgroup (bn_tail.h); kind 0: none
};

constexpr int CC = 32;              // input channels per chunk: 3*CC = 96-deep contraction per trip
constexpr int WP = 3 * CC + 4;      // weight tile pitch: 36*row mod 64 walks all 16 four-bank groups
constexpr int MAX_CR = 256;         // channels of the activated sequence tile one workgroup keeps in LDS
constexpr int ZROW = ROWS;          // index of the all-zero row of that tile

// ELU for the staging path.  Relative error <= 3e-7 over the whole range: a degree-6 Taylor polynomial
// where exp(z)-1 would cancel (|z| < 0.25, truncation z^7/5040), v_exp_f32 elsewhere (|result| >= 0.22).
__device__ __forceinline__ float elu_stage(float z) {
  if (z > 0.f) return z;
  const float poly = z * (1.f + z * (0.5f + z * (1.f / 6 + z * (1.f / 24 + z * (1.f / 120 + z * (1.f / 720))))));
  return z > -0.25f ? poly : __expf(z) - 1.f;
}

// im2col of the activated input for the backward's weight gradient: col[(b,t)][ci*3+tap] = a[t-(2-tap)*d][ci], from the
// sequence tile in the LDS.  Round 4: the column tiles of a sequence SHARE the write (each takes a slice of the 16-B quads;
// it used to be the first tile's job alone, one 4-B store per element: 5-16 us of the launch on one workgroup in 16), 16 B
// per lane.  (cr is a multiple of 4, so a row's 3*cr floats are whole quads; K*4 and cz0*12 bytes keep them 16-B aligned.)
__device__ __forceinline__ void write_col(float* col, int K, const float* a_lds, int AP, int T, int cr, int d, int tid) {
  const int run4 = (cr * 3) >> 2, total = T * run4;
  const int per = (total + (int)gridDim.y - 1) / (int)gridDim.y;
  const int q_end = min(total, ((int)blockIdx.y + 1) * per);
  int q = (int)blockIdx.y * per + tid;
  const int dr = 256 / run4, dc = 256 - dr * run4;
  int r = q / run4, c = q - r * run4;                     // (row, quad in row), advanced by increments
  for (; q < q_end; q += 256) {
    const int k4 = c << 2;
    f32x4 v;
#pragma unroll
    for (int m = 0; m < 4; ++m) {
      const int kk = k4 + m, ci = kk / 3, tap = kk - ci * 3;
      const int ts = r - (2 - tap) * d;
      v[m] = ts >= 0 ? a_lds[ts * AP + ci] : 0.f;
    }
    store4(col + (long)r * K + k4, v);
    r += dr;
    c += dc;
    if (c >= run4) { c -= run4; ++r; }
  }
}

// gridDim.z > 1 splits the input channels over workgroups (the 1024->16 layer: 64 workgroups walking
// K = 3072 was one long chain of load latencies): split z writes its partial tile to y + z*slab_stride
// and pcaa_splitk_reduce_stats finishes the sum and the statistics.
__global__ __launch_bounds__(256) void dtc_fwd_kernel(DtcFwdParams p) {
  __shared__ __attribute__((aligned(16))) float a_lds[(ROWS + 1) * (MAX_CR + 4)];   // later: the 4 partial tiles
  __shared__ __attribute__((aligned(16))) float Ws[32 * WP];
  __shared__ float red[2][8][32];
  __shared__ int tail_flag;
  static_assert(4 * ROWS * 33 <= (ROWS + 1) * (MAX_CR + 4), "partial tiles alias the sequence tile");

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x, n0 = blockIdx.y * 32;
  const int T = p.T, cin = p.cin, K = cin * 3, d = p.dil;
  const bool split = gridDim.z > 1;
  float* yout = p.y + (long)blockIdx.z * p.slab_stride;
  // this workgroup's input-channel range [cz0, cz0+cr)
  const int per_z = ((cin + CC - 1) / CC + (int)gridDim.z - 1) / (int)gridDim.z * CC;
  const int cz0 = blockIdx.z * per_z;
  const int cr = max(0, min(cin, cz0 + per_z) - cz0);
  const int AP = cr + 4;

  DTC_T0();
  // ---- the activated sequence tile, once
  {
    const int q4 = cr >> 2;
    const bool act = p.scale != nullptr;
    for (int q = tid; q < T * q4; q += 256) {
      const int r = q / q4, c4 = (q - r * q4) << 2;
      f32x4 v = load4(p.src + ((long)b * T + r) * cin + cz0 + c4);
      if (act) {
        const f32x4 sc = load4(p.scale + cz0 + c4), sh = load4(p.shift + cz0 + c4);
        v.x = elu_stage(fmaf(sc.x, v.x, sh.x));
        v.y = elu_stage(fmaf(sc.y, v.y, sh.y));
        v.z = elu_stage(fmaf(sc.z, v.z, sh.z));
        v.w = elu_stage(fmaf(sc.w, v.w, sh.w));
      }
      *reinterpret_cast<f32x4*>(&a_lds[r * AP + c4]) = v;
    }
    for (int q = tid; q < (ROWS + 1 - T) * AP; q += 256) a_lds[T * AP + q] = 0.f;     // rows T..31 and the zero row
  }
  f32x4 rw[3];
  auto load_w = [&](int c0) {
    // 32 columns x 96 contiguous floats W[n0+c][(cz0+c0)*3 ...]: float4 q -> (column c = q / 24, run offset f)
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = tid + j * 256;
      const int c = q / 24, f = (q - c * 24) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + c < p.cout && c0 * 3 + f < cr * 3) v = load4(p.W + (long)(n0 + c) * K + (long)(cz0 + c0) * 3 + f);
      rw[j] = v;
    }
  };
  auto store_w = [&]() {
    // scatter tap-major: run element kk = ci_l*3 + tap -> Ws[c][tap*CC + ci_l]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = tid + j * 256;
      const int c = q / 24, f = (q - c * 24) << 2;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int kk = f + m, ci_l = kk / 3, tap = kk - ci_l * 3;
        Ws[c * WP + tap * CC + ci_l] = rw[j][m];
      }
    }
  };
  if (cr > 0) load_w(0);
  __syncthreads();
  DTC_MARK(0);      // staging
  if (p.col != nullptr) write_col(p.col + (long)b * T * K + (long)cz0 * 3, K, a_lds, AP, T, cr, d, tid);

  DTC_MARK(1);      // im2col
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < cr; c0 += CC) {
    if (c0 > 0) __syncthreads();          // previous trip's readers are done with the weight tile
    store_w();
    __syncthreads();
    if (c0 + CC < cr) load_w(c0 + CC);    // in flight while this chunk is consumed
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gg = wave * 3 + g;        // 12 groups of 8 per chunk: tap = gg / 4, 8 channels each
      const int tap = gg >> 2, ci_l = ((gg & 3) << 3) + (half << 2);
      const int rs = l31 - (2 - tap) * d;
      const float* ap = (rs >= 0 && c0 + ci_l < cr) ? &a_lds[rs * AP + c0 + ci_l] : &a_lds[ZROW * AP];
      const f32x4 av = *reinterpret_cast<const f32x4*>(ap);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(&Ws[l31 * WP + tap * CC + ci_l]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wv.w, acc, 0, 0, 0);
    }
  }
  DTC_MARK(2);      // contraction
  // combine the four waves' partial tiles (accumulator i of lane (l31, half) is row (i&3) + 8*(i>>2) + 4*half,
  // column l31); the sequence tile is dead by now
  __syncthreads();
  float (*part)[ROWS][33] = reinterpret_cast<float (*)[ROWS][33]>(a_lds);
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][(i & 3) + 8 * (i >> 2) + 4 * half][l31] = acc[i];
  __syncthreads();
  const int colx = tid & 31, rg = tid >> 5, gn = n0 + colx;
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rg + 8 * i;
    const float v = (part[0][r][colx] + part[1][r][colx]) + (part[2][r][colx] + part[3][r][colx]);
    if (r < T && gn < p.cout) {
      yout[((long)b * T + r) * p.cout + gn] = v;
      s1 += v;
      s2 += v * v;
    }
  }
  DTC_MARK(3);      // combine + store
  if (p.stats != nullptr && !split) {
    red[0][rg][colx] = s1;
    red[1][rg][colx] = s2;
    __syncthreads();
    if (tid < 64) {
      const int stat = tid >> 5, c = tid & 31;
      if (n0 + c < p.cout) {
        double v = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) v += (double)red[stat][g][c];
        unsafeAtomicAdd(&p.stats[((long)(b % p.nrep) * 2 + stat) * p.cout + n0 + c], v);
      }
    }
  }
  DTC_MARK(4);      // statistics
  bn_tail_run(p.tail, tid, 256, gridDim.x * gridDim.y * gridDim.z, &tail_flag);
  DTC_MARK(5);      // tail
}


// ------------------------------------------------------------------ round 4: the bf16 throughput mode's variant
// Same staging, same implicit im2col, same statistics -- but the contraction runs on v_mfma_f32_32x32x16_bf16 (the
// operands are rounded to bf16 as the fragments are built: the activated tile stays fp32 in the LDS, so the im2col
// matrix for the weight gradient is unchanged; the weight chunk is stored as bf16), and a workgroup owns 128 output
// channels -- one 32-column block per wave, the whole contraction in that wave, no cross-wave reduction -- so a
// sequence's tile is staged and activated by 4 workgroups instead of 16 (256 -> 512 layer).  A 96-deep chunk is 6 MFMAs
// of 32 cycles per wave instead of 12 of 64: the contraction leaves the kernel's critical path (layer 6: 22 us of 45).
constexpr int NCT = 128;            // output channels per workgroup
constexpr int WPH = 3 * CC + 8;     // bf16 pitch of a weight-tile row (208 B)

__device__ __forceinline__ bf16x8 cvt8(const float* p) {
  const f32x4 a = *reinterpret_cast<const f32x4*>(p), b = *reinterpret_cast<const f32x4*>(p + 4);
  bf16x8 r;
  r[0] = (bf16_t)a.x; r[1] = (bf16_t)a.y; r[2] = (bf16_t)a.z; r[3] = (bf16_t)a.w;
  r[4] = (bf16_t)b.x; r[5] = (bf16_t)b.y; r[6] = (bf16_t)b.z; r[7] = (bf16_t)b.w;
  return r;
}

__global__ __launch_bounds__(256) void dtc_fwd_bf16_kernel(DtcFwdParams p) {
  __shared__ __attribute__((aligned(16))) float a_lds[(ROWS + 1) * (MAX_CR + 4)];
  __shared__ __attribute__((aligned(16))) bf16_t Ws[NCT * WPH];
  __shared__ int tail_flag;

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x, n0 = blockIdx.y * NCT;
  const int T = p.T, cin = p.cin, K = cin * 3, d = p.dil;
  const bool split = gridDim.z > 1;
  float* yout = p.y + (long)blockIdx.z * p.slab_stride;
  const int per_z = ((cin + CC - 1) / CC + (int)gridDim.z - 1) / (int)gridDim.z * CC;
  const int cz0 = blockIdx.z * per_z;
  const int cr = max(0, min(cin, cz0 + per_z) - cz0);
  const int AP = cr + 4;
  {
    const int q4 = cr >> 2;
    const bool act = p.scale != nullptr;
    for (int q = tid; q < T * q4; q += 256) {
      const int r = q / q4, c4 = (q - r * q4) << 2;
      f32x4 v = load4(p.src + ((long)b * T + r) * cin + cz0 + c4);
      if (act) {
        const f32x4 sc = load4(p.scale + cz0 + c4), sh = load4(p.shift + cz0 + c4);
        v.x = elu_stage(fmaf(sc.x, v.x, sh.x));
        v.y = elu_stage(fmaf(sc.y, v.y, sh.y));
        v.z = elu_stage(fmaf(sc.z, v.z, sh.z));
        v.w = elu_stage(fmaf(sc.w, v.w, sh.w));
      }
      *reinterpret_cast<f32x4*>(&a_lds[r * AP + c4]) = v;
    }
    for (int q = tid; q < (ROWS + 1 - T) * AP; q += 256) a_lds[T * AP + q] = 0.f;
  }
  // weight chunk: 128 columns x 96 contiguous floats W[n0 + c][(cz0 + c0) * 3 ...], 12 float4 per thread
  f32x4 rw[12];
  auto load_w = [&](int c0) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int q = tid + j * 256;
      const int c = q / 24, f = (q - c * 24) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (n0 + c < p.cout && c0 * 3 + f < cr * 3) v = load4(p.W + (long)(n0 + c) * K + (long)(cz0 + c0) * 3 + f);
      rw[j] = v;
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int q = tid + j * 256;
      const int c = q / 24, f = (q - c * 24) << 2;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int kk = f + m, ci_l = kk / 3, tap = kk - ci_l * 3;
        Ws[c * WPH + tap * CC + ci_l] = (bf16_t)rw[j][m];
      }
    }
  };
  if (cr > 0) load_w(0);
  __syncthreads();
  if (p.col != nullptr) write_col(p.col + (long)b * T * K + (long)cz0 * 3, K, a_lds, AP, T, cr, d, tid);
  const bool active = n0 + wave * 32 < p.cout;      // (wave-uniform) this wave's 32 columns exist
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < cr; c0 += CC) {
    if (c0 > 0) __syncthreads();
    store_w();
    __syncthreads();
    if (c0 + CC < cr) load_w(c0 + CC);
    if (active) {
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        // k' = tap * 32 + ci_l: step st covers k' = 16 st .. 16 st + 15, this lane's eight are 16 st + 8 half ...
        const int tap = st >> 1, ci_l = ((st & 1) << 4) + (half << 3);
        const int rs = l31 - (2 - tap) * d;
        const float* ap = (rs >= 0 && c0 + ci_l < cr) ? &a_lds[rs * AP + c0 + ci_l] : &a_lds[ZROW * AP];
        const bf16x8 av = cvt8(ap);
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&Ws[(wave * 32 + l31) * WPH + tap * CC + ci_l]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wv, acc, 0, 0, 0);
      }
    }
  }
  // accumulator i of lane (l31, half) is row (i & 3) + 8 (i >> 2) + 4 half, column l31 of the wave's block
  const int gn = n0 + wave * 32 + l31;
  float s1 = 0.f, s2 = 0.f;
  if (active && gn < p.cout) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = (i & 3) + 8 * (i >> 2) + 4 * half;
      if (r < T) {
        const float v = acc[i];
        yout[((long)b * T + r) * p.cout + gn] = v;
        s1 += v;
        s2 += v * v;
      }
    }
  }
  if (p.stats != nullptr && !split) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (active && half == 0 && gn < p.cout) {
      unsafeAtomicAdd(&p.stats[((long)(b % p.nrep) * 2 + 0) * p.cout + gn], (double)s1);
      unsafeAtomicAdd(&p.stats[((long)(b % p.nrep) * 2 + 1) * p.cout + gn], (double)s2);
    }
  }
  bn_tail_run(p.tail, tid, 256, gridDim.x * gridDim.y * gridDim.z, &tail_flag);
}

// ------------------------------------------------------------------ backward w.r.t. the layer input
//   da[(b,t)][ci] = sum_{co,tap} dy[b][t+(2-tap)*d][co] * W[co][ci][tap]        (rows past the sequence end: zero)
// The adjoint of the forward as the same kernel shape: the gradient sequence tile dy[T][cout range] is staged in
// LDS once, the taps are row offsets (now forwards in time) of the A-fragment reads, the contraction runs over
// (tap, co) in 32-channel chunks and the weight tile of a chunk is the same 96-float run per co as in the
// forward, scattered TRANSPOSED (row = input channel ci, k' = tap*32 + co).  Replaces a 128x128-tile GEMM on a
// handful of workgroups (dcol = dy . W) plus the col2im pass that summed its three taps.
struct DtcDgradParams {
  const float* dy;      // [B*T, cout], or null: dy = coef0*dz + coef1*y + coef2 is formed while the tile is staged
  const float* dz;      // [B*T, cout]  d(loss)/d(pre-activation z) of THIS layer      (dy == null)
  const float* y;       // [B*T, cout]  this layer's bias-free pre-BN output             (dy == null)
  const float* coef;    // [3][cout]    pcaa_bn_bwd_finalize's coefficients             (dy == null)
  float* dy_out;        // [B*T, cout] or null: the staged dy, written by the first column tile (for the wgrad)
  const float* W;       // [cout, cin*3]
  float* out;           // [B*T, cin]: da, or with the epilogue dz of the layer BELOW (gridDim.z > 1: slabs)
  // epilogue (layer below): out = da * ELU'(ep_y*ep_scale + ep_shift), statistics {sum dz, sum dz*yhat}
  const float* ep_y; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
  double* ep_stats; int nrep;
  int B, T, cin, cout, dil;
  long slab_stride;
  BnTail tail;          // the BatchNorm-backward finalize of ``ep_stats`` (bn_tail.h); kind 0: none
};

constexpr int DG_MAX_CR = 512;                       // contraction channels per workgroup (dynamic LDS)
constexpr int DG_PART_FLOATS = 4 * ROWS * 33;        // the four waves' partial tiles alias the sequence tile
static inline int dg_tile_floats(int cr) { const int a = (ROWS + 1) * (cr + 4); return a > DG_PART_FLOATS ? a : DG_PART_FLOATS; }

__global__ __launch_bounds__(256) void dtc_dgrad_kernel(DtcDgradParams p, int tile_floats) {
  extern __shared__ __attribute__((aligned(16))) float dg_smem[];
  float* a_lds = dg_smem;                            // [(ROWS+1)][cr+4], later part[4][ROWS][33]
  float* Ws = dg_smem + tile_floats;                 // [32][WP]
  float* red = Ws + 32 * WP;                         // [2][8][32]

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x, n0 = blockIdx.y * 32;                 // n0: first input channel (output column)
  const int T = p.T, cin = p.cin, cout = p.cout, d = p.dil;
  const long wrow = (long)cin * 3;
  float* out = p.out + (long)blockIdx.z * p.slab_stride;
  const int per_z = ((cout + CC - 1) / CC + (int)gridDim.z - 1) / (int)gridDim.z * CC;
  const int cz0 = blockIdx.z * per_z;
  const int cr = max(0, min(cout, cz0 + per_z) - cz0);
  const int AP = cr + 4;
  DTC_T0();
  {
    const int q4 = cr >> 2;
    const bool form = p.dy == nullptr;
    const bool keep = p.dy_out != nullptr && blockIdx.y == 0;
    for (int q = tid; q < T * q4; q += 256) {
      const int r = q / q4, c4 = (q - r * q4) << 2;
      const long g = ((long)b * T + r) * cout + cz0 + c4;
      f32x4 v;
      if (form) {
        const f32x4 k0 = load4(p.coef + cz0 + c4), k1 = load4(p.coef + cout + cz0 + c4),
                    k2 = load4(p.coef + 2 * cout + cz0 + c4);
        v = k0 * load4(p.dz + g) + k1 * load4(p.y + g) + k2;
      } else {
        v = load4(p.dy + g);
      }
      *reinterpret_cast<f32x4*>(&a_lds[r * AP + c4]) = v;
      if (keep) store4(p.dy_out + g, v);
    }
    for (int q = tid; q < (ROWS + 1 - T) * AP; q += 256) a_lds[T * AP + q] = 0.f;
  }
  f32x4 rw[3];
  auto load_w = [&](int c0) {
    // 32 contraction channels co x 96 contiguous floats W[cz0+c0+co_l][(n0 ...)*3 ...]
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = tid + j * 256;
      const int co_l = q / 24, f = (q - co_l * 24) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c0 + co_l < cr && (long)n0 * 3 + f < wrow) v = load4(p.W + (long)(cz0 + c0 + co_l) * wrow + (long)n0 * 3 + f);
      rw[j] = v;
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int j = 0; j < 3; ++j) {
      const int q = tid + j * 256;
      const int co_l = q / 24, f = (q - co_l * 24) << 2;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int kk = f + m, ci_l = kk / 3, tap = kk - ci_l * 3;
        Ws[ci_l * WP + tap * CC + co_l] = rw[j][m];
      }
    }
  };
  if (cr > 0) load_w(0);
  __syncthreads();
  DTC_MARK(8);      // staging
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < cr; c0 += CC) {
    if (c0 > 0) __syncthreads();
    store_w();
    __syncthreads();
    if (c0 + CC < cr) load_w(c0 + CC);
#pragma unroll
    for (int g = 0; g < 3; ++g) {
      const int gg = wave * 3 + g;
      const int tap = gg >> 2, co_l = ((gg & 3) << 3) + (half << 2);
      const int rs = l31 + (2 - tap) * d;
      const float* ap = (rs < T && c0 + co_l < cr) ? &a_lds[rs * AP + c0 + co_l] : &a_lds[ZROW * AP];
      const f32x4 av = *reinterpret_cast<const f32x4*>(ap);
      const f32x4 wv = *reinterpret_cast<const f32x4*>(&Ws[l31 * WP + tap * CC + co_l]);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wv.x, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wv.y, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wv.z, acc, 0, 0, 0);
      acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wv.w, acc, 0, 0, 0);
    }
  }
  __syncthreads();
  DTC_MARK(9);      // contraction
  float (*part)[ROWS][33] = reinterpret_cast<float (*)[ROWS][33]>(a_lds);
#pragma unroll
  for (int i = 0; i < 16; ++i) part[wave][(i & 3) + 8 * (i >> 2) + 4 * half][l31] = acc[i];
  __syncthreads();
  const int colx = tid & 31, rg = tid >> 5, gn = n0 + colx;
  const bool ep = p.ep_stats != nullptr;
  float esc = 0.f, esh = 0.f, emu = 0.f, ers = 0.f;
  if (ep && gn < cin) { esc = p.ep_scale[gn]; esh = p.ep_shift[gn]; emu = p.ep_mean[gn]; ers = p.ep_rstd[gn]; }
  float s1 = 0.f, s2 = 0.f;
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    const int r = rg + 8 * i;
    if (r < T && gn < cin) {
      float v = (part[0][r][colx] + part[1][r][colx]) + (part[2][r][colx] + part[3][r][colx]);
      const long g = ((long)b * T + r) * cin + gn;
      if (ep) {
        // first half of the BatchNorm+ELU backward of the layer below: dz = da * ELU'(z), its two column sums
        const float yb = p.ep_y[g];
        v *= elu_grad_from_pre(fmaf(yb, esc, esh));
        s1 += v;
        s2 += v * ((yb - emu) * ers);
      }
      out[g] = v;
    }
  }
  DTC_MARK(10);     // combine + epilogue + store
  if (ep) {
    red[(0 * 8 + rg) * 32 + colx] = s1;
    red[(1 * 8 + rg) * 32 + colx] = s2;
    __syncthreads();
    if (tid < 64) {
      const int stat = tid >> 5, c = tid & 31;
      if (n0 + c < cin) {
        double v = 0.0;
#pragma unroll
        for (int g = 0; g < 8; ++g) v += (double)red[(stat * 8 + g) * 32 + c];
        unsafeAtomicAdd(&p.ep_stats[((long)(b % p.nrep) * 2 + stat) * cin + n0 + c], v);
      }
    }
  }
  DTC_MARK(11);     // statistics
  bn_tail_run(p.tail, tid, 256, gridDim.x * gridDim.y * gridDim.z, reinterpret_cast<int*>(red + 2 * 8 * 32));
  DTC_MARK(12);     // tail
}

// the bf16 throughput mode's adjoint (round 4): 128 input channels per workgroup, one 32-column block per wave,
// v_mfma_f32_32x32x16_bf16 on fragments rounded as they are built (see dtc_fwd_bf16_kernel)
__global__ __launch_bounds__(256) void dtc_dgrad_bf16_kernel(DtcDgradParams p, int tile_floats) {
  extern __shared__ __attribute__((aligned(16))) float dg_smem[];
  float* a_lds = dg_smem;                                            // [(ROWS+1)][cr+4]
  bf16_t* Ws = reinterpret_cast<bf16_t*>(dg_smem + tile_floats);     // [NCT][WPH]
  int* flag = reinterpret_cast<int*>(Ws + NCT * WPH);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int b = blockIdx.x, n0 = blockIdx.y * NCT;                 // n0: first input channel (output column)
  const int T = p.T, cin = p.cin, cout = p.cout, d = p.dil;
  const long wrow = (long)cin * 3;
  float* out = p.out + (long)blockIdx.z * p.slab_stride;
  const int per_z = ((cout + CC - 1) / CC + (int)gridDim.z - 1) / (int)gridDim.z * CC;
  const int cz0 = blockIdx.z * per_z;
  const int cr = max(0, min(cout, cz0 + per_z) - cz0);
  const int AP = cr + 4;
  {
    const int q4 = cr >> 2;
    const bool form = p.dy == nullptr;
    const bool keep = p.dy_out != nullptr && blockIdx.y == 0;
    for (int q = tid; q < T * q4; q += 256) {
      const int r = q / q4, c4 = (q - r * q4) << 2;
      const long g = ((long)b * T + r) * cout + cz0 + c4;
      f32x4 v;
      if (form) {
        const f32x4 k0 = load4(p.coef + cz0 + c4), k1 = load4(p.coef + cout + cz0 + c4),
                    k2 = load4(p.coef + 2 * cout + cz0 + c4);
        v = k0 * load4(p.dz + g) + k1 * load4(p.y + g) + k2;
      } else {
        v = load4(p.dy + g);
      }
      *reinterpret_cast<f32x4*>(&a_lds[r * AP + c4]) = v;
      if (keep) store4(p.dy_out + g, v);
    }
    for (int q = tid; q < (ROWS + 1 - T) * AP; q += 256) a_lds[T * AP + q] = 0.f;
  }
  // weight chunk: 32 contraction channels co x (128 input channels x 3 taps = 384 contiguous floats)
  // W[cz0 + c0 + co_l][(n0 ...) * 3 ...]: 12 float4 per thread, scattered transposed: Ws[ci_l][tap * 32 + co_l]
  f32x4 rw[12];
  auto load_w = [&](int c0) {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int q = tid + j * 256;
      const int co_l = q / 96, f = (q - co_l * 96) << 2;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (c0 + co_l < cr && (long)n0 * 3 + f < wrow) v = load4(p.W + (long)(cz0 + c0 + co_l) * wrow + (long)n0 * 3 + f);
      rw[j] = v;
    }
  };
  auto store_w = [&]() {
#pragma unroll
    for (int j = 0; j < 12; ++j) {
      const int q = tid + j * 256;
      const int co_l = q / 96, f = (q - co_l * 96) << 2;
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int kk = f + m, ci_l = kk / 3, tap = kk - ci_l * 3;
        Ws[ci_l * WPH + tap * CC + co_l] = (bf16_t)rw[j][m];
      }
    }
  };
  if (cr > 0) load_w(0);
  __syncthreads();
  const bool active = n0 + wave * 32 < cin;
  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;
  for (int c0 = 0; c0 < cr; c0 += CC) {
    if (c0 > 0) __syncthreads();
    store_w();
    __syncthreads();
    if (c0 + CC < cr) load_w(c0 + CC);
    if (active) {
#pragma unroll
      for (int st = 0; st < 6; ++st) {
        const int tap = st >> 1, co_l = ((st & 1) << 4) + (half << 3);
        const int rs = l31 + (2 - tap) * d;
        const float* ap = (rs < T && c0 + co_l < cr) ? &a_lds[rs * AP + c0 + co_l] : &a_lds[ZROW * AP];
        const bf16x8 av = cvt8(ap);
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&Ws[(wave * 32 + l31) * WPH + tap * CC + co_l]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wv, acc, 0, 0, 0);
      }
    }
  }
  const int gn = n0 + wave * 32 + l31;
  const bool ep = p.ep_stats != nullptr;
  float esc = 0.f, esh = 0.f, emu = 0.f, ers = 0.f;
  if (ep && gn < cin) { esc = p.ep_scale[gn]; esh = p.ep_shift[gn]; emu = p.ep_mean[gn]; ers = p.ep_rstd[gn]; }
  float s1 = 0.f, s2 = 0.f;
  if (active && gn < cin) {
#pragma unroll
    for (int i = 0; i < 16; ++i) {
      const int r = (i & 3) + 8 * (i >> 2) + 4 * half;
      if (r < T) {
        float v = acc[i];
        const long g = ((long)b * T + r) * cin + gn;
        if (ep) {
          const float yb = p.ep_y[g];
          v *= elu_grad_from_pre(fmaf(yb, esc, esh));
          s1 += v;
          s2 += v * ((yb - emu) * ers);
        }
        out[g] = v;
      }
    }
  }
  if (ep) {
    s1 += __shfl_xor(s1, 32, 64);
    s2 += __shfl_xor(s2, 32, 64);
    if (active && half == 0 && gn < cin) {
      unsafeAtomicAdd(&p.ep_stats[((long)(b % p.nrep) * 2 + 0) * cin + gn], (double)s1);
      unsafeAtomicAdd(&p.ep_stats[((long)(b % p.nrep) * 2 + 1) * cin + gn], (double)s2);
    }
  }
  bn_tail_run(p.tail, tid, 256, gridDim.x * gridDim.y * gridDim.z, flag);
}

// ------------------------------------------------------------------ round 4: two sequences per workgroup (layers 5, 6)
// The per-phase trace of the kernels above (tools/dtc_lab.py --trace, profiles/r04_dtc_trace.txt) put 72 % of the
// 256 -> 512 layer's 42 us into the chunk loop -- 3.3 us per 96-deep chunk for 0.35 us of MFMA: every one of the
// 1 024 (sequence, 32-column) workgroups streams its own 96 KB weight slice through a CU that holds three or four of
// them, and stages the same 30 KB sequence tile as its 15 siblings.  Here a workgroup owns TWO sequences (64 rows) and
// NT output channels; a weight chunk is fetched once per pair (half the weight traffic, and with NT = 64 half the
// staging), requested TWO chunks ahead into registers and published through a double-buffered LDS tile (one barrier per
// chunk).  One body serves the forward and its adjoint (ADJ: anti-causal row offsets, dy formed on load, the
// BatchNorm+ELU backward's first half in the epilogue, weights read along their rows) and both matrix pipes
// (BF: v_mfma_f32_32x32x16_bf16 on operands rounded as the fragments are built, the throughput mode).
//   NT = 64: wave = (sequence, 32-column block), whole contraction in the wave, results leave from the registers;
//   NT = 32: wave = (sequence, half of each chunk's k-groups), two partial tiles combined through the LDS.
// Contraction channels beyond PAIR_KC are staged in further passes over the same LDS tile (the adjoint of layer 6).
struct DtcPairParams {
  const float* src;      // forward: [B*T, kc] block input / previous pre-BN output;  adjoint: dy or null
  const float* scale;    // forward: BatchNorm+ELU of the previous layer on load (null: as is)
  const float* shift;
  const float* dz; const float* y; const float* coef;   // adjoint with src == null: tile = coef0*dz + coef1*y + coef2
  float* keep;           // adjoint: the formed dy written out (by the first column tile) or null
  const float* W;        // [cout, cin*3], k = ci*3 + tap
  float* out;            // [B*T, nc]
  float* col;            // forward: im2col of the activated input or null
  double* stats; int nrep;
  const float* ep_y; const float* ep_scale; const float* ep_shift; const float* ep_mean; const float* ep_rstd;
  int B, T, kc, nc, dil; // kc: contraction channels, nc: output channels
  BnTail tail;
};

constexpr int PAIR_KC = 256;

template <bool ADJ, bool BF, int NT>
__global__ __launch_bounds__(256) void dtc_pair_kernel(DtcPairParams p, int tile_floats) {
  extern __shared__ __attribute__((aligned(16))) float pr_smem[];
  constexpr int WROW = BF ? WPH / 2 : WP;              // floats per weight-tile row (bf16 rows: 104 halves = 52 floats)
  constexpr int NJ = NT * 24 / 256;                    // float4 per thread and chunk: NT columns x 96 floats
  float* tile = pr_smem;                               // [2][ROWS + 1][AP]; NT = 32: later part[2][2][ROWS][33]
  float* Wsf = pr_smem + tile_floats;                  // [2][NT][WROW]
  float* red = Wsf + 2 * NT * WROW;                    // [2][8][32]
  int* flag = reinterpret_cast<int*>(red + 2 * 8 * 32);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, l31 = lane & 31, half = lane >> 5;
  const int sq = wave >> 1, wsub = wave & 1;           // sequence of the pair; column block (NT = 64) or k-half (NT = 32)
  const int b0 = 2 * blockIdx.x, n0 = blockIdx.y * NT;
  const int T = p.T, kc = p.kc, nc = p.nc, d = p.dil;
  const int AP = min(kc, PAIR_KC) + 4;
  const long wrow = (long)(ADJ ? nc : kc) * 3;         // floats per row of W
  const int nch = (kc + CC - 1) / CC;
  float* tile_s = tile + sq * (ROWS + 1) * AP;
  DTC_T0();

  f32x4 rw0[NJ], rw1[NJ];
  auto load_w = [&](f32x4 (&rw)[NJ], int c0g) {
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = tid + j * 256;
      f32x4 v = {0.f, 0.f, 0.f, 0.f};
      if (!ADJ) {
        // NT columns x 96 contiguous floats W[n0 + c][c0g * 3 ...]
        const int c = q / 24, f = (q - c * 24) << 2;
        if (n0 + c < nc && c0g * 3 + f < kc * 3) v = load4(p.W + (long)(n0 + c) * wrow + (long)c0g * 3 + f);
      } else {
        // 32 contraction rows x (NT * 3) contiguous floats W[c0g + co_l][n0 * 3 ...]
        constexpr int RUN4 = NT * 3 / 4;
        const int co_l = q / RUN4, f = (q - co_l * RUN4) << 2;
        if (c0g + co_l < kc && (long)n0 * 3 + f < wrow) v = load4(p.W + (long)(c0g + co_l) * wrow + (long)n0 * 3 + f);
      }
      rw[j] = v;
    }
  };
  auto store_w = [&](const f32x4 (&rw)[NJ], int buf) {
    float* wf = Wsf + buf * NT * WROW;
    bf16_t* wh = reinterpret_cast<bf16_t*>(wf);
#pragma unroll
    for (int j = 0; j < NJ; ++j) {
      const int q = tid + j * 256;
      int row0, f;          // forward: row0 = the column, k' from f;  adjoint: row0 = contraction channel, column from f
      if (!ADJ) { row0 = q / 24; f = (q - row0 * 24) << 2; }
      else { constexpr int RUN4 = NT * 3 / 4; row0 = q / RUN4; f = (q - row0 * RUN4) << 2; }
#pragma unroll
      for (int m = 0; m < 4; ++m) {
        const int kk = f + m, a = kk / 3, tap = kk - a * 3;
        const int col = ADJ ? a : row0, kl = ADJ ? row0 : a;      // LDS row = output column, k' = tap * 32 + channel
        if (BF) wh[col * WPH + tap * CC + kl] = (bf16_t)rw[j][m];
        else wf[col * WP + tap * CC + kl] = rw[j][m];
      }
    }
  };
  load_w(rw0, 0);
  if (nch > 1) load_w(rw1, CC);

  f32x16 acc;
#pragma unroll
  for (int i = 0; i < 16; ++i) acc[i] = 0.f;

  auto contract = [&](int buf, int cl, int kr) {
    // one 96-deep chunk: tile-local channels cl .. cl + 31 of the kr staged ones
    const float* wf = Wsf + buf * NT * WROW;
    const int wcol = (NT == 64 ? wsub * 32 : 0) + l31;
    if (!BF) {
      constexpr int NG = NT == 64 ? 12 : 6;
#pragma unroll
      for (int g = 0; g < NG; ++g) {
        const int gg = NT == 64 ? g : wsub * 6 + g;     // 12 groups of 8 per chunk: tap = gg / 4, 8 channels each
        const int tap = gg >> 2, kl = ((gg & 3) << 3) + (half << 2);
        const int rs = ADJ ? l31 + (2 - tap) * d : l31 - (2 - tap) * d;
        const bool ok = (ADJ ? rs < T : rs >= 0) && cl + kl < kr;
        const f32x4 av = *reinterpret_cast<const f32x4*>(ok ? &tile_s[rs * AP + cl + kl] : &tile_s[ZROW * AP]);
        const f32x4 wv = *reinterpret_cast<const f32x4*>(&wf[wcol * WP + tap * CC + kl]);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, wv.x, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, wv.y, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.z, wv.z, acc, 0, 0, 0);
        acc = __builtin_amdgcn_mfma_f32_32x32x2f32(av.w, wv.w, acc, 0, 0, 0);
      }
    } else {
      const bf16_t* wh = reinterpret_cast<const bf16_t*>(wf);
      constexpr int NS = NT == 64 ? 6 : 3;
#pragma unroll
      for (int g = 0; g < NS; ++g) {
        const int st = NT == 64 ? g : wsub * 3 + g;     // k' = 16 st .. 16 st + 15, this lane's eight: + 8 half
        const int tap = st >> 1, kl = ((st & 1) << 4) + (half << 3);
        const int rs = ADJ ? l31 + (2 - tap) * d : l31 - (2 - tap) * d;
        const bool ok = (ADJ ? rs < T : rs >= 0) && cl + kl < kr;
        const bf16x8 av = cvt8(ok ? &tile_s[rs * AP + cl + kl] : &tile_s[ZROW * AP]);
        const bf16x8 wv = *reinterpret_cast<const bf16x8*>(&wh[wcol * WPH + tap * CC + kl]);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av, wv, acc, 0, 0, 0);
      }
    }
  };

  int c = 0;                                             // chunk index over the whole contraction
  for (int k0 = 0; k0 < kc; k0 += PAIR_KC) {
    const int kr = min(kc - k0, PAIR_KC);                // channels staged in this pass
    if (k0 > 0) __syncthreads();                         // the previous pass's readers are done with the tiles
    // ---- the two sequence tiles.  One workgroup per CU: nothing hides a load's latency but the loads themselves, so a
    // thread requests NB quads (adjoint: of dz and of y) before it transforms the first one
    {
      constexpr int NB = ADJ ? 8 : 16;                  // 16 loads in flight per thread either way: the whole pair in one round at 256 channels
      const int q4 = kr >> 2, R2 = 2 * T;
      // quad q = tid + 256 i of the pair's 2 T rows x q4 quads: (row, quad-in-row) advanced by increments -- the two
      // integer divisions per quad were a third of this phase (tools/dtc_lab.py --trace)
      const int dr = 256 / q4, dc = 256 - dr * q4;
      int R = tid / q4, cq = tid - R * q4;
      // the per-channel vectors of this thread's first quad: with q4 | 256 (every layer of the block) they serve all its
      // quads -- loaded per quad they sat, as dependent L2 round trips, in the middle of the transform loop (4 us of 11)
      const int cq0 = cq;
      f32x4 h0 = {1.f, 1.f, 1.f, 1.f}, h1 = {0.f, 0.f, 0.f, 0.f}, h2 = {0.f, 0.f, 0.f, 0.f};
      if (R < R2) {
        if (!ADJ) {
          if (p.scale != nullptr) { h0 = load4(p.scale + k0 + (cq0 << 2)); h1 = load4(p.shift + k0 + (cq0 << 2)); }
        } else if (p.src == nullptr) {
          h0 = load4(p.coef + k0 + (cq0 << 2));
          h1 = load4(p.coef + kc + k0 + (cq0 << 2));
          h2 = load4(p.coef + 2 * kc + k0 + (cq0 << 2));
        }
      }
      while (R < R2) {
        f32x4 v[NB], w[NB];
        int Rj[NB], cj[NB];
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          Rj[j] = R;
          cj[j] = cq;
          const int s = R >= T ? 1 : 0, r = R - s * T, c4 = cq << 2;
          v[j] = f32x4{0.f, 0.f, 0.f, 0.f};
          w[j] = v[j];
          if (R < R2 && b0 + s < p.B) {
            const long g = ((long)(b0 + s) * T + r) * kc + k0 + c4;
            if (ADJ && p.src == nullptr) {
              v[j] = load4(p.dz + g);
              w[j] = load4(p.y + g);
            } else {
              v[j] = load4(p.src + g);
            }
          }
          R += dr;
          cq += dc;
          if (cq >= q4) { cq -= q4; ++R; }
        }
#pragma unroll
        for (int j = 0; j < NB; ++j) {
          const int s = Rj[j] >= T ? 1 : 0, r = Rj[j] - s * T, c4 = cj[j] << 2;
          if (Rj[j] < R2) {
            f32x4 x = v[j];
            if (b0 + s < p.B) {
              const long g = ((long)(b0 + s) * T + r) * kc + k0 + c4;
              if (!ADJ) {
                if (p.scale != nullptr) {
                  f32x4 sc = h0, sh = h1;
                  if (cj[j] != cq0) { sc = load4(p.scale + k0 + c4); sh = load4(p.shift + k0 + c4); }
                  x.x = elu_stage(fmaf(sc.x, x.x, sh.x));
                  x.y = elu_stage(fmaf(sc.y, x.y, sh.y));
                  x.z = elu_stage(fmaf(sc.z, x.z, sh.z));
                  x.w = elu_stage(fmaf(sc.w, x.w, sh.w));
                }
              } else {
                if (p.src == nullptr) {
                  f32x4 c0v = h0, c1v = h1, c2v = h2;
                  if (cj[j] != cq0) { c0v = load4(p.coef + k0 + c4); c1v = load4(p.coef + kc + k0 + c4); c2v = load4(p.coef + 2 * kc + k0 + c4); }
                  x = c0v * x + c1v * w[j] + c2v;
                }
                if (p.keep != nullptr && blockIdx.y == 0) store4(p.keep + g, x);
              }
            }
            *reinterpret_cast<f32x4*>(&tile[s * (ROWS + 1) * AP + r * AP + c4]) = x;      // B odd: the missing sequence is zeros
          }
        }
      }
      for (int q = tid; q < 2 * (ROWS + 1 - T) * AP; q += 256) {                           // rows T..31 and the zero row, both tiles
        const int s = q >= (ROWS + 1 - T) * AP ? 1 : 0;
        tile[s * (ROWS + 1) * AP + T * AP + (q - s * (ROWS + 1 - T) * AP)] = 0.f;
      }
    }
    __syncthreads();
    DTC_MARK(0);      // staging
    if (!ADJ && p.col != nullptr) {
      for (int s = 0; s < 2; ++s)
        if (b0 + s < p.B)
          write_col(p.col + (long)(b0 + s) * T * (kc * 3) + (long)k0 * 3, kc * 3, tile + s * (ROWS + 1) * AP, AP, T, kr, d, tid);
    }
    DTC_MARK(1);      // im2col
    for (int cl = 0; cl < kr; cl += 2 * CC) {
      store_w(rw0, 0);
      __syncthreads();
      if (c + 2 < nch) load_w(rw0, (c + 2) * CC);
      contract(0, cl, kr);
      ++c;
      if (cl + CC < kr) {
        store_w(rw1, 1);
        __syncthreads();
        if (c + 2 < nch) load_w(rw1, (c + 2) * CC);
        contract(1, cl + CC, kr);
        ++c;
      }
    }
    DTC_MARK(2);      // contraction
  }

  // ---- results.  accumulator i of lane (l31, half) is row (i & 3) + 8 (i >> 2) + 4 half, column l31 of the wave's block
  float s1 = 0.f, s2 = 0.f;
  float esc = 0.f, esh = 0.f, emu = 0.f, ers = 0.f;
  const bool ep = ADJ && p.stats != nullptr;
  auto emit = [&](float v, int b, int r, int gn) {
    const long g = ((long)b * T + r) * nc + gn;
    if (!ADJ) {
      p.out[g] = v;
      s1 += v;
      s2 += v * v;
    } else {
      if (ep) {
        // first half of the BatchNorm+ELU backward of the layer below: dz = da * ELU'(z), its two column sums
        const float yb = p.ep_y[g];
        v *= elu_grad_from_pre(fmaf(yb, esc, esh));
        s1 += v;
        s2 += v * ((yb - emu) * ers);
      }
      p.out[g] = v;
    }
  };
  const bool want_stats = p.stats != nullptr;
  if (NT == 64) {
    const int gn = n0 + wsub * 32 + l31, b = b0 + sq;
    if (ep && gn < nc) { esc = p.ep_scale[gn]; esh = p.ep_shift[gn]; emu = p.ep_mean[gn]; ers = p.ep_rstd[gn]; }
    if (gn < nc && b < p.B) {
#pragma unroll
      for (int i = 0; i < 16; ++i) {
        const int r = (i & 3) + 8 * (i >> 2) + 4 * half;
        if (r < T) emit(acc[i], b, r, gn);
      }
    }
    DTC_MARK(3);      // store
    if (want_stats) {
      s1 += __shfl_xor(s1, 32, 64);
      s2 += __shfl_xor(s2, 32, 64);
      if (half == 0) {
        red[(0 * 8 + wave) * 32 + l31] = s1;
        red[(1 * 8 + wave) * 32 + l31] = s2;
      }
      __syncthreads();
      if (tid < 128) {
        const int stat = tid >> 6, cc = tid & 63, cb = cc >> 5;
        if (n0 + cc < nc) {
          const double v = (double)red[(stat * 8 + cb) * 32 + (cc & 31)] + (double)red[(stat * 8 + 2 + cb) * 32 + (cc & 31)];
          unsafeAtomicAdd(&p.stats[((long)(blockIdx.x % p.nrep) * 2 + stat) * nc + n0 + cc], v);
        }
      }
    }
  } else {
    __syncthreads();                                     // the sequence tiles are dead
    float* part = tile;                                  // [2][2][ROWS][33]
#pragma unroll
    for (int i = 0; i < 16; ++i) part[((sq * 2 + wsub) * ROWS + (i & 3) + 8 * (i >> 2) + 4 * half) * 33 + l31] = acc[i];
    __syncthreads();
    const int colx = tid & 31, rg = tid >> 5, gn = n0 + colx;
    if (ep && gn < nc) { esc = p.ep_scale[gn]; esh = p.ep_shift[gn]; emu = p.ep_mean[gn]; ers = p.ep_rstd[gn]; }
    if (gn < nc) {
#pragma unroll
      for (int s = 0; s < 2; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) {
          const int r = rg + 8 * i;
          if (r < T && b0 + s < p.B)
            emit(part[((s * 2 + 0) * ROWS + r) * 33 + colx] + part[((s * 2 + 1) * ROWS + r) * 33 + colx], b0 + s, r, gn);
        }
    }
    DTC_MARK(3);      // combine + store
    if (want_stats) {
      red[(0 * 8 + rg) * 32 + colx] = s1;
      red[(1 * 8 + rg) * 32 + colx] = s2;
      __syncthreads();
      if (tid < 64) {
        const int stat = tid >> 5, cc = tid & 31;
        if (n0 + cc < nc) {
          double v = 0.0;
#pragma unroll
          for (int g = 0; g < 8; ++g) v += (double)red[(stat * 8 + g) * 32 + cc];
          unsafeAtomicAdd(&p.stats[((long)(blockIdx.x % p.nrep) * 2 + stat) * nc + n0 + cc], v);
        }
      }
    }
  }
  DTC_MARK(4);        // statistics
  bn_tail_run(p.tail, tid, 256, gridDim.x * gridDim.y, flag);
  DTC_MARK(5);
}

static inline int pair_tile_floats(int kc) {
  const int a = 2 * (ROWS + 1) * ((kc < PAIR_KC ? kc : PAIR_KC) + 4), part = 2 * 2 * ROWS * 33;
  return a > part ? a : part;
}

// the pair kernels take the wide layers: at least four chunks of contraction, whole chunks, at least 64 output channels
static inline bool pair_takes(int kc, int nc, int ksplit, bool adj) {
  // PCAA_DTC_PAIR: 0 = off, fwd / adj = that direction only (lab), anything else = both
  static const int which = [] {
    const char* e = getenv("PCAA_DTC_PAIR");
    if (e == nullptr) return 3;
    if (e[0] == '0') return 0;
    if (e[0] == 'f') return 1;
    if (e[0] == 'a') return 2;
    return 3;
  }();
  return (which & (adj ? 2 : 1)) != 0 && ksplit == 1 && kc >= 128 && kc % CC == 0 && nc >= 64 && nc % 4 == 0;
}

template <bool ADJ, bool BF, int NT>
static int launch_pair(const DtcPairParams& p, hipStream_t s) {
  constexpr int WROW = BF ? WPH / 2 : WP;
  auto kern = dtc_pair_kernel<ADJ, BF, NT>;
  const int tile = pair_tile_floats(p.kc);
  const size_t lds = (size_t)(tile + 2 * NT * WROW + 2 * 8 * 32 + 4) * sizeof(float);
  static bool configured = false;
  if (!configured) {
    const size_t cap = (size_t)(pair_tile_floats(PAIR_KC) + 2 * NT * WROW + 2 * 8 * 32 + 4) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)cap) != hipSuccess)
      return 1;
    configured = true;
  }
  hipLaunchKernelGGL(kern, dim3((p.B + 1) / 2, (p.nc + NT - 1) / NT), dim3(256), lds, s, p, tile);
  return 0;
}

template <bool ADJ>
static int launch_pair_any(const DtcPairParams& p, bool bf16, hipStream_t s) {
  // 64-column workgroups when they still fill the chip (the 256 -> 512 layer's forward: 32 pairs x 8), else 32
  const bool wide = (long)((p.B + 1) / 2) * (p.nc / 64) >= 192 && p.nc % 64 == 0;
  if (wide) return bf16 ? launch_pair<ADJ, true, 64>(p, s) : launch_pair<ADJ, false, 64>(p, s);
  return bf16 ? launch_pair<ADJ, true, 32>(p, s) : launch_pair<ADJ, false, 32>(p, s);
}

}  // namespace

extern "C" int pcaa_dtc_conv_supported(int T, int cin, int cout) {
  return (T >= 1 && T <= ROWS && cin >= 4 && cin % 4 == 0 && cout >= 16 && cout % 16 == 0) ? 1 : 0;
}

/* smallest / recommended split of the input channels over workgroups: a workgroup keeps at most MAX_CR
 * channels of its sequence in LDS; few workgroups with a long contraction are cut further */
extern "C" int pcaa_dtc_conv_ksplit(int B, int cin, int cout) {
  const int chunks = (cin + CC - 1) / CC;
  int ks = (cin + MAX_CR - 1) / MAX_CR;
  if ((long)B * ((cout + 31) / 32) <= 128 && chunks >= 16) ks = ks > 8 ? ks : 8;
  return ks < chunks ? ks : chunks;
}

static int dtc_conv_fwd_impl(bool bf16, const float* src, const float* scale, const float* shift, const float* W, float* y,
                             float* col, double* stats, int nrep, int B, int T, int cin, int cout, int dilation,
                             int ksplit, long slab_stride, void* stream) {
  PCAA_CHECK_ARG(src && W && y && B >= 1 && dilation >= 1 && ksplit >= 1, "pcaa_dtc_conv_fwd: bad args");
  PCAA_CHECK_ARG(pcaa_dtc_conv_supported(T, cin, cout), "pcaa_dtc_conv_fwd: needs T <= %d, cin %% 4 == 0, cout %% 16 == 0",
                 ROWS);
  PCAA_CHECK_ARG((scale == nullptr) == (shift == nullptr), "pcaa_dtc_conv_fwd: scale and shift go together");
  PCAA_CHECK_ARG(stats == nullptr || nrep >= 1, "pcaa_dtc_conv_fwd: nrep");
  PCAA_CHECK_ARG(((uintptr_t)W % 16) == 0 && ((uintptr_t)src % 16) == 0 && (scale == nullptr || (((uintptr_t)scale % 16) == 0 &&
                 ((uintptr_t)shift % 16) == 0)), "pcaa_dtc_conv_fwd: src, W, scale, shift must be 16-B aligned");
  const int chunks = (cin + CC - 1) / CC;
  const int per_z = (chunks + ksplit - 1) / ksplit * CC;
  PCAA_CHECK_ARG(ksplit <= chunks && per_z <= MAX_CR, "pcaa_dtc_conv_fwd: ksplit must keep <= %d channels per workgroup "
                 "(pcaa_dtc_conv_ksplit)", MAX_CR);
  PCAA_CHECK_ARG(ksplit == 1 || (stats == nullptr && slab_stride >= (long)B * T * cout),
                 "pcaa_dtc_conv_fwd: ksplit > 1 writes slabs (no statistics): slab_stride >= B*T*cout");
  if (pair_takes(cin, cout, ksplit, false)) {
    DtcPairParams pp{src, scale, shift, nullptr, nullptr, nullptr, nullptr, W, y, col, stats, nrep,
                     nullptr, nullptr, nullptr, nullptr, nullptr, B, T, cin, cout, dilation,
                     stats != nullptr ? pcaa_take_bn_tail(stats) : BnTail{}};
    if (launch_pair_any<false>(pp, bf16, as_stream(stream)) != 0) {
      pcaa_rearm_bn_tail(pp.tail);
      pcaa_set_error("pcaa_dtc_conv_fwd: cannot raise the dynamic LDS limit");
      return PCAA_ERR_LAUNCH;
    }
    PCAA_RETURN_LAUNCH_STATUS("pcaa_dtc_conv_fwd");
  }
  DtcFwdParams p{src, scale, shift, W, y, col, stats, B, T, cin, cout, dilation, nrep, ksplit > 1 ? slab_stride : 0,
                 (stats != nullptr && ksplit == 1) ? pcaa_take_bn_tail(stats) : BnTail{}};
  if (bf16) hipLaunchKernelGGL(dtc_fwd_bf16_kernel, dim3(B, (cout + NCT - 1) / NCT, ksplit), dim3(256), 0, as_stream(stream), p);
  else hipLaunchKernelGGL(dtc_fwd_kernel, dim3(B, (cout + 31) / 32, ksplit), dim3(256), 0, as_stream(stream), p);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_dtc_conv_fwd");
}
extern "C" int pcaa_dtc_conv_fwd(const float* src, const float* scale, const float* shift, const float* W, float* y,
                                 float* col, double* stats, int nrep, int B, int T, int cin, int cout, int dilation,
                                 int ksplit, long slab_stride, void* stream) {
  return dtc_conv_fwd_impl(false, src, scale, shift, W, y, col, stats, nrep, B, T, cin, cout, dilation, ksplit, slab_stride, stream);
}
extern "C" int pcaa_dtc_conv_fwd_bf16(const float* src, const float* scale, const float* shift, const float* W, float* y,
                                      float* col, double* stats, int nrep, int B, int T, int cin, int cout, int dilation,
                                      int ksplit, long slab_stride, void* stream) {
  return dtc_conv_fwd_impl(true, src, scale, shift, W, y, col, stats, nrep, B, T, cin, cout, dilation, ksplit, slab_stride, stream);
}

/* channel split for the dgrad (its contraction runs over the OUTPUT channels of the convolution; a workgroup
 * keeps up to 512 of them in LDS) */
extern "C" int pcaa_dtc_conv_dgrad_ksplit(int B, int cin, int cout) {
  (void)B; (void)cin;
  const int ks = (cout + DG_MAX_CR - 1) / DG_MAX_CR;
  return ks < 1 ? 1 : ks;
}

static int dtc_conv_dgrad_impl(bool bf16, const float* dy, const float* dz, const float* y, const float* coef, float* dy_out,
                               const float* W, float* out, const float* ep_y, const float* ep_scale,
                               const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                               double* ep_stats, int nrep, int B, int T, int cin, int cout, int dilation,
                               int ksplit, long slab_stride, void* stream) {
  PCAA_CHECK_ARG(W && out && B >= 1 && dilation >= 1 && ksplit >= 1, "pcaa_dtc_conv_dgrad: bad args");
  PCAA_CHECK_ARG((dy != nullptr) != (dz != nullptr && y != nullptr && coef != nullptr),
                 "pcaa_dtc_conv_dgrad: either dy, or dz + y + coef");
  PCAA_CHECK_ARG(T >= 1 && T <= ROWS && cin >= 4 && cin % 4 == 0 && cout >= 4 && cout % 4 == 0,
                 "pcaa_dtc_conv_dgrad: needs T <= %d, cin %% 4 == 0, cout %% 4 == 0", ROWS);
  PCAA_CHECK_ARG(((uintptr_t)W % 16) == 0 && (!dy || ((uintptr_t)dy % 16) == 0) && (!dz || (((uintptr_t)dz % 16) == 0 &&
                 ((uintptr_t)y % 16) == 0 && ((uintptr_t)coef % 16) == 0)) && (!dy_out || ((uintptr_t)dy_out % 16) == 0),
                 "pcaa_dtc_conv_dgrad: 16-B alignment");
  const int chunks = (cout + CC - 1) / CC;
  const int per_z = (chunks + ksplit - 1) / ksplit * CC;
  PCAA_CHECK_ARG(ksplit <= chunks && per_z <= DG_MAX_CR, "pcaa_dtc_conv_dgrad: ksplit must keep <= %d channels per "
                 "workgroup (pcaa_dtc_conv_dgrad_ksplit)", DG_MAX_CR);
  PCAA_CHECK_ARG(ksplit == 1 || slab_stride >= (long)B * T * cin, "pcaa_dtc_conv_dgrad: slab_stride >= B*T*cin");
  const bool ep = ep_stats != nullptr;
  PCAA_CHECK_ARG(!ep || (ksplit == 1 && ep_y && ep_scale && ep_shift && ep_mean && ep_rstd && nrep >= 1),
                 "pcaa_dtc_conv_dgrad: the epilogue needs ksplit == 1 and ep_y, ep_scale, ep_shift, ep_mean, ep_rstd");
  if (pair_takes(cout, cin, ksplit, true)) {
    DtcPairParams pp{dy, nullptr, nullptr, dz, y, coef, dy_out, W, out, nullptr, ep_stats, nrep,
                     ep_y, ep_scale, ep_shift, ep_mean, ep_rstd, B, T, cout, cin, dilation,
                     ep ? pcaa_take_bn_tail(ep_stats) : BnTail{}};
    if (launch_pair_any<true>(pp, bf16, as_stream(stream)) != 0) {
      pcaa_rearm_bn_tail(pp.tail);
      pcaa_set_error("pcaa_dtc_conv_dgrad: cannot raise the dynamic LDS limit");
      return PCAA_ERR_LAUNCH;
    }
    PCAA_RETURN_LAUNCH_STATUS("pcaa_dtc_conv_dgrad");
  }
  const int tile = dg_tile_floats(per_z);
  DtcDgradParams p{dy, dz, y, coef, dy_out, W, out, ep_y, ep_scale, ep_shift, ep_mean, ep_rstd, ep_stats, nrep,
                   B, T, cin, cout, dilation, ksplit > 1 ? slab_stride : 0,
                   ep ? pcaa_take_bn_tail(ep_stats) : BnTail{}};
  if (bf16) {
    const size_t lds16 = (size_t)tile * sizeof(float) + (size_t)NCT * WPH * sizeof(bf16_t) + 16;
    static bool configured16 = false;
    if (!configured16) {
      const size_t cap = (size_t)dg_tile_floats(DG_MAX_CR) * sizeof(float) + (size_t)NCT * WPH * sizeof(bf16_t) + 16;
      if (hipFuncSetAttribute(reinterpret_cast<const void*>(dtc_dgrad_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                              (int)cap) != hipSuccess) {
        pcaa_rearm_bn_tail(p.tail);
        pcaa_set_error("pcaa_dtc_conv_dgrad_bf16: cannot raise the dynamic LDS limit");
        return PCAA_ERR_LAUNCH;
      }
      configured16 = true;
    }
    hipLaunchKernelGGL(dtc_dgrad_bf16_kernel, dim3(B, (cin + NCT - 1) / NCT, ksplit), dim3(256), lds16, as_stream(stream), p, tile);
    PCAA_RETURN_LAUNCH_STATUS("pcaa_dtc_conv_dgrad_bf16");
  }
  const size_t lds = (size_t)(tile + 32 * WP + 2 * 8 * 32 + 4) * sizeof(float);     // + the finalize flag
  static bool configured = false;
  if (!configured) {
    const size_t cap = (size_t)(dg_tile_floats(DG_MAX_CR) + 32 * WP + 2 * 8 * 32 + 4) * sizeof(float);
    if (hipFuncSetAttribute(reinterpret_cast<const void*>(dtc_dgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                            (int)cap) != hipSuccess) {
      pcaa_rearm_bn_tail(p.tail);
      pcaa_set_error("pcaa_dtc_conv_dgrad: cannot raise the dynamic LDS limit");
      return PCAA_ERR_LAUNCH;
    }
    configured = true;
  }
  hipLaunchKernelGGL(dtc_dgrad_kernel, dim3(B, (cin + 31) / 32, ksplit), dim3(256), lds, as_stream(stream), p, tile);
  PCAA_RETURN_LAUNCH_STATUS("pcaa_dtc_conv_dgrad");
}
extern "C" int pcaa_dtc_conv_dgrad(const float* dy, const float* dz, const float* y, const float* coef, float* dy_out,
                                   const float* W, float* out, const float* ep_y, const float* ep_scale,
                                   const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                                   double* ep_stats, int nrep, int B, int T, int cin, int cout, int dilation,
                                   int ksplit, long slab_stride, void* stream) {
  return dtc_conv_dgrad_impl(false, dy, dz, y, coef, dy_out, W, out, ep_y, ep_scale, ep_shift, ep_mean, ep_rstd, ep_stats, nrep,
                             B, T, cin, cout, dilation, ksplit, slab_stride, stream);
}
extern "C" int pcaa_dtc_conv_dgrad_bf16(const float* dy, const float* dz, const float* y, const float* coef, float* dy_out,
                                        const float* W, float* out, const float* ep_y, const float* ep_scale,
                                        const float* ep_shift, const float* ep_mean, const float* ep_rstd,
                                        double* ep_stats, int nrep, int B, int T, int cin, int cout, int dilation,
                                        int ksplit, long slab_stride, void* stream) {
  return dtc_conv_dgrad_impl(true, dy, dz, y, coef, dy_out, W, out, ep_y, ep_scale, ep_shift, ep_mean, ep_rstd, ep_stats, nrep,
                             B, T, cin, cout, dilation, ksplit, slab_stride, stream);
}
