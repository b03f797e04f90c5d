// A BatchNorm "finalize" carried by the launch that PRODUCES its statistics.
//
// Train-mode BatchNorm is a grid-wide dependency: every workgroup adds its column sums into the fp64 replica
// buffer, and only when all of them have can the per-channel coefficients be formed.  Rounds 1-2 did that in a
// launch of its own (bn_finalize_kernel / bn_bwd_finalize_kernel: 20 launches of 5-8 us per step, each on the critical
// path between two big kernels).  Here the LAST workgroup to arrive does it:
//
//   every wave: s_waitcnt vmcnt(0)            -- its statistics atomics have been performed (they execute at the
//                                                memory side, MI355X_MICROARCH.md "Global float atomics")
//   workgroup barrier; lane 0: relaxed agent-scope fetch_add on the arrival counter; the workgroup that draws
//   nblocks-1 is the finalizer: every replica word is then read with an agent-scope (sc1) load -- neither this CU's
//   L1 nor this XCD's L2 can serve a stale copy (cdna_hip_programming.md Guideline 16, the "every payload byte
//   written write-through / by atomics and drained, every load sc1" form: no release, no cache invalidate; nobody
//   ever waits for another workgroup, so no placement or dispatch-order assumption).  The counter is zeroed with the statistics (ops.StatsPool clears its whole used range once per step)
//   and reset by the finalizer, so a replayed hipGraph starts from zero as well.
//
// The arithmetic is the one of the stand-alone kernels (elementwise.hip), which remain for SyncBN (the all-reduce of
// the statistics sits between producer and finalize) and for producers that do not carry a tail.
#pragma once
#include "common.h"

struct BnTail {
  int kind;                 // 0: none, 1: forward coefficients, 2: backward coefficients
  int nrep, ch;
  unsigned* counter;
  const double* stats;      // [nrep][2][ch]
  double inv_count, unbias;
  const float* lin_bias;    // forward (may be null)
  const float* gamma;
  const float* beta;        // forward
  float* running_mean;      // forward (may be null together with running_var / nbt)
  float* running_var;
  long long* nbt;
  float momentum, eps;
  float* scale;             // forward outputs
  float* shift;
  float* mean;              // forward: output; backward: input
  float* rstd;
  float* coef;              // backward outputs: [3][ch]
  float* dgamma;            // (may be null)
  float* dbeta;
};

// armed by pcaa_bn_tail_arm_fwd / _bwd, taken (and disarmed) by the next launcher that can carry it
BnTail pcaa_take_bn_tail(const double* stats);
void pcaa_rearm_bn_tail(const BnTail& t);      // give a taken tail back when the launch did not happen

__device__ __forceinline__ double bn_tail_ld(const double* p) {
  return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// the finalize of ONE channel from its two reduced sums (== bn_finalize_kernel / bn_bwd_finalize_kernel, elementwise.hip)
__device__ __forceinline__ void bn_tail_channel(const BnTail& t, int c, double s1, double s2) {
  const int ch = t.ch;
  if (t.kind == 1) {
    const double m0 = s1 * t.inv_count;
    double var = s2 * t.inv_count - m0 * m0;
    if (var < 0.0) var = 0.0;
    const double mean = m0 + (t.lin_bias ? (double)t.lin_bias[c] : 0.0);
    const double rstd = 1.0 / sqrt(var + (double)t.eps);
    t.scale[c] = (float)((double)t.gamma[c] * rstd);
    t.shift[c] = (float)((double)t.beta[c] - m0 * (double)t.gamma[c] * rstd);
    t.mean[c] = (float)m0;
    t.rstd[c] = (float)rstd;
    if (t.running_mean != nullptr) {
      t.running_mean[c] = (1.f - t.momentum) * t.running_mean[c] + t.momentum * (float)mean;
      t.running_var[c] = (1.f - t.momentum) * t.running_var[c] + t.momentum * (float)(var * t.unbias);
    }
  } else {
    if (t.dbeta) t.dbeta[c] = (float)s1;
    if (t.dgamma) t.dgamma[c] = (float)s2;
    const double c1 = s1 * t.inv_count, c2 = s2 * t.inv_count;
    const double rs = t.rstd[c], mu = t.mean[c];
    const double g = (double)t.gamma[c] * rs;
    t.coef[0 * ch + c] = (float)g;
    t.coef[1 * ch + c] = (float)(-g * c2 * rs);
    t.coef[2 * ch + c] = (float)(-g * c1 + g * c2 * rs * mu);
  }
}

// Call from EVERY thread of EVERY workgroup of the launch, after the thread's last statistics atomic has been issued
// (uniform control flow).  ``flag`` is one int of LDS that nothing else uses until the call returns.
__device__ __forceinline__ void bn_tail_run(const BnTail& t, int tid, int nthreads, unsigned nblocks, int* flag) {
  if (t.kind == 0) return;
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (tid == 0) {
    // No release fence: the only bytes handed over are the statistics, and those were written by atomics, which
    // execute at the memory side and leave nothing in this XCD's L2 (an agent-scope release here is a write-back of
    // the WHOLE L2 -- measured: with the decoder's Adam streaming on a side stream it cost the temporal block's
    // backward +190 us per step).
    const unsigned prev = __hip_atomic_fetch_add(t.counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    *flag = prev == nblocks - 1 ? 1 : 0;
  }
  __syncthreads();
  if (*flag == 0) return;
  // Every load of a handed-over word below is an agent-scope (sc1) load, which neither this CU's L1 nor this XCD's
  // L2 serves from a stale copy: no cache invalidate is needed (Guideline 16, "every load sc1"); the fence only keeps
  // the compiler from moving them above the counter.
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  if (tid == 0) {
    __hip_atomic_store(t.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (t.kind == 1 && t.nbt != nullptr) *t.nbt += 1;
  }
  const int ch = t.ch;
  for (int c = tid; c < ch; c += nthreads) {
    double s1 = 0.0, s2 = 0.0;
    int r = 0;
    for (; r + 8 <= t.nrep; r += 8) {          // 16 loads in flight (same summation order as replica_sums)
      double a[8], b[8];
#pragma unroll
      for (int u = 0; u < 8; ++u) {
        a[u] = bn_tail_ld(t.stats + ((long)(r + u) * 2 + 0) * ch + c);
        b[u] = bn_tail_ld(t.stats + ((long)(r + u) * 2 + 1) * ch + c);
      }
#pragma unroll
      for (int u = 0; u < 8; ++u) { s1 += a[u]; s2 += b[u]; }
    }
    for (; r < t.nrep; ++r) {
      s1 += bn_tail_ld(t.stats + ((long)r * 2 + 0) * ch + c);
      s2 += bn_tail_ld(t.stats + ((long)r * 2 + 1) * ch + c);
    }
    bn_tail_channel(t, c, s1, s2);
  }
}
