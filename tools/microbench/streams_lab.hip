// A HIP stream confined to part of the chip.  One step has kernels that need the matrix pipes (the weight-gradient
// products, which only feed the optimizer) and kernels that only need HBM (the BatchNorm-backward elementwise passes
// on the backward's critical chain): side by side on disjoint compute units they overlap, on the same compute units a
// resident 512-thread GEMM workgroup owns the register file and the other kernel simply waits.  The mask is the
// queue's CU mask (hipExtStreamCreateWithCUMask); bit i is compute unit i / 8 of XCD i % 8, so the low n bits (n a
// multiple of 8) give every XCD the same share and the GEMMs' XCD-aware block orders keep their meaning; bits
// [first_cu, first_cu + n_cus) in general, so that two streams can own complementary parts.
// LAB ONLY (round 4: moved out of the product library / include/pcaa_hip.h -- no product caller; tools/overlap_lab.py
// builds this file into tools/microbench/libstreams_lab.so and binds it itself).
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#define PCAA_OK 0
#define PCAA_ERR_INVALID_ARG 1
#define PCAA_ERR_LAUNCH 2
static void pcaa_set_error(const char* fmt, ...) { (void)fmt; }
#define PCAA_CHECK_ARG(cond, ...) do { if (!(cond)) { fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); return PCAA_ERR_INVALID_ARG; } } while (0)

extern "C" int lab_stream_create_masked(int first_cu, int n_cus, void** stream) {
  PCAA_CHECK_ARG(stream != nullptr, "lab_stream_create_masked: null");
  int dev = 0, total = 0;
  if (hipGetDevice(&dev) != hipSuccess ||
      hipDeviceGetAttribute(&total, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess) {
    pcaa_set_error("lab_stream_create_masked: no device");
    return PCAA_ERR_LAUNCH;
  }
  PCAA_CHECK_ARG(n_cus >= 8 && first_cu >= 0 && first_cu + n_cus <= total && (n_cus % 8) == 0 && (first_cu % 8) == 0,
                 "lab_stream_create_masked: first_cu and n_cus must be multiples of 8 within [0, %d] (got %d + %d)", total,
                 first_cu, n_cus);
  uint32_t mask[32] = {0};
  PCAA_CHECK_ARG(total <= 32 * 32, "lab_stream_create_masked: %d compute units", total);
  for (int i = first_cu; i < first_cu + n_cus; ++i) mask[i >> 5] |= 1u << (i & 31);
  hipStream_t s = nullptr;
  const hipError_t e = hipExtStreamCreateWithCUMask(&s, (uint32_t)((total + 31) / 32), mask);
  if (e != hipSuccess) {
    pcaa_set_error("lab_stream_create_masked: hipExtStreamCreateWithCUMask: %s", hipGetErrorString(e));
    return PCAA_ERR_LAUNCH;
  }
  *stream = s;
  return PCAA_OK;
}

extern "C" int lab_stream_destroy(void* stream) {
  if (stream != nullptr && hipStreamDestroy(reinterpret_cast<hipStream_t>(stream)) != hipSuccess) {
    pcaa_set_error("lab_stream_destroy: hipStreamDestroy failed");
    return PCAA_ERR_LAUNCH;
  }
  return PCAA_OK;
}
