// A kernel that keeps N CUs busy for a fixed time (every workgroup takes a CU's whole LDS and sleeps a fixed number
// of rounds): stands in for a collective's persistent kernels when timing the GEMM's tile loop beside one.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC -o libhog.so hog.hip
#include <hip/hip_runtime.h>
__global__ void hog_kernel(int rounds, int* out) {
  extern __shared__ int lds[];
  lds[threadIdx.x] = threadIdx.x;
  for (int i = 0; i < rounds; ++i) __builtin_amdgcn_s_sleep(127);
  if (lds[threadIdx.x] == -1) out[0] = 1;
}
extern "C" int hog_launch(int workgroups, int rounds, int* out, void* stream) {
  static bool cfg = false;
  if (!cfg) { (void)hipFuncSetAttribute((const void*)hog_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, 163840); cfg = true; }
  hipLaunchKernelGGL(hog_kernel, dim3(workgroups), dim3(64), 163840, (hipStream_t)stream, rounds, out);
  return (int)hipGetLastError();
}
