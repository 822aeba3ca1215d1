// Which SIMD does wave w of a workgroup land on?  Prints HW_ID.SIMD_ID / CU_ID of the waves of
// a few 256- and 512-thread workgroups (one workgroup per CU, as the dense kernels launch).
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void k(unsigned *out) {
  unsigned id;
  asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(id));
  if ((threadIdx.x & 63) == 0) out[blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64] = id;
}
int main() {
  unsigned *d, h[8 * 64];
  (void)hipMalloc(&d, sizeof(h));
  for (int nt = 256; nt <= 512; nt *= 2) {
    hipLaunchKernelGGL(k, dim3(8), dim3(nt), 0, 0, d);
    (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
    for (int b = 0; b < 4; ++b) {
      printf("threads %d block %d: ", nt, b);
      for (int w = 0; w < nt / 64; ++w) {
        const unsigned v = h[b * (nt / 64) + w];
        printf("w%d simd %u cu %u wave %u | ", w, (v >> 4) & 3, (v >> 8) & 15, v & 15);
      }
      printf("\n");
    }
  }
  return 0;
}
