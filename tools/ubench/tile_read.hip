// HBM read rate of the training GEMM's operand walk against a plain contiguous walk of the same matrix [M, K] fp32:
//   mode 0: the GEMM's pattern -- a workgroup owns 128 consecutive rows and reads them in k-steps of 32 floats (one 128-byte line of every
//           K*4-byte row per step), 8 lanes per line;   mode 1: the same workgroup reads its 128 rows contiguously (whole rows, 1 KiB per
//           wave instruction);   mode 2: mode 0 after the workgroup has touched its block contiguously once (warms L2 / the memory-side cache).
//   hipcc --offload-arch=gfx950 -O3 tile_read.hip -o tile_read && ./tile_read
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CHECK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

template <int MODE>
__global__ void __launch_bounds__(256, 2) k_read(const float4* __restrict__ a, int K4, float* __restrict__ out) {
  const int tid = threadIdx.x;
  const long row0 = (long)blockIdx.x * 128;
  float acc = 0.f;
  if (MODE == 1 || MODE == 2) {
    const float4* blk = a + row0 * K4;
    for (int i = tid; i < 128 * K4; i += 256) { const float4 v = blk[i]; acc += v.x + v.w; }
  }
  if (MODE == 0 || MODE == 2) {
    for (int k4 = 0; k4 < K4; k4 += 8) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        const float4 v = a[(row0 + (tid >> 3) + 32 * i) * K4 + k4 + (tid & 7)];
        acc += v.x + v.w;
      }
      __syncthreads();
    }
  }
  if (acc == 123.456f) out[0] = acc;
}

int main() {
  const long M = 3072L * 64 * 15;
  for (int K : {128, 256}) {
    float4* a; float* out;
    CHECK(hipMalloc(&a, M * K * 4)); CHECK(hipMalloc(&out, 4));
    CHECK(hipMemset(a, 0, M * K * 4));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int mode = 0; mode < 3; ++mode) {
      float ms = 0.f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (mode == 0) hipLaunchKernelGGL(k_read<0>, dim3(M / 128), dim3(256), 0, 0, a, K / 4, out);
        if (mode == 1) hipLaunchKernelGGL(k_read<1>, dim3(M / 128), dim3(256), 0, 0, a, K / 4, out);
        if (mode == 2) hipLaunchKernelGGL(k_read<2>, dim3(M / 128), dim3(256), 0, 0, a, K / 4, out);
        hipEventRecord(e1); hipEventSynchronize(e1); hipEventElapsedTime(&ms, e0, e1);
      }
      printf("K %d mode %d: %.1f us  %.2f TB/s (matrix bytes / time)\n", K, mode, ms * 1e3, (double)M * K * 4 / ms / 1e9);
    }
    hipFree(a); hipFree(out);
  }
  return 0;
}
