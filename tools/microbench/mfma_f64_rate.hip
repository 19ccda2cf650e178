// GPU box microbenchmark: issue rate of v_mfma_f64_16x16x4_f64 on gfx950 -- the matrix-pipe ceiling of the fused row kernels
// (phase B, phase C, k_phaseC_multi).  Every SIMD of every CU runs `waves` waves of back-to-back MFMAs on 8 independent
// accumulators; in-kernel cycles per MFMA (s_memtime is a constant 100 MHz tick, so the wall clock is used instead) and
// the chip's TFLOP/s (2048 flop per MFMA and wave).
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/mfma_f64_rate.hip -o /tmp/mfma_f64_rate && /tmp/mfma_f64_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(1024) k_mfma(double* out, int iters, double a0, double b0) {
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = a0 + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  double* out;
  hipMalloc(&out, sizeof(double) * cus * 1024);
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000;
  for (int waves_per_simd = 1; waves_per_simd <= 4; waves_per_simd *= 2) {
    const int threads = 256 * waves_per_simd;
    hipLaunchKernelGGL(k_mfma, dim3(cus), dim3(threads), 0, 0, out, 100, 1.0, 1e-3);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_mfma, dim3(cus), dim3(threads), 0, 0, out, iters, 1.0, 1e-3);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0;
    hipEventElapsedTime(&ms, e0, e1);
    const double mfmas_per_simd = double(iters) * 8 * waves_per_simd;
    const double flops = mfmas_per_simd * 4 * cus * 2048.0;
    printf("waves/SIMD %d: %.3f ms, %.1f TFLOP/s fp64, %.1f ns per MFMA and SIMD (= %.1f cycles at %.2f GHz nominal)\n", waves_per_simd,
           ms, flops / (ms * 1e-3) / 1e12, ms * 1e6 / mfmas_per_simd, ms * 1e6 / mfmas_per_simd * prop.clockRate * 1e-6,
           prop.clockRate * 1e-6);
  }
  return 0;
}
