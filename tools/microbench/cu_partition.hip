// GPU box microbenchmark (round 5): can an fp64-matrix-bound kernel and a streaming kernel share the chip side by side?
// The closing pass of a group (k_phaseC_multi) is bound by the fp64 matrix pipe and leaves HBM at 4.5 TB/s; every other kernel
// of the iteration streams and leaves the matrix pipe mostly idle.  Both kinds fill a CU's LDS, so
// they could only run beside each other on disjoint CUs.  Here: streams created with hipExtStreamCreateWithCUMask, a
// streaming kernel (two 12-GiB arrays read, one written, like phase B) and a matrix kernel (back-to-back
// v_mfma_f64_16x16x4_f64, 139 KB of LDS per block so that nothing else fits its CU), alone and together:
//   * does the mask take effect (the matrix kernel's time on 1/2, 1/4 of the CUs)?
//   * what does the streaming kernel reach on 3/4, 1/2 of the CUs?
//   * and both at once?
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/cu_partition.hip -o /tmp/cu_partition && /tmp/cu_partition
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <cstdint>
#include <vector>
typedef double dv2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

__global__ void __launch_bounds__(256) k_stream(const dv2* a, const dv2* b, dv2* out, long ntiles) {  // a tile = 4 KB per stream
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += (long)gridDim.x * 4) {
    dv2 va[4], vb[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) va[k] = __builtin_nontemporal_load(a + t * 256 + lane + 64 * k);
#pragma unroll
    for (int k = 0; k < 4; ++k) vb[k] = __builtin_nontemporal_load(b + t * 256 + lane + 64 * k);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(va[k] + vb[k], out + t * 256 + lane + 64 * k);
  }
}

__global__ void __launch_bounds__(512) k_matrix(double* out, int iters, double a0, double b0) {  // 8 waves: 2 per SIMD
  extern __shared__ double lds[];
  if (threadIdx.x == 0) lds[0] = a0;  // the LDS is only there to keep other blocks off the CU
  __syncthreads();
  d4 acc[8];
  for (int i = 0; i < 8; ++i) acc[i] = d4{0.0, 0.0, 0.0, 0.0};
  double a = lds[0] + threadIdx.x * 1e-9, b = b0;
  for (int it = 0; it < iters; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
  }
  double s = 0.0;
  for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

static double now_ms() {
  return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

struct Masked {
  hipStream_t s = nullptr;
  int cus = 0;
};
// how = 0: the first `n` mask bits; 1: bits spread evenly (every k-th)
static Masked make_stream(int total_cus, int n, int how, int offset) {
  std::vector<uint32_t> mask((total_cus + 31) / 32, 0u);
  Masked m;
  for (int i = 0; i < n; ++i) {
    int bit = how == 0 ? (offset + i) % total_cus : (int)(((long)i * total_cus) / n + offset) % total_cus;
    if (!(mask[bit / 32] >> (bit % 32) & 1u)) ++m.cus;
    mask[bit / 32] |= 1u << (bit % 32);
  }
  if (hipExtStreamCreateWithCUMask(&m.s, (uint32_t)mask.size(), mask.data()) != hipSuccess) {
    printf("hipExtStreamCreateWithCUMask failed: %s\n", hipGetErrorString(hipGetLastError()));
    m.s = nullptr;
  }
  return m;
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const long bytes = 12L << 30;
  void *a, *b, *o;
  double* mout;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, bytes) != hipSuccess ||
      hipMalloc(&mout, sizeof(double) * 512 * 1024) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  hipMemset(a, 1, bytes);
  hipMemset(b, 1, bytes);
  const size_t lds = 139 * 1024;
  hipFuncSetAttribute((const void*)k_matrix, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  const int iters = 60000;  // per wave: 8 x iters MFMAs
  auto stream_ms = [&](hipStream_t s, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_stream, dim3(grid), dim3(256), 0, s, (const dv2*)a, (const dv2*)b, (dv2*)o, bytes / 4096);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms;
  };
  auto matrix_ms = [&](hipStream_t s, int grid) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipEventRecord(e0, s);
    hipLaunchKernelGGL(k_matrix, dim3(grid), dim3(512), lds, s, mout, iters, 1.0, 1e-3);
    hipEventRecord(e1, s);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    return (double)ms;
  };
  auto tbs = [&](double ms) { return 3.0 * bytes / ms / 1e9; };
  auto tflops = [&](double ms, int grid) { return (double)iters * 8 * 8 * grid * 2048.0 / (ms * 1e-3) / 1e12; };
  printf("%d CUs\n", cus);
  // warm-up and baselines on the null stream (all CUs)
  stream_ms(nullptr, 1024);
  matrix_ms(nullptr, cus);
  const double s_all = stream_ms(nullptr, 1024), m_all = matrix_ms(nullptr, cus);
  printf("all CUs: streaming %.2f ms = %.2f TB/s; matrix kernel (%d blocks) %.2f ms = %.1f TFLOP/s\n", s_all, tbs(s_all), cus, m_all,
         tflops(m_all, cus));
  for (int how = 0; how < 2; ++how) {
    printf("--- mask bits %s\n", how == 0 ? "contiguous" : "spread evenly");
    for (int quarter = 1; quarter <= 2; ++quarter) {  // matrix kernel on 1/4, 1/2 of the CUs; streaming on the rest
      const int n_m = cus * quarter / 4, n_s = cus - n_m;
      Masked sm = how == 0 ? make_stream(cus, n_m, 0, 0) : make_stream(cus, n_m, 1, 0);
      // the complement of the matrix stream's mask
      std::vector<uint32_t> mask((cus + 31) / 32, 0u);
      {
        std::vector<char> used(cus, 0);
        for (int i = 0; i < n_m; ++i) used[how == 0 ? i : (int)(((long)i * cus) / n_m)] = 1;
        for (int i = 0; i < cus; ++i)
          if (!used[i]) mask[i / 32] |= 1u << (i % 32);
      }
      Masked ss;
      ss.cus = n_s;
      if (hipExtStreamCreateWithCUMask(&ss.s, (uint32_t)mask.size(), mask.data()) != hipSuccess) ss.s = nullptr;
      if (!sm.s || !ss.s) return 1;
      // the same number of blocks as on the whole chip: if the mask works, the matrix kernel takes cus / n_m times as long
      const double m_alone = matrix_ms(sm.s, cus);
      const double m_fit = matrix_ms(sm.s, n_m);  // one block per CU of the partition
      const double s_alone = stream_ms(ss.s, 1024);
      printf("matrix on %d CUs: %d blocks %.2f ms (x%.2f of all CUs), %d blocks %.2f ms = %.1f TFLOP/s; streaming on %d CUs alone: %.2f ms = %.2f TB/s\n",
             sm.cus, cus, m_alone, m_alone / m_all, n_m, m_fit, tflops(m_fit, n_m), n_s, s_alone, tbs(s_alone));
      // together: the matrix kernel sized to last about as long as the streaming kernel
      const int reps_m = (int)(s_alone / m_fit + 0.5) > 0 ? (int)(s_alone / m_fit + 0.5) : 1;
      hipEvent_t a0, a1, b0, b1;
      hipEventCreate(&a0); hipEventCreate(&a1); hipEventCreate(&b0); hipEventCreate(&b1);
      hipDeviceSynchronize();
      const double t0 = now_ms();
      hipEventRecord(a0, sm.s);
      for (int r = 0; r < reps_m; ++r) hipLaunchKernelGGL(k_matrix, dim3(n_m), dim3(512), lds, sm.s, mout, iters, 1.0, 1e-3);
      hipEventRecord(a1, sm.s);
      hipEventRecord(b0, ss.s);
      hipLaunchKernelGGL(k_stream, dim3(1024), dim3(256), 0, ss.s, (const dv2*)a, (const dv2*)b, (dv2*)o, bytes / 4096);
      hipEventRecord(b1, ss.s);
      hipDeviceSynchronize();
      const double wall = now_ms() - t0;
      float ms_m, ms_s;
      hipEventElapsedTime(&ms_m, a0, a1);
      hipEventElapsedTime(&ms_s, b0, b1);
      printf("  together: matrix x%d %.2f ms (alone %.2f) = %.1f TFLOP/s, streaming %.2f ms = %.2f TB/s (alone %.2f), wall %.2f ms\n", reps_m, ms_m,
             reps_m * m_fit, tflops(ms_m / reps_m, n_m), ms_s, tbs(ms_s), tbs(s_alone), wall);
      hipStreamDestroy(sm.s);
      hipStreamDestroy(ss.s);
    }
  }
  return 0;
}
