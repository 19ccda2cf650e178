// GPU box microbenchmark: how often ONE wave can issue a v_fma_f64, alone on its SIMD and with a second wave beside it.
// The stencil's step is 288 fp64 FMAs per lane; its stamps (profiles/r04_stencil_pipe.txt) show a wave spending 8 cycles
// per FMA in the arithmetic phases.  Is that the instruction (then two waves per SIMD are needed for the 4-cycle rate and a
// third buys nothing for the arithmetic), or the stencil's mix (LDS reads between the FMAs, six dependent chains)?
//   MODE 0: 24 independent accumulators, operands in registers
//   MODE 1: 6 accumulators (the stencil's three colours x re / im), each a dependent chain
//   MODE 2: as 1, with the multiplier operands read from LDS (one ds_read_b128 per 4 FMAs, as the link entries are)
//   MODE 3: as 1, every FMA a v_fmac_f64_dpp whose multiplier is lane n's value of the lane's row of 16 (row_newbcast:n) --
//           a link entry held once per site, spread over the site's 16 lanes, instead of read from LDS by all 16
//           (the result is checked: lane l must see the value of lane (l & 48) + n)
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/fma_f64_issue.hip -o /tmp/fma_f64_issue && /tmp/fma_f64_issue
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) k_fma(double* out, long long* cycles, int iters, double seed) {
  extern __shared__ double lds[];
  const int lane = threadIdx.x & 63;
  for (int i = threadIdx.x; i < 2048; i += 256) lds[i] = seed + i * 1e-9;
  __syncthreads();
  double a[24];
#pragma unroll
  for (int k = 0; k < 24; ++k) a[k] = seed * (k + 1);
  double x = seed + lane * 1e-6, y = seed - lane * 1e-6;
  const dv2* L = reinterpret_cast<const dv2*>(lds) + (lane >> 4);  // 4 distinct addresses per wave: a broadcast read, like the links
  __builtin_amdgcn_sched_barrier(0);
  const long long t0 = __builtin_amdgcn_s_memtime();
  __builtin_amdgcn_sched_barrier(0);
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int k = 0; k < 24; ++k) a[k] = __builtin_fma(x, y, a[k]);
    } else if (MODE == 1) {
#pragma unroll
      for (int r = 0; r < 16; ++r)
#pragma unroll
        for (int k = 0; k < 6; ++k) a[k] = __builtin_fma(x, y, a[k]);
    } else if (MODE == 3) {
#pragma unroll
      for (int r = 0; r < 16; ++r) {
        asm volatile("v_fmac_f64_dpp %0, %6, %7 row_newbcast:0 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %1, -%6, %8 row_newbcast:3 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %2, %6, %7 row_newbcast:5 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %3, %6, %8 row_newbcast:9 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %4, -%6, %7 row_newbcast:12 row_mask:0xf bank_mask:0xf\n\t"
                     "v_fmac_f64_dpp %5, %6, %8 row_newbcast:15 row_mask:0xf bank_mask:0xf"
                     : "+v"(a[0]), "+v"(a[1]), "+v"(a[2]), "+v"(a[3]), "+v"(a[4]), "+v"(a[5])
                     : "v"(x), "v"(y), "v"(x));
      }
    } else {
#pragma unroll
      for (int r = 0; r < 24; ++r) {   // one link entry (16 B) feeds 4 FMAs: u.x * p.x, u.y * p.y, u.x * p.y, u.y * p.x
        const dv2 u = L[(r * 4) & 511];
        a[(r % 3) * 2] = __builtin_fma(u.x, x, a[(r % 3) * 2]);
        a[(r % 3) * 2] = __builtin_fma(-u.y, y, a[(r % 3) * 2]);
        a[(r % 3) * 2 + 1] = __builtin_fma(u.x, y, a[(r % 3) * 2 + 1]);
        a[(r % 3) * 2 + 1] = __builtin_fma(u.y, x, a[(r % 3) * 2 + 1]);
      }
    }
  }
  __builtin_amdgcn_sched_barrier(0);
  const long long t1 = __builtin_amdgcn_s_memtime();
  double s = 0;
#pragma unroll
  for (int k = 0; k < 24; ++k) s += a[k];
  if (s == 1.2345e300) out[0] = s;
  if (lane == 0) cycles[blockIdx.x * 4 + (threadIdx.x >> 6)] = t1 - t0;
  if (MODE == 3 && blockIdx.x == 0 && threadIdx.x < 64 && iters == 1) {  // semantics check (one trip): a[0] = seed + x(lane row*16 + 0) * y ...
    out[8 + lane * 2] = a[0];
    out[8 + lane * 2 + 1] = a[3];
  }
}

template <int MODE>
void run(double* out, long long* cyc, int cus, int waves_per_simd) {
  const int iters = 2000, fmas = 96 * iters;
  const size_t lds = waves_per_simd == 1 ? 100 * 1024 : (waves_per_simd == 2 ? 70 * 1024 : 40 * 1024);  // blocks per CU through the LDS size
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(k_fma<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds));
  const int grid = cus * waves_per_simd;
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  hipLaunchKernelGGL(k_fma<MODE>, dim3(grid), dim3(256), lds, 0, out, cyc, 50, 1.0);
  (void)hipDeviceSynchronize();
  (void)hipEventRecord(e0);
  hipLaunchKernelGGL(k_fma<MODE>, dim3(grid), dim3(256), lds, 0, out, cyc, iters, 1.0);
  (void)hipEventRecord(e1);
  (void)hipEventSynchronize(e1);
  float ms = 0;
  (void)hipEventElapsedTime(&ms, e0, e1);
  long long h[64];
  (void)hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
  double mean = 0;
  for (int i = 0; i < 64; ++i) mean += h[i] / 64.0;
  // s_memtime counts at a constant 100 MHz on this chip: wave-cycles from the launch time and the chip clock instead
  printf("mode %d, %d wave(s) per SIMD: %8.3f ms for %d FMAs per lane = %6.2f ns per FMA per wave; s_memtime ticks per wave %.0f (%.4f per FMA)\n", MODE,
         waves_per_simd, ms, fmas, ms * 1e6 / fmas, mean, mean / fmas);
}

int main() {
  hipDeviceProp_t prop;
  (void)hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  double* out;
  long long* cyc;
  (void)hipMalloc(&out, 8 * (8 + 128));
  (void)hipMalloc(&cyc, sizeof(long long) * cus * 3 * 4);
  printf("%d CUs, clock %d MHz (4 cycles per FMA at that clock = %.2f ns)\n", cus, prop.clockRate / 1000, 4e6 / prop.clockRate);
  for (int w : {1, 2, 3}) {
    run<0>(out, cyc, cus, w);
    run<1>(out, cyc, cus, w);
    run<2>(out, cyc, cus, w);
    run<3>(out, cyc, cus, w);
  }
  // row_newbcast semantics: one trip of mode 3, lane l: a[0] = 1 + 16 x(row(l), 0) y(l), a[3] = 4 + 16 x(row(l), 9) x(l)
  const size_t lds = 40 * 1024;
  hipLaunchKernelGGL(k_fma<3>, dim3(1), dim3(256), lds, 0, out, cyc, 1, 1.0);
  double h[8 + 128];
  (void)hipMemcpy(h, out, sizeof(h), hipMemcpyDeviceToHost);
  int bad = 0;
  for (int l = 0; l < 64; ++l) {
    const double xl = 1.0 + l * 1e-6, yl = 1.0 - l * 1e-6;
    auto xs = [](int q) { return 1.0 + q * 1e-6; };
    double a0 = 1.0, a3 = 4.0;
    for (int r = 0; r < 16; ++r) {
      a0 = __builtin_fma(xs((l & 48) + 0), yl, a0);
      a3 = __builtin_fma(xs((l & 48) + 9), xl, a3);
    }
    if (h[8 + 2 * l] != a0 || h[8 + 2 * l + 1] != a3) ++bad;
  }
  printf("row_newbcast:n on v_fmac_f64_dpp reads lane n of the lane's own row of 16: %s (%d of 64 lanes differ)\n", bad ? "NO" : "yes", bad);
  return 0;
}
