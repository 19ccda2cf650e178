// GPU box microbenchmark: what a CU's vector-memory path takes per instruction -- `global_load_dwordx4` into registers against
// `global_load_lds_dwordx4` (LDS-DMA) into LDS -- on data that stays in the L2, with 8 waves per CU (two per SIMD, the
// stencil's occupancy) and 1 .. 8 of them issuing.  Each wave re-reads 8 KB of its own (1 KB per instruction, 16 B per
// lane), 8 instructions in flight, L2-resident data, so neither HBM nor the latency is the limit: the result is the issue /
// address / return rate of the path itself.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/lds_dma_rate.hip -o /tmp/lds_dma_rate && /tmp/lds_dma_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));

// MODE 0: loads into registers; 1: LDS-DMA (m0-based destination, saved / restored like the product does); 2: LDS-DMA, m0 set once per group
template <int MODE>
__global__ void __launch_bounds__(256) k_rate(const char* __restrict__ in, int iters, int active_waves, double* out) {
  extern __shared__ char lds[];
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int wid = blockIdx.x * 4 + wave;
  if ((wid & 7) >= active_waves) return;  // (wave ids 8 c .. 8 c + 7 share a CU only if the dispatcher places blocks 2 c, 2 c + 1 together: an assumption of this reading)
  const char* base = in + static_cast<long>(wid) * 8192 + lane * 16;  // 8 KB per wave, re-read every trip: 64 KB per CU (past its 32 KB L1), 2 MB per XCD (inside its L2)
  const unsigned ldsbase = __builtin_amdgcn_readfirstlane(static_cast<unsigned>(reinterpret_cast<uintptr_t>((__attribute__((address_space(3))) char*)lds)) + wave * 8192);
  dv2 acc = {0.0, 0.0};
  for (int it = 0; it < iters; ++it) {
    const char* p = base;
    if (MODE == 0) {
      dv2 v[8];
#pragma unroll
      for (int k = 0; k < 8; ++k) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(v[k]) : "v"(p + k * 1024));
      asm volatile("s_waitcnt vmcnt(0)" : "+v"(v[0]), "+v"(v[1]), "+v"(v[2]), "+v"(v[3]), "+v"(v[4]), "+v"(v[5]), "+v"(v[6]), "+v"(v[7]));
#pragma unroll
      for (int k = 0; k < 8; ++k) acc += v[k];
    } else if (MODE == 1) {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        unsigned keep;
        asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                     : "=&s"(keep) : "v"(p + k * 1024), "s"(ldsbase + k * 1024));
      }
      asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
      unsigned keep;
      asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %1" : "=&s"(keep) : "s"(ldsbase));
#pragma unroll
      for (int k = 0; k < 8; ++k)
        asm volatile("s_nop 0\n\tglobal_load_lds_dwordx4 %0, off\n\ts_add_u32 m0, m0, 1024" : : "v"(p + k * 1024));
      asm volatile("s_mov_b32 m0, %0\n\ts_waitcnt vmcnt(0)" : : "s"(keep) : "memory");
    }
  }
  if (MODE != 0) {
    __syncthreads();
    acc += reinterpret_cast<dv2*>(lds)[threadIdx.x];
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;
}

template <int MODE>
void run(const char* in, double* out, int cus, int active) {
  const int iters = 4000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int grid = cus * 2;  // 8 waves per CU
  hipFuncSetAttribute(reinterpret_cast<const void*>(k_rate<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, 65536);
  hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 65536, 0, in, 200, active, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_rate<MODE>, dim3(grid), dim3(256), 65536, 0, in, iters, active, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  // per CU: `active` waves x iters x 8 instructions of 1 KB
  const double instr_per_cu = static_cast<double>(active) * iters * 8;
  const double ns_per_instr = ms * 1e6 / instr_per_cu;
  printf("%-34s %d of 8 waves per CU issuing: %7.2f ms, %6.1f ns per wave-instruction per CU (%5.1f cycles at 2.0 GHz), %6.1f GB/s per CU\n",
         MODE == 0 ? "global_load_dwordx4 -> registers" : (MODE == 1 ? "global_load_lds_dwordx4 (m0 saved)" : "global_load_lds_dwordx4 (m0 once)"),
         active, ms, ns_per_instr, ns_per_instr * 2.0, 1024.0 / ns_per_instr);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const long bytes = static_cast<long>(cus) * 8 * 8192;
  char* in;
  double* out;
  hipMalloc(&in, bytes);
  hipMalloc(&out, 64);
  hipMemset(in, 0, bytes);
  for (int active : {1, 2, 4, 8}) {
    run<0>(in, out, cus, active);
    run<1>(in, out, cus, active);
    run<2>(in, out, cus, active);
  }
  return 0;
}
