// GPU box microbenchmark: what a READ-ONLY stream reaches on this chip, and how it depends on the loads in flight per CU.
// Each wave sums `unroll` independent 1 KB loads (16 B per lane) per loop trip over a 16 GiB array, persistent grid of
// `blocks_per_cu` x 256 CUs blocks of 256 threads.  Little's law on the result: bytes in flight per CU = rate x latency.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/read_rate.hip -o /tmp/read_rate && /tmp/read_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));

template <int UNROLL>
__global__ void __launch_bounds__(256) k_read(const dv2* __restrict__ in, long n_chunks, double* out) {  // chunk = 64 lanes x 16 B = 1 KB
  const int lane = threadIdx.x & 63;
  const long wave = (long)blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = (long)gridDim.x * 4;
  dv2 acc = {0.0, 0.0};
  for (long c = wave * UNROLL; c + UNROLL <= n_chunks; c += nwaves * UNROLL) {
    dv2 v[UNROLL];
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) v[k] = __builtin_nontemporal_load(in + (c + k) * 64 + lane);
#pragma unroll
    for (int k = 0; k < UNROLL; ++k) acc += v[k];
  }
  if (acc.x + acc.y == 1.2345e300) out[0] = acc.x;  // keep the loads
}

template <int UNROLL>
void run(const dv2* in, long n_chunks, double* out, int blocks_per_cu, int cus) {
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  hipLaunchKernelGGL(k_read<UNROLL>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, in, n_chunks, out);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  hipLaunchKernelGGL(k_read<UNROLL>, dim3(cus * blocks_per_cu), dim3(256), 0, 0, in, n_chunks, out);
  hipEventRecord(e1);
  hipEventSynchronize(e1);
  float ms = 0;
  hipEventElapsedTime(&ms, e0, e1);
  const double gb = n_chunks * 1024.0 / 1e9;
  printf("loads in flight per wave %2d, waves per CU %2d (= %4d KB in flight per CU): %.2f ms, %.2f TB/s read\n", UNROLL, 4 * blocks_per_cu,
         UNROLL * 4 * blocks_per_cu, ms, gb / ms);
}

int main() {
  hipDeviceProp_t prop;
  hipGetDeviceProperties(&prop, 0);
  const int cus = prop.multiProcessorCount;
  const long bytes = 16L << 30, n_chunks = bytes / 1024;
  dv2* in;
  double* out;
  hipMalloc(&in, bytes);
  hipMalloc(&out, 64);
  hipMemset(in, 0, bytes);
  for (int bpc : {1, 2, 4, 8}) {
    run<1>(in, n_chunks, out, bpc, cus);
    run<2>(in, n_chunks, out, bpc, cus);
    run<4>(in, n_chunks, out, bpc, cus);
    run<8>(in, n_chunks, out, bpc, cus);
    run<16>(in, n_chunks, out, bpc, cus);
  }
  return 0;
}
