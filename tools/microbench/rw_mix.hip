// GPU box microbenchmark (round 5): what rate does a streaming kernel reach for a given MIX of read and written streams?
// The row kernels with two fields read and one written (phase B, k_phaseC_p0) run at 5.2-5.3 TB/s whatever their access shape
// (profiles/r05_phaseC_p0.txt), phase C with three read and two written at 5.8.  Here: R arrays read, W arrays written
// (the sum of the reads), nothing else, 1 KB contiguous per wave-instruction, persistent grids of 512 .. 4096 blocks of 256
// threads, all the loads of a tile in flight before the first use; fields of 12 GiB like the 64^4 x 768 B ones.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/rw_mix.hip -o /tmp/rw_mix && /tmp/rw_mix
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double dv2 __attribute__((ext_vector_type(2)));
struct Ptrs { const dv2* in[4]; dv2* out[4]; };

template <int R, int W>
__global__ void __launch_bounds__(256) k_mix(Ptrs p, long ntiles) {  // a tile = 4 KB per stream
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += (long)gridDim.x * 4) {
    dv2 v[R][4];
#pragma unroll
    for (int s = 0; s < R; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k) v[s][k] = __builtin_nontemporal_load(p.in[s] + t * 256 + lane + 64 * k);
    dv2 acc[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) {
      acc[k] = v[0][k];
#pragma unroll
      for (int s = 1; s < R; ++s) acc[k] += v[s][k];
    }
    if (W == 0) {  // keep the loads alive without a store stream
      if (acc[0].x + acc[1].x + acc[2].x + acc[3].x == 1.2345e300) p.out[0][t] = acc[0];
    }
#pragma unroll
    for (int s = 0; s < W; ++s)
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(acc[k] + (double)s, p.out[s] + t * 256 + lane + 64 * k);
  }
}

template <int R, int W>
void run(const Ptrs& p, long bytes, hipEvent_t e0, hipEvent_t e1) {
  for (int grid : {512, 1024, 2048, 4096}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      hipLaunchKernelGGL((k_mix<R, W>), dim3(grid), dim3(256), 0, 0, p, bytes / 4096);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("read %d  written %d  grid %4d: %.2f ms  %.2f TB/s (read %.2f + written %.2f)\n", R, W, grid, best,
           (R + W) * (double)bytes / best / 1e9, R * (double)bytes / best / 1e9, W * (double)bytes / best / 1e9);
  }
}

int main() {
  const long bytes = 12L << 30;
  Ptrs p{};
  for (int s = 0; s < 4; ++s) {
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed at %d\n", s); return 1; }
    hipMemset(a, 1, bytes);
    p.in[s] = (const dv2*)a; p.out[s] = (dv2*)b;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  run<1, 0>(p, bytes, e0, e1);
  run<2, 0>(p, bytes, e0, e1);
  run<1, 1>(p, bytes, e0, e1);
  run<2, 1>(p, bytes, e0, e1);
  run<3, 1>(p, bytes, e0, e1);
  run<3, 2>(p, bytes, e0, e1);
  run<4, 4>(p, bytes, e0, e1);
  return 0;
}
