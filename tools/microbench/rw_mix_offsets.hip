// GPU box microbenchmark (round 5): does the RELATIVE OFFSET of the streams of a two-reads-one-write kernel matter?
// Fields of the lattices that matter are exactly 12 GiB and come back to back from the allocator, so every stream of a row
// kernel is at the same offset modulo any power of two at any moment.  Here each of the three streams (two read, one
// written; 12 GiB each, round-robin 4-KB tiles, 1 KB contiguous per wave-instruction) gets its own byte offset into its
// allocation.   hipcc -O3 --offload-arch=gfx950 tools/microbench/rw_mix_offsets.hip -o /tmp/rwo && /tmp/rwo
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));
__global__ void __launch_bounds__(256) k_mix(const dv2* a, const dv2* b, dv2* c, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += (long)gridDim.x * 4) {
    dv2 x[4], y[4];
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = __builtin_nontemporal_load(a + t * 256 + lane + 64 * k);
#pragma unroll
    for (int k = 0; k < 4; ++k) y[k] = __builtin_nontemporal_load(b + t * 256 + lane + 64 * k);
#pragma unroll
    for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(x[k] + y[k], c + t * 256 + lane + 64 * k);
  }
}
int main() {
  const long bytes = 12L << 30, slack = 64L << 20;
  char *A, *B, *C;
  if (hipMalloc(&A, bytes + slack) != hipSuccess || hipMalloc(&B, bytes + slack) != hipSuccess || hipMalloc(&C, bytes + slack) != hipSuccess) return 1;
  (void)hipMemset(A, 1, bytes + slack); (void)hipMemset(B, 1, bytes + slack);
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  auto run = [&](long oa, long ob, long oc) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_mix, dim3(1024), dim3(256), 0, 0, (const dv2*)(A + oa), (const dv2*)(B + ob), (dv2*)(C + oc), bytes / 4096);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("offsets read %8ld read %8ld written %8ld: %.2f ms  %.2f TB/s\n", oa, ob, oc, best, 3.0 * bytes / best / 1e9);
  };
  const long P[] = {256, 1024, 2048, 4096, 4352, 8192, 16384, 32768, 65536, 131072, 262144, 524288, 1048576, 2097152, 4194304, 8388608, 33554432};
  run(0, 0, 0);
  for (long p : P) run(0, 0, p);
  for (long p : {4352L, 65536L, 1048576L}) { run(0, p, 0); run(0, p, p); run(0, p, 2 * p); }
  run(0, 0, 0);
  return 0;
}
