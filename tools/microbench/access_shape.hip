// GPU box microbenchmark: does the ACCESS SHAPE of the fused row kernels cost HBM rate?
// Phase B / C load a 16-row x 256-B tile per wave as four instructions of 16 separate 64-B pieces (lane = (row, k-slot):
// what the MFMA operand layout wants); a plain copy loads 1 KB contiguous per instruction.  Copies `streams` arrays to
// `streams` others with either shape, persistent grid of 1024 blocks x 256 threads, 16 B per lane and instruction.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/access_shape.hip -o /tmp/access_shape && /tmp/access_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef double dv2 __attribute__((ext_vector_type(2)));
struct Ptrs { const dv2* in[9]; dv2* out[9]; };

template <bool TILE>
__global__ void __launch_bounds__(256) k_copy(Ptrs p, int streams, long ntiles) {  // a tile = 16 rows x 16 elements of 16 B = 4 KB
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  for (long t = (long)blockIdx.x * 4 + wave; t < ntiles; t += (long)gridDim.x * 4) {
    for (int s = 0; s < streams; ++s) {
      const dv2* src = p.in[s] + t * 256;
      dv2* dst = p.out[s] + t * 256;
      dv2 v[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = TILE ? (lane & 15) * 16 + (lane >> 4) + 4 * k : lane + 64 * k;
        v[k] = __builtin_nontemporal_load(src + e);
      }
#pragma unroll
      for (int k = 0; k < 4; ++k) {
        const int e = TILE ? (lane & 15) * 16 + (lane >> 4) + 4 * k : lane + 64 * k;
        __builtin_nontemporal_store(v[k], dst + e);
      }
    }
  }
}

int main() {
  const long bytes = 12L << 30;  // one 64^4 x 768 B field
  const int maxs = 9;
  std::vector<void*> bufs;
  Ptrs p{};
  for (int s = 0; s < maxs; ++s) {
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed at %d\n", s); return 1; }
    hipMemset(a, 1, bytes);
    p.in[s] = (const dv2*)a; p.out[s] = (dv2*)b;
  }
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int streams : {1, 3, 9})
    for (int tile = 0; tile < 2; ++tile) {
      float best = 1e9f;
      for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        if (tile) hipLaunchKernelGGL(k_copy<true>, dim3(1024), dim3(256), 0, 0, p, streams, bytes / 4096);
        else hipLaunchKernelGGL(k_copy<false>, dim3(1024), dim3(256), 0, 0, p, streams, bytes / 4096);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
      }
      printf("streams %d  shape %-22s %.2f ms  %.2f TB/s (read + written)\n", streams, tile ? "16 rows x 64 B pieces" : "1 KB contiguous", best,
             2.0 * streams * bytes / best / 1e9);
    }
  return 0;
}
