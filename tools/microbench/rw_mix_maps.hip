#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));
struct Ptrs { const dv2* in[4]; dv2* out[4]; };
// MAP 0: tiles dealt round-robin to waves (tile = i*G*4 + block*4 + wave)   [the row kernels]
// MAP 1: each block owns one contiguous range of tiles; its 4 waves round-robin inside
// MAP 2: round-robin in units of 4 tiles (16 KB per stream and wave before the next stream)
// NT: nontemporal loads/stores or plain
template <int R, int W, int MAP, bool NT>
__global__ void __launch_bounds__(256) k_mix(Ptrs p, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long G = gridDim.x;
  const long per = (ntiles + G - 1) / G;
  long t0, t1, step;
  if (MAP == 1) { t0 = blockIdx.x * per + wave; t1 = (blockIdx.x + 1) * per < ntiles ? (blockIdx.x + 1) * per : ntiles; step = 4; }
  else { t0 = (long)blockIdx.x * 4 + wave; t1 = ntiles; step = G * 4; }
  constexpr int U = MAP == 2 ? 4 : 1;
  if (MAP == 2) { t0 *= 4; step *= 4; }
  for (long t = t0; t < t1; t += step) {
    dv2 v[R][4 * U];
#pragma unroll
    for (int s = 0; s < R; ++s)
#pragma unroll
      for (int k = 0; k < 4 * U; ++k) { const dv2* q = p.in[s] + t * 256 + lane + 64 * k; v[s][k] = NT ? __builtin_nontemporal_load(q) : *q; }
#pragma unroll
    for (int k = 0; k < 4 * U; ++k) {
      dv2 acc = v[0][k];
#pragma unroll
      for (int s = 1; s < R; ++s) acc += v[s][k];
#pragma unroll
      for (int s = 0; s < W; ++s) { dv2* q = p.out[s] + t * 256 + lane + 64 * k; if (NT) __builtin_nontemporal_store(acc + (double)s, q); else *q = acc + (double)s; }
    }
  }
}
template <int R, int W, int MAP, bool NT>
void run(const Ptrs& p, long bytes, hipEvent_t e0, hipEvent_t e1, const char* what) {
  for (int grid : {1024, 4096}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL((k_mix<R, W, MAP, NT>), dim3(grid), dim3(256), 0, 0, p, bytes / 4096);
      (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
      float ms; (void)hipEventElapsedTime(&ms, e0, e1); if (ms < best) best = ms;
    }
    printf("R%d W%d %-34s grid %4d: %.2f ms  %.2f TB/s\n", R, W, what, grid, best, (R + W) * (double)bytes / best / 1e9);
  }
}
int main(int argc, char** argv) {
  const long bytes = 12L << 30;
  const long pad = argc > 1 ? atol(argv[1]) : 0;  // extra bytes in front of every second allocation (relative placement)
  Ptrs p{};
  for (int s = 0; s < 4; ++s) {
    void *a, *b;
    if (hipMalloc(&a, bytes + pad) != hipSuccess || hipMalloc(&b, bytes + pad) != hipSuccess) return 1;
    (void)hipMemset(a, 1, bytes);
    p.in[s] = (const dv2*)a; p.out[s] = (dv2*)((char*)b + pad);
  }
  hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
  printf("pad %ld\n", pad);
  run<2, 1, 0, true>(p, bytes, e0, e1, "round-robin tiles, nontemporal");
  run<2, 1, 0, false>(p, bytes, e0, e1, "round-robin tiles, plain");
  run<2, 1, 1, true>(p, bytes, e0, e1, "contiguous range per block, nt");
  run<2, 1, 2, true>(p, bytes, e0, e1, "round-robin 16 KB units, nt");
  run<2, 1, 2, false>(p, bytes, e0, e1, "round-robin 16 KB units, plain");
  Ptrs q = p; q.out[0] = (dv2*)p.in[1];  // write over the second input (in place, like phase B's Q)
  run<2, 1, 0, true>(q, bytes, e0, e1, "in place over input 1, nt");
  return 0;
}
