// GPU box microbenchmark (round 5): does it pay to separate reads and writes IN TIME?
// A streaming kernel with two arrays read and one written (phase B, k_phaseC_p0) reaches 5.0-5.3 TB/s where pure reads reach
// 6.7-7.1 (rw_mix.hip).  If the difference is the memory's read <-> write turnaround, batching helps: every block reads N
// tile pairs, keeps the N results in LDS (N x 4 KB), then writes them in one burst -- block-local phases -- and, with a
// grid-wide barrier between the phases (one block per CU, all resident; bounded spins), the whole chip alternates between
// reading 2 x N x 4 KB x 256 and writing N x 4 KB x 256.
//   hipcc -O3 --offload-arch=gfx950 tools/microbench/rw_phased.hip -o /tmp/rw_phased && /tmp/rw_phased
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double dv2 __attribute__((ext_vector_type(2)));

__global__ void __launch_bounds__(256) k_plain(const dv2* a, const dv2* b, dv2* out, long ntiles) {
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

// N tiles per block and phase; 512 threads = 8 waves, wave w takes tiles w, w + 8, ... of the block's chunk
template <int N, bool GRID_SYNC>
__global__ void __launch_bounds__(512) k_phased(const dv2* a, const dv2* b, dv2* out, long ntiles, unsigned* counter) {
  extern __shared__ dv2 lds[];  // N tiles of 256 dv2
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long chunks = ntiles / N;  // (ntiles is a multiple of N * gridDim.x here)
  unsigned phase = 0;
  bool alone = false;  // thread 0: a barrier timed out once -- no further waiting (every block still counts its arrivals)
  auto grid_barrier = [&]() {
    if (!GRID_SYNC) return;
    __syncthreads();
    if (threadIdx.x == 0) {
      __threadfence();
      atomicAdd(counter, 1u);
      ++phase;
      const unsigned want = phase * gridDim.x;
      if (!alone) {
        alone = true;
        for (int spin = 0; spin < 200000; ++spin) {  // bounded: a block that cannot see the others goes on alone
          if (__atomic_load_n(counter, __ATOMIC_RELAXED) >= want) {
            alone = false;
            break;
          }
          __builtin_amdgcn_s_sleep(1);
        }
      }
    }
    __syncthreads();
  };
  for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
    const long t0 = c * N;
    for (int i = wave; i < N; i += 8) {
      dv2 va[4], vb[4];
#pragma unroll
      for (int k = 0; k < 4; ++k) va[k] = __builtin_nontemporal_load(a + (t0 + i) * 256 + lane + 64 * k);
#pragma unroll
      for (int k = 0; k < 4; ++k) vb[k] = __builtin_nontemporal_load(b + (t0 + i) * 256 + lane + 64 * k);
#pragma unroll
      for (int k = 0; k < 4; ++k) lds[i * 256 + lane + 64 * k] = va[k] + vb[k];
    }
    __syncthreads();
    grid_barrier();
    for (int i = wave; i < N; i += 8) {
#pragma unroll
      for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(lds[i * 256 + lane + 64 * k], out + (t0 + i) * 256 + lane + 64 * k);
    }
    __syncthreads();
    grid_barrier();
  }
}

// The same chunks without LDS: MODE 0 = a wave loads, adds and stores tile by tile (contiguous chunks only);
// MODE 1 = a wave loads ALL its N/8 tile pairs, then stores all its results (register-batched); MODE 2 = MODE 1 with a block
// barrier between the loads and the stores (the block's stores go out together).
template <int N, int MODE>
__global__ void __launch_bounds__(512) k_chunked(const dv2* a, const dv2* b, dv2* out, long ntiles) {
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const long chunks = ntiles / N;
  constexpr int PER = N / 8;
  for (long c = blockIdx.x; c < chunks; c += gridDim.x) {
    const long t0 = c * N;
    if (MODE == 0) {
      for (int i = wave; i < N; i += 8) {
        dv2 va[4], vb[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) va[k] = __builtin_nontemporal_load(a + (t0 + i) * 256 + lane + 64 * k);
#pragma unroll
        for (int k = 0; k < 4; ++k) vb[k] = __builtin_nontemporal_load(b + (t0 + i) * 256 + lane + 64 * k);
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(va[k] + vb[k], out + (t0 + i) * 256 + lane + 64 * k);
      }
    } else {
      dv2 va[PER][4], vb[PER][4];
#pragma unroll
      for (int j = 0; j < PER; ++j) {
#pragma unroll
        for (int k = 0; k < 4; ++k) va[j][k] = __builtin_nontemporal_load(a + (t0 + wave + 8 * j) * 256 + lane + 64 * k);
#pragma unroll
        for (int k = 0; k < 4; ++k) vb[j][k] = __builtin_nontemporal_load(b + (t0 + wave + 8 * j) * 256 + lane + 64 * k);
      }
#pragma unroll
      for (int j = 0; j < PER; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) va[j][k] += vb[j][k];
      if (MODE == 2) __syncthreads();
#pragma unroll
      for (int j = 0; j < PER; ++j)
#pragma unroll
        for (int k = 0; k < 4; ++k) __builtin_nontemporal_store(va[j][k], out + (t0 + wave + 8 * j) * 256 + lane + 64 * k);
    }
  }
}

template <int N, int MODE>
static void run_chunked(const dv2* a, const dv2* b, dv2* o, long bytes, int grid, hipEvent_t e0, hipEvent_t e1) {
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_chunked<N, MODE>), dim3(grid), dim3(512), 0, 0, a, b, o, bytes / 4096);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  static const char* const names[3] = {"tile by tile", "register-batched", "register-batched + block barrier"};
  printf("chunks of %2d tiles per block, %s, grid %d: %.2f ms  %.2f TB/s\n", N, names[MODE], grid, best, 3.0 * bytes / best / 1e9);
}

template <int N, bool GS>
static void run_phased(const dv2* a, const dv2* b, dv2* o, long bytes, unsigned* counter, int grid, hipEvent_t e0, hipEvent_t e1) {
  (void)hipFuncSetAttribute((const void*)k_phased<N, GS>, hipFuncAttributeMaxDynamicSharedMemorySize, N * 4096);
  float best = 1e9f;
  for (int rep = 0; rep < 3; ++rep) {
    (void)hipMemset(counter, 0, sizeof(unsigned));
    (void)hipEventRecord(e0);
    hipLaunchKernelGGL((k_phased<N, GS>), dim3(grid), dim3(512), N * 4096, 0, a, b, o, bytes / 4096, counter);
    (void)hipEventRecord(e1);
    (void)hipEventSynchronize(e1);
    float ms;
    (void)hipEventElapsedTime(&ms, e0, e1);
    if (ms < best) best = ms;
  }
  printf("phased, %2d tiles (%3d KB) per block and phase, %s, grid %d: %.2f ms  %.2f TB/s\n", N, N * 4, GS ? "grid-wide phases" : "block-local phases",
         grid, best, 3.0 * bytes / best / 1e9);
}

int main() {
  const long bytes = 12L << 30;  // 3145728 tiles: a multiple of 32 * 256
  void *a, *b, *o;
  unsigned* counter;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess || hipMalloc(&o, bytes) != hipSuccess ||
      hipMalloc(&counter, sizeof(unsigned)) != hipSuccess) {
    printf("alloc failed\n");
    return 1;
  }
  (void)hipMemset(a, 1, bytes);
  (void)hipMemset(b, 1, bytes);
  hipEvent_t e0, e1;
  (void)hipEventCreate(&e0);
  (void)hipEventCreate(&e1);
  for (int grid : {512, 1024}) {
    float best = 1e9f;
    for (int rep = 0; rep < 3; ++rep) {
      (void)hipEventRecord(e0);
      hipLaunchKernelGGL(k_plain, dim3(grid), dim3(256), 0, 0, (const dv2*)a, (const dv2*)b, (dv2*)o, bytes / 4096);
      (void)hipEventRecord(e1);
      (void)hipEventSynchronize(e1);
      float ms;
      (void)hipEventElapsedTime(&ms, e0, e1);
      if (ms < best) best = ms;
    }
    printf("plain streaming, grid %d: %.2f ms  %.2f TB/s\n", grid, best, 3.0 * bytes / best / 1e9);
  }
  run_phased<8, false>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, counter, 256, e0, e1);
  run_phased<16, false>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, counter, 256, e0, e1);
  run_phased<32, false>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, counter, 256, e0, e1);
  run_phased<16, false>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, counter, 512, e0, e1);
  for (int grid : {256, 512}) {
    run_chunked<16, 0>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, grid, e0, e1);
    run_chunked<32, 0>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, grid, e0, e1);
    run_chunked<16, 1>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, grid, e0, e1);
    run_chunked<32, 1>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, grid, e0, e1);
    run_chunked<16, 2>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, grid, e0, e1);
    run_chunked<32, 2>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, grid, e0, e1);
  }
  run_phased<32, true>((const dv2*)a, (const dv2*)b, (dv2*)o, bytes, counter, 256, e0, e1);
  return 0;
}
