// Probe: how fast can 256 workgroups write a 32 MiB fp16 matrix (the GEMM epilogue floor)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned int v4u __attribute__((ext_vector_type(4)));
// tile-shaped: block b writes a 256x256 fp16 tile of a 4096x4096 matrix, 16 B per lane, 128-B row segments per 8 lanes
__global__ __launch_bounds__(512) void tile_store(v4u *D, int N) {
  const int bn = blockIdx.x % (N / 256), bm = blockIdx.x / (N / 256);
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const int wm = wave / 4, wn = wave % 4;
  v4u v = {threadIdx.x, blockIdx.x, 1u, 2u};
  for (int i = 0; i < 16; ++i) {
    const int row = bm * 256 + wm * 128 + i * 8 + (lane >> 3);
    const int col = bn * 256 + wn * 64 + (lane & 7) * 8;
    D[((size_t)row * N + col) / 8] = v;
  }
}
// flat: fully linear 16 B per lane
__global__ __launch_bounds__(512) void flat_store(v4u *D, size_t n16) {
  v4u v = {threadIdx.x, blockIdx.x, 1u, 2u};
  for (size_t i = blockIdx.x * 512 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 512) D[i] = v;
}
int main() {
  const int N = 4096; size_t bytes = (size_t)N * N * 2;
  v4u *D; hipMalloc(&D, bytes);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  for (int mode = 0; mode < 3; ++mode) {
    for (int rep = 0; rep < 3; ++rep) {
      hipEventRecord(e0);
      for (int i = 0; i < 20; ++i) {
        if (mode == 0) hipLaunchKernelGGL(tile_store, dim3(256), dim3(512), 0, 0, D, N);
        else hipLaunchKernelGGL(flat_store, dim3(mode == 1 ? 256 : 2048), dim3(512), 0, 0, D, bytes / 16);
      }
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      printf("mode %d: %.2f us per 32 MiB  (%.2f TB/s)\n", mode, ms / 20 * 1e3, bytes / (ms / 20 * 1e-3) / 1e12);
    }
  }
  return 0;
}
