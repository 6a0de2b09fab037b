// Control for roofline.traffic: what rocprofv3's FETCH_SIZE / WRITE_SIZE report on gfx950 for streams of KNOWN size, by
// load type -- the GEMM kernels read their operands with LDS-DMA (global_load_lds_dwordx4), and the x2 correction of
// MI355X_MICROARCH.md was calibrated on 16-byte-per-lane loads into registers.  Three kernels, each touching `mb` MiB exactly once:
//   stream_lds_dma   global_load_lds_dwordx4, 1 KiB per wave instruction (the GEMMs' operand path)
//   stream_vgpr      global_load_dwordx4 into registers (the calibration case of the guide)
//   stream_store     global_store_dwordx4 (the GEMM's output path: 16-byte stores)
// Run each under `rocprofv3 --kernel-trace --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` (tools/profile_bench.sh) and divide the counter
// (KiB) by mb * 1024: that ratio is the correction for that access type.  Build: hipcc --offload-arch=gfx950 -O2 fetch_control.cpp
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e), __LINE__); return 1; } } while (0)
typedef unsigned v4u __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void stream_lds_dma(const uint8_t *src, size_t bytes, unsigned *sink) {
  __shared__ __attribute__((aligned(16))) char lds[4 * 1024];
  const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
  const unsigned m0v = __builtin_amdgcn_readfirstlane((unsigned)(size_t)(__attribute__((address_space(3))) char *)lds + wave * 1024);
  const size_t stride = (size_t)gridDim.x * 4096;
  for (size_t off = (size_t)blockIdx.x * 4096 + wave * 1024; off < bytes; off += stride) {
    const uint8_t *p = src + off + lane * 16;
    asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off" ::"s"(m0v), "v"(p) : "memory");
  }
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  if (sink && threadIdx.x == 0) sink[blockIdx.x] = *reinterpret_cast<unsigned *>(lds);
}
__global__ __launch_bounds__(256) void stream_vgpr(const uint8_t *src, size_t bytes, unsigned *sink) {
  const size_t stride = (size_t)gridDim.x * 4096;
  v4u acc = {0, 0, 0, 0};
  for (size_t off = (size_t)blockIdx.x * 4096 + threadIdx.x * 16; off < bytes; off += stride) acc ^= *reinterpret_cast<const v4u *>(src + off);
  if (sink && (acc.x ^ acc.y ^ acc.z ^ acc.w) == 0x12345u) sink[blockIdx.x] = 1;
}
__global__ __launch_bounds__(256) void stream_store(uint8_t *dst, size_t bytes) {
  const size_t stride = (size_t)gridDim.x * 4096;
  const v4u v = {blockIdx.x, threadIdx.x, 3u, 4u};
  for (size_t off = (size_t)blockIdx.x * 4096 + threadIdx.x * 16; off < bytes; off += stride) *reinterpret_cast<v4u *>(dst + off) = v;
}

int main(int argc, char **argv) {
  const size_t mb = argc > 1 ? (size_t)atoi(argv[1]) : 512, bytes = mb << 20;
  const int reps = argc > 2 ? atoi(argv[2]) : 5;
  uint8_t *buf; unsigned *sink;
  CK(hipMalloc(&buf, bytes)); CK(hipMalloc(&sink, 4096 * 4));
  CK(hipMemset(buf, 1, bytes));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int k = 0; k < 3; ++k) {
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
      CK(hipEventRecord(e0, 0));
      if (k == 0) hipLaunchKernelGGL(stream_lds_dma, dim3(2048), dim3(256), 0, 0, buf, bytes, sink);
      else if (k == 1) hipLaunchKernelGGL(stream_vgpr, dim3(2048), dim3(256), 0, 0, buf, bytes, sink);
      else hipLaunchKernelGGL(stream_store, dim3(2048), dim3(256), 0, 0, buf, bytes);
      CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1)); if (ms < best) best = ms;
    }
    printf("%-15s %zu MiB per launch, %d launches, best %.1f us = %.2f TB/s\n", k == 0 ? "stream_lds_dma" : (k == 1 ? "stream_vgpr" : "stream_store"),
           mb, reps, best * 1e3, bytes / (best * 1e-3) / 1e12);
  }
  return 0;
}
