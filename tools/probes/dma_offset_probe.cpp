// Does the instruction offset of global_load_lds_dwordx4 move the LDS destination as well as the global source (gfx950)?
// And the saddr form (SGPR base + VGPR 32-bit offset).  hipcc --offload-arch=gfx950 -O2 tools/probes/dma_offset_probe.cpp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
extern __shared__ char lds[];
__global__ void k(const unsigned *src, unsigned *out) {
  for (int i = threadIdx.x; i < 4096; i += 64) reinterpret_cast<unsigned *>(lds)[i] = 0xDEAD0000u + i;
  __syncthreads();
  const unsigned m0v = 4096;                         // LDS byte offset 4096
  const char *g = reinterpret_cast<const char *>(src) + threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off offset:1024\n\ts_waitcnt vmcnt(0)" ::"s"(m0v), "v"(g) : "memory");
  // saddr form: SGPR base + VGPR offset, imm 2048, M0 = 8192
  const unsigned m1v = 8192;
  const unsigned voff = threadIdx.x * 16;
  asm volatile("s_mov_b32 m0, %0\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, %2 offset:2048\n\ts_waitcnt vmcnt(0)" ::"s"(m1v), "v"(voff), "s"(src) : "memory");
  __syncthreads();
  for (int i = threadIdx.x; i < 4096; i += 64) out[i] = reinterpret_cast<unsigned *>(lds)[i];
}
int main() {
  std::vector<unsigned> h(4096);
  for (int i = 0; i < 4096; ++i) h[i] = i;           // dword i holds i
  unsigned *src, *out; (void)hipMalloc(&src, 16384); (void)hipMalloc(&out, 16384);
  (void)hipMemcpy(src, h.data(), 16384, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 16384, 0, src, out);
  (void)hipMemcpy(h.data(), out, 16384, hipMemcpyDeviceToHost);
  int first = -1, last = -1;
  for (int i = 0; i < 4096; ++i) if ((h[i] >> 16) != 0xDEAD) { if (first < 0) first = i; last = i; }
  printf("changed dwords: first %d last %d\n", first, last);
  for (int i = 0; i < 4096; ++i) if ((h[i] >> 16) != 0xDEAD && (i % 256 == 0 || i == first)) printf("  lds dword %d (byte %d) = src dword %u\n", i, i * 4, h[i]);
  return 0;
}
