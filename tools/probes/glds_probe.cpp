// Probe: what LDS image does global_load_lds_{ushort,dword,dwordx4} produce on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef const __attribute__((address_space(1))) void *gptr_t;
typedef __attribute__((address_space(3))) void *lptr_t;
__global__ void k(const uint16_t *src, uint32_t *out) {
  extern __shared__ __attribute__((aligned(16))) char lds[];
  const int lane = threadIdx.x;
  for (int i = lane; i < 1024; i += 64) ((uint32_t *)lds)[i] = 0xDEADBEEF;
  __syncthreads();
  __builtin_amdgcn_global_load_lds((gptr_t)(src + lane), (lptr_t)(lds), 2, 0, 0);          // ushort
  __builtin_amdgcn_global_load_lds((gptr_t)(src + 2 * lane), (lptr_t)(lds + 1024), 4, 0, 0); // dword
  __builtin_amdgcn_global_load_lds((gptr_t)(src + 8 * lane), (lptr_t)(lds + 2048), 16, 0, 0); // x4
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  __syncthreads();
  for (int i = lane; i < 1024; i += 64) out[i] = ((uint32_t *)lds)[i];
}
int main() {
  uint16_t h[1024]; for (int i = 0; i < 1024; ++i) h[i] = (uint16_t)(0x1000 + i);
  uint16_t *d; uint32_t *o; hipMalloc(&d, sizeof(h)); hipMalloc(&o, 4096);
  hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(1), dim3(64), 4096, 0, d, o);
  uint32_t r[1024]; hipMemcpy(r, o, 4096, hipMemcpyDeviceToHost);
  printf("ushort region dwords 0..7:"); for (int i = 0; i < 8; ++i) printf(" %08x", r[i]); printf("\n");
  printf("ushort region dwords 30..35:"); for (int i = 30; i < 36; ++i) printf(" %08x", r[i]); printf("\n");
  printf("ushort region dwords 62..66:"); for (int i = 62; i < 67; ++i) printf(" %08x", r[i]); printf("\n");
  printf("dword region 0..3:"); for (int i = 256; i < 260; ++i) printf(" %08x", r[i]); printf("\n");
  printf("x4 region 0..7:"); for (int i = 512; i < 520; ++i) printf(" %08x", r[i]); printf("\n");
  return 0;
}
