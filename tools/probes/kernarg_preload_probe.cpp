// Does kernel-argument preloading (hipcc -mllvm -amdgpu-kernarg-preload-count=N: the command processor writes the first N dwords of the
// kernarg segment into SGPRs at wave launch) shorten a dependent small launch?  A HIP graph of CHAIN dependent launches of a kernel that
// needs a pointer argument before it can issue its first load; between replays a 512 MB fill evicts L2 / Infinity Cache so that the
// kernarg blocks are cold, as they are in a decode step that streams 100 MB of weights per layer.
//   hipcc --offload-arch=gfx950 -O3 tools/probes/kernarg_preload_probe.cpp -o build/tools/kp0
//   hipcc --offload-arch=gfx950 -O3 -mllvm -amdgpu-kernarg-preload-count=16 tools/probes/kernarg_preload_probe.cpp -o build/tools/kp1
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
__global__ void step(const int *in, int *out, const int *tab, int n, int k) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) out[i] = in[i] + tab[(i + k) & 1023];
}
int main() {
  const int n = 16 * 256, CHAIN = 64;
  int *a, *b, *tab; char *big;
  CK(hipMalloc(&a, n * 4)); CK(hipMalloc(&b, n * 4)); CK(hipMalloc(&tab, 4096)); CK(hipMalloc(&big, 512u << 20));
  CK(hipMemset(a, 0, n * 4)); CK(hipMemset(tab, 0, 4096));
  hipStream_t s; CK(hipStreamCreate(&s));
  hipGraph_t g; hipGraphExec_t ge;
  CK(hipStreamBeginCapture(s, hipStreamCaptureModeGlobal));
  for (int i = 0; i < CHAIN; ++i) {
    hipLaunchKernelGGL(step, dim3(16), dim3(256), 0, s, (const int *)(i & 1 ? b : a), (i & 1 ? a : b), (const int *)tab, n, i);
  }
  CK(hipStreamEndCapture(s, &g)); CK(hipGraphInstantiate(&ge, g, nullptr, nullptr, 0));
  hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  for (int cold = 0; cold < 2; ++cold) {
    float best = 1e9f, sum = 0;
    for (int r = 0; r < 12; ++r) {
      if (cold) CK(hipMemsetAsync(big, r, 512u << 20, s));
      CK(hipEventRecord(e0, s)); CK(hipGraphLaunch(ge, s)); CK(hipEventRecord(e1, s)); CK(hipStreamSynchronize(s));
      float ms; CK(hipEventElapsedTime(&ms, e0, e1));
      if (r >= 2) { best = ms < best ? ms : best; sum += ms; }
    }
    printf("%s: %.2f us per dependent launch (best), %.2f (mean of 10)\n", cold ? "caches evicted before each replay" : "back to back replays", best * 1e3f / CHAIN, sum * 1e2f / CHAIN);
  }
  return 0;
}
