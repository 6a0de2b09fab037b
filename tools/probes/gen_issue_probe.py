#!/usr/bin/env python3
"""Generates build/tools/gen/issue_probe.cpp (not committed: `make -C atom_amd/csrc probes`): MFMA / VALU co-issue probe for gfx950 (MI355X).

Question: does VALU work hide under the MFMA of a SIMD, and under which stream shapes?  Every kernel
is ONE hand-written instruction stream (inline asm on hard-coded registers v64..v127, so the order is exactly what is
written and 4 waves fit a SIMD), run on all 256 CUs with 1..4 waves per SIMD (one 256..1024-thread workgroup per CU).
Reported: shader cycles (s_memtime) and ns per MFMA per SIMD.

Register map: A fragment v[64:69], B fragment v[70:75], E8M0 scale v76 (=127), v77 = s1 (float), v78 = s2 (float),
accumulators v[80:83] v[84:87] v[88:91] v[92:95] (16x16 results) or v[80:95] v[96:111] (32x32 results),
v112..v119 independent VALU targets, v120..v127 running sums.
"""
import sys

CLOB = ", ".join('"v%d"' % i for i in range(64, 128))

MF = {   # name -> (asm with {d} = dst regs, {c} = C operand), result regs, cycles at peak
    "bf6s16": ("v_mfma_scale_f32_16x16x128_f8f6f4 {d}, v[64:69], v[70:75], {c}, v76, v76 op_sel_hi:[0,0,0] cbsz:3 blgp:3", 4),
    "bf6u16": ("v_mfma_f32_16x16x128_f8f6f4 {d}, v[64:69], v[70:75], {c} cbsz:3 blgp:3", 4),
    "i8_16": ("v_mfma_i32_16x16x64_i8 {d}, v[64:67], v[70:73], {c}", 4),
    "bf6s32": ("v_mfma_scale_f32_32x32x64_f8f6f4 {d}, v[64:69], v[70:75], {c}, v76, v76 op_sel_hi:[0,0,0] cbsz:3 blgp:3", 16),
    "bf16_32": ("v_mfma_f32_32x32x16_bf16 {d}, v[64:67], v[70:73], {c}", 16),
}


def acc(kind, i):
    n = MF[kind][1]
    if n == 4:
        b = 80 + 4 * (i % 4)
        return b, "v[%d:%d]" % (b, b + 3)
    b = 80 + 16 * (i % 2)
    return b, "v[%d:%d]" % (b, b + 15)


def M(kind, i, c="0"):
    b, d = acc(kind, i)
    return MF[kind][0].format(d=d, c=(d if c == "acc" else c))


def I(n, start=0):   # n independent fma on v112..v119
    return ["v_fma_f32 v%d, v%d, v77, v78" % (112 + (start + j) % 8, 112 + (start + j) % 8) for j in range(n)]


def I_pk(n):   # n independent packed fma on the pairs v[112:113] .. v[118:119]
    return ["v_pk_fma_f32 v[%d:%d], v[%d:%d], v[78:79], v[120:121]" % (112 + 2 * (j % 4), 113 + 2 * (j % 4), 112 + 2 * (j % 4), 113 + 2 * (j % 4))
            for j in range(n)]


def deq(kind, i, mix=False, pk=False):   # de-quantisation of accumulator i: n multiplies, then n fma into v120.. (round-robin)
    b, _ = acc(kind, i)
    n = MF[kind][1]
    if pk:   # packed FP32: two elements per instruction; the token scale v77 is broadcast to both halves by op_sel_hi
        out = ["v_pk_mul_f32 v[%d:%d], v[%d:%d], v[76:77] op_sel:[0,1] op_sel_hi:[1,1]" % (b + r, b + r + 1, b + r, b + r + 1) for r in range(0, n, 2)]
        for r in range(0, n, 2):
            s_ = 120 + (r % 8)
            out.append("v_pk_fma_f32 v[%d:%d], v[%d:%d], v[78:79], v[%d:%d]" % (s_, s_ + 1, b + r, b + r + 1, s_, s_ + 1))
        return out
    out = ["v_mul_f32 v%d, v%d, v77" % (b + r, b + r) for r in range(n)]
    for r in range(n):
        s = 120 + (r % 8)
        if mix:
            out.append("v_fma_mix_f32 v%d, v%d, v78, v%d op_sel_hi:[0,1,0]" % (s, b + r, s))
        else:
            out.append("v_fma_f32 v%d, v%d, v78, v%d" % (s, b + r, s))
    return out


def patterns(kind):
    n = MF[kind][1]
    P = {}
    if n == 4:
        P["M_only"] = ([M(kind, i) for i in range(8)], 8, 0)
        for v in (2, 4, 6, 8, 12):
            body = []
            for i in range(8):
                body += [M(kind, i)] + I(v)
            P["M+%dI" % v] = (body, 8, 8 * v)
        body = []
        for i in range(4):
            body += [M(kind, 2 * i), M(kind, 2 * i + 1)] + I(16)
        P["2M+16I"] = (body, 8, 64)
        body = []
        for i in range(2):
            body += [M(kind, 4 * i + j) for j in range(4)] + I(32)
        P["4M+32I"] = (body, 8, 64)
        # the GEMM's dependency structure: tile t's 4 multiplies + 4 fma read the MFMA issued one unit earlier
        body = []
        for i in range(8):
            body += [M(kind, i + 1)] + deq(kind, i)
        P["M(t+1),deq(t)"] = (body, 8, 64)
        body = []
        for i in range(8):
            body += [M(kind, i + 1)] + deq(kind, i, mix=True)
        P["M(t+1),deq_mix(t)"] = (body, 8, 64)
        body = []
        for i in range(8):   # the same with packed FP32 de-quantisation: 2 v_pk_mul_f32 + 2 v_pk_fma_f32 per tile
            body += [M(kind, i + 1)] + deq(kind, i, pk=True)
        P["M(t+1),deq_pk(t)"] = (body, 8, 32)
        for la in (2, 3):
            body = []
            for i in range(8):
                body += [M(kind, i + la)] + deq(kind, i, pk=True)
            P["M(t+%d),deq_pk(t)" % la] = (body, 8, 32)
        body = []
        for i in range(4):
            body += [M(kind, 2 * i + 2), M(kind, 2 * i + 3)]
            d0, d1 = deq(kind, 2 * i, pk=True), deq(kind, 2 * i + 1, pk=True)
            body += d0[:2] + d1[:2] + d0[2:] + d1[2:]
        P["2M(p+1),deq_pk(p)"] = (body, 8, 32)
        body = []
        for i in range(4):   # pairs, one pair ahead (the r01 product kernel): accs {0,1} / {2,3}
            body += [M(kind, 2 * i + 2), M(kind, 2 * i + 3)]
            d0, d1 = deq(kind, 2 * i), deq(kind, 2 * i + 1)
            body += d0[:4] + d1[:4] + d0[4:] + d1[4:]
        P["2M(p+1),deq(p)"] = (body, 8, 64)
        body = []
        for i in range(8):   # no look-ahead: the dependent VALU needs the MFMA->VALU wait states in software
            body += [M(kind, i), "s_nop 7"] + deq(kind, i)
        P["M(t),nop8,deq(t)"] = (body, 8, 64)
        body = []
        for i in range(8):   # look-ahead 2
            body += [M(kind, i + 2)] + deq(kind, i)
        P["M(t+2),deq(t)"] = (body, 8, 64)
    else:
        zc = "acc" if kind == "bf16_32" else "0"
        P["M_only"] = ([M(kind, i, zc) for i in range(4)], 4, 0)
        for v in (3, 5, 8, 16):
            body = []
            for i in range(4):
                body += [M(kind, i, zc)] + I(v)
            P["M+%dI" % v] = (body, 4, 4 * v)
        if kind == "bf6s32":
            body = []
            for i in range(2):   # the 32x32x64 GEMM kernel: (C=0, accumulate) chain of tile t+1, then tile t's 16 + 16
                body += [M(kind, i + 1, "0"), M(kind, i + 1, "acc")] + deq(kind, i)
            P["2Mchain(t+1),deq(t)"] = (body, 4, 64)
    P["VALU_only"] = (I(64), 0, 64)
    P["PK_only"] = (I_pk(64), 0, 64)
    return P


def main():
    out = []
    out.append("// GENERATED by tools/probes/gen_issue_probe.py -- do not edit.  Build: hipcc --offload-arch=gfx950 -O2 issue_probe.cpp")
    out.append("#include <hip/hip_runtime.h>\n#include <cstdio>\n#include <cstring>")
    out.append("#define CLOB %s" % CLOB)
    out.append(r'''
#define INIT_ASM \
  "v_mov_b32 v64, 0x0c30c30c\n v_mov_b32 v65, 0x30c30c30\n v_mov_b32 v66, 0xc30c30c3\n v_mov_b32 v67, 0x0c30c30c\n" \
  "v_mov_b32 v68, 0x30c30c30\n v_mov_b32 v69, 0xc30c30c3\n v_mov_b32 v70, 0x0c30c30c\n v_mov_b32 v71, 0x30c30c30\n" \
  "v_mov_b32 v72, 0xc30c30c3\n v_mov_b32 v73, 0x0c30c30c\n v_mov_b32 v74, 0x30c30c30\n v_mov_b32 v75, 0xc30c30c3\n" \
  "v_mov_b32 v76, 127\n v_mov_b32 v77, 0x3f800347\n v_mov_b32 v78, 0x3f000000\n v_mov_b32 v79, 0x3f000000\n" \
  "v_mov_b32 v80, 0\n v_mov_b32 v81, 0\n v_mov_b32 v82, 0\n v_mov_b32 v83, 0\n v_mov_b32 v84, 0\n v_mov_b32 v85, 0\n v_mov_b32 v86, 0\n v_mov_b32 v87, 0\n" \
  "v_mov_b32 v88, 0\n v_mov_b32 v89, 0\n v_mov_b32 v90, 0\n v_mov_b32 v91, 0\n v_mov_b32 v92, 0\n v_mov_b32 v93, 0\n v_mov_b32 v94, 0\n v_mov_b32 v95, 0\n" \
  "v_mov_b32 v96, 0\n v_mov_b32 v97, 0\n v_mov_b32 v98, 0\n v_mov_b32 v99, 0\n v_mov_b32 v100, 0\n v_mov_b32 v101, 0\n v_mov_b32 v102, 0\n v_mov_b32 v103, 0\n" \
  "v_mov_b32 v104, 0\n v_mov_b32 v105, 0\n v_mov_b32 v106, 0\n v_mov_b32 v107, 0\n v_mov_b32 v108, 0\n v_mov_b32 v109, 0\n v_mov_b32 v110, 0\n v_mov_b32 v111, 0\n" \
  "v_mov_b32 v112, 1.0\n v_mov_b32 v113, 2.0\n v_mov_b32 v114, 0.5\n v_mov_b32 v115, 4.0\n v_mov_b32 v116, 1.0\n v_mov_b32 v117, 2.0\n v_mov_b32 v118, 0.5\n v_mov_b32 v119, 4.0\n" \
  "v_mov_b32 v120, 0\n v_mov_b32 v121, 0\n v_mov_b32 v122, 0\n v_mov_b32 v123, 0\n v_mov_b32 v124, 0\n v_mov_b32 v125, 0\n v_mov_b32 v126, 0\n v_mov_b32 v127, 0\n"
#define FINI_ASM "v_add_f32 %0, v80, v120\n v_add_f32 %0, %0, v112\n v_add_f32 %0, %0, v84\n v_add_f32 %0, %0, v96\n v_add_f32 %0, %0, v121\n"
extern __shared__ char dyn_lds[];
#define KERNEL(NAME, BODY)                                                                          \
  __global__ __launch_bounds__(1024) void NAME(float *out, int iters, unsigned long long *cyc) {   \
    asm volatile(INIT_ASM ::: CLOB);                                                                \
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();                                     \
    for (int i = 0; i < iters; ++i) asm volatile(BODY ::: CLOB);                                    \
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();                                     \
    float r;                                                                                        \
    asm volatile(FINI_ASM : "=v"(r) :: CLOB);                                                       \
    if (blockIdx.x == 9 && threadIdx.x == 0) cyc[0] = t1 - t0;                                      \
    if (r == 123.456f) dyn_lds[threadIdx.x] = 1;                                                    \
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;                                                 \
  }
''')
    table = []
    kid = 0
    for kind in MF:
        for name, (body, nm, nv) in patterns(kind).items():
            if name in ("VALU_only", "PK_only") and kind != "bf6s16":
                continue
            fn = "k%d" % kid
            kid += 1
            asm = "\\n\"\n    \"".join(body)
            out.append('KERNEL(%s,\n    "%s\\n")' % (fn, asm))
            table.append((fn, kind, name, nm, nv))
    # two roles on one SIMD: even waves run MFMA only, odd waves VALU only (wave id = threadIdx.x / 64; waves w and w+4 share a SIMD)
    mbody = "\\n\"\n    \"".join([M("bf6s16", i) for i in range(8)])
    vbody = "\\n\"\n    \"".join(I(64))
    out.append(r'''
__global__ __launch_bounds__(1024) void k_roles(float *out, int iters, unsigned long long *cyc) {
  asm volatile(INIT_ASM ::: CLOB);
  const int role = (threadIdx.x >> 8) & 1;     // waves 0-3: MFMA only; 4-7: VALU only; 8-11: MFMA; 12-15: VALU
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
  if (role == 0) { for (int i = 0; i < iters; ++i) asm volatile("%s\n" ::: CLOB); }
  else           { for (int i = 0; i < iters; ++i) asm volatile("%s\n" ::: CLOB); }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float r;
  asm volatile(FINI_ASM : "=v"(r) :: CLOB);
  if (blockIdx.x == 9 && (threadIdx.x == 0 || threadIdx.x == 256)) cyc[threadIdx.x >> 8] = t1 - t0;
  if (r == 123.456f) dyn_lds[threadIdx.x] = 1;
  out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}
''' % (mbody, vbody))
    out.append("struct Row { void (*fn)(float *, int, unsigned long long *); const char *kind, *name; int nm, nv; };")
    out.append("static Row rows[] = {")
    for fn, kind, name, nm, nv in table:
        out.append('  {%s, "%s", "%s", %d, %d},' % (fn, kind, name, nm, nv))
    out.append("};")
    out.append(r'''
int main(int argc, char **argv) {
  const char *only = argc > 1 ? argv[1] : nullptr;
  float *out; hipMalloc(&out, 256 * 1024 * 4);
  unsigned long long *cyc; hipMalloc(&cyc, 16);
  hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
  const int iters = 8000;
  const size_t lds = 96 * 1024;   // one workgroup per CU
  printf("# kind pattern waves/SIMD | ns/iter cyc/iter GHz | per SIMD: cyc/MFMA ns/MFMA | cyc/VALU\n");
  for (auto &r : rows) {
    if (only && strcmp(only, r.kind)) continue;
    hipFuncSetAttribute((const void *)r.fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    for (int w = 1; w <= 4; ++w) {
      hipLaunchKernelGGL(r.fn, dim3(256), dim3(256 * w), lds, 0, out, 1000, cyc);
      hipEventRecord(e0);
      hipLaunchKernelGGL(r.fn, dim3(256), dim3(256 * w), lds, 0, out, iters, cyc);
      hipEventRecord(e1); hipEventSynchronize(e1);
      float ms; hipEventElapsedTime(&ms, e0, e1);
      unsigned long long hc; hipMemcpy(&hc, cyc, 8, hipMemcpyDeviceToHost);
      const double ns = ms * 1e6 / iters, cy = (double)hc / iters;
      printf("%-8s %-20s w=%d | %8.1f %8.1f %.2f |", r.kind, r.name, w, ns, cy, cy / ns);
      if (r.nm) printf(" %6.1f %6.2f |", cy / (r.nm * w), ns / (r.nm * w)); else printf("      -      - |");
      if (r.nv) printf(" %5.2f", cy / (r.nv * w));
      printf("\n");
    }
  }
  // roles: 2 and 4 waves per SIMD, half of them MFMA-only, half VALU-only
  hipFuncSetAttribute((const void *)k_roles, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  for (int w = 2; w <= 4; w += 2) {
    hipLaunchKernelGGL(k_roles, dim3(256), dim3(256 * w), lds, 0, out, 1000, cyc);
    hipEventRecord(e0);
    hipLaunchKernelGGL(k_roles, dim3(256), dim3(256 * w), lds, 0, out, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long hc[2]; hipMemcpy(hc, cyc, 16, hipMemcpyDeviceToHost);
    printf("roles (bf6s16 8 MFMA per iter | 64 fma per iter) w=%d: wall %.1f ns/iter; MFMA wave %.1f cyc/iter = %.1f cyc/MFMA/SIMD, VALU wave %.1f cyc/iter = %.2f cyc/VALU/SIMD\n",
           w, ms * 1e6 / iters, (double)hc[0] / iters, (double)hc[0] / iters / (8 * w / 2), (double)hc[1] / iters, (double)hc[1] / iters / (64 * w / 2));
  }
  return 0;
}
''')
    open(sys.argv[1] if len(sys.argv) > 1 else "build/tools/gen/issue_probe.cpp", "w").write("\n".join(out) + "\n")


if __name__ == "__main__":
    main()
