"""CPU-only checks of the drop-in boundary: the shared library loads and exports every symbol include/atom_hip.h
declares (no compute calls without a GPU), and the host-side helpers agree with the oracle."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "atom_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(atom_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from atom_amd import _lib
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    L = _lib.lib()
    decl = _declared_symbols()
    assert len(decl) >= 9
    for name in decl:
        assert hasattr(L, name), f"{name} declared in include/atom_hip.h but not exported"
        assert name in _lib.SIGNATURES, f"{name} has no ctypes signature"
    assert set(_lib.SIGNATURES) == set(decl)
    assert b"gfx950" in L.atom_version()
    assert L.atom_strerror(0) == b"ok"


def test_scale_size_matches_reference_formula():
    from atom_amd import _lib, ops
    from oracle import atom_oracle as O
    L = _lib.lib()
    for m in list(range(0, 70)) + [100, 127, 128, 4095, 4096, 65536]:
        assert L.atom_scale_size(m, 0) == ops.scale_size(m) == O.scale_size(m)
        assert L.atom_scale_size(m, 1) == m


def test_no_cpu_fallback():
    """Product ops refuse CPU tensors instead of silently computing on the host."""
    import torch
    from atom_amd import ops
    from atom_amd._lib import AtomHipError
    with pytest.raises(AtomHipError):
        ops.reorder_fp16_i4(torch.zeros((2, 256), dtype=torch.float16), None)
    with pytest.raises(AtomHipError):
        ops.quant_weight_w4(torch.zeros((4, 256), dtype=torch.float16))


def test_product_does_not_import_oracle():
    for dirpath, _, files in os.walk(os.path.join(ROOT, "atom_amd")):
        for f in files:
            if f.endswith(".py"):
                txt = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in re.sub(r'""".*?"""', "", txt, flags=re.S).replace("# ", ""), f


def test_f6_summation_order_query():
    """atom_gemm_w4a4_f6_order is a host-side function of the shape alone: K steps in order for shapes that fill the chip with
    256x256 / 128x128 tiles and -- round 5 -- for up to 512 tiles of 64x64 (the mid-size-batch kernel), two ranges beyond that up to 256
    tiles of 128x128; 0 for an unsupported shape.  (What the GPU tests restate the arithmetic for: oracle gemm_core nsplit = -order.)"""
    from atom_amd import _lib
    order = _lib.lib().atom_gemm_w4a4_f6_order
    assert order(4096, 4096, 4096) == 1 and order(65536, 11008, 4096) == 1 and order(2048, 4096, 4096) == 1
    assert order(1024, 4096, 4096) == 2 and order(768, 4096, 11008) == 2 and order(256, 13824, 5120) == 2
    assert order(300, 1088, 1152) == 1 and order(512, 4096, 4096) == 1 and order(256, 4096, 11008) == 1 and order(100, 4096, 4096) == 1
    assert order(64, 13824, 5120) == 1 and order(513, 4096, 4096) == 2     # 512 / 576 tiles of 64x64
    assert order(1500, 1408, 640) == 1                        # fewer than 8 K steps: no K groups
    assert order(16, 4096, 4000) == 0 and order(0, 4096, 4096) == 0 and order(16, 32, 4096) == 0


def test_packed_route_query():
    """atom_gemm_w4a4_ws_recodes / atom_gemm_w4a4_workspace_bytes are host-side functions of the shape: packed operands are re-coded to
    the F6 format in the caller's workspace from 257 rows (N >= 2048, K >= 1024), from 129 where the decode-batch kernel does not
    take the shape; the workspace holds the weight's records + fp32 scales first, then the activation's records."""
    from atom_amd import _lib
    L = _lib.lib()
    rec, wsb = L.atom_gemm_w4a4_ws_recodes, L.atom_gemm_w4a4_workspace_bytes
    assert rec(4096, 4096, 4096) == 1 and rec(768, 4096, 4096) == 1 and rec(257, 4096, 4096) == 1 and rec(300, 11008, 4096) == 1
    assert rec(256, 4096, 4096) == 0 and rec(16, 4096, 4096) == 0 and rec(128, 11008, 4096) == 0
    assert rec(256, 11008, 4096) == 1                         # 129..256 rows outside the decode-batch kernel's shapes
    assert rec(4096, 1024, 4096) == 0 and rec(4096, 4096, 896) == 0
    G = (4096 - 128) // 128
    assert wsb(300, 4096, 4096) == G * 512 * 104 + G * 4096 * 108
    assert rec(0, 4096, 4096) == 0 and rec(300, 4096, 4000) == 0
    # with the weight's BF6 form cached (ATOM_WS_WEIGHT_CACHED): every shape from 129 rows, and from 17 rows what the decode-batch kernel
    # does not take
    recc = L.atom_gemm_w4a4_ws_recodes_cached
    assert recc(4096, 4096, 4096) == 1 and recc(256, 4096, 4096) == 1 and recc(129, 4096, 4096) == 1
    assert recc(64, 13824, 5120) == 1 and recc(64, 5120, 13824) == 1 and recc(100, 8192, 8192) == 1
    assert recc(64, 4096, 4096) == 0 and recc(128, 11008, 4096) == 0 and recc(16, 13824, 5120) == 0 and recc(64, 1024, 13824) == 0
    assert wsb(64, 13824, 5120) > 0 and L.atom_gemm_w4a4_packed_order(64, 13824, 5120, 2) == 1
    assert L.atom_gemm_w4a4_packed_order(64, 5120, 13824, 1) > 100 and wsb(64, 5120, 13824) >= 8 * 64 * 5120 * 4   # split K without the flag
    # 8 .. 16 rows at K_total > 14464 (Llama-70B down_proj): split K through the workspace WITHOUT the flag, the plain entry point's
    # kernel with it -- the partial sums would land on the cached weight otherwise (tests/test_gpu_gemm.py)
    assert L.atom_gemm_w4a4_packed_order(12, 8192, 28672, 1) > 100 and L.atom_gemm_w4a4_packed_order(12, 8192, 28672, 2) == L.atom_gemm_w4a4_packed_order(12, 8192, 28672, 0)


def test_dispatch_queries_are_consistent_over_random_shapes():
    """The host-side dispatch queries agree with each other for any shape: a shape that re-codes without a cached weight re-codes with
    one; a re-coding shape has room for both operands' BF6 forms in the workspace the library asks for; every summation order is one
    the oracle restates (1, 2, 8, 63, 64, 100 + splits); unsupported shapes answer 0 everywhere."""
    import random
    from atom_amd import _lib
    L = _lib.lib()
    rnd = random.Random(5)
    for _ in range(4000):
        M = rnd.choice([1, 2, 3, 7, 16, 17, 33, 64, 65, 128, 129, 200, 256, 257, 300, 512, 513, 700, 1024, 2048, 4096, rnd.randrange(1, 5000)])
        N = 64 * rnd.randrange(1, 260)
        K = 128 * rnd.randrange(2, 120)
        rec, recc, wsb = L.atom_gemm_w4a4_ws_recodes(M, N, K), L.atom_gemm_w4a4_ws_recodes_cached(M, N, K), L.atom_gemm_w4a4_workspace_bytes(M, N, K)
        assert rec in (0, 1) and recc in (0, 1) and recc >= rec, (M, N, K)
        G = K // 128 - 1
        f6 = G * (-(-M // 256) * 256) * 104 + G * (-(-N // 256) * 256) * 108
        if recc:
            assert wsb >= f6, (M, N, K, wsb, f6)
        for w in (0, 1, 2):
            o = L.atom_gemm_w4a4_packed_order(M, N, K, w)
            assert o in (1, 2, 8, 63, 64) or 101 < o <= 108, (M, N, K, w, o)
            # a call that asserts ATOM_WS_WEIGHT_CACHED never splits K through the workspace: its head is the cached weight
            assert not (w == 2 and o > 100), (M, N, K, o)
        if recc:
            assert L.atom_gemm_w4a4_packed_order(M, N, K, 2) == L.atom_gemm_w4a4_f6_order(M, N, K) in (1, 2), (M, N, K)
        assert L.atom_gemm_w4a4_f6_order(M, N, K) in (1, 2)
    for bad in ((0, 4096, 4096), (16, 4096, 4000), (16, 32, 4096), (16, 4096, 128)):
        assert L.atom_gemm_w4a4_ws_recodes_cached(*bad) == 0 and L.atom_gemm_w4a4_workspace_bytes(*bad) == 0 and L.atom_gemm_w4a4_f6_order(*bad) == 0


def test_fused_quantiser_shape_query():
    """atom_gemm_w4a4_multi_q_fits(q_op, ...): one or two tokens, the shapes atom_gemm_w4a4_multi takes, and the launcher's own bounds.
    Round 6 (csrc/gemvq_w4a4.hip: 1024 threads per workgroup): per thread at most three 4-channel quantiser tasks per token row
    (K_total <= 12,288) and, for the three ops that stage fp16 rows and the norm weight in LDS, two 16-byte row chunks (M x K_total <=
    16,384); at most 8 weight chunks per lane.  The merge form (atom_gemm_w4a4_multi_merge_q_fits): 2 .. 16 KV splits, M x K_total <= 8192."""
    from atom_amd import _lib
    L = _lib.lib()
    fq, fm, fmq = L.atom_gemm_w4a4_multi_q_fits, L.atom_gemm_w4a4_multi_fits, L.atom_gemm_w4a4_multi_merge_q_fits
    RE, RN, AR, SM = _lib.Q_REORDER, _lib.Q_RMSNORM, _lib.Q_ADD_RMSNORM, _lib.Q_SILU_MUL
    assert fq(RN, 1, 4096, 3, 4096) == 1 and fq(RE, 2, 4096, 1, 4096) == 1 and fq(AR, 1, 11008, 2, 4096) == 1 and fq(SM, 2, 4096, 1, 11008) == 1
    assert fq(RE, 3, 4096, 1, 4096) == 0 and fq(RE, 0, 4096, 1, 4096) == 0 and fm(3, 4096, 1, 4096) == 1
    assert fq(0, 1, 4096, 1, 4096) == 0 and fq(5, 1, 4096, 1, 4096) == 0
    # 13824 / 4 tasks per row > 3 x 1024: not the dot-product form; one token still fits the decode-batch form (864 slots <= 3 x 512)
    assert fq(SM, 2, 5120, 1, 13824) == 0 and fq(SM, 1, 5120, 1, 13824) == 1
    assert fq(SM, 1, 5120, 1, 12288) == 1 and fq(SM, 2, 5120, 1, 12288) == 1
    assert fq(RE, 1, 4096, 4, 4096) == 0 and fq(RE, 1, 24, 1, 4096) == 0 and fq(RE, 1, 4096, 1, 4000) == 0
    # two tokens: the row-staging ops up to hidden 8192 (M x K / 8 <= 2048 chunks), SiLU x up beyond
    for h in (6656, 8192, 11008):
        ok = 1 if h <= 8192 else 0
        assert fq(RE, 2, 4096, 1, h) == ok and fq(RN, 2, 4096, 1, h) == ok and fq(AR, 2, 4096, 1, h) == ok and fq(SM, 2, 4096, 1, h) == 1
        assert fq(RE, 1, 4096, 1, h) == 1
    assert fq(RE, 2, 4096, 1, 8192) == 1 and fq(RE, 2, 4096, 1, 8320) == 0
    assert fmq(1, 4096, 1, 4096, 8) == 1 and fmq(1, 4096, 1, 4096, 16) == 1 and fmq(1, 8192, 1, 8192, 2) == 1
    assert fmq(2, 4096, 1, 4096, 4) == 0                     # two tokens with K <= 4096 run the decode-batch kernel, not the dot-product form
    assert fmq(1, 4096, 1, 4096, 1) == 0 and fmq(1, 4096, 1, 4096, 17) == 0 and fmq(2, 4096, 1, 8192, 8) == 0 and fmq(3, 4096, 1, 4096, 8) == 0


def test_decode_split_policy_query():
    """atom_batch_decode_i4_splits / _workspace_bytes (host-side functions of the shape): how many waves share a (sequence, head)'s KV range.
    Round 6: splits of at least 4 tiles of 16 tokens; one round of at most 3072 waves where the pairs allow it.  Llama-7B heads,
    context 1024 = 64 pages of 16: 16 waves per pair at batch 1-4 that leave 4 partial states (4 waves per workgroup merge in LDS), 12 at 8, 6 at 16 (the splits are then waves of 12-wave workgroups that merge
    in LDS), 3 from 32 on (two rounds of a third of the range beat one round of all of it); the workspace holds batch x heads x splits partial states of 130 floats."""
    from atom_amd import _lib
    L = _lib.lib()
    sp = lambda b, pages, heads=32, page=16: L.atom_batch_decode_i4_splits(b, heads, page, pages)
    assert [sp(b, 64) for b in (1, 2, 4, 8, 16, 32, 64, 128)] == [4, 4, 4, 12, 6, 3, 3, 3]
    assert sp(1, 0) == 1 and sp(1, 4) == 1 and sp(1, 8) == 2 and sp(1, 16) == 1 and sp(1, 256) <= 64   # unknown / short / long contexts (16 pages: 4 waves, one workgroup, no partial state)
    assert 4 <= sp(1, 64, page=32) <= 32                                                 # 32-token pages: 128 tiles
    for b in (1, 8, 16, 64):
        s_ = sp(b, 64)
        assert L.atom_batch_decode_i4_workspace_bytes(b, 32, 16, 64) == (b * 32 * s_ * 130 * 4 if s_ > 1 else 0)
    assert sp(0, 64) == 0 and L.atom_batch_decode_i4_workspace_bytes(0, 32, 16, 64) == 0


def test_bf6_convert_result_never_overlaps_its_sources_at_an_offset(tmp_path):
    """Guard against a code-generation trap of hipcc (ROCm 7.2): v_cvt_scalef32_2xpk16_bf6_f32 reads two 16-register sources over
    several passes and writes a 6-register result; the register allocator may place the result INSIDE a source at an offset
    (v[2:7] <- v[0:15]) and the instruction then overwrites elements it has not read yet (wrong BF6 codes, silently).  Result
    and source at the same base register are fine.  Disassemble every gfx950 code object of the product library and check."""
    import re
    import shutil
    import subprocess
    from atom_amd import _lib
    objdump = "/opt/rocm/lib/llvm/bin/llvm-objdump"
    if not os.path.exists(objdump):
        pytest.skip("llvm-objdump not found")
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    objs = sorted(p for p in tmp_path.iterdir() if "gfx950" in p.name)
    assert objs, "no gfx950 code objects found in the library"
    pat = re.compile(r"v_cvt_scalef32_2xpk16_bf6_f32 v\[(\d+):(\d+)\], v\[(\d+):(\d+)\], v\[(\d+):(\d+)\]")
    seen = 0
    for o in objs:
        asm = subprocess.run([objdump, "-d", str(o)], check=True, capture_output=True, text=True).stdout
        for m in pat.finditer(asm):
            d0, d1, a0, a1, b0, b1 = map(int, m.groups())
            seen += 1
            for s0, s1 in ((a0, a1), (b0, b1)):
                overlap = not (d1 < s0 or s1 < d0)
                assert not overlap or d0 == s0, f"{m.group(0)} in {o.name}: result overlaps a source at an offset"
    assert seen >= 100          # the quantisers, the re-coding kernel and the fused gate/up epilogue all use it


def test_no_kernel_keeps_its_tile_in_scratch_memory(tmp_path):
    """hipcc may demote a register array to scratch (private) memory without reporting a spill -- round 3 met it on the two-K-group GEMM
    kernel: 134 VGPRs + 272 B of scratch per lane, 8x slower, vgpr_spill_count 0.  The code objects carry the size
    (.private_segment_fixed_size): none for the tile kernels the dispatch uses, at most a few spilled registers anywhere."""
    import re
    import shutil
    import subprocess
    from atom_amd import _lib
    objdump, readelf = "/opt/rocm/lib/llvm/bin/llvm-objdump", "/opt/rocm/lib/llvm/bin/llvm-readelf"
    if not (os.path.exists(objdump) and os.path.exists(readelf)):
        pytest.skip("llvm-objdump / llvm-readelf not found")
    if not os.path.exists(_lib.LIB_PATH):
        _lib.build()
    so = shutil.copy(_lib.LIB_PATH, tmp_path / "lib.so")
    subprocess.run([objdump, "--offloading", str(so)], cwd=tmp_path, check=True, capture_output=True)
    sizes = {}
    for o in sorted(p for p in tmp_path.iterdir() if "gfx950" in p.name):
        notes = subprocess.run([readelf, "--notes", str(o)], check=True, capture_output=True, text=True).stdout
        for name, size in re.findall(r"\.name:\s+(\S+)\n\s+\.private_segment_fixed_size:\s+(\d+)", notes):
            sizes[name] = int(size)
    assert len(sizes) > 100
    for name, size in sizes.items():
        assert size <= 64, f"{name}: {size} B of scratch per lane"
        if any(k in name for k in ("gemm_w4a4_f6q_kernel", "gemm_w4a4_f6qk_kernel", "gemm_w4a4_f6q2_kernel", "gemm_w4a4_f6x16_kernel",
                                   "gemv_w4a4", "act_quant2_kernel", "batch_decode_kernel")):
            assert size == 0, f"{name}: {size} B of scratch per lane"
