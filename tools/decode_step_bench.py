"""One decode step of a Llama-7B decoder layer on the real-kernel operator surface (atom_amd.e2e.LlamaDecoderLayer, the
mirror of the reference's punica llama.py:247-292): W4A4 GEMMs, fused RMSNorm / reorder / SiLU*mul quantisers, INT4 paged KV
cache, RoPE-fused batch decode attention.  Reports the layer latency per batch size and where it goes (CUDA events around
the ops of atom_amd.ops).  Run on the GPU box:  python tools/decode_step_bench.py [context_len [batch,batch,...]]"""
import os
import sys
import types

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from atom_amd import ops  # noqa: E402
from atom_amd.e2e import LlamaDecoderLayer  # noqa: E402
from atom_amd.utils import BatchLenInfo, BatchedKvCacheInt4, KvCacheInt4, KvPoolInt4  # noqa: E402

TIMED = ["rmsnorm_fp16_i4", "add_rmsnorm_fp16_i4", "reorder_fp16_i4", "activate_fp16_i4", "dense_layer_gemm_i4_fp16",
         "dense_layer_gemm_i4_o4", "dense_layer_gemm_i4_f32", "append_kv_i4", "quant_append_kv_i4", "batch_decode_i4"]


def main(ctx=1024, batches=(1, 4, 8, 16, 32, 64, 128), hidden=4096, heads=32, inter=11008, iters=20):
    dev = torch.device("cuda")
    cfg = types.SimpleNamespace(hidden_size=hidden, num_attention_heads=heads, intermediate_size=inter, rms_norm_eps=1e-5,
                                rope_theta=1e4)
    layer = LlamaDecoderLayer(cfg, layer_idx=0).cuda()
    g = torch.Generator().manual_seed(1)
    for mod in layer.modules():
        if type(mod).__name__ == "LinearInt4":
            mod.load_fp16_weight((torch.randn(mod.out_features, mod.in_features, generator=g) * 0.05).half().cuda())
        elif type(mod).__name__ == "LlamaRMSNormInt4":
            mod.weight.data = (1 + 0.1 * torch.randn(mod.weight.shape, generator=g)).half().cuda()

    spans = {}
    orig = {n: getattr(ops, n) for n in TIMED if hasattr(ops, n)}

    def wrap(name, f):
        def inner(*a, **k):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            r = f(*a, **k)
            e1.record()
            spans.setdefault(name, []).append((e0, e1))
            return r
        return inner

    print(f"# Llama-7B decoder layer, one decode step, context {ctx} tokens per sequence, INT4 paged KV (page 16)")
    print(f"# {'batch':>5} {'eager us':>9} {'graph us':>9} | " + " ".join(f"{n.replace('dense_layer_', '').replace('_fp16_i4', ''):>14}" for n in orig))
    for bsz in batches:
        pool = KvPoolInt4(num_layers=1, num_heads=heads, head_dim=hidden // heads, capacity=bsz * (ctx // 16 + 2), block_len=16,
                          device=dev)
        pool.buf.random_(0, 255)                     # any codes, small finite (scale, zero) parameters: timing only
        pool.param.copy_(torch.rand(pool.param.shape, device=dev).mul_(0.05).add_(0.01).half())
        cs = [KvCacheInt4(pool, ctx) for _ in range(bsz)]
        for c in cs:
            c.acquire_one()
        kv = BatchedKvCacheInt4(cs)
        x = (torch.randn(bsz, hidden, device=dev) * 0.7).half()
        blen = BatchLenInfo([], bsz, dev)
        with torch.no_grad():
            for _ in range(3):
                layer(x, blen, None, kv)
            torch.cuda.synchronize()
            t0, t1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            t0.record()
            for _ in range(iters):
                layer(x, blen, None, kv)
            t1.record()
            torch.cuda.synchronize()
            total = t0.elapsed_time(t1) * 1e3 / iters
            # the same step captured into a HIP graph and replayed: GPU time without the Python / launch overhead per op
            graph_us = float("nan")
            try:
                gr = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gr):
                    layer(x, blen, None, kv)
                gr.replay()
                torch.cuda.synchronize()
                t0.record()
                for _ in range(iters):
                    gr.replay()
                t1.record()
                torch.cuda.synchronize()
                graph_us = t0.elapsed_time(t1) * 1e3 / iters
            except Exception as e:      # noqa: BLE001
                print("  (graph capture failed:", type(e).__name__, str(e)[:100], ")")
            for n, f in orig.items():
                setattr(ops, n, wrap(n, f))
            spans.clear()
            for _ in range(5):
                layer(x, blen, None, kv)
            torch.cuda.synchronize()
            for n, f in orig.items():
                setattr(ops, n, f)
        per = {n: sum(a.elapsed_time(b) for a, b in v) * 1e3 / 5 for n, v in spans.items()}
        print(f"  {bsz:>5} {total:>9.1f} {graph_us:>9.1f} | " + " ".join(f"{per.get(n, 0.0):>14.1f}" for n in orig))
        for c in cs:
            c.release()


if __name__ == "__main__":
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024,
         tuple(int(b) for b in sys.argv[2].split(",")) if len(sys.argv) > 2 else (1, 4, 8, 16, 32, 64, 128))
