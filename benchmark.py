#!/usr/bin/env python3
"""Benchmark CLI with the interface of the reference's benchmark.py (reference bench:21-42, bench:99-138):

    python benchmark.py [--causal] [--mask-prob P] [--only-forwards | --only-backwards] [--num-times K]

Sweeps seq_len 128..8192 at batch 4, heads 8, dim 64 for float32 and float16 (the reference's grid) plus bfloat16, and
prints, per sequence length, the fused op's time, the time of the plain PyTorch composite (`plain_cosine_sim_attention`,
"baseline", what the reference compares against) and "slower" = fused / baseline exactly like the reference's table
(values < 1 mean the fused op is faster).  Two extra columns: algorithmic TFLOP/s of the fused op and the time of
torch.nn.functional.scaled_dot_product_attention (softmax flash attention of the same shape) for orientation.

Protocol (reference pb:7-56): 10 warm-up calls, mean of --num-times device-event timed calls; forward+backward window =
fn(); out.sum().backward().  Needs a GPU (this is a measurement tool of the product, it never touches the oracle).
"""
import argparse
import json
import os
import sys

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from flash_cosine_sim_attention_amd import flash_cosine_sim_attention, plain_cosine_sim_attention   # noqa: E402

SEQ_LENS = (128, 256, 512, 1024, 2048, 4096, 8192)
BATCH, HEADS, DIM = 4, 8, 64


def timed(fn, num_times, backwards, only_backwards):
    """mean ms per call; the timed window follows the reference: forward, forward+backward, or backward only."""
    def once(record):
        if only_backwards:
            out = fn()
            loss = out.sum()
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record(); loss.backward(); e.record()
        else:
            s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            s.record()
            out = fn()
            if backwards:
                out.sum().backward()
            e.record()
        if record:
            torch.cuda.synchronize()
            return s.elapsed_time(e)
        return 0.0
    for _ in range(10):
        once(False)
    torch.cuda.synchronize()
    return sum(once(True) for _ in range(num_times)) / num_times


def graph_replay_ms(fn, zero, backwards, num_times):
    """mean ms per replay of the window captured in a HIP graph (the library allocates nothing and launches on the caller's
    stream only, so the op is capturable after a warm-up)."""
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        for _ in range(3):
            out = fn()
            if backwards:
                out.sum().backward()
    torch.cuda.current_stream().wait_stream(side)
    zero()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = fn()
        if backwards:
            out.sum().backward()
    for _ in range(10):
        graph.replay()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    torch.cuda.synchronize()
    s.record()
    for _ in range(num_times):
        graph.replay()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / num_times


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--causal", default=False, action="store_true")
    ap.add_argument("--mask-prob", type=float, default=0.)
    ap.add_argument("--only-forwards", default=False, action="store_true")
    ap.add_argument("--only-backwards", default=False, action="store_true")
    ap.add_argument("--num-times", default=20, type=int)
    ap.add_argument("--dtypes", default="float32,float16,bfloat16")
    ap.add_argument("--json", default=None, help="also write the table as JSON to this path")
    ap.add_argument("--seq-lens", type=int, nargs="+", default=list(SEQ_LENS), help="sequence lengths (default: the reference's sweep)")
    ap.add_argument("--dim-head", type=int, default=64, help="head dimension (16, 32, 64, 96, 128; the reference's sweep uses 64)")
    ap.add_argument("--no-baseline", default=False, action="store_true", help="skip the plain PyTorch composite column (dims sweep)")
    ap.add_argument("--hip-graph", default=False, action="store_true",
                    help="extra column: the fused op's window captured once in a HIP graph (torch.cuda.graph) and replayed -- what a "
                         "launch-bound caller (short sequences) gets by capturing its step")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        sys.exit("a GPU must be available to run the benchmark")
    assert 0 <= args.mask_prob < 1
    assert not (args.only_forwards and args.only_backwards)
    assert not (args.causal and args.mask_prob > 0), "mask should not be given if causal"
    backwards = not args.only_forwards
    dim = args.dim_head
    rows = []
    for name in args.dtypes.split(","):
        dtype = getattr(torch, name)
        print("-" * 100)
        print(f"{name}\t\tbatch: {BATCH}\theads: {HEADS}\tdim {dim}\t"
              f"{'causal ' if args.causal else ''}{'mask-prob %.2f ' % args.mask_prob if args.mask_prob else ''}"
              f"{'forward' if args.only_forwards else 'backward' if args.only_backwards else 'forward+backward'}")
        print("-" * 100)
        # one untimed call per dtype: the first launch of a kernel instantiation loads its code object (tens of ms)
        w = torch.randn(BATCH, HEADS, 256, dim, dtype=dtype, device="cuda", requires_grad=True)
        flash_cosine_sim_attention(w, w, w, causal=args.causal).sum().backward()
        torch.cuda.synchronize()
        for seq in args.seq_lens:
            q, k, v = (torch.randn(BATCH, HEADS, seq, dim, dtype=dtype, device="cuda").requires_grad_(backwards) for _ in range(3))
            mask = None
            if args.mask_prob > 0:
                mask = torch.zeros((BATCH, seq), device="cuda").uniform_(0, 1) > args.mask_prob
            kw = dict(causal=args.causal, mask=mask)

            def zero():
                q.grad = k.grad = v.grad = None

            def fused():
                zero()
                return flash_cosine_sim_attention(q, k, v, **kw)

            def baseline():
                zero()
                return plain_cosine_sim_attention(q, k, v, **kw)

            def sdpa():
                zero()
                am = None if mask is None else mask[:, None, None, :]
                return torch.nn.functional.scaled_dot_product_attention(q, k, v, attn_mask=am, is_causal=args.causal)

            t_fused = timed(fused, args.num_times, backwards, args.only_backwards)
            t_graph = graph_replay_ms(fused, zero, backwards, args.num_times) if args.hip_graph and not args.only_backwards else None
            try:
                t_base = None if args.no_baseline else timed(baseline, args.num_times, backwards, args.only_backwards)
            except torch.OutOfMemoryError:
                torch.cuda.empty_cache()
                t_base = None
            try:
                t_sdpa = timed(sdpa, args.num_times, backwards, args.only_backwards)
            except Exception:        # e.g. no f32 flash kernel
                t_sdpa = None
            frac = (seq + 1) / (2.0 * seq) if args.causal else 1.0
            mult = 4 if args.only_forwards else 10 if args.only_backwards else 14
            tflops = mult * BATCH * HEADS * seq * seq * dim * frac / (t_fused * 1e-3) / 1e12
            slower = t_fused / t_base if t_base else 0.0
            print(f"seq_len: {seq}\tslower: {slower:.2f}x\tkernel: {t_fused:.3f}ms\tbaseline: "
                  f"{('skipped' if args.no_baseline else 'oom') if t_base is None else '%.3fms' % t_base}\t{tflops:7.1f} TFLOP/s\tsdpa: "
                  f"{'n/a' if t_sdpa is None else '%.3fms' % t_sdpa}" + ("" if t_graph is None else f"\thip-graph replay: {t_graph:.3f}ms"))
            rows.append(dict(dtype=name, seq_len=seq, kernel_ms=t_fused, baseline_ms=t_base, sdpa_ms=t_sdpa, tflops=tflops, graph_ms=t_graph))
            del q, k, v
            torch.cuda.empty_cache()
    if args.json:
        json.dump(dict(args=vars(args), rows=rows), open(args.json, "w"), indent=1)


if __name__ == "__main__":
    main()
