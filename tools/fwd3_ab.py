"""Interleaved same-process A/B of the D = 128 forward forms: the C ABI's debug knob fcsa_debug_forward_form (include/fcsa.h) switches the form between launches, so one process times
the lean 32-row form ("0") and the product's wide form ("").  (The development snapshots of round 5 -- commit "fcsa_fwd3 v2-v5" -- also
understood ring depths, row-sum forms and ablations: "r3", "r4d", "rx" ...; the tables they produced are profiles/r05_fwd3_ab_v*.txt,
r05_fwd3_ablations_v*.txt.)  HIP-event timing of forward-only calls under no_grad, `rounds` interleaved rounds of `iters`.
usage: python tools/fwd3_ab.py [--shapes B,H,N,M,causal ...] [--variants 0 r3 ...] [--dtype bf16]"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--shapes", nargs="*", default=["4,8,4096,4096,1", "4,8,4096,4096,0", "16,8,2048,2048,1", "2,8,8192,8192,1"])
    ap.add_argument("--variants", nargs="*", default=["0", "on"])      # "on" = the product's dispatch (variable unset)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--rounds", type=int, default=4)
    ap.add_argument("--iters", type=int, default=30)
    args = ap.parse_args()
    import flash_cosine_sim_attention_amd as F
    from flash_cosine_sim_attention_amd import _lib
    dt = {"bf16": torch.bfloat16, "f16": torch.float16}[args.dtype]
    print("device:", torch.cuda.get_device_name(0), "dtype:", args.dtype)
    for shp in args.shapes:
        B, H, N, M, causal = (int(x) for x in shp.split(","))
        g = torch.Generator(device="cuda").manual_seed(0)
        q = torch.randn((B, H, N, 128), device="cuda", dtype=dt, generator=g)
        k = torch.randn((B, H, M, 128), device="cuda", dtype=dt, generator=g)
        v = torch.randn((B, H, M, 128), device="cuda", dtype=dt, generator=g)
        frac = 1.0
        if causal:
            frac = sum(min(M, i + (M - N) + 1) for i in range(N)) / (N * M)
        flops = 4.0 * B * H * N * M * 128 * frac
        times = {vn: [] for vn in args.variants}
        outs = {}
        with torch.no_grad():
            for r in range(args.rounds + 1):
                for vn in args.variants:
                    _lib.forward_form(0 if vn == "0" else 1)
                    for _ in range(5):
                        o = F.flash_cosine_sim_attention(q, k, v, causal=bool(causal))
                    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    s.record()
                    for _ in range(args.iters):
                        o = F.flash_cosine_sim_attention(q, k, v, causal=bool(causal))
                    e.record()
                    torch.cuda.synchronize()
                    if r > 0:                                   # round 0 = warm-up
                        times[vn].append(s.elapsed_time(e) / args.iters * 1e3)
                    outs[vn] = o
        _lib.forward_form(1)
        base = outs[args.variants[0]].float()
        print(f"== (B,H,N,M)=({B},{H},{N},{M}) causal={causal}  {flops / 1e9:.1f} GFLOP")
        for vn in args.variants:
            ts = sorted(times[vn])
            med = ts[len(ts) // 2]
            d = (outs[vn].float() - base).abs().max().item()
            nan = int((~torch.isfinite(outs[vn])).sum().item())
            print(f"   {vn:10s} median {med:8.1f} us  min {ts[0]:8.1f}  {flops / med / 1e6:7.1f} TFLOP/s   max|o - o[{args.variants[0]}]| {d:.3e}  non-finite {nan}")


if __name__ == "__main__":
    main()
