#!/usr/bin/env python3
"""bench.py -- headline benchmark of the hot path (BASELINE.json metric).

    python bench.py --gpus N --steps K --warmup W

A "step" is one forward + backward of flash_cosine_sim_attention on one batch of synthetic
input at the configuration the metric is quoted on (BASELINE.json configs[2] / SURVEY "C3"):
    q, k, v ~ N(0,1), shape (B=4, H=8, N=4096, D=64), bf16, causal=True, scale=8, groups=1,
    l2norm_qk=True; backward driven by a fixed random dO (inputs resident in HBM).
For N > 1 GPUs (launched with torch.distributed.run, one rank per GPU) every rank runs the same
per-GPU workload -- the op is embarrassingly parallel over (batch, head), there is no collective on the
data path ("replicas only") -- so scaling is "weak" and `value` is the
sum over ranks of algorithmic FLOPs / max-over-ranks time.

FLOP convention: GEMM FLOPs only, fwd 4*B*H*N*M*D, bwd 10*B*H*N*M*D, times the causal
fraction n_valid/(N*M).  Rank 0 prints ONE JSON line.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

WORKLOAD = dict(name="C3 fwd+bwd (4,8,4096,64) bf16 causal", B=4, H=8, N=4096, M=4096, D=64,
                dtype="bf16", causal=True, scale=8.0, groups=1)
MFMA_PEAK_TFLOPS = 2500.0         # dense bf16/f16 MFMA peak, MI355X_MICROARCH.md (AMD figure excl. sparsity)


def causal_fraction(n, m):
    diff = m - n
    nv = sum(min(m, max(0, i + diff + 1)) for i in range(n))
    return nv / float(n * m)


def flops(w, fwd=True, bwd=True):
    per = (4 if fwd else 0) + (10 if bwd else 0)
    f = per * w["B"] * w["H"] * w["N"] * w["M"] * w["D"]
    return f * (causal_fraction(w["N"], w["M"]) if w["causal"] else 1.0)


def lib_sha256():
    """sha256 of the loaded libfcsa_hip.so: ties PMC measurements in profiles/ to the exact binary they were taken on."""
    import hashlib
    from flash_cosine_sim_attention_amd import _lib
    return hashlib.sha256(open(_lib.LIB_PATH, "rb").read()).hexdigest()


def aggregate_over_ranks(elapsed_s, flops_local, dist=None, device="cpu"):
    """Whole-job figures from per-rank ones: time = MAX over ranks, work = SUM over ranks (replicas, no data-path
    collective).  Returns (elapsed_max_s, flops_total).  `dist` is torch.distributed (initialised) or None."""
    if dist is None or not dist.is_initialized() or dist.get_world_size() == 1:
        return float(elapsed_s), float(flops_local)
    t = torch.tensor([elapsed_s], device=device, dtype=torch.float64)
    f = torch.tensor([flops_local], device=device, dtype=torch.float64)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(f, op=dist.ReduceOp.SUM)
    return float(t.item()), float(f.item())


# ---- the other BASELINE configs and the reference's own timing protocol, in the SAME driver-run line (rank 0) -------------
# Shapes / flags are BASELINE.json `configs` (C2, C4, C5) plus C5 at scale 8 (the table's default scale; runs the
# dynamic-shift forward), C3 at D = 128 and a learned-bias case.  Protocol = the reference's benchmark.py:7-56: 10 warm-up
# calls, then the mean of 20 device-event timed calls; the forward+backward window is `fn(); out.backward(dO)` like the
# headline step.  Each entry also times torch SDPA (softmax flash attention) on the same tensors.
EXTRA_CONFIGS = {
    "C2": dict(q=(4, 8, 1024, 64), kv=(4, 8, 1024, 64), dtype="f16", causal=False, mask=False, scale=8, groups=1, bwd=False),
    "C4": dict(q=(1, 8, 1024, 64), kv=(1, 8, 8192, 64), dtype="f16", causal=False, mask=True, scale=8, groups=1, bwd=True),
    "C5": dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype="bf16", causal=True, mask=False, scale=1, groups=8, bwd=True),
    "C5s8": dict(q=(4, 8, 2048, 128), kv=(4, 2048, 128), dtype="bf16", causal=True, mask=False, scale=8, groups=8, bwd=True),
    "C3_d128": dict(q=(4, 8, 4096, 128), kv=(4, 8, 4096, 128), dtype="bf16", causal=True, mask=False, scale=8, groups=1, bwd=True),
    # float16 at scale 16: beyond the constant exponent window (scale * groups > 11), i.e. the forward's online per-row reference
    "C3_f16_scale16": dict(q=(4, 8, 4096, 64), kv=(4, 8, 4096, 64), dtype="f16", causal=True, mask=False, scale=16, groups=1, bwd=True),
    "C2_bias": dict(q=(4, 8, 1024, 64), kv=(4, 8, 1024, 64), dtype="f16", causal=False, mask=False, scale=8, groups=1, bwd=True, bias=True),
}
_DT = {"bf16": torch.bfloat16, "f16": torch.float16, "f32": torch.float32}


def event_time_ms(fn, iters=20, warm=10):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(iters):
        fn()
    e.record()
    torch.cuda.synchronize()
    return s.elapsed_time(e) / iters


def protocol_time_ms(fn):
    """The reference protocol (10 warm-ups, mean of 20 event-timed calls).  A 20-call window of a small configuration is 0.5 - 3 ms long --
    shorter than the chip's clock ramp after the host-side pause in front of it, and the round-6 evidence session caught one such window at
    5x its usual value -- so windows below 5 ms are measured five times (no further warm-up) and the MEDIAN window is reported."""
    t = event_time_ms(fn)
    if t * 20 < 5.0:
        ts = sorted([t] + [event_time_ms(fn, warm=0) for _ in range(4)])
        t = ts[len(ts) // 2]
    return t


def run_extra_config(F, c, sdpa=True):
    dt = _DT[c["dtype"]]
    g = torch.Generator(device="cuda").manual_seed(0)
    B, H, N, D = c["q"]
    M = c["kv"][-2]
    q = torch.randn(c["q"], device="cuda", dtype=dt, generator=g).requires_grad_(c["bwd"])
    k = torch.randn(c["kv"], device="cuda", dtype=dt, generator=g).requires_grad_(c["bwd"])
    v = torch.randn(c["kv"], device="cuda", dtype=dt, generator=g).requires_grad_(c["bwd"])
    mask = (torch.rand((B, M), device="cuda", generator=g) > 0.25) if c["mask"] else None      # README protocol: 25 % masked
    do = torch.randn(c["q"], device="cuda", dtype=dt, generator=g)
    bias = (0.5 * torch.randn((H, N, M), device="cuda", dtype=dt, generator=g)).requires_grad_() if c.get("bias") else None
    unit = B * H * N * M * D * (causal_fraction(N, M) if c["causal"] else 1.0)
    kw = dict(mask=mask, attn_bias=bias, causal=c["causal"], scale=c["scale"], groups=c["groups"])

    def fwd():
        with torch.no_grad():
            return F.flash_cosine_sim_attention(q, k, v, **kw)

    def fb():
        q.grad = k.grad = v.grad = None
        if bias is not None:
            bias.grad = None
        F.flash_cosine_sim_attention(q, k, v, **kw).backward(do)

    r = {}
    t_f = protocol_time_ms(fwd)
    r["fwd_ms"] = round(t_f, 4)
    r["fwd_tflops"] = round(4 * unit / t_f / 1e9, 1)
    if c["bwd"]:
        t_fb = protocol_time_ms(fb)
        r["ms"] = round(t_fb, 4)
        r["tflops"] = round(14 * unit / t_fb / 1e9, 1)
    else:
        r["ms"], r["tflops"] = r["fwd_ms"], r["fwd_tflops"]
    if sdpa:
        try:
            ke, ve = (k, v) if k.dim() == 4 else (k[:, None].expand(B, H, M, D), v[:, None].expand(B, H, M, D))
            am = None if mask is None else mask[:, None, None, :].expand(B, 1, N, M)
            if bias is not None:
                am = bias.detach()[None].expand(B, H, N, M)                  # additive float mask (SDPA gives no bias gradient)
            sd = torch.nn.functional.scaled_dot_product_attention

            def sfwd():
                with torch.no_grad():
                    return sd(q, ke, ve, attn_mask=am, is_causal=c["causal"])

            def sfb():
                q.grad = k.grad = v.grad = None
                sd(q, ke, ve, attn_mask=am, is_causal=c["causal"]).backward(do)

            ts = protocol_time_ms(sfb if c["bwd"] else sfwd)
            r["sdpa_ms"] = round(ts, 4)
            r["vs_flash_sdpa"] = round(ts / r["ms"], 2)
        except Exception as ex:                                   # pragma: no cover
            r["sdpa_error"] = repr(ex)[:120]
    return r


def accuracy_report(F, w, q, k, v, do, slices):
    """The metric's second half.  Forward and gradients of the HIP op on (batch, head) slices of the headline workload against a
    float64 evaluation of the same function of the same bf16 inputs (plain softmax form + torch.autograd, exact l2norm), beside
    the reference-style PyTorch composite (plain_cosine_sim_attention, py:75-126) evaluated in the SAME dtype on the same slices --
    the comparator for north_star's "within 1e-3 rel." -- and the HIP op against float64 math on the 16-bit OPERANDS of the S
    product (c1 * q^, k^ rounded to bf16: the rounding every bf16 implementation shares)."""
    # exact() below is the headline's function only (square causal mask, one l2norm group): refuse anything else instead of
    # reporting a number for another function than the one the op was called with
    assert w["causal"] and w["groups"] == 1 and q.shape[-2] == k.shape[-2], "accuracy_report covers the causal, groups=1, N == M headline"

    def stats(got, ref):
        d = got.double() - ref.double()
        return {"rel_l2": float(d.norm() / ref.double().norm()), "max_abs": float(d.abs().max())}

    def exact(qs, ks, vs, dos, operand_dtype=None):
        qs, ks, vs = (t.double().requires_grad_() for t in (qs, ks, vs))
        qn, kn = torch.nn.functional.normalize(qs, dim=-1), torch.nn.functional.normalize(ks, dim=-1)
        if operand_dtype is not None:      # straight-through rounding of the operands (the cast's gradient is the identity)
            c1 = w["scale"] * 1.4426950408889634
            qn, kn = (qn * c1).to(operand_dtype).double() / c1, kn.to(operand_dtype).double()
        sc = (qn @ kn.t()) * w["scale"]
        sc = sc.masked_fill(torch.ones_like(sc, dtype=torch.bool).triu(1), float("-inf"))
        o = torch.softmax(sc, -1) @ vs
        (o * dos.double()).sum().backward()
        return o.detach(), qs.grad, ks.grad, vs.grad

    q.grad = k.grad = v.grad = None
    o = F.flash_cosine_sim_attention(q, k, v, causal=w["causal"], scale=w["scale"], groups=w["groups"])
    o.backward(do)
    names = ("o", "dq", "dk", "dv")
    acc = {n: {"hip": [], "torch_same_dtype": [], "hip_vs_16bit_operands": []} for n in names}
    for (b, h) in slices:
        sl = lambda t: t.detach()[b, h]
        ref = exact(sl(q), sl(k), sl(v), sl(do))
        ref16 = exact(sl(q), sl(k), sl(v), sl(do), operand_dtype=q.dtype)
        got = (sl(o), q.grad[b, h], k.grad[b, h], v.grad[b, h])
        qc, kc, vc = (t.detach()[b:b + 1, h:h + 1].clone().requires_grad_() for t in (q, k, v))
        oc = F.plain_cosine_sim_attention(qc, kc, vc, causal=w["causal"], scale=w["scale"], groups=w["groups"])
        oc.backward(do[b:b + 1, h:h + 1])
        comp = (oc.detach()[0, 0], qc.grad[0, 0], kc.grad[0, 0], vc.grad[0, 0])
        for n, g_, c_, r_, r16 in zip(names, got, comp, ref, ref16):
            acc[n]["hip"].append(stats(g_, r_))
            acc[n]["torch_same_dtype"].append(stats(c_, r_))
            acc[n]["hip_vs_16bit_operands"].append(stats(g_, r16))
    worst = lambda lst: {k_: float("%.3e" % max(x[k_] for x in lst)) for k_ in ("rel_l2", "max_abs")}
    out = {leg: {n: worst(acc[n][leg]) for n in names} for leg in ("hip", "torch_same_dtype", "hip_vs_16bit_operands")}
    out["slices"] = [list(s_) for s_ in slices]
    out["what"] = ("worst over the slices; hip / torch_same_dtype: against float64 softmax(scale * l2norm(q) l2norm(k)^T) v + autograd on the "
                   "same bf16 inputs and dO (torch_same_dtype = plain_cosine_sim_attention composite in bf16); hip_vs_16bit_operands: against "
                   "float64 math on c1 * q^, k^ rounded to bf16")
    return out


def reference_protocol(F, w, dt):
    """The reference's timing protocol on the headline workload (flash_cosine_sim_attention/benchmark.py:7-56, 46-48):
    10 warm-ups, mean of 20 calls, each timed by its own device-event pair around `out = fn(); out.sum().backward()`."""
    g = torch.Generator(device="cuda").manual_seed(1)
    shp = (w["B"], w["H"], w["N"], w["D"])
    q, k, v = (torch.randn(shp, device="cuda", dtype=dt, generator=g).requires_grad_() for _ in range(3))

    def once():
        s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        q.grad = k.grad = v.grad = None
        s.record()
        out = F.flash_cosine_sim_attention(q, k, v, causal=w["causal"], scale=w["scale"], groups=w["groups"])
        out.sum().backward()
        e.record()
        return s, e

    for _ in range(10):
        once()
    torch.cuda.synchronize()
    ev = [once() for _ in range(20)]
    torch.cuda.synchronize()
    ms = sum(s.elapsed_time(e) for s, e in ev) / len(ev)
    return {"ms": round(ms, 4), "tflops": round(flops(w) / ms / 1e9, 1),
            "what": "reference protocol: 10 warm-ups, mean of 20 event-timed `out = fn(); out.sum().backward()` (includes the .sum() "
                    "kernel and the ones-expand of its backward)"}


def small_n(F):
    """host-bound sizes: f16 causal forward+backward, B4 H8 D64, N = 128 / 256 / 512, eager, against SDPA (reference protocol)."""
    out = {}
    for n in (128, 256, 512):
        c = dict(q=(4, 8, n, 64), kv=(4, 8, n, 64), dtype="f16", causal=True, mask=False, scale=8, groups=1, bwd=True)
        r = run_extra_config(F, c)
        out[str(n)] = {"ms": r["ms"], "sdpa_ms": r.get("sdpa_ms"), "vs_flash_sdpa": r.get("vs_flash_sdpa")}
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)     # reference protocol averages 20 (benchmark.py:26); 200 steps = 80 ms
                                                          # make one timed window robust against a single scheduling hiccup
    ap.add_argument("--warmup", type=int, default=10)     # reference protocol: 10 warm-ups (benchmark.py:11)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-kernel-events", action="store_true", help="skip the instrumented pass for the roofline object")
    ap.add_argument("--no-extra-configs", action="store_true", help="skip the C2 / C4 / C5 / ... entries and the reference-protocol leg")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        print(json.dumps({"error": "no GPU visible; bench.py measures the HIP path only"}))
        sys.exit(2)
    torch.cuda.set_device(local_rank)
    dist = None
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist_mod.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        dist = dist_mod

    import flash_cosine_sim_attention_amd as F
    from flash_cosine_sim_attention_amd import _lib

    w = WORKLOAD
    dt = torch.bfloat16
    g = torch.Generator(device="cuda").manual_seed(rank)
    shp = (w["B"], w["H"], w["N"], w["D"])
    q, k, v = (torch.randn(shp, device="cuda", dtype=dt, generator=g).requires_grad_() for _ in range(3))
    do = torch.randn(shp, device="cuda", dtype=dt, generator=g)

    def step():
        q.grad = k.grad = v.grad = None
        o = F.flash_cosine_sim_attention(q, k, v, causal=w["causal"], scale=w["scale"], groups=w["groups"])
        o.backward(do)
        return o

    def barrier():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    # Runtime pre-warm (setup, untimed, not part of the W warm-up steps): the first ~second of a fresh process on a
    # fresh box pays HIP module loading, allocator growth and clock ramp-up (measured: 1.2 ms/step in the first
    # process vs 0.45 ms afterwards).  Run the step for a fixed wall time so the numbers below describe the
    # steady state whatever W the caller picked.
    t_pre = time.perf_counter()
    while time.perf_counter() - t_pre < 1.5:
        for _ in range(10):
            step()
        torch.cuda.synchronize()
    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed, flops_per_step_all_ranks = aggregate_over_ranks(elapsed, flops(w), dist, "cuda")
    ms_per_step = elapsed / args.steps * 1e3
    value = flops_per_step_all_ranks / (ms_per_step * 1e-3) / 1e12

    # ---- roofline: dominant kernel, timed with HIP events on the launch stream (library hook) ----------
    roofline = None
    kernels = None
    if rank == 0 and not args.no_kernel_events:
        isteps = min(args.steps, 50)                      # instrumented pass (event pair around every launch)
        _lib.profile_enable(True)
        for _ in range(isteps):
            step()
        torch.cuda.synchronize()
        _lib.profile_enable(False)
        stats = _lib.profile_collect()
        kernels = {s["name"]: dict(calls=s["calls"], avg_us=round(s["total_ms"] / max(s["calls"], 1) * 1e3, 2),
                                   per_step_us=round(s["total_ms"] / isteps * 1e3, 2)) for s in stats}
        # algorithmic GEMM FLOPs each attention kernel is responsible for, per launch:
        #   fwd: QK^T + PV = 4 BHNMD;  bwd_dkv: S, dP, dV, dK = 8 BHNMD;  bwd_dq: dQ = 2 BHNMD (its S/dP recompute
        #   is not algorithmic work).  All times the causal fraction.
        unit = w["B"] * w["H"] * w["N"] * w["M"] * w["D"] * (causal_fraction(w["N"], w["M"]) if w["causal"] else 1.0)
        alg = {"fwd": 4 * unit, "bwd_dkv": 8 * unit, "bwd_dq": 2 * unit}
        cand = [s for s in stats if s["name"] in alg]
        if cand:
            dom = max(cand, key=lambda s: s["total_ms"])
            avg_s = dom["total_ms"] / dom["calls"] * 1e-3
            ach = alg[dom["name"]] / avg_s / 1e12
            roofline = {"kernel": dom["name"], "bound": "mfma", "achieved": round(ach, 2), "peak": MFMA_PEAK_TFLOPS,
                        "unit": "TFLOP/s", "frac": round(ach / MFMA_PEAK_TFLOPS, 4), "traffic": None,
                        "avg_launch_us": round(avg_s * 1e6, 2), "algorithmic_gflop_per_launch": round(alg[dom["name"]] / 1e9, 2),
                        "timing": "hipEvent pairs recorded by libfcsa_hip on the launch stream over %d steps" % isteps}
            # The event pairs around every launch are a DIFFERENT regime than the headline's timed region (they serialise the
            # launches and cost clock: the instrumented kernels of a step sum to more than the un-instrumented step).  Say so in
            # the line, and give the whole-step fraction -- un-instrumented, all kernels and gaps included -- beside it.
            ksum_us = sum(s["total_ms"] for s in stats) / isteps * 1e3
            roofline["regime"] = ("instrumented pass (an event pair around every launch), separate from the timed region: its kernels sum to "
                                  "%.1f us per step, the un-instrumented step takes %.1f us" % (ksum_us, ms_per_step * 1e3))
            roofline["kernel_sum_per_step_us_instrumented"] = round(ksum_us, 2)
            roofline["step_us_uninstrumented"] = round(ms_per_step * 1e3, 2)
            scale_u = min(1.0, ms_per_step * 1e3 / ksum_us) if ksum_us > 0 else 1.0
            roofline["avg_launch_us_scaled_to_step"] = round(avg_s * 1e6 * scale_u, 2)
            roofline["frac_scaled_to_step"] = round(ach / scale_u / MFMA_PEAK_TFLOPS, 4)
            roofline["whole_step"] = {"achieved": round(value / max(world, 1), 2), "frac": round(value / max(world, 1) / MFMA_PEAK_TFLOPS, 4),
                                      "what": "all algorithmic FLOPs of fwd+bwd / un-instrumented ms_per_step (per GPU), every kernel and launch gap included"}
            # HBM bytes per launch of that kernel: PMC counters cannot be read from inside this process, so the value is the
            # committed measurement of this same command (tools/gpu_pmc.sh: separate rocprofv3 --pmc passes, FETCH_SIZE x2
            # gfx950 correction) -- but ONLY if it was taken on the very library binary that is loaded now (sha256 recorded by
            # the PMC run); after any kernel change it reads null until the PMC passes are repeated.
            import glob
            src, hit = _lib.source_sha256(), None
            for tfile in sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_pmc_traffic.json")), reverse=True):
                rec = json.load(open(tfile))
                tk = rec.get("kernels", {}).get(dom["name"] + "_kernel")
                if tk and rec.get("src_sha256") == src:
                    hit = (os.path.relpath(tfile, ROOT), tk, rec.get("derived", {}).get(dom["name"] + "_kernel"))
                    break
            if hit:
                # north_star: "rocprof-reported HBM GB/s": the PMC bytes of EVERY kernel of the step over this run's event-timed
                # launch durations (peak 8 TB/s); the l2norm pass is the one HBM-bound kernel of the step
                rec_k = json.load(open(os.path.join(ROOT, hit[0]))).get("kernels", {})
                hbm = {}
                for name, kst in kernels.items():
                    tk2 = rec_k.get(name + "_kernel")
                    if tk2 and kst["avg_us"] > 0:
                        gbps = tk2["total_bytes"] / (kst["avg_us"] * 1e-6) / 1e9
                        hbm[name] = {"bytes_per_launch": round(tk2["total_bytes"]), "avg_us": kst["avg_us"], "GB/s": round(gbps, 1),
                                     "frac_of_8TBps": round(gbps / 8000.0, 4)}
                roofline["hbm"] = hbm
                roofline["traffic"] = round(hit[1]["total_bytes"])
                roofline["traffic_source"] = ("%s (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, bytes/launch; measured on a build of the same "
                                              "kernel sources + Makefile, source sha256 %s)" % (hit[0], src[:12]))
                if hit[2]:      # matrix-pipe occupancy and effective clock of the same PMC passes (tools/pmc_summary.py)
                    roofline["mfma_busy"] = hit[2].get("mfma_busy")
                    roofline["effective_clock_ghz"] = hit[2].get("effective_clock_ghz")
                    if "mfma_busy_memtime" in hit[2]:      # the kernel's own clock (s_memtime, -DFCSA_TRACE_WG build; tools/pmc_clock_crosscheck.py)
                        roofline["mfma_busy_memtime"] = hit[2]["mfma_busy_memtime"]
                        roofline["effective_clock_ghz_memtime"] = hit[2]["effective_clock_ghz_memtime"]
                    roofline["mfma_busy_what"] = ("SQ_VALU_MFMA_BUSY_CYCLES / (4 SIMDs x CUs x GRBM_GUI_ACTIVE per XCD), effective clock = GRBM_GUI_ACTIVE "
                                                  "per XCD / kernel duration of the rocprofv3 kernel trace: the pipe's occupancy in REAL clocks next to `frac`, "
                                                  "which prices the same launch against the 2.4 GHz nominal peak.  GRBM_GUI_ACTIVE of a profiled dispatch includes its set-up (the HBM-bound l2norm kernel "
                                                  "reads 4.2 GHz that way): *_memtime uses the median workgroup duration in s_memtime ticks of the trace build instead")
            else:
                roofline["traffic_source"] = ("null: no profiles/r*_pmc_traffic.json was measured on these kernel sources (source sha256 %s; "
                                              "library sha256 %s)" % (src[:12], lib_sha256()[:12]))

    # ---- stock softmax flash attention on the same box, shape, dtype, causal flag (the ">= 1.2x" target) -------
    sdpa = None
    if rank == 0:
        qs, ks, vs = (t.detach().clone().requires_grad_() for t in (q, k, v))

        def sdpa_step():
            qs.grad = ks.grad = vs.grad = None
            torch.nn.functional.scaled_dot_product_attention(qs, ks, vs, is_causal=w["causal"]).backward(do)

        try:
            for _ in range(10):
                sdpa_step()
            torch.cuda.synchronize()
            n_sd = min(args.steps, 50)
            t1 = time.perf_counter()
            for _ in range(n_sd):
                sdpa_step()
            torch.cuda.synchronize()
            sd_ms = (time.perf_counter() - t1) / n_sd * 1e3
            sdpa = {"ms_per_step": round(sd_ms, 4), "tflops": round(flops(w) / (sd_ms * 1e-3) / 1e12, 2), "steps": n_sd,
                    "what": "torch.nn.functional.scaled_dot_product_attention fwd+bwd, same tensors / dtype / causal flag, same FLOP convention"}
        except Exception as e:                                   # pragma: no cover
            sdpa = {"error": repr(e)[:200]}

    # ---- max |delta| vs a PyTorch f32 evaluation of the same math on (b,h) slices ----------------------
    max_delta = None
    if rank == 0:
        with torch.no_grad():
            o = F.flash_cosine_sim_attention(q, k, v, causal=w["causal"], scale=w["scale"])
            md = 0.0
            for (b, h) in ((0, 0), (w["B"] - 1, w["H"] - 1)):
                qs, ks, vs = q[b, h].float(), k[b, h].float(), v[b, h].float()
                s = torch.nn.functional.normalize(qs, dim=-1) @ torch.nn.functional.normalize(ks, dim=-1).t() * w["scale"]
                s = s.masked_fill(torch.ones_like(s, dtype=torch.bool).triu(1), float("-inf"))
                ref = torch.softmax(s, -1) @ vs
                md = max(md, (o[b, h].float() - ref).abs().max().item())
            max_delta = md

    accuracy = None
    if rank == 0:
        try:
            accuracy = accuracy_report(F, w, q, k, v, do, ((0, 0), (w["B"] - 1, w["H"] - 1)))
        except Exception as ex:                                   # pragma: no cover
            accuracy = {"error": repr(ex)[:200]}

    # ---- the other configs, the reference's timing protocol, host-bound sizes (rank 0; ~2 s) -----------------------------
    configs = ref_proto = small = None
    if rank == 0 and world == 1 and not args.no_extra_configs:      # (N > 1 runs report the headline only: the other ranks are waiting)
        configs = {}
        for name, c in EXTRA_CONFIGS.items():
            try:
                configs[name] = run_extra_config(F, c)
            except Exception as ex:                                # pragma: no cover
                configs[name] = {"error": repr(ex)[:200]}
        ref_proto = reference_protocol(F, w, dt)
        small = small_n(F)

    # ---- CPU baseline: ports of the reference's CPU-runnable paths on the host cores, bounded samples ----------
    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        from oracle import torch_cpu_port as P
        sample = (1, w["H"], w["N"], w["D"])                 # one batch element of the same workload
        secs = P.time_fwd_bwd(sample, sample, dt, causal=w["causal"], scale=w["scale"], reps=2)
        wf = dict(w, B=1)
        cpu = {"value": round(flops(wf) / secs / 1e12, 4), "unit": "TFLOP/s", "cores": torch.get_num_threads(),
               "kind": "port", "sample": "1 of 4 batch elements of the workload: (1,8,4096,64) bf16 causal fwd+bwd, "
               "PyTorch CPU port of plain_cosine_sim_attention + autograd, best of 2 after 1 warm-up (%.2f s)" % secs}
        # second leg: the reference's tiled CPU forward ("reference CPU path", py:130-241), forward only, same sample
        t_tiled = P.time_forward(P.tiled_forward_cpu, sample, sample, dt, w["causal"], reps=2, scale=w["scale"])
        cpu["tiled_forward"] = {"value": round(flops(wf, bwd=False) / t_tiled / 1e12, 4), "unit": "TFLOP/s", "seconds": round(t_tiled, 3),
                                "sample": "(1,8,4096,64) bf16 causal forward, port of the reference's tiled CPU path (512 x 512 tiles, f32 inside)"}
        # third leg: BASELINE config C1, the reference's own CPU-runnable case: plain attention forward, (1,8,1024,64) f32
        c1 = dict(B=1, H=8, N=1024, M=1024, D=64, causal=False)
        t_c1 = P.time_forward(P.plain_attention_cpu, (1, 8, 1024, 64), (1, 8, 1024, 64), torch.float32, False, reps=3, scale=8.0)
        cpu["c1_plain_forward"] = {"value": round(flops(c1, bwd=False) / t_c1 / 1e12, 4), "unit": "TFLOP/s", "seconds": round(t_c1, 4),
                                   "sample": "C1: plain_cosine_sim_attention port, forward, (1,8,1024,64) f32 non-causal, best of 3"}

    if rank == 0:
        out = {
            "metric": "attention TFLOP/s fwd+bwd (B=4,H=8,N=4096,D=64 bf16) + max-|delta| vs PyTorch ref",
            "value": round(value, 2), "unit": "TFLOP/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(ms_per_step, 4), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": "bf16", "data": "synthetic",
            "config": {"workload": w["name"], "causal": True, "scale": 8, "groups": 1, "l2norm_qk": True,
                       "per_gpu_batch": w["B"], "parallelism": "replicas x%d (no collective)" % world,
                       "flop_convention": "GEMM only: fwd 4*BHNMD + bwd 10*BHNMD, x causal fraction 4097/8192",
                       "algorithmic_gflop_per_step_per_gpu": round(flops(w) / 1e9, 2)},
            "max_abs_delta_vs_pytorch_f32": max_delta,
            "accuracy": accuracy,
            "roofline": roofline,
            "vs_flash_sdpa": (round(sdpa["ms_per_step"] / ms_per_step, 3) if sdpa and "ms_per_step" in sdpa else None),
            "flash_sdpa": sdpa,
            "cpu_baseline": cpu,
            "kernels": kernels,
            "configs": configs,
            "reference_protocol": ref_proto,
            "small_n_f16_causal_fwdbwd": small,
        }
        print(json.dumps(out))
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
