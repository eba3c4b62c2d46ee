"""The reference's third test (tests/test.py:129-161, `test_output_equal_cuda_and_cpu_forward`): the SAME public entry point called on
device tensors (the HIP kernels) and on `.cpu()` copies (the package's tiled CPU forward, cpu.py <-> py:130-241), same grid --
(causal, mask) x attn_bias x seq_len {63, 127} x dim_head {32, 64, 96, 128} x {f32, f16} x attn_bias_batch_dim x single_head_kv --
plus bf16.  Tolerances: the reference asserts max-abs 1e-4 (f32) / 1e-1 (f16) (test.py:139); here 1.5e-5 / 3e-3 (f16) / 2.4e-2 (bf16):
both sides round q^, k^, P and the output to the 16-bit type, independently."""
import pytest
import torch

pytestmark = pytest.mark.gpu

ATOL = {torch.float32: 1.5e-5, torch.float16: 3e-3, torch.bfloat16: 2.4e-2}      # measured 8.5e-6 / 1.95e-3 / 1.56e-2 (one output ulp at |o| ~ 2 ... 4)


def _grid():
    out = []
    for causal, mask in ((True, False), (False, True), (False, False)):
        for bias in (True, False):
            for n in (63, 127):
                for d in (32, 64, 96, 128):
                    for dt in (torch.float32, torch.float16, torch.bfloat16):
                        for bias_batch in (False, True):
                            for single in (False, True):
                                if bias_batch and not bias:
                                    continue      # (attn_bias_batch_dim without a bias is the same problem)
                                out.append(dict(causal=causal, mask=mask, bias=bias, n=n, d=d, dt=dt, bias_batch=bias_batch, single=single))
    return out


def _id(c):
    return "%s%s%s_n%d_d%d_%s%s%s" % ("causal" if c["causal"] else "full", "_mask" if c["mask"] else "", "_bias" if c["bias"] else "", c["n"], c["d"],
                                      str(c["dt"]).split(".")[-1], "_bb" if c["bias_batch"] else "", "_skv" if c["single"] else "")


@pytest.mark.parametrize("c", _grid(), ids=_id)
def test_output_equal_device_and_cpu_forward(c):
    import flash_cosine_sim_attention_amd as F
    import tolerances as T
    batch, heads, n, d, dt = 4, 8, c["n"], c["d"], c["dt"]
    g = torch.Generator().manual_seed(n * 1000 + d)
    kv_shape = (batch, heads, n, d) if not c["single"] else (batch, n, d)
    q = torch.randn(batch, heads, n, d, generator=g).to(dt)
    k = torch.randn(kv_shape, generator=g).to(dt)
    v = torch.randn(kv_shape, generator=g).to(dt)
    mask = torch.randint(0, 2, (batch, n), generator=g).bool() if c["mask"] else None
    if mask is not None:
        mask[:, 0] = True                                             # (rows without a valid key: the two paths' known divergence)
    bias = torch.randn(batch if c["bias_batch"] else heads, n, n, generator=g).to(dt) if c["bias"] else None
    kw = dict(causal=c["causal"], attn_bias_batch_dim=c["bias_batch"])
    dev = lambda t: None if t is None else t.cuda()
    out_dev = F.flash_cosine_sim_attention(dev(q), dev(k), dev(v), mask=dev(mask), attn_bias=dev(bias), **kw)
    out_cpu = F.flash_cosine_sim_attention(q, k, v, mask=mask, attn_bias=bias, **kw)
    assert out_dev.is_cuda and not out_cpu.is_cuda and out_dev.dtype == out_cpu.dtype == dt
    err = (out_dev.cpu().float() - out_cpu.float()).abs().max().item()
    assert T.check("gpu-vs-cpu-path/max-abs", str(dt).split(".")[-1], err, ATOL[dt]), f"max-abs {err:.3e}"
