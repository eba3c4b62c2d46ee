"""CPU-tensor behaviour of the operator (reference: flash_cosine_sim_attention.py:322-323 -> py:130-241): the package's own
forward-only blockwise path in flash_cosine_sim_attention_amd/cpu.py, checked against the reference's `o_tiled` /
`o_plain` golden fixtures and the numpy oracle.  CPU only; the oracle is the checker, never the thing tested."""
import os

import numpy as np
import pytest
import torch

import cases as C
import flash_cosine_sim_attention_amd as F
from flash_cosine_sim_attention_amd import cpu as P

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
TOL = {"f32": 2e-5, "f16": 2e-3, "bf16": 1.6e-2}       # output rounding of the 16-bit types dominates


def _valid_rows(case, inp):
    n, m, b = case["n"], case["m"], case["b"]
    ok = np.ones((b, n), dtype=bool)
    if case["causal"]:
        ok &= (np.arange(n)[None, :] + (m - n)) >= 0
    if inp["mask"] is not None:
        ok &= inp["mask"].numpy().any(-1)[:, None]
    return ok if case["merged"] else ok[:, None, :]


@pytest.mark.parametrize("case", C.CASES, ids=lambda c: c["name"])
def test_cpu_path_matches_reference_fixture(case):
    inp = C.make_inputs(case)
    gold = dict(np.load(os.path.join(GOLD, case["name"] + ".npz")))
    o = F.flash_cosine_sim_attention(inp["q"], inp["k"], inp["v"], mask=inp["mask"], attn_bias=inp["attn_bias"], **C.op_kwargs(case))
    assert o.shape == inp["q"].shape and o.dtype == inp["q"].dtype and o.device.type == "cpu"
    got = o.double().numpy()
    ok = np.broadcast_to(_valid_rows(case, inp)[..., None], got.shape)
    ref = gold["o_tiled"] if case["tiled_ok"] else gold["o_plain"]     # o_tiled: the reference's own CPU path, where it is usable
    # (`wide` cases: the dtype rounding of q^, k^ -- this path normalises in the input dtype like the reference, py:57-65 -- is
    #  amplified by the logit range scale * groups; 1 for every other case)
    cond = C.logit_cond(case["dtype"], case["scale"], case["groups"], case["l2norm"])
    assert np.abs(np.where(ok, got - ref, 0.0)).max() <= TOL[case["dtype"]] * cond * (1.25 if cond > 1 else 1.0)      # (max-abs over 2e4 values)
    assert np.abs(np.where(ok, 0.0, got)).max() == 0.0                 # rows without a valid key are exactly 0


@pytest.mark.parametrize("blocks", [(256, 1024), (7, 13), (64, 64), (1000, 5)])
def test_cpu_path_is_block_size_invariant_and_skips_the_right_causal_blocks(blocks):
    """Causal N > 512 is where the reference's tiled path goes wrong (its block skip is inverted, py:215); the result here
    must equal the O(N*M) composite for every block shape."""
    torch.manual_seed(1)
    q, k, v = torch.randn(1, 2, 700, 32), torch.randn(1, 2, 900, 32), torch.randn(1, 2, 900, 32)
    ref = F.plain_cosine_sim_attention(q.double(), k.double(), v.double(), causal=True, scale=6, groups=2)
    o = P.attention_forward_cpu(q, k, v, causal=True, scale=6, groups=2, row_block=blocks[0], key_block=blocks[1])
    assert (o.double() - ref).abs().max() <= 2e-5


def test_cpu_path_single_head_kv_mask_bias_merged():
    torch.manual_seed(2)
    q = torch.randn(3, 4, 50, 16)
    k, v = torch.randn(3, 70, 16), torch.randn(3, 70, 16)
    mask = torch.rand(3, 70) > 0.3
    bias = torch.randn(4, 50, 70)
    ref = F.plain_cosine_sim_attention(q.double(), k.double(), v.double(), mask=mask, attn_bias=bias.double())
    o = F.flash_cosine_sim_attention(q, k, v, mask=mask, attn_bias=bias)
    assert (o.double() - ref).abs().max() <= 2e-5
    qm, km, vm = torch.randn(6, 20, 32), torch.randn(6, 33, 32), torch.randn(6, 33, 32)
    bm = torch.randn(6, 20, 33)
    ref = F.plain_cosine_sim_attention(qm.double(), km.double(), vm.double(), attn_bias=bm.double())
    assert (F.flash_cosine_sim_attention(qm, km, vm, attn_bias=bm).double() - ref).abs().max() <= 2e-5


def test_cpu_path_is_forward_only_and_validates():
    q = torch.randn(1, 2, 8, 16)
    with pytest.raises(RuntimeError, match="forward-only"):
        F.flash_cosine_sim_attention(q.clone().requires_grad_(), q, q)
    with pytest.raises(ValueError):
        F.flash_cosine_sim_attention(q, q, q, mask=torch.ones(1, 8, dtype=torch.bool), causal=True)
    with pytest.raises(ValueError):
        F.flash_cosine_sim_attention(q, q, q, groups=3)


def test_product_cpu_path_does_not_touch_the_oracle():
    import inspect
    from flash_cosine_sim_attention_amd import ops
    for mod in (P, ops):
        assert "oracle" not in inspect.getsource(mod).replace("`oracle/`", "")


def test_l2norm_exports_match_reference_fixture():
    """l2norm_tensors (public export): golden qn / kn of the reference, incl. groups."""
    for name in ("g01_dense_d64_n63_f32", "g16_groups2_scale1_d64_f32"):
        case = C.BY_NAME[name]
        inp = C.make_inputs(case)
        gold = dict(np.load(os.path.join(GOLD, name + ".npz")))
        qn, kn = F.l2norm_tensors(inp["q"], inp["k"], groups=case["groups"])
        assert np.abs(qn.double().numpy() - gold["qn"]).max() <= 1e-6
        assert np.abs(kn.double().numpy() - gold["kn"]).max() <= 1e-6
