"""The C ABI works on CALLER-OWNED buffers (include/fcsa.h): outputs, saved state and workspaces of exactly the sizes the header states.
GPU AddressSanitizer is not available for gfx950 here, so this is the bounds check the suite can make: every buffer of a forward +
backward call is carved out of one arena pre-filled with a byte pattern, with guard bands between the buffers and workspaces of EXACTLY
fcsa_*_workspace_bytes; after the calls (a) every guard byte still holds the pattern -- nothing was written outside a buffer, whatever
the tile tails, split windows, padding lanes (D = 96) or slab layouts did; (b) the inputs are bit-identical (they are `const` in the
ABI); (c) no element of o / dq / dk / dv / d_bias is a NaN: the fill pattern 0xFF makes every unwritten element one, and workspace that
was read before it was written would propagate it.  Shapes are ragged on
purpose and cover the forms with their own address arithmetic: split-key forward + combine, split dQ / dK/dV slabs (causal windows
too), single-headed K/V, bias, key masks, the fused and the slab l2norm paths, every head dim and dtype."""
import ctypes as C

import pytest
import torch

pytestmark = pytest.mark.gpu

DT = {"f16": torch.float16, "bf16": torch.bfloat16, "f32": torch.float32}
FILL = 0xFF          # every 16- and 32-bit float made of it is a NaN: an element that was never written (or workspace read before written) shows as one
GUARD = 4096


class Arena:
    def __init__(self, nbytes):
        self.buf = torch.full((nbytes,), FILL, device="cuda", dtype=torch.uint8)
        self.off = GUARD
        self.used = []          # (offset, nbytes)

    def take(self, shape, dtype):
        n = 1
        for s in shape: n *= s
        nbytes = n * torch.empty((), dtype=dtype).element_size()
        off = (self.off + 255) // 256 * 256
        assert off + nbytes + GUARD <= self.buf.numel(), "arena too small"
        self.used.append((off, nbytes))
        self.off = off + nbytes + GUARD
        return self.buf[off:off + nbytes].view(dtype).view(shape)

    def guards_intact(self):
        keep = torch.ones(self.buf.numel(), device="cuda", dtype=torch.bool)
        for off, n in self.used: keep[off:off + n] = False
        return bool((self.buf[keep] == FILL).all().item())


CASES = [
    # id, dtype, B, H, Hk, N, M, D, kwargs
    ("bf16_d64_ragged_causal", "bf16", 2, 3, 3, 333, 333, 64, dict(causal=True)),
    ("f16_d128_mask_single_kv", "f16", 2, 3, 1, 130, 515, 128, dict(mask=True)),
    ("bf16_d96_one_group_fused", "bf16", 1, 2, 2, 257, 300, 96, dict()),
    ("bf16_d96_two_groups_slabs", "bf16", 1, 2, 2, 200, 129, 96, dict(groups=2, scale=4.0)),
    ("f32_d32_causal_m_gt_n", "f32", 1, 2, 2, 100, 190, 32, dict(causal=True)),
    ("f32_d96_mask", "f32", 1, 2, 2, 70, 131, 96, dict(mask=True)),
    ("f16_d16_groups2", "f16", 2, 2, 2, 129, 65, 16, dict(groups=2)),
    ("bf16_d64_bias_causal", "bf16", 1, 3, 3, 200, 260, 64, dict(bias=True, causal=True)),
    ("f16_d64_bias_batch_mask", "f16", 2, 2, 2, 100, 130, 64, dict(bias=True, bias_batch=True, mask=True)),
    ("bf16_d64_split_forward_keys", "bf16", 1, 2, 2, 40, 2500, 64, dict()),                 # split-key forward + combine, split-key dQ
    ("f16_d64_split_forward_mask", "f16", 1, 4, 4, 300, 2100, 64, dict(mask=True)),
    ("bf16_d64_split_queries_dkv", "bf16", 1, 4, 4, 3000, 200, 64, dict()),                 # split-query dK/dV slabs
    ("bf16_d64_causal_windows", "bf16", 1, 2, 2, 2200, 2200, 64, dict(causal=True)),        # causal split windows in all three kernels
    ("bf16_d128_causal_windows_single_kv", "bf16", 1, 4, 1, 2300, 2300, 128, dict(causal=True)),
    ("f16_d32_causal_windows_m_gt_n", "f16", 1, 3, 3, 1500, 2600, 32, dict(causal=True)),
    ("f16_d64_online_shift", "f16", 1, 2, 2, 300, 450, 64, dict(scale=16.0, causal=True)),
    ("bf16_d128_no_l2norm", "bf16", 1, 2, 2, 129, 257, 128, dict(l2norm=False, scale=1.0)),
    ("bf16_d128_wide_forward", "bf16", 4, 8, 8, 2048, 2048, 128, dict(causal=True)),        # fwd3_kernel + lean backward on a chip-filling grid
    ("bf16_d64_one_row_decode", "bf16", 1, 1, 1, 1, 8192, 64, dict()),
    ("f16_d64_inference_only", "f16", 2, 2, 2, 129, 300, 64, dict(backward=False)),
]


@pytest.mark.parametrize("name,dtype,B,H,Hk,N,M,D,kw", CASES, ids=[c[0] for c in CASES])
def test_calls_stay_inside_their_buffers(name, dtype, B, H, Hk, N, M, D, kw):
    from flash_cosine_sim_attention_amd import _lib
    lib = _lib.load()
    dt = DT[dtype]
    causal, mask, bias = kw.get("causal", False), kw.get("mask", False), kw.get("bias", False)
    bias_batch, l2norm, groups = kw.get("bias_batch", False), kw.get("l2norm", True), kw.get("groups", 1)
    scale, backward = kw.get("scale", 8.0), kw.get("backward", True)
    prob = _lib.problem(dt, (B, H, Hk, N, M, D), causal, bias_batch, l2norm, groups, scale)
    fws_n = int(lib.fcsa_forward_workspace_bytes(C.byref(prob)))
    bws_n = int(lib.fcsa_backward_workspace_bytes(C.byref(prob))) if backward else 0
    es = torch.empty((), dtype=dt).element_size()
    nbias = (B if bias_batch else H) * N * M if bias else 0
    total = (6 * B * H * N * D + 7 * B * Hk * M * D + 2 * nbias) * es + (B * H * N * (1 + groups) + B * Hk * M * groups) * 4 \
        + B * M + fws_n + bws_n + 40 * (GUARD + 256)
    ar = Arena(total)
    g = torch.Generator(device="cuda").manual_seed(hash(name) % 10000)

    def rnd(shape):
        t = ar.take(shape, dt)
        t.copy_(torch.randn(shape, device="cuda", dtype=torch.float32, generator=g).to(dt))
        return t

    q, k, v, do = rnd((B, H, N, D)), rnd((B, Hk, M, D)), rnd((B, Hk, M, D)), rnd((B, H, N, D))
    if not l2norm:
        q.copy_(torch.nn.functional.normalize(q.float(), dim=-1).to(dt))
        k.copy_(torch.nn.functional.normalize(k.float(), dim=-1).to(dt))
    mk = None
    if mask:
        mk = ar.take((B, M), torch.bool)
        mk.copy_(torch.rand((B, M), device="cuda", generator=g) > 0.3)
        mk[:, 0] = True
    ab = None
    if bias:
        ab = ar.take(((B if bias_batch else H), N, M), dt)
        ab.copy_((0.5 * torch.randn(ab.shape, device="cuda", generator=g)).to(dt))
    inputs = [t for t in (q, k, v, do, mk, ab) if t is not None]
    before = [t.clone() for t in inputs]

    o = ar.take((B, H, N, D), dt)
    inv_l = ar.take((B, H, N), torch.float32) if backward else None
    need_qn = bool(lib.fcsa_forward_needs_qn(C.byref(prob), 1 if backward else 0))
    qn = ar.take((B, H, N, D), dt) if need_qn else None
    kn = ar.take((B, Hk, M, D), dt) if l2norm else None
    rq = ar.take((B, H, N, groups), torch.float32) if (l2norm and backward) else None
    rk = ar.take((B, Hk, M, groups), torch.float32) if (l2norm and backward) else None
    fws = ar.take((fws_n,), torch.uint8) if fws_n else None
    ptr = lambda t: None if t is None else t.data_ptr()
    stream = torch.cuda.current_stream().cuda_stream
    norm = _lib.NormState(ptr(qn), ptr(kn), ptr(rq), ptr(rk))
    fa = _lib.ForwardArgs(prob, _lib.tensor4(q), _lib.tensor4(k), _lib.tensor4(v), _lib.tensor4(o), ptr(inv_l), ptr(mk), ptr(ab),
                          norm, ptr(fws), fws_n, stream)
    _lib.check(lib.fcsa_forward(C.byref(fa)), "fcsa_forward")
    outs = {"o": o}
    if backward:
        ws = ar.take((max(bws_n, 1),), torch.uint8)
        dq, dk, dv = ar.take((B, H, N, D), dt), ar.take((B, Hk, M, D), dt), ar.take((B, Hk, M, D), dt)
        db = ar.take(ab.shape, dt) if ab is not None else None
        ba = _lib.BackwardArgs(prob, _lib.tensor4(do), _lib.tensor4(o), ptr(inv_l), _lib.tensor4(q), _lib.tensor4(k), _lib.tensor4(v),
                               ptr(mk), ptr(ab), norm, _lib.tensor4(dq), _lib.tensor4(dk), _lib.tensor4(dv), ptr(db), ws.data_ptr(), bws_n, stream)
        _lib.check(lib.fcsa_backward(C.byref(ba)), "fcsa_backward")
        outs.update(dq=dq, dk=dk, dv=dv)
        if db is not None: outs["d_bias"] = db
    torch.cuda.synchronize()

    assert ar.guards_intact(), "a byte outside the call's buffers was written"
    for t, b in zip(inputs, before):
        assert torch.equal(t, b), "an input buffer was modified"
    for nm, t in outs.items():
        bad = int((~torch.isfinite(t.float())).sum().item())
        assert bad == 0, f"{nm}: {bad} element(s) are NaN -- never written (the arena's fill pattern), or computed from unwritten workspace"


def test_arena_detects_a_stray_write():
    """negative control of the check itself: one byte behind a buffer, one NaN-pattern element left in an output"""
    ar = Arena(64 * 1024)
    t = ar.take((100,), torch.float16)
    t.zero_()
    assert ar.guards_intact()
    off, n = ar.used[0]
    ar.buf[off + n] = 0          # first guard byte behind the buffer
    assert not ar.guards_intact()
    u = Arena(64 * 1024).take((10,), torch.bfloat16)
    assert not torch.isfinite(u.float()).any()          # the fill pattern reads as NaN in every float type the ABI takes


PADDED = [
    # id, dtype, B, H, Hk, N, M, D, kwargs -- every tensor of the call (inputs AND outputs) has rows of D + PAD elements and a head-major memory
    # order ([H, B, N, D + PAD] storage viewed as [B, H, N, D]): the ABI takes element strides per tensor (include/fcsa.h, fcsa_tensor)
    ("bf16_d64_causal_padded_rows", "bf16", 2, 3, 3, 200, 200, 64, dict(causal=True)),
    ("f16_d128_mask_single_kv_padded", "f16", 2, 3, 1, 130, 300, 128, dict(mask=True)),
    ("bf16_d96_padded", "bf16", 1, 2, 2, 140, 170, 96, dict()),
    ("f32_d32_padded", "f32", 1, 2, 2, 100, 130, 32, dict(causal=True)),
    ("bf16_d64_split_layout_padded", "bf16", 1, 2, 2, 40, 2500, 64, dict()),      # (a split-eligible problem: strided layouts must still be right, split or not)
]


@pytest.mark.parametrize("name,dtype,B,H,Hk,N,M,D,kw", PADDED, ids=[c[0] for c in PADDED])
def test_strided_inputs_and_outputs_through_the_c_abi(name, dtype, B, H, Hk, N, M, D, kw):
    """Strided o / dq / dk / dv (and inputs): the padding elements behind every row keep the arena's pattern, and the results equal the
    contiguous call's (bit for bit where both calls take the same form)."""
    from flash_cosine_sim_attention_amd import _lib
    lib = _lib.load()
    dt = DT[dtype]
    PAD = 16
    causal, mask = kw.get("causal", False), kw.get("mask", False)
    prob = _lib.problem(dt, (B, H, Hk, N, M, D), causal, False, True, 1, 8.0)
    fws_n = int(lib.fcsa_forward_workspace_bytes(C.byref(prob)))
    bws_n = int(lib.fcsa_backward_workspace_bytes(C.byref(prob)))
    es = torch.empty((), dtype=dt).element_size()
    per = lambda h, l: B * h * l * (D + PAD) * es
    ar = Arena(2 * (6 * per(H, N) + 7 * per(Hk, M) + (B * H * N * 2 + B * Hk * M) * 4 + fws_n + bws_n) + B * M + 80 * (GUARD + 256))
    g = torch.Generator(device="cuda").manual_seed(hash(name) % 10000)
    stream = torch.cuda.current_stream().cuda_stream
    ptr = lambda t: None if t is None else t.data_ptr()

    def padded(h, l):          # [B, h, l, D] view of [h, B, l, D + PAD] storage
        return ar.take((h, B, l, D + PAD), dt).permute(1, 0, 2, 3)[..., :D]

    def plain(h, l):
        return ar.take((B, h, l, D), dt)

    vals = [torch.randn((B, h, l, D), device="cuda", dtype=torch.float32, generator=g).to(dt) for h, l in ((H, N), (Hk, M), (Hk, M), (H, N))]
    mk = None
    if mask:
        mk = ar.take((B, M), torch.bool)
        mk.copy_(torch.rand((B, M), device="cuda", generator=g) > 0.3)
        mk[:, 0] = True
    results = []
    for make in (plain, padded):
        q, k, v, do = make(H, N), make(Hk, M), make(Hk, M), make(H, N)
        for t, x in zip((q, k, v, do), vals): t.copy_(x)
        o, dq, dk, dv = make(H, N), make(H, N), make(Hk, M), make(Hk, M)
        inv_l = ar.take((B, H, N), torch.float32)
        qn, kn = ar.take((B, H, N, D), dt), ar.take((B, Hk, M, D), dt)
        rq, rk = ar.take((B, H, N, 1), torch.float32), ar.take((B, Hk, M, 1), torch.float32)
        fws = ar.take((fws_n,), torch.uint8) if fws_n else None
        ws = ar.take((max(bws_n, 1),), torch.uint8)
        norm = _lib.NormState(ptr(qn), ptr(kn), ptr(rq), ptr(rk))
        t4 = lambda t: _lib.Tensor(t.data_ptr(), t.stride(0), t.stride(1), t.stride(2))
        fa = _lib.ForwardArgs(prob, t4(q), t4(k), t4(v), t4(o), ptr(inv_l), ptr(mk), None, norm, ptr(fws), fws_n, stream)
        _lib.check(lib.fcsa_forward(C.byref(fa)), "fcsa_forward")
        ba = _lib.BackwardArgs(prob, t4(do), t4(o), ptr(inv_l), t4(q), t4(k), t4(v), ptr(mk), None, norm, t4(dq), t4(dk), t4(dv), None,
                               ws.data_ptr(), bws_n, stream)
        _lib.check(lib.fcsa_backward(C.byref(ba)), "fcsa_backward")
        torch.cuda.synchronize()
        results.append([t.clone() for t in (o, dq, dk, dv)])
    # the padding behind every row was carved out of "used" ranges: check it explicitly, then the guards between buffers
    raw = ar.buf
    assert ar.guards_intact(), "a byte outside the call's buffers was written"
    # padding columns of the padded tensors: re-derive them from the arena by scanning every (h, B, l, D + PAD) block taken by `padded`
    for (off, n) in ar.used:
        blk = raw[off:off + n]
        for h, l in ((H, N), (Hk, M)):
            if n == per(h, l):
                pad = blk.view(dt).view(h, B, l, D + PAD)[..., D:].contiguous().view(torch.uint8)
                assert bool((pad == FILL).all().item()), "a padding element behind a row was written"
    for nm, a, b in zip(("o", "dq", "dk", "dv"), results[0], results[1]):
        assert torch.isfinite(a.float()).all(), nm
        # (bit-identical where both calls take the same form; a layout without stride0 == heads * stride1 runs the UN-split backward /
        #  forward -- include/fcsa.h -- and then differs from the contiguous call's split form by the order of f32 partial sums)
        rel = ((a.float() - b.float()).norm() / b.float().norm().clamp_min(1e-30)).item()
        assert rel <= {"f32": 1e-5, "f16": 1e-3, "bf16": 6e-3}[dtype], f"{nm}: the strided call differs from the contiguous one (rel-L2 {rel:.2e})"
        if "split" not in name and "single_kv" not in name:
            assert torch.equal(a, b), f"{nm}: same form, but not bit-identical"
