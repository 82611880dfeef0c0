"""GPU parity tests of the prior's training operators (through the C-ABI): the
hand-written backward kernels against torch autograd of the specification in
oracle/prior_oracle.py on the CPU (layers: parity unpinned, the reference's layer
package is absent; LabelSmoothingLoss: pinned by the fixture generated from the
reference class).  Tolerance: 2e-4 of the gradient tensor's max for fp32."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 2e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close(a, b, tol=TOL, what=""):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, f"{what}: {tuple(a.shape)} vs {tuple(b.shape)}"
    assert torch.isfinite(a).all(), f"{what}: non-finite values"
    err = (a - b).abs().max() / b.abs().max().clamp(min=1e-12)
    assert err <= tol, f"{what}: max err / max|ref| = {err:.3e}"


def _spec_attention(q, k, v, rel, H, Cq, Ck, Ek, mask):
    from oracle import prior_oracle as P
    Sq, B, d = q.shape
    Sk = k.shape[0]
    hd = d // H
    hq = q.reshape(Sq, B, H, hd).permute(1, 2, 0, 3)
    hk = k.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    hv = v.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    logits = hq @ hk.transpose(-1, -2)
    if rel is not None:
        qe = torch.einsum("bhid,hrd->bhir", hq, rel)
        idx = P.rel_index(Sq, Sk, Cq, Ck, Ek)
        logits = logits + qe.gather(3, idx.expand(B, H, Sq, Sk))
    logits = logits / math.sqrt(hd)
    if mask is not None:
        logits = logits + mask
    return (torch.softmax(logits, -1) @ hv).permute(2, 0, 1, 3).reshape(Sq, B, d)


@pytest.mark.parametrize("hd,H,Sq,Sk,Cq,Ck,mode,bias", [
    (16, 4, 33, 33, 1, 1, 1, True), (16, 4, 33, 33, 1, 1, 2, True), (32, 2, 132, 33, 4, 1, 0, True),
    (64, 2, 200, 200, 1, 1, 1, True), (32, 3, 260, 260, 4, 4, 1, True), (64, 2, 77, 150, 2, 1, 0, True),
    (64, 2, 150, 150, 1, 1, 1, False), (32, 2, 40, 70, 1, 2, 0, False),
])
def test_rel_attention_backward_against_spec(hd, H, Sq, Sk, Cq, Ck, mode, bias):
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors._train import RelAttentionFn
    torch.manual_seed(hd + Sq + mode)
    d, B = hd * H, 2
    Eq, Ek = -(-Sq // Cq), -(-Sk // Ck)
    self_attn = Sq == Sk and Cq == Ck
    if self_attn:
        a = torch.randn(Sq, B, 3 * d, requires_grad=True)
        b = None
        q, k, v = a[..., :d], a[..., d:2 * d], a[..., 2 * d:]
    else:
        a = torch.randn(Sq, B, d, requires_grad=True)
        b = torch.randn(Sk, B, 2 * d, requires_grad=True)
        q, k, v = a, b[..., :d], b[..., d:]
    rel = (torch.randn(H, Eq + Ek - 1, hd) * 0.5).requires_grad_(True) if bias else None
    mask = None
    if mode == 1:
        mask = P.causal_mask(Sq)
    elif mode == 2:
        mask = P.causal_mask(Sq).t()
    w = torch.randn(Sq, B, d)
    ref = _spec_attention(q, k, v, rel, H, Cq, Ck, Ek, mask)
    (ref * w).sum().backward()

    dev = _dev()
    ga = a.detach().to(dev).requires_grad_(True)
    gb = b.detach().to(dev).requires_grad_(True) if b is not None else None
    grel = rel.detach().to(dev).requires_grad_(True) if bias else None
    got = RelAttentionFn.apply(ga, gb, grel, H, Cq, Ck, Ek, mode, None)
    _close(got, ref, 1e-4, "forward")
    (got * w.to(dev)).sum().backward()
    _close(ga.grad, a.grad, TOL, "d(q|k|v)" if self_attn else "dq")
    if gb is not None:
        _close(gb.grad, b.grad, TOL, "d(k|v)")
    if bias:
        _close(grel.grad, rel.grad, TOL, "d rel_embeddings")
    # dense additive mask path
    if mask is not None:
        ga2 = a.detach().to(dev).requires_grad_(True)
        grel2 = rel.detach().to(dev).requires_grad_(True) if bias else None
        got2 = RelAttentionFn.apply(ga2, None, grel2, H, Cq, Ck, Ek, 0, mask.to(dev).contiguous())
        (got2 * w.to(dev)).sum().backward()
        _close(ga2.grad, a.grad, TOL, "dense mask: d(q|k|v)")
        if bias:
            _close(grel2.grad, rel.grad, TOL, "dense mask: d rel_embeddings")
    # bit-reproducible
    ga3 = a.detach().to(dev).requires_grad_(True)
    gb3 = b.detach().to(dev).requires_grad_(True) if b is not None else None
    grel3 = rel.detach().to(dev).requires_grad_(True) if bias else None
    (RelAttentionFn.apply(ga3, gb3, grel3, H, Cq, Ck, Ek, mode, None) * w.to(dev)).sum().backward()
    assert torch.equal(ga3.grad, ga.grad), "attention backward is not deterministic"
    if bias and Cq == 1 and Ck == 1:
        assert torch.equal(grel3.grad, grel.grad), "d rel_embeddings is not deterministic"


def test_linear_layernorm_loss_backward_against_torch(golden_dir):
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors._train import LinearFn, LayerNormFn, label_smoothing_loss
    dev = _dev()
    torch.manual_seed(3)
    # linear (+relu, +residual), K multiple of 32 and not
    for K, N, relu, res in ((96, 50, True, True), (40, 64, False, False), (512, 2048, True, False)):
        x = torch.randn(37, 3, K, requires_grad=True)
        W = (torch.randn(N, K) * 0.1).requires_grad_(True)
        b = torch.randn(N, requires_grad=True)
        r = torch.randn(37, 3, N, requires_grad=True) if res else None
        y = torch.nn.functional.linear(x, W, b)
        if res:
            y = y + r
        if relu:
            y = torch.relu(y)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        gx, gW, gb_ = (t.detach().to(dev).requires_grad_(True) for t in (x, W, b))
        gr = r.detach().to(dev).requires_grad_(True) if res else None
        packed = _ops.pack_linear_weight(gW)
        got = LinearFn.apply(gx, gW, gb_, gr, relu, packed,
                             lambda: _ops.pack_linear_weight(gW.detach().t().contiguous()))
        _close(got, y, 1e-5, "linear fwd")
        (got * w.to(dev)).sum().backward()
        _close(gx.grad, x.grad, TOL, "linear dx")
        _close(gW.grad, W.grad, TOL, "linear dW")
        _close(gb_.grad, b.grad, TOL, "linear db")
        if res:
            _close(gr.grad, r.grad, TOL, "linear dres")
    # layernorm (+residual)
    for D, rows, res in ((96, (37, 3), True), (512, (130, 2), False)):
        x = torch.randn(*rows, D, requires_grad=True)
        r = torch.randn(*rows, D, requires_grad=True) if res else None
        g = torch.randn(D, requires_grad=True)
        be = torch.randn(D, requires_grad=True)
        y = torch.nn.functional.layer_norm(x + r if res else x, (D,), g, be, 1e-5)
        w = torch.randn_like(y)
        (y * w).sum().backward()
        gx, gg, gbe = (t.detach().to(dev).requires_grad_(True) for t in (x, g, be))
        gr = r.detach().to(dev).requires_grad_(True) if res else None
        got = LayerNormFn.apply(gx, gr, gg, gbe, 1e-5)
        _close(got, y, 1e-5, "layernorm fwd")
        (got * w.to(dev)).sum().backward()
        _close(gx.grad, x.grad, TOL, "layernorm dx")
        _close(gg.grad, g.grad, TOL, "layernorm dgamma")
        _close(gbe.grad, be.grad, TOL, "layernorm dbeta")
        if res:
            _close(gr.grad, r.grad, TOL, "layernorm dres")
    # label smoothing: fixture from the reference class (dim=1, [B,K,F,T]) + autograd of the oracle
    from oracle import prior_oracle as P
    z = np.load(golden_dir / "prior_wrapper.npz")
    pred = torch.from_numpy(z["ls::pred"])
    tgt = torch.from_numpy(z["ls::target"])
    for sm in ("0.1", "0.0"):  # LabelSmoothingLoss(32, sm, dim=1) of the reference (oracle/make_golden.py)
        got = label_smoothing_loss(pred.to(dev), tgt.to(dev), 32, float(sm), dim=1)
        _close(got, z["ls::loss_" + sm], 1e-5, "label smoothing (reference fixture)")
    pred = torch.randn(3, 17, 5, 4, requires_grad=True)
    tgt = torch.randint(0, 17, (3, 5, 4))
    for sm in (0.0, 0.1):
        ref = P.label_smoothing_loss(pred, tgt, 17, sm, dim=1)
        pred.grad = None
        (ref * 1.7).backward()
        gp = pred.detach().to(dev).requires_grad_(True)
        got = label_smoothing_loss(gp, tgt.to(dev), 17, sm, dim=1)
        _close(got, ref, 1e-5, "label smoothing loss")
        (got * 1.7).backward()
        _close(gp.grad, pred.grad, TOL, "label smoothing gradient")
