"""GPU parity tests of the prior's training operators (through the C-ABI): the
hand-written backward kernels against torch autograd of the specification in
oracle/prior_oracle.py on the CPU (layers: parity unpinned, the reference's layer
package is absent; LabelSmoothingLoss: pinned by the fixture generated from the
reference class).  Tolerance: 2e-4 of the gradient tensor's max for fp32."""
import math

import numpy as np
import pytest
import torch


def _free_port() -> int:
    """A TCP port nobody listens on right now (asked from the OS) for a rendezvous."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]


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
    # degenerate / boundary shapes: one query (over two keys: with a single key every gradient but dV is
    # identically zero and a relative comparison is meaningless), exactly one tile, a tile boundary + 1
    (16, 2, 1, 2, 1, 1, 0, True), (64, 1, 1, 97, 1, 1, 0, True), (32, 2, 64, 64, 1, 1, 2, True),
    # one or two rows / keys beyond the last full 128-row block: unmasked -> the one-row / one-key kernels (with and
    # without relative logits, every head dim, Cq = 2 with a ragged key count, B H not a multiple of 8); causal ->
    # the ragged query block is block 0; the banded G products with rows ordered (query, batch)
    (32, 3, 257, 257, 1, 1, 0, True), (16, 5, 258, 130, 2, 1, 0, True), (64, 3, 385, 257, 1, 1, 0, True),
    (64, 3, 257, 258, 1, 1, 0, False), (64, 3, 257, 257, 1, 1, 1, True), (32, 2, 300, 300, 1, 1, 2, True),
    (64, 2, 129, 129, 1, 1, 1, True),
])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_rel_attention_backward_against_spec(hd, H, Sq, Sk, Cq, Ck, mode, bias, precision, monkeypatch):
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors._train import RelAttentionFn
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", precision)   # exact-fp32 and split-bf16 kernels, one tolerance
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


# ------------------------------------------------------------------ whole model
COMMON = dict(n_class=32, channel=8, kernel_size=5, n_block=1, n_res_block=1, res_channel=8,
              d_model=64, embeddings_dim=8, positional_embeddings_dim=8,
              use_relative_transformer=True, predict_frequencies_first=True,
              conditional_model=True, class_conditioning_prepend_to_dummy_input=True,
              class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
              class_conditioning_embedding_dim_per_modality={"instrument_family_str": 16, "pitch": 16},
              conditional_model_nhead=4, conditional_model_num_encoder_layers=2,
              conditional_model_num_decoder_layers=3)


def _cpu_prepare(model, P, code_map, kind, cls, mask_map=None):
    """Plain-torch restatement of VQNSynthTransformer.prepare_data (priors/transformer.py:419-680) on
    leaf copies P of the parameters: Embedding -> Linear -> (+ positional embeddings) -> start symbol
    with the class-conditioning embeddings written into its leading dims."""
    F_ = torch.nn.functional
    src = kind == "source"
    helper = model.source_codemaps_helper if src else model.target_codemaps_helper
    seq = helper.to_sequence(code_map)
    if mask_map is not None:
        seq = seq.masked_fill(helper.to_sequence(mask_map), model.mask_token_index)
    emb = F_.linear(F_.embedding(seq, P[f"{kind}_embed.weight"]), P[f"{kind}_embeddings_linear.weight"],
                    P[f"{kind}_embeddings_linear.bias"])
    if src:
        freq = P["source_positional_embeddings_frequency"].repeat(1, 1, model.source_duration, 1)
        pos = torch.cat([freq, freq], 3).reshape(1, model.source_frequencies, model.source_duration, -1)
    else:
        freq = P["target_positional_embeddings_frequency"].repeat(1, 1, model.target_duration, 1)
        patch = P["target_positional_embeddings_patch"].repeat(1, model.source_frequencies, model.source_duration, 1)
        pos = torch.cat([freq, patch], 3).reshape(1, model.target_frequencies, model.target_duration, -1)
    B = code_map.shape[0]
    x = torch.cat([emb, helper.to_sequence(pos).expand(B, -1, -1)], 2)
    start = P[f"{kind}_start_symbol"].repeat(B, 1, 1)
    for name, c in cls.items():
        e = F_.embedding(c, P[f"class_conditioning_embedding_layers.{name}.weight"]).squeeze(1)
        p0 = model.class_conditioning_start_positions_per_modality[name]
        start[:, :, p0:p0 + e.shape[1]] = e.unsqueeze(1)
    return torch.cat([start, x], 1)


# BASELINE config 4's model: d_model 512, 6 + 8 layers, 8 heads, 512 classes, [32,32] top map = 1025-token sequences
FULL = dict(COMMON, n_class=512, channel=256, n_block=4, n_res_block=4, res_channel=256, d_model=512,
            embeddings_dim=32, positional_embeddings_dim=16,
            class_conditioning_embedding_dim_per_modality={"instrument_family_str": 64, "pitch": 64},
            conditional_model_nhead=8, conditional_model_num_encoder_layers=6, conditional_model_num_decoder_layers=8)


@pytest.mark.parametrize("level,linear_precision,size", [
    ("top", "f32", "toy"), ("top", "bf16x6", "toy"), ("bottom", "f32", "toy"), ("bottom", "bf16x6", "toy"),
    ("top", "f32", "full"), ("top", "bf16x6", "full"), ("bottom", "bf16x6", "full")])
def test_prior_training_step_gradients_against_spec(level, linear_precision, size, monkeypatch):
    """loss.backward() of one training batch (train_autoregressive_model.py:178-257 semantics, dropout 0):
    every parameter gradient of the HIP path against torch autograd of the CPU specification.  size 'full' =
    BASELINE config 4's priors at B = 1: top (seq 1025, d_model 512, 6 + 8 layers) and bottom ([64,64] map: 4100
    target rows, 4 tokens per event, 4100 x 4100 causal self-attention and 4100 x 1025 cross-attention)."""
    from oracle import prior_oracle as P_
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors.transformer import (
        SelfAttentiveVQTransformer, UpsamplingVQTransformer)
    from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
    monkeypatch.setattr(_ops, "LINEAR_PRECISION", linear_precision)
    torch.manual_seed(7)
    full = size == "full"
    cfg, st, sb, K = (FULL, [32, 32], [64, 64], 512) if full else (COMMON, [8, 4], [16, 8], 32)
    if level == "top":
        model = SelfAttentiveVQTransformer(shape=st, condition_shape=st, self_conditional_model=True,
                                           add_mask_token_to_symbols=True, **cfg)
    else:
        model = UpsamplingVQTransformer(shape=sb, condition_shape=st, **cfg)
    for m in model.modules():
        if hasattr(m, "dropout") and isinstance(m.dropout, float):
            m.dropout = 0.0
    B = 1 if full else 3
    g = torch.Generator().manual_seed(8)
    cls = {"instrument_family_str": torch.randint(0, 11, (B, 1), generator=g),
           "pitch": torch.randint(0, 61, (B, 1), generator=g)}
    top = torch.randint(0, K, (B, *st), generator=g)
    bottom = torch.randint(0, K, (B, *sb), generator=g)
    mask = torch.rand(B, *st, generator=g) < 0.5
    target, cond = (top, top) if level == "top" else (bottom, top)

    # ---- CPU specification with autograd
    P = {k: v.detach().clone().requires_grad_(True) for k, v in model.named_parameters()}
    src = _cpu_prepare(model, P, cond, "source", cls, mask if level == "top" else None)
    tgt = _cpu_prepare(model, P, target, "target", cls)
    H = model.conditional_model_nhead
    Ce, Ee = model.source_num_channels, model.source_num_events_with_start_symbol
    Cd, Ed = model.target_num_channels, model.target_num_events_with_start_symbol
    s, t = src.transpose(0, 1), tgt.transpose(0, 1)
    enc_mask = P_.causal_mask(s.shape[0]).t() if model.self_conditional_model else None
    memory = P_.encoder(s, P, "transformer.encoder.", model.conditional_model_num_encoder_layers, H, Ce, Ee, enc_mask)
    out = P_.decoder(t, memory, P, "transformer.decoder.", model.conditional_model_num_decoder_layers, H,
                     Cd, Ed, Ce, Ee, P_.causal_mask(t.shape[0]), None)
    start = model.target_start_symbol.shape[1]
    logits = torch.nn.functional.linear(out[start - 1:-1].transpose(0, 1),
                                        P["project_transformer_outputs_to_logits.weight"],
                                        P["project_transformer_outputs_to_logits.bias"])
    ref_map = model.target_codemaps_helper.to_time_frequency_map(logits, permute_output_as_logits=True)
    ref_loss = P_.label_smoothing_loss(ref_map, target, K, 0.1, dim=1)
    ref_loss.backward()

    # ---- HIP path
    dev = _dev()
    model = model.to(dev).train()
    dcls = {k: v.to(dev) for k, v in cls.items()}
    src_g, tgt_g = model.to_sequences(target.to(dev), condition=cond.to(dev), class_conditioning=dcls,
                                      mask=mask.to(dev) if level == "top" else None)
    _close(src_g, src, 1e-5, "prepared source")
    _close(tgt_g, tgt, 1e-5, "prepared target")
    logits_g, _ = model(tgt_g, condition=src_g)
    _close(logits_g, logits, 5e-4 if full else 1e-4, "logits")
    loss = LabelSmoothingLoss(K, 0.1, dim=1)(
        model.to_time_frequency_map(logits_g, kind="target", permute_output_as_logits=True), target.to(dev))
    _close(loss, ref_loss, 1e-5, "loss")
    loss.backward()
    checked = 0
    for name, p in model.named_parameters():
        ref = P[name].grad
        if ref is None:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name
            continue
        assert p.grad is not None, f"{name}: no gradient"
        if linear_precision == "f32" and not full:
            # relative-embedding gradients: long cancelling sums, ~sqrt(B S) fp32 roundings on either side
            _close(p.grad, ref, 2e-3 if name.endswith("rel_embeddings") else 5e-4, name)
        else:
            # A feed-forward pre-activation within rounding of zero may be rectified differently by two correct
            # fp32-grade implementations (expected for a couple of the 2.4 M activations here); one such unit moves
            # single rows of the neighbouring gradients by percents.  The six-term GEMMs are therefore compared in
            # the root-mean-square sense; their element-wise accuracy is pinned by the 'f32' run of this test
            # together with test_linear_precisions_against_fp64.
            g, r = p.grad.detach().double().cpu(), ref.double()
            assert torch.isfinite(g).all(), name
            rms = float((g - r).pow(2).mean().sqrt() / r.pow(2).mean().sqrt().clamp(min=1e-30))
            assert rms <= 3e-3, f"{name}: rms error / rms|ref| = {rms:.3e}"
        checked += 1
    assert checked >= 60
    # inference is untouched by the training machinery: eval forward records nothing
    model.eval()
    lg, _ = model(tgt_g.detach(), condition=src_g.detach())
    assert not lg.requires_grad
    _close(lg, logits, 5e-4 if full else 1e-4, "eval logits")


def test_run_model_epoch_decreases_loss():
    """`run_model` (train_autoregressive_model.py:119-372): two epochs over a tiny fixed dataset with
    Adam reduce the loss; returned sums are sample-weighted."""
    import argparse
    import train_autoregressive_model as T
    from interactive_spectrogram_inpainting.priors.sequence_mask import BernoulliSequenceMask
    from interactive_spectrogram_inpainting.priors.transformer import SelfAttentiveVQTransformer
    from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
    dev = _dev()
    torch.manual_seed(11)
    model = SelfAttentiveVQTransformer(shape=[8, 4], condition_shape=[8, 4], self_conditional_model=True,
                                       add_mask_token_to_symbols=True, **COMMON).to(dev)
    data = T.SyntheticCodes(12, [8, 4], [16, 8], 32, {"instrument_family_str": 11, "pitch": 61}, seed=1)
    loader = torch.utils.data.DataLoader(data, batch_size=5, shuffle=False)
    opt = torch.optim.Adam(model.parameters(), lr=2e-3)
    crit = LabelSmoothingLoss(32, 0.05, dim=1)
    sampler = BernoulliSequenceMask(0.5, sequence_duration=model.source_transformer_sequence_length,
                                    mask_token_index=model.mask_token_index)
    args = argparse.Namespace(hier="top")
    losses = []
    for epoch in range(4):
        loss_sum, acc_sum, n = T.run_model(args, epoch, loader, model, opt, None, dev, crit, is_training=True,
                                           mask_sampler=sampler, clip_grad_norm=1.0)
        assert n == 12 and 0.0 <= acc_sum / n <= 1.0
        losses.append(loss_sum / n)
    assert losses[-1] < losses[0], losses
    v_loss, v_acc, n = T.run_model(args, 0, loader, model, opt, None, dev, crit, is_training=False, mask_sampler=sampler)
    assert n == 12 and math.isfinite(v_loss)
    assert all(p.grad is None or torch.isfinite(p.grad).all() for p in model.parameters())


def test_linear_precisions_against_fp64():
    """The GEMM behind every Linear of the prior, in each product mode, against float64: the six-term
    split must be at least as accurate as the fp32 matrix pipe."""
    from interactive_spectrogram_inpainting.priors import _ops
    dev = _dev()
    torch.manual_seed(0)
    saved = _ops.LINEAR_PRECISION
    try:
        for M, K, N in ((387, 2048, 64), (1030, 512, 1536), (129, 2048, 64), (100, 256, 40)):
            x, W, b = torch.randn(M, K), torch.randn(N, K) / K ** 0.5, torch.randn(N)
            ref = x.double() @ W.double().t() + b.double()
            err = {}
            for prec in ("f32", "bf16x6", "bf16x3", "f16x3"):
                _ops.LINEAR_PRECISION = prec
                y = _ops.linear(x.to(dev), _ops.pack_linear_weight(W.to(dev), range_check="now"), b.to(dev), N)
                err[prec] = float((y.double().cpu() - ref).abs().max() / ref.abs().max())
            assert err["f32"] <= 5e-6 and err["bf16x6"] <= 1.5 * err["f32"] + 1e-7 and err["bf16x3"] <= 2e-5, err
            assert err["f16x3"] <= 1.5 * err["f32"] + 1e-7, err      # the default: fp32-grade
        # a weight matrix beyond the f16 pieces' range is recognised at pack time and runs the six-term bf16 split
        _ops.LINEAR_PRECISION = "f16x3"
        x, W = torch.randn(64, 256), torch.randn(40, 256)
        W[3, 7] = 100.0
        pw = _ops.pack_linear_weight(W.to(dev), range_check="now")
        assert pw.isi_f16_ok is False
        # the monitor of a weight under training: exact limit for inference weights, HALF the limit (and no device
        # read-back until ~256 versions later) while it is being trained
        wp = torch.nn.Parameter(torch.randn(40, 256, device=dev) * 0.1)
        mon = _ops.WeightRange()
        assert mon.update(wp, False) is True and mon.update(wp, True) is True
        with torch.no_grad():
            wp[3, 7] = 40.0
        assert mon.update(wp, True) is True and _ops.WeightRange().update(wp, False) is False
        with torch.no_grad():
            wp[3, 7] = 100.0
        assert mon.update(wp, True) is False
        y = _ops.linear(x.to(dev), pw, None, 40)
        ref = x.double() @ W.double().t()
        assert torch.isfinite(y).all() and float((y.double().cpu() - ref).abs().max() / ref.abs().max()) < 5e-6
    finally:
        _ops.LINEAR_PRECISION = saved


@pytest.mark.parametrize("Sq,Sk,Cq,Ck,mode", [(1025, 1025, 1, 1, 1), (1025, 1025, 1, 1, 2), (4100, 1025, 4, 1, 0),
                                              (4100, 4100, 4, 4, 1)])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_rel_attention_backward_at_baseline_sizes(Sq, Sk, Cq, Ck, mode, precision, monkeypatch):
    """BASELINE config 4 shapes (head_dim 64, 8 heads): S = 1025 causal / anti-causal self-attention, the bottom
    prior's 4100 x 1025 cross-attention with 4 tokens per event and its 4100 x 4100 causal self-attention: forward and
    every gradient against autograd of the specification."""
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors._train import RelAttentionFn
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", precision)
    hd, H, B = 64, 8, 1
    d = hd * H
    torch.manual_seed(Sq + mode)
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
    rel = (torch.randn(H, Eq + Ek - 1, hd) * 0.5).requires_grad_(True)
    mask = P.causal_mask(Sq) if mode == 1 else P.causal_mask(Sq).t() if mode == 2 else None
    w = torch.randn(Sq, B, d)
    ref = _spec_attention(q, k, v, rel, H, Cq, Ck, Ek, mask)
    (ref * w).sum().backward()
    dev = _dev()
    ga = a.detach().to(dev).requires_grad_(True)
    gb = b.detach().to(dev).requires_grad_(True) if b is not None else None
    grel = rel.detach().to(dev).requires_grad_(True)
    got = RelAttentionFn.apply(ga, gb, grel, H, Cq, Ck, Ek, mode, None)
    _close(got, ref, 1e-4, "forward")
    (got * w.to(dev)).sum().backward()
    _close(ga.grad, a.grad, TOL, "d(q|k|v)" if self_attn else "dq")
    if gb is not None:
        _close(gb.grad, b.grad, TOL, "d(k|v)")
    # a relative-embedding row collects ~S (x Cq Ck) contributions of either sign: fp32 summation-order noise
    _close(grel.grad, rel.grad, 1e-3, "d rel_embeddings")


@pytest.mark.parametrize("Sq,Sk,Cq,Ck,mode", [(1025, 1025, 1, 1, 1), (260, 260, 4, 4, 1), (132, 33, 4, 1, 0)])
def test_rel_attention_backward_bf16_mode(Sq, Sk, Cq, Ck, mode, monkeypatch):
    """precision = 'bf16' (north_star config 4: relative attention on "MFMA bf16"): the backward runs single-term bf16
    products like the forward of that mode (csrc/rel_attention_bwd_f32.hip, ONE = true).  Operands carry 2^-9 relative
    rounding: every gradient within 3e-2 (rms) of autograd of the specification, and less accurate than -- i.e.
    actually different from -- the three-term mode."""
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors._train import RelAttentionFn
    hd, H, B = 64, 8, 1
    d = hd * H
    torch.manual_seed(Sq + 3 * mode)
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
    rel = (torch.randn(H, Eq + Ek - 1, hd) * 0.5).requires_grad_(True)
    mask = P.causal_mask(Sq) if mode == 1 else None
    w = torch.randn(Sq, B, d)
    ref = _spec_attention(q, k, v, rel, H, Cq, Ck, Ek, mask)
    (ref * w).sum().backward()
    dev = _dev()

    def run(prec):
        monkeypatch.setattr(_ops, "ATTENTION_PRECISION", prec)
        ga = a.detach().to(dev).requires_grad_(True)
        gb = b.detach().to(dev).requires_grad_(True) if b is not None else None
        grel = rel.detach().to(dev).requires_grad_(True)
        got = RelAttentionFn.apply(ga, gb, grel, H, Cq, Ck, Ek, mode, None)
        (got * w.to(dev)).sum().backward()
        grads = [ga.grad.cpu()] + ([gb.grad.cpu()] if gb is not None else []) + [grel.grad.cpu()]
        refs = [a.grad] + ([b.grad] if b is not None else []) + [rel.grad]
        return [float((g_ - r_).double().pow(2).mean().sqrt() / r_.double().pow(2).mean().sqrt()) for g_, r_ in zip(grads, refs)]

    e1 = run("bf16")
    e3 = run("bf16x3")
    assert all(e < 3e-2 for e in e1), e1
    assert all(x3 < x1 for x1, x3 in zip(e1, e3)), (e1, e3)


@pytest.mark.parametrize("shape", [(8200, 512, 512), (1030, 1536, 512), (300, 496, 2048), (257, 40, 64), (4100, 2048, 512)])
@pytest.mark.parametrize("precision", ["bf16x3", "f16x3"])
def test_linear_gemm_kernel_equals_implicit_gemm(shape, precision):
    """`gemm_split_kernel` (the linear layers' own GEMM: two-stage LDS ring, one barrier per K chunk) forms the same
    products in the same order as the 1x1 implicit-GEMM convolution it replaces: bit-identical results, with bias,
    residual and ReLU, ragged M / N tiles and a row-strided input; and fp32-grade against fp64."""
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.priors import _ops
    M, N, K = shape
    dev = _dev()
    g = torch.Generator().manual_seed(M + N)
    xw = torch.randn(M, K + 32, generator=g).to(dev)
    x = xw[:, 16:16 + K]                                   # rows of stride K + 32, 64-byte aligned start
    W = (torch.randn(N, K, generator=g) / K ** 0.5).to(dev)
    b = torch.randn(N, generator=g).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    pw = _ops.pack_linear_weight(W, range_check="now")
    for relu, r in ((False, None), (True, res)):
        got = _ops.linear(x, pw, b, N, relu=relu, residual=r, precision=precision)
        with _hip.knob("ISI_NO_GEMM_KERNEL", 1):
            old = _ops.linear(x, pw, b, N, relu=relu, residual=r, precision=precision)
        # up to 16 rows beyond a multiple of 128 (of at least 1024) take the few-row fp32 kernel instead of a whole extra
        # column of tiles: those rows are only held to the fp64 bound below
        tail = M % 128 if (M % 128 <= 16 and M >= 1024) else 0
        assert torch.equal(got[:M - tail], old[:M - tail])
        ref = x.double() @ W.double().t() + b.double() + (r.double() if r is not None else 0)
        ref = torch.relu(ref) if relu else ref
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        assert err < (2e-5 if precision == "bf16x3" else 2e-6), err
        if precision == "bf16x3" and r is None and N % 32 == 0:
            # the input-gradient operand W^T with its split-bf16 pair copy (one launch) against the transposing copy
            gy = torch.randn(M, N, generator=g).to(dev)
            a = _ops.linear(gy, _ops.pack_linear_weight_t(W), None, K, precision="bf16x3")
            b_ = _ops.linear(gy, _ops.pack_linear_weight(W.t().contiguous()), None, K, precision="bf16x3")
            assert torch.equal(a[:M - tail], b_[:M - tail]) and torch.allclose(a, b_, rtol=0, atol=2e-5 * float(b_.abs().max()))
        if precision == "f16x3":    # inference weights carry their split-f16 pair copy: staged by plain copies, same bits
            pw16 = _ops.pack_linear_weight(W, range_check="now", with_f16=True)
            assert pw16.isi_w16 and torch.equal(_ops.linear(x, pw16, b, N, relu=relu, residual=r, precision=precision), got)


@pytest.mark.parametrize("shape", [(8200, 512, 512), (1030, 1536, 512), (4100, 512, 2048), (300, 128, 128),
                                   # (N/128)(K/128) = 400 tiles in (384, 512): the row-major kernel wants 2 splits where the
                                   # generic count is 1 -- the workspace sizer has to know
                                   (520, 2560, 2560)])
def test_linear_weight_gradient_kernel(shape):
    """`linear_wgrad_kernel` (row-major operands in LDS, transposing fragment reads, two-stage ring) against the
    per-tap kernel it replaces and fp64: dW = dY^T X and db = column sums of dY, ragged last row chunk included."""
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.priors import _train as PT
    M, N, K = shape
    dev = _dev()
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g)
    dy = torch.randn(M, N, generator=g)
    dw_ref, db_ref = dy.double().t() @ x.double(), dy.double().sum(0)
    dw, db = PT.linear_wgrad(x.to(dev), dy.to(dev))
    with _hip.knob("ISI_NO_GEMM_KERNEL", 1):
        dw_old, db_old = PT.linear_wgrad(x.to(dev), dy.to(dev))
    rel = lambda a, b: float((a.double().cpu() - b).abs().max() / b.abs().max())
    assert rel(dw, dw_ref) < 2e-5 and rel(dw_old, dw_ref) < 2e-5, (rel(dw, dw_ref), rel(dw_old, dw_ref))
    assert rel(db, db_ref) < 1e-5 and rel(db_old, db_ref) < 1e-5


def test_feed_forward_fused_dropout():
    """The hidden dropout of a feed-forward block inside linear1's GEMM epilogue (isi_linear_f32: mask = hash of (seed,
    index), nothing stored) and its backward inside linear2's gated input-gradient GEMM: the kept fraction, the scaling of
    the kept units, and every gradient against an fp64 computation with the same mask (read off the output's zeros)."""
    from interactive_spectrogram_inpainting.priors import _ops, _train as PT
    from VQCPCB.transformer.transformer_custom import _LinearParams
    dev = _dev()
    torch.manual_seed(11)
    M, d, ff, p = 1030, 256, 1024, 0.25
    assert _ops.fused_tails_ok(M, ff, d) and _ops.fused_tails_ok(M, ff, d)
    l1, l2 = _LinearParams(d, ff).to(dev), _LinearParams(ff, d).to(dev)
    x = torch.randn(M, d, device=dev, requires_grad=True)
    w = torch.randn(M, d, device=dev)
    h = l1.run(x, relu=True, grad_pre_gated=True, dropout_p=p)
    y = l2.run(h, rectified_input=True, input_keep_scale=1.0 / (1.0 - p))
    (y * w).sum().backward()
    # forward: kept units are relu(x W1^T + b1) / (1 - p), about p of the positive ones are dropped
    pre = torch.relu(x.detach().double() @ l1.weight.detach().double().t() + l1.bias.detach().double())
    kept = (h.detach() != 0)
    pos = pre > 1e-6
    frac = 1.0 - kept[pos].double().mean().item()
    assert abs(frac - p) < 0.01, frac
    assert torch.allclose(h.detach().double()[kept], pre[kept] / (1 - p), rtol=1e-5, atol=1e-6)
    assert not torch.equal(kept[:512], kept[512:1024]), "the mask must not repeat along the rows"
    # backward: fp64 autograd with the same mask
    xr = x.detach().double().requires_grad_(True)
    W1, b1 = l1.weight.detach().double().requires_grad_(True), l1.bias.detach().double().requires_grad_(True)
    W2, b2 = l2.weight.detach().double().requires_grad_(True), l2.bias.detach().double().requires_grad_(True)
    hr = torch.relu(xr @ W1.t() + b1) * kept.double() / (1 - p)
    ((hr @ W2.t() + b2) * w.double()).sum().backward()
    for got, ref, name in ((x.grad, xr.grad, "dx"), (l1.weight.grad, W1.grad, "dW1"), (l1.bias.grad, b1.grad, "db1"),
                           (l2.weight.grad, W2.grad, "dW2"), (l2.bias.grad, b2.grad, "db2")):
        err = float((got.double() - ref).abs().max() / ref.abs().max())
        assert err < 3e-5, (name, err)
    # two calls draw two masks; the same torch seed reproduces a mask
    torch.manual_seed(5)
    h1 = l1.run(x, relu=True, grad_pre_gated=True, dropout_p=p).detach()
    h2 = l1.run(x, relu=True, grad_pre_gated=True, dropout_p=p).detach()
    torch.manual_seed(5)
    h3 = l1.run(x, relu=True, grad_pre_gated=True, dropout_p=p).detach()
    assert not torch.equal(h1 != 0, h2 != 0) and torch.equal(h1, h3)


def test_layernorm_fused_dropout():
    """LayerNorm(dropout(x) + residual) with the dropout inside the kernels (isi_layernorm_dropout_f32 / _bwd_f32): the
    output and every gradient against fp64 autograd with the same mask (read off the x gradient's zeros), the kept
    fraction, mask reproducibility under torch.manual_seed."""
    from interactive_spectrogram_inpainting.priors import _train as PT
    dev = _dev()
    torch.manual_seed(21)
    M, D, p = 777, 512, 0.3
    x = torch.randn(M, D, device=dev, requires_grad=True)
    r = torch.randn(M, D, device=dev, requires_grad=True)
    g = (torch.rand(D, device=dev) + 0.5).requires_grad_(True)
    b = torch.randn(D, device=dev, requires_grad=True)
    w = torch.randn(M, D, device=dev)
    torch.manual_seed(3)
    y = PT.LayerNormFn.apply(x, r, g, b, 1e-5, p)
    (y * w).sum().backward()
    kept = x.grad != 0                                   # (a kept element's gradient is exactly 0 with probability 0)
    frac = 1.0 - kept.double().mean().item()
    assert abs(frac - p) < 0.01, frac
    xr, rr = x.detach().double().requires_grad_(True), r.detach().double().requires_grad_(True)
    gr, br = g.detach().double().requires_grad_(True), b.detach().double().requires_grad_(True)
    yr = torch.nn.functional.layer_norm(xr * kept.double() / (1 - p) + rr, (D,), gr, br, 1e-5)
    (yr * w.double()).sum().backward()
    for got, ref, name in ((y, yr, "y"), (x.grad, xr.grad, "dx"), (r.grad, rr.grad, "dres"), (g.grad, gr.grad, "dgamma"),
                           (b.grad, br.grad, "dbeta")):
        err = float((got.detach().double() - ref.detach()).abs().max() / ref.detach().abs().max())
        assert err < 2e-5, (name, err)
    torch.manual_seed(3)
    y2 = PT.LayerNormFn.apply(x, r, g, b, 1e-5, p)
    y3 = PT.LayerNormFn.apply(x, r, g, b, 1e-5, p)
    assert torch.equal(y2, y) and not torch.equal(y3, y)
    # p = 0 is the plain kernel
    y0 = PT.LayerNormFn.apply(x, r, g, b, 1e-5, 0.0)
    ref0 = torch.nn.functional.layer_norm(x.detach().double() + r.detach().double(), (D,), g.detach().double(), b.detach().double(), 1e-5)
    assert float((y0.double() - ref0).abs().max()) < 1e-4


@pytest.mark.parametrize("Sq,Sk,mode", [(1025, 1025, 1), (1025, 1025, 0), (300, 300, 2), (513, 640, 0)])
@pytest.mark.parametrize("keep_logits", [True, False])
def test_attention_backward_with_poisoned_workspace(Sq, Sk, mode, keep_logits, monkeypatch):
    """G is zeroed only in the margins of its band (kMargin columns either side) and its band is stored whole by the
    kernels: with the workspace filled with NaN before the call, the gradients must be finite and equal to those
    of the full zero-fill (ISI_ATTN_FULL_ZERO=1) -- both with the forward's logits kept (the key-stationary kernel stores
    dS into G, the query-stationary one reads it back) and with everything recomputed."""
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.priors import _ops, _train as PT
    monkeypatch.setattr(_ops, "SAVE_ATTENTION_LOGITS", keep_logits)
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", "bf16x3")
    dev = _dev()
    torch.manual_seed(Sq + mode)
    H, hd, B = 4, 64, 3
    d = H * hd
    q = torch.randn(Sq, B, d, device=dev, requires_grad=True)
    kv = torch.randn(Sk, B, 2 * d, device=dev, requires_grad=True)
    rel = (torch.randn(H, Sq + Sk - 1, hd, device=dev) * 0.3).requires_grad_(True)
    w = torch.randn(Sq, B, d, device=dev)

    def grads(fill, full_zero):
        for t in (q, kv, rel):
            t.grad = None
        monkeypatch.setattr(PT.RelAttentionFn, "workspace_fill", fill)
        with _hip.knob("ISI_ATTN_FULL_ZERO", full_zero):
            out = PT.RelAttentionFn.apply(q, kv, rel, H, 1, 1, Sk, mode, None)
            (out * w).sum().backward()
        return [t.grad.clone() for t in (q, kv, rel)]
    ref = grads(None, 1)
    got = grads(float("nan"), 0)
    for a, b, name in zip(got, ref, ("dq", "dkv", "drel")):
        assert torch.isfinite(a).all(), name
        # the sums over key groups are float atomics: their order, not their terms, differs from run to run
        assert float((a - b).abs().max()) <= 1e-5 * float(b.abs().max()), name


def _toy_training_setup(dropout, seed=11, lr=1e-3, capturable=False):
    from interactive_spectrogram_inpainting.priors.transformer import SelfAttentiveVQTransformer
    from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
    from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
    dev = _dev()
    torch.manual_seed(seed)
    model = SelfAttentiveVQTransformer(shape=[8, 4], condition_shape=[8, 4], self_conditional_model=True,
                                       add_mask_token_to_symbols=True, **COMMON).to(dev).train()
    for m in model.modules():
        if hasattr(m, "dropout") and isinstance(m.dropout, float):
            m.dropout = dropout
    g = torch.Generator().manual_seed(seed + 1)
    B = 4
    cls = {"instrument_family_str": torch.randint(0, 11, (B, 1), generator=g).to(dev),
           "pitch": torch.randint(0, 61, (B, 1), generator=g).to(dev)}
    batches = [(torch.randint(0, 32, (B, 8, 4), generator=g).to(dev), (torch.rand(B, 8, 4, generator=g) < 0.5).to(dev))
               for _ in range(4)]
    opt = make_adam(model.parameters(), lr=lr, **({"capturable": True} if capturable else {}))
    crit = LabelSmoothingLoss(32, 0.1, dim=1)

    def step(code, mask):
        opt.zero_grad(set_to_none=True)
        src, tgt = model.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
        logits, _ = model(tgt, condition=src)
        loss = crit(model.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), code)
        loss.backward()
        opt.step()
        return loss
    return model, step, batches


def test_graphed_training_step_equals_eager():
    """A training step recorded into a HIP graph (utils/training/graphed_step.py) launches the eager step's kernels: after
    one eager warm-up step on batch 0 and replays on batches 1..3 the parameters equal those of four eager steps on the same
    batches (dropout off; float atomics of the unmasked attention backward: their order, not their terms, may differ), the
    losses of the replays equal the eager ones, and a batch with a symbol outside the table raises the reference's
    IndexError after the fact."""
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
    model_e, step_e, batches = _toy_training_setup(0.0)
    losses_e = [float(step_e(*b).detach()) for b in batches]
    model_g, step_g, _ = _toy_training_setup(0.0, capturable=True)
    static = (batches[0][0].clone(), batches[0][1].clone())
    graphed = GraphedTrainingStep(step_g, static, warmup=1, index_limits={0: 32})
    try:
        losses_g = [float(graphed(*b).detach()) for b in batches[1:]]
        for a, b in zip(losses_g, losses_e[1:]):
            assert abs(a - b) <= 2e-5 * abs(b), (losses_g, losses_e)
        for (name, pe), pg in zip(model_e.named_parameters(), model_g.parameters()):
            d = float((pe.detach() - pg.detach()).abs().max())
            assert d <= 2e-4 * max(1e-3, float(pe.abs().max())), (name, d)
        bad = batches[0][0].clone()
        bad[0, 0, 0] = 32
        with pytest.raises(IndexError):      # from the call itself if the verdict has arrived by then, else from finish()
            graphed(bad, batches[0][1])
            graphed.finish()
        graphed.finish()
    finally:
        _ops.set_dropout_seed_base(None)
    # eager code after the replays sees the replayed parameters (version-keyed caches were marked stale)
    model_g.eval()
    with torch.no_grad():
        src, tgt = model_g.to_sequences(batches[1][0], condition=batches[1][0],
                                        class_conditioning={"instrument_family_str": torch.zeros(4, 1, dtype=torch.long, device=_dev()),
                                                            "pitch": torch.zeros(4, 1, dtype=torch.long, device=_dev())},
                                        mask=batches[1][1])
        out_g, _ = model_g(tgt, condition=src)
    assert torch.isfinite(out_g).all()


def test_graphed_training_step_draws_fresh_dropout_masks():
    """The fused dropouts' seeds are launch constants of a recorded step; the device-resident counter the first node of the
    graph advances (isi_set_dropout_seed_base) gives every replay its own masks: with a learning rate of zero two replays
    on one batch give different losses at p = 0.3 and identical ones at p = 0."""
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
    for p, differ in ((0.3, True), (0.0, False)):
        _, step, batches = _toy_training_setup(p, lr=0.0, capturable=True)
        graphed = GraphedTrainingStep(step, (batches[0][0].clone(), batches[0][1].clone()), warmup=1)
        try:
            a = float(graphed(*batches[1]).detach())
            b = float(graphed(*batches[1]).detach())
            graphed.finish()
        finally:
            _ops.set_dropout_seed_base(None)
        assert (a != b) == differ, (p, a, b)


def test_replayed_steps_stay_finite_over_many_recordings():
    """Regression (round 5): ~3 % of the toy model's replayed steps came back with NaN parameter gradients -- reductions that
    read unwritten partial sums -- whenever freed memory held NaNs: a hipMemsetAsync NODE (the attention backward's
    zero-fill of its band buffer) inside the replayed HIP graph was not ordered with the kernels around it (ROCm 7.2, graph
    packet capture; 0 of 300 with DEBUG_CLR_GRAPH_PACKET_CAPTURE=0, and 0 of 300 since the library zero-fills with a kernel).
    25 recordings with NaN-filled freed blocks in front of each: every gradient of every first replay is finite."""
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
    for i in range(25):
        junk = torch.full((1 << (20 + i % 6),), float("nan"), device=_dev())
        del junk
        model, step, batches = _toy_training_setup(0.0, lr=0.0, capturable=True)
        graphed = GraphedTrainingStep(step, (batches[0][0].clone(), batches[0][1].clone()), warmup=1)
        try:
            loss = float(graphed(*batches[1]).detach())
            bad = [n for n, q in model.named_parameters() if q.grad is not None and not torch.isfinite(q.grad).all()]
            graphed.finish()
        finally:
            _ops.set_dropout_seed_base(None)
        assert math.isfinite(loss) and not bad, (i, loss, bad[:4])


def test_weight_leaving_the_f16_range_mid_training_is_loud():
    """a weight under training is compared against the split-f16 operand range only every ~256 versions
    (`_ops.WeightRange`).  A weight that leaves the range INSIDE that window must not give silently wrong products: the
    lagged monitor still says "in range", the GEMM runs split-f16 products, the weight's f16 pieces overflow and the affected
    output column is non-finite -- loud.  An exact re-check (inference weights) sends the same weight to the six-term mode."""
    from interactive_spectrogram_inpainting.priors import _ops
    dev = _dev()
    torch.manual_seed(3)
    w = (torch.randn(512, 512, device=dev) * 0.05).requires_grad_(True)
    x = torch.randn(300, 512, device=dev)
    mon = _ops.WeightRange()
    assert mon.update(w, inference=False)
    with torch.no_grad():
        w[7, 3] = 100.0            # |w| >= 64: beyond the f16 pieces' range (operands are scaled by 2^10)
    assert mon.update(w, inference=False)          # inside the window: not looked at again
    y = _ops.linear(x, _ops.pack_linear_weight(w, range_check=True), None, 512, precision="f16x3")
    assert not torch.isfinite(y[:, 7]).all()
    assert torch.isfinite(y[:, :7]).all() and torch.isfinite(y[:, 8:]).all()
    assert not mon.update(w, inference=True)       # an exact check sees it
    y6 = _ops.linear(x, _ops.pack_linear_weight(w, range_check=False), None, 512, precision="bf16x6")
    ref = x.double() @ w.detach().double().t()
    assert float((y6.double() - ref).abs().max()) <= 1e-4 * float(ref.abs().max())


def test_wt_pack_group_releases_dead_models():
    """the batched W^T re-pack (`_ops._WT_GROUP`) holds its weights weakly -- a trained model that is deleted
    takes its entries (and their 2 N K-float output buffers) with it at the next request or purge; slices of a fused
    in-projection parameter are owned by that parameter."""
    import gc
    from interactive_spectrogram_inpainting.priors import _ops
    grp = _ops._WT_GROUP
    grp.purge()
    before = len(grp.entries)
    model, step, batches = _toy_training_setup(0.0)
    step(*batches[0])
    step(*batches[1])
    torch.cuda.synchronize()
    grown = len(grp.entries)
    assert grown > before, "the training step did not go through the batched W^T pack"
    mem_with = torch.cuda.memory_allocated()
    del model, step, batches
    gc.collect()
    assert grp.purge() == before, (grp.purge(), before)
    torch.cuda.empty_cache()
    assert torch.cuda.memory_allocated() < mem_with
    # a second model after the first one is gone: its table holds only live rows
    model, step, batches = _toy_training_setup(0.0)
    step(*batches[0])
    assert len(grp.entries) == grown
    dev_tables = [t for t in grp.tables.values()]
    assert dev_tables and len(dev_tables[0][0]) == len(grp.entries)
    del model, step, batches
    gc.collect()
    grp.purge()


_PRIOR_FORCED_COLLECTIVES_SCRIPT = r"""
import os, sys, pathlib
root = pathlib.Path(sys.argv[1])
sys.path.insert(0, str(root)); sys.path.insert(0, str(root / "interactive-spectrogram-inpainting_amd")); sys.path.insert(0, str(root / "tests"))
import torch, torch.distributed as dist
dist.init_process_group("nccl", init_method="tcp://127.0.0.1:%s" % sys.argv[2], rank=0, world_size=1)
import test_prior_train_gpu as T
from interactive_spectrogram_inpainting.priors import _ops
from interactive_spectrogram_inpainting.utils.distributed import GradBucketReducer
from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam

def setup(capturable):
    model, _, batches = T._toy_training_setup(0.0, capturable=capturable)
    red = GradBucketReducer(model.parameters(), bucket_mb=0.05, force_collectives=True)     # several buckets, real all-reduces
    opt = make_adam(model.parameters(), lr=1e-3, **({"capturable": True} if capturable else {}))
    crit = LabelSmoothingLoss(32, 0.1, dim=1)
    cls = {"instrument_family_str": torch.zeros(4, 1, dtype=torch.long, device="cuda:0"), "pitch": torch.zeros(4, 1, dtype=torch.long, device="cuda:0")}
    def step(code, mask):
        red.zero()
        src, tgt = model.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
        logits, _ = model(tgt, condition=src)
        loss = crit(model.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), code)
        loss.backward()
        red.finish()
        opt.step()
        return loss
    return model, red, step, batches

me, re_, se, batches = setup(False)
le = [float(se(*b).detach()) for b in batches]
mg, rg, sg, _ = setup(True)
g = GraphedTrainingStep(sg, (batches[0][0].clone(), batches[0][1].clone()), warmup=1, index_limits={0: 32},
                        range_params=[p for p in mg.parameters() if p.dim() == 2], range_check_every=2)
try:
    lg = [float(g(*b).detach()) for b in batches[1:]]
    g.finish()
finally:
    _ops.set_dropout_seed_base(None)
assert len(rg.buckets) >= 3 and g.n_segments == len(rg.buckets) + 2, (len(rg.buckets), g.n_segments)
for a, b in zip(lg, le[1:]):
    assert abs(a - b) <= 2e-5 * abs(b), (lg, le)
for (n, pe), pg in zip(me.named_parameters(), mg.parameters()):
    d = float((pe.detach() - pg.detach()).abs().max())
    assert d <= 2e-4 * max(1e-3, float(pe.abs().max())), (n, d)
print("PRIOR SEGMENTS", g.n_segments, "OK")
dist.destroy_process_group()
"""


def test_graphed_prior_step_with_real_rccl_collectives_between_segments():
    """The prior's data-parallel step under graph replay with RCCL itself: a 1-rank "nccl" group, GradBucketReducer with
    force_collectives -- its bucket all-reduces are launched from autograd's post-accumulate hooks (device thread), each of
    which cuts the recording.  Segments = buckets + 2; the replays reproduce the eager losses and parameters; the
    weight-range re-check (every 2 replays here) runs without a verdict change."""
    import pathlib
    import subprocess
    import sys
    root = pathlib.Path(__file__).resolve().parents[1]
    out = subprocess.run([sys.executable, "-c", _PRIOR_FORCED_COLLECTIVES_SCRIPT, str(root), str(_free_port())], capture_output=True,
                         text=True, timeout=600)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-3000:])
    assert "PRIOR SEGMENTS" in out.stdout and "OK" in out.stdout


def test_aligned_decoder_layer_own_specification():
    """`use_aligned_decoder=True` (priors/transformer.py:388-396; the layer's definition is in the absent package: OUR
    specification, parity unpinned): cross-attention restricted to the source tokens of the target token's own event.  The
    aligned layer equals the plain decoder layer handed the explicit alignment mask, forward and backward; an output row does
    not move when the memory rows of OTHER events change; the wrapper builds and trains with the flag; KV-cached sampling
    refuses it."""
    from VQCPCB.transformer.transformer_custom import TransformerAlignedDecoderLayerCustom, TransformerDecoderLayerCustom
    dev = _dev()
    torch.manual_seed(5)
    kw = dict(d_model=64, nhead=4, num_channels_encoder=1, num_events_encoder=9, num_channels_decoder=4, num_events_decoder=9)
    al = TransformerAlignedDecoderLayerCustom(**kw).to(dev).train()
    pl = TransformerDecoderLayerCustom(**kw).to(dev).train()
    pl.load_state_dict(al.state_dict())
    al.dropout = pl.dropout = 0.0
    St, Ss, B = 36, 9, 3
    tgt = torch.randn(St, B, 64, device=dev, requires_grad=True)
    mem = torch.randn(Ss, B, 64, device=dev, requires_grad=True)
    mask = al.alignment_mask(St, Ss, dev)
    assert mask.shape == (St, Ss) and int((mask == 0).sum()) == St        # one source event per target token
    out_a = al(tgt, mem, "causal")
    out_p = pl(tgt, mem, "causal", mask)
    _close(out_a, out_p, 1e-6, "aligned == plain + alignment mask")
    w = torch.randn_like(out_a)
    ga = torch.autograd.grad((out_a * w).sum(), [tgt, mem] + list(al.parameters()), allow_unused=True)
    gp = torch.autograd.grad((out_p * w).sum(), [tgt, mem] + list(pl.parameters()), allow_unused=True)
    for x, y in zip(ga, gp):
        if x is not None:
            _close(x, y, 1e-5, "aligned backward")
    # locality: target tokens of event 2 (rows 8..11) only see memory row 2
    with torch.no_grad():
        mem2 = mem.detach().clone()
        mem2[[0, 1, 3, 4, 5, 6, 7, 8]] += 1.0
        o1, o2 = al(tgt.detach(), mem.detach(), "causal"), al(tgt.detach(), mem2, "causal")
        # (self-attention is causal over the target only: rows 8..11 depend on memory through their own cross-attention alone)
        assert float((o1[8:12] - o2[8:12]).abs().max()) < 1e-6
        assert float((o1[12:16] - o2[12:16]).abs().max()) > 1e-3
    # a caller's additive memory_mask: combined with the alignment ONCE per (mask tensor, version) -- the same tensor is
    # handed to the attention on every forward, so its classification is cached too --, re-formed when the mask is edited in
    # place; a mask that leaves a target token without any source column is rejected (it would be a NaN softmax)
    extra = torch.zeros(St, Ss, device=dev)
    with torch.no_grad():
        c1 = al._combined(mask, extra)
        assert al._combined(mask, extra) is c1
        o3 = al(tgt.detach(), mem.detach(), "causal", extra)
        assert al._combined(mask, extra) is c1
        _close(o3, o1, 1e-6, "zero memory_mask")
        extra[8:12, 2] = float("-inf")           # removes the only aligned key of event 2 (in place: version bump)
        with pytest.raises(ValueError):
            al(tgt.detach(), mem.detach(), "causal", extra)
    with pytest.raises(ValueError):                # 40 target tokens = 10 events against 9 source events
        al.alignment_mask(40, 9, dev)
    # the wrapper with the flag: builds, one training step runs, sampling refuses
    from interactive_spectrogram_inpainting.priors.transformer import UpsamplingVQTransformer
    from interactive_spectrogram_inpainting.priors._decode import IncrementalDecoder
    m = UpsamplingVQTransformer(shape=[8, 8], condition_shape=[4, 4], use_aligned_decoder=True, **{k: v for k, v in COMMON.items()}).to(dev).train()
    g = torch.Generator().manual_seed(1)
    top = torch.randint(0, 32, (2, 4, 4), generator=g).to(dev)
    bot = torch.randint(0, 32, (2, 8, 8), generator=g).to(dev)
    cls = {"instrument_family_str": torch.zeros(2, 1, dtype=torch.long, device=dev), "pitch": torch.zeros(2, 1, dtype=torch.long, device=dev)}
    src, tgt_ = m.to_sequences(bot, condition=top, class_conditioning=cls)
    logits, memory = m(tgt_, condition=src)
    logits.square().mean().backward()
    assert all(torch.isfinite(p.grad).all() for p in m.parameters() if p.grad is not None)
    with pytest.raises(NotImplementedError):
        IncrementalDecoder(m, memory.detach(), 2)
