"""GPU parity tests of the transformer prior (through the C-ABI).

The wrapper arithmetic (embeddings, positions, start symbols, filtering, masks)
is pinned by fixtures generated from the reference; the layers are checked against
the specification in oracle/prior_oracle.py (parity unpinned: the reference's layer
package is absent).  Tolerance 1e-4 of the tensor max for fp32 activations."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
TOL = 1e-4


def _dev():
    assert torch.cuda.is_available()
    return torch.device("cuda:0")


def _close(a, b, tol=TOL, what=""):
    a = torch.as_tensor(a).detach().float().cpu()
    b = torch.as_tensor(b).detach().float().cpu()
    assert a.shape == b.shape, f"{what}: {tuple(a.shape)} vs {tuple(b.shape)}"
    fin = torch.isfinite(b)
    assert torch.equal(torch.isfinite(a), fin), f"{what}: finite pattern differs"
    err = (a[fin] - b[fin]).abs().max() / b[fin].abs().max().clamp(min=1e-12)
    assert err <= tol, f"{what}: max err / max|ref| = {err:.3e}"


COMMON = dict(n_class=32, channel=8, kernel_size=5, n_block=1, n_res_block=1, res_channel=8,
              d_model=64, embeddings_dim=8, positional_embeddings_dim=8,
              use_relative_transformer=True, predict_frequencies_first=True,
              conditional_model=True, class_conditioning_prepend_to_dummy_input=True,
              class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
              class_conditioning_embedding_dim_per_modality={"instrument_family_str": 16, "pitch": 16},
              conditional_model_nhead=4, conditional_model_num_encoder_layers=2,
              conditional_model_num_decoder_layers=3)


def _models(golden_dir, load=True):
    from interactive_spectrogram_inpainting.priors.transformer import (
        SelfAttentiveVQTransformer, UpsamplingVQTransformer)
    z = np.load(golden_dir / "prior_wrapper.npz")
    torch.manual_seed(5)
    top = SelfAttentiveVQTransformer(shape=[8, 4], condition_shape=[8, 4], self_conditional_model=True,
                                     add_mask_token_to_symbols=True, **COMMON)
    bottom = UpsamplingVQTransformer(shape=[16, 8], condition_shape=[8, 4], **COMMON)
    if load:
        for name, m in (("top", top), ("bottom", bottom)):
            sd = {k[len(name) + 5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "::w::")}
            missing = m.load_state_dict(sd, strict=False)
            assert all(k.startswith("transformer.") for k in missing.missing_keys) and not missing.unexpected_keys
    return z, top.to(_dev()).eval(), bottom.to(_dev()).eval()


@pytest.mark.parametrize("hd,H,Sq,Sk,Cq,Ck,mode", [
    (16, 4, 33, 33, 1, 1, 1), (16, 4, 33, 33, 1, 1, 2), (16, 4, 132, 33, 4, 1, 0),
    (64, 2, 200, 200, 1, 1, 1), (32, 3, 260, 260, 4, 4, 1), (64, 2, 77, 150, 2, 1, 0),
    (16, 2, 1, 1, 1, 1, 0), (64, 1, 1, 97, 1, 1, 0), (32, 2, 64, 64, 1, 1, 2), (64, 2, 129, 129, 1, 1, 1),
    # one or two rows beyond the last full 128-row block: unmasked -> the one-row kernel (every head dim, two tokens per
    # query event, B H = 6 / 10 pairs on the 8 XCD residue classes); causal -> the ragged block is block 0
    (32, 3, 257, 257, 1, 1, 0), (16, 5, 258, 130, 2, 1, 0), (64, 3, 385, 300, 1, 1, 0), (64, 3, 257, 257, 1, 1, 1),
    (32, 2, 300, 300, 1, 1, 1),
])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_rel_attention_against_spec(hd, H, Sq, Sk, Cq, Ck, mode, precision, monkeypatch):
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors import _ops
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", precision)  # both kernels hold the same tolerance
    torch.manual_seed(hd + Sq)
    d, B = hd * H, 2
    Eq, Ek = -(-Sq // Cq), -(-Sk // Ck)
    q, k, v = torch.randn(Sq, B, d), torch.randn(Sk, B, d), torch.randn(Sk, B, d)
    rel = torch.randn(H, Eq + Ek - 1, hd) * 0.5
    # reference through the oracle's attention with identity projections
    eye = torch.eye(d)
    sd = {"in_proj_weight": torch.cat([eye, eye, eye]), "in_proj_bias": torch.zeros(3 * d),
          "out_proj.weight": eye, "out_proj.bias": torch.zeros(d), "rel_embeddings": rel}
    mask = None
    if mode == 1:
        mask = P.causal_mask(Sq)
    elif mode == 2:
        mask = P.causal_mask(Sq).t()
    # q and k/v come from different tensors: emulate with a block-diagonal trick
    ref_logits_in = P.attention  # noqa
    hq = q.reshape(Sq, B, H, hd).permute(1, 2, 0, 3)
    hk = k.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    hv = v.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    logits = hq @ hk.transpose(-1, -2)
    qe = torch.einsum("bhid,hrd->bhir", hq, rel)
    idx = P.rel_index(Sq, Sk, Cq, Ck, Ek)
    logits = (logits + qe.gather(3, idx.expand(B, H, Sq, Sk))) / math.sqrt(hd)
    if mask is not None:
        logits = logits + mask
    ref = (torch.softmax(logits, -1) @ hv).permute(2, 0, 1, 3).reshape(Sq, B, d)
    dev = _dev()
    got = _ops.rel_attention(q.to(dev), k.to(dev), v.to(dev), rel.to(dev), H, Cq, Ck, Ek, mask_mode=mode)
    _close(got, ref, TOL, "rel_attention")
    # dense additive mask path == predicate path
    if mask is not None:
        got2 = _ops.rel_attention(q.to(dev), k.to(dev), v.to(dev), rel.to(dev), H, Cq, Ck, Ek, mask_mode=0,
                                  dense_mask=mask.to(dev).contiguous())
        _close(got2, ref, TOL, "rel_attention dense mask")
    # no-bias variant
    ref_nb = (torch.softmax((hq @ hk.transpose(-1, -2)) / math.sqrt(hd) + (mask if mask is not None else 0), -1)
              @ hv).permute(2, 0, 1, 3).reshape(Sq, B, d)
    _close(_ops.rel_attention(q.to(dev), k.to(dev), v.to(dev), None, H, Cq, Ck, Ek, mask_mode=mode), ref_nb,
           TOL, "no_bias")
    # decode kernel: last causal row / arbitrary row
    pos = Sq - 1
    nk = pos + 1 if mode == 1 else Sk
    if mode != 2 and nk <= Sk:
        row = _ops.rel_attention_decode(q[pos].to(dev), k.to(dev), v.to(dev), rel.to(dev), H, nk, pos, Cq, Ck, Ek)
        _close(row, ref[pos], TOL, "decode row")


@pytest.mark.parametrize("hd,H,B,S,mode", [(64, 2, 2, 200, 1), (64, 8, 8, 1025, 1), (64, 4, 3, 385, 0), (32, 4, 2, 300, 2),
                                           (32, 16, 4, 1025, 1), (64, 1, 1, 1, 0), (64, 2, 2, 130, 1)])
def test_rel_attention_plane_staged_kernel(hd, H, B, S, mode):
    """rel_attention_fwd3.hip (K / V / e as 16-bit planes in isi_attn_args.workspace, LDS-DMA tiles, two wave groups half a
    step apart) against the register-staged kernel it replaces (ISI_ATTN_NO_FWD3=1): the same products in the same
    precision -- outputs and log-sum-exps agree to accumulation order, the kept logits of the three-term mode to 2e-4
    (base-2 units) -- in all three 16-bit modes, REPEATED: the kernel's first versions failed one run in four at this size
    (profiles/r06_attention_skew_race.txt)."""
    import ctypes as C
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.priors import _ops
    dev = _dev()
    torch.manual_seed(S + hd)
    d = hd * H
    q, k, v = (torch.randn(S, B, d, device=dev) for _ in range(3))
    rel = torch.randn(H, 2 * S - 1, hd, device=dev) * 0.5
    ld = (S + 31) // 32 * 32
    saved = _ops.ATTENTION_PRECISION
    try:
        with _hip.knob("ISI_ATTN_FWD3_ALL", 1):
            for prec, tol in (("bf16x3", 2e-5), ("bf16", 2e-3), ("f16", 3e-4)):
                _ops.ATTENTION_PRECISION = prec
                lse0, lg0 = torch.empty(B, H, S, device=dev), torch.zeros(B, H, S, ld, device=dev)
                with _hip.knob("ISI_ATTN_NO_FWD3", 1):
                    ref = _ops.rel_attention(q, k, v, rel, H, 1, 1, S, mask_mode=mode, lse=lse0, logits=lg0)
                for rep in range(6 if S > 1000 else 2):
                    lse, lg = torch.empty(B, H, S, device=dev), torch.zeros(B, H, S, ld, device=dev)
                    got = _ops.rel_attention(q, k, v, rel, H, 1, 1, S, mask_mode=mode, lse=lse, logits=lg)
                    err = float((got - ref).abs().max() / ref.abs().max())
                    assert err < tol, f"{prec} run {rep}: output differs from the register-staged kernel's by {err:.2e}"
                    assert float((lse - lse0).abs().max()) < 50 * tol
                    if prec == "bf16x3":
                        i, j = torch.arange(S, device=dev)[:, None], torch.arange(S, device=dev)[None, :]
                        allowed = (j <= i) if mode == 1 else (j >= i) if mode == 2 else torch.ones(S, S, dtype=torch.bool, device=dev)
                        if mode == 0 and S % 128 in (1, 2) and S >= 256:
                            allowed[S - S % 128:] = False      # (the one-row kernel of the last rows keeps no logits)
                        dl = float((lg[..., :S] - lg0[..., :S])[:, :, allowed].abs().max())
                        assert dl < 2e-4, f"kept logits differ by {dl:.2e}"
        # the workspace contract: none for several channels per event / the exact-fp32 mode / (by default) single-term modes
        a = _ops._attn_args(q, k, v, rel, torch.empty(S, B, d, device=dev), S, S, B, H, hd, 1, 1, S, mode, None)
        L = _hip.lib()
        for prec, want in ((1, True), (0, False), (2, False)):
            a.precision = prec
            assert (L.isi_rel_attention_workspace_bytes(C.byref(a)) > 0) == want
        a.precision, a.Cq = 1, 2
        assert L.isi_rel_attention_workspace_bytes(C.byref(a)) == 0
        a.Cq = 1
        need = L.isi_rel_attention_workspace_bytes(C.byref(a))
        ws = torch.empty(need, dtype=torch.uint8, device=dev)
        a.q_ss, a.q_sb, a.q_sh = q.stride(0), q.stride(1), hd
        a.k_ss, a.k_sb, a.k_sh = k.stride(0), k.stride(1), hd
        a.v_ss, a.v_sb, a.v_sh = v.stride(0), v.stride(1), hd
        a.o_ss, a.o_sb, a.o_sh = B * d, d, hd
        a.workspace, a.workspace_bytes = ws.data_ptr(), need - 256
        assert L.isi_rel_attention_f32(C.byref(a), _hip.stream_ptr(dev)) == -1      # ISI_E_INVALID: workspace too small
    finally:
        _ops.ATTENTION_PRECISION = saved


def test_small_ops_against_torch():
    from interactive_spectrogram_inpainting.priors import _ops
    dev = _dev()
    torch.manual_seed(2)
    x = torch.randn(37, 3, 96)
    W, b = torch.randn(50, 96) * 0.1, torch.randn(50)
    r = torch.randn(37, 3, 50)
    ref = torch.relu(torch.nn.functional.linear(x, W, b) + r)
    got = _ops.linear(x.to(dev), _ops.pack_linear_weight(W.to(dev)), b.to(dev), 50, relu=True, residual=r.to(dev))
    _close(got, ref, 1e-5, "linear")
    x2 = torch.randn(5, 96)
    got = _ops.linear_rows(x2.to(dev), W.to(dev), b.to(dev), relu=False, residual=r[0, :, :][:1].expand(5, 50).contiguous().to(dev))
    _close(got, torch.nn.functional.linear(x2, W, b) + r[0, :1], 1e-5, "linear_rows")
    g, be = torch.randn(96), torch.randn(96)
    _close(_ops.layernorm(x.to(dev), g.to(dev), be.to(dev)), torch.nn.functional.layer_norm(x, (96,), g, be), 1e-5, "layernorm")


def test_filtering_and_sampling_against_reference(golden_dir):
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors import _ops
    import sample as S
    z = np.load(golden_dir / "filtering.npz")
    logits = torch.from_numpy(z["logits"])
    dev = _dev()
    for key in z.files:
        if key == "logits":
            continue
        k, p = key[1:].split("_p")
        k, p = int(k), float(p)
        got = S.top_k_top_p_filtering(logits.clone().to(dev), top_k=k, top_p=p)
        _close(got, z[key], 1e-6, f"filtering {key}")
        assert torch.equal(P.top_k_top_p_filtering(logits, k, p), torch.from_numpy(z[key]))
    torch.manual_seed(3)
    rows = torch.randn(6, 32) * 2
    u = torch.rand(6)
    for (T, k, p) in [(1.0, 0, 0.0), (0.7, 5, 0.0), (1.3, 0, 0.8)]:
        filt = P.top_k_top_p_filtering(rows / T, k, p)
        ref = P.sample_from_uniform(torch.softmax(filt, -1), u)
        got = _ops.sample_rows(rows.to(dev), T, k, p, u)
        assert torch.equal(got.cpu(), ref), (T, k, p)


def test_wrapper_sequences_against_reference(golden_dir):
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    cls = {k[5:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("cls::")}
    code = torch.from_numpy(z["top::code"]).to(dev)
    mask = torch.from_numpy(z["top::mask"]).to(dev)
    src, tgt = top.to_sequences(code, code, class_conditioning=cls, mask=mask)
    _close(src, z["top::src"], 1e-6, "top src"); _close(tgt, z["top::tgt"], 1e-6, "top tgt")
    tidx = z["top::tidx"].tolist()
    src, tgt = top.to_sequences(code, code, class_conditioning=cls, time_indexes_source=tidx, time_indexes_target=tidx)
    _close(src, z["top::src_tidx"], 1e-6, "top src tidx"); _close(tgt, z["top::tgt_tidx"], 1e-6, "top tgt tidx")
    bcode = torch.from_numpy(z["bottom::code"]).to(dev)
    src, tgt = bottom.to_sequences(bcode, code, class_conditioning=cls)
    _close(src, z["bottom::src"], 1e-6, "bottom src"); _close(tgt, z["bottom::tgt"], 1e-6, "bottom tgt")
    # the layer constructor arguments the reference hands to the (absent) layer package
    enc, dec = z["top::enc_layer_args"], z["bottom::dec_layer_args"]
    l0 = top.transformer.encoder.layers[0].self_attn
    assert [l0.d_model, l0.nhead, l0.Cq, l0.Eq] == enc.tolist()
    d0 = bottom.transformer.decoder.layers[0]
    assert [d0.self_attn.d_model, d0.self_attn.nhead, d0.multihead_attn.Ck, d0.multihead_attn.Ek,
            d0.self_attn.Cq, d0.self_attn.Eq] == dec.tolist()


def test_positional_class_conditioning_against_reference(golden_dir):
    """positional_class_conditioning=True (reference priors/transformer.py:270-271,331,345,555-560,660-663): class
    embeddings appended to every row and to the start symbols -- sequences pinned to the reference's, parameter shapes
    equal to its state dict; the model then runs (forward, KV-cached sampling == full-pass sampling)."""
    import sample as S
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors.transformer import (
        SelfAttentiveVQTransformer, UpsamplingVQTransformer, Seq2SeqInputKind)
    z = np.load(golden_dir / "prior_wrapper_positional.npz")
    dev = _dev()
    common = {k: v for k, v in COMMON.items() if k != "class_conditioning_prepend_to_dummy_input"}
    common["positional_class_conditioning"] = True
    torch.manual_seed(6)
    top = SelfAttentiveVQTransformer(shape=[8, 4], condition_shape=[8, 4], self_conditional_model=True,
                                     add_mask_token_to_symbols=True, **common)
    bottom = UpsamplingVQTransformer(shape=[16, 8], condition_shape=[8, 4], **common)
    for name, m in (("top", top), ("bottom", bottom)):
        sd = {k[len(name) + 5:]: torch.from_numpy(z[k]) for k in z.files if k.startswith(name + "::w::")}
        missing = m.load_state_dict(sd, strict=False)      # strict shapes: a width mismatch raises here
        assert all(k.startswith("transformer.") for k in missing.missing_keys) and not missing.unexpected_keys
        assert m.embeddings_effective_dim == int(z[name + "::effective_dim"]) == 64 - 8 - 32
    top, bottom = top.to(dev).eval(), bottom.to(dev).eval()
    cls = {k[5:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("cls::")}
    code = torch.from_numpy(z["top::code"]).to(dev)
    mask = torch.from_numpy(z["top::mask"]).to(dev)
    src, tgt = top.to_sequences(code, code, class_conditioning=cls, mask=mask)
    _close(src, z["top::src"], 1e-6, "top src"); _close(tgt, z["top::tgt"], 1e-6, "top tgt")
    bcode = torch.from_numpy(z["bottom::code"]).to(dev)
    srcb, tgtb = bottom.to_sequences(bcode, code, class_conditioning=cls)
    _close(srcb, z["bottom::src"], 1e-6, "bottom src"); _close(tgtb, z["bottom::tgt"], 1e-6, "bottom tgt")
    logits, _ = bottom(tgtb, srcb)
    assert logits.shape == (2, 128, 32) and torch.isfinite(logits).all()
    # sampling: KV-cached loop == full pass per token (same uniforms)
    B = 2
    g = torch.Generator().manual_seed(3)
    uni = torch.rand(top.target_transformer_sequence_length, B, generator=g)
    cond = {"pitch": torch.tensor([20]), "instrument_family_str": torch.tensor([3])}
    got = S.sample_model(top, dev, B, [8, 4], temperature=1.0, class_conditioning=cond, top_p_sampling_p=0.9, uniforms=uni)
    clsd = {k: v.long().expand(B).reshape(B, 1).to(dev) for k, v in cond.items()}
    codemap = torch.full((B, 8, 4), top.mask_token_index, dtype=torch.int64, device=dev)
    srcs, tgts = top.to_sequences(codemap.clamp(max=top.n_class_target - 1), codemap, class_conditioning=clsd)
    seq = top.target_codemaps_helper.to_sequence(codemap).clone()
    memory = None
    for i in range(seq.shape[1]):
        lg, memory = top(tgts, srcs, memory=memory)
        s_i = _ops.sample_rows(lg[:, i].contiguous(), 1.0, 0, 0.9, uni[i])
        seq[:, i] = s_i
        emb = top.embed_data(s_i, Seq2SeqInputKind.Target)
        tgts[:, i + 1, :top.embeddings_effective_dim] = emb
    assert torch.equal(got, top.target_codemaps_helper.to_time_frequency_map(seq))


def _oracle_logits(model, src, tgt):
    from oracle import prior_oracle as P
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    H = model.conditional_model_nhead
    Ce, Ee = model.source_num_channels, model.source_num_events_with_start_symbol
    Cd, Ed = model.target_num_channels, model.target_num_events_with_start_symbol
    s, t = src.cpu().transpose(0, 1), tgt.cpu().transpose(0, 1)
    St = t.shape[0]
    enc_mask = P.causal_mask(s.shape[0]).t() if model.self_conditional_model else None
    memory = P.encoder(s, sd, "transformer.encoder.", model.conditional_model_num_encoder_layers, H, Ce, Ee, enc_mask)
    out = P.decoder(t, memory, sd, "transformer.decoder.", model.conditional_model_num_decoder_layers, H,
                    Cd, Ed, Ce, Ee, P.causal_mask(St), None)
    start = model.target_start_symbol.shape[1]
    out = out[start - 1:-1].transpose(0, 1)
    return torch.nn.functional.linear(out, sd["project_transformer_outputs_to_logits.weight"],
                                      sd["project_transformer_outputs_to_logits.bias"]), memory, out


def test_prior_forward_and_incremental_decoding(golden_dir):
    from interactive_spectrogram_inpainting.priors._decode import IncrementalDecoder
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    cls = {k[5:]: torch.from_numpy(z[k]).to(dev) for k in z.files if k.startswith("cls::")}
    code = torch.from_numpy(z["top::code"]).to(dev)
    bcode = torch.from_numpy(z["bottom::code"]).to(dev)
    mask = torch.from_numpy(z["top::mask"]).to(dev)
    for name, m, (src, tgt) in (("top", top, top.to_sequences(code, code, class_conditioning=cls, mask=mask)),
                                ("bottom", bottom, bottom.to_sequences(bcode, code, class_conditioning=cls))):
        ref_logits, ref_mem, _ = _oracle_logits(m, src, tgt)
        logits, memory = m(tgt, src)
        assert logits.shape == ref_logits.shape
        _close(memory, ref_mem, TOL, name + " memory")
        _close(logits, ref_logits, TOL, name + " logits")
        # the reference wrapper's calling convention with explicit float masks also works
        out2, *_ = m.transformer.decoder(tgt.transpose(0, 1), memory, tgt_mask=m.causal_mask.to(dev), memory_mask=None)
        start = m.target_start_symbol.shape[1]
        _close(m.project_transformer_outputs_to_logits.run(out2[start - 1:-1].transpose(0, 1).contiguous()),
               ref_logits, TOL, name + " logits (dense causal mask tensor)")
        # incremental rows == full pass rows
        dec = IncrementalDecoder(m, memory, tgt.shape[0])
        x = tgt.transpose(0, 1).contiguous()
        for p in range(x.shape[0] - 1):
            row = dec.step(p, x[p])
            i = p - (start - 1)
            if i >= 0 and i % 7 == 0:
                _close(dec.logits(row), ref_logits[:, i], TOL, f"{name} incremental logits @ {i}")


def test_sample_model_matches_full_pass_sampling(golden_dir):
    """KV-cached sampling == the reference's loop (full decoder pass per token,
    sample.py:268-305) when both draw from the same uniforms."""
    import sample as S
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors.transformer import Seq2SeqInputKind
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    B = 2
    g = torch.Generator().manual_seed(9)
    init = torch.randint(0, 32, (B, 8, 4), generator=g)
    mask = torch.zeros(1, 8, 4, dtype=torch.bool)
    mask[:, :, 1:3] = True                                       # regenerate the two middle columns
    cls = {"pitch": torch.tensor([20]), "instrument_family_str": torch.tensor([3])}
    S_len = top.target_transformer_sequence_length
    uni = torch.rand(S_len, B, generator=g)
    got = S.sample_model(top, dev, B, [8, 4], temperature=0.9, class_conditioning=cls, initial_code=init.clone(),
                         mask=mask, top_p_sampling_p=0.8, uniforms=uni)
    assert got.shape == (B, 8, 4) and got.dtype == torch.int64
    keep = ~mask.expand(B, -1, -1)
    assert torch.equal(got.cpu()[keep], init[keep]), "unmasked positions must keep initial_code"
    # reference-loop semantics with the full forward
    clsd = {k: v.long().expand(B).reshape(B, 1).to(dev) for k, v in cls.items()}
    codemap = init.clone().to(dev)
    src, tgt = top.to_sequences(codemap, codemap, class_conditioning=clsd, mask=mask.to(dev))
    seq = top.target_codemaps_helper.to_sequence(codemap).clone()
    mseq = top.target_codemaps_helper.to_sequence(mask.to(dev))[0].cpu().numpy()
    memory = None
    for i, is_masked in enumerate(mseq):
        if not is_masked:
            continue
        logits, memory = top(tgt, src, memory=memory)
        s = _ops.sample_rows(logits[:, i].contiguous(), 0.9, 0, 0.8, uni[i])
        seq[:, i] = s
        tgt[:, i + 1, :top.embeddings_effective_dim] = top.embed_data(s, Seq2SeqInputKind.Target)
    ref = top.target_codemaps_helper.to_time_frequency_map(seq)
    assert torch.equal(got, ref)
    # the opt-in execution modes of the decoding loop (hipGraph replay of the per-position launch
    # sequence) draws the same codes
    from interactive_spectrogram_inpainting import _hip
    with _hip.knob("ISI_PRIOR_GRAPH", 1):
        alt = S.sample_model(top, dev, B, [8, 4], temperature=0.9, class_conditioning=cls,
                             initial_code=init.clone(), mask=mask, top_p_sampling_p=0.8, uniforms=uni)
    assert torch.equal(alt, ref), "ISI_PRIOR_GRAPH"
    # round 5: W consecutive positions per graph (windows of sampled / kept positions, mixed windows as direct launches)
    for w_ in (3, 4, 8):
        with _hip.knob("ISI_PRIOR_GRAPH", w_):
            alt = S.sample_model(top, dev, B, [8, 4], temperature=0.9, class_conditioning=cls,
                                 initial_code=init.clone(), mask=mask, top_p_sampling_p=0.8, uniforms=uni)
        assert torch.equal(alt, ref), f"ISI_PRIOR_GRAPH={w_}"
    # bottom prior conditioned on the sampled top map: unmasked call runs and stays in range (equality with the
    # full-pass loop: test_bottom_prior_sample_model_matches_full_pass_sampling)
    out_b = S.sample_model(bottom, dev, B, [16, 8], temperature=1.0, condition=got, class_conditioning=cls,
                           generator=torch.Generator().manual_seed(1))
    assert out_b.shape == (B, 16, 8) and int(out_b.min()) >= 0 and int(out_b.max()) < 32


def _full_pass_sampling(model, codemap, condition, clsd, mask, uniforms, temperature, top_k, top_p):
    """The reference's loop (sample.py:268-336) on the build's full forward: one complete decoder pass per masked
    token in the target helper's order, the token drawn from row i of the logits with the given uniform, its embedding
    written into the next decoder input row (priors/transformer.py:848-872).  Returns the [B, F, T] map."""
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors.transformer import Seq2SeqInputKind
    src, tgt = model.to_sequences(codemap, condition, class_conditioning=clsd, mask=mask)
    seq = model.target_codemaps_helper.to_sequence(codemap).clone()
    S_len = model.target_transformer_sequence_length
    mseq = model.target_codemaps_helper.to_sequence(mask).reshape(-1, S_len)[0].cpu().numpy()
    start_len = model.target_start_symbol.shape[1]
    memory = None
    for i, is_masked in enumerate(mseq):
        if not is_masked:
            continue
        logits, memory = model(tgt, src, memory=memory)
        s = _ops.sample_rows(logits[:, i].contiguous(), temperature, top_k, top_p, uniforms[i])
        seq[:, i] = s
        if i + start_len < tgt.shape[1]:
            tgt[:, i + start_len, :model.embeddings_effective_dim] = model.embed_data(s, Seq2SeqInputKind.Target)
    return model.target_codemaps_helper.to_time_frequency_map(seq), int(mseq.sum())


def test_bottom_prior_sample_model_matches_full_pass_sampling(golden_dir):
    """VERDICT r02 missing 1: `sample_model` on the UpsamplingVQTransformer (zig-zag target order, 4 target tokens per
    source event, cached cross-attention over the top map) == the reference's loop with one full pass per token and the
    same uniforms, at [16, 8] with batch 2; unmasked codes are kept (reference sample.py:268-336,
    priors/codemaps_helpers.py:108-243)."""
    import sample as S
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    B = 2
    g = torch.Generator().manual_seed(31)
    cond = torch.randint(0, 32, (B, 8, 4), generator=g)
    init = torch.randint(0, 32, (B, 16, 8), generator=g)
    cls = {"pitch": torch.tensor([20]), "instrument_family_str": torch.tensor([3])}
    clsd = {k: v.long().expand(B).reshape(B, 1).to(dev) for k, v in cls.items()}
    S_len = bottom.target_transformer_sequence_length
    assert S_len == 128 and bottom.target_start_symbol.shape[1] == 4
    for mask in (torch.ones(1, 16, 8, dtype=torch.bool),                      # everything
                 _window_mask(16, 8, slice(3, 13), slice(2, 6))):            # a window: prefix prefill + kept tail
        uni = torch.rand(S_len, B, generator=g)
        for top_p, top_k, temp in ((0.8, 0, 0.9), (0.0, 0, 1.0), (0.0, 5, 1.1)):
            got = S.sample_model(bottom, dev, B, [16, 8], temperature=temp, condition=cond, class_conditioning=cls,
                                 initial_code=init.clone(), mask=mask, top_p_sampling_p=top_p,
                                 top_k_sampling_k=top_k, uniforms=uni)
            keep = ~mask.expand(B, -1, -1)
            assert torch.equal(got.cpu()[keep], init[keep]), "unmasked positions must keep initial_code"
            ref, n_masked = _full_pass_sampling(bottom, init.clone().to(dev), cond.to(dev), clsd, mask.to(dev), uni,
                                                temp, top_k, top_p)
            assert n_masked == int(mask.sum())
            assert torch.equal(got, ref), f"{(got != ref).sum().item()} of {n_masked * B} sampled codes differ"


def _window_mask(F, T, fs, ts):
    m = torch.zeros(1, F, T, dtype=torch.bool)
    m[:, fs, ts] = True
    return m


def test_bottom_prior_sample_model_at_baseline_size_matches_full_pass_sampling():
    """BASELINE config 5, bottom half: KV-cached sampling on the [64,64] bottom map (4096 tokens + 4 start rows,
    d_model 512, 6 + 8 layers; conditioned on a [32,32] top map) with a 64-token mask == the reference's loop (one
    full 4100-row decoder pass per masked token) drawing from the same uniforms.  Covers what the [16,8] case cannot:
    the long key splits of the cached self-attention (two-pass form beyond 144 keys per split), the batched prefill of
    a 2000-row prefix and the 1025-row cached cross-attention."""
    import sample as S
    bottom = _full_bottom()
    dev = _dev()
    B = 1
    g = torch.Generator().manual_seed(23)
    cond = torch.randint(0, 512, (B, 32, 32), generator=g)
    init = torch.randint(0, 512, (B, 64, 64), generator=g)
    mask = _window_mask(64, 64, slice(10, 42), slice(34, 36))               # 32 frequencies x 2 frames = 64 tokens
    cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
    clsd = {k: v.long().expand(B).reshape(B, 1).to(dev) for k, v in cls.items()}
    uni = torch.rand(bottom.target_transformer_sequence_length, B, generator=g)
    got = S.sample_model(bottom, dev, B, [64, 64], temperature=1.0, condition=cond, class_conditioning=cls,
                         initial_code=init.clone(), mask=mask, top_p_sampling_p=0.8, uniforms=uni)
    keep = ~mask.expand(B, -1, -1)
    assert torch.equal(got.cpu()[keep], init[keep]), "unmasked positions must keep initial_code"
    ref, n_masked = _full_pass_sampling(bottom, init.clone().to(dev), cond.to(dev), clsd, mask.to(dev), uni, 1.0, 0, 0.8)
    assert n_masked == 64
    # a draw sits on a CDF step: rounding differences between the cached row and the full pass may move a code only
    # if the uniform falls within float rounding of a step -- identical in practice
    assert (got != ref).sum().item() <= 1, f"{(got != ref).sum().item()} of 64 sampled codes differ"


def test_predictive_sampling_equals_sequential_gumbel_max(golden_dir):
    """use_predictive_sampling (reference sample.py:251-261,268-342): forecasts + skipped steps give exactly the map of
    token-by-token Gumbel-max sampling (argmax(log softmax(filtered logits_i) + g_i), one full pass per token) with the
    same noise; unmasked positions keep their codes."""
    import sample as S
    from interactive_spectrogram_inpainting.priors.transformer import Seq2SeqInputKind
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    for B, masked_cols in ((1, slice(1, 3)), (2, slice(0, 4))):
        g = torch.Generator().manual_seed(21 + B)
        init = torch.randint(0, 32, (B, 8, 4), generator=g)
        mask = torch.zeros(1, 8, 4, dtype=torch.bool)
        mask[:, :, masked_cols] = True
        cls = {"pitch": torch.tensor([20]), "instrument_family_str": torch.tensor([3])}
        S_len = top.target_transformer_sequence_length
        u = torch.rand(B, S_len, top.n_class_target, generator=g).clamp_(1e-9, 1 - 1e-9)
        gumbel = -torch.log(-torch.log(u))
        got = S.sample_model(top, dev, B, [8, 4], temperature=0.9, class_conditioning=cls, initial_code=init.clone(),
                             mask=mask, top_p_sampling_p=0.8, use_predictive_sampling=True, gumbel_noise=gumbel)
        assert got.shape == (B, 8, 4) and got.dtype == torch.int64
        keep = ~mask.expand(B, -1, -1)
        assert torch.equal(got.cpu()[keep], init[keep])
        assert 0.0 <= top.predictive_sampling_correct_ratio <= 1.0
        # sequential reference: one full pass per masked token, argmax with that token's noise
        clsd = {k: v.long().expand(B).reshape(B, 1).to(dev) for k, v in cls.items()}
        codemap = init.clone().to(dev)
        src, tgt = top.to_sequences(codemap, codemap, class_conditioning=clsd, mask=mask.to(dev))
        seq = top.target_codemaps_helper.to_sequence(codemap).clone()
        mseq = top.target_codemaps_helper.to_sequence(mask.to(dev))[0].cpu().numpy()
        memory = None
        start = top.target_start_symbol.shape[1]
        for i, is_masked in enumerate(mseq):
            if not is_masked:
                continue
            logits, memory = top(tgt, src, memory=memory)
            li = S.top_k_top_p_filtering(logits[:, i] / 0.9, top_k=0, top_p=0.8)
            s_i = torch.argmax(torch.log(torch.softmax(li, -1)) + gumbel[:, i].to(dev), -1)
            seq[:, i] = s_i
            emb = top.embed_data(s_i, Seq2SeqInputKind.Target)
            tgt[:, i + start, :top.embeddings_effective_dim] = emb
            if top.self_conditional_model:
                src[:, i + top.source_start_symbol.shape[1], :top.embeddings_effective_dim] = emb
        ref = top.target_codemaps_helper.to_time_frequency_map(seq)
        assert torch.equal(got, ref), B


def test_sample_model_edge_masks(golden_dir):
    """Empty mask (nothing to resample), a single masked token, the last token only, batch 1 vs the same row of
    a batch: the KV-cached loop keeps every unmasked code and its draws do not depend on the batch composition."""
    import sample as S
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    g = torch.Generator().manual_seed(11)
    init = torch.randint(0, 32, (3, 8, 4), generator=g)
    cls = {"pitch": torch.tensor([20]), "instrument_family_str": torch.tensor([3])}
    S_len = top.target_transformer_sequence_length
    uni = torch.rand(S_len, 3, generator=g)
    none = torch.zeros(1, 8, 4, dtype=torch.bool)
    out = S.sample_model(top, dev, 3, [8, 4], temperature=1.0, class_conditioning=cls, initial_code=init.clone(),
                         mask=none, uniforms=uni)
    assert torch.equal(out.cpu(), init)
    for pos in ((0, 0), (3, 2), (7, 3)):
        m = none.clone()
        m[0, pos[0], pos[1]] = True
        out = S.sample_model(top, dev, 3, [8, 4], temperature=1.0, class_conditioning=cls, initial_code=init.clone(),
                             mask=m, uniforms=uni)
        keep = ~m.expand(3, -1, -1)
        assert torch.equal(out.cpu()[keep], init[keep]) and 0 <= int(out.min()) and int(out.max()) < 32
        # row 1 alone, with its own column of uniforms, reproduces row 1 of the batch
        solo = S.sample_model(top, dev, 1, [8, 4], temperature=1.0, class_conditioning=cls,
                              initial_code=init[1:2].clone(), mask=m, uniforms=uni[:, 1:2].contiguous())
        assert torch.equal(solo[0], out[1])


def test_cached_attention_two_splits_in_one_workgroup():
    """Round 6: batches of 17 .. 63 sequences split the cached attention's keys in two; both halves now run in ONE workgroup
    that also merges them (no partials in memory, no combine launch).  Bit for bit the two-launch form
    (ISI_DECODE_ATTN_SEPARATE_SPLITS=1) at every key count -- one key, a half that stays empty, the split boundary, the full
    cache -- with and without the relative table, and the torch formula."""
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting import _hip
    from oracle import prior_oracle as P
    dev = _dev()
    H, hd, S = 8, 64, 1024
    d = H * hd
    g = torch.Generator().manual_seed(31)
    for B in (32, 48):
        k = torch.randn(S, B, d, generator=g).to(dev)
        v = torch.randn(S, B, d, generator=g).to(dev)
        rel = (torch.randn(H, 2 * S - 1, hd, generator=g) * 0.2).to(dev)
        k[600:] = float("nan")                     # rows beyond the keys in use may hold anything
        v[600:] = float("nan")
        for nk in (1, 100, 144, 145, 300, 599, 600):
            q = torch.randn(B, d, generator=g).to(dev)
            for r in (rel, None):
                one = _ops.rel_attention_decode(q, k, v, r, H, nk, nk - 1, 1, 1, S)
                with _hip.knob("ISI_DECODE_ATTN_SEPARATE_SPLITS", 1):
                    two = _ops.rel_attention_decode(q, k, v, r, H, nk, nk - 1, 1, 1, S)
                assert torch.isfinite(one).all() and torch.equal(one, two), (B, nk, r is None)
            # the formula (relative rows: key j of query position p reads table row p - j + Ek - 1)
            hq = q.view(B, H, hd).cpu().double()
            hk = k[:nk].view(nk, B, H, hd).permute(1, 2, 0, 3).cpu().double()
            hv = v[:nk].view(nk, B, H, hd).permute(1, 2, 0, 3).cpu().double()
            idx = (nk - 1) - torch.arange(nk) + S - 1
            logit = (torch.einsum("bhd,bhjd->bhj", hq, hk) + torch.einsum("bhd,hjd->bhj", hq, rel.cpu().double()[:, idx])) / math.sqrt(hd)
            ref = torch.einsum("bhj,bhjd->bhd", torch.softmax(logit, -1), hv).reshape(B, d)
            got = _ops.rel_attention_decode(q, k, v, rel, H, nk, nk - 1, 1, 1, S)
            _close(got, ref.float(), TOL, "cached attention")
    k2 = torch.randn(S, 32, d, generator=g).to(dev)     # the full cache
    v2 = torch.randn(S, 32, d, generator=g).to(dev)
    q = torch.randn(32, d, generator=g).to(dev)
    for nk in (512, 513, 1024):
        one = _ops.rel_attention_decode(q, k2, v2, rel, H, nk, nk - 1, 1, 1, S)
        with _hip.knob("ISI_DECODE_ATTN_SEPARATE_SPLITS", 1):
            two = _ops.rel_attention_decode(q, k2, v2, rel, H, nk, nk - 1, 1, 1, S)
        assert torch.equal(one, two), nk


def test_decode_graph_cache_and_statistics_handoff(golden_dir):
    """Round 6.  (a) `isi_prior_sample_run` keeps the graph executables of its last 8 argument sets: more sets than that (ten
    temperatures), revisited afterwards, give the codes of direct launches (ISI_PRIOR_GRAPH=0) every time -- entries are
    dropped and rebuilt without a stale replay.  (b) The LayerNorm statistics handed from the launch that normalises a row as
    its input to the launch that normalises it as its residual (batch 1, rows-in-registers and tile kernels): the same codes
    as with every launch computing its own (ISI_DECODE_NO_STAT_HANDOFF=1)."""
    import sample as S
    from interactive_spectrogram_inpainting import _hip
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    g = torch.Generator().manual_seed(21)
    cls = {"pitch": torch.tensor([20]), "instrument_family_str": torch.tensor([3])}
    uni = torch.rand(top.target_transformer_sequence_length, 20, generator=g)
    temps = [0.8 + 0.05 * i for i in range(10)]
    run = lambda t, B=1: S.sample_model(top, dev, B, [8, 4], temperature=t, class_conditioning=cls, top_p_sampling_p=0.9,
                                        uniforms=uni[:, :B].contiguous())
    with _hip.knob("ISI_PRIOR_GRAPH", 0):
        direct = [run(t) for t in temps]
    for _ in range(2):                                   # the second sweep meets entries the first one's tail pushed out
        for t, ref in zip(temps, direct):
            assert torch.equal(run(t), ref), t
    assert len({tuple(d.flatten().tolist()) for d in direct}) > 1      # (the temperatures do change the draws)
    for B in (1, 5, 20):
        with _hip.knob("ISI_DECODE_NO_STAT_HANDOFF", 1):
            own = run(1.0, B)
        assert torch.equal(run(1.0, B), own), B
    with _hip.knob("ISI_DECODE_MFMA_ROWS", 1):           # the tile kernels on a batch of 5
        tiled = run(1.0, 5)
        with _hip.knob("ISI_DECODE_STATS_GLOBAL", 1), _hip.knob("ISI_DECODE_NO_STAT_HANDOFF", 1):
            assert torch.equal(run(1.0, 5), tiled)
    assert torch.equal(tiled, run(1.0, 5))


def test_sample_model_large_batch(golden_dir):
    """More than 8 sequences per call (the row kernels take groups of 8 rows): every row equals its single-sequence run."""
    import sample as S
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    B = 11
    g = torch.Generator().manual_seed(13)
    cls = {"pitch": torch.tensor([20]), "instrument_family_str": torch.tensor([3])}
    uni = torch.rand(top.target_transformer_sequence_length, B, generator=g)
    out = S.sample_model(top, dev, B, [8, 4], temperature=1.0, class_conditioning=cls, top_p_sampling_p=0.9, uniforms=uni)
    assert out.shape == (B, 8, 4)
    for b in (0, 7, 8, 10):
        solo = S.sample_model(top, dev, 1, [8, 4], temperature=1.0, class_conditioning=cls, top_p_sampling_p=0.9,
                              uniforms=uni[:, b:b + 1].contiguous())
        assert torch.equal(solo[0], out[b]), b
    # round 5: more than ISI_DECODE_MFMA_ROWS (16) rows run each stage as 32-row GEMM tiles on the fp32 matrix pipe
    # (csrc/prior_decode.hip: row_mfma32_kernel; the 256-row chunks of the B = 259 call below); with the switch at 8 this
    # batch of 11 takes the tile kernel, and with it at 1 a batch of 3 as well: same codes every way
    from interactive_spectrogram_inpainting import _hip
    with _hip.knob("ISI_DECODE_MFMA_ROWS", 8):
        tiled = S.sample_model(top, dev, B, [8, 4], temperature=1.0, class_conditioning=cls, top_p_sampling_p=0.9, uniforms=uni)
    assert torch.equal(tiled, out)
    with _hip.knob("ISI_DECODE_MFMA_ROWS", 1):
        three = S.sample_model(top, dev, 3, [8, 4], temperature=1.0, class_conditioning=cls, top_p_sampling_p=0.9,
                               uniforms=uni[:, :3].contiguous())
        bottom_three = S.sample_model(bottom, dev, 3, [16, 8], temperature=1.0, condition=three, class_conditioning=cls,
                                      top_p_sampling_p=0.9, generator=torch.Generator().manual_seed(5))
    assert torch.equal(three, out[:3])
    ref_bottom = S.sample_model(bottom, dev, 3, [16, 8], temperature=1.0, condition=three, class_conditioning=cls,
                                top_p_sampling_p=0.9, generator=torch.Generator().manual_seed(5))
    assert torch.equal(bottom_three, ref_bottom)
    # more than 256 sequences: decoded in chunks of 256, every row still equals its single-sequence run
    B = 259
    uni = torch.rand(top.target_transformer_sequence_length, B, generator=g)
    out = S.sample_model(top, dev, B, [8, 4], temperature=1.0, class_conditioning=cls, top_p_sampling_p=0.9, uniforms=uni)
    assert out.shape == (B, 8, 4)
    for b in (0, 255, 256, 258):
        solo = S.sample_model(top, dev, 1, [8, 4], temperature=1.0, class_conditioning=cls, top_p_sampling_p=0.9,
                              uniforms=uni[:, b:b + 1].contiguous())
        assert torch.equal(solo[0], out[b]), b


def test_inpainting_operations(golden_dir):
    """The compute behind the reference's /timerange-change, /generate, /erase and /get-audio routes
    (flask_server.py:376-443,685-931,1003-1021): regenerated zones stay inside the mask and the model
    window, everything else is returned untouched, same seed -> same result."""
    import inpainting
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    from GANsynth_pytorch.spectrograms_helper import SpectrogramsHelper
    z, top, bottom = _models(golden_dir)
    dev = _dev()
    cls = {k[5:]: torch.from_numpy(z[k])[:1].to(dev) for k in z.files if k.startswith("cls::")}
    g = torch.Generator().manual_seed(4)
    T_top, T_bot = 8, 16                                   # codemaps twice as long as the models' windows
    top_code = torch.randint(0, 32, (1, 8, T_top), generator=g).to(dev)
    bottom_code = torch.randint(0, 32, (1, 16, T_bot), generator=g).to(dev)
    mask = torch.zeros(1, 8, 4, dtype=torch.bool)
    mask[0, 2:6, 1:3] = True
    start = 3

    def run(layer, seed, m=mask):
        return inpainting.timerange_change(top, bottom, top_code, bottom_code, m, layer, start, 1.0, cls, cls, dev,
                                           generator=torch.Generator().manual_seed(seed), top_p_sampling_p=0.9)
    new_top, new_bottom = run("top", 1)
    assert new_top.shape == top_code.shape and new_bottom.shape == bottom_code.shape
    full_top = torch.zeros(1, 8, T_top, dtype=torch.bool, device=dev)
    full_top[..., start:start + 4] = mask.to(dev)
    assert torch.equal(new_top[~full_top], top_code[~full_top])
    assert int(new_top.min()) >= 0 and int(new_top.max()) < 32
    full_bot = full_top.repeat_interleave(2, -2).repeat_interleave(2, -1)
    assert torch.equal(new_bottom[~full_bot], bottom_code[~full_bot])
    assert (new_bottom[full_bot] != bottom_code[full_bot]).any()
    again_top, again_bottom = run("top", 1)
    assert torch.equal(again_top, new_top) and torch.equal(again_bottom, new_bottom)
    # bottom layer only: the top codemap is returned as is
    mask_b = torch.zeros(1, 16, 8, dtype=torch.bool)
    mask_b[0, 4:9, 2:7] = True
    t2, b2 = run("bottom", 2, mask_b)
    assert torch.equal(t2, top_code)
    fb = torch.zeros(1, 16, T_bot, dtype=torch.bool, device=dev)
    fb[..., 2 * start:2 * start + 8] = mask_b.to(dev)
    assert torch.equal(b2[~fb], bottom_code[~fb]) and (b2[fb] != bottom_code[fb]).any()
    with pytest.raises(ValueError):
        run("middle", 0)
    # generate from scratch
    gt, gb = inpainting.generate(top, bottom, 1.0, cls, cls, dev, generator=torch.Generator().manual_seed(3))
    assert gt.shape == (1, 8, 4) and gb.shape == (1, 16, 8) and int(gt.max()) < 32 and int(gb.max()) < 32
    # erase + audio through a VQ-VAE whose code maps have the models' shapes ([1,2,64,32] -> 8x4 / 16x8)
    torch.manual_seed(9)
    vq = VQVAE(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16,
               num_embeddings=64, resolution_factors={"bottom": 4, "top": 2}).to(dev).eval()
    et, eb = inpainting.erase(vq, gt, gb, mask[0], amplitude=0.5, start_index_top=0)
    assert et.shape == gt.shape and eb.shape == gb.shape and et.dtype == torch.int64
    helper = SpectrogramsHelper(16000, 128, 32, 128).to(dev)           # 64 bins = the decoded height
    audio = inpainting.codes_to_audio(vq, helper, gt, gb)
    assert audio.shape == (1, 32 * 32) and torch.isfinite(audio).all()


def test_flask_routes_round_trip(golden_dir):
    """Same routes / query arguments / JSON schema as the reference server (flask_server.py:376-443,
    685-1021; request example locustfile.py:4-17), served by Flask's test client."""
    import json
    import flask_server
    import inpainting
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    from GANsynth_pytorch.spectrograms_helper import SpectrogramsHelper
    z, top, bottom = _models(golden_dir)
    dev = _dev()

    class Enc:                                   # stands in for sklearn's LabelEncoder
        def __init__(self, classes):
            self.classes = list(classes)

        def transform(self, values):
            return np.array([self.classes.index(v) for v in values])

        def inverse_transform(self, indexes):
            return np.array([self.classes[int(i)] for i in indexes], dtype=object)
    encoders = {"pitch": Enc(range(24, 85)), "instrument_family_str": Enc([f"fam{i}" for i in range(11)])}
    torch.manual_seed(9)
    vq = VQVAE(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16,
               num_embeddings=64, resolution_factors={"bottom": 4, "top": 2}).to(dev).eval()
    helper = SpectrogramsHelper(16000, 128, 32, 128).to(dev)
    # the code database of /sample-from-dataset: items as utils.datasets.lmdb_dataset.LMDBDataset yields them
    gdb = torch.Generator().manual_seed(3)
    database = [(torch.randint(0, 64, (8, 3 + i % 3), generator=gdb), torch.randint(0, 64, (16, 2 * (3 + i % 3)), generator=gdb),
                 {"pitch": torch.tensor([36 + i]), "instrument_family_str": torch.tensor([i % 11])}) for i in range(20)]
    app = flask_server.create_app(vq, top, bottom, encoders, dev, spectrograms_helper=helper, top_p=0.9, seed=0,
                                  codes_dataset=database, spectrograms_upsampling_factor=2)
    c = app.test_client()
    q = "pitch=60&instrument_family_str=fam3&temperature=1.0"
    r = c.get("/generate?" + q)
    assert r.status_code == 200
    body = r.get_json()
    assert np.array(body["top_code"]).shape == (8, 4) and np.array(body["bottom_code"]).shape == (16, 8)
    assert body["top_conditioning"]["pitch"][0][0] == 60 and body["bottom_conditioning"]["instrument_family_str"][3][2] == "fam3"
    mask = np.zeros((8, 4), dtype=bool)
    mask[1:5, 0:2] = True
    payload = dict(body, mask=mask.tolist())
    r2 = c.post("/timerange-change?layer=top&start_index_top=0&uniform_sampling=False&" + q.replace("60", "72"),
                data=json.dumps(payload))
    assert r2.status_code == 200
    b2 = r2.get_json()
    t0, t2 = np.array(body["top_code"]), np.array(b2["top_code"])
    assert (t0[~mask] == t2[~mask]).all()
    assert b2["bottom_conditioning"]["pitch"][2][0] == 72 and b2["bottom_conditioning"]["pitch"][0][0] == 60
    r3 = c.post("/erase?eraser_amplitude=0.3&start_index_top=0", data=json.dumps(payload))
    assert r3.status_code == 200 and np.array(r3.get_json()["bottom_code"]).shape == (16, 8)
    r4 = c.post("/get-audio", data=json.dumps(body))
    assert r4.status_code == 200 and r4.mimetype == "audio/wav" and r4.data[:4] == b"RIFF"
    assert len(r4.data) == 44 + 2 * 32 * 32
    # /top-conditioned-sample (flask_server.py:1049-1115): one bottom codemap per pitch of the range from the bottom prior,
    # batched through the native decoder, decoded and returned as a zip of WAV files -- the end-to-end consumer of batched
    # bottom-prior decoding
    import io
    import zipfile
    r5 = c.post("/top-conditioned-sample?instrument_family_str=fam3&min_pitch=60&max_pitch=66&temperature=1.0&top_p=0.9",
                data=json.dumps(body))
    assert r5.status_code == 200 and r5.mimetype == "application/zip"
    with zipfile.ZipFile(io.BytesIO(r5.data)) as zf:
        names = zf.namelist()
        assert names == [f"fam3-{p_}.wav" for p_ in range(60, 66)]
        wavs = [zf.read(n) for n in names]
    assert all(w[:4] == b"RIFF" and len(w) == 44 + 2 * 32 * 32 for w in wavs)
    assert len({w[44:] for w in wavs}) > 1, "every pitch got the same audio"
    assert c.post("/top-conditioned-sample?instrument_family_str=fam3&min_pitch=66&max_pitch=60&temperature=1.0",
                  data=json.dumps(body)).status_code == 400
    # the compute on its own: per-row conditioning, fixed uniforms -> the rows equal single-row calls (batched == batch 1)
    from sample import sample_model
    cls = {"pitch": torch.tensor([3, 10, 17]), "instrument_family_str": torch.tensor([3])}
    top_code = torch.tensor(body["top_code"]).unsqueeze(0).to(dev)
    u = torch.rand(bottom.target_transformer_sequence_length, 3, generator=torch.Generator().manual_seed(5))
    codes3, audio3 = inpainting.top_conditioned_sample(vq, bottom, helper, top_code, 1.0, cls, dev, top_p_sampling_p=0.9, uniforms=u)
    assert codes3.shape == (3, 16, 8) and audio3.shape == (3, 32 * 32) and torch.isfinite(audio3).all()
    for i in range(3):
        one = sample_model(bottom, dev, 1, bottom.shape, 1.0, condition=top_code,
                           class_conditioning={"pitch": cls["pitch"][i:i + 1], "instrument_family_str": cls["instrument_family_str"]},
                           top_p_sampling_p=0.9, uniforms=u[:, i:i + 1])
        assert torch.equal(one[0], codes3[i])
    # /analyze-audio (flask_server.py:624-667): a WAV upload -> spectrogram -> VQVAE.encode -> codes (+ constant
    # conditioning maps); the codes equal those of the library call on the same samples
    wav = flask_server._wav_bytes(torch.sin(torch.arange(1500) * 0.05) * 0.5, 16000)
    r6 = c.post("/analyze-audio?pitch=64&instrument_family_str=fam2", data={"audio": (io.BytesIO(wav), "note.wav")},
                content_type="multipart/form-data")
    assert r6.status_code == 200
    b6 = r6.get_json()
    # 1500 samples -> rounded to 6 columns of the top map (256 samples each)
    assert np.array(b6["top_code"]).shape == (8, 6) and np.array(b6["bottom_code"]).shape == (16, 12)
    assert b6["top_conditioning"]["pitch"][0][0] == 64 and b6["bottom_conditioning"]["instrument_family_str"][0][0] == "fam2"
    x, rate = flask_server._read_wav(wav)
    assert rate == 16000 and x.numel() == 1500
    res_n = inpainting.top_resolution_n(vq, top, bottom, helper, dev)
    assert res_n == 32 * 32 // 4
    dur = inpainting.adapt_duration(1500, 16000, 4.0, res_n, 4)
    assert dur == 6 * res_n and inpainting.adapt_duration(300, 16000, 4.0, res_n, 4) == 4 * res_n   # at least one window
    assert inpainting.adapt_duration(10 ** 6, 16000, 4.0, res_n, 4) == 64000                          # at most 4 s
    t6, b6c = inpainting.analyze_audio(vq, helper, x, dur, dev)
    assert t6[0].cpu().tolist() == b6["top_code"] and b6c[0].cpu().tolist() == b6["bottom_code"]
    bad = c.post("/analyze-audio?pitch=64&instrument_family_str=fam2", data={"audio": (io.BytesIO(b"nonsense"), "x.wav")},
                 content_type="multipart/form-data")
    assert bad.status_code == 400
    # /sample-from-dataset (flask_server.py:333-372,446-514): a stored pair meeting the constraints, cut / continued with its
    # last column to `duration_top`; item i carries pitch class index 36 + i = pitch 60 + i, family i % 11
    r7 = c.get("/sample-from-dataset?duration_top=4&pitch=65")               # class index 41 = item 5 (3 + 5 % 3 = 5 columns: cut)
    assert r7.status_code == 200
    b7 = r7.get_json()
    assert b7["top_code"] == database[5][0][:, :4].tolist() and b7["bottom_code"] == database[5][1][:, :8].tolist()
    assert b7["top_conditioning"]["pitch"][0][0] == 65 and b7["bottom_conditioning"]["instrument_family_str"][1][1] == "fam5"
    r8 = c.get("/sample-from-dataset?duration_top=6&pitch_class=0&octave=6&instrument_family_str=fam1")   # pitch 72 = item 12, 3 columns
    assert r8.status_code == 200
    b8 = np.array(r8.get_json()["top_code"])
    assert b8.shape == (8, 6) and (b8[:, :3] == database[12][0].numpy()).all() and (b8[:, 3:] == b8[:, 2:3]).all()
    assert np.array(r8.get_json()["bottom_code"]).shape == (16, 12)
    assert c.get("/sample-from-dataset?duration_top=4&pitch=100").status_code == 404
    # /test-generate (flask_server.py:517-552): random codemaps of the models' shapes
    r9 = c.get("/test-generate?pitch=50&instrument_family_str=fam0")
    t9 = np.array(r9.get_json()["top_code"])
    assert r9.status_code == 200 and t9.shape == (8, 4) and t9.min() >= 0 and t9.max() < 64
    assert r9.get_json()["bottom_conditioning"]["pitch"][15][7] == 50
    # /get-spectrogram-image (flask_server.py:1024-1046): PNG of the decoded log-magnitude, upsampled twice
    import struct
    import zlib
    r10 = c.post("/get-spectrogram-image", data=json.dumps(body))
    assert r10.status_code == 200 and r10.mimetype == "image/png" and r10.data[:8] == b"\x89PNG\r\n\x1a\n"
    w_, h_, depth, colour = struct.unpack(">IIBB", r10.data[16:26])
    assert (w_, h_, depth, colour) == (2 * 32, 2 * 64, 8, 2)                 # decode_code of [8,4] / [16,8] codes: 64 x 32 bins
    n_idat = struct.unpack(">I", r10.data[33:37])[0]
    assert r10.data[37:41] == b"IDAT"
    raw = np.frombuffer(zlib.decompress(r10.data[41:41 + n_idat]), dtype=np.uint8).reshape(h_, 1 + 3 * w_)
    assert (raw[:, 0] == 0).all()
    px = raw[:, 1:].reshape(h_, w_, 3)
    spec = torch.nn.functional.interpolate(vq.decode_code(torch.tensor(body["top_code"])[None].to(dev),
                                                          torch.tensor(body["bottom_code"])[None].to(dev))[0, 0][None, None],
                                           mode="bilinear", scale_factor=2)[0, 0].flip(0).cpu().numpy()
    iy, ix = np.unravel_index(spec.argmax(), spec.shape)
    assert tuple(px[iy, ix]) == (253, 231, 37)                               # the maximum is drawn in viridis' last colour
    iy, ix = np.unravel_index(spec.argmin(), spec.shape)
    assert tuple(px[iy, ix]) == (68, 1, 84)


# ---------------------------------------------------------------------------------------------------------------
# BASELINE sizes (configs 4 / 5): seq 1025 self-attention at head_dim 64 x 8 heads, the bottom prior's 4100 x 1025
# cross-attention with four tokens per event, the full d_model 512 / 6 + 8 layer priors, sampling on a [32,32] map.

FULL = dict(n_class=512, channel=256, kernel_size=5, n_block=4, n_res_block=4, res_channel=256, d_model=512,
            embeddings_dim=32, positional_embeddings_dim=16, use_relative_transformer=True,
            predict_frequencies_first=True, conditional_model=True, class_conditioning_prepend_to_dummy_input=True,
            class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
            class_conditioning_embedding_dim_per_modality={"instrument_family_str": 64, "pitch": 64})


def _full_top(seed=2):
    from interactive_spectrogram_inpainting.priors.transformer import SelfAttentiveVQTransformer
    torch.manual_seed(seed)
    return SelfAttentiveVQTransformer(shape=[32, 32], condition_shape=[32, 32], self_conditional_model=True,
                                      add_mask_token_to_symbols=True, **FULL).to(_dev()).eval()


def _full_bottom(seed=3):
    from interactive_spectrogram_inpainting.priors.transformer import UpsamplingVQTransformer
    torch.manual_seed(seed)
    return UpsamplingVQTransformer(shape=[64, 64], condition_shape=[32, 32], **FULL).to(_dev()).eval()


@pytest.mark.parametrize("Sq,Sk,Cq,Ck,mode", [(1025, 1025, 1, 1, 1), (1025, 1025, 1, 1, 2), (4100, 1025, 4, 1, 0),
                                              (4100, 4100, 4, 4, 1)])
@pytest.mark.parametrize("precision", ["f32", "bf16x3"])
def test_rel_attention_at_baseline_sizes(Sq, Sk, Cq, Ck, mode, precision, monkeypatch):
    """The attention shapes the metric is quoted on: top prior self-attention (S = 1025, causal in the decoder,
    anti-causal in the self-conditional encoder), bottom prior cross-attention 4100 x 1025 (4 tokens per event) and
    its causal self-attention at S = 4100; head_dim 64, 8 heads."""
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors import _ops
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", precision)
    hd, H, B = 64, 8, 1
    d = hd * H
    torch.manual_seed(Sq + Sk + mode)
    Eq, Ek = -(-Sq // Cq), -(-Sk // Ck)
    q, k, v = torch.randn(Sq, B, d), torch.randn(Sk, B, d), torch.randn(Sk, B, d)
    rel = torch.randn(H, Eq + Ek - 1, hd) * 0.5
    hq = q.reshape(Sq, B, H, hd).permute(1, 2, 0, 3)
    hk = k.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    hv = v.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    logits = hq @ hk.transpose(-1, -2)
    qe = torch.einsum("bhid,hrd->bhir", hq, rel)
    logits = (logits + qe.gather(3, P.rel_index(Sq, Sk, Cq, Ck, Ek).expand(B, H, Sq, Sk))) / math.sqrt(hd)
    del qe
    if mode == 1:
        logits += P.causal_mask(Sq)
    elif mode == 2:
        logits += P.causal_mask(Sq).t()
    ref = (torch.softmax(logits, -1) @ hv).permute(2, 0, 1, 3).reshape(Sq, B, d)
    del logits
    dev = _dev()
    got = _ops.rel_attention(q.to(dev), k.to(dev), v.to(dev), rel.to(dev), H, Cq, Ck, Ek, mask_mode=mode)
    _close(got, ref, TOL, f"rel_attention {Sq}x{Sk} mode {mode} {precision}")


def test_full_size_priors_forward_against_spec():
    """One forward of the top prior ([32,32]: 1025-token source and target, d_model 512, 6 encoder + 8 decoder layers,
    8 heads) and of the bottom prior ([64,64] conditioned on [32,32]: 4100-token target) at B = 1 against the
    specification in oracle/prior_oracle.py: encoder memory and logits within 5e-4 of their maxima (north_star: 1e-3)."""
    dev = _dev()
    g = torch.Generator().manual_seed(17)
    top_code = torch.randint(0, 512, (1, 32, 32), generator=g).to(dev)
    bottom_code = torch.randint(0, 512, (1, 64, 64), generator=g).to(dev)
    mask = (torch.rand(1, 32, 32, generator=g) < 0.5).to(dev)
    cls = {"pitch": torch.tensor([[24]], device=dev), "instrument_family_str": torch.tensor([[3]], device=dev)}
    top, bottom = _full_top(), _full_bottom()
    for name, m, (src, tgt) in (("top", top, top.to_sequences(top_code, top_code, class_conditioning=cls, mask=mask)),
                                ("bottom", bottom, bottom.to_sequences(bottom_code, top_code, class_conditioning=cls))):
        assert tgt.shape[1] == (1025 if name == "top" else 4100) and src.shape[1] == 1025
        ref_logits, ref_mem, _ = _oracle_logits(m, src, tgt)
        logits, memory = m(tgt, src)
        _close(memory, ref_mem, 5e-4, name + " memory")
        _close(logits, ref_logits, 5e-4, name + " logits")
        err = ((logits.cpu() - ref_logits).abs().max() / ref_logits.abs().max()).item()
        print(f"[{name} prior, d_model 512, 6+8 layers] max |logits - spec| / max |spec| = {err:.2e}")


def test_sample_model_at_baseline_size_matches_full_pass_sampling():
    """BASELINE config 5: KV-cached sampling on the [32,32] top map (1024 tokens, d_model 512, 6 + 8 layers) with a
    32-token mask (one column of the map) == the reference's loop (one full decoder pass per masked token,
    sample.py:268-305) drawing from the same uniforms; unmasked codes are kept."""
    import sample as S
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors.transformer import Seq2SeqInputKind
    top = _full_top()
    dev = _dev()
    B = 1
    g = torch.Generator().manual_seed(19)
    init = torch.randint(0, 512, (B, 32, 32), generator=g)
    mask = torch.zeros(1, 32, 32, dtype=torch.bool)
    mask[:, :, 17] = True                                      # 32 tokens: column 17
    cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
    uni = torch.rand(top.target_transformer_sequence_length, B, generator=g)
    got = S.sample_model(top, dev, B, [32, 32], temperature=1.0, class_conditioning=cls, initial_code=init.clone(),
                         mask=mask, top_p_sampling_p=0.8, uniforms=uni)
    keep = ~mask.expand(B, -1, -1)
    assert torch.equal(got.cpu()[keep], init[keep]), "unmasked positions must keep initial_code"
    clsd = {k: v.long().expand(B).reshape(B, 1).to(dev) for k, v in cls.items()}
    codemap = init.clone().to(dev)
    src, tgt = top.to_sequences(codemap, codemap, class_conditioning=clsd, mask=mask.to(dev))
    seq = top.target_codemaps_helper.to_sequence(codemap).clone()
    mseq = top.target_codemaps_helper.to_sequence(mask.to(dev))[0].cpu().numpy()
    assert int(mseq.sum()) == 32
    memory = None
    for i, is_masked in enumerate(mseq):
        if not is_masked:
            continue
        logits, memory = top(tgt, src, memory=memory)
        s = _ops.sample_rows(logits[:, i].contiguous(), 1.0, 0, 0.8, uni[i])
        seq[:, i] = s
        tgt[:, i + 1, :top.embeddings_effective_dim] = top.embed_data(s, Seq2SeqInputKind.Target)
    ref = top.target_codemaps_helper.to_time_frequency_map(seq)
    # a draw sits on a CDF step: rounding differences between the cached row and the full pass may move a code only
    # if the uniform falls within float rounding of a step -- identical in practice
    assert (got != ref).sum().item() <= 1, f"{(got != ref).sum().item()} of 32 sampled codes differ"


@pytest.mark.parametrize("M,N,K", [(1, 512, 512), (3, 192, 64), (8, 512, 2048), (17, 512, 512), (20, 48, 96), (32, 1536, 512),
                                   (33, 512, 2048), (40, 1024, 768), (64, 2048, 512), (256, 512, 512)])
def test_decode_stage_kernels_against_torch(M, N, K):
    """isi_decode_stage_f32 = one stage of the KV-cached decoding loop on M new rows (csrc/prior_decode.hip): LayerNorm of the
    input rows folded in, bias, LayerNorm-ed residual, ReLU -- against torch in fp64, through every kernel family: the
    one-row GEMV, the GEMV looping over groups of rows (ISI_DECODE_MFMA_ROWS = 256), and the 32-row fp32-MFMA tiles
    (ISI_DECODE_MFMA_ROWS = 1; K in one chunk, in several, with a short last chunk; N not a multiple of 32; the K chunks
    side by side with the finishing launch when a workspace is given, one after the other without).  A row of a batch
    on the GEMV family equals the row alone up to the compiler's contraction of the same operations (a few ulp)."""
    import ctypes as C
    from interactive_spectrogram_inpainting import _hip
    dev = _dev()
    L = _hip.lib()
    g = torch.Generator().manual_seed(M * 7 + N + K)
    x = torch.randn(M, K, generator=g).to(dev)
    W = (torch.randn(N, K, generator=g) / math.sqrt(K)).to(dev)
    bias = torch.randn(N, generator=g).to(dev)
    ln_g, ln_b = (1 + 0.1 * torch.randn(K, generator=g)).to(dev), (0.1 * torch.randn(K, generator=g)).to(dev)
    res = torch.randn(M, N, generator=g).to(dev)
    res_g, res_b = (1 + 0.1 * torch.randn(N, generator=g)).to(dev), (0.1 * torch.randn(N, generator=g)).to(dev)
    s = C.c_void_p(torch.cuda.current_stream().cuda_stream)

    def run(ln, rs, rln, relu, rows_switch, with_ws=True, m=M, x_=None):
        x_ = x if x_ is None else x_
        out = torch.full((m, N), float("nan"), device=dev)
        nws = L.isi_decode_stage_workspace_floats(m, N, K)
        ws = torch.empty(nws, device=dev) if with_ws else None
        with _hip.knob("ISI_DECODE_MFMA_ROWS", rows_switch):
            _hip.check(L.isi_decode_stage_f32(x_.data_ptr(), K, ln_g.data_ptr() if ln else None, ln_b.data_ptr() if ln else None,
                                              W.data_ptr(), bias.data_ptr(), res.data_ptr() if rs else None, N,
                                              res_g.data_ptr() if rln else None, res_b.data_ptr() if rln else None,
                                              out.data_ptr(), N, m, N, K, int(relu), 1e-5,
                                              ws.data_ptr() if ws is not None else None, nws if ws is not None else 0, s),
                       "isi_decode_stage_f32")
        return out

    def ref(ln, rs, rln, relu):
        xd = x.double()
        if ln:
            xd = torch.nn.functional.layer_norm(xd, (K,), ln_g.double(), ln_b.double(), 1e-5)
        y = xd @ W.double().t() + bias.double()
        if rs:
            r = res.double()
            if rln:
                r = torch.nn.functional.layer_norm(r, (N,), res_g.double(), res_b.double(), 1e-5)
            y = y + r
        return torch.relu(y) if relu else y

    for ln, rs, rln, relu in ((True, True, N <= 512, False), (False, True, False, True), (True, False, False, True),
                              (False, False, False, False)):
        want = ref(ln, rs, rln, relu)
        scale = float(want.abs().max())
        gemv = run(ln, rs, rln, relu, 256)
        assert float((gemv.double() - want).abs().max()) < 2e-5 * scale, ("gemv", ln, rs, rln, relu)
        if M >= 2 and K % 8 == 0:
            for with_ws in (True, False):
                tiles = run(ln, rs, rln, relu, 1, with_ws)
                assert float((tiles.double() - want).abs().max()) < 2e-5 * scale, ("tiles", ln, rs, rln, relu, with_ws)
    if 1 < M <= 40:       # a row of a batch against the same row alone (GEMV family: same operations, same order)
        batch = run(True, True, N <= 512, False, 256)
        for r_ in (0, M - 1):
            res_row = res[r_:r_ + 1].clone()
            out1 = torch.full((1, N), float("nan"), device=dev)
            _hip.check(L.isi_decode_stage_f32(x[r_:r_ + 1].contiguous().data_ptr(), K, ln_g.data_ptr(), ln_b.data_ptr(), W.data_ptr(),
                                              bias.data_ptr(), res_row.data_ptr(), N, res_g.data_ptr() if N <= 512 else None,
                                              res_b.data_ptr() if N <= 512 else None, out1.data_ptr(), N, 1, N, K, 0, 1e-5,
                                              None, 0, s), "isi_decode_stage_f32")
            assert float((out1[0] - batch[r_]).abs().max()) <= 4e-6 * float(batch[r_].abs().max()), r_


def test_batched_decoding_on_matrix_tiles_at_baseline_size():
    """Round 5: with more than 16 sequences a decoding stage runs as 32-row tiles on the fp32 matrix pipe
    (csrc/prior_decode.hip: row_mfma32_kernel; K = 512 in one LDS chunk, K = 2048 -- linear2 -- in four, LayerNorm folded into
    the staging, residual LayerNorm in the epilogue).  BASELINE config 5's top prior (d_model 512, 6 + 8 layers, [32,32] map), a
    32-token mask, B = 20: every checked row draws the codes of its single-sequence run (GEMV kernels; the tile kernel sums in
    another order, so a draw sitting on a CDF step may move: at most one code per row)."""
    import sample as S
    top = _full_top()
    dev = _dev()
    B = 20
    g = torch.Generator().manual_seed(29)
    init = torch.randint(0, 512, (B, 32, 32), generator=g)
    mask = torch.zeros(1, 32, 32, dtype=torch.bool)
    mask[:, :, 11] = True
    cls = {"pitch": torch.tensor([30]), "instrument_family_str": torch.tensor([5])}
    uni = torch.rand(top.target_transformer_sequence_length, B, generator=g)
    out = S.sample_model(top, dev, B, [32, 32], temperature=1.0, class_conditioning=cls, initial_code=init.clone(), mask=mask,
                         top_p_sampling_p=0.8, uniforms=uni)
    keep = ~mask.expand(B, -1, -1)
    assert torch.equal(out.cpu()[keep], init[keep])
    for b in (0, 7, 19):
        solo = S.sample_model(top, dev, 1, [32, 32], temperature=1.0, class_conditioning=cls, initial_code=init[b:b + 1].clone(),
                              mask=mask, top_p_sampling_p=0.8, uniforms=uni[:, b:b + 1].contiguous())
        assert (solo[0] != out[b]).sum().item() <= 1, (b, (solo[0] != out[b]).sum().item())


@pytest.mark.parametrize("Sq,Sk,Cq,Ck,mode", [(1025, 1025, 1, 1, 1), (4100, 1025, 4, 1, 0), (200, 200, 1, 1, 2), (77, 150, 2, 1, 0)])
def test_rel_attention_bf16_mode(Sq, Sk, Cq, Ck, mode, monkeypatch):
    """precision = 'bf16' (north_star: relative-attention contractions on MFMA bf16, single term): operands rounded
    to bf16 (2^-9 relative), fp32 accumulation / logits / softmax.  Error against the fp32 specification is that of
    the operand rounding -- checked at 2e-2 of the output's maximum here and REPORTED by bench.py's attention leg
    next to the three-term split's 1e-5; masks, skew and ragged tiles as in the other modes."""
    from oracle import prior_oracle as P
    from interactive_spectrogram_inpainting.priors import _ops
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", "bf16")
    hd, H, B = 64, 8, 1
    d = hd * H
    torch.manual_seed(Sq + mode)
    Eq, Ek = -(-Sq // Cq), -(-Sk // Ck)
    q, k, v = torch.randn(Sq, B, d), torch.randn(Sk, B, d), torch.randn(Sk, B, d)
    rel = torch.randn(H, Eq + Ek - 1, hd) * 0.5
    hq = q.reshape(Sq, B, H, hd).permute(1, 2, 0, 3)
    hk = k.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    hv = v.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    logits = hq @ hk.transpose(-1, -2)
    qe = torch.einsum("bhid,hrd->bhir", hq, rel)
    logits = (logits + qe.gather(3, P.rel_index(Sq, Sk, Cq, Ck, Ek).expand(B, H, Sq, Sk))) / math.sqrt(hd)
    if mode == 1:
        logits += P.causal_mask(Sq)
    elif mode == 2:
        logits += P.causal_mask(Sq).t()
    ref = (torch.softmax(logits, -1) @ hv).permute(2, 0, 1, 3).reshape(Sq, B, d)
    dev = _dev()
    got = _ops.rel_attention(q.to(dev), k.to(dev), v.to(dev), rel.to(dev), H, Cq, Ck, Ek, mask_mode=mode)
    err = ((got.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert torch.isfinite(got).all() and err < 2e-2, err
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", "bf16x3")
    x3 = _ops.rel_attention(q.to(dev), k.to(dev), v.to(dev), rel.to(dev), H, Cq, Ck, Ek, mask_mode=mode)
    assert ((x3.cpu() - ref).abs().max() / ref.abs().max()).item() < err, "the three-term split must be more accurate"


def _attention_spec(q, k, v, rel, H, Cq, Ck, Ek, mode):
    """softmax((q k^T + skew(q e^T)) / sqrt(hd) + mask) v on the CPU (oracle/prior_oracle.py's index map)."""
    from oracle import prior_oracle as P
    Sq, B, d = q.shape
    Sk, hd = k.shape[0], d // H
    hq = q.reshape(Sq, B, H, hd).permute(1, 2, 0, 3)
    hk = k.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    hv = v.reshape(Sk, B, H, hd).permute(1, 2, 0, 3)
    logits = hq @ hk.transpose(-1, -2)
    if rel is not None:
        qe = torch.einsum("bhid,hrd->bhir", hq, rel)
        logits = logits + qe.gather(3, P.rel_index(Sq, Sk, Cq, Ck, Ek).expand(B, H, Sq, Sk))
    logits = logits / math.sqrt(hd)
    if mode == 1:
        logits += P.causal_mask(Sq)
    elif mode == 2:
        logits += P.causal_mask(Sq).t()
    return (torch.softmax(logits, -1) @ hv).permute(2, 0, 1, 3).reshape(Sq, B, d)


@pytest.mark.parametrize("Sq,Sk,Cq,Ck,mode", [(1025, 1025, 1, 1, 1), (4100, 1025, 4, 1, 0), (200, 200, 1, 1, 2), (77, 150, 2, 1, 0)])
def test_rel_attention_f16_mode(Sq, Sk, Cq, Ck, mode, monkeypatch):
    """precision = 'f16' (round 4): single-term products like 'bf16' -- a third of the three-term split's matrix work --
    with operands rounded to f16's 11 significand bits: the error against the fp32 specification has to stay inside
    north_star's 1e-3 of the output's maximum (the bf16 mode's 8 bits give ~3e-3) and below the bf16 mode's."""
    from interactive_spectrogram_inpainting.priors import _ops
    hd, H, B = 64, 8, 1
    d = hd * H
    torch.manual_seed(Sq + mode)
    Eq, Ek = -(-Sq // Cq), -(-Sk // Ck)
    q, k, v = torch.randn(Sq, B, d), torch.randn(Sk, B, d), torch.randn(Sk, B, d)
    rel = torch.randn(H, Eq + Ek - 1, hd) * 0.5
    ref = _attention_spec(q, k, v, rel, H, Cq, Ck, Ek, mode)
    dev = _dev()
    errs = {}
    for prec in ("f16", "bf16"):
        monkeypatch.setattr(_ops, "ATTENTION_PRECISION", prec)
        got = _ops.rel_attention(q.to(dev), k.to(dev), v.to(dev), rel.to(dev), H, Cq, Ck, Ek, mask_mode=mode)
        assert torch.isfinite(got).all()
        errs[prec] = ((got.cpu() - ref).abs().max() / ref.abs().max()).item()
    assert errs["f16"] < 1e-3 and errs["f16"] < errs["bf16"], errs


@pytest.mark.parametrize("hd,H,B,Sq,Sk,Cq,Ck,mode", [
    (64, 8, 40, 300, 300, 1, 1, 1),     # 320 (batch, head) pairs > 256 CUs: one persistent workgroup walks a pair's three blocks
    (32, 8, 33, 513, 513, 1, 1, 1),     # 264 pairs, five blocks incl. the one-row ragged block, three-term at 64-key tiles
    (64, 3, 11, 1025, 1025, 1, 1, 2),   # 33 pairs x 9 blocks > 256: snake order, anti-causal (block 0 heaviest)
    (64, 8, 5, 900, 640, 2, 2, 0),      # unmasked cross-attention, general channel layout, pairs not a multiple of 8
    (16, 4, 9, 700, 700, 1, 1, 1),
])
@pytest.mark.parametrize("precision", ["bf16x3", "f16"])
def test_rel_attention_persistent_blocks(hd, H, B, Sq, Sk, Cq, Ck, mode, precision, monkeypatch):
    """rel_attention_fwd2.hip deals a (batch, head) pair's query blocks to fewer, persistent workgroups once the launch
    would exceed one workgroup per CU (snake order over the blocks sorted by cost; the next block's first key step is
    requested during the last step of the current one): against the exact-fp32 kernel, which runs one block per
    workgroup, and against the round-3 kernel behind ISI_ATTN_OLD_FWD."""
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.priors import _ops
    torch.manual_seed(Sq + B)
    dev = _dev()
    d = hd * H
    Eq, Ek = -(-Sq // Cq), -(-Sk // Ck)
    q, k, v = (torch.randn(s, B, d, device=dev) for s in (Sq, Sk, Sk))
    rel = torch.randn(H, Eq + Ek - 1, hd, device=dev) * 0.5
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", "f32")
    lse0 = torch.empty(B, H, Sq, device=dev)
    ref = _ops.rel_attention(q, k, v, rel, H, Cq, Ck, Ek, mask_mode=mode, lse=lse0)
    monkeypatch.setattr(_ops, "ATTENTION_PRECISION", precision)
    lse = torch.empty(B, H, Sq, device=dev)
    got = _ops.rel_attention(q, k, v, rel, H, Cq, Ck, Ek, mask_mode=mode, lse=lse)
    tol = 3e-5 if precision == "bf16x3" else 2.5e-3
    _close(got, ref, tol, f"persistent blocks {precision}")
    assert (lse - lse0).abs().max().item() < 30 * tol
    if precision == "bf16x3":
        with _hip.knob("ISI_ATTN_OLD_FWD", 1):
            old = _ops.rel_attention(q, k, v, rel, H, Cq, Ck, Ek, mask_mode=mode)
        _close(got, old, 3e-5, "64-key-tile kernel vs round-3 kernel")
