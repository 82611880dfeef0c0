"""Generate tests/golden/*.npz by running the REFERENCE itself (build container only).

TEST INFRASTRUCTURE.  Imports the reference package from /root/reference with
`sys.modules` stubs for its absent third-party dependencies (fastai,
GANsynth_pytorch, discretization), runs its own `VQVAE` / `QuantizedBottleneck`
/ `Rosinality*` / codemap helper classes on fixed-seed inputs and stores inputs
+ weights + outputs as small .npz fixtures.  The fixtures are data only; no
reference source travels.  Run:

    PYTHONDONTWRITEBYTECODE=1 python oracle/make_golden.py

The reference never runs on the GPU box: tests read only the .npz files.
"""
from __future__ import annotations

import os
import sys
import types
import pathlib

import numpy as np
import torch
from torch import nn

sys.dont_write_bytecode = True
REF = pathlib.Path("/root/reference")
OUT = pathlib.Path(__file__).resolve().parent.parent / "tests" / "golden"


def _install_stubs():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        if "." in name:  # expose as attribute of the parent package too
            parent, leaf = name.rsplit(".", 1)
            if parent in sys.modules:
                setattr(sys.modules[parent], leaf, m)
        return m

    class _Dummy(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()

    def _noop(*a, **k):
        return None

    def _delegates(*a, **k):
        def deco(f):
            return f
        return deco

    mod("fastai")
    mod("fastai.vision")
    unet = mod("fastai.vision.models.unet", UnetBlock=_Dummy, _get_sz_change_idxs=_noop)
    xresnet = mod("fastai.vision.models.xresnet", XResNet=_Dummy, delegates=_delegates)
    mod("fastai.vision.models", unet=unet, xresnet=xresnet)
    mod("fastai.layers", BatchNorm=_Dummy, ConvLayer=_Dummy, SequentialEx=_Dummy,
        PixelShuffle_ICNR=_Dummy, SigmoidRange=_Dummy, ResBlock=_Dummy)
    mod("fastai.torch_core", apply_init=_noop, defaults=types.SimpleNamespace(activation=nn.ReLU),
        Module=_Dummy)
    mod("fastai.callback")
    mod("fastai.callback.hook", model_sizes=_noop, dummy_eval=_noop)
    mod("GANsynth_pytorch")
    mod("GANsynth_pytorch.loader", make_masked_phase_transform=_noop)

    class DataNormalizerStatistics:
        def __init__(self, **kw):
            self.__dict__.update(kw)

    mod("GANsynth_pytorch.normalizer", DataNormalizer=_Dummy,
        DataNormalizerStatistics=DataNormalizerStatistics)
    mod("discretization", ProductVectorQuantizer=_Dummy)

    # the prior's layers live in the absent VQCPCB package: stand-ins that only
    # record their constructor arguments (the wrapper around them is what we pin)
    class _Layer(nn.Module):
        def __init__(self, *a, **k):
            super().__init__()
            self.kwargs = k

    mod("VQCPCB")
    mod("VQCPCB.transformer")
    mod("VQCPCB.transformer.transformer_custom", TransformerCustom=_Layer, TransformerDecoderCustom=_Layer,
        TransformerEncoderCustom=_Layer, TransformerDecoderLayerCustom=_Layer,
        TransformerEncoderLayerCustom=_Layer, TransformerAlignedDecoderLayerCustom=_Layer)


def _np(sd):
    return {k: v.detach().cpu().numpy() for k, v in sd.items()}


def _save(name, **arrays):
    OUT.mkdir(parents=True, exist_ok=True)
    path = OUT / name
    np.savez_compressed(path, **arrays)
    print(f"wrote {path} ({path.stat().st_size / 1024:.1f} KiB)")


@torch.no_grad()
def _calibrate_codebooks(model, x, seed):
    """Replace each codebook by a random subset of that level's own
    pre-quantisation vectors (non-degenerate code usage)."""
    g = torch.Generator().manual_seed(seed)
    enc_b = model.enc_b(x.clone())
    enc_t = model.enc_t(enc_b)
    z_t = model.quantize_conv_t(enc_t).permute(0, 2, 3, 1)
    flat = z_t.reshape(-1, model.embed_dim)
    pick = torch.randint(0, flat.shape[0], (model.n_embed_t,), generator=g)
    model.quantize_t.embed.copy_(flat[pick].t())
    model.quantize_t.embed_avg.copy_(model.quantize_t.embed)
    q_t, *_ = model.quantize_t(z_t)
    dec_t = model.dec_t(q_t.permute(0, 3, 1, 2))
    z_b = model.quantize_conv_b(torch.cat([dec_t, enc_b], 1)).permute(0, 2, 3, 1)
    flat = z_b.reshape(-1, model.embed_dim)
    pick = torch.randint(0, flat.shape[0], (model.n_embed_b,), generator=g)
    model.quantize_b.embed.copy_(flat[pick].t())
    model.quantize_b.embed_avg.copy_(model.quantize_b.embed)


@torch.no_grad()
def vqvae_fixture(name, ctor_kwargs, in_shape, seed):
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    torch.manual_seed(seed)
    model = VQVAE(**ctor_kwargs).eval()
    x = torch.randn(*in_shape)
    _calibrate_codebooks(model, torch.randn(*in_shape), seed + 100)
    sd = {k: v.clone() for k, v in model.state_dict().items()}

    enc_b = model.enc_b(x.clone())
    enc_t = model.enc_t(enc_b.clone())
    z_t = model.quantize_conv_t(enc_t)
    q_t, q_b, diff, id_t, id_b, perp_t, perp_b = model.encode(x.clone())
    dec, diff2, perp_t2, perp_b2, id_t2, id_b2 = model(x.clone())
    assert torch.equal(id_t, id_t2) and torch.equal(id_b, id_b2)
    dec_code = model.decode_code(id_t, id_b)
    used_t = len(torch.unique(id_t)) / model.n_embed_t
    used_b = len(torch.unique(id_b)) / model.n_embed_b
    print(f"{name}: code usage top {used_t:.2f} bottom {used_b:.2f}, "
          f"perplexity {perp_t.item():.1f}/{perp_b.item():.1f}")
    arrays = {"w::" + k: v for k, v in _np(sd).items()}
    arrays.update(
        x=x.numpy(), enc_b=enc_b.numpy(), enc_t=enc_t.numpy(), z_t=z_t.numpy(),
        quant_t=q_t.contiguous().numpy(), quant_b=q_b.contiguous().numpy(),
        diff=diff.numpy(), id_t=id_t.numpy(), id_b=id_b.numpy(),
        perplexity_t=perp_t.numpy(), perplexity_b=perp_b.numpy(),
        dec=dec.numpy(), dec_code=dec_code.numpy(),
        cfg_in_channel=np.int64(ctor_kwargs["in_channel"]),
        cfg_num_hidden_channels=np.int64(ctor_kwargs.get("num_hidden_channels", 128)),
        cfg_n_res_block=np.int64(ctor_kwargs.get("n_res_block", 2)),
        cfg_num_residual_channels=np.int64(ctor_kwargs.get("num_residual_channels", 32)),
        cfg_embed_dim=np.int64(ctor_kwargs.get("embed_dim", 64)),
        cfg_num_embeddings=np.int64(ctor_kwargs.get("num_embeddings", 512)),
        cfg_factor_bottom=np.int64(ctor_kwargs.get("resolution_factors", {"bottom": 4})["bottom"]),
        cfg_factor_top=np.int64(ctor_kwargs.get("resolution_factors", {"top": 2})["top"]),
        cfg_groups=np.int64(ctor_kwargs.get("groups", 1)),
    )
    _save(name, **arrays)


@torch.no_grad()
def resblock_fixture():
    from interactive_spectrogram_inpainting.vqvae.encoder_decoder import RosinalityResBlock
    torch.manual_seed(11)
    blk = RosinalityResBlock(16, 8).eval()
    x = torch.randn(2, 16, 5, 7)
    x_in = x.clone()
    y = blk(x_in)
    # x_in has been overwritten by relu(x): proof of the in-place semantics
    assert torch.equal(x_in, torch.relu(x))
    _save("resblock.npz", x=x.numpy(), y=y.numpy(), x_after=x_in.numpy(),
          **{"w::" + k: v for k, v in _np(blk.state_dict()).items()})


@torch.no_grad()
def layer_fixtures():
    """Each layer type alone at odd sizes."""
    torch.manual_seed(12)
    out = {}
    specs = {
        "conv_k4s2": (nn.Conv2d(6, 10, 4, stride=2, padding=1), (2, 6, 10, 14)),
        "conv_k4s2_odd": (nn.Conv2d(3, 5, 4, stride=2, padding=1), (1, 3, 9, 13)),
        "conv_k3": (nn.Conv2d(7, 9, 3, padding=1), (2, 7, 5, 11)),
        "conv_k1": (nn.Conv2d(12, 6, 1), (2, 12, 3, 5)),
        "convT_k4s2": (nn.ConvTranspose2d(6, 10, 4, stride=2, padding=1), (2, 6, 5, 7)),
        "convT_k4s2_c2": (nn.ConvTranspose2d(8, 2, 4, stride=2, padding=1), (1, 8, 6, 9)),
    }
    for name, (layer, shape) in specs.items():
        x = torch.randn(*shape)
        y = layer(x)
        out[name + "::x"] = x.numpy()
        out[name + "::y"] = y.numpy()
        out[name + "::weight"] = layer.weight.detach().numpy()
        out[name + "::bias"] = layer.bias.detach().numpy()
    _save("layers.npz", **out)


@torch.no_grad()
def quantizer_fixtures():
    from interactive_spectrogram_inpainting.vqvae.bottleneck import QuantizedBottleneck
    torch.manual_seed(13)
    # (1) Gaussian case at the default D=64, K=512
    q = QuantizedBottleneck(64, 512).eval()
    z = torch.randn(3, 8, 16, 64)
    quant, diff, ind, perp = q(z)
    out = dict(g_z=z.numpy(), g_embed=q.embed.numpy(), g_quant=quant.numpy(), g_diff=diff.numpy(),
               g_ind=ind.numpy(), g_perp=perp.numpy(),
               g_embed_code=q.embed_code(ind).numpy())
    # (2) engineered exact tie: codes 5 and 9 identical, inputs equal to them
    q2 = QuantizedBottleneck(8, 16).eval()
    q2.embed[:, 9] = q2.embed[:, 5]
    q2.embed[:, 12] = q2.embed[:, 2]
    z2 = torch.randn(1, 4, 6, 8)
    z2[0, 0, 0] = q2.embed[:, 5]
    z2[0, 1, 2] = q2.embed[:, 12] * 1.0
    z2[0, 3, 3] = 0.5 * (q2.embed[:, 1] + q2.embed[:, 3])
    quant2, diff2, ind2, perp2 = q2(z2)
    assert ind2[0, 0, 0].item() == 5 and ind2[0, 1, 2].item() == 2
    out.update(t_z=z2.numpy(), t_embed=q2.embed.numpy(), t_quant=quant2.numpy(),
               t_diff=diff2.numpy(), t_ind=ind2.numpy(), t_perp=perp2.numpy())
    # (3) train mode: EMA buffers after 1 and 2 steps
    torch.manual_seed(14)
    q3 = QuantizedBottleneck(16, 32).train()
    e0 = q3.embed.clone()
    out["e_embed0"] = e0.numpy()
    import warnings
    for step in (1, 2):
        zt = torch.randn(2, 5, 7, 16)
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            quant3, diff3, ind3, perp3 = q3(zt)
        out[f"e_z{step}"] = zt.numpy()
        out[f"e_ind{step}"] = ind3.numpy()
        out[f"e_embed{step}"] = q3.embed.clone().numpy()
        out[f"e_cluster_size{step}"] = q3.cluster_size.clone().numpy()
        out[f"e_embed_avg{step}"] = q3.embed_avg.clone().numpy()
        out[f"e_diff{step}"] = diff3.numpy()
        out[f"e_perp{step}"] = perp3.numpy()
    # (4) train mode with index corruption (bottleneck.py:63-73); the offsets come from the CPU default generator
    torch.manual_seed(15)
    q4 = QuantizedBottleneck(16, 32, corruption_weights=[0.1, 0.8, 0.1]).train()
    out["c_embed0"] = q4.embed.clone().numpy()
    zc = torch.randn(2, 5, 7, 16)
    out["c_z"] = zc.numpy()
    torch.manual_seed(16)      # state of the default generator when forward draws the offsets
    with warnings.catch_warnings():
        warnings.simplefilter("ignore")
        quant4, diff4, ind4, perp4 = q4(zc)
    out.update(c_ind=ind4.numpy(), c_quant=quant4.numpy(), c_diff=diff4.numpy(), c_perp=perp4.numpy(),
               c_embed1=q4.embed.clone().numpy(), c_cluster_size1=q4.cluster_size.clone().numpy(),
               c_embed_avg1=q4.embed_avg.clone().numpy())
    _save("quantizer.npz", **out)


@torch.no_grad()
def quantizer_large_fixtures():
    """65 536 near-tie-free Gaussian vectors through the REFERENCE's QuantizedBottleneck (D=64, K=512, its own
    randn codebook): only the codebook and the int16 indices are stored; the vectors are regenerated from the seed by
    oracle.vqvae_oracle.near_tie_free_vectors (shared with the test)."""
    from interactive_spectrogram_inpainting.vqvae.bottleneck import QuantizedBottleneck
    sys.path.insert(0, str(OUT.parent.parent))
    from oracle import vqvae_oracle as O
    torch.manual_seed(17)
    q = QuantizedBottleneck(64, 512).eval()
    n, seed, scale, min_gap = 65536, 18, 1.0, 1e-5
    vec = O.near_tie_free_vectors(q.embed, n, seed, scale=scale, min_gap=min_gap)
    ind = torch.cat([q(vec[lo:lo + 8192])[2] for lo in range(0, n, 8192)])
    assert ind.shape == (n,) and int(ind.max()) < 512
    _save("quantizer_large.npz", embed=q.embed.numpy(), ind=ind.numpy().astype(np.int16), n=np.int64(n),
          seed=np.int64(seed), scale=np.float64(scale), min_gap=np.float64(min_gap))


@torch.no_grad()
def codemap_fixtures():
    from interactive_spectrogram_inpainting.priors.codemaps_helpers import (
        SimpleCodemapsHelper, ZigZagCodemapsHelper)
    import inspect
    out = {}
    print("ZigZag signature:", inspect.signature(ZigZagCodemapsHelper.__init__))
    for (F_, T_) in [(4, 4), (4, 3), (64, 64), (32, 128), (256, 32)]:
        cm = torch.arange(F_ * T_).reshape(1, F_, T_)
        h = SimpleCodemapsHelper(F_, T_)
        seq = h.to_sequence(cm)
        assert torch.equal(h.to_time_frequency_map(seq), cm)
        out[f"simple_{F_}x{T_}"] = seq.numpy()
    for (F_, T_, pf, pt) in [(4, 4, 2, 2), (64, 64, 2, 2), (32, 128, 2, 2), (256, 32, 2, 2),
                             (16, 8, 4, 2)]:
        cm = torch.arange(F_ * T_).reshape(1, F_, T_)
        h = ZigZagCodemapsHelper(F_, T_, pf, pt)
        seq = h.to_sequence(cm)
        assert torch.equal(h.to_time_frequency_map(seq), cm)
        out[f"zigzag_{F_}x{T_}_{pf}x{pt}"] = seq.numpy()
        # 4-D (embedding) input
        cm4 = torch.arange(2 * F_ * T_ * 3).reshape(2, F_, T_, 3)
        seq4 = h.to_sequence(cm4)
        assert torch.equal(h.to_time_frequency_map(seq4), cm4)
        if F_ * T_ <= 1024:
            out[f"zigzag4d_{F_}x{T_}_{pf}x{pt}"] = seq4.numpy()
            out[f"zigzag4d_logits_{F_}x{T_}_{pf}x{pt}"] = h.to_time_frequency_map(
                seq4, permute_output_as_logits=True).numpy()
    _save("codemaps.npz", **out)


@torch.no_grad()
def prior_wrapper_fixtures():
    """Everything of the prior that IS in the reference tree: token / positional /
    class embeddings, start symbols, sequence construction (to_sequences), causal
    mask, the constructor arguments it hands to the (absent) transformer layers,
    and the label-smoothing loss."""
    from interactive_spectrogram_inpainting.priors.transformer import (
        SelfAttentiveVQTransformer, UpsamplingVQTransformer)
    from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
    common = dict(n_class=32, channel=8, kernel_size=5, n_block=1, n_res_block=1, res_channel=8,
                  d_model=64, embeddings_dim=8, positional_embeddings_dim=8,
                  use_relative_transformer=True, predict_frequencies_first=True,
                  conditional_model=True, class_conditioning_prepend_to_dummy_input=True,
                  class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
                  class_conditioning_embedding_dim_per_modality={"instrument_family_str": 16, "pitch": 16},
                  conditional_model_nhead=4, conditional_model_num_encoder_layers=2,
                  conditional_model_num_decoder_layers=3)
    out = {}
    torch.manual_seed(31)
    top = SelfAttentiveVQTransformer(shape=[8, 4], condition_shape=[8, 4], self_conditional_model=True,
                                     add_mask_token_to_symbols=True, **common).eval()
    bottom = UpsamplingVQTransformer(shape=[16, 8], condition_shape=[8, 4], **common).eval()
    for name, m in (("top", top), ("bottom", bottom)):
        for k, v in m.state_dict().items():
            out[f"{name}::w::{k}"] = v.numpy()
        out[f"{name}::causal_mask"] = m.causal_mask.numpy()
        enc_layer = m.transformer.kwargs["custom_encoder"].kwargs["encoder_layer"].kwargs
        dec_layer = m.transformer.kwargs["custom_decoder"].kwargs["decoder_layer"].kwargs
        out[f"{name}::enc_layer_args"] = np.array([enc_layer["d_model"], enc_layer["nhead"],
                                                   enc_layer["num_channels"], enc_layer["num_events"]])
        out[f"{name}::dec_layer_args"] = np.array([
            dec_layer["d_model"], dec_layer["nhead"], dec_layer["num_channels_encoder"],
            dec_layer["num_events_encoder"], dec_layer["num_channels_decoder"], dec_layer["num_events_decoder"]])
        print(name, "enc", enc_layer, "dec", dec_layer)
    B = 2
    g = torch.Generator().manual_seed(32)
    cls = {"instrument_family_str": torch.randint(0, 11, (B, 1), generator=g),
           "pitch": torch.randint(0, 61, (B, 1), generator=g)}
    top_code = torch.randint(0, 32, (B, 8, 4), generator=g)
    mask = torch.rand(B, 8, 4, generator=g) < 0.4
    src, tgt = top.to_sequences(top_code, top_code, class_conditioning=cls, mask=mask)
    out.update({"top::code": top_code.numpy(), "top::mask": mask.numpy(), "top::src": src.numpy(),
                "top::tgt": tgt.numpy()})
    tidx = [1, 2, 3, 3]
    src2, tgt2 = top.to_sequences(top_code, top_code, class_conditioning=cls, mask=None,
                                  time_indexes_source=tidx, time_indexes_target=tidx)
    out.update({"top::tidx": np.array(tidx), "top::src_tidx": src2.numpy(), "top::tgt_tidx": tgt2.numpy()})
    bottom_code = torch.randint(0, 32, (B, 16, 8), generator=g)
    src3, tgt3 = bottom.to_sequences(bottom_code, top_code, class_conditioning=cls)
    out.update({"bottom::code": bottom_code.numpy(), "bottom::src": src3.numpy(), "bottom::tgt": tgt3.numpy()})
    for k, v in cls.items():
        out[f"cls::{k}"] = v.numpy()
    # label smoothing loss (utils/losses/prediction.py)
    pred = torch.randn(3, 32, 5, 7, generator=g)
    target = torch.randint(0, 32, (3, 5, 7), generator=g)
    out["ls::pred"], out["ls::target"] = pred.numpy(), target.numpy()
    out["ls::loss_0.1"] = LabelSmoothingLoss(32, 0.1, dim=1)(pred, target).numpy()
    out["ls::loss_0.0"] = LabelSmoothingLoss(32, 0.0, dim=1)(pred, target).numpy()
    _save("prior_wrapper.npz", **out)


@torch.no_grad()
def prior_wrapper_positional_fixtures():
    """positional_class_conditioning=True (priors/transformer.py:270-271,304-345,555-560,660-663): the class embeddings
    are appended to EVERY position (and to the start symbols) instead of being written into the start symbol only;
    effective embedding and start-symbol widths shrink by the class-conditioning width."""
    from interactive_spectrogram_inpainting.priors.transformer import SelfAttentiveVQTransformer, UpsamplingVQTransformer
    common = dict(n_class=32, channel=8, kernel_size=5, n_block=1, n_res_block=1, res_channel=8,
                  d_model=64, embeddings_dim=8, positional_embeddings_dim=8,
                  use_relative_transformer=True, predict_frequencies_first=True,
                  conditional_model=True, positional_class_conditioning=True,
                  class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
                  class_conditioning_embedding_dim_per_modality={"instrument_family_str": 16, "pitch": 16},
                  conditional_model_nhead=4, conditional_model_num_encoder_layers=2,
                  conditional_model_num_decoder_layers=3)
    out = {}
    torch.manual_seed(41)
    top = SelfAttentiveVQTransformer(shape=[8, 4], condition_shape=[8, 4], self_conditional_model=True,
                                     add_mask_token_to_symbols=True, **common).eval()
    bottom = UpsamplingVQTransformer(shape=[16, 8], condition_shape=[8, 4], **common).eval()
    for name, m in (("top", top), ("bottom", bottom)):
        for k, v in m.state_dict().items():
            out[f"{name}::w::{k}"] = v.numpy()
        out[f"{name}::effective_dim"] = np.array(m.embeddings_effective_dim)
    B = 2
    g = torch.Generator().manual_seed(42)
    cls = {"instrument_family_str": torch.randint(0, 11, (B, 1), generator=g),
           "pitch": torch.randint(0, 61, (B, 1), generator=g)}
    top_code = torch.randint(0, 32, (B, 8, 4), generator=g)
    mask = torch.rand(B, 8, 4, generator=g) < 0.4
    src, tgt = top.to_sequences(top_code, top_code, class_conditioning=cls, mask=mask)
    out.update({"top::code": top_code.numpy(), "top::mask": mask.numpy(), "top::src": src.numpy(), "top::tgt": tgt.numpy()})
    bottom_code = torch.randint(0, 32, (B, 16, 8), generator=g)
    src3, tgt3 = bottom.to_sequences(bottom_code, top_code, class_conditioning=cls)
    out.update({"bottom::code": bottom_code.numpy(), "bottom::src": src3.numpy(), "bottom::tgt": tgt3.numpy()})
    for k, v in cls.items():
        out[f"cls::{k}"] = v.numpy()
    _save("prior_wrapper_positional.npz", **out)


@torch.no_grad()
def filtering_fixtures():
    """top_k_top_p_filtering (sample.py:36-65), imported with its heavy script
    dependencies stubbed."""
    import types as _t
    for name in ("soundfile", "torchaudio", "torchvision", "torchvision.utils"):
        if name not in sys.modules:
            sys.modules[name] = _t.ModuleType(name)
    sys.modules["torchvision.utils"].save_image = lambda *a, **k: None
    gs = _t.ModuleType("GANsynth_pytorch.spectrograms_helper")
    gs.SpectrogramsHelper = gs.MelSpectrogramsHelper = object
    sys.modules["GANsynth_pytorch.spectrograms_helper"] = gs
    gl = sys.modules["GANsynth_pytorch.loader"]
    for n in ("WavToSpectrogramDataLoader", "MaskedPhaseWavToSpectrogramDataLoader"):
        setattr(gl, n, object)
    try:
        import lmdb  # noqa: F401
    except ImportError:
        sys.modules["lmdb"] = _t.ModuleType("lmdb")
    import sample as ref_sample
    g = torch.Generator().manual_seed(41)
    logits = torch.randn(2, 5, 32, generator=g) * 2
    out = {"logits": logits.numpy()}
    for (k, p) in [(0, 0.0), (5, 0.0), (0, 0.8), (8, 0.6), (40, 0.0), (0, 0.05)]:
        out[f"k{k}_p{p}"] = ref_sample.top_k_top_p_filtering(logits.clone(), top_k=k, top_p=p).numpy()
    _save("filtering.npz", **out)


def scheduler_fixtures():
    """LR (and beta1) traces of the schedules the training scripts use
    (utils/training/scheduler.py: CycleScheduler, get_cosine_schedule_with_warmup)."""
    from interactive_spectrogram_inpainting.utils.training.scheduler import (
        CycleScheduler, get_cosine_schedule_with_warmup)
    out = {}
    for tag, n_iter, kw in (("a", 10, {}), ("b", 37, dict(divider=10, warmup_proportion=0.45)),
                            ("c", 8, dict(momentum=None, phase=("cos", "linear")))):
        w = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.Adam([w], lr=1.0)
        # The reference class calls `_LRScheduler.__init__` FIRST, which itself calls `self.step()`
        # before `lr_phase` exists (AttributeError on every torch release that has that base class);
        # the schedule it describes is recorded here by skipping the base constructor.
        base_init = torch.optim.lr_scheduler._LRScheduler.__init__
        torch.optim.lr_scheduler._LRScheduler.__init__ = lambda self, optimizer, *a, **k: setattr(self, "optimizer", optimizer)
        try:
            sch = CycleScheduler(opt, 3e-4, n_iter=n_iter, **kw)
        finally:
            torch.optim.lr_scheduler._LRScheduler.__init__ = base_init
        lrs, moms = [], []
        for _ in range(2 * n_iter + 3):   # runs past the end of the cycle: it restarts
            lr, mom = sch.step()
            lrs.append(lr)
            moms.append(opt.param_groups[0]["betas"][0])
        out[f"cycle_{tag}::lr"], out[f"cycle_{tag}::beta1"] = np.array(lrs), np.array(moms)
        out[f"cycle_{tag}::args"] = np.array([n_iter, kw.get("divider", 25), kw.get("warmup_proportion", 0.3)])
    for tag, warm, total, cycles in (("a", 5, 40, 0.5), ("b", 0, 12, 0.5), ("c", 3, 20, 1.5)):
        w = torch.nn.Parameter(torch.zeros(1))
        opt = torch.optim.SGD([w], lr=0.01)
        sch = get_cosine_schedule_with_warmup(opt, warm, total, num_cycles=cycles)
        lrs = [opt.param_groups[0]["lr"]]
        for _ in range(total + 4):
            opt.step()
            sch.step()
            lrs.append(opt.param_groups[0]["lr"])
        out[f"cosine_{tag}::lr"] = np.array(lrs)
        out[f"cosine_{tag}::args"] = np.array([warm, total, cycles])
    _save("schedulers.npz", **out)


def time_indexes_fixtures():
    """`make_time_indexes` (flask_server.py:670-682) is a pure function living in a script whose imports
    (flask_cors, GANsynth_pytorch, ...) are absent: its definition is extracted from the reference source
    with `ast` and executed here, and only its outputs are stored."""
    import ast
    from typing import List
    src = (REF / "flask_server.py").read_text()
    fn = next(n for n in ast.parse(src).body if isinstance(n, ast.FunctionDef) and n.name == "make_time_indexes")
    ns = {"List": List}
    exec(compile(ast.Module(body=[fn], type_ignores=[]), "flask_server.py", "exec"), ns)
    out, cases = {}, []
    for codemap_duration, transformer_duration in ((32, 32), (64, 32), (96, 32), (40, 32), (128, 64), (33, 32),
                                                   (12, 4), (16, 8)):
        for start in sorted({0, 1, (codemap_duration - transformer_duration) // 2,
                             codemap_duration - transformer_duration}):
            if start < 0:
                continue
            cases.append((start, codemap_duration, transformer_duration))
            out[f"ti::{start}_{codemap_duration}_{transformer_duration}"] = np.array(
                ns["make_time_indexes"](start, codemap_duration, transformer_duration), dtype=np.int64)
    out["cases"] = np.array(cases, dtype=np.int64)
    _save("time_indexes.npz", **out)


def spectral_loss_fixtures():
    """Values and d/d(audio_pred) of the reference's MultiscaleSpectralLoss (utils/losses/spectral.py) with the
    DDSP and the Jukebox parameter sets and with L2Loss.  The reference calls torch.stft without
    `return_complex`, which current torch rejects for real input: the call is given the legacy real-pair
    output (return_complex=False, still available) -- the format `magnitude()` (`t.norm(2, dim=-1)`) expects."""
    import types as _t
    for name in ("GANsynth_pytorch", "GANsynth_pytorch.spectrograms_helper"):
        if name not in sys.modules:
            sys.modules[name] = _t.ModuleType(name)
    if not hasattr(sys.modules["GANsynth_pytorch.spectrograms_helper"], "SpectrogramsHelper"):
        sys.modules["GANsynth_pytorch.spectrograms_helper"].SpectrogramsHelper = object
    from interactive_spectrogram_inpainting.utils.losses import spectral as R
    import warnings
    legacy = torch.stft

    def stft_legacy(*a, **k):
        k.setdefault("return_complex", False)
        return legacy(*a, **k)

    out = {}
    g = torch.Generator().manual_seed(41)
    pred = torch.randn(2, 6000, generator=g) * 0.3
    target = pred * 0.7 + torch.randn(2, 6000, generator=g) * 0.2
    out["pred"], out["target"] = pred.numpy(), target.numpy()
    cases = {"ddsp": R.DDSPMultiscaleSpectralLoss_kwargs, "jukebox": R.JukeboxMultiscaleSpectralLoss_kwargs,
             "l2": dict(n_ffts=[256, 512], window_lengths=[200, 512], overlap_ratio=0.75, loss=R.L2Loss(),
                        lin_loss_alpha=0.5, log_loss_alpha=2.0)}
    torch.stft = stft_legacy
    try:
        with warnings.catch_warnings():
            warnings.simplefilter("ignore")
            for name, kw in cases.items():
                m = R.MultiscaleSpectralLoss(**kw)
                p = pred.clone().requires_grad_(True)
                loss = m(p, target)
                loss.backward()
                out[f"{name}::loss"] = loss.detach().numpy()
                out[f"{name}::grad"] = p.grad.numpy()
    finally:
        torch.stft = legacy
    _save("spectral_loss.npz", **out)


@torch.no_grad()
def unquantized_fixtures():
    """VQVAE(disable_quantization=True): UnquantizedBottleneck at both levels (bottleneck.py:107-119, selected at
    vqvae.py:152-160)."""
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    kw = dict(in_channel=2, num_hidden_channels=32, n_res_block=1, num_residual_channels=8, embed_dim=16,
              num_embeddings=64, resolution_factors={"bottom": 4, "top": 2})
    torch.manual_seed(25)
    model = VQVAE(disable_quantization=True, **kw).eval()
    x = torch.randn(2, 2, 32, 40)
    q_t, q_b, diff, id_t, id_b, perp_t, perp_b = model.encode(x.clone())
    dec, diff2, perp_t2, perp_b2, id_t2, id_b2 = model(x.clone())
    assert id_t is None and id_b is None and id_t2 is None
    arrays = {"w::" + k: v for k, v in _np(model.state_dict()).items()}
    arrays.update(x=x.numpy(), quant_t=q_t.contiguous().numpy(), quant_b=q_b.contiguous().numpy(), diff=diff.numpy(),
                  perplexity_t=perp_t.numpy(), perplexity_b=perp_b.numpy(), dec=dec.numpy(),
                  dec_from_quant=model.decode(q_t, q_b).numpy())
    _save("vqvae_unquantized.npz", **arrays)


def train_trajectory_fixtures():
    """BASELINE config 1: the reference's own training-step semantics (train_vqvae.py:168-192:
    model.zero_grad(); out, latent_loss, perplexity_t, perplexity_b, *_ = model(img); loss = MSE(out, img) +
    0.25 * latent_loss.mean(); loss.backward(); Adam(lr 3e-4).step()) run on the imported reference VQ-VAE for two
    steps of batch 8: per-step losses / perplexities / code indices, and the codebooks and a few parameters after the
    second step.  Two cases: the reduced configuration (weights stored) and the DEFAULT constructor at the NSynth
    shape [8,2,128,512] (initial weights = those of vqvae_default_tiny.npz, inputs regenerated from the seed)."""
    import warnings
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    out = {}

    def run(tag, model, batches):
        model.train()
        opt = torch.optim.Adam(model.parameters(), lr=3e-4)
        crit = nn.MSELoss()
        for step, img in enumerate(batches):
            model.zero_grad()
            with warnings.catch_warnings():
                warnings.simplefilter("ignore")
                o, latent_loss, perp_t, perp_b, id_t, id_b = model(img.clone())
            recon = crit(o, img)
            latent = latent_loss.mean()
            loss = recon + 0.25 * latent
            loss.backward()
            opt.step()
            out[f"{tag}::recon{step}"] = recon.detach().numpy()
            out[f"{tag}::latent{step}"] = latent.detach().numpy()
            out[f"{tag}::loss{step}"] = loss.detach().numpy()
            out[f"{tag}::perp_t{step}"] = perp_t.detach().numpy()
            out[f"{tag}::perp_b{step}"] = perp_b.detach().numpy()
            out[f"{tag}::id_t{step}"] = id_t.numpy().astype(np.int16)
            out[f"{tag}::id_b{step}"] = id_b.numpy().astype(np.int16)
            print(tag, step, "loss", loss.item(), "recon", recon.item(), "latent", latent.item(),
                  "perplexity", perp_t.item(), perp_b.item())
        sd = model.state_dict()
        for k in ("quantize_t.embed", "quantize_b.embed", "quantize_t.cluster_size", "quantize_b.cluster_size",
                  "quantize_t.embed_avg", "quantize_b.embed_avg", "quantize_conv_t.weight", "quantize_conv_t.bias",
                  "enc_b.blocks.0.weight", "dec.blocks.0.bias", "enc_t.blocks.3.conv.3.weight"):
            out[f"{tag}::after::{k}"] = sd[k].detach().numpy().copy()

    # (1) reduced configuration, weights stored with the fixture
    torch.manual_seed(41)
    kw = dict(in_channel=2, num_hidden_channels=32, n_res_block=2, num_residual_channels=8, embed_dim=16,
              num_embeddings=64, resolution_factors={"bottom": 4, "top": 2})
    small = VQVAE(**kw).eval()
    _calibrate_codebooks(small, torch.randn(4, 2, 32, 48), 141)
    for k, v in _np(small.state_dict()).items():
        out["small::w::" + k] = v.copy()       # .numpy() shares storage with the parameters the steps below update
    xs = [torch.randn(8, 2, 32, 48) for _ in range(2)]
    for i, xb in enumerate(xs):
        out[f"small::x{i}"] = xb.numpy()
    run("small", small, xs)

    # (2) default constructor at the NSynth shape; initial weights = vqvae_default_tiny.npz
    z = np.load(OUT / "vqvae_default_tiny.npz")
    full = VQVAE(in_channel=2, resolution_factors={"bottom": 4, "top": 2})
    full.load_state_dict({k[3:]: torch.from_numpy(z[k]) for k in z.files if k.startswith("w::")})
    g = torch.Generator().manual_seed(4242)
    xs = [torch.randn(8, 2, 128, 512, generator=g) for _ in range(2)]
    for i, xb in enumerate(xs):
        xb[:, 1].tanh_()                       # channel 1 ~ mel-IF in [-1, 1] (SURVEY 8d synthetic input)
        out[f"full::x{i}_head"] = xb.reshape(-1)[:64].numpy().copy()      # proves the regenerated inputs are the same
        out[f"full::x{i}_sum"] = np.float64(xb.double().sum().item())
    out["full::x_seed"] = np.int64(4242)
    run("full", full, xs)
    _save("train_trajectory.npz", **out)


def vqvae_f16_fixtures():
    """The deepest down-sampling the reference's encoder / decoder offer (resolution factor 16 at the bottom level:
    vqvae/encoder_decoder.py:57-75 and :158-176), reduced widths; top level a further factor 2."""
    vqvae_fixture("vqvae_f16_f2.npz",
                  dict(in_channel=2, num_hidden_channels=16, n_res_block=1,
                       num_residual_channels=8, embed_dim=8, num_embeddings=32,
                       resolution_factors={"bottom": 16, "top": 2}),
                  (1, 2, 64, 96), seed=25)


def main():
    os.environ["PYTHONDONTWRITEBYTECODE"] = "1"
    _install_stubs()
    sys.path.insert(0, str(REF))
    torch.set_num_threads(4)
    if len(sys.argv) > 1:   # regenerate only the named groups, e.g. `make_golden.py schedulers`
        for name in sys.argv[1:]:
            globals()[name + "_fixtures"]()
        return
    # reduced config
    vqvae_fixture("vqvae_small.npz",
                  dict(in_channel=2, num_hidden_channels=32, n_res_block=2,
                       num_residual_channels=8, embed_dim=16, num_embeddings=64,
                       resolution_factors={"bottom": 4, "top": 2}),
                  (2, 2, 32, 48), seed=21)
    # default config at tiny spatial size
    vqvae_fixture("vqvae_default_tiny.npz",
                  dict(in_channel=2, resolution_factors={"bottom": 4, "top": 2}),
                  (2, 2, 32, 64), seed=22)
    # other resolution factors (8 bottom / 4 top) on a reduced config
    vqvae_fixture("vqvae_f8_f4.npz",
                  dict(in_channel=2, num_hidden_channels=16, n_res_block=1,
                       num_residual_channels=8, embed_dim=8, num_embeddings=32,
                       resolution_factors={"bottom": 8, "top": 4}),
                  (1, 2, 64, 64), seed=23)
    # grouped convolutions (groups=2 in every down / up-sampling conv and the encoders' 3x3)
    vqvae_fixture("vqvae_groups2.npz",
                  dict(in_channel=2, num_hidden_channels=32, n_res_block=1,
                       num_residual_channels=8, embed_dim=16, num_embeddings=64, groups=2,
                       resolution_factors={"bottom": 4, "top": 2}),
                  (2, 2, 32, 32), seed=24)
    vqvae_f16_fixtures()
    resblock_fixture()
    layer_fixtures()
    quantizer_fixtures()
    quantizer_large_fixtures()
    codemap_fixtures()
    prior_wrapper_fixtures()
    prior_wrapper_positional_fixtures()
    filtering_fixtures()
    scheduler_fixtures()
    time_indexes_fixtures()
    spectral_loss_fixtures()
    unquantized_fixtures()
    train_trajectory_fixtures()


if __name__ == "__main__":
    main()
