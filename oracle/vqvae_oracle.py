"""CPU oracle for the VQ-VAE-2 hot path  --  TEST INFRASTRUCTURE, NOT PRODUCT.

This file restates, in plain functional PyTorch-CPU fp32, the arithmetic of
the reference's `VQVAE.forward` path.  Only `tests/`, `__graft_entry__.smoke()`
and `bench.py`'s `cpu_baseline` leg may import it; the product package
(`interactive-spectrogram-inpainting_amd/`) never does.

Pinned: `tests/test_oracle_golden.py` checks every function here against the
fixtures in `tests/golden/` which were produced by importing the reference
itself in the build container (`oracle/make_golden.py`).

Reference citations (paths relative to /root/reference):
  res_block      interactive_spectrogram_inpainting/vqvae/encoder_decoder.py:18-35
  encoder        .../vqvae/encoder_decoder.py:38-126
  decoder        .../vqvae/encoder_decoder.py:129-227
  quantize       .../vqvae/bottleneck.py:53-104
  encode/decode  .../vqvae/vqvae.py:245-302

All tensors are NCHW fp32 like the reference; state dict keys are the
reference's own (`enc_b.blocks.0.weight`, `quantize_t.embed`, ...).
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional, Tuple

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
StateDict = Mapping[str, Tensor]

# number of stride-2 stages for each supported resolution factor
_STAGES = {2: 1, 4: 2, 8: 3, 16: 4}


def _down_channels(in_channel: int, channel: int, factor: int):
    """(cin, cout) of every strided conv of an encoder
    (encoder_decoder.py:53-113)."""
    if factor == 16:
        chans = [in_channel, channel // 4, channel // 2, 3 * channel // 4, channel]
    elif factor == 8:
        chans = [in_channel, channel // 2, channel // 2, channel]
    elif factor == 4:
        chans = [in_channel, channel // 2, channel]
    elif factor == 2:
        chans = [in_channel, channel // 2]
    else:
        raise ValueError(f"Unexpected resolution factor {factor}")
    return list(zip(chans[:-1], chans[1:]))


def _up_channels(channel: int, out_channel: int, factor: int):
    """(cin, cout) of every transposed conv of a decoder
    (encoder_decoder.py:153-216)."""
    if factor == 16:
        chans = [channel, 3 * channel // 4, channel // 2, channel // 4, out_channel]
    elif factor == 8:
        chans = [channel, channel // 2, channel // 2, out_channel]
    elif factor == 4:
        chans = [channel, channel // 2, out_channel]
    elif factor == 2:
        chans = [channel, out_channel]
    else:
        raise ValueError(f"Unexpected resolution factor {factor}")
    return list(zip(chans[:-1], chans[1:]))


def _conv(x: Tensor, w: Tensor, b: Tensor, **kw) -> Tensor:
    """Conv2d whose `groups` (encoder_decoder.py:58-112) is read off the weight shape [Cout, Cin/groups, k, k]."""
    return F.conv2d(x, w, b, groups=x.shape[1] // w.shape[1], **kw)


def _conv_t(x: Tensor, w: Tensor, b: Tensor, **kw) -> Tensor:
    """ConvTranspose2d, `groups` (encoder_decoder.py:159-215) from the weight shape [Cin, Cout/groups, k, k]."""
    return F.conv_transpose2d(x, w, b, groups=b.shape[0] // w.shape[1], **kw)


def res_block(x: Tensor, sd: StateDict, prefix: str) -> Tensor:
    """RosinalityResBlock (encoder_decoder.py:18-35).

    The first ReLU of the block is in-place, so it overwrites the tensor the
    residual connection later reads: the block returns
    relu(x) + conv1x1(relu(conv3x3(relu(x)))), NOT x + ... .
    """
    r = F.relu(x)
    h = F.conv2d(r, sd[prefix + "conv.1.weight"], sd[prefix + "conv.1.bias"], padding=1)
    h = F.relu(h)
    h = F.conv2d(h, sd[prefix + "conv.3.weight"], sd[prefix + "conv.3.bias"])
    return h + r


def encoder(x: Tensor, sd: StateDict, prefix: str, factor: int, n_res_block: int,
            use_local_kernels: bool = False) -> Tensor:
    """RosinalityEncoder.forward (encoder_decoder.py:38-126)."""
    k = 2 if use_local_kernels else 4
    idx = 0
    n_stages = _STAGES[factor]
    for s in range(n_stages):
        x = _conv(x, sd[f"{prefix}blocks.{idx}.weight"], sd[f"{prefix}blocks.{idx}.bias"],
                  stride=2, padding=1)
        idx += 1
        # every strided conv is followed by a ReLU (encoder_decoder.py:53-113)
        x = F.relu(x)
        idx += 1
    assert k == sd[f"{prefix}blocks.0.weight"].shape[-1]
    x = _conv(x, sd[f"{prefix}blocks.{idx}.weight"], sd[f"{prefix}blocks.{idx}.bias"],
              padding=1)
    idx += 1
    for _ in range(n_res_block):
        x = res_block(x, sd, f"{prefix}blocks.{idx}.")
        idx += 1
    return F.relu(x)


def decoder(x: Tensor, sd: StateDict, prefix: str, factor: int, n_res_block: int) -> Tensor:
    """RosinalityDecoder.forward (encoder_decoder.py:129-227)."""
    idx = 0
    x = F.conv2d(x, sd[f"{prefix}blocks.{idx}.weight"], sd[f"{prefix}blocks.{idx}.bias"],
                 padding=1)
    idx += 1
    for _ in range(n_res_block):
        x = res_block(x, sd, f"{prefix}blocks.{idx}.")
        idx += 1
    x = F.relu(x)
    idx += 1
    n_stages = _STAGES[factor]
    for s in range(n_stages):
        x = _conv_t(x, sd[f"{prefix}blocks.{idx}.weight"],
                    sd[f"{prefix}blocks.{idx}.bias"], stride=2, padding=1)
        idx += 1
        if s != n_stages - 1:
            x = F.relu(x)
            idx += 1
    return x


def quantize(z: Tensor, embed: Tensor) -> Tuple[Tensor, Tensor, Tensor, Tensor]:
    """QuantizedBottleneck.forward in eval mode (bottleneck.py:53-101).

    z      [..., D]  (channels-last)
    embed  [D, K]    (column = code)
    returns (straight-through quantize [..., D], diff [], idx int64 [...],
             perplexity [])
    """
    dim, n_embed = embed.shape
    flat = z.reshape(-1, dim)
    dist = (flat.pow(2).sum(1, keepdim=True)
            - 2 * flat @ embed
            + embed.pow(2).sum(0, keepdim=True))
    _, ind = (-dist).max(1)
    onehot_mean = torch.bincount(ind, minlength=n_embed).to(flat.dtype) / flat.shape[0]
    ind = ind.view(*z.shape[:-1])
    q = F.embedding(ind, embed.t())
    diff = (q - z).pow(2).mean()
    q_st = z + (q - z)
    perplexity = torch.exp(-torch.sum(onehot_mean * torch.log(onehot_mean.clamp(min=1e-7))))
    return q_st, diff, ind, perplexity


def ema_update(flat: Tensor, ind: Tensor, embed: Tensor, cluster_size: Tensor,
               embed_avg: Tensor, decay: float = 0.99, eps: float = 1e-5
               ) -> Tuple[Tensor, Tensor, Tensor]:
    """Train-mode EMA codebook update (bottleneck.py:79-92); returns the new
    (embed, cluster_size, embed_avg) without mutating the arguments."""
    n_embed = embed.shape[1]
    onehot = F.one_hot(ind.reshape(-1), n_embed).to(flat.dtype)
    cs = cluster_size * decay + (1 - decay) * onehot.sum(0)
    ea = embed_avg * decay + (1 - decay) * (flat.t() @ onehot)
    n = cs.sum()
    cs_norm = (cs + eps) / (n + n_embed * eps) * n
    return ea / cs_norm.unsqueeze(0), cs, ea


def quantize_train(z: Tensor, embed: Tensor, cluster_size: Tensor, embed_avg: Tensor, decay: float = 0.99,
                   eps: float = 1e-5, corruption_weights=None):
    """QuantizedBottleneck.forward in train mode with autograd semantics
    (bottleneck.py:53-101): search with the current codebook, optional index corruption
    (`:63-73`: offsets multinomial(weights) - 1 in {-1, 0, +1}, drawn from the CPU default
    generator, added modulo K), EMA update of the buffers, commitment `diff`, straight-through output.
    Returns (q_st, diff, ind, perplexity, (embed', cluster_size', embed_avg'))."""
    dim, n_embed = embed.shape
    with torch.no_grad():
        flat = z.detach().reshape(-1, dim)
        dist = flat.pow(2).sum(1, keepdim=True) - 2 * flat @ embed + embed.pow(2).sum(0, keepdim=True)
        _, ind = (-dist).max(1)
        if corruption_weights is not None:
            offsets = torch.multinomial(torch.Tensor(corruption_weights), ind.numel(), replacement=True) - 1
            ind = (ind + offsets.reshape(ind.shape)) % n_embed
        new = ema_update(flat, ind, embed, cluster_size, embed_avg, decay, eps)
        onehot_mean = torch.bincount(ind, minlength=n_embed).to(flat.dtype) / flat.shape[0]
        perplexity = torch.exp(-torch.sum(onehot_mean * torch.log(onehot_mean.clamp(min=1e-7))))
    ind = ind.view(*z.shape[:-1])
    q = F.embedding(ind, embed.t())
    diff = (q.detach() - z).pow(2).mean()
    q_st = z + (q - z).detach()
    return q_st, diff, ind, perplexity, new


def forward_train(x: Tensor, sd: StateDict, cfg: "Config"):
    """Train-mode VQVAE.forward: (dec, diff, id_t, id_b, new_buffers) with autograd through sd."""
    fb, ft = cfg.resolution_factors["bottom"], cfg.resolution_factors["top"]
    enc_b = encoder(x, sd, "enc_b.", fb, cfg.n_res_block)
    enc_t = encoder(enc_b, sd, "enc_t.", ft, cfg.n_res_block)
    z_t = F.conv2d(enc_t, sd["quantize_conv_t.weight"], sd["quantize_conv_t.bias"]).permute(0, 2, 3, 1)
    if cfg.disable_quantization:      # identity bottlenecks: nothing to update, gradients pass straight through
        dec_t = decoder(z_t.permute(0, 3, 1, 2), sd, "dec_t.", ft, cfg.n_res_block)
        z_b = F.conv2d(torch.cat([dec_t, enc_b], 1), sd["quantize_conv_b.weight"], sd["quantize_conv_b.bias"])
        return decode(z_t.permute(0, 3, 1, 2), z_b, sd, cfg), torch.zeros(1, 1), None, None, (None, None)
    q_t, diff_t, id_t, _, new_t = quantize_train(z_t, sd["quantize_t.embed"], sd["quantize_t.cluster_size"],
                                                 sd["quantize_t.embed_avg"],
                                                 corruption_weights=cfg.corruption_weights["top"])
    q_t = q_t.permute(0, 3, 1, 2)
    dec_t = decoder(q_t, sd, "dec_t.", ft, cfg.n_res_block)
    z_b = F.conv2d(torch.cat([dec_t, enc_b], 1), sd["quantize_conv_b.weight"],
                   sd["quantize_conv_b.bias"]).permute(0, 2, 3, 1)
    q_b, diff_b, id_b, _, new_b = quantize_train(z_b, sd["quantize_b.embed"], sd["quantize_b.cluster_size"],
                                                 sd["quantize_b.embed_avg"],
                                                 corruption_weights=cfg.corruption_weights["bottom"])
    q_b = q_b.permute(0, 3, 1, 2)
    dec = decode(q_t, q_b, sd, cfg)
    return dec, diff_t.unsqueeze(0) + diff_b.unsqueeze(0), id_t, id_b, (new_t, new_b)


def unquantized(z: Tensor) -> Tuple[Tensor, Tensor, None, Tensor]:
    """UnquantizedBottleneck.forward (bottleneck.py:107-119): identity, diff zeros(1), no indices, perplexity [inf]."""
    return z, torch.zeros((1,), dtype=z.dtype), None, torch.as_tensor([float("inf")])


def embed_code(ind: Tensor, embed: Tensor) -> Tensor:
    """QuantizedBottleneck.embed_code (bottleneck.py:103-104)."""
    return F.embedding(ind, embed.t())


class Config:
    """Constructor arguments that shape the arithmetic (vqvae.py:65-93)."""

    def __init__(self, in_channel=2, num_hidden_channels=128, n_res_block=2,
                 num_residual_channels=32, embed_dim=64, num_embeddings=512,
                 resolution_factors=None, adapt_quantized_durations=True, groups=1, corruption_weights=None,
                 disable_quantization=False):
        self.disable_quantization = disable_quantization  # vqvae.py:152-160: UnquantizedBottleneck at both levels
        self.groups = groups                              # vqvae.py:76: down / up-sampling convs and the encoders' 3x3
        self.corruption_weights = dict(corruption_weights or {"top": None, "bottom": None})   # vqvae.py:84-85
        self.in_channel = in_channel
        self.num_hidden_channels = num_hidden_channels
        self.n_res_block = n_res_block
        self.num_residual_channels = num_residual_channels
        self.embed_dim = embed_dim
        self.num_embeddings = num_embeddings
        self.resolution_factors = dict(resolution_factors or {"bottom": 4, "top": 2})
        self.adapt_quantized_durations = adapt_quantized_durations


def encode(x: Tensor, sd: StateDict, cfg: Config):
    """VQVAE.encode (vqvae.py:251-278) without the GANSynth normaliser."""
    fb, ft = cfg.resolution_factors["bottom"], cfg.resolution_factors["top"]
    enc_b = encoder(x, sd, "enc_b.", fb, cfg.n_res_block)
    enc_t = encoder(enc_b, sd, "enc_t.", ft, cfg.n_res_block)

    z_t = F.conv2d(enc_t, sd["quantize_conv_t.weight"], sd["quantize_conv_t.bias"]).permute(0, 2, 3, 1)
    q_t, diff_t, id_t, perp_t = unquantized(z_t) if cfg.disable_quantization else quantize(z_t, sd["quantize_t.embed"])
    q_t = q_t.permute(0, 3, 1, 2)

    dec_t = decoder(q_t, sd, "dec_t.", ft, cfg.n_res_block)
    if cfg.adapt_quantized_durations:
        w = min(dec_t.shape[-1], enc_b.shape[-1])
        dec_t = dec_t[..., :w]
        enc_b = enc_b[..., :w]
    cat = torch.cat([dec_t, enc_b], 1)
    z_b = F.conv2d(cat, sd["quantize_conv_b.weight"], sd["quantize_conv_b.bias"]).permute(0, 2, 3, 1)
    q_b, diff_b, id_b, perp_b = unquantized(z_b) if cfg.disable_quantization else quantize(z_b, sd["quantize_b.embed"])
    q_b = q_b.permute(0, 3, 1, 2)
    diff = diff_t.unsqueeze(0) + diff_b.unsqueeze(0)
    return q_t, q_b, diff, id_t, id_b, perp_t, perp_b


def decode(q_t: Tensor, q_b: Tensor, sd: StateDict, cfg: Config) -> Tensor:
    """VQVAE.decode (vqvae.py:280-286), post_process being the identity when
    no normaliser statistics / min magnitude are configured."""
    up = q_t
    n_up = int(math.log2(cfg.resolution_factors["top"]))
    for i in range(n_up):
        up = F.conv_transpose2d(up, sd[f"upsample_top_to_bottom.{i}.weight"],
                                sd[f"upsample_top_to_bottom.{i}.bias"], stride=2, padding=1)
    quant = torch.cat([up, q_b], 1)
    return decoder(quant, sd, "dec.", cfg.resolution_factors["bottom"], cfg.n_res_block)


def decode_code(id_t: Tensor, id_b: Tensor, sd: StateDict, cfg: Config) -> Tensor:
    """VQVAE.decode_code (vqvae.py:288-295)."""
    q_t = embed_code(id_t, sd["quantize_t.embed"]).permute(0, 3, 1, 2)
    q_b = embed_code(id_b, sd["quantize_b.embed"]).permute(0, 3, 1, 2)
    return decode(q_t, q_b, sd, cfg)


def forward(x: Tensor, sd: StateDict, cfg: Config):
    """VQVAE.forward (vqvae.py:245-249): (dec, diff, perp_t, perp_b, id_t, id_b)."""
    q_t, q_b, diff, id_t, id_b, perp_t, perp_b = encode(x, sd, cfg)
    dec = decode(q_t, q_b, sd, cfg)
    return dec, diff, perp_t, perp_b, id_t, id_b


def init_state_dict(cfg: Config, seed: int = 1) -> Dict[str, Tensor]:
    """Random fp32 weights with the reference's key names and shapes and the
    default torch init distributions (uniform(-1/sqrt(fan_in), ..)); used for
    synthetic benchmarks only.  Not bit-equal to nn.Module's init stream."""
    g = torch.Generator().manual_seed(seed)
    sd: Dict[str, Tensor] = {}

    def conv(name, cout, cin, k, groups=1):
        bound = 1.0 / math.sqrt(cin // groups * k * k)
        sd[name + ".weight"] = (torch.rand(cout, cin // groups, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    def convT(name, cin, cout, k, groups=1):
        bound = 1.0 / math.sqrt(cout // groups * k * k)
        sd[name + ".weight"] = (torch.rand(cin, cout // groups, k, k, generator=g) * 2 - 1) * bound
        sd[name + ".bias"] = (torch.rand(cout, generator=g) * 2 - 1) * bound

    C, R, D = cfg.num_hidden_channels, cfg.num_residual_channels, cfg.embed_dim

    def enc(prefix, cin, factor):
        idx = 0
        for (a, b) in _down_channels(cin, C, factor):
            conv(f"{prefix}blocks.{idx}", b, a, 4, cfg.groups)
            idx += 2
            last = b
        conv(f"{prefix}blocks.{idx}", C, last, 3, cfg.groups)
        idx += 1
        for _ in range(cfg.n_res_block):
            conv(f"{prefix}blocks.{idx}.conv.1", R, C, 3)
            conv(f"{prefix}blocks.{idx}.conv.3", C, R, 1)
            idx += 1

    def dec(prefix, cin, cout, factor):
        idx = 0
        conv(f"{prefix}blocks.{idx}", C, cin, 3)
        idx += 1
        for _ in range(cfg.n_res_block):
            conv(f"{prefix}blocks.{idx}.conv.1", R, C, 3)
            conv(f"{prefix}blocks.{idx}.conv.3", C, R, 1)
            idx += 1
        idx += 1  # ReLU
        for (a, b) in _up_channels(C, cout, factor):
            convT(f"{prefix}blocks.{idx}", a, b, 4, cfg.groups)
            idx += 2

    fb, ft = cfg.resolution_factors["bottom"], cfg.resolution_factors["top"]
    enc("enc_b.", cfg.in_channel, fb)
    enc("enc_t.", C, ft)
    conv("quantize_conv_t", D, C, 1)
    dec("dec_t.", D, D, ft)
    conv("quantize_conv_b", D, D + C, 1)
    for i in range(int(math.log2(ft))):
        convT(f"upsample_top_to_bottom.{i}", D, D, 4)
    dec("dec.", D + D, cfg.in_channel, fb)
    for lvl in ("t", "b"):
        e = torch.randn(D, cfg.num_embeddings, generator=g)
        sd[f"quantize_{lvl}.embed"] = e
        sd[f"quantize_{lvl}.cluster_size"] = torch.zeros(cfg.num_embeddings)
        sd[f"quantize_{lvl}.embed_avg"] = e.clone()
    return sd


@torch.no_grad()
def pick_distinct_rows(flat: Tensor, K: int, g: torch.Generator, min_rel: float = 0.1) -> Tensor:
    """Row indices of K WELL-SEPARATED vectors of `flat` [N, D]: rows are visited in a random order and one is
    accepted when it lies farther than `min_rel * |row|` from every accepted row; if the pool runs out, the
    remainder is filled farthest-point first.  (Drawing K rows WITH replacement from a few thousand highly
    correlated encoder outputs gave exact duplicates and code pairs 1e-3 apart: every vector assigned to such a
    pair is a coin toss of the fp32 distance formula, which made "bit-exact indices" untestable at scale.)"""
    N = flat.shape[0]
    assert N >= K, (N, K)
    order = torch.randperm(N, generator=g)
    cand = flat[order].double()
    norm = cand.norm(dim=1)
    mind = torch.full((N,), float("inf"), dtype=torch.float64)
    chosen = []
    for i in range(N):
        if len(chosen) == K:
            break
        if mind[i] > min_rel * norm[i]:
            chosen.append(i)
            mind = torch.minimum(mind, (cand - cand[i]).norm(dim=1))
    while len(chosen) < K:
        i = int(mind.argmax())
        chosen.append(i)
        mind = torch.minimum(mind, (cand - cand[i]).norm(dim=1))
    return order[torch.tensor(chosen)]


def _standardize_1x1(sd: Dict[str, Tensor], prefix: str, z: Tensor, target_std: float) -> None:
    """Rescale the 1x1 convolution `prefix` so that its output channels have zero mean and `target_std` deviation
    over the calibration vectors z [..., D] (an exact re-parametrisation of the layer: w/s, (b - mean)/s)."""
    flat = z.reshape(-1, z.shape[-1])
    mu, sg = flat.mean(0), flat.std(0).clamp(min=1e-12) / target_std
    sd[prefix + "weight"] = sd[prefix + "weight"] / sg.view(-1, 1, 1, 1)
    sd[prefix + "bias"] = (sd[prefix + "bias"] - mu) / sg


def calibrate_codebooks(sd: Dict[str, Tensor], cfg: Config, x: Tensor, seed: int = 3, standardize: bool = True) -> None:
    """Make a randomly-initialised model NON-DEGENERATE on a calibration batch, in place:

    * `standardize`: the pre-quantisation vectors of a random-init network are one common offset plus a tiny spread
      (top level of the default model: |mean| 0.75, spread 0.03), so that every vector is nearly equidistant from every
      code RELATIVE to |z|^2 + |e|^2 -- the magnitude at which the reference's fp32 formula |z|^2 - 2 z.e + |e|^2 rounds
      (bottleneck.py:56-60): 5 % of the top vectors were then coin tosses of that formula's last bit.  A trained model
      has no such offset; `quantize_conv_t/b` are re-parametrised to zero-mean outputs of deviation 0.5.
    * each level's codebook is re-seeded from that level's own pre-quantisation vectors (a random-init VQ-VAE maps
      everything to 1-2 codes), WELL SEPARATED (`pick_distinct_rows`)."""
    g = torch.Generator().manual_seed(seed)
    fb, ft = cfg.resolution_factors["bottom"], cfg.resolution_factors["top"]
    K = cfg.num_embeddings

    def pick(flat):
        return pick_distinct_rows(flat, K, g) if flat.shape[0] >= K else torch.randint(0, flat.shape[0], (K,), generator=g)

    enc_b = encoder(x, sd, "enc_b.", fb, cfg.n_res_block)
    enc_t = encoder(enc_b, sd, "enc_t.", ft, cfg.n_res_block)
    conv_t = lambda: F.conv2d(enc_t, sd["quantize_conv_t.weight"], sd["quantize_conv_t.bias"]).permute(0, 2, 3, 1)
    if standardize:
        _standardize_1x1(sd, "quantize_conv_t.", conv_t(), 0.5)
    z_t = conv_t()
    flat = z_t.reshape(-1, cfg.embed_dim)
    sd["quantize_t.embed"] = flat[pick(flat)].t().contiguous()
    sd["quantize_t.embed_avg"] = sd["quantize_t.embed"].clone()
    q_t, _, _, _ = quantize(z_t, sd["quantize_t.embed"])
    dec_t = decoder(q_t.permute(0, 3, 1, 2), sd, "dec_t.", ft, cfg.n_res_block)
    w = min(dec_t.shape[-1], enc_b.shape[-1])
    cat = torch.cat([dec_t[..., :w], enc_b[..., :w]], 1)
    conv_b = lambda: F.conv2d(cat, sd["quantize_conv_b.weight"], sd["quantize_conv_b.bias"]).permute(0, 2, 3, 1)
    if standardize:
        _standardize_1x1(sd, "quantize_conv_b.", conv_b(), 0.5)
    z_b = conv_b()
    flat = z_b.reshape(-1, cfg.embed_dim)
    sd["quantize_b.embed"] = flat[pick(flat)].t().contiguous()
    sd["quantize_b.embed_avg"] = sd["quantize_b.embed"].clone()


def near_tie_free_vectors(embed: Tensor, n: int, seed: int, scale: float = 1.0, min_gap: float = 1e-5) -> Tensor:
    """`n` seeded Gaussian vectors [n, D] none of which is a NEAR-TIE of the codebook `embed` [D, K]: the float64
    distances to the best and second-best code differ by more than `min_gap * (|z|^2 + |e|^2)`, two decades above the
    rounding of the reference's fp32 formula |z|^2 - 2 z.e + |e|^2 (bottleneck.py:56-60).  Every correct fp32
    implementation must therefore return the reference's indices BIT-EXACTLY on them.  Deterministic: seeded draw
    of 1.1 n vectors, the near-ties (about 0.1 %) dropped, the first n kept."""
    g = torch.Generator().manual_seed(seed)
    D = embed.shape[0]
    pool = torch.randn(int(n * 1.1) + 64, D, generator=g) * scale
    e = embed.double()
    e2 = e.pow(2).sum(0)
    keep = []
    for lo in range(0, pool.shape[0], 8192):
        z = pool[lo:lo + 8192].double()
        z2 = z.pow(2).sum(1, keepdim=True)
        d = z2 - 2 * z @ e + e2
        best, idx = d.topk(2, dim=1, largest=False)
        gap = (best[:, 1] - best[:, 0]) / (z2[:, 0] + e2[idx[:, 0]])
        keep.append(gap > min_gap)
    keep = torch.cat(keep)
    out = pool[keep][:n]
    assert out.shape[0] == n
    return out.contiguous()


def near_tie_gaps(z_vecs: Tensor, embed: Tensor, got: Tensor, ref: Tensor) -> Tensor:
    """For every position where `got` differs from `ref`: the float64 distance gap between the two candidate codes,
    normalised by the magnitude of the terms the reference's fp32 formula |z|^2 - 2 z.e + |e|^2 cancels
    (bottleneck.py:56-60 evaluates the distance with absolute error ~ulp(|z|^2)).  Empty when nothing differs."""
    bad = (got != ref).reshape(-1).nonzero().reshape(-1)
    if bad.numel() == 0:
        return torch.zeros(0, dtype=torch.float64)
    flat = z_vecs.reshape(-1, z_vecs.shape[-1]).double()[bad]
    e = embed.double()
    x2 = flat.pow(2).sum(1, keepdim=True)
    d = x2 - 2 * flat @ e + e.pow(2).sum(0, keepdim=True)
    dg = d.gather(1, got.reshape(-1)[bad].unsqueeze(1))
    dr = d.gather(1, ref.reshape(-1)[bad].unsqueeze(1))
    scale = x2 + e.pow(2).sum(0)[ref.reshape(-1)[bad]].unsqueeze(1)
    return ((dg - dr).abs() / scale).reshape(-1)


def teacher_forced_code_check(x: Tensor, sd: StateDict, cfg: Config, id_t: Tensor, id_b: Tensor, eps: float = 2e-6,
                              chunk: int = 16) -> Dict[str, object]:
    """Codes of another implementation (`id_t`, `id_b` for the batch `x`) against this oracle, level by level, so
    that a moved near-tie at the top never hides the bottom level behind it (tests/test_hip_parity.py):
      top     z_t of the oracle from x                          -> indices that differ from the oracle's arg-min
      bottom  z_b of the oracle from the GIVEN id_t (embed_code -> dec_t -> cat(enc_b) -> 1x1 convolution)
    Every differing index is then checked to be a NEAR-TIE of the reference's own fp32 distance formula (normalised
    float64 gap < eps, `near_tie_gaps`).  Returns counts, the largest gap and `certified` (all gaps < eps)."""
    fb, ft = cfg.resolution_factors["bottom"], cfg.resolution_factors["top"]
    moved = {"top": 0, "bottom": 0}
    worst = 0.0
    n_t = n_b = 0
    for s0 in range(0, x.shape[0], chunk):
        xs, gt, gb = x[s0:s0 + chunk], id_t[s0:s0 + chunk], id_b[s0:s0 + chunk]
        enc_b = encoder(xs, sd, "enc_b.", fb, cfg.n_res_block)
        enc_t = encoder(enc_b, sd, "enc_t.", ft, cfg.n_res_block)
        z_t = F.conv2d(enc_t, sd["quantize_conv_t.weight"], sd["quantize_conv_t.bias"]).permute(0, 2, 3, 1)
        ref_t = quantize(z_t, sd["quantize_t.embed"])[2]
        g = near_tie_gaps(z_t, sd["quantize_t.embed"], gt, ref_t)
        moved["top"] += g.numel()
        q_t = embed_code(gt, sd["quantize_t.embed"]).permute(0, 3, 1, 2)
        dec_t = decoder(q_t, sd, "dec_t.", ft, cfg.n_res_block)
        if cfg.adapt_quantized_durations:
            w = min(dec_t.shape[-1], enc_b.shape[-1])
            dec_t, enc_b = dec_t[..., :w], enc_b[..., :w]
        z_b = F.conv2d(torch.cat([dec_t, enc_b], 1), sd["quantize_conv_b.weight"], sd["quantize_conv_b.bias"]).permute(0, 2, 3, 1)
        ref_b = quantize(z_b, sd["quantize_b.embed"])[2]
        g2 = near_tie_gaps(z_b, sd["quantize_b.embed"], gb, ref_b)
        moved["bottom"] += g2.numel()
        for t in (g, g2):
            if t.numel():
                worst = max(worst, float(t.max()))
        n_t += ref_t.numel()
        n_b += ref_b.numel()
    return {"top_moved": moved["top"], "of_top": n_t, "bottom_moved_teacher_forced": moved["bottom"], "of_bottom": n_b,
            "largest_normalised_gap": worst, "near_tie_threshold": eps, "certified_near_ties": bool(worst < eps),
            "samples": int(x.shape[0])}
