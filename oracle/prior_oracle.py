"""CPU oracle for the autoregressive prior  --  TEST INFRASTRUCTURE, NOT PRODUCT.

PARITY UNPINNED for the transformer layers: the reference instantiates
`VQCPCB.transformer.transformer_custom.*` (priors/transformer.py:12-15,370-417),
a third-party package that is neither in the reference tree nor pinned by any
manifest, and the reference has no test or golden vector at that boundary.  This
file therefore *is* the specification of those layers for this repository
(Music-Transformer-style relative attention, post-norm layers); the HIP kernels
are checked against it.  The wrapper arithmetic around the layers (embeddings,
positions, start symbols, logits head, top-k/top-p filtering, label-smoothing
loss) IS in the reference tree and is pinned by `tests/golden/prior_wrapper.npz`
generated from the reference (`oracle/make_golden.py`).

Specification of the layers (all fp32, eval mode = no dropout):

  MultiheadRelativeAttention(d_model, nhead, bias_type, Cq, Eq, Ck, Ek)
      q,k,v = x Wq^T+bq, m Wk^T+bk, m Wv^T+bv   (in_proj_weight [3d,d], in_proj_bias [3d])
      head h, query position i, key position j:
        logit[i,j] = ( q_i . k_j  +  q_i . e[h, r(i,j)] ) / sqrt(d_head)  + mask[i,j]
        r(i,j)     = floor(i / Cq) - floor(j / Ck) + (Ek - 1)      in [0, Eq + Ek - 2]
      `floor(i / C)` is the *event* (time step) of a token when C tokens
      ("channels") share one event (priors/transformer.py:210-219,353-359);
      rel_embeddings e: [nhead, Eq + Ek - 1, d_head]; bias_type 'no_bias' drops the e term.
      out = softmax_j(logit) v  Wo^T + bo
  EncoderLayer:  x = LN1(x + SelfAttn(x));  x = LN2(x + W2 relu(W1 x + b1) + b2)
  DecoderLayer:  x = LN1(x + SelfAttn(x, causal));  x = LN2(x + CrossAttn(x, memory));
                 x = LN3(x + FFN(x))
  LayerNorm eps 1e-5; dim_feedforward 2048.
"""
from __future__ import annotations

import math
from typing import Dict, Mapping, Optional

import torch
import torch.nn.functional as F

Tensor = torch.Tensor
NEG_INF = float("-inf")


def rel_index(Sq: int, Sk: int, Cq: int, Ck: int, Ek: int) -> Tensor:
    i = torch.arange(Sq).unsqueeze(1) // Cq
    j = torch.arange(Sk).unsqueeze(0) // Ck
    return i - j + (Ek - 1)


def attention(x: Tensor, mem: Tensor, sd: Mapping[str, Tensor], prefix: str, nhead: int,
              Cq: int, Eq: int, Ck: int, Ek: int, mask: Optional[Tensor], bias: bool = True) -> Tensor:
    """x [Sq,B,d] queries, mem [Sk,B,d] keys/values (time-major like the reference's call sites)."""
    Sq, B, d = x.shape
    Sk = mem.shape[0]
    hd = d // nhead
    W, b = sd[prefix + "in_proj_weight"], sd[prefix + "in_proj_bias"]
    q = F.linear(x, W[:d], b[:d]).reshape(Sq, B, nhead, hd).permute(1, 2, 0, 3)
    k = F.linear(mem, W[d:2 * d], b[d:2 * d]).reshape(Sk, B, nhead, hd).permute(1, 2, 0, 3)
    v = F.linear(mem, W[2 * d:], b[2 * d:]).reshape(Sk, B, nhead, hd).permute(1, 2, 0, 3)
    logits = q @ k.transpose(-1, -2)                                   # [B,H,Sq,Sk]
    if bias:
        e = sd[prefix + "rel_embeddings"]                              # [H, Eq+Ek-1, hd]
        qe = torch.einsum("bhid,hrd->bhir", q, e)                      # [B,H,Sq,R]
        idx = rel_index(Sq, Sk, Cq, Ck, Ek).clamp(0, e.shape[1] - 1)
        logits = logits + qe.gather(3, idx.expand(B, nhead, Sq, Sk))
    logits = logits / math.sqrt(hd)
    if mask is not None:
        logits = logits + mask
    p = torch.softmax(logits, dim=-1)
    o = (p @ v).permute(2, 0, 1, 3).reshape(Sq, B, d)
    return F.linear(o, sd[prefix + "out_proj.weight"], sd[prefix + "out_proj.bias"])


def _ffn(x, sd, prefix):
    return F.linear(F.relu(F.linear(x, sd[prefix + "linear1.weight"], sd[prefix + "linear1.bias"])),
                    sd[prefix + "linear2.weight"], sd[prefix + "linear2.bias"])


def _ln(x, sd, prefix):
    return F.layer_norm(x, (x.shape[-1],), sd[prefix + "weight"], sd[prefix + "bias"], 1e-5)


def causal_mask(S: int) -> Tensor:
    """priors/transformer.py:483-500: 0 where j <= i, -inf elsewhere."""
    return torch.zeros(S, S).masked_fill(torch.triu(torch.ones(S, S), 1) == 1, NEG_INF)


def encoder(src: Tensor, sd, prefix: str, num_layers: int, nhead: int, C: int, E: int,
            mask: Optional[Tensor]) -> Tensor:
    x = src
    for l in range(num_layers):
        p = f"{prefix}layers.{l}."
        x = _ln(x + attention(x, x, sd, p + "self_attn.", nhead, C, E, C, E, mask), sd, p + "norm1.")
        x = _ln(x + _ffn(x, sd, p), sd, p + "norm2.")
    return x


def decoder(tgt: Tensor, memory: Tensor, sd, prefix: str, num_layers: int, nhead: int,
            Cd: int, Ed: int, Ce: int, Ee: int, tgt_mask: Optional[Tensor],
            memory_mask: Optional[Tensor], cross_bias: bool = True) -> Tensor:
    x = tgt
    for l in range(num_layers):
        p = f"{prefix}layers.{l}."
        x = _ln(x + attention(x, x, sd, p + "self_attn.", nhead, Cd, Ed, Cd, Ed, tgt_mask), sd, p + "norm1.")
        x = _ln(x + attention(x, memory, sd, p + "multihead_attn.", nhead, Cd, Ed, Ce, Ee, memory_mask,
                              bias=cross_bias), sd, p + "norm2.")
        x = _ln(x + _ffn(x, sd, p), sd, p + "norm3.")
    return x


def top_k_top_p_filtering(logits: Tensor, top_k: int = 0, top_p: float = 0.0) -> Tensor:
    """sample.py:36-65 on a copy (the reference filters in place)."""
    logits = logits.clone()
    top_k = min(top_k, logits.size(-1))
    if top_k > 0:
        kth = torch.topk(logits, top_k)[0][..., -1, None]
        logits[logits < kth] = NEG_INF
    if top_p > 0.0:
        sorted_logits, sorted_indices = torch.sort(logits, descending=True)
        cum = torch.cumsum(F.softmax(sorted_logits, dim=-1), dim=-1)
        remove = cum > top_p
        remove[..., 1:] = remove[..., :-1].clone()
        remove[..., 0] = 0
        logits[remove.scatter(-1, sorted_indices, remove)] = NEG_INF
    return logits


def sample_from_uniform(probs: Tensor, u: Tensor) -> Tensor:
    """Inverse-CDF categorical draw with host-supplied uniforms u in [0,1):
    the first index whose inclusive cumulative probability exceeds u * total."""
    cdf = torch.cumsum(probs, dim=-1)
    total = cdf[..., -1:]
    idx = (cdf <= u.unsqueeze(-1) * total).sum(-1)
    return idx.clamp(max=probs.shape[-1] - 1)


def label_smoothing_loss(pred: Tensor, target: Tensor, num_classes: int, smoothing: float, dim: int = 1) -> Tensor:
    """utils/losses/prediction.py:14-20."""
    logp = pred.log_softmax(dim=dim)
    true = torch.full_like(logp, smoothing / (num_classes - 1))
    true.scatter_(1, target.unsqueeze(1), 1.0 - smoothing)
    return torch.mean(torch.sum(-true * logp, dim=dim))
