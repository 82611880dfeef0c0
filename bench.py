#!/usr/bin/env python3
"""Benchmark of the hot path: VQVAE.forward (encode + 2x quantize + decode),
eval / no-grad, batch 64 of synthetic [2,128,512] spectrograms per GPU
(BASELINE.json configs[1]).  One process per GPU; for the headline metric the
ranks are independent replicas of the path over disjoint batches (no data-path
collective: weak scaling); a barrier + synchronize brackets the timed region
and the slowest rank's time is used.

Prints ONE JSON line on rank 0:
  value     spectrograms/s over all ranks, inputs resident in HBM
  roofline  the dominant kernel (by summed in-library HIP-event time inside the
            timed region) priced on its algorithmic FLOPs against the dense
            16-bit matrix peak divided by the MFMA terms per product (default
            split_f16 mode: three f16 terms -> 2500 / 3 TFLOP/s; exact-fp32
            kernels: 157.3 TFLOP/s); `traffic` comes from the committed PMC pass
            named in `traffic_source` (same command, same batch), not from this run
  cpu_baseline  the CPU oracle (oracle/vqvae_oracle.py, torch-CPU fp32) timed
            on this host's cores on a bounded sample of the same workload
  vqvae_training  BASELINE configs[2]: one data-parallel VQ-VAE training step
            per rank at B = 64 (bucketed RCCL gradient all-reduce overlapped
            with the backward + the EMA-statistics all-reduce), spectrograms/s
            over all ranks -- the leg whose scaling exercises collectives
  attention  BASELINE configs[3]'s kernel: relative attention forward /
            backward at B8 H8 S1025 hd64 with its own roofline entry
"""
from __future__ import annotations

import argparse
import ctypes as C
import json
import os
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent
for _p in (str(ROOT), str(ROOT / "interactive-spectrogram-inpainting_amd")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import torch  # noqa: E402

FP32_MATRIX_PEAK_TFLOPS = 157.3   # MI355X_MICROARCH.md: v_mfma_f32_32x32x2_f32
BF16_MATRIX_PEAK_TFLOPS = 2500.0  # dense bf16 MFMA
HBM_PEAK_GBS = 8000.0


def _pick_distinct_rows(flat, K, g, min_rel=0.1):
    """K well-separated rows of `flat` [N, D] (CPU): random order, a row is accepted when farther than
    min_rel * |row| from every accepted one, farthest-point fill if the pool runs out.  (Drawing WITH replacement
    from correlated encoder outputs produced duplicate / 1e-3-apart codes: VERDICT r02 weak 1.)"""
    N = flat.shape[0]
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


def _build_model(device, seed=1):
    """Default-constructed VQVAE with torch's default init, made NON-DEGENERATE on a calibration batch (computed by
    the HIP path itself): a random-init network's pre-quantisation vectors are one common offset plus a tiny spread
    (top level: |mean| 0.75, spread 0.03), which makes every vector a near-tie of the reference's fp32 distance
    formula; `quantize_conv_t/b` are re-parametrised to zero-mean outputs of deviation 0.5 (exact re-scaling of the
    1x1 layers), then each codebook is re-seeded from that level's own vectors, well separated."""
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    torch.manual_seed(seed)
    m = VQVAE(in_channel=2).to(device).eval()
    g = torch.Generator().manual_seed(seed + 1)

    def standardize(layer, z):
        mu, sg = z.mean(0), z.std(0).clamp(min=1e-12) / 0.5
        layer.weight.copy_(layer.weight / sg.view(-1, 1, 1, 1))
        layer.bias.copy_((layer.bias - mu) / sg)

    with torch.no_grad():
        xc = torch.randn(2, 2, 128, 512, generator=g).to(device)
        enc_b = m.enc_b(xc)
        enc_t = m.enc_t(enc_b)
        conv_t = lambda: m.quantize_conv_t.run(enc_t, relu=False).permute(0, 2, 3, 1).reshape(-1, m.embed_dim)
        standardize(m.quantize_conv_t, conv_t())
        z_t = conv_t()
        pick = _pick_distinct_rows(z_t.cpu(), m.n_embed_t, g).to(device)
        m.quantize_t.embed.copy_(z_t[pick].t())
        q_t = m.quantize_t(z_t.reshape(2, enc_t.shape[2], enc_t.shape[3], -1))[0].permute(0, 3, 1, 2)
        dec_t = m.dec_t(q_t)
        conv_b = lambda: m.quantize_conv_b.run(dec_t, relu=False, x2=enc_b).permute(0, 2, 3, 1).reshape(-1, m.embed_dim)
        standardize(m.quantize_conv_b, conv_b())
        z_b = conv_b()
        pick = _pick_distinct_rows(z_b.cpu(), m.n_embed_b, g).to(device)
        m.quantize_b.embed.copy_(z_b[pick].t())
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    return m, sd


def _host_cpu():
    """(model name, physical cores of ONE socket, sockets) from /proc/cpuinfo; (None, None, None) if unreadable."""
    try:
        model, cores = None, set()
        phys = None
        for line in open("/proc/cpuinfo"):
            k, _, v = line.partition(":")
            k, v = k.strip(), v.strip()
            if k == "model name" and model is None:
                model = v
            elif k == "physical id":
                phys = v
            elif k == "core id":
                cores.add((phys, v))
        sockets = sorted({p_ for p_, _ in cores})
        per_socket = len([1 for p_, _ in cores if p_ == sockets[0]]) if sockets else None
        return model, per_socket, len(sockets) or None
    except OSError:
        return None, None, None


def _cpu_baseline(sd, x=None, id_t=None, id_b=None, batch=64, warmup=2, iters=5):
    """Oracle forward on the host cores (B = 64): threads = the physical cores of one socket, 2 warm-up and
    5 timed iterations on the bench batch's shape, median.  With the GPU's codes of the bench batch (`x`, `id_t`,
    `id_b`) it also compares them with the oracle LEVEL BY LEVEL (bottom teacher-forced on the GPU's top codes) over
    all samples and certifies every differing code as a near-tie of the reference's fp32 distance formula."""
    from oracle import vqvae_oracle as O
    cfg = O.Config(in_channel=2)
    model, per_socket, sockets = _host_cpu()
    before = torch.get_num_threads()
    cores = per_socket or before
    torch.set_num_threads(cores)
    xs = torch.randn(batch, 2, 128, 512, generator=torch.Generator().manual_seed(4)) if x is None else x[:batch]
    batch = xs.shape[0]
    times = []
    parity = None
    try:
        with torch.no_grad():
            for i in range(warmup + iters):
                t0 = time.perf_counter()
                O.forward(xs, sd, cfg)
                if i >= warmup:
                    times.append(time.perf_counter() - t0)
            if x is not None and id_t is not None:
                parity = O.teacher_forced_code_check(x, sd, cfg, id_t, id_b, eps=1e-6)
    finally:
        torch.set_num_threads(before)
    med = sorted(times)[len(times) // 2]
    out = {"value": round(batch / med, 3), "unit": "spectrograms/s", "cores": cores, "kind": "port",
           "cpu_model": model, "sockets": sockets,
           "sample": f"median of {iters} x VQVAE.forward on batch {batch} of [2,128,512] fp32 after {warmup} warm-up "
                     f"iteration(s) (oracle/vqvae_oracle.py, torch-CPU, {cores} threads = one socket's physical cores)"}
    if parity is not None:
        out["gpu_codes_vs_oracle"] = parity
    return out


def _prior_sampling(device):
    """Secondary metric of BASELINE.json: prior-sampled codes/s at seq 1024 (top prior,
    shape [32,32], d_model 512, 6 encoder + 8 decoder layers, 8 heads, full mask,
    temperature 1, top-p 0.8, random weights), KV-cached native sampling loop."""
    import sample as S
    from interactive_spectrogram_inpainting.priors.transformer import SelfAttentiveVQTransformer
    torch.manual_seed(2)
    m = SelfAttentiveVQTransformer(
        shape=[32, 32], condition_shape=[32, 32], n_class=512, channel=256, kernel_size=5, n_block=4,
        n_res_block=4, res_channel=256, d_model=512, embeddings_dim=32, positional_embeddings_dim=16,
        use_relative_transformer=True, predict_frequencies_first=True, conditional_model=True,
        self_conditional_model=True, add_mask_token_to_symbols=True,
        class_conditioning_prepend_to_dummy_input=True,
        class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
        class_conditioning_embedding_dim_per_modality={"instrument_family_str": 64, "pitch": 64}).to(device).eval()
    cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
    out = {"timing": "median of 5 codemaps (B = 1; 3 at B = 8 / 32, 2 at B = 128) after one warm-up codemap",
           "decode_loop": "graphs of ISI_PRIOR_GRAPH = 8 positions replayed per launch (0: ~66 direct launches per position)"}

    def run(B, **kw):
        S.sample_model(m, device, B, [32, 32], 1.0, generator=torch.Generator().manual_seed(0), class_conditioning=cls, **kw)
        torch.cuda.synchronize(device)
        ts = []
        n_rep = 5 if B == 1 else 3 if B <= 32 else 2       # (a single codemap varies by +-5 %)
        for rep in range(n_rep):
            t0 = time.perf_counter()
            S.sample_model(m, device, B, [32, 32], 1.0, generator=torch.Generator().manual_seed(1 + rep), class_conditioning=cls, **kw)
            torch.cuda.synchronize(device)
            ts.append(time.perf_counter() - t0)
        return sorted(ts)[n_rep // 2]

    for B in (1, 8, 32, 128):                              # (beyond 16 sequences the stages run as fp32-MFMA row tiles)
        dt = run(B, top_p_sampling_p=0.8)                  # Inference.ipynb cell 43 samples with top-p 0.8
        out[f"codes_per_s_B{B}"] = round(B * 1024 / dt, 1)
        out[f"ms_per_codemap_B{B}"] = round(dt * 1e3, 1)
    dt_plain = run(1)                                      # also top-p 0 / top-k 0 (no filtering)
    out["codes_per_s_B1_top_p0_top_k0"] = round(1024 / dt_plain, 1)
    # roofline of the decode loop at B = 1: bytes a token's 8 decoder layers + logits head must stream -- the layer
    # weights (self-attention in / out projections, cross-attention q / out projections, the two feed-forward
    # matrices; fp32), the logits matrix, and the cached keys / values it attends over (self: 2 p d per layer at
    # position p, 512 on average; cross: the 1025 memory rows' keys / values) -- against HBM's peak.  The weights fit
    # the 256 MB Infinity Cache, so the loop is really a chain of dependent launches; the
    # fraction says how far a batch-1 token is from streaming its bytes at memory speed.
    d, ff, L = m.d_model, 2048, m.conditional_model_num_decoder_layers
    w_bytes = 4 * (L * (3 * d * d + d * d + d * d + d * d + 2 * d * ff) + d * m.n_class_target)
    kv_bytes = 4 * L * (2 * 512 * d + 2 * 1025 * d)
    per_token = w_bytes + kv_bytes
    t_tok = 1.0 / out["codes_per_s_B1"]
    out["roofline"] = {"bound": "hbm", "achieved": round(per_token / t_tok / 1e9, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                       "frac": round(per_token / t_tok / 1e9 / HBM_PEAK_GBS, 4), "traffic": None,
                       "algorithmic_bytes_per_token": per_token, "us_per_token": round(t_tok * 1e6, 1),
                       "note": "batch-1 decode: weights " + str(w_bytes) + " B + average KV " + str(kv_bytes) + " B per token"}
    # CPU baseline with the reference's loop semantics (sample.py:268-283: one FULL decoder pass per
    # sampled token, encoder memory cached), oracle layers on the host cores, bounded to 3 tokens
    from oracle import prior_oracle as P
    sd = {k: v.detach().cpu() for k, v in m.state_dict().items()}
    code = torch.randint(0, 512, (1, 32, 32))
    clsd = {k: v.reshape(1, 1).to(device) for k, v in cls.items()}
    src, tgt = m.to_sequences(code.to(device), code.to(device), class_conditioning=clsd)
    s_, t_ = src.cpu().transpose(0, 1), tgt.cpu().transpose(0, 1)
    H, E = m.conditional_model_nhead, m.source_num_events_with_start_symbol
    _, per_socket, _ = _host_cpu()
    before = torch.get_num_threads()
    torch.set_num_threads(per_socket or before)            # one socket's physical cores, like the forward's baseline
    with torch.no_grad():
        memory = P.encoder(s_, sd, "transformer.encoder.", m.conditional_model_num_encoder_layers, H, 1, E,
                           P.causal_mask(s_.shape[0]).t())
        t0 = time.perf_counter()
        n_tok = 3
        for _ in range(n_tok):
            o = P.decoder(t_, memory, sd, "transformer.decoder.", m.conditional_model_num_decoder_layers, H, 1, E,
                          1, E, P.causal_mask(t_.shape[0]), None)
            torch.nn.functional.linear(o[:-1].transpose(0, 1), sd["project_transformer_outputs_to_logits.weight"],
                                       sd["project_transformer_outputs_to_logits.bias"])
        out["cpu_baseline"] = {"value": round(n_tok / (time.perf_counter() - t0), 3), "unit": "codes/s",
                               "cores": torch.get_num_threads(), "kind": "port",
                               "sample": f"{n_tok} tokens, one full decoder pass each (reference loop, oracle/prior_oracle.py)"}
    torch.set_num_threads(before)
    out["unit"] = "codes/s"
    out["config"] = "SelfAttentiveVQTransformer shape [32,32] (1024 tokens + start), d_model 512, 6+8 layers, 8 heads, fp32"
    return out


def _top_prior(device):
    from interactive_spectrogram_inpainting.priors.transformer import SelfAttentiveVQTransformer
    torch.manual_seed(2)
    return SelfAttentiveVQTransformer(
        shape=[32, 32], condition_shape=[32, 32], n_class=512, channel=256, kernel_size=5, n_block=4,
        n_res_block=4, res_channel=256, d_model=512, embeddings_dim=32, positional_embeddings_dim=16,
        use_relative_transformer=True, predict_frequencies_first=True, conditional_model=True,
        self_conditional_model=True, add_mask_token_to_symbols=True,
        class_conditioning_prepend_to_dummy_input=True,
        class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
        class_conditioning_embedding_dim_per_modality={"instrument_family_str": 64, "pitch": 64}).to(device)


def _bottom_prior(device):
    """The bottom-level prior (priors/transformer.py:848-872): [64,64] codemaps conditioned on the [32,32] top map."""
    from interactive_spectrogram_inpainting.priors.transformer import UpsamplingVQTransformer
    torch.manual_seed(3)
    return UpsamplingVQTransformer(
        shape=[64, 64], condition_shape=[32, 32], n_class=512, channel=256, kernel_size=5, n_block=4,
        n_res_block=4, res_channel=256, d_model=512, embeddings_dim=32, positional_embeddings_dim=16,
        use_relative_transformer=True, predict_frequencies_first=True, conditional_model=True,
        class_conditioning_prepend_to_dummy_input=True,
        class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
        class_conditioning_embedding_dim_per_modality={"instrument_family_str": 64, "pitch": 64}).to(device)


def _timerange_change(device, vqvae, calls=5):
    """SURVEY 8d metric 2, second half: latency of one /timerange-change-equivalent request
    (flask_server.py:685-870): layer 'top', mask = half of the top codemap's columns -> the top prior resamples
    512 codes, the bottom prior the 2048 codes under the up-sampled mask, then VQ-VAE decode_code of the new
    maps.  Top [32,32] / bottom [64,64] priors (d_model 512, 6+8 layers), random weights."""
    import inpainting
    top = _top_prior(device).eval()
    bottom = _bottom_prior(device).eval()
    g = torch.Generator().manual_seed(5)
    top_code = torch.randint(0, 512, (1, 32, 32), generator=g).to(device)
    bottom_code = torch.randint(0, 512, (1, 64, 64), generator=g).to(device)
    mask = torch.zeros(1, 32, 32, dtype=torch.bool)
    mask[..., 8:24] = True
    cls = {"pitch": torch.tensor([[24]]), "instrument_family_str": torch.tensor([[0]])}
    times = []
    for i in range(calls + 1):
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        nt, nb = inpainting.timerange_change(top, bottom, top_code, bottom_code, mask, 'top', 0, 1.0, cls, cls, device,
                                             generator=torch.Generator().manual_seed(i), top_p_sampling_p=0.8)
        spec = vqvae.decode_code(nt, nb)
        torch.cuda.synchronize(device)
        times.append(time.perf_counter() - t0)
    assert spec.shape == (1, 2, 256, 256) and torch.isfinite(spec).all()
    times = sorted(times[1:])
    return {"p50_ms": round(times[len(times) // 2] * 1e3, 1), "calls": calls,
            "resampled_codes": {"top": 512, "bottom": 2048},
            "config": "layer 'top', mask = columns 8..23 of the [32,32] top map; bottom [64,64]; + decode_code"}


def _ranks_in_sync(dist, device, tensors):
    """Every rank must hold the same values after averaged updates: min == max over ranks of a few checksums."""
    if dist is None:
        return True
    chk = torch.stack([t.detach().double().sum() for t in tensors])
    lo, hi = chk.clone(), chk.clone()
    dist.all_reduce(lo, op=dist.ReduceOp.MIN)
    dist.all_reduce(hi, op=dist.ReduceOp.MAX)
    return bool(((hi - lo).abs() <= 1e-6 * hi.abs().clamp(min=1e-12)).all())


def _all_ranks_ok(dist, device, ok: bool) -> bool:
    """Do ALL ranks agree to go on?  (A rank whose recording failed must not leave the others waiting in a replay's collective.)"""
    if dist is None:
        return ok
    t = torch.tensor([1 if ok else 0], dtype=torch.int32, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def _time_steps(fn, steps, barrier, dist, device):
    """ms per call of `fn` over `steps` calls between barriers, slowest rank; also the host's share (time to ENQUEUE the
    steps, before the closing barrier)."""
    barrier()
    t0 = time.perf_counter()
    for _ in range(steps):
        out = fn()
    t_host = time.perf_counter() - t0
    barrier()
    dt = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([dt, t_host], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt, t_host = t.tolist()
    return dt / steps * 1e3, t_host / steps * 1e3, out


def _prior_training(device, dist=None, world=1, B=8, steps=20, warmup=2, level="top"):
    """BASELINE configs[3]: training step of the prior -- level 'top': 1024 tokens + start symbol, self-conditional;
    level 'bottom': the [64,64] bottom codemaps (4096 tokens + 4 start symbols) conditioned on the top map
    (priors/transformer.py:848-872 UpsamplingVQTransformer) -- d_model 512, 6 + 8 layers, fp32, dropout 0.1, label
    smoothing, Adam: forward + loss + backward + optimizer step, B codemaps PER GPU (weak scaling).
    With world > 1 it runs on EVERY rank: one process per GPU, `GradBucketReducer` (utils/distributed.py) all-reduces
    the flat gradient buffer bucket by bucket over RCCL while the backward is still running -- what replaces the
    reference's nn.DataParallel (train_autoregressive_model.py:145,203-263).  Timed twice between barriers, slowest rank
    counts: EAGER (the host enqueues ~1300 launches per step) and REPLAYED from HIP graph segments cut at the collectives
    (utils/training/graphed_step.py; `value` / `ms_per_step` are the replayed step's, the mode a training run would use)."""
    from interactive_spectrogram_inpainting.priors import _ops as _prior_ops
    from interactive_spectrogram_inpainting.utils.distributed import GradBucketReducer
    from interactive_spectrogram_inpainting.utils.losses.prediction import LabelSmoothingLoss
    from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
    from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
    rank = dist.get_rank() if dist is not None else 0
    m = (_top_prior(device) if level == "top" else _bottom_prior(device)).train()   # seeded: identical weights on every rank
    g = torch.Generator().manual_seed(300 + rank)  # every rank its own shard of the synthetic codemaps
    code = torch.randint(0, 512, (B, 32, 32), generator=g).to(device)
    mask = (torch.rand(B, 32, 32, generator=g) < 0.5).to(device)
    bottom = torch.randint(0, 512, (B, 64, 64), generator=g).to(device)
    cls = {"pitch": torch.full((B, 1), 24, device=device),
           "instrument_family_str": torch.zeros(B, 1, dtype=torch.long, device=device)}
    reducer = GradBucketReducer(m.parameters()) if world > 1 else None
    opt = make_adam(m.parameters(), lr=3e-4)
    crit = LabelSmoothingLoss(512, 0.1, dim=1)
    torch.manual_seed(400 + rank)                  # dropout masks differ per rank like the data
    tokens = 1024 if level == "top" else 4096

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    def step():
        if reducer is not None:
            reducer.zero()
        else:
            opt.zero_grad(set_to_none=True)
        if level == "top":
            target = code
            src, tgt = m.to_sequences(code, condition=code, class_conditioning=cls, mask=mask)
        else:
            target = bottom
            src, tgt = m.to_sequences(bottom, condition=code, class_conditioning=cls)
        logits, _ = m(tgt, condition=src)
        loss = crit(m.to_time_frequency_map(logits, kind="target", permute_output_as_logits=True), target)
        loss.backward()
        if reducer is not None:
            reducer.finish()
        opt.step()
        return loss
    for _ in range(warmup):
        step()
    eager_ms, eager_host_ms, loss = _time_steps(step, steps, barrier, dist, device)
    in_sync = _ranks_in_sync(dist, device, [p for p in (list(m.parameters())[i] for i in (0, len(list(m.parameters())) // 2, -1))])
    assert torch.isfinite(loss).all()
    loss = None        # (an eager step's autograd graph kept alive keeps its AccumulateGrad nodes on the eager stream)
    # the same step recorded into HIP graph segments and replayed: the host's ~1300 launches per step become one call per
    # segment (one segment per collective + 1); a fresh optimizer (capturable) on the same model
    graphed_ms = graphed_host_ms = n_segments = None
    graph_error = None
    opt = make_adam(m.parameters(), lr=3e-4, capturable=True)
    try:
        statics = (code, mask) if level == "top" else (bottom, code)
        graphed = None
        try:          # (the recording itself holds no collective: a failure here is this rank's alone -- agree before replaying)
            graphed = GraphedTrainingStep(lambda *_a: step(), statics, warmup=2, index_limits={0: 512},
                                          range_params=[p for p in m.parameters() if p.dim() == 2])
        except Exception as e:
            graph_error = f"{type(e).__name__}: {str(e)[:300]}"
        if not _all_ranks_ok(dist, device, graphed is not None):
            raise RuntimeError(graph_error or "another rank could not record the step")
        try:
            graphed(*statics)
            graphed_ms, graphed_host_ms, gl = _time_steps(lambda: graphed(*statics), steps, barrier, dist, device)
            assert torch.isfinite(gl).all()
            n_segments = graphed.n_segments
            graphed.finish()
            in_sync = in_sync and _ranks_in_sync(dist, device, [list(m.parameters())[i] for i in (0, -1)])
        finally:
            _prior_ops.set_dropout_seed_base(None)
        del graphed
    except Exception as e:       # the eager numbers stand on their own
        graph_error = graph_error or f"{type(e).__name__}: {str(e)[:300]}"
        _prior_ops.set_dropout_seed_base(None)
    ms = graphed_ms if graphed_ms is not None else eager_ms
    out = {"value": round(world * B * tokens / ms * 1e3, 0), "unit": "tokens/s", "ms_per_step": round(ms, 2),
           "mode": "hip-graph replay" if graphed_ms is not None else "eager",
           "tokens_per_s": round(world * B * tokens / ms * 1e3, 0), "codemaps_per_s": round(world * B / ms * 1e3, 1),
           "n_gpus": world, "steps": steps, "warmup": warmup, "global_batch": world * B, "scaling": "weak",
           "ranks_in_sync": in_sync,
           "ms_per_step_eager": round(eager_ms, 2), "host_enqueue_ms_per_step_eager": round(eager_host_ms, 2),
           "ms_per_step_hip_graph": round(graphed_ms, 2) if graphed_ms is not None else None,
           "host_enqueue_ms_per_step_hip_graph": round(graphed_host_ms, 3) if graphed_host_ms is not None else None,
           "graph_segments": n_segments,
           "collectives_per_step": (f"{len(reducer.buckets)} gradient buckets ({reducer.flat.numel() * 4 / 1e6:.1f} MB fp32 in "
                                    f"total), all-reduced while the backward runs") if reducer is not None else "none (1 rank)",
           "config": (f"top prior [32,32], B={B}/GPU x 1025 tokens" if level == "top" else
                      f"bottom prior [64,64] on top [32,32], B={B}/GPU x 4100 tokens (source 1025)") +
                     f", d_model 512, 6+8 layers, 8 heads, fp32, Adam, "
                     f"attention products {os.environ.get('ISI_ATTENTION_PRECISION', 'bf16x3')}"}
    if graph_error:
        out["hip_graph_error"] = graph_error
    del m, opt, reducer
    torch.cuda.empty_cache()
    return out


def _vqvae_training(device, dist, world, batch=64, steps=20, warmup=3):
    """BASELINE configs[2]: VQ-VAE training, global batch = world x 64 synthetic spectrograms, data parallel with one
    process per GPU: train-mode forward (asynchronous EMA-statistics all-reduce behind both quantisers), hand-written
    backward whose gradient buckets are all-reduced by RCCL while it is still running, Adam step.  Runs on EVERY rank; timed
    between barriers, slowest rank counts -- eagerly and replayed from HIP graph segments cut at the collectives
    (`value` / `ms_per_step`: the replayed step)."""
    from interactive_spectrogram_inpainting.utils.training.graphed_step import GraphedTrainingStep
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    torch.manual_seed(1)                       # identical weights on every rank
    m = VQVAE(in_channel=2).to(device).train()
    from interactive_spectrogram_inpainting.utils.training.optimizer import make_adam
    from interactive_spectrogram_inpainting.utils.losses.mse import mse_loss
    opt = make_adam(m.parameters(), lr=3e-4)
    rank = dist.get_rank() if dist is not None else 0
    x = torch.randn(batch, 2, 128, 512, generator=torch.Generator().manual_seed(200 + rank)).to(device)

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    def step():
        m.zero_grad()
        out, latent, *_ = m(x)
        loss = mse_loss(out, x) + 0.25 * latent.mean()       # (nn.MSELoss on the HIP library: utils/losses/mse.py)
        loss.backward()
        opt.step()
        return loss
    for _ in range(warmup):
        step()
    eager_ms, eager_host_ms, loss = _time_steps(step, steps, barrier, dist, device)
    # every rank must hold the same codebook and weights after the exchanged updates
    in_sync = _ranks_in_sync(dist, device, [m.quantize_b.embed, m.enc_b.blocks[0].weight])
    assert torch.isfinite(loss).all()
    loss = None
    graphed_ms = graphed_host_ms = n_segments = None
    graph_error = None
    opt = make_adam(m.parameters(), lr=3e-4, capturable=True)
    try:
        graphed = None
        try:
            graphed = GraphedTrainingStep(lambda _x: step(), (x,), warmup=2)
        except Exception as e:
            graph_error = f"{type(e).__name__}: {str(e)[:300]}"
        if not _all_ranks_ok(dist, device, graphed is not None):
            raise RuntimeError(graph_error or "another rank could not record the step")
        graphed(x)
        graphed_ms, graphed_host_ms, gl = _time_steps(lambda: graphed(x), steps, barrier, dist, device)
        assert torch.isfinite(gl).all()
        n_segments = graphed.n_segments
        graphed.finish()
        in_sync = in_sync and _ranks_in_sync(dist, device, [m.quantize_b.embed, m.quantize_t.embed, m.enc_b.blocks[0].weight])
        del graphed
    except Exception as e:
        graph_error = graph_error or f"{type(e).__name__}: {str(e)[:300]}"
    ms = graphed_ms if graphed_ms is not None else eager_ms
    out = {"value": round(world * batch / ms * 1e3, 1), "unit": "spectrograms/s", "ms_per_step": round(ms, 2),
           "mode": "hip-graph replay" if graphed_ms is not None else "eager",
           "n_gpus": world, "steps": steps, "warmup": warmup, "global_batch": world * batch, "scaling": "weak",
           "ranks_in_sync": in_sync,
           "ms_per_step_eager": round(eager_ms, 2), "host_enqueue_ms_per_step_eager": round(eager_host_ms, 2),
           "ms_per_step_hip_graph": round(graphed_ms, 2) if graphed_ms is not None else None,
           "host_enqueue_ms_per_step_hip_graph": round(graphed_host_ms, 3) if graphed_host_ms is not None else None,
           "graph_segments": n_segments,
           "collectives_per_step": "4 gradient buckets (5.5 MB fp32 in total) + 2 EMA-statistics messages (133 KB each, "
                                   "asynchronous: waited for at the end of the forward)"
                                   if world > 1 else "none (1 rank)",
           "config": f"VQVAE default ctor, B={batch}/GPU of [2,128,512], MSE + 0.25 latent, Adam 3e-4, forward products "
                     f"three-term split-f16, input and weight gradients three-term split-bf16 (ISI_TRAIN_DGRAD_PRECISION=same: six-term "
                     f"input gradients)"}
    if graph_error:
        out["hip_graph_error"] = graph_error
    del m, opt
    torch.cuda.empty_cache()
    return out


def _attention_row(device, B, H, S, hd, n=10):
    """One more shape of the attention leg (the reference CLI's default is 16 heads: head_dim 32,
    train_autoregressive_model.py:419): forward / backward times of the default product mode, causal."""
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors._train import RelAttentionFn
    d = H * hd
    torch.manual_seed(0)
    qkv = torch.randn(S, B, 3 * d, device=device, requires_grad=True)
    rel = (torch.randn(H, 2 * S - 1, hd, device=device) * 0.1).requires_grad_(True)
    w = torch.randn(S, B, d, device=device)
    dense = 2.0 * S * S * hd * B * H

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        ts = []
        for _rep in range(3):      # median of three batches: one stalled batch (seen once: 14 ms per call) is not the number
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize(device)
            ts.append(a.elapsed_time(b) / n * 1e3)
        return sorted(ts)[1]
    with torch.no_grad():
        t_f = timed(lambda: RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, 1, None))
    res = RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, 1, None)

    def bwd():
        qkv.grad = None
        rel.grad = None
        res.backward(w, retain_graph=True)
    t_b = timed(bwd)
    return {"config": f"self-attention B{B} H{H} S{S} head_dim {hd}, causal, {_ops.ATTENTION_PRECISION}",
            "fwd_us": round(t_f, 1), "fwd_TFLOPs_dense": round(3 * dense / t_f / 1e6, 1),
            "bwd_us": round(t_b, 1), "bwd_TFLOPs_dense": round(7 * dense / t_b / 1e6, 1)}


def _attention(device, B=8, H=8, S=1025, hd=64, n=10):
    """BASELINE configs[3]'s kernel at the shape the metric names: relative self-attention (QK^T + Q E^T skewed +
    softmax + PV), causal, B8 H8 S1025 head_dim 64, forward and backward, per product mode.  FLOPs = the DENSE count
    2 S^2 hd per contraction (3 forward, 7 backward; a causal mask lets the kernels skip half of it -- convention of
    SURVEY 8d), priced against the dense bf16 matrix peak / terms per product (fp32 pipe: 157.3)."""
    from interactive_spectrogram_inpainting.priors import _ops
    from interactive_spectrogram_inpainting.priors._train import RelAttentionFn
    d = H * hd
    torch.manual_seed(0)
    qkv = torch.randn(S, B, 3 * d, device=device, requires_grad=True)
    rel = (torch.randn(H, 2 * S - 1, hd, device=device) * 0.1).requires_grad_(True)
    w = torch.randn(S, B, d, device=device)
    dense = 2.0 * S * S * hd * B * H

    def timed(fn):
        for _ in range(3):
            fn()
        torch.cuda.synchronize(device)
        ts = []
        for _rep in range(3):      # median of three batches (a one-off stall of a batch is not the number)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            a.record()
            for _ in range(n):
                fn()
            b.record()
            torch.cuda.synchronize(device)
            ts.append(a.elapsed_time(b) / n * 1e3)
        return sorted(ts)[1]   # us
    saved = _ops.ATTENTION_PRECISION
    out = {"config": f"self-attention B{B} H{H} S{S} head_dim {hd}, causal, relative logits, fp32 I/O",
           "flops_convention": "dense 2 S^2 hd per contraction: 3 forward, 7 backward (causal kernels skip half)"}
    with torch.no_grad():
        _ops.ATTENTION_PRECISION = "f32"
        ref = RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, 1, None)
    try:
        for prec in _ops.ATTENTION_PRECISIONS:
            _ops.ATTENTION_PRECISION = prec
            terms = _ops.ATTENTION_TERMS[prec]
            peak = BF16_MATRIX_PEAK_TFLOPS / terms if terms else FP32_MATRIX_PEAK_TFLOPS
            with torch.no_grad():
                t_f = timed(lambda: RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, 1, None))
                o = RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, 1, None)
            entry = {"fwd_us": round(t_f, 1), "fwd_TFLOPs_dense": round(3 * dense / t_f / 1e6, 1),
                     "peak_TFLOPs": round(peak, 1), "fwd_frac": round(3 * dense / t_f / 1e6 / peak, 4),
                     "max_err_vs_f32_over_max": float((o - ref).abs().max() / ref.abs().max())}
            res = RelAttentionFn.apply(qkv, None, rel, H, 1, 1, S, 1, None)

            def bwd():
                qkv.grad = None
                rel.grad = None
                res.backward(w, retain_graph=True)
            t_b = timed(bwd)
            entry.update(bwd_us=round(t_b, 1), bwd_TFLOPs_dense=round(7 * dense / t_b / 1e6, 1),
                         bwd_frac=round(7 * dense / t_b / 1e6 / peak, 4))
            out[prec] = entry
    finally:
        _ops.ATTENTION_PRECISION = saved
    out["default_precision"] = saved
    out["backward_logits"] = ("kept by the forward (isi_attn_args.logits), read by the backward"
                              if _ops.SAVE_ATTENTION_LOGITS else "recomputed by the backward")
    out["H16_hd32"] = _attention_row(device, B, 16, S, 32, n)
    dflt = out[saved]
    out["roofline"] = {"bound": "mfma", "kernel": f"rel_attention forward ({saved})", "achieved": dflt["fwd_TFLOPs_dense"],
                       "peak": dflt["peak_TFLOPs"], "unit": "TFLOP/s", "frac": dflt["fwd_frac"], "traffic": None,
                       "frac_of_2500_bf16_peak": round(dflt["fwd_TFLOPs_dense"] / BF16_MATRIX_PEAK_TFLOPS, 4)}
    return out


def _frontend(device, B=64):
    """Audio -> mel/IF spectrogram -> audio (n_fft 2048, hop 512, 4 s clips at 16 kHz): DFT and mel
    projections as fp32 GEMMs + HBM-bound polar / scan / transpose kernels."""
    from GANsynth_pytorch.spectrograms_helper import MelSpectrogramsHelper
    h = MelSpectrogramsHelper(16000, 2048, 512, 2048).to(device)
    x = torch.randn(B, 64000, device=device) * 0.1
    spec = h.to_spectrogram(x)
    out = {}
    for name, fn in (("to_spectrogram", lambda: h.to_spectrogram(x)), ("to_audio", lambda: h.to_audio(spec))):
        for _ in range(2):
            fn()
        torch.cuda.synchronize(device)
        t0 = time.perf_counter()
        for _ in range(5):
            fn()
        torch.cuda.synchronize(device)
        dt = (time.perf_counter() - t0) / 5
        out[name] = {"ms": round(dt * 1e3, 3), "clips_per_s": round(B / dt, 0)}
    out["config"] = f"B={B} clips x 64000 samples -> [B,2,1024,125] (mel), fp32"
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=64, help="spectrograms per GPU per step")
    ap.add_argument("--timer-stride", type=int, default=8,
                    help="per-kernel HIP-event timers on every n-th timed step (1: every step)")
    ap.add_argument("--spinup-ms", type=float, default=150.0,
                    help="untimed forwards before the warm-up steps until the clocks are steady")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-prior", action="store_true", help="skip the secondary prior-sampling metric")
    ap.add_argument("--no-train", action="store_true", help="skip the data-parallel VQ-VAE training leg")
    ap.add_argument("--prior-batch", type=int, default=8, help="codemaps per GPU and step of the prior's training legs")
    ap.add_argument("--prior-steps", type=int, default=20)
    ap.add_argument("--train-steps", type=int, default=20, help="timed steps of the VQ-VAE training leg")
    ap.add_argument("--prior-bottom-batch", type=int, default=2, help="codemaps per step of the bottom prior's training row")
    ap.add_argument("--prior-bottom-steps", type=int, default=8)
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("launch with: python -m torch.distributed.run --nproc-per-node N bench.py --gpus N")
        args.gpus = world
    # ISI_BENCH_BACKEND=gloo: dry run of the multi-rank code path on a box with fewer GPUs than ranks (ranks share
    # devices, collectives go through the host) -- a functional check only, never a measurement
    backend = os.environ.get("ISI_BENCH_BACKEND", "nccl")
    if backend != "nccl":
        local_rank %= max(1, torch.cuda.device_count())      # (device_count does not initialise the GPU)
    device = torch.device("cuda", local_rank)
    dist = None
    pg_before_gpu = None
    if world > 1:
        # the launcher path: the process group comes up BEFORE this process makes its first GPU call of its own (the
        # rendezvous must not depend on a device that another rank is still opening)
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        pg_before_gpu = not torch.cuda.is_initialized()
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=device)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)
    assert torch.cuda.is_available(), "bench.py needs an MI355X"
    torch.cuda.set_device(local_rank)
    if dist is not None:          # ... on EVERY rank
        flag = torch.tensor([1.0 if pg_before_gpu else 0.0], device=device if backend == "nccl" else "cpu")
        dist.all_reduce(flag, op=dist.ReduceOp.MIN)
        pg_before_gpu = bool(flag.item() == 1.0)

    from interactive_spectrogram_inpainting import _hip
    model, sd = _build_model(device)
    default_precision = model.conv_precision
    x = torch.randn(args.batch, 2, 128, 512, generator=torch.Generator().manual_seed(100 + rank)).to(device)
    L = _hip.lib()

    def barrier():
        torch.cuda.synchronize(device)
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize(device)

    with torch.no_grad():
        # device spin-up, untimed and independent of --warmup: from idle the part needs ~15 forwards (35 ms) to
        # reach its steady clock (first forwards 3.0 ms, steady 2.3 ms; tools/bench_ramp.py prints the ramp)
        t_spin = time.perf_counter()
        while time.perf_counter() - t_spin < args.spinup_ms * 1e-3:
            out = model(x)
            torch.cuda.synchronize(device)
        for _ in range(args.warmup):
            out = model(x)
        barrier()
        # kernel timers on every `--timer-stride`-th step of the timed region only: a timed launch carries a
        # start / stop event pair and costs ~4 us of stream time (4 % of a step if every launch is timed)
        # The library takes a launch's event pair from a pool that grows on demand: the pool is filled HERE, by as many
        # instrumented forwards as the timed region will sample -- created inside it, the ~50 events of a sampled forward
        # cost that step 0.4-0.7 ms of host time (the headline lost 3-9 % to it, box by box: 1.95 ms against the 1.77 ms of
        # the same loop without timers, `alt_precision_single_gpu.split_f16`).
        L.isi_prof_enable(1)
        for _ in range((args.steps + args.timer_stride - 1) // args.timer_stride):
            out = model(x)
        torch.cuda.synchronize(device)
        L.isi_prof_enable(1)          # records cleared, the pool stays
        L.isi_prof_enable(0)
        barrier()
        timed_steps = 0
        t0 = time.perf_counter()
        for i in range(args.steps):
            sampled = i % args.timer_stride == 0
            if sampled:
                L.isi_prof_enable(2)
                timed_steps += 1
            out = model(x)
            if sampled:
                L.isi_prof_enable(0)
        barrier()
        dt = time.perf_counter() - t0
    assert torch.isfinite(out[0]).all()

    if dist is not None:
        t = torch.tensor([dt], dtype=torch.float64, device=device)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        dt = t.item()

    # ---- BASELINE configs[2] on every rank: data-parallel VQ-VAE training step (collectives inside)
    train_leg = None
    if not args.no_train:
        try:
            train_leg = _vqvae_training(device, dist, world, batch=args.batch, steps=args.train_steps,
                                        warmup=min(3, args.train_steps))
        except Exception as e:      # a secondary leg must not take the headline metric with it ...
            if world > 1:           # ... but with collectives inside, a rank that skips the rest of the leg would leave
                raise               # the others waiting in an all-reduce until the RCCL timeout: let torchrun tear down
            train_leg = {"error": f"{type(e).__name__}: {e}"[:300]}

    # ---- BASELINE configs[3] on every rank when N > 1: data-parallel training step of the top prior (collectives inside)
    prior_dp_leg = None
    if not args.no_prior and not args.no_train and world > 1:
        prior_dp_leg = _prior_training(device, dist, world, B=args.prior_batch, steps=args.prior_steps,
                                       warmup=min(2, args.prior_steps))      # a failing rank tears the job down, as above

    # ---- per-kernel HIP-event timing recorded inside the timed region
    kernels = []
    for kid in range(L.isi_prof_num_kernels()):
        n, ms, fl, by = C.c_longlong(), C.c_double(), C.c_double(), C.c_double()
        _hip.check(L.isi_prof_read(kid, C.byref(n), C.byref(ms), C.byref(fl), C.byref(by)), "isi_prof_read")
        if n.value:
            kernels.append({"kernel": L.isi_prof_kernel_name(kid).decode(), "launches": n.value,
                            "ms": ms.value, "flops": fl.value, "bytes": by.value})
    # the forward's single dominant kernel has its own timer id; its launches also belong to the split-f16 convolution
    # family, whose line (`roofline`) is kept comparable with the earlier rounds
    single = next((k for k in kernels if k["kernel"].startswith("conv_pair_kernel<128, true, 0>")), None)
    if single is not None:
        fam = next((k for k in kernels if k["kernel"].startswith("conv_pair_kernel<..>")), None)
        kernels.remove(single)
        if fam is None:
            kernels.append(dict(single, kernel="conv_pair_kernel<..> + convT_pair_kernel<..> + conv_igemm_f32_kernel<..,f16x3>"))
        else:
            for key in ("launches", "ms", "flops", "bytes"):
                fam[key] += single[key]
    kernels.sort(key=lambda k: -k["ms"])

    if rank == 0:
        dom = kernels[0]
        # HBM bytes per launch of the dominant kernel from the committed PMC passes
        # (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE on this same command, FETCH doubled as the
        # gfx950 guide prescribes; tools/pmc_traffic.py): a measured constant, not re-measured here
        traffic = None
        traffic_file = next((f for f in ("profiles/r06_pmc_hbm_traffic.json", "profiles/r05_pmc_hbm_traffic.json", "profiles/r04_pmc_hbm_traffic.json", "profiles/r03_pmc_hbm_traffic.json", "profiles/r02_pmc_hbm_traffic.json",
                                         "profiles/r01_pmc_hbm_traffic.json") if (ROOT / f).exists()), "profiles/none")
        try:
            if args.batch != 64:
                raise ValueError("the committed PMC pass was taken at batch 64")
            pmc = json.load(open(ROOT / traffic_file))
            # rocprof names of the launches the dominant profiling id covers (template tail = MODE, PREC)
            # (f16x3: 3 = pieces of the weights computed while staging, 4 = prepared at pack time, ISI_CONV_W16)
            # (..., MODE, PREC, OUTP>: 4 = f16x3 with pack-time weight pieces, OUTP = output written as pairs)
            tail = {"bf16x6": (", 0, 2, ",), "bf16x3": (", 0, 1, ",), "f16x3": (", 0, 3, ", ", 0, 4, ")}
            want = next((t for k, t in tail.items() if k in dom["kernel"]), ("<128, 128, 2, 2, 0, 0, ",))
            tot_b = tot_n = 0.0
            for name, d in pmc.items():
                mine = ("conv_igemm_f32_kernel" in name and any(t in name for t in want)) or \
                       ("f16x3" in dom["kernel"] and ("conv_pair_kernel<" in name or "convT_pair_kernel<" in name))
                if mine and d.get("hbm_bytes_per_launch_corrected") is not None:    # (None: the two passes disagree on it)
                    n_d = d.get("dispatches", d.get("dispatches_fetch_pass", 0))
                    tot_b += d["hbm_bytes_per_launch_corrected"] * n_d
                    tot_n += n_d
            if tot_n:
                traffic = round(tot_b / tot_n)
        except (OSError, ValueError, KeyError):
            pass
        achieved = dom["flops"] / (dom["ms"] * 1e-3) / 1e12
        # a split kernel spends three (six) bf16 / f16 MFMA products per algorithmic product; both 16-bit types run
        # at the same dense matrix rate
        terms = 6.0 if "bf16x6" in dom["kernel"] else 3.0 if "f16x3" in dom["kernel"] else 0.0
        peak = BF16_MATRIX_PEAK_TFLOPS / terms if terms else FP32_MATRIX_PEAK_TFLOPS
        roof = {"bound": "mfma", "kernel": dom["kernel"], "achieved": round(achieved, 2),
                "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(achieved / peak, 4), "traffic": traffic,
                "traffic_source": (f"{traffic_file} (committed rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this "
                                   f"command at batch 64, FETCH doubled per the gfx950 guide; not measured in this run)")
                                  if traffic is not None else None,
                "algorithmic_bytes_per_launch": round(dom["bytes"] / dom["launches"]),
                "avg_launch_us": round(dom["ms"] * 1e3 / dom["launches"], 2),
                "launches_per_step": dom["launches"] // timed_steps,
                "timed_launches": dom["launches"],
                "share_of_step_time": round((dom["ms"] / timed_steps) / (dt * 1e3 / args.steps), 3),
                "hbm_algorithmic_GBs": round(dom["bytes"] / (dom["ms"] * 1e-3) / 1e9, 1)}
        if single is not None and single["launches"]:
            s_ach = single["flops"] / (single["ms"] * 1e-3) / 1e12
            roof["dominant_single_kernel"] = {
                "kernel": single["kernel"], "achieved": round(s_ach, 2), "peak": round(peak, 1), "unit": "TFLOP/s",
                "frac": round(s_ach / peak, 4), "avg_launch_us": round(single["ms"] * 1e3 / single["launches"], 2),
                "launches_per_step": single["launches"] // timed_steps,
                "algorithmic_bytes_per_launch": round(single["bytes"] / single["launches"]),
                "share_of_step_time": round((single["ms"] / timed_steps) / (dt * 1e3 / args.steps), 3)}
        # the whole forward against BOTH ceilings ("report achieved = max(bytes / (t BW_peak), flops / (t FLOP_peak))
        # and both terms"): algorithmic FLOPs / bytes summed over the timed kernels of one step
        tot_fl = sum(k["flops"] for k in kernels) / timed_steps
        tot_by = sum(k["bytes"] for k in kernels) / timed_steps
        t_step = dt / args.steps
        roof["whole_forward"] = {
            "flops": round(tot_fl), "bytes": round(tot_by),
            "TFLOPs": round(tot_fl / t_step / 1e12, 1), "frac_of_mfma_ceiling": round(tot_fl / t_step / 1e12 / peak, 4),
            "GBs": round(tot_by / t_step / 1e9, 1), "frac_of_hbm_peak": round(tot_by / t_step / 1e9 / HBM_PEAK_GBS, 4),
            "max_frac": round(max(tot_fl / t_step / 1e12 / peak, tot_by / t_step / 1e9 / HBM_PEAK_GBS), 4)}
        # the same figures as flat keys (the driver's parsed record keeps only the scalar entries of `roofline`)
        for key in ("TFLOPs", "frac_of_mfma_ceiling", "GBs", "frac_of_hbm_peak", "max_frac"):
            roof["whole_forward_" + key] = roof["whole_forward"][key]
        line = {
            "metric": "spectrograms/sec VQ-VAE fwd+quantize @B64",
            "value": round(world * args.batch * args.steps / dt, 2),
            "unit": "spectrograms/s",
            "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "spinup_ms": args.spinup_ms,
            **({"process_group_before_first_gpu_call": pg_before_gpu} if world > 1 else {}),
            "ms_per_step": round(dt * 1e3 / args.steps, 3),
            "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": {"f32": "f32", "split_f16": "f32 (products split into two f16 pieces, three terms)"}.get(
                default_precision, "f32 (products split into bf16 pieces: x6 / x3)"),
            "precision": {"mode": default_precision,
                          "note": "fp32 data and fp32 accumulation everywhere; split_f16 (default): every product as "
                                  "hi.hi+hi.lo+lo.hi of two 11-bit f16 pieces of the operands (scaled by powers of "
                                  "two) on the f16 matrix pipe -- per-product error ~2^-23, error vs fp64 at or below "
                                  "the fp32 pipe's own; codebook search: candidates on the f16 pipe, the decision "
                                  "between the two best in fp32 (index differences from the fp32 CPU reference are at "
                                  "the level of the reference's own rounding, DESIGN section 2); "
                                  "operand range |activation| < 16384, |weight| < 64 (weights checked at "
                                  "plan time, an activation beyond it gives NaN / index -1). split_bf16: six-term "
                                  "bf16 split (no range limit) in index-feeding layers, three-term in the final "
                                  "decoder. The other modes are timed in alt_precision_single_gpu"},
            "data": "synthetic",
            "config": {"workload": "VQVAE.forward (encode + quantize x2 + decode), eval, default ctor "
                                   "(128 hidden, 2 res blocks, D=64, K=512, factors 4/2)",
                       "batch_per_gpu": args.batch, "input": "[2,128,512] fp32 randn",
                       "weights": "random init, codebooks calibrated on encoder outputs",
                       "parallelism": f"replicas x{world} (no data-path collective)"},
            "roofline": roof,
            "kernel_timers": f"HIP events on every launch of {timed_steps} of the {args.steps} timed steps (every "
                             f"{args.timer_stride}th)",
            "kernels": [{"kernel": k["kernel"], "launches_per_step": k["launches"] // timed_steps,
                         "ms_per_step": round(k["ms"] / timed_steps, 4),
                         "TFLOPs": round(k["flops"] / (k["ms"] * 1e-3) / 1e12, 2),
                         "GBs": round(k["bytes"] / (k["ms"] * 1e-3) / 1e9, 1)} for k in kernels],
        }
        # the other precision modes, each compared with the all-fp32 outputs: 'f32' = every product exact,
        # 'bf16x3' = every convolution split-bf16 (may move near-tie indices: not parity-safe, never `value`)
        alt = {}
        with torch.no_grad():
            model.conv_precision = "f32"
            ref_out = [o.clone() for o in model(x)]
            for mode in ("f32", "bf16x3_decoder", "split_bf16", "bf16x3", "split_f16"):
                model.conv_precision = mode
                for _ in range(5):
                    o = model(x)
                torch.cuda.synchronize(device)
                t0 = time.perf_counter()
                for _ in range(10):
                    o = model(x)
                torch.cuda.synchronize(device)
                dt2 = (time.perf_counter() - t0) / 10
                # the decoder's own arithmetic error is measured TEACHER-FORCED on the fp32 mode's codes (a code that
                # moves -- only a float-rounding near-tie can, tests/test_hip_parity.py certifies each -- changes a whole
                # patch of the reconstruction and would hide it); `dec_max_err_over_max` is the end-to-end figure
                dec_tf = model.decode_code(ref_out[4], ref_out[5])
                alt[mode] = {"spectrograms_per_s": round(args.batch / dt2, 1), "ms_per_step": round(dt2 * 1e3, 3),
                             "dec_teacher_forced_max_err_over_max": float((dec_tf - ref_out[0]).abs().max() / ref_out[0].abs().max()),
                             "dec_max_err_over_max": float((o[0] - ref_out[0]).abs().max() / ref_out[0].abs().max()),
                             "codes_moved": {"top": int((o[4] != ref_out[4]).sum()), "of_top": o[4].numel(),
                                             "bottom": int((o[5] != ref_out[5]).sum()), "of_bottom": o[5].numel()},
                             "id_t_agreement": float((o[4] == ref_out[4]).float().mean()),
                             "id_b_agreement": float((o[5] == ref_out[5]).float().mean())}
            model.conv_precision = default_precision
        line["alt_precision_single_gpu"] = alt
        if train_leg is not None:
            line["vqvae_training_single_gpu" if world == 1 else "vqvae_training_dp"] = train_leg
        if prior_dp_leg is not None:
            line["prior_training_dp"] = prior_dp_leg
        # secondary legs and the CPU baseline at N = 1 only: at N > 1 the other ranks wait in the final barrier
        def leg(name, fn):          # a secondary leg must not take the headline metric with it
            try:
                line[name] = fn()
            except Exception as e:
                line[name] = {"error": f"{type(e).__name__}: {e}"[:300]}
            torch.cuda.empty_cache()
        if not args.no_prior and world == 1:
            leg("attention", lambda: _attention(device))
            leg("frontend", lambda: _frontend(device))
            leg("prior_sampling", lambda: _prior_sampling(device))
            if "error" not in line["prior_sampling"]:
                try:
                    line["prior_sampling"]["timerange_change"] = _timerange_change(device, model)
                except Exception as e:
                    line["prior_sampling"]["timerange_change"] = {"error": f"{type(e).__name__}: {e}"[:300]}
                torch.cuda.empty_cache()
            leg("prior_training_single_gpu", lambda: _prior_training(device, B=args.prior_batch, steps=args.prior_steps,
                                                                    warmup=min(2, args.prior_steps)))
            if args.prior_bottom_steps > 0:
                leg("prior_training_bottom", lambda: _prior_training(device, B=args.prior_bottom_batch, steps=args.prior_bottom_steps,
                                                                    warmup=min(2, args.prior_bottom_steps), level="bottom"))
        if not args.no_cpu_baseline and world == 1:
            # the timed model's own codes of the bench batch go to the oracle (all samples, level by level)
            with torch.no_grad():
                chk = model(x)
            leg("cpu_baseline", lambda: _cpu_baseline(sd, x.cpu(), chk[4].cpu(), chk[5].cpu(), batch=args.batch))
            cb = line["cpu_baseline"]
            if "gpu_codes_vs_oracle" in cb:     # north_star: bit-exact code indices (or certified near-ties of the reference's rounding)
                g = cb.pop("gpu_codes_vs_oracle")
                line["codes_moved_vs_oracle"] = {"top": g["top_moved"], "bottom_teacher_forced": g["bottom_moved_teacher_forced"],
                                                 "of_top": g["of_top"], "of_bottom": g["of_bottom"], "mode": default_precision,
                                                 "samples": g["samples"]}
                line["certified_near_ties"] = g["certified_near_ties"]
                line["codes_largest_normalised_gap"] = g["largest_normalised_gap"]
        print(json.dumps(line), flush=True)
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
