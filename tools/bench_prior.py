#!/usr/bin/env python3
"""Prior sampling micro-benchmark (BASELINE metric 2): codes/s of `sample_model`
on the top prior, shape [32,32] (1024 tokens), B=1, full mask, d_model 512,
6 encoder + 8 decoder layers, 8 heads, random weights."""
import pathlib
import sys
import time

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT / "interactive-spectrogram-inpainting_amd"))
import torch  # noqa: E402
import sample as S  # noqa: E402
from interactive_spectrogram_inpainting.priors.transformer import SelfAttentiveVQTransformer  # noqa: E402


def build(device, shape=(32, 32)):
    torch.manual_seed(2)
    m = SelfAttentiveVQTransformer(
        shape=list(shape), condition_shape=list(shape), n_class=512, channel=256, kernel_size=5, n_block=4,
        n_res_block=4, res_channel=256, d_model=512, embeddings_dim=32, positional_embeddings_dim=16,
        use_relative_transformer=True, predict_frequencies_first=True, conditional_model=True,
        self_conditional_model=True, add_mask_token_to_symbols=True,
        class_conditioning_prepend_to_dummy_input=True,
        class_conditioning_num_classes_per_modality={"instrument_family_str": 11, "pitch": 61},
        class_conditioning_embedding_dim_per_modality={"instrument_family_str": 64, "pitch": 64})
    return m.to(device).eval()


def main():
    dev = torch.device("cuda:0")
    m = build(dev)
    cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
    for top_p in (0.0, 0.8):
        S.sample_model(m, dev, 1, [32, 32], 1.0, class_conditioning=cls, top_p_sampling_p=top_p,
                       generator=torch.Generator().manual_seed(0))  # warm-up
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        out = S.sample_model(m, dev, 1, [32, 32], 1.0, class_conditioning=cls, top_p_sampling_p=top_p,
                             generator=torch.Generator().manual_seed(1))
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(f"top_p={top_p}: {1024 / dt:8.1f} codes/s  ({dt * 1e3:.1f} ms per 1024-token codemap), "
              f"unique codes {out.unique().numel()}")
    # full teacher-forced forward (training-shaped pass) for the attention roofline
    code = torch.randint(0, 512, (4, 32, 32), device=dev)
    clsd = {k: v.expand(4).reshape(4, 1).to(dev) for k, v in cls.items()}
    src, tgt = m.to_sequences(code, code, class_conditioning=clsd)
    for _ in range(2):
        m(tgt, src)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        m(tgt, src)
    torch.cuda.synchronize()
    print(f"full forward B=4 S=1025: {(time.perf_counter() - t0) / 5 * 1e3:.2f} ms")


if __name__ == "__main__":
    main()
