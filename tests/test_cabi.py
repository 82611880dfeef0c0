"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports
every symbol include/isi_hip.h declares, struct layouts agree, the Python
modules mirror the reference's state_dict, and nothing computes off-GPU."""
import ctypes
import pathlib
import re

import numpy as np
import pytest
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _declared_functions():
    text = (ROOT / "include" / "isi_hip.h").read_text()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(isi_[a-zA-Z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    from interactive_spectrogram_inpainting import _hip
    lib = _hip.lib()
    names = _declared_functions()
    assert len(names) >= 15
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/isi_hip.h but not exported"
        assert n in _hip.SIGNATURES, f"{n} has no ctypes signature"
    assert set(_hip.SIGNATURES) == set(names)
    assert lib.isi_version().startswith(b"isi_hip gfx950")


def test_struct_layouts_match():
    from interactive_spectrogram_inpainting import _hip
    lib = _hip.lib()
    structs = [_hip.isi_src, _hip.isi_dst, _hip.isi_conv_w, _hip.isi_encoder_w, _hip.isi_decoder_w,
               _hip.isi_codebook_w, _hip.isi_vqvae_w, _hip.isi_vqvae_out]
    for i, st in enumerate(structs):
        assert lib.isi_abi_struct_bytes(i) == ctypes.sizeof(st)
    assert lib.isi_abi_struct_bytes(99) == 0


def test_host_side_argument_validation_needs_no_gpu():
    from interactive_spectrogram_inpainting import _hip
    lib = _hip.lib()
    assert lib.isi_packed_conv_weight_floats(128, 128, 3, 3) == 128 * 1152
    assert lib.isi_packed_conv_weight_floats(64, 2, 4, 4) == 64 * 32
    assert lib.isi_packed_convT_k4s2_weight_floats(64, 2) == 4 * 2 * 256
    assert lib.isi_vq_num_partials(262144) == 256
    assert lib.isi_vq_num_partials(100) == 1
    # null pointers are rejected before any launch
    assert lib.isi_vq_nearest_f32(None, None, None, None, None, None, None, 10, 64, 512, None) == -1
    assert b"null" in lib.isi_last_error()
    assert lib.isi_conv2d_f32(None, None, None, None, None, None, 1, 1, 1, 1, 1, 1, 1, 0, 0, None) == -1


def test_vqvae_state_dict_matches_reference(golden_dir):
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    z = np.load(golden_dir / "vqvae_default_tiny.npz")
    ref = {k[3:]: z[k].shape for k in z.files if k.startswith("w::")}
    m = VQVAE(in_channel=2)
    sd = m.state_dict()
    assert set(sd) == set(ref)
    for k, shp in ref.items():
        assert tuple(sd[k].shape) == shp, k
    assert sum(p.numel() for p in m.parameters()) == 1386818
    # loads the reference's own weights, including a DDP-style `module.` prefix
    m.load_state_dict({k: torch.from_numpy(z["w::" + k]) for k in ref})
    z8 = np.load(golden_dir / "vqvae_f8_f4.npz")
    m8 = VQVAE(in_channel=2, num_hidden_channels=16, n_res_block=1, num_residual_channels=8,
               embed_dim=8, num_embeddings=32, resolution_factors={"bottom": 8, "top": 4})
    assert set(m8.state_dict()) == {k[3:] for k in z8.files if k.startswith("w::")}


def test_instantiation_parameters_round_trip(tmp_path):
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    m = VQVAE(in_channel=2, num_hidden_channels=32, num_residual_channels=8, embed_dim=16,
              num_embeddings=64)
    pj, pw = tmp_path / "p.json", tmp_path / "w.pt"
    m.store_instantiation_parameters(pj)
    torch.save({"model": {"module." + k: v for k, v in m.state_dict().items()}}, pw)
    m2 = VQVAE.from_parameters_and_weights(pj, pw)
    for k, v in m.state_dict().items():
        assert torch.equal(v, m2.state_dict()[k])


def test_no_cpu_fallback():
    from interactive_spectrogram_inpainting import _hip
    from interactive_spectrogram_inpainting.vqvae.vqvae import VQVAE
    m = VQVAE(in_channel=2, num_hidden_channels=32, num_residual_channels=8, embed_dim=16,
              num_embeddings=64).eval()
    with pytest.raises(_hip.HipLibraryError):
        m(torch.randn(1, 2, 16, 16))
    with pytest.raises(_hip.HipLibraryError):
        m.decode_code(torch.zeros(1, 2, 2, dtype=torch.int64), torch.zeros(1, 4, 4, dtype=torch.int64))
    with pytest.raises(NotImplementedError):      # unusable in the reference itself (see encoder_decoder._LOCAL_KERNELS)
        VQVAE(in_channel=2, use_local_kernels=True)
    g2 = VQVAE(in_channel=2, num_hidden_channels=32, num_residual_channels=8, embed_dim=16, num_embeddings=64, groups=2)
    assert g2.enc_b.blocks[0].weight.shape == (16, 1, 4, 4) and g2.dec.blocks[-1].weight.shape == (16, 1, 4, 4)
    dense = g2.enc_b.blocks[2].dense_weight()     # grouped conv = dense conv with the block-diagonal weight
    assert dense.shape == (32, 16, 4, 4) and not dense[:16, 8:].any() and not dense[16:, :8].any()
    assert torch.equal(g2.enc_b.blocks[2].grouped(dense), g2.enc_b.blocks[2].weight)


def test_missing_library_fails_loudly(monkeypatch, tmp_path):
    from interactive_spectrogram_inpainting import _hip
    monkeypatch.setenv("ISI_HIP_LIBRARY", str(tmp_path / "nope.so"))
    monkeypatch.setattr(_hip, "_lib", None)
    with pytest.raises(_hip.HipLibraryError):
        _hip.lib()


def test_oversize_and_bad_shapes_are_rejected_before_any_launch():
    """Limits are enforced on the host (no GPU needed): tensors of 4 GiB or more,
    codebooks that do not fit the kernel's tiling, bad quantiser shapes."""
    import ctypes as C
    from interactive_spectrogram_inpainting import _hip
    lib = _hip.lib()
    fake = 0x10000  # non-null, 16-byte aligned, never dereferenced on the host
    # 1 x 4 x 40000 x 40000 fp32 = 25.6 GB: beyond the 32-bit offset range of the kernels
    H = W = 40000
    src = _hip.isi_src(fake, 4, H * W * 4, 1, W * 4, 4)
    dst = _hip.isi_dst(fake, H * W * 4, 1, W * 4, 4)
    rc = lib.isi_conv2d_f32(C.byref(src), None, fake, None, None, C.byref(dst), 1, H, W, 4, 3, 3, 1, 1, 0, None)
    assert rc == -4 and b"4 GiB" in lib.isi_last_error()
    # more than 2^31 output pixels
    src = _hip.isi_src(fake, 4, 0, 1, 0, 4)
    rc = lib.isi_conv2d_f32(C.byref(src), None, fake, None, None, C.byref(dst), 4096, 1024, 1024, 4, 1, 1, 1, 0, 0, None)
    assert rc == -4
    # empty output
    src = _hip.isi_src(fake, 4, 16, 1, 8, 4)
    assert lib.isi_conv2d_f32(C.byref(src), None, fake, None, None, C.byref(dst), 1, 2, 2, 4, 4, 4, 2, 0, 0, None) == -1
    # quantiser: a codebook that does not fit in LDS, unsupported embed_dim, empty input
    assert lib.isi_vq_nearest_f32(fake, fake, fake, fake, fake, fake, fake, 100, 64, 1000, None) == -4
    assert lib.isi_vq_nearest_f32(fake, fake, fake, fake, fake, fake, fake, 100, 64, 0, None) == -1
    assert lib.isi_vq_nearest_f32(fake, fake, fake, fake, fake, fake, fake, 100, 24, 512, None) == -4
    assert lib.isi_vq_nearest_f32(fake, fake, fake, fake, fake, fake, fake, 0, 64, 512, None) == -1
    # fused residual block outside its range -> caller must compose two convolutions
    assert lib.isi_resblock_fusable(128, 32) == 1 and lib.isi_resblock_fusable(16, 8) == 0
    assert lib.isi_resblock_fusable(256, 32) == 0 and lib.isi_resblock_fusable(128, 64) == 0
    assert lib.isi_resblock_f32(fake, fake, fake, fake, fake, fake, 1, 4, 4, 16, 8, 0, None) == -4
    # sampler limits
    assert lib.isi_sample_row_f32(fake, 2048, 1, 2048, 1.0, 0, 0.0, fake, fake, None, None) == -4
    assert lib.isi_sample_row_f32(fake, 512, 1, 512, 0.0, 0, 0.0, fake, fake, None, None) == -1   # temperature 0
