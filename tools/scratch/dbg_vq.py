import sys, pathlib
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo/interactive-spectrogram-inpainting_amd")
import numpy as np, torch
import test_hip_parity as T
gd = pathlib.Path("/root/repo/tests/golden")
for name in ["vqvae_default_tiny.npz", "vqvae_small.npz"]:
    z = np.load(gd / name)
    m = T._model_from_golden(z)
    x = torch.from_numpy(z["x"]).cuda()
    q_t, q_b, diff, id_t, id_b, p_t, p_b = m.encode(x)
    for nm, got, ref in (("t", id_t.cpu(), torch.from_numpy(z["id_t"])), ("b", id_b.cpu(), torch.from_numpy(z["id_b"]))):
        bad = (got != ref).nonzero()
        print(name, nm, "codes", got.numel(), "mismatches", bad.shape[0])
        for b in bad[:5]:
            print("   at", b.tolist(), "got", int(got[tuple(b)]), "ref", int(ref[tuple(b)]))
    # certify using the module's own pre-quantisation z: recompute via oracle on CPU
    from oracle import vqvae_oracle as O
    sd = {k: torch.from_numpy(z[k]) for k in z.files if k.startswith("sd.")} if any(k.startswith("sd.") for k in z.files) else None
    print("keys", [k for k in z.files][:12])
