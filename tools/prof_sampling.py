#!/usr/bin/env python3
"""One B = 1 codemap of the top prior (bench.py's sampling configuration) for `rocprofv3 --kernel-trace --stats`."""
import pathlib
import sys

sys.path.insert(0, str(pathlib.Path(__file__).resolve().parents[1]))
import torch  # noqa: E402

import bench  # noqa: E402
import sample as S  # noqa: E402

if __name__ == "__main__":
    dev = torch.device("cuda", 0)
    m = bench._top_prior(dev).eval()
    cls = {"pitch": torch.tensor([24]), "instrument_family_str": torch.tensor([0])}
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    S.sample_model(m, dev, B, [32, 32], 1.0, generator=torch.Generator().manual_seed(0), class_conditioning=cls,
                   top_p_sampling_p=0.8)
    torch.cuda.synchronize(dev)
