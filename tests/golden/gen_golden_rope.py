"""Rotary-table golden vectors from the REFERENCE rotary producer (build container only).

    python tests/golden/gen_golden_rope.py

``lite_llama/models/rotary_embedding.py:34-137``: default and llama3-scaled (Llama-3.1 style: factor 8, low / high
frequency factors 1 / 4, original context 8192, theta 5e5, head_dim 128) inverse frequencies, and the fp16 cos / sin
rows the reference hands out for a spread of positions (short, around the original context, far beyond it).
Saved: plain arrays only.
"""

import os
import sys

sys.path.insert(0, "/root/reference")

import numpy as np  # noqa: E402
import torch  # noqa: E402

HERE = os.path.dirname(os.path.abspath(__file__))


def main():
    from lite_llama.models.rotary_embedding import RotaryEmbedding

    positions = torch.tensor([[0, 1, 2, 17, 511, 2047, 8191, 8192, 20000, 65535, 131071]])
    out = {"positions": positions.numpy()}
    cfgs = {
        "default": dict(head_dim=128, hidden_size=4096, num_heads=32, rope_theta=500000.0, rope_type="default"),
        "llama3": dict(head_dim=128, hidden_size=4096, num_heads=32, rope_theta=500000.0, rope_type="llama3", factor=8.0,
                       low_freq_factor=1.0, high_freq_factor=4.0, original_max_position_embeddings=8192),
    }
    for name, cfg in cfgs.items():
        rot = RotaryEmbedding(cfg)
        cos, sin = rot(torch.zeros(1, dtype=torch.float16), positions)
        out[f"{name}.inv_freq"] = rot.inv_freq.numpy()
        out[f"{name}.cos"] = cos.numpy()
        out[f"{name}.sin"] = sin.numpy()
        out[f"{name}.scaling"] = np.array([cfg.get("factor", 0.0), cfg.get("low_freq_factor", 0.0),
                                           cfg.get("high_freq_factor", 0.0),
                                           cfg.get("original_max_position_embeddings", 0)], dtype=np.float64)
    np.savez(os.path.join(HERE, "rotary_tables.npz"), **out)
    print("rotary_tables.npz:", {k: v.shape for k, v in out.items()})


if __name__ == "__main__":
    main()
