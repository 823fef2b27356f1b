"""Rotary producer against tables recorded from the reference's ``RotaryEmbedding``
(lite_llama/models/rotary_embedding.py:34-137; generator tests/golden/gen_golden_rope.py): default and llama3-scaled
inverse frequencies bit-exact, fp16 cos / sin rows exact -- on the direct path and through the position-indexed
cache the decode step reads."""

import numpy as np
import pytest
import torch

from tests import _golden as G


@pytest.mark.parametrize("kind", ["default", "llama3"])
def test_rotary_tables_match_reference(kind):
    from lite_llama_amd.model import RotaryEmbedding, tiny_geometry

    d = np.load(G.GOLDEN_DIR + "/rotary_tables.npz")
    kw = {}
    if kind == "llama3":
        f, lo, hi, orig = d["llama3.scaling"].tolist()
        kw = dict(rope_type="llama3", rope_scaling=dict(factor=f, low_freq_factor=lo, high_freq_factor=hi,
                                                        original_max_position_embeddings=int(orig)))
    geo = tiny_geometry(hidden_size=4096, num_heads=32, num_kv_heads=8, head_dim=128, rope_theta=500000.0, **kw)
    rot = RotaryEmbedding(geo)
    assert np.array_equal(rot.inv_freq.numpy(), d[f"{kind}.inv_freq"])
    pos = torch.from_numpy(d["positions"])
    cos, sin = rot(torch.zeros(1, dtype=torch.float16), pos)
    assert np.array_equal(cos.numpy(), d[f"{kind}.cos"]) and np.array_equal(sin.numpy(), d[f"{kind}.sin"])
    # the decode-step form: rows of the position-indexed cache are the very same values
    rot.ensure(8200, "cpu")
    small = pos[:, pos[0] < 8200]
    tables = rot(torch.zeros(1, dtype=torch.float16), small.reshape(-1, 1))
    c2, s2 = tables.materialise()
    n = small.shape[1]
    assert np.array_equal(c2.reshape(n, -1).numpy(), d[f"{kind}.cos"][0, :n])
    assert np.array_equal(s2.reshape(n, -1).numpy(), d[f"{kind}.sin"][0, :n])
