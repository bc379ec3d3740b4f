"""CPU: the oracle (oracle/l4p_oracle.py) against the golden vectors produced by the REAL reference
(tools/gen_golden.py; tests/golden/*.npz).  Tolerance 1e-4 relative-to-max (measured <= 1e-6);
tracker labels / prompt labels / re-seeded query times bit-exact."""
import os

import numpy as np
import pytest
import torch

from l4p_amd.weights import ModelCfg, seeded_state_dict
from oracle.l4p_oracle import OracleModel, encoder_forward
from tests.golden_utils import make_batch, sample_indices

GOLD = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ALL = ["flow_2d_backward", "track_2d", "depth", "dyn_mask", "camray"]


@pytest.fixture(scope="module")
def mini():
    cfg = ModelCfg.mini()
    return cfg, seeded_state_dict(cfg)


def _cmp(name, y, g):
    y = y.detach().float().reshape(-1)
    g = torch.from_numpy(np.asarray(g)).float().reshape(-1)
    s = y[sample_indices(y.numel())] if y.numel() > 4096 else y
    assert s.shape == g.shape, (name, s.shape, g.shape)
    err = (s - g).abs().max() / (g.abs().max() + 1e-30)
    assert err <= 1e-4, (name, float(err))


@pytest.mark.parametrize("case,T,tasks,nq", [("mini_T16_all", 16, ALL, 8),
                                             ("mini_T32_stitch", 32, ["depth", "flow_2d_backward", "dyn_mask", "track_2d"], 12)])
def test_oracle_matches_reference_goldens(mini, case, T, tasks, nq):
    cfg, sd = mini
    gold = np.load(os.path.join(GOLD, case + ".npz"))
    batch = make_batch(T, nq)
    trace = []
    with torch.no_grad():
        out = OracleModel(sd, cfg, use_intrinsics=True).forward(batch, tasks, trace=trace)
        feats = encoder_forward(sd, batch["rgb_b3thw"][:, :, :16], cfg)
    for li in sorted(set([0, 1, cfg.depth] + list(cfg.hooks))):
        _cmp(f"feat{li}", feats[li], gold[f"feat{li}"])
    keys = [k for k in gold.files if not k.startswith("feat") and not k.startswith("trace")]
    assert sorted(keys) == sorted(out.keys())
    for k in keys:
        _cmp(k, out[k], gold[k])
    nwin = len([k for k in gold.files if k.endswith("_labels") and "prompt" not in k])
    assert nwin == len(trace) == (T - 16) // 8 + 1
    for w in range(nwin):
        assert np.array_equal(gold[f"trace{w}_labels"], trace[w]["labels"].float().numpy()), w
        assert np.array_equal(gold[f"trace{w}_prompt_labels"], trace[w]["prompt_labels"].float().numpy()), w
        assert np.array_equal(gold[f"trace{w}_queries"][:, 0], trace[w]["queries"][:, 0].numpy()), w
        _cmp(f"trace{w}_queries", trace[w]["queries"], gold[f"trace{w}_queries"])


def test_full_size_goldens_are_present_and_pinned():
    """The full-size (1408 x 40) fixtures were produced by the reference and checked against the oracle at
    generation time (oracle_vs_reference_full.json); re-running them on CPU takes minutes, so here only the
    recorded agreement is asserted.  The GPU suite compares the engine with these vectors."""
    import json

    rep = json.load(open(os.path.join(GOLD, "oracle_vs_reference_full.json")))["full_T16_all"]
    assert max(v["oracle_rel_err"] for v in rep["tensors"].values()) <= 1e-4
    assert rep.get("tracker_state_windows", 0) >= 1
    g = np.load(os.path.join(GOLD, "full_T16_all.npz"))
    assert "depth_est_b1thw" in g.files and "feat36" in g.files
