"""CPU: the package's vectorised host integer code (ssr_speech_amd.layout) against the reference's
golden vectors and against the oracle."""
import glob
import os

import numpy as np
import pytest
import torch

import ssr_speech_amd  # noqa: F401
from ssr_speech_amd import layout as LY
from ssr_speech_amd import weights as W
from oracle import lm as O


def test_build_layout_matches_reference(golden_dir):
    g = np.load(os.path.join(golden_dir, "layout.npz"))
    args = W.lm_args_tiny()
    for name in ("tts", "mid", "start", "end", "two", "three", "insert"):
        cated, mp, num_task, nmi = LY.build_layout(g[f"{name}_y"], g[f"{name}_mi"], args)
        assert np.array_equal(cated, g[f"{name}_cated"]), name
        assert mp == g[f"{name}_mask_position"].tolist(), name
        pat = LY.delay_pattern(g[f"{name}_span"], args.empty_token)
        assert np.array_equal(pat, g[f"{name}_pattern"])
        assert np.array_equal(LY.undelay(pat, args.empty_token), g[f"{name}_reverted"])


@pytest.mark.parametrize("name", ["tts_greedy_cfg5", "edit_mid_greedy", "edit_start_greedy", "edit_2span_greedy", "edit_3span_greedy"])
def test_assemble_matches_reference(golden_dir, name):
    g = np.load(os.path.join(golden_dir, f"lm_{name}.npz"))
    d, h, nl, v = (int(t) for t in g["cfg"])
    args = W.lm_args_tiny(d_model=d, nhead=h, layers=nl, vocab=v)
    sd = O.reference_params(W.lm_state_dict(args, seed=int(g["weight_seed"])))
    kw = {k[3:]: (g[k].tolist() if g[k].ndim else g[k].item()) for k in g.files if k.startswith("kw_")}
    trace = {}
    torch.manual_seed(int(g["torch_seed"]))
    O.inference(sd, args, torch.from_numpy(g["x"]), torch.from_numpy(g["y"]), torch.from_numpy(g["mask_interval"]), trace=trace, **kw)
    steps = torch.stack(trace["samples"]).numpy()          # [n_steps, K] post-state-machine samples
    y = g["y"][0].T
    cated, mp, num_task, nmi = LY.build_layout(y, g["mask_interval"][0], args)
    # split the step stream into spans at the all-eog rows
    ends = [i + 1 for i in range(len(steps)) if steps[i, -1] == args.eog]
    assert len(ends) == num_task
    spans = [steps[a:b] for a, b in zip([0] + ends[:-1], ends)]
    res, marks, masks, nmi_out = LY.assemble(y, spans, nmi, args)
    assert np.array_equal(res[None], g["res"])
    assert np.array_equal(marks[None], g["marks"])
    assert np.array_equal(np.asarray(masks), g["masks"])
    assert np.array_equal(np.asarray(nmi_out), g["non_mask_intervals"])
