"""Decode knobs of the five scripted state-machine cases of tests/golden/sampler_script.npz (oracle/make_golden.py::script_knobs)."""
SCRIPT_KNOBS = {
    "greedy_cfg": dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=3, aug_text=True),
    "topk_topp": dict(top_k=12, top_p=0.8, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=3, aug_text=True),
    "topp_temp": dict(top_k=0, top_p=0.7, temperature=2.0, stop_repetition=2, cfg_coef=1.3, cfg_stride=1, aug_text=True),
    "silence": dict(top_k=1, top_p=1.0, temperature=1.0, stop_repetition=1, cfg_coef=1.5, cfg_stride=1, aug_text=True),
    "nocfg_topk": dict(top_k=5, top_p=0.95, temperature=1.0, stop_repetition=2, cfg_coef=1.5, cfg_stride=3, aug_text=False),
}
