"""CPU: the product path must never route through the oracle or a CPU fallback: no file of the package imports, opens or
executes anything under oracle/, and the compute entry points fail loudly without the HIP library / a GPU."""
import os
import re

import pytest
import torch

import ssr_speech_amd  # noqa: F401

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(ROOT, "ssr-speech_amd")


def test_package_sources_never_touch_the_oracle():
    bad = []
    for dp, _, fs in os.walk(PKG):
        for f in fs:
            if f.endswith((".py", ".hip", ".h", ".cpp")):
                txt = open(os.path.join(dp, f), errors="ignore").read()
                for m in re.finditer(r"^\s*(from|import)\s+oracle\b|oracle[/.](lm|codec|ref_import|make_golden)|/root/reference", txt, re.M):
                    # doc strings may *mention* that the CPU restatement lives in oracle/: only code references count
                    line = txt[txt.rfind("\n", 0, m.start()) + 1: txt.find("\n", m.end())]
                    if re.match(r"\s*(from|import)\s", line) or "open(" in line or "sys.path" in line:
                        bad.append((f, line.strip()))
    assert not bad, bad


def test_inference_refuses_to_run_without_a_gpu():
    if torch.cuda.is_available():
        pytest.skip("needs a CPU-only box")
    from ssr_speech_amd import weights as W
    from ssr_speech_amd.models.ssr import SSR_Speech
    from ssr_speech_amd.codec.wmencodec import WMEncodecModel
    m = SSR_Speech(W.lm_args_tiny())
    x = torch.zeros(1, 5, dtype=torch.long)
    y = torch.zeros(1, 9, 4, dtype=torch.long)
    with pytest.raises(RuntimeError, match="GPU"):
        m.inference(x, torch.LongTensor([5]), x, torch.LongTensor([5]), y, y, torch.LongTensor([[[9, 9]]]))
    cfg = W.codec_config_tiny()
    with pytest.raises(RuntimeError, match="GPU"):
        WMEncodecModel(cfg, W.codec_state_dict(cfg), "cpu")


def test_missing_library_is_a_loud_error(monkeypatch, tmp_path):
    from ssr_speech_amd import _lib
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", str(tmp_path / "nope.so"))
    with pytest.raises(_lib.SsrHipUnavailable):
        _lib.lib()
