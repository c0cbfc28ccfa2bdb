"""ORACLE tooling (build container only): import the REAL reference from /root/reference so the
restatements in oracle/ can be pinned against it and golden vectors generated.

Nothing here is used on the GPU box (the reference does not travel); nothing from the reference is
copied. Two training-only third-party modules that are not installed offline are stubbed:
`torchmetrics.classification.MulticlassAccuracy` (models/ssr.py:12,181-189 — training metric) and
`flashy.distrib.broadcast_tensors` (quantization/core_vq.py:140,158 — k-means init broadcast).
"""
import importlib
import importlib.util
import os
import sys
import types

import torch

REF = "/root/reference"


def available() -> bool:
    return os.path.isdir(os.path.join(REF, "models"))


def import_lm():
    """-> module `models.ssr` of the reference."""
    if "torchmetrics" not in sys.modules:
        tm = types.ModuleType("torchmetrics")
        tmc = types.ModuleType("torchmetrics.classification")

        class MulticlassAccuracy(torch.nn.Module):
            def __init__(self, *a, **k):
                super().__init__()

        tmc.MulticlassAccuracy = MulticlassAccuracy
        tm.classification = tmc
        sys.modules["torchmetrics"] = tm
        sys.modules["torchmetrics.classification"] = tmc
    if REF not in sys.path:
        sys.path.insert(0, REF)
    from models import ssr  # noqa

    return ssr


def import_codec():
    """-> (seanet module, quantization package, wmencodec module) of the reference's audiocraft,
    loaded file-by-file because `import audiocraft` needs xformers/omegaconf/flashy/julius/av."""
    if "flashy" not in sys.modules:
        fl = types.ModuleType("flashy")
        fl.distrib = types.SimpleNamespace(broadcast_tensors=lambda *a, **k: None)
        sys.modules["flashy"] = fl
    R = REF + "/audiocraft/audiocraft"
    for name, path in [("audiocraft", R), ("audiocraft.modules", R + "/modules"), ("audiocraft.models", R + "/models")]:
        if name not in sys.modules:
            p = types.ModuleType(name)
            p.__path__ = [path]
            sys.modules[name] = p

    def load(name, path):
        if name in sys.modules:
            return sys.modules[name]
        spec = importlib.util.spec_from_file_location(name, path)
        mod = importlib.util.module_from_spec(spec)
        sys.modules[name] = mod
        spec.loader.exec_module(mod)
        return mod

    load("audiocraft.modules.conv", R + "/modules/conv.py")
    load("audiocraft.modules.lstm", R + "/modules/lstm.py")
    seanet = load("audiocraft.modules.seanet", R + "/modules/seanet.py")
    qt = importlib.import_module("audiocraft.quantization")
    wm = load("audiocraft.models.wmencodec", R + "/models/wmencodec.py")
    return seanet, qt, wm
