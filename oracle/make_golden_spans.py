"""Fixture for the CLI's multi-span edit path: word-aligned edit intervals -> morphed spans -> mask_interval, recorded
from the REFERENCE's own statements. The reference keeps this logic inside `main()` (inference_v2.py:284-317: the `> 3 editings` check,
the nested `combine_spans`, the `morphed_span` / `mask_interval` assignments), so nothing can be imported: this script parses the
reference file, pulls exactly those statements out of `main`'s syntax tree and executes them on a list of cases. Only inputs and
outputs are stored (tests/golden/edit_spans.json). Run in the build container only (needs /root/reference):
  python oracle/make_golden_spans.py"""
import ast
import json
import os
from argparse import Namespace

import torch

SRC = "/root/reference/inference_v2.py"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "edit_spans.json")

CASES = [
    # (intervals in seconds, sub_amount, audio_dur, codec_sr)
    ([(1.00, 1.40)], 0.12, 7.93, 50),
    ([(0.05, 0.30)], 0.12, 7.93, 50),                       # start clipped at 0
    ([(7.60, 7.90)], 0.12, 7.93, 50),                       # end clipped at the file's length
    ([(1.00, 1.40), (3.00, 3.50)], 0.12, 7.93, 50),         # two edits, far apart
    ([(3.00, 3.50), (1.00, 1.40)], 0.12, 7.93, 50),         # given out of order
    ([(1.00, 1.40), (1.80, 2.10)], 0.12, 7.93, 50),         # 0.16 s apart after the margins: merged (threshold 0.2)
    ([(1.00, 1.40), (1.85, 2.10)], 0.12, 7.93, 50),         # 0.21 s apart: kept
    ([(1.00, 1.40), (1.84, 2.10)], 0.12, 7.93, 50),         # exactly at the threshold (>=)
    ([(0.50, 0.90), (2.00, 2.60), (5.00, 5.35)], 0.12, 7.93, 50),
    ([(0.50, 0.90), (1.20, 2.60), (2.70, 5.35)], 0.12, 7.93, 50),   # three edits that chain into one
    ([(0.50, 2.90), (1.00, 1.20), (4.00, 4.30)], 0.12, 7.93, 50),   # one edit inside another
    ([(1.013, 1.377)], 0.0, 7.93, 50),                      # rounding of seconds * codec_sr
    ([(1.01, 1.37), (2.49, 2.51)], 0.08, 5.0, 75),
    ([(0.1, 0.2), (0.5, 0.6), (1.0, 1.1), (2.0, 2.1)], 0.12, 7.93, 50),   # four edits: RuntimeError
]


def reference_statements():
    tree = ast.parse(open(SRC).read())
    main = next(n for n in tree.body if isinstance(n, ast.FunctionDef) and n.name == "main")
    found = {}
    for node in ast.walk(main):
        if isinstance(node, ast.FunctionDef) and node.name == "combine_spans":
            found["combine"] = node
        if isinstance(node, ast.If) and "len(orig_spans) > 3" in ast.unparse(node.test):
            found["too_many"] = node
        if isinstance(node, ast.Assign) and isinstance(node.targets[0], ast.Name):
            name, src = node.targets[0].id, ast.unparse(node)
            if name == "morphed_span" and "sub_amount" in src:
                found["grow"] = node
            if name == "morphed_span" and "combine_spans" in src:
                found["merge"] = node
            if name == "mask_interval" and "codec_sr" in src and "grow" in found and "frames" not in found:
                found["frames"] = node
            if name == "mask_interval" and "LongTensor" in src and "frames" in found and "tensor" not in found:
                found["tensor"] = node
    order = ["too_many", "combine", "grow", "merge", "frames", "tensor"]
    assert all(k in found for k in order), sorted(found)
    return ast.Module(body=[found[k] for k in order], type_ignores=[])


def run_reference(code, spans, sub_amount, audio_dur, codec_sr):
    ns = {"orig_spans": list(spans), "starting_intervals": [a for a, _ in spans], "ending_intervals": [b for _, b in spans],
          "args": Namespace(sub_amount=sub_amount, codec_sr=codec_sr), "audio_dur": audio_dur, "torch": torch}
    try:
        exec(code, ns)
    except RuntimeError as e:
        return {"error": str(e)}
    return {"morphed_span": [[float(a), float(b)] for a, b in ns["morphed_span"]], "mask_interval": ns["mask_interval"].tolist()}


if __name__ == "__main__":
    code = compile(ast.fix_missing_locations(reference_statements()), SRC, "exec")
    out = []
    for spans, sub, dur, sr in CASES:
        out.append({"spans": [list(s) for s in spans], "sub_amount": sub, "audio_dur": dur, "codec_sr": sr, "expect": run_reference(code, spans, sub, dur, sr)})
    json.dump({"source": "inference_v2.py:284-317 of the reference, executed statement by statement", "torch": torch.__version__, "cases": out},
              open(OUT, "w"), indent=1)
    print("wrote", OUT, len(out), "cases")
