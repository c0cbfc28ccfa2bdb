"""TEST INFRASTRUCTURE (build container only): extract the reference's argparse surfaces as data fixtures.

  tests/golden/cli_flags.json     <- /root/reference/inference_v2.py   (`parser.add_argument(...)` calls, :158-188)
  tests/golden/encode_flags.json  <- /root/reference/data/encode.py    (:5-19)

Only the flag table is recorded ({flag, type, default, store_true, choices}); nothing of the reference's code is copied.
Run: python -m oracle.make_golden_flags"""
import ast
import json
import os

REF = "/root/reference"
GOLD = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden")


def flags_of(path):
    tree = ast.parse(open(path).read())
    out = []
    for node in ast.walk(tree):
        if isinstance(node, ast.Call) and isinstance(node.func, ast.Attribute) and node.func.attr == "add_argument":
            flag = ast.literal_eval(node.args[0])
            kw = {k.arg: k.value for k in node.keywords}
            rec = {"flag": flag, "store_true": False, "choices": None}
            if "action" in kw:
                rec["store_true"] = ast.literal_eval(kw["action"]) == "store_true"
            if "type" in kw:
                rec["type"] = kw["type"].id
            if "default" in kw:
                rec["default"] = str(ast.literal_eval(kw["default"]))
            if "choices" in kw:
                rec["choices"] = ast.literal_eval(kw["choices"])
            out.append((node.lineno, rec))
    return [r for _, r in sorted(out, key=lambda t: t[0])]


def main():
    for src, dst in (("inference_v2.py", "cli_flags.json"), ("data/encode.py", "encode_flags.json")):
        recs = flags_of(os.path.join(REF, src))
        with open(os.path.join(GOLD, dst), "w") as f:
            json.dump(recs, f, indent=1)
        print(f"  {dst}: {len(recs)} flags")


if __name__ == "__main__":
    main()
