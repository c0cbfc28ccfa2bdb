"""CPU: small consistency checks between the sources and the documents a maintainer reads."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _read(*parts):
    with open(os.path.join(ROOT, *parts)) as f:
        return f.read()


def test_every_environment_switch_is_listed_in_integration_md():
    names = set()
    for fn in glob.glob(os.path.join(ROOT, "ssr-speech_amd", "csrc", "*.hip")) + glob.glob(os.path.join(ROOT, "ssr-speech_amd", "csrc", "*.h")):
        names |= set(re.findall(r'getenv(?:_flag)?\("(SSRHIP_[A-Z0-9_]+)"', open(fn).read()))
    for fn in glob.glob(os.path.join(ROOT, "ssr-speech_amd", "**", "*.py"), recursive=True):
        names |= set(re.findall(r'environ(?:\.get\(|\[)"(SSRHIP_[A-Z0-9_]+)"', open(fn).read()))
    assert len(names) >= 15, sorted(names)
    doc = _read("INTEGRATION.md")
    missing = sorted(n for n in names if n not in doc)
    assert not missing, f"environment switches read by the code but not documented in INTEGRATION.md §4: {missing}"


def test_profiles_named_in_bench_exist():
    src = _read("bench.py")
    for rel in set(re.findall(r"profiles/[A-Za-z0-9_./]+\.(?:md|json|csv)", src)):
        assert os.path.exists(os.path.join(ROOT, rel)), rel


def test_design_cites_existing_tools_and_tests():
    doc = _read("DESIGN.md")
    for rel in set(re.findall(r"`(tools/[A-Za-z0-9_]+\.(?:hip|py|sh))`", doc)):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
    for rel in set(re.findall(r"`(tests/[A-Za-z0-9_]+\.py)", doc)):
        assert os.path.exists(os.path.join(ROOT, rel)), rel
