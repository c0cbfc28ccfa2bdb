"""Scan the ISA of every kernel of the library for SERIALIZED global loads: a `global_load` / `buffer_load` (not an LDS-DMA) followed within
four instructions by `s_waitcnt vmcnt(0)` — one dependent memory round trip. A handful is normal (a prologue, a bias); dozens in one kernel
mean an epilogue or gather that compiled to load -> wait -> use -> store per element (round 4: 64 per lane in the residual-block kernel's
epilogue = 30 of a workgroup's 49 us). CPU only: hipcc cross-compiles gfx950.

    python tools/isa_scan.py [min_count]        # default 6
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "ssr-speech_amd", "csrc")


def serialized_loads(asm):
    """kernel symbol -> number of loads that are waited for (vmcnt(0)) within four instructions of their issue"""
    out, kern, last = {}, None, -99
    for n, line in enumerate(asm.split("\n")):
        m = re.match(r"^(_Z\w+):", line)
        if m:
            kern, last = m.group(1), -99
            continue
        t = line.strip()
        if kern is None or not t or t.startswith(";") or t.startswith("."):
            continue
        if (t.startswith("global_load") or t.startswith("buffer_load")) and " lds" not in t:
            last = n
        elif t.startswith("s_waitcnt") and "vmcnt(0)" in t and n - last <= 4:
            out[kern] = out.get(kern, 0) + 1
            last = -99
    return out


def main():
    lo = int(sys.argv[1]) if len(sys.argv) > 1 else 6
    with tempfile.TemporaryDirectory() as tmp:
        for f in sorted(os.listdir(CSRC)):
            if not f.endswith(".hip"):
                continue
            s = os.path.join(tmp, f[:-4] + ".s")
            subprocess.run(["hipcc", "-O3", "-std=c++17", "--offload-arch=gfx950", f"-I{ROOT}/include", f"-I{CSRC}", "-ffp-contract=off", "-S",
                            "--cuda-device-only", os.path.join(CSRC, f), "-o", s], check=True, capture_output=True)
            for k, v in sorted(serialized_loads(open(s).read()).items(), key=lambda kv: -kv[1]):
                if v >= lo:
                    print(f"{f:20s} {v:4d}  {k}")


if __name__ == "__main__":
    main()
