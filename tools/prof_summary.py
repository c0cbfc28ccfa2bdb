"""Summarise a rocprofv3 --kernel-trace CSV: per (kernel, grid) call count / avg / min duration, and the
busy fraction of a decode-only window. With --gemm-log <file> (the SSRHIP_GEMM_LOG of the same run: one "M N K batch gx gy gz split" line
per GEMM launch, in call order) every GEMM row also carries its problem size and its fp32-equivalent TFLOP/s = 2 M N K batch / avg
(launches are matched to log lines per launch grid, in order: streams may reorder launches of DIFFERENT grids, not of the same one).
Usage: python tools/prof_summary.py <kernel_trace.csv> [out.md] [--gemm-log gemm.log]"""
import csv
import sys
from collections import defaultdict


def short(name):
    for k in ("resblock_split_dma_kernel", "lstm_step_split_kernel", "gemv_pair_merge_kernel", "gemv_pair_kernel", "gemv_segu_kernel", "gemv_rows_xreg_kernel", "gemv_rows_stream_kernel", "attn_rows_kernel", "conv_one_out_mfma_kernel", "conv_few_out_kernel", "conv_cin1_vec_kernel", "lstm_step_wide_kernel", "resblock_chain_split_kernel", "resblock_chain_kernel", "attn_prefill_kernel", "gemv_seg_kernel", "gemv_fast_kernel", "gemv_mfma_kernel", "gemv_kernel", "resblock64_kernel", "lstm_step_mfma_kernel", "lstm_step_kernel", "rvq_encode_mfma_kernel", "conv_cin1_kernel", "attn_decode_kernel", "attn_combine_kernel", "sample_kernel", "embed_kernel", "gemm_split_dma_kernel", "gemm_split_kernel", "gemm_kernel", "layernorm_kernel", "kv_scatter_kernel"):
        if k in name:
            return k + (name[name.index(k) + len(k):].split("(")[0] if "<" in name else "")
    return "torch:" + name.split("<")[0].split("(")[0][-40:]


GEMM_KERNELS = ("gemm_split_dma_kernel", "gemm_split_kernel", "gemm_kernel")


def main(path, out=None, gemm_log=None):
    rows = list(csv.DictReader(open(path)))
    rows.sort(key=lambda r: int(r["Start_Timestamp"]))
    shapes = []
    if gemm_log:
        shapes = [tuple(int(v) for v in ln.split()) for ln in open(gemm_log) if ln.strip()]
    gemm_rows = [r for r in rows if any(k in r["Kernel_Name"] for k in GEMM_KERNELS)]
    shape_of = {}
    if shapes:
        by_grid_log, by_grid_trace = defaultdict(list), defaultdict(list)
        for sh in shapes:
            by_grid_log[(sh[4], sh[5], sh[6], sh[7])].append(sh[:4])
        for r in gemm_rows:
            g = (int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r.get("Grid_Size_Z", 1) or 1), 1 if "gemm_split" in r["Kernel_Name"] else 0)
            by_grid_trace[g].append(r)
        for g, rs in by_grid_trace.items():
            if len(by_grid_log.get(g, [])) == len(rs):
                for r, sh in zip(rs, by_grid_log[g]):
                    shape_of[id(r)] = sh
        if len(shape_of) != len(gemm_rows):
            print(f"({len(gemm_rows) - len(shape_of)} of {len(gemm_rows)} GEMM launches not matched to a log line)")
    agg = defaultdict(list)
    for r in rows:
        d = int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        key = (short(r["Kernel_Name"]), int(r["Grid_Size_X"]) // max(int(r["Workgroup_Size_X"]), 1), int(r["Grid_Size_Y"]), int(r.get("Grid_Size_Z", 1) or 1),
               int(r["LDS_Block_Size"]), shape_of.get(id(r)))
        agg[key].append(d)
    lines = ["| kernel | blocks_x | grid_y | grid_z | lds | M x N x K x batch | calls | avg us | min us | total ms | TFLOP/s (fp32-equivalent) |", "|---|---|---|---|---|---|---|---|---|---|---|"]
    for key, ds in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        if key[0].startswith("torch:") and sum(ds) < 2e6:
            continue
        sh = key[5]
        avg = sum(ds) / len(ds)
        tf = f"{2.0 * sh[0] * sh[1] * sh[2] * sh[3] / avg / 1e3:.1f}" if sh else ""
        lines.append(f"| {key[0]} | {key[1]} | {key[2]} | {key[3]} | {key[4]} | {' x '.join(str(v) for v in sh) if sh else ''} | {len(ds)} | {avg / 1e3:.2f} | {min(ds) / 1e3:.2f} | {sum(ds) / 1e6:.2f} | {tf} |")
    ours = [r for r in rows if "ssrhip" in r["Kernel_Name"] or "anonymous" in r["Kernel_Name"]]
    ours.sort(key=lambda r: int(r["Start_Timestamp"]))
    dec = [r for r in ours if "gemm_kernel" not in r["Kernel_Name"]]
    # a window of 100 consecutive graph-replayed decode steps out of the timed region: from one sampler launch to the
    # sampler launch 100 steps later, picked where the gaps are smallest (the eager / per-slot timing passes have large gaps)
    samp = [i for i, r in enumerate(dec) if "sample_kernel" in r["Kernel_Name"]]
    best = None
    # only windows of WHOLE steps: the per-category timing passes of bench.py replay graphs that hold one kernel category (e.g. 50 sampler
    # launches back to back) — such a window has far fewer launches than 100 steps; the whole-step windows share the most common count
    counts = {}
    for j in range(0, max(len(samp) - 100, 0), 10):
        counts[samp[j + 100] - samp[j]] = counts.get(samp[j + 100] - samp[j], 0) + 1
    full = max(counts, key=lambda k: (counts[k], k)) if counts else 0
    for j in range(0, max(len(samp) - 100, 0), 10):
        i0, i1 = samp[j], samp[j + 100]
        if i1 - i0 != full:
            continue
        span = int(dec[i1]["End_Timestamp"]) - int(dec[i0]["End_Timestamp"])
        if best is None or span < best[0]:
            best = (span, i0, i1)
    if best:
        span, i0, i1 = best
        win = dec[i0 + 1: i1 + 1]
        busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in win)
        lines.append("")
        lines.append(f"100 consecutive decode steps ({len(win)} launches): span {span / 1e6:.3f} ms = {span / 1e5:.2f} us/step, kernel-busy {busy / 1e6:.3f} ms "
                     f"({100 * busy / span:.1f} %), avg gap {(span - busy) / len(win) / 1e3:.2f} us")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__" and "--gemm-log" in sys.argv:
    i = sys.argv.index("--gemm-log")
    _gl = sys.argv[i + 1]
    del sys.argv[i:i + 2]
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None, _gl)
    sys.exit(0)
if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
