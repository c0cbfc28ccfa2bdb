"""Summarise a rocprofv3 --pmc counter_collection CSV: average counter value per kernel launch, converted to bytes with the
gfx950 corrections of /opt/skills/guides/MI355X_MICROARCH.md §HBM (FETCH_SIZE/WRITE_SIZE are in KiB; FETCH_SIZE reports
exactly half of the bytes of a wide coalesced streaming read -> doubled).
Usage: python tools/pmc_summary.py <counter_collection.csv> [out.md]"""
import csv
import sys
from collections import defaultdict


def short(name):
    for k in ("resblock_split_dma_kernel", "lstm_step_split_kernel", "gemv_pair_merge_kernel", "gemv_pair_kernel", "gemv_segu_kernel", "gemv_rows_xreg_kernel", "gemv_rows_stream_kernel", "attn_rows_kernel", "attn_prefill_kernel", "conv_one_out_mfma_kernel", "conv_few_out_kernel", "conv_cin1_vec_kernel",
              "resblock64_kernel", "resblock_chain_split_kernel", "resblock_chain_kernel", "lstm_step_wide_kernel",
              "gemv_seg_kernel", "gemv_fast_kernel", "gemv_mfma_kernel", "gemv_kernel", "attn_decode_kernel", "attn_combine_kernel", "sample_kernel", "gemm_split_dma_kernel", "gemm_split_kernel", "gemm_kernel",
              "lstm_step_mfma_kernel", "lstm_step_kernel", "rvq_encode_mfma_kernel", "rvq_encode_kernel"):
        if k in name:
            tail = name[name.index(k) + len(k):]
            return k + (tail.split("(")[0] if tail.startswith("<") else "")
    return None


def main(path, out=None):
    agg = defaultdict(lambda: defaultdict(list))
    for r in csv.DictReader(open(path)):
        n = short(r["Kernel_Name"])
        if n is None:
            continue
        key = (n, int(r["Grid_Size"]) // max(int(r["Workgroup_Size"]), 1) if "Grid_Size" in r else 0)
        agg[key][r["Counter_Name"]].append(float(r["Counter_Value"]))
    lines = ["| kernel | workgroups | counter | launches | avg value | avg bytes/launch (gfx950-corrected) |", "|---|---|---|---|---|---|"]
    for key in sorted(agg, key=lambda k: (k[0], k[1])):
        for cn, vs in agg[key].items():
            avg = sum(vs) / len(vs)
            if cn == "FETCH_SIZE":
                b = avg * 1024 * 2
            elif cn == "WRITE_SIZE":
                b = avg * 1024
            else:
                b = float("nan")
            tail = f"{b / 1e6:.2f} MB" if b == b else "(not a byte counter)"
            lines.append(f"| {key[0]} | {key[1]} | {cn} | {len(vs)} | {avg:.1f} | {tail} |")
    txt = "\n".join(lines)
    print(txt)
    if out:
        open(out, "w").write(txt + "\n")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
