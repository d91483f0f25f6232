"""Summarise rocprofv3 --pmc passes into profiles/<name>.json (per kernel: mean per launch; FETCH_SIZE / WRITE_SIZE in KB).

usage: python tools/pmc_summary.py OUT.json DIR_FETCH DIR_WRITE [note]
Each DIR is the -d output of one `rocprofv3 --pmc <COUNTER> --kernel-trace --output-format csv`
pass (separate passes, as the MI355X guide prescribes); every *counter_collection.csv below it is read.
"""
import csv
import glob
import json
import os
import re
import sys

SHORT = (("k_tower", "k_tower"), ("k_conv3x3_b<16", "k_conv3x3_b<16>"), ("k_conv3x3_b<128", "k_conv3x3_b<128>"),
         ("k_heads", "k_heads"), ("k_search_round", "k_search_round"), ("k_play_move", "k_play_move"),
         ("k_end_search", "k_end_search"), ("k_new_games", "k_new_games"), ("k_pack_samples", "k_pack_samples"),
         ("k_finalize", "k_finalize"), ("k_policy_fc", "k_policy_fc"), ("k_value_head", "k_value_head"),
         ("k_refill", "k_refill"), ("Cijk_", "policy_fc_gemm(hipBLASLt)"))


def short(name):
    for key, s in SHORT:
        if key in name:
            return s
    return re.sub(r"\(.*", "", name)[:60]


def read(d):
    acc = {}
    for path in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        with open(path, newline="") as f:
            for row in csv.DictReader(f):
                k = (short(row["Kernel_Name"]), row["Counter_Name"])
                a = acc.setdefault(k, [0.0, set()])
                a[0] += float(row["Counter_Value"])
                a[1].add(row.get("Dispatch_Id") or row.get("Correlation_Id"))
    return acc


def main():
    out, dirs, note = sys.argv[1], sys.argv[2:4], (sys.argv[4] if len(sys.argv) > 4 else "")
    kernels = {}
    for d in dirs:
        for (kname, counter), (total, ids) in read(d).items():
            e = kernels.setdefault(kname, {})
            # FETCH_SIZE / WRITE_SIZE count kilobytes; every other counter (SQ_*) is a plain count
            unit = "_KB" if counter in ("FETCH_SIZE", "WRITE_SIZE") else ""
            e["%s%s_mean_per_launch" % (counter, unit)] = total / max(len(ids), 1)
            e["launches_%s" % counter] = len(ids)
    # which build the passes profiled: sha256 of the in-tree HIP library (16 hex digits), as bench.py reports for the one it runs
    import hashlib
    lib = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "chinesechessai_amd", "csrc", "libxq_hip.so")
    build = hashlib.sha256(open(lib, "rb").read()).hexdigest()[:16] if os.path.exists(lib) else None
    json.dump({"note": note, "build_libxq_hip_sha16": build, "kernels": kernels}, open(out, "w"), indent=1)
    for k, v in sorted(kernels.items()):
        print(k, v)


if __name__ == "__main__":
    main()
