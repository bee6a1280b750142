#!/usr/bin/env python3
"""Condenses rocprofv3 output under gpurun_out/ into the tracked files under profiles/.

  profiles/<tag>_kernel_stats.csv  : --kernel-trace --stats summary, library kernels only
  profiles/traffic.json            : HBM bytes per launch of the dominant kernel from the PMC passes

FETCH_SIZE / WRITE_SIZE are in KiB.  On gfx950 FETCH_SIZE reports half the bytes of a streaming
read; the factor is CALIBRATED here on kernels of known byte counts in the same run
(morton_encode30: 12 B read per point; global_histogram: 4 B per key) and applied to the pass.
WRITE_SIZE is exact for streaming stores (checked on morton_encode30: 8 B per point).
"""
import argparse
import collections
import csv
import glob
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def counter_means(pattern, counter):
    out = collections.defaultdict(list)
    for path in glob.glob(pattern, recursive=True):
        for r in csv.DictReader(open(path)):
            if r["Counter_Name"] == counter:
                out[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in out.items()}


def pick(d, needle):
    for k, v in d.items():
        if needle in k:
            return v
    return None


def summarize_hybrid(args, n, library_rows):
    """Kernel statistics and per-launch HBM traffic of the MRS_ALGO_HYBRID schedule's kernels."""
    root = os.path.join(ROOT, args.hybrid_root)
    for d in sorted(glob.glob(os.path.join(root, "prof_*"))):
        if not os.path.isdir(d):
            continue
        name = os.path.basename(d)[len("prof_"):]
        for path in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
            with open(os.path.join(ROOT, "profiles", f"{args.tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
                csv.writer(f).writerows(library_rows(path))
    fetch = counter_means(os.path.join(root, "pmc_fetch", "**", "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_means(os.path.join(root, "pmc_write", "**", "*counter_collection.csv"), "WRITE_SIZE")
    if not fetch or not write:
        return
    cal_enc = 12.0 * n / (pick(fetch, "morton_encode30") * 1024.0)
    cal_hist = 4.0 * n / (pick(fetch, "hybrid_histogram_kernel") * 1024.0)
    cal_w = 8.0 * n / (pick(write, "morton_encode30") * 1024.0)
    factor = 0.5 * (cal_enc + cal_hist)

    def traffic(needle):
        f, w = pick(fetch, needle), pick(write, needle)
        return {"read_bytes_per_launch": f * 1024.0 * factor, "write_bytes_per_launch": w * 1024.0 * cal_w,
                "bytes_per_launch": f * 1024.0 * factor + w * 1024.0 * cal_w, "FETCH_SIZE_KiB": f, "WRITE_SIZE_KiB": w}

    # the last but one template argument of hybrid_pass_kernel names the launch: 0 pass A, 1 pass B
    a = traffic("1024, 32, 0, ")
    b = traffic("1024, 32, 1, ")
    entry = {
        "kernel": "hybrid_pass_kernel (mean of pass A and pass B)",
        "n": n,
        "bytes_per_launch": 0.5 * (a["bytes_per_launch"] + b["bytes_per_launch"]),
        "algorithmic_bytes_per_launch": 16.0 * n,
        "pass_a": a,
        "pass_b": b,
        "local_sort": traffic("hybrid_local_sort_kernel"),
        "histogram": traffic("hybrid_histogram_kernel"),
        "fetch_calibration": {"morton_encode30 (12 B/point)": cal_enc, "hybrid_histogram (4 B/key)": cal_hist, "applied": factor},
        "write_calibration": {"morton_encode30 (8 B/point)": cal_w},
        "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of tools/sort_once.py --algo hybrid --end-bit 30",
        "measured": args.measured,
    }
    path = os.path.join(ROOT, "profiles", "traffic.json")
    whole = json.load(open(path)) if os.path.exists(path) else {}
    whole["hybrid_pass"] = entry
    with open(path, "w") as f:
        json.dump(whole, f, indent=1)
    print(json.dumps(entry, indent=1))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--tag", default="r01")
    ap.add_argument("--stats-dir", default="gpurun_out/prof_r01b")
    ap.add_argument("--fetch-dir", default="gpurun_out/pmc_fetch")
    ap.add_argument("--write-dir", default="gpurun_out/pmc_write")
    ap.add_argument("--log2n", type=int, default=28)
    ap.add_argument("--root", default="", help="directory written by tools/collect_profiles.sh (gpurun_out/TAG): every "
                                               "prof_NAME below it becomes profiles/TAG_NAME_kernel_stats.csv, and its "
                                               "pmc_fetch / pmc_write feed traffic.json")
    ap.add_argument("--measured", default="", help="free text stored in traffic.json (date, commit)")
    ap.add_argument("--hybrid-root", default="", help="directory written by tools/collect_hybrid_profiles.sh: its prof_NAME "
                                                      "become profiles/TAG_NAME_kernel_stats.csv and its pmc_fetch / pmc_write "
                                                      "the `hybrid_pass` entry of traffic.json (the rest of the file is kept)")
    args = ap.parse_args()
    os.makedirs(os.path.join(ROOT, "profiles"), exist_ok=True)
    n = 1 << args.log2n

    def library_rows(path):
        rows = list(csv.reader(open(path)))
        # library kernels, RCCL's and the runtime's copy/fill kernels
        return [rows[0]] + [r for r in rows[1:] if "mrs::" in r[0] or "rocclr" in r[0] or "nccl" in r[0].lower()]

    if args.root:
        for d in sorted(glob.glob(os.path.join(ROOT, args.root, "prof_*"))):
            if not os.path.isdir(d):
                continue
            name = os.path.basename(d)[len("prof_"):]
            for path in glob.glob(os.path.join(d, "**", "*kernel_stats.csv"), recursive=True):
                with open(os.path.join(ROOT, "profiles", f"{args.tag}_{name}_kernel_stats.csv"), "w", newline="") as f:
                    csv.writer(f).writerows(library_rows(path))
        args.fetch_dir = os.path.join(args.root, "pmc_fetch")
        args.write_dir = os.path.join(args.root, "pmc_write")
    else:
        for path in glob.glob(os.path.join(ROOT, args.stats_dir, "**", "*kernel_stats.csv"), recursive=True):
            with open(os.path.join(ROOT, "profiles", f"{args.tag}_kernel_stats.csv"), "w", newline="") as f:
                csv.writer(f).writerows(library_rows(path))

    if args.hybrid_root:
        summarize_hybrid(args, n, library_rows)
        return

    fetch = counter_means(os.path.join(ROOT, args.fetch_dir, "**", "*counter_collection.csv"), "FETCH_SIZE")
    write = counter_means(os.path.join(ROOT, args.write_dir, "**", "*counter_collection.csv"), "WRITE_SIZE")
    if fetch and write:
        enc_f, hist_f = pick(fetch, "morton_encode30"), pick(fetch, "global_histogram_banked_kernel") or pick(fetch, "global_histogram_kernel")
        enc_w = pick(write, "morton_encode30")
        cal_enc = 12.0 * n / (enc_f * 1024.0)
        cal_hist = 4.0 * n / (hist_f * 1024.0)
        cal_w = 8.0 * n / (enc_w * 1024.0)
        factor = 0.5 * (cal_enc + cal_hist)
        pass_f, pass_w = pick(fetch, "digit_pass_kernel<1"), pick(write, "digit_pass_kernel<1")
        read_b = pass_f * 1024.0 * factor
        write_b = pass_w * 1024.0 * cal_w
        out = {
            "kernel": "digit_pass_kernel<onesweep>",
            "n": n,
            "bytes_per_launch": read_b + write_b,
            "read_bytes_per_launch": read_b,
            "write_bytes_per_launch": write_b,
            "algorithmic_bytes_per_launch": 16.0 * n,
            "FETCH_SIZE_KiB": pass_f,
            "WRITE_SIZE_KiB": pass_w,
            "fetch_calibration": {"morton_encode30 (12 B/point)": cal_enc, "global_histogram (4 B/key)": cal_hist,
                                  "applied": factor},
            "write_calibration": {"morton_encode30 (8 B/point)": cal_w},
            "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate runs of tools/sort_once.py",
            "measured": args.measured,
        }
        with open(os.path.join(ROOT, "profiles", "traffic.json"), "w") as f:
            json.dump(out, f, indent=1)
        print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
