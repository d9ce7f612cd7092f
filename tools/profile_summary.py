#!/usr/bin/env python3
"""Reduce rocprofv3 outputs of `bench.py --no-secondary --no-cpu-baseline` to the small files kept under profiles/.

  python tools/profile_summary.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> <bench_json> <out_dir>

- <stats_dir>: output directory of `rocprofv3 --kernel-trace --stats`  -> <tag>_bench_kernel_stats.csv
- <pmc_*_dir>: output directories of two SEPARATE `rocprofv3 --pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
  -> <tag>_pmc_FETCH_SIZE.csv, <tag>_pmc_WRITE_SIZE.csv (per launch of the factor kernel) and <tag>_pmc_summary.json
Counter units are KB (MI355X_MICROARCH.md, HBM / rocprofv3 section).
"""
import csv
import glob
import json
import os
import shutil
import sys

KERNEL = "slip_factor_kernel"


def find(d, suffix):
    hits = sorted(glob.glob(os.path.join(d, "**", "*" + suffix), recursive=True))
    return hits[0] if hits else None


def pmc_rows(d, counter):
    path = find(d, "counter_collection.csv")
    rows = []
    if path:
        for r in csv.DictReader(open(path)):
            if KERNEL in r.get("Kernel_Name", "") and r.get("Counter_Name") == counter:
                rows.append(dict(Kernel_Name=r["Kernel_Name"], Counter_Name=counter, Counter_Value=float(r["Counter_Value"]),
                                 Grid_Size=r.get("Grid_Size", ""), Workgroup_Size=r.get("Workgroup_Size", "")))
    return rows


def main():
    tag, stats_dir, fdir, wdir, bench_json, out = sys.argv[1:7]
    os.makedirs(out, exist_ok=True)
    ks = find(stats_dir, "kernel_stats.csv")
    if ks:
        shutil.copy(ks, os.path.join(out, f"{tag}_bench_kernel_stats.csv"))
    ds = find(stats_dir, "domain_stats.csv")
    if ds:
        shutil.copy(ds, os.path.join(out, f"{tag}_bench_domain_stats.csv"))
    summary = {}
    total = 0.0
    for counter, d in (("FETCH_SIZE", fdir), ("WRITE_SIZE", wdir)):
        rows = pmc_rows(d, counter)
        with open(os.path.join(out, f"{tag}_pmc_{counter}.csv"), "w") as f:
            w = csv.DictWriter(f, fieldnames=["Kernel_Name", "Counter_Name", "Counter_Value", "Grid_Size", "Workgroup_Size"])
            w.writeheader()
            for r in rows:
                w.writerow(r)
        vals = [r["Counter_Value"] for r in rows]
        mean = sum(vals) / len(vals) if vals else None
        # the first factorisation of a handle grows its slabs and relaunches: those partial launches are short;
        # the median is the full-window launch the bench times
        med = sorted(vals)[len(vals) // 2] if vals else None
        summary[counter] = dict(per_launch_values_KB=vals, mean_KB=mean, median_KB=med)
        total = total + med * 1024 if (med is not None and total is not None) else None
    summary["hbm_bytes_per_launch"] = total
    summary["command"] = ("rocprofv3 --pmc FETCH_SIZE|WRITE_SIZE --output-format csv -- python3 bench.py --steps 3 --warmup 1 "
                          "--no-cpu-baseline --no-secondary (two separate passes)")
    summary["note"] = ("Per launch of slip_factor_kernel (the grid of column workers) on the C4 window; counters are in KB. "
                       "MI355X_MICROARCH.md: FETCH_SIZE under-reports wide (16 B/lane) streaming reads by 2x; this kernel's reads "
                       "are 4-8 B per lane and uncalibrated, so the read side lies between FETCH_SIZE and 2*FETCH_SIZE. "
                       "The workers' frontier polls (sc1 loads of one line) and the release write-backs are included.")
    try:
        b = json.loads(open(bench_json).read().strip().splitlines()[-1])
        summary["algorithmic"] = dict(B_read=b["roofline"]["algorithmic_read_bytes"], B_write=b["roofline"]["algorithmic_write_bytes"])
        # bench.py quotes this profile's traffic only while its kernel time still matches the kernel being timed
        summary["kernel_ms_per_launch"] = b["roofline"]["kernel_ms_per_launch"]
        summary["workers"] = b["roofline"].get("workers"); summary["waves"] = b["roofline"].get("waves")
        shutil.copy(bench_json, os.path.join(out, f"{tag}_bench.json"))
    except Exception as e:          # the bench line is optional for this reduction
        summary["bench_json_error"] = str(e)
    json.dump(summary, open(os.path.join(out, f"{tag}_pmc_summary.json"), "w"), indent=1)
    print(json.dumps(summary)[:600])


if __name__ == "__main__":
    main()
