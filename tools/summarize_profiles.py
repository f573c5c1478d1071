#!/usr/bin/env python3
"""Turn gpurun_out/prof_<tag>/ (tools/profile_gpu.sh) into the committed summaries under profiles/:

  profiles/<round>_<tag>_kernel_stats.csv   rocprofv3 --kernel-trace --stats summary, verbatim
  profiles/<round>_<tag>_traffic.json       per-kernel HBM bytes per launch from the FETCH_SIZE and
                                            WRITE_SIZE passes

Counter handling follows /opt/skills/guides/MI355X_MICROARCH.md (HBM section): the two counters are
collected in separate passes; they are reported in KiB-like units of the L2's memory-side requests and
are uncalibrated for 8-byte-per-lane accesses, so each is scaled by the factor that makes the library's
streaming copy kernel (k_copy_probe, exactly 8*n bytes read and 8*n written per launch, n*8 well above
the 256 MB Infinity Cache only for the large workloads) come out right.  The factors are stored in the
JSON next to the raw averages.

usage: tools/summarize_profiles.py <round> <tag> <workload> <copy_bytes_read_per_launch>
"""
import csv
import glob
import json
import os
import shutil
import sys
from collections import defaultdict

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


def counter_avgs(d, counter):
    files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
    acc = defaultdict(lambda: [0.0, 0])
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = r["Kernel_Name"].split("(")[0]
            acc[k][0] += float(r["Counter_Value"])
            acc[k][1] += 1
    # one row per dispatch and counter dimension instance: normalise by dispatches
    disp = defaultdict(set)
    for f in files:
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") == counter:
                disp[r["Kernel_Name"].split("(")[0]].add(r["Dispatch_Id"])
    return {k: v[0] / max(1, len(disp[k])) for k, v in acc.items()}


def main():
    rnd, tag, workload, copy_bytes = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
    dst = os.path.join(ROOT, "profiles")
    os.makedirs(dst, exist_ok=True)
    for f in glob.glob(os.path.join(src, "trace", "**", "*kernel_stats.csv"), recursive=True):
        shutil.copy(f, os.path.join(dst, f"{rnd}_{tag}_kernel_stats.csv"))
    fetch = counter_avgs(os.path.join(src, "fetch"), "FETCH_SIZE")
    write = counter_avgs(os.path.join(src, "write"), "WRITE_SIZE")
    out = {"workload": workload, "source": f"rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes), {tag}",
           "kernels": {}}
    cf = cw = None
    if "k_copy_probe" in fetch and fetch["k_copy_probe"] > 0:
        cf = copy_bytes / fetch["k_copy_probe"]
    if "k_copy_probe" in write and write["k_copy_probe"] > 0:
        cw = copy_bytes / write["k_copy_probe"]
    out["calibration"] = {"kernel": "k_copy_probe", "bytes_read_per_launch": copy_bytes,
                          "bytes_written_per_launch": copy_bytes,
                          "FETCH_SIZE_raw": fetch.get("k_copy_probe"), "WRITE_SIZE_raw": write.get("k_copy_probe"),
                          "bytes_per_FETCH_SIZE_unit": cf, "bytes_per_WRITE_SIZE_unit": cw}
    for k in sorted(set(fetch) | set(write)):
        fr, wr = fetch.get(k), write.get(k)
        e = {"FETCH_SIZE_raw_per_launch": fr, "WRITE_SIZE_raw_per_launch": wr}
        if cf and cw and fr is not None and wr is not None:
            e["hbm_bytes_per_launch"] = fr * cf + wr * cw
        out["kernels"][k] = e
    with open(os.path.join(dst, f"{rnd}_{tag}_traffic.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", os.path.join(dst, f"{rnd}_{tag}_traffic.json"))


if __name__ == "__main__":
    main()
