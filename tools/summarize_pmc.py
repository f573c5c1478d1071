#!/usr/bin/env python3
"""gpurun_out/pmc_<tag>/ (tools/profile_pmc.sh) -> profiles/<round>_<tag>_traffic.json.

Per kernel: the average FETCH_SIZE and WRITE_SIZE per launch (separate rocprofv3 --pmc passes), and the
HBM bytes per launch derived from them.  Calibration as /opt/skills/guides/MI355X_MICROARCH.md asks
("calibrate on a known byte count in your own access pattern"): the library's streaming copy kernel
k_copy_probe moves exactly `copy_bytes` in and `copy_bytes` out per launch with the access pattern of
the model's kernels (8 bytes per lane); the factors bytes/FETCH_SIZE-unit and bytes/WRITE_SIZE-unit that
make its counters come out right are applied to every kernel and stored in the JSON.  On arrays that fit
the 256 MB Infinity Cache the memory-side counters can undercount (guide: "Infinity-Cache hits appear to
be counted" is uncertain there), so the large workloads are the ones to trust.

usage: tools/summarize_pmc.py <round> <tag> <workload> <copy_bytes_read_per_launch>
"""
import csv
import glob
import json
import os
import sys
from collections import defaultdict

ROOT = os.path.abspath(os.path.join(os.path.dirname(__file__), ".."))


# variants of one kernel (LDS / register forms, chunk sizes) are reported under the kernel's name
VARIANTS = {"k_s3uv_col_l": "k_s3uv_col", "k_s3uv_col_l10": "k_s3uv_col", "k_s3uv_col_r32": "k_s3uv_col", "k_s3uv_col_r52": "k_s3uv_col", "k_s3uv_couple_l": "k_s3uv_couple",
            "k_s3t_col_l": "k_s3t_col", "k_s3t_col_l10": "k_s3t_col", "k_s3t_col_n30": "k_s3t_col",
            "k_uv3dmix2_m": "k_uv3dmix2_s", "k_t3dmix2_m": "k_t3dmix2_s", "k_mp_vdiff_l": "k_mp_vdiff", "k_mp_limapply": "k_mp_apply", "k_lmd_col2": "k_lmd_col", "k_omega_l": "k_omega", "k_wvel_f": "k_wvel",
            "k_rhs3d_lds": "k_rhs3d_pt", "k_pre_new_m": "k_pre_new", "k_pre_new_m4": "k_pre_new"}


def canon(raw):
    """rocprofv3 kernel name -> the name DESIGN.md / bench.py use for the kernel"""
    k = raw.split("(")[0].replace("void ", "").strip()
    if k.startswith("k_tadv_lds<0"):          # LDS-tiled tracer advection: mode 0 = pre_step3d's predictor,
        return "k_pre_t3"
    if k.startswith("k_tadv_lds<1"):          # 1 = step3d_t's corrector advection
        return "k_s3t_hv"
    k = k.split("<")[0].strip()
    if k in ("k_step2d_a", "k_step2d_ac", "k_step2d_am", "k_step2d_b", "k_step2d_c", "k_step2d_d"):     # sub-tile variants of one kernel
        k = "k_step2d"
    if k in ("k_step2d_pair_a",):
        k = "k_step2d_pair"
    return VARIANTS.get(k, k)


def counter_avgs(d, counter):
    tot, disp = defaultdict(float), defaultdict(set)
    for f in glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True):
        for r in csv.DictReader(open(f)):
            if r.get("Counter_Name") != counter:
                continue
            k = canon(r["Kernel_Name"])
            tot[k] += float(r["Counter_Value"])
            disp[k].add(r["Dispatch_Id"])
    return {k: tot[k] / len(disp[k]) for k in tot}, {k: len(disp[k]) for k in tot}


def main():
    rnd, tag, workload, copy_bytes = sys.argv[1], sys.argv[2], sys.argv[3], float(sys.argv[4])
    src = os.path.join(ROOT, "gpurun_out", f"pmc_{tag}")
    fetch, nf = counter_avgs(os.path.join(src, "fetch"), "FETCH_SIZE")
    write, nw = counter_avgs(os.path.join(src, "write"), "WRITE_SIZE")
    cf = copy_bytes / fetch["k_copy_probe"] if fetch.get("k_copy_probe") else None
    cw = copy_bytes / write["k_copy_probe"] if write.get("k_copy_probe") else None
    out = {"workload": workload,
           "source": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes (tools/profile_pmc.sh)",
           "calibration": {"kernel": "k_copy_probe", "bytes_read_per_launch": copy_bytes,
                           "bytes_written_per_launch": copy_bytes, "FETCH_SIZE_raw": fetch.get("k_copy_probe"),
                           "WRITE_SIZE_raw": write.get("k_copy_probe"), "bytes_per_FETCH_SIZE_unit": cf,
                           "bytes_per_WRITE_SIZE_unit": cw},
           "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        e = {"launches_sampled": nf.get(k, 0), "FETCH_SIZE_raw_per_launch": fetch.get(k),
             "WRITE_SIZE_raw_per_launch": write.get(k)}
        if cf and cw and k in fetch and k in write:
            e["hbm_bytes_per_launch"] = fetch[k] * cf + write[k] * cw
        out["kernels"][k] = e
    dst = os.path.join(ROOT, "profiles", f"{rnd}_{tag}_traffic.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=1)
    print("wrote", dst)


if __name__ == "__main__":
    main()
