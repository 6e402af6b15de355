#!/usr/bin/env python3
"""HBM bytes per launch of one kernel from two rocprofv3 PMC passes (FETCH_SIZE, WRITE_SIZE collected in SEPARATE runs,
as MI355X_MICROARCH.md's HBM section prescribes: the two counters do not fit one pass, and counter passes must not be
combined with tracing other than --kernel-trace).

  rocprofv3 --kernel-trace --pmc FETCH_SIZE --kernel-include-regex 'k_ll' -d out/fetch -- python3 tools/profile_ll.py ...
  rocprofv3 --kernel-trace --pmc WRITE_SIZE --kernel-include-regex 'k_ll' -d out/write -- python3 tools/profile_ll.py ...
  python tools/pmc_traffic.py --kernel k_ll_fused4 --sites 10000000 --fetch out/fetch --write out/write \
         --expect-read-bytes 1.0e9 --out profiles/traffic_cfg3.json

Units and corrections (the guide's): the counters are in KB; on gfx950 FETCH_SIZE tallies the 128-byte requests of wide
coalesced streaming reads at 64 bytes, so it is doubled; WRITE_SIZE reads exactly for 16-byte-per-lane stores.  Other
access widths are uncalibrated: give --expect-read-bytes (a byte count known from the kernel's access pattern) and the
file records how far 2 x FETCH_SIZE is from it; the factor itself is never tuned to fit.
bench.py uses the file only if `kernel` and `sites` match the run."""
import argparse
import csv
import glob
import json
import os
import statistics


def counter_values(path, name, kernel):
    files = [path] if os.path.isfile(path) else glob.glob(os.path.join(path, "**", "*counter_collection.csv"), recursive=True)
    vals, kname = [], None
    for f in files:
        with open(f, newline="") as fh:
            for row in csv.DictReader(fh):
                if row.get("Counter_Name") == name and kernel in row.get("Kernel_Name", ""):
                    vals.append(float(row["Counter_Value"]))
                    kname = row["Kernel_Name"]
    return vals, kname


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--kernel", required=True, action="append", help="substring of the kernel name (repeat to add the kernels of one step up)")
    ap.add_argument("--source-key", default="", help="key of bench.py's KERNEL_SOURCES: the git blob hashes of those files are recorded, "
                    "and bench.py uses the result only on a tree whose files hash the same")
    ap.add_argument("--sites", type=int, required=True)
    ap.add_argument("--fetch", required=True, help="output directory (or csv) of the FETCH_SIZE pass")
    ap.add_argument("--write", required=True, help="output directory (or csv) of the WRITE_SIZE pass")
    ap.add_argument("--expect-read-bytes", type=float, default=None)
    ap.add_argument("--config", default="")
    ap.add_argument("--out", required=True)
    a = ap.parse_args()
    per_kernel, read_b, write_b, names, launches = {}, 0.0, 0.0, [], []
    for kern in a.kernel:
        fv, fk = counter_values(a.fetch, "FETCH_SIZE", kern)
        wv, wk = counter_values(a.write, "WRITE_SIZE", kern)
        if not fv or not wv:
            raise SystemExit("pmc_traffic: no %s rows for kernel '%s'" % ("FETCH_SIZE" if not fv else "WRITE_SIZE", kern))
        fetch_kb, write_kb = statistics.median(fv), statistics.median(wv)
        name = (fk or kern).replace("void ", "").split("(")[0]
        per_kernel[name] = {"FETCH_SIZE_KB": fetch_kb, "WRITE_SIZE_KB": write_kb, "hbm_read_bytes_corrected": 2.0 * fetch_kb * 1024.0,
                            "hbm_write_bytes": write_kb * 1024.0, "launches_measured": [len(fv), len(wv)]}
        read_b += 2.0 * fetch_kb * 1024.0
        write_b += write_kb * 1024.0
        names.append(name)
        launches.append([len(fv), len(wv)])
    out = {"kernel": names[0] if len(names) == 1 else " + ".join(names), "sites": a.sites, "config": a.config,
           "launches_measured": launches[0] if len(launches) == 1 else launches,
           "hbm_read_bytes_corrected": read_b, "hbm_write_bytes": write_b, "hbm_bytes_per_launch": read_b + write_b,
           "bytes_per_site": (read_b + write_b) / a.sites,
           "correction": "FETCH_SIZE x 2 (gfx950 tallies 128-byte requests at 64 bytes; MI355X_MICROARCH.md, HBM section), WRITE_SIZE as read; separate PMC passes"}
    if len(names) == 1:
        out.update({"FETCH_SIZE_KB": per_kernel[names[0]]["FETCH_SIZE_KB"], "WRITE_SIZE_KB": per_kernel[names[0]]["WRITE_SIZE_KB"]})
    else:
        out["per_kernel"] = per_kernel
    if a.source_key:
        import sys
        sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
        import bench
        out["source_key"] = a.source_key
        out["source_blobs"] = bench.source_hashes(a.source_key)
    if a.expect_read_bytes:
        out["expected_read_bytes"] = a.expect_read_bytes
        out["read_vs_expected"] = read_b / a.expect_read_bytes
    with open(a.out, "w") as f:
        json.dump(out, f, indent=1)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
