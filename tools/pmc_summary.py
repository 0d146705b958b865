"""Summarise two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE; separate runs, as
MI355X_MICROARCH.md prescribes) into profiles/<tag>_pmc_traffic.json:
per kernel, average per-launch FETCH_SIZE / WRITE_SIZE (KiB as reported) and the corrected HBM bytes
    traffic = 2 * FETCH_SIZE * 1024 + WRITE_SIZE * 1024
(gfx950: FETCH_SIZE reports half of a wide coalesced 16 B/lane read; WRITE_SIZE matched our known
write volumes exactly, e.g. lstm_fwd: 24 fp32 per position-direction-unit = 6.304 GB).

    python tools/pmc_summary.py gpurun_out/pmc_FETCH_SIZE/pmc_counter_collection.csv \
        gpurun_out/pmc_WRITE_SIZE/pmc_counter_collection.csv profiles/r01_pmc_traffic.json \
        [commit bench_py_sha16 command]      (provenance recorded under "collected")"""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return agg


def main(fetch_csv, write_csv, out_json, commit="", bench_sha16="", command=""):
    f, w = per_kernel(fetch_csv), per_kernel(write_csv)
    out = {}
    for k in sorted(set(f) | set(w), key=lambda k: -(sum(f.get(k, [0])) + sum(w.get(k, [0])))):
        fa = sum(f[k]) / len(f[k]) if k in f else 0.0
        wa = sum(w[k]) / len(w[k]) if k in w else 0.0
        out[k] = {"launches": len(f.get(k, w.get(k))), "fetch_size_kib_avg": fa, "write_size_kib_avg": wa,
                  "hbm_bytes_per_launch_corrected": 2 * fa * 1024 + wa * 1024,
                  "hbm_bytes_per_launch_raw": fa * 1024 + wa * 1024}
    json.dump({"source": "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, separate passes, bench.py --steps 1 --warmup 1",
               "correction": "traffic = 2*FETCH_SIZE + WRITE_SIZE (MI355X_MICROARCH.md, HBM section)",
               "collected": {"commit": commit, "bench_py_sha16": bench_sha16, "command": command},
               "kernels": out}, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main(*sys.argv[1:7])
