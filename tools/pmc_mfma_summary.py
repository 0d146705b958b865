"""Summarise one rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE pass into
profiles/<tag>_pmc_mfma.json: per kernel, MFMA-busy fraction
    frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE_per_XCD * n_simd)
SQ_VALU_MFMA_BUSY_CYCLES sums the busy cycles of all SIMDs (32 cycles per 32x32x16 bf16 MFMA,
MI355X_MICROARCH.md); GRBM_GUI_ACTIVE is reported summed over the 8 XCDs, so the kernel's cycle count
is that value / 8 and there are 256 CUs x 4 SIMDs = 1024 matrix pipes.

    python tools/pmc_mfma_summary.py gpurun_out/pmc_mfma/runc/*_counter_collection.csv profiles/r01_pmc_mfma.json"""
import collections
import csv
import json
import sys

N_XCD, N_SIMD = 8, 1024


def main(path, out_json, commit="", bench_sha16="", command=""):
    busy, act, ns = collections.defaultdict(float), collections.defaultdict(float), collections.defaultdict(float)
    n = collections.Counter()
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        v = float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_VALU_MFMA_BUSY_CYCLES":
            busy[k] += v
            n[k] += 1
            ns[k] += int(r["End_Timestamp"]) - int(r["Start_Timestamp"])
        elif r["Counter_Name"] == "GRBM_GUI_ACTIVE":
            act[k] += v
    out = {}
    for k in sorted(busy, key=lambda k: -busy[k]):
        if busy[k] <= 0:
            continue
        cyc = act[k] / N_XCD
        out[k] = {"launches": n[k], "mfma_busy_cycles_per_launch": busy[k] / n[k],
                  "kernel_cycles_per_launch": cyc / n[k], "avg_us_under_profiler": ns[k] / n[k] / 1e3,
                  "mfma_busy_frac": busy[k] / (cyc * N_SIMD)}
    json.dump({"source": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE, bench.py --steps 1 --warmup 1",
               "formula": "mfma_busy_frac = SQ_VALU_MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 XCDs * 1024 SIMDs)",
               "collected": {"commit": commit, "bench_py_sha16": bench_sha16, "command": command},
               "kernels": out}, open(out_json, "w"), indent=1)
    for k, v in list(out.items())[:12]:
        print(f"{k[:56]:56s} n={v['launches']:3d} busy_frac={v['mfma_busy_frac']:.3f} cycles={v['kernel_cycles_per_launch']:.3e} us={v['avg_us_under_profiler']:.0f}")


if __name__ == "__main__":
    main(*sys.argv[1:6])
