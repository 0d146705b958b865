"""Fill `roofline.traffic` (HBM bytes per launch of the dominant kernel class, from the rocprofv3 PMC passes summarised by
tools/pmc_summary.py) and `roofline.mfma_busy_pmc` into the bench line of a per-model tool (tools/bench_dpccn.py,
bench_tfgridnet.py, bench_convtasnet.py), whose own run cannot collect counters.  The class -> kernel-name map follows the
library's HIP-event brackets (ws_prof_begin sites in wesep_amd/csrc).

    python tools/r04_attach_traffic.py line.json pmc_traffic.json [pmc_mfma.json] > line_with_traffic.json"""
import json
import sys

CLASS_KERNELS = {
    "lstm_fwd": ("lstm_fwd",),
    "lstm_bwd": ("lstm_bwd",),
    "gemm_nt": ("gemm_nt", "gemm_p2b", "gemm_b2p", "conv3x3_kernel", "conv3x3_fwd"),
    "gemm_tn": ("gemm_tn", "conv_wgrad", "conv3x3_wgrad"),
}


def main(line_path, traffic_path, mfma_path=None):
    line = json.loads([l for l in open(line_path).read().splitlines() if l.startswith("{")][-1])
    roof = line.get("roofline") or {}
    cls = (roof.get("kernel") or "").split(" ")[0]
    keys = CLASS_KERNELS.get(cls, (cls,))
    doc = json.load(open(traffic_path))
    hit = [v for k, v in doc["kernels"].items() if any(s in k for s in keys)]
    n = sum(v["launches"] for v in hit)
    if n:
        roof["traffic"] = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in hit) / n
        roof["traffic_over_algorithmic"] = roof["traffic"] / roof["bytes_per_launch"] if roof.get("bytes_per_launch") else None
    src = {"traffic": dict(doc.get("collected", {}), kernels_matched=sorted({s for s in keys}), launches=n)}
    tot = sum(v["hbm_bytes_per_launch_corrected"] * v["launches"] for v in doc["kernels"].values())
    src["traffic"]["all_kernels_gb_over_collected_steps"] = tot / 1e9
    if mfma_path:
        md = json.load(open(mfma_path))
        mh = [v for k, v in md["kernels"].items() if any(s in k for s in keys)]
        mn = sum(v["launches"] for v in mh)
        if mn:
            roof["mfma_busy_pmc"] = sum(v["mfma_busy_frac"] * v["launches"] for v in mh) / mn
        src["mfma"] = md.get("collected", {})
    roof["counters_from"] = src
    line["roofline"] = roof
    print(json.dumps(line))


if __name__ == "__main__":
    main(*sys.argv[1:4])
