"""Static resource usage of every gfx950 kernel in wesep_amd/csrc (VGPRs, scratch = spills, LDS, waves/SIMD) from
hipcc's -Rpass-analysis=kernel-resource-usage remarks; no GPU needed.  Writes a markdown table.

    python tools/kernel_resources.py > profiles/r01_kernel_resources.md"""
import glob
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from wesep_amd.build import FLAGS  # noqa: E402  (the library's own compile flags)
rows = []
for src in sorted(glob.glob(os.path.join(ROOT, "wesep_amd", "csrc", "*.hip"))):
    out = subprocess.run(["hipcc"] + FLAGS + ["-c", src, "-o", os.devnull,
                          "-Rpass-analysis=kernel-resource-usage"], capture_output=True, text=True).stderr
    cur = None
    for line in out.splitlines():
        m = re.search(r"remark: .*?(Function Name|Name): (\S+)", line)
        if m:
            demangled = subprocess.run(["c++filt", m.group(2)], capture_output=True, text=True).stdout.strip()
            cur = {"file": os.path.basename(src), "kernel": re.sub(r"\(.*", "", demangled)}
            rows.append(cur)
            continue
        for key, pat in (("vgpr", r"\bVGPRs: (\d+)"), ("agpr", r"AGPRs: (\d+)"), ("sgpr", r"SGPRs: (\d+)"),
                         ("scratch", r"ScratchSize \[bytes/lane\]: (\d+)"), ("occ", r"Occupancy \[waves/SIMD\]: (\d+)"),
                         ("lds", r"LDS Size \[bytes/block\]: (\d+)")):
            m = re.search(pat, line)
            if m and cur is not None:
                cur[key] = int(m.group(1))
print("# Static kernel resources (gfx950, hipcc -O3; `python tools/kernel_resources.py`)\n")
print("Scratch > 0 means register spills.  Occupancy is the compiler's bound from registers alone (LDS and the launch "
      "bounds can lower it).\n")
print("| file | kernel | VGPRs | AGPRs | SGPRs | scratch B/lane | LDS B/block | waves/SIMD |")
print("|---|---|---|---|---|---|---|---|")
for r in rows:
    print(f"| {r['file']} | `{r['kernel']}` | {r.get('vgpr', '')} | {r.get('agpr', '')} | {r.get('sgpr', '')} | "
          f"{r.get('scratch', '')} | {r.get('lds', '')} | {r.get('occ', '')} |")
spills = [r for r in rows if r.get("scratch", 0) > 0]
print(f"\n{len(rows)} kernels, {len(spills)} with spills: " + ", ".join(f"`{r['kernel']}` ({r['scratch']} B)" for r in spills))
