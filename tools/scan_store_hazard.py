"""Scan the gfx950 ISA of every kernel in wesep_amd/csrc for the store-data hazard found in round 3
(profiles/r03_store_hazard.md):

    buffer_store_dwordx3/x4  v[a:b], vaddr, s[rsrc], sN offen ...     <- soffset is an SGPR
    <VALU / LDS-read / VMEM-load instruction that WRITES one of v[a..b]>   within the next `--window` instructions

hipcc's hazard recognizer pads a VALU write behind a >64-bit MUBUF store only when the store has NO register soffset
(GCNHazardRecognizer::createsVALUHazard); on the MI355X the store with a register soffset was observed to pick up the
overwritten registers as well (sparse wrong dwords in the lanes / dwords read last).  Global / flat stores are padded
unconditionally and are reported separately (`--all`) only for completeness.

    python tools/scan_store_hazard.py [--window 2] [files.hip ...]      # exit status 1 when a hit is found
"""
import argparse
import glob
import os
import re
import subprocess
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(os.path.dirname(HERE), "wesep_amd", "csrc")

STORE = re.compile(r"^\s*buffer_store_dwordx([34])\s+v\[(\d+):(\d+)\],\s*(\S+),\s*s\[\d+:\d+\],\s*(\S+)\s+(.*)$")
VREG = re.compile(r"v\[(\d+):(\d+)\]|v(\d+)")
NOT_INSTR = re.compile(r"^\s*(;|\.|$)|^\S+:")          # comments, directives, labels


def dests(line):
    """VGPRs an instruction writes (first operand of VALU / ds_read / loads / mfma / v_readlane excluded)."""
    m = re.match(r"^\s*(\S+)\s+(.*)$", line)
    if not m:
        return set()
    op, rest = m.group(1), m.group(2)
    if op.startswith(("s_", "buffer_store", "global_store", "flat_store", "ds_write", "scratch_store", "v_cmp",
                      "v_readlane", "v_readfirstlane", "buffer_wbl2", "buffer_inv", "ds_add", "ds_bpermute")):
        if not op.startswith("ds_bpermute"):
            return set()
    first = rest.split(",")[0].strip()
    m2 = VREG.match(first)
    if not m2:
        return set()
    if m2.group(3) is not None:
        return {int(m2.group(3))}
    return set(range(int(m2.group(1)), int(m2.group(2)) + 1))


def scan(path, window):
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "k.s")
        subprocess.check_call(["hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", path,
                               "-o", out], stderr=subprocess.DEVNULL)
        lines = open(out).read().splitlines()
    hits, kernel, nstores, nsoff = [], None, 0, 0
    instrs = []
    for ln in lines:
        if re.match(r"^_Z\S+:|^[A-Za-z_]\w*:\s*(;.*)?$", ln) and not ln.startswith(".L"):
            kernel = ln.split(":")[0]
        if NOT_INSTR.match(ln):
            continue
        instrs.append((kernel, ln))
    for i, (k, ln) in enumerate(instrs):
        m = STORE.match(ln)
        if not m:
            continue
        nstores += 1
        soff = m.group(5)
        if not soff.startswith("s"):
            continue
        nsoff += 1
        data = set(range(int(m.group(2)), int(m.group(3)) + 1))
        seen = 0
        for k2, nxt in instrs[i + 1:]:
            op = nxt.split()[0]
            if op in ("s_nop", "s_waitcnt") or op.startswith(";"):
                if op == "s_nop":
                    seen += 1 + int(nxt.split()[1])
                continue
            if op in ("s_barrier", "s_endpgm") or op.startswith("s_cbranch") or op.startswith("s_branch"):
                break
            seen += 1
            if seen > window:
                break
            w = dests(nxt) & data
            if w:
                hits.append((k, ln.strip(), nxt.strip(), seen))
                break
    return hits, nstores, nsoff


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("files", nargs="*")
    ap.add_argument("--window", type=int, default=2, help="instruction slots behind the store that count as 'next'")
    a = ap.parse_args()
    files = a.files or sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    total = 0
    for f in files:
        hits, ns, nsoff = scan(f, a.window)
        print(f"{os.path.basename(f):24s} {ns:4d} 12/16-byte buffer stores, {nsoff:4d} with an SGPR soffset, "
              f"{len(hits):3d} followed within {a.window} slots by a write to their data registers")
        for k, st, nx, d in hits:
            print(f"    {k}\n        {st}\n        +{d}: {nx}")
        total += len(hits)
    return 1 if total else 0


if __name__ == "__main__":
    sys.exit(main())
