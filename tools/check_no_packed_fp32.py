"""Disassemble every gfx950 code object inside wesep_amd/libwesep_hip.so and count packed-FP32 instructions
(v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32).  The library is built without them (wesep_amd/build.py NO_PACKED_FP32;
profiles/r03_kernel_race.md): exit status 1 if any is found, or if no MFMA kernel was seen (= the walk missed the code).

    python tools/check_no_packed_fp32.py [path/to/lib.so]
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib):
    """gfx950 ELF images of every offload bundle in the library's .hip_fatbin section."""
    with tempfile.TemporaryDirectory() as td:
        fat = os.path.join(td, "fat.bin")
        subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "-O", "binary", "--only-section=.hip_fatbin", lib, fat])
        blob = open(fat, "rb").read()
    out, at = [], blob.find(MAGIC)
    while at >= 0:
        n, = struct.unpack_from("<Q", blob, at + len(MAGIC))
        p = at + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, p)
            triple = blob[p + 24:p + 24 + tlen].decode()
            p += 24 + tlen
            if "gfx950" in triple and size:
                out.append(blob[at + off:at + off + size])
        at = blob.find(MAGIC, at + len(MAGIC))
    return out


def histogram(lib):
    hist = {}
    for co in code_objects(lib):
        with tempfile.NamedTemporaryFile(suffix=".co") as f:
            f.write(co)
            f.flush()
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", f.name], capture_output=True, text=True).stdout
        for m in re.finditer(r"\b(v_pk_[a-z0-9_]+|v_mfma_[a-z0-9_]+)", dis):
            hist[m.group(1)] = hist.get(m.group(1), 0) + 1
    return hist


def main():
    here = os.path.dirname(os.path.abspath(__file__))
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(os.path.dirname(here), "wesep_amd", "libwesep_hip.so")
    hist = histogram(lib)
    for k in sorted(hist):
        print(f"{hist[k]:7d}  {k}")
    packed = {k: v for k, v in hist.items() if re.fullmatch(r"v_pk_(mul|fma|add)_f32", k)}
    mfma = sum(v for k, v in hist.items() if k.startswith("v_mfma"))
    if packed or not mfma:
        print("FAIL:", packed or "no MFMA instruction found -- the code objects were not disassembled")
        return 1
    print("ok: no packed FP32 arithmetic in", lib)
    return 0


if __name__ == "__main__":
    sys.exit(main())
