"""TEST INFRASTRUCTURE.  Builds oracle/_ref/libref_fbank.so: the reference's own C++ kaldi-style fbank
(runtime/frontend/fbank.h + fft.cc) compiled with g++ from the sources where they lie under /root/reference.
No reference source is copied into the repo; the output directory is git-ignored (it still ships to the GPU box).
The rest of the reference runtime (separate_engine, feature_pipeline) needs libtorch/gflags/glog fetched over the
network by cmake and is unbuildable here."""
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
REF_RT = "/root/reference/runtime"
OUT = os.path.join(HERE, "_ref", "libref_fbank.so")


def available() -> bool:
    return os.path.exists(OUT)


def build(force=False) -> bool:
    """-> True when the library exists afterwards; False when /root/reference is absent (GPU box) and nothing was
    prebuilt."""
    srcs = [os.path.join(HERE, "ref_fbank_driver.cc"), os.path.join(REF_RT, "frontend", "fft.cc")]
    if not os.path.isdir(REF_RT):
        return available()
    if available() and not force and all(os.path.getmtime(OUT) >= os.path.getmtime(s) for s in srcs):
        return True
    os.makedirs(os.path.dirname(OUT), exist_ok=True)
    cmd = ["g++", "-O2", "-std=c++14", "-shared", "-fPIC", "-I", REF_RT, "-I", os.path.join(HERE, "ref_shim"),
           *srcs, "-o", OUT]
    subprocess.check_call(cmd)
    return True


if __name__ == "__main__":
    print("built" if build(force=True) else "reference sources absent", OUT)
