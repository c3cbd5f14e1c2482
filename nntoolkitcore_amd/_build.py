"""Build libnntoolkitcore_hip.so in-tree: C host layer with gcc, kernels with hipcc for gfx950.

The shared object lands in ``nntoolkitcore_amd/lib/`` (git-ignored, but it travels to
the GPU box with the gpurun snapshot).  hipcc cross-compiles without a GPU.
"""
import os
import subprocess
import sys

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
OBJ = os.path.join(PKG, "lib", "obj")
LIB = os.path.join(PKG, "lib", "libnntoolkitcore_hip.so")

HOST_SRC = ["runtime.c", "activation.c", "conv_1d.c", "recurrent.c", "dense.c", "spectrogram.c", "mel.c", "train.c"]
HIP_SRC = ["runtime.hip", "conv1d.hip", "conv1d_s2.hip", "conv1d_flatk.hip", "recurrent.hip", "recurrent_rr.hip", "recurrent_fk.hip", "frag3.hip", "spectrogram.hip", "dist.hip", "conv1d_grad.hip", "train.hip"]
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
ARCH = "gfx950"


def source_hash():
    """sha256 (first 16 hex digits) over the kernel + host sources the library is built from: the identity a
    profile (profiles/*_pmc_traffic.json) is stamped with, so bench.py never prints another library's counters."""
    import hashlib
    h = hashlib.sha256()
    files = [os.path.join(CSRC, "host", f) for f in HOST_SRC] + [os.path.join(CSRC, "hip", f) for f in HIP_SRC]
    files += [os.path.join(CSRC, "hip", "nntk_shim.h"), os.path.join(CSRC, "hip", "nntk_common.hpp"),
              os.path.join(CSRC, "hip", "conv1d_kernels.hpp"), os.path.join(CSRC, "hip", "recurrent_rr_common.hpp"),
              os.path.join(CSRC, "host", "nntk_internal.h"), os.path.join(ROOT, "include", "nntoolkitcore_hip.h")]
    for f in sorted(files):
        h.update(os.path.basename(f).encode())
        with open(f, "rb") as fh:
            h.update(fh.read())
    return h.hexdigest()[:16]


def _newer(src_list, target):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in src_list)


def _run(cmd):
    res = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if res.returncode != 0:
        sys.stderr.write(res.stdout)
        raise RuntimeError("build step failed: " + " ".join(cmd))
    return res.stdout


def build(force=False, verbose=False):
    os.makedirs(OBJ, exist_ok=True)
    headers = [os.path.join(ROOT, "include", "nntoolkitcore_hip.h"),
               os.path.join(CSRC, "hip", "nntk_shim.h"),
               os.path.join(CSRC, "hip", "nntk_common.hpp"),
               os.path.join(CSRC, "hip", "conv1d_kernels.hpp"),
               os.path.join(CSRC, "hip", "recurrent_rr_common.hpp"),
               os.path.join(CSRC, "host", "nntk_internal.h")]
    objs, jobs = [], []
    for f in HOST_SRC:
        src = os.path.join(CSRC, "host", f)
        obj = os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            jobs.append(["gcc", "-O2", "-fPIC", "-std=gnu11", "-Wall", "-Wextra", "-Wno-unused-parameter",
                         "-fvisibility=default", "-I" + os.path.join(ROOT, "include"), "-c", src, "-o", obj])
    for f in HIP_SRC:
        src = os.path.join(CSRC, "hip", f)
        obj = os.path.join(OBJ, f + ".o")
        objs.append(obj)
        if force or _newer([src] + headers, obj):
            jobs.append([HIPCC, "-O3", "--offload-arch=" + ARCH, "-fPIC", "-std=c++17", "-Wall",
                         "-Wno-unused-function"] + os.environ.get("NNTK_EXTRA_HIPFLAGS", "").split() +
                        ["-c", src, "-o", obj])
    # the identity of the sources this library is built from, embedded so that a consumer (bench.py's roofline.traffic gate) asks the
    # LOADED binary, not the source tree: a stale .so next to edited sources must not borrow the new sources' profile
    bid_src = os.path.join(OBJ, "build_id.c")
    bid_obj = os.path.join(OBJ, "build_id.c.o")
    bid_txt = 'const char *nntk_build_source_hash(void) { return "%s"; }\n' % source_hash()
    objs.append(bid_obj)
    if force or not os.path.exists(bid_src) or open(bid_src).read() != bid_txt or not os.path.exists(bid_obj):
        with open(bid_src, "w") as fh:
            fh.write(bid_txt)
        jobs.append(["gcc", "-O2", "-fPIC", "-fvisibility=default", "-c", bid_src, "-o", bid_obj])
    if jobs:
        # the translation units are independent: compile them side by side (the two conv1d units take minutes each)
        from concurrent.futures import ThreadPoolExecutor
        workers = max(1, min(len(jobs), int(os.environ.get("NNTK_BUILD_JOBS", "0")) or (os.cpu_count() or 4)))
        with ThreadPoolExecutor(workers) as pool:
            for out in pool.map(_run, jobs):
                if verbose and out:
                    print(out)
    if force or _newer(objs, LIB):
        _run([HIPCC, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB] + objs)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
