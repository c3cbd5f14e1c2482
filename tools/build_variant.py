#!/usr/bin/env python3
"""A second copy of the library with SOME translation units recompiled with extra hipcc flags (diagnostics / A-B variants); every
other object is the product build's.   usage: python tools/build_variant.py <out.so> <unit.hip[,unit.hip...]> <extra hipcc flags...>
e.g.  python tools/build_variant.py gpurun_out/libs/libstamps.so recurrent_rr.hip,recurrent_rr4.hip -DNNTK_REC_STAMPS"""
import glob, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from nntoolkitcore_amd import _build

out, units, flags = sys.argv[1], sys.argv[2].split(","), sys.argv[3:]
_build.build()
os.makedirs(os.path.dirname(os.path.abspath(out)), exist_ok=True)
tmp = os.path.abspath(out) + ".obj"
os.makedirs(tmp, exist_ok=True)
procs, objs = [], []
for u in units:
    o = os.path.join(tmp, u + ".o")
    objs.append(o)
    procs.append(subprocess.Popen([_build.HIPCC, "-O3", "--offload-arch=" + _build.ARCH, "-fPIC", "-std=c++17", "-Wno-unused-function"] + flags +
                                  ["-c", os.path.join(_build.CSRC, "hip", u), "-o", o]))
assert all(p.wait() == 0 for p in procs)
others = [o for o in glob.glob(os.path.join(_build.OBJ, "*.o")) if os.path.basename(o)[:-2] not in units]
subprocess.check_call([_build.HIPCC, "--offload-arch=" + _build.ARCH, "-shared", "-fPIC", "-o", out] + objs + others)
print("built", out)
