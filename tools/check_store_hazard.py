#!/usr/bin/env python3
"""ISA lint for a gfx950 hazard the compiler does not cover: a buffer store of more than 64 bits of data whose SOFFSET
is an SGPR, followed within two instructions by a VALU write of one of its data registers.  LLVM's hazard recognizer
treats only the immediate-soffset form as hazardous (GCNHazardRecognizer::createsVALUHazard); on MI355X the SGPR form
corrupts the stored data too (seen in conv1d.hip's epilogue: lanes 12-15 / 28-31 of each half stored the NEXT loop
iteration's channel index instead of the result).  Kernels therefore keep soffset immediate on wide stores.

usage: python tools/check_store_hazard.py [file.hip ...]     (default: every csrc/hip/*.hip)   exit 1 on a finding"""
import glob, os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
STORE = re.compile(r'buffer_store_dwordx[34]\s+v\[(\d+):(\d+)\],\s*\S+,\s*s\[\d+:\d+\],\s*(s\d+|ttmp\d+|m0)\b')


def scan(asm_text):
    lines = asm_text.split("\n")
    kernel, found = None, []
    for n, l in enumerate(lines):
        km = re.match(r'(_Z\w+):', l)
        if km:
            kernel = km.group(1)
        m = STORE.search(l)
        if not m:
            continue
        lo, hi = int(m.group(1)), int(m.group(2))
        cnt = 0
        for k in range(n + 1, min(n + 12, len(lines))):
            t = lines[k].strip()
            if not t or t[0] in ";." or t.endswith(":"):
                continue
            cnt += 1
            regs = []
            w = re.match(r'v_\w+\s+v(\d+)\b', t)
            w2 = re.match(r'v_\w+\s+v\[(\d+):(\d+)\]', t)
            if w: regs = [int(w.group(1))]
            if w2: regs = list(range(int(w2.group(1)), int(w2.group(2)) + 1))
            if any(lo <= r <= hi for r in regs):
                found.append((kernel, n + 1, l.strip(), t))
                break
            if t.startswith("s_nop") or cnt >= 2:
                break
    return found


def main():
    files = sys.argv[1:] or sorted(glob.glob(os.path.join(ROOT, "nntoolkitcore_amd", "csrc", "hip", "*.hip")))
    bad = 0
    from concurrent.futures import ThreadPoolExecutor
    with tempfile.TemporaryDirectory() as td:
        def compile_one(f):
            out = os.path.join(td, os.path.basename(f) + ".s")
            subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-function",
                                   "--cuda-device-only", "-S", "-I", os.path.dirname(f), f, "-o", out], stderr=subprocess.DEVNULL)
            return out
        with ThreadPoolExecutor(4) as ex:
            outs = list(ex.map(compile_one, files))
        for f, out in zip(files, outs):
            res = scan(open(out).read())
            print("%s: %d wide SGPR-soffset stores with a data register overwritten right behind them" % (os.path.basename(f), len(res)))
            for kern, ln, st, wr in res[:5]:
                print("   %s  line %d: %s  ->  %s" % (kern, ln, st, wr))
            bad += len(res)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
