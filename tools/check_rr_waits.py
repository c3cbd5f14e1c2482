#!/usr/bin/env python3
"""ISA check for lstm_rr_kernel's COUNTED vmcnt wait (recurrent_rr.hip, `arrive`): the publishing wave waits for its three
write-through stores with s_waitcnt vmcnt(N), N = the vector-memory instructions it issues between those stores and the wait.
If the compiler drops or adds one (dead x loads in the last half-steps did), N is wrong: too large and the flag can overtake the
data.  This script compiles the file and, for every such wait of every instantiation, counts the vector-memory instructions
between the store group and the wait in the ISA; exit 1 on a mismatch.   usage: python tools/check_rr_waits.py"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "nntoolkitcore_amd", "csrc", "hip", "recurrent_rr.hip")


def main():
    with tempfile.TemporaryDirectory() as td:
        out = os.path.join(td, "rr.s")
        subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-function",
                               "--cuda-device-only", "-S", SRC, "-o", out] + os.environ.get("RR_EXTRA", "").split(), stderr=subprocess.DEVNULL)
        txt = open(out).read()
    bad = total = 0
    for kname in re.findall(r'^(_Z1[34](?:lstm|gru)_rr_kernel\w+):', txt, re.M):
        a = txt.index("\n" + kname + ":")
        s = txt[a:txt.index("s_endpgm", a)].split("\n")
        for i, l in enumerate(s):
            first = "buffer_store_dwordx4" in l and "sc1" in l and not ("buffer_store_dwordx4" in s[i - 1] and "sc1" in s[i - 1])
            if not first:
                continue
            # the two parities of a publication are the arms of an if / else: the second arm's group is reached from the first
            # arm's position too, so every group is followed to ITS wait and both must agree with the count
            younger = 0
            for j in range(i, min(i + 1500, len(s))):
                t = s[j].strip()
                if re.match(r'(buffer_|global_|scratch_|flat_)', t) and not ("buffer_store_dwordx4" in t and "sc1" in t):
                    younger += 1
                m = re.match(r's_waitcnt vmcnt\((\d+)\)$', t)
                if m and "ASMSTART" in s[j - 1]:
                    total += 1
                    if int(m.group(1)) != younger:
                        bad += 1
                        print("%s: wait vmcnt(%s) at +%d but %d vector-memory instructions follow the publication" % (kname, m.group(1), j, younger))
                    break
    print("lstm_rr_kernel / gru_rr_kernel: %d counted waits checked, %d mismatches" % (total, bad))
    return 1 if bad or total == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
