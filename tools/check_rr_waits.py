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
    pbad, ptotal = check_polls(txt)
    print("lstm_rr_kernel / gru_rr_kernel: %d flag polls checked, %d violations" % (ptotal, pbad))
    return 1 if bad or total == 0 or pbad or ptotal == 0 else 0


def regs_of(tok):
    """VGPR numbers an operand token names: v12 -> {12}, v[14:15] -> {14, 15}"""
    m = re.match(r'v\[(\d+):(\d+)\]', tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r'v(\d+)$', tok)
    return {int(m.group(1))} if m else set()


def check_polls(txt):
    """The flag poll is two asm statements a k step apart: `global_load_dword vN, ..., off sc1` (poll_a) and `s_waitcnt vmcnt(0)` with
    vN as an in/out operand (poll_b).  The compiler does not know vN is in flight in between, so nothing enforces that it leaves the
    register alone.  Checked here for every poll of every instantiation: between the load and the first asm vmcnt(0) wait
    that follows it, no instruction names vN as an operand and no branch is taken; and no rr kernel uses scratch (a spill of vN would
    be a hidden read + write)."""
    bad = total = 0
    for kname in re.findall(r'^(_Z1[34](?:lstm|gru)_rr_kernel\w+):', txt, re.M):
        a = txt.index("\n" + kname + ":")
        s = txt[a:txt.index("s_endpgm", a)].split("\n")
        m = re.search(r'\.amdhsa_kernel %s\b.*?\.end_amdhsa_kernel' % re.escape(kname), txt, re.S)
        if m:
            ps = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(0))
            if ps and int(ps.group(1)) != 0:
                bad += 1
                print("%s: private_segment_fixed_size %s (scratch in use: a spill may touch the poll register)" % (kname, ps.group(1)))
        for i, l in enumerate(s):
            t = l.strip()
            pm = re.match(r'global_load_dword (v\d+), v\[\d+:\d+\], off sc1$', t)
            if not (pm and "ASMSTART" in s[i - 1]):
                continue
            total += 1
            reg = int(pm.group(1)[1:])
            closed = False
            for j in range(i + 1, min(i + 400, len(s))):
                u = s[j].strip()
                if not u or u.startswith(";") or u.startswith("."):
                    if re.match(r'\.LBB', u):
                        print("%s: a branch target between the flag load (v%d) and its wait" % (kname, reg)); bad += 1; closed = True
                        break
                    continue
                if u == "s_waitcnt vmcnt(0)" and "ASMSTART" in s[j - 1]:
                    closed = True
                    break
                if re.match(r's_(c?branch|setpc|swappc|call)', u):
                    print("%s: `%s` between the flag load (v%d) and its wait" % (kname, u, reg)); bad += 1; closed = True
                    break
                ops = re.split(r'[,\s]+', u)[1:]
                if any(reg in regs_of(o) for o in ops):
                    print("%s: `%s` touches v%d while the flag load is in flight" % (kname, u, reg)); bad += 1; closed = True
                    break
            if not closed:
                print("%s: no asm vmcnt(0) wait within 400 lines of the flag load into v%d" % (kname, reg)); bad += 1
    return bad, total


if __name__ == "__main__":
    sys.exit(main())
