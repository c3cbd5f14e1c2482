#!/usr/bin/env python3
"""ISA checks for the hand-off protocols of recurrent_rr.hip.  Flag protocol: lstm_rr_kernel's COUNTED vmcnt wait (recurrent_rr.hip, `arrive`): the publishing wave waits for its three
write-through stores with s_waitcnt vmcnt(N), N = the vector-memory instructions it issues between those stores and the wait.
If the compiler drops or adds one (dead x loads in the last half-steps did), N is wrong: too large and the flag can overtake the
data.  This script compiles the file and, for every such wait of every instantiation, counts the vector-memory instructions
between the store group and the wait in the ISA; exit 1 on a mismatch.   usage: python tools/check_rr_waits.py"""
import os, re, subprocess, sys, tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRCS = [os.path.join(ROOT, "nntoolkitcore_amd", "csrc", "hip", f) for f in ("recurrent_rr.hip", "recurrent_fk.hip", "frag3.hip", "conv1d.hip")]


def main():
    txt = ""
    with tempfile.TemporaryDirectory() as td:
        procs = []
        for k, src in enumerate(SRCS):                  # (the four units compile side by side)
            out = os.path.join(td, "rr%d.s" % k)
            procs.append((out, subprocess.Popen(["/opt/rocm/bin/hipcc", "-O3", "--offload-arch=gfx950", "-std=c++17", "-Wno-unused-function",
                                                 "--cuda-device-only", "-S", src, "-o", out] + os.environ.get("RR_EXTRA", "").split(),
                                                stderr=subprocess.DEVNULL)))
        for out, pr in procs:
            if pr.wait() != 0:
                raise SystemExit("hipcc failed on " + out)
            txt += open(out).read() + "\n"
    bad = total = 0
    for kname in re.findall(r'^(_Z1[345](?:lstm|gru)_rr_kernel\w+):', txt, re.M):
        a = txt.index("\n" + kname + ":")
        s = txt[a:txt.index(".Lfunc_end", a)].split("\n")           # (a kernel may hold several s_endpgm and out-of-line blocks behind them)
        b_, t_ = check_counted_waits(kname, s)
        bad += b_
        total += t_
    print("lstm_rr_kernel / gru_rr_kernel: %d counted waits checked, %d mismatches" % (total, bad))
    pbad, ptotal = check_polls(txt)
    print("lstm_rr_kernel / gru_rr_kernel: %d flag polls checked, %d violations" % (ptotal, pbad))
    qbad, qtotal = check_pending(txt)
    print("lstm_rr_kernel / gru_rr_kernel: %d pending-pattern looks checked, %d defects" % (qtotal, qbad))
    mbad, mtotal = check_marks(txt)
    print("pending-pattern kernels (rr KH = 4, fk): %d mark groups checked, %d not covered by a vmcnt wait before the second barrier / the next publication" % (mtotal, mbad))
    fbad, ftotal = check_split_fma(txt)
    print("bf16 x 3 splits (rr, fk, frag3, conv): %d conversions checked, %d fed by a fused multiply-add of an image" % (ftotal, fbad))
    sbad, stotal = check_scratch(txt)
    print("register-resident / MFMA kernels (rr, fk, dense_frag3, conv1d): %d kernels checked, %d with a private segment (spills)" % (stotal, sbad))
    return 1 if (bad or total == 0 or pbad or ptotal == 0 or qbad or qtotal == 0 or mbad or mtotal == 0 or fbad or ftotal == 0
                 or sbad or stotal == 0) else 0


def check_scratch(txt):
    """No kernel of these units may spill: the register-resident kernels and dense_frag3_kernel sit at the 512-register limit on purpose, and a
    change that tips them over still compiles, still passes every test -- and runs 1.1 to 10 x slower (the '#pragma unroll' that gives up
    silently and leaves a loop indexing the accumulators dynamically is the usual cause: recurrent_fk.hip fk_for, frag3.hip f3_for)."""
    bad = total = 0
    for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', txt, re.S):
        ps = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(2))
        total += 1
        if ps and ps.group(1) != "0":
            bad += 1
            print("%s: private_segment_fixed_size %s" % (m.group(1), ps.group(1)))
    return bad, total


def is_pub_store(t):
    return "buffer_store_dwordx4" in t and "sc1" in t


def is_vmem(t):
    return re.match(r'(buffer_|global_|scratch_|flat_)', t) is not None


def check_counted_waits(kname, s):
    """For every publication (a group of write-through 16-byte stores) follow the CONTROL FLOW to the asm `s_waitcnt vmcnt(N)` that
    guards its flag: N must equal the smallest number of vector-memory instructions on any path from the stores to the wait.  (The
    stores have completed once at most as many operations are outstanding as the wave issued after them; N larger than that and the
    flag can overtake the data.  Arms another wave takes -- the output wave's stores -- only add instructions, so the minimum is the
    publishing wave's own path; hipcc places such arms out of line, which is why this walks labels and branches, not lines.)"""
    ins = [l.strip() for l in s]
    labels = {}
    for i, t in enumerate(ins):
        m = re.match(r'(\.LBB\w+):', t)
        if m:
            labels[m.group(1)] = i
    bad = total = 0
    for i, t in enumerate(ins):
        if not (is_pub_store(t) and not is_pub_store(ins[i - 1])):
            continue
        j = i
        while is_pub_store(ins[j]):
            j += 1
        best, found = {}, {}                       # instruction index -> fewest vector-memory instructions on a path reaching it
        stack = [(j, 0)]
        while stack:
            k, cnt = stack.pop()
            steps = 0
            while k < len(ins) and steps < 6000:
                if best.get(k, 1 << 30) <= cnt:
                    break
                best[k] = cnt
                u = ins[k]
                steps += 1
                if is_pub_store(u):                # the next publication: this path never raised the flag (not the publishing wave's)
                    break
                m = re.match(r's_waitcnt vmcnt\((\d+)\)$', u)
                # the arrival's wait is the asm wait that is followed by the flag store (the poll's asm wait is not)
                if m and k > 0 and "ASMSTART" in ins[k - 1] and any(x.startswith("global_store_dword ") for x in ins[k + 1:k + 16]):
                    key = (k, int(m.group(1)))
                    found[key] = min(found.get(key, 1 << 30), cnt)
                    break
                if is_vmem(u):
                    cnt += 1
                m = re.match(r's_branch\s+(\.LBB\w+)', u)
                if m:
                    k = labels[m.group(1)]
                    continue
                m = re.match(r's_cbranch_\w+\s+(\.LBB\w+)', u)
                if m:
                    stack.append((labels[m.group(1)], cnt))
                if u.startswith("s_endpgm") or u.startswith("s_setpc"):
                    break
                k += 1
        # paths that bypass this publication's own arrival (its `if` arm) run on to later half-steps' arrivals: infeasible for the
        # publishing wave, and always longer -- the nearest arrival is this publication's
        if found:
            key = min(found, key=lambda kk: found[kk])
            found = {key: found[key]}
        for (k, n_wait), cnt in found.items():
            total += 1
            if n_wait != cnt:
                bad += 1
                print("%s: publication at +%d, wait vmcnt(%d) at +%d, but the shortest path between them issues %d vector-memory instructions"
                      % (kname, i, n_wait, k, cnt))
        if not found:
            # a publication whose flag is never raised: only the very last one of a sequence (its consumers do not exist)
            pass
    return bad, total


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
    for kname in re.findall(r'^(_Z1[345](?:lstm|gru)_rr_kernel\w+):', txt, re.M):
        a = txt.index("\n" + kname + ":")
        s = txt[a:txt.index(".Lfunc_end", a)].split("\n")
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


def check_pending(txt):
    """The KH = 4 instantiations hand h over WITHOUT flags: a consumer looks at the fragments it fetched and takes a word of 0xffffffff
    for "not written yet" (recurrent_rr.hip, probe_h / settle_h).  That is only sound if every WORD of the three fragments of a k step
    is looked at (nothing is assumed about how a 16-byte store becomes visible).  For each such kernel: every `v_cmp_eq_u32 -1, vN`
    sits on top of a v_max_u32 / v_max3_u32 tree with exactly twelve leaf registers; there are at least 2 x 4 k steps x 2 halves of
    them (the steady-state loop and the peeled last half-steps, fast path + the second look of the slow path); the kernel contains no
    asm vmcnt wait (the flag protocol's drain) and uses no scratch.  (Which registers hold a fragment changes from one peeled copy to
    the next, so the leaves are not traced back to the loads; the bit-for-bit shard / whole-batch tests do that at run time.)"""
    bad = total = 0
    for kname in re.findall(r'^(_Z1[34](?:lstm|gru)_rr_kernelILi4E\w+):', txt, re.M):
        a = txt.index("\n" + kname + ":")
        s = [l.strip() for l in txt[a:txt.index(".Lfunc_end", a)].split("\n")]
        m = re.search(r'\.amdhsa_kernel %s\b.*?\.end_amdhsa_kernel' % re.escape(kname), txt, re.S)
        ps = re.search(r'\.amdhsa_private_segment_fixed_size (\d+)', m.group(0)) if m else None
        if ps and int(ps.group(1)) != 0:
            bad += 1
            print("%s: scratch in use" % kname)
        looks = 0
        for i, t in enumerate(s):
            cm = re.match(r'v_cmp_eq_u32_e\d+ \S+ -1, v(\d+)$', t)
            if not cm:
                continue
            total += 1
            looks += 1
            want = {int(cm.group(1))}
            for j in range(i - 1, max(i - 40, 0), -1):
                mm = re.match(r'v_max3?_u32(?:_e\d+)? v(\d+), (.*)$', s[j])
                if not mm or int(mm.group(1)) not in want:
                    continue
                want.discard(int(mm.group(1)))
                for o in re.split(r',\s*', mm.group(2)):
                    r = regs_of(o.strip())
                    if len(r) == 1:
                        want |= r                  # (a source that no earlier max in the window defines stays in `want`: a leaf)
            if len(want) != 12:
                bad += 1
                print("%s: the look at +%d covers %d registers %s, not the twelve words of three fragments" % (kname, i, len(want), sorted(want)))
        if looks < 32:
            bad += 1
            print("%s: only %d looks" % (kname, looks))
        if any("ASMSTART" in s[k - 1] and re.match(r's_waitcnt vmcnt', t) for k, t in enumerate(s)):
            bad += 1
            print("%s: an asm vmcnt wait (the flag protocol) in a pending-pattern kernel" % kname)
    return bad, total


def kernels(txt, pattern):
    for kname in re.findall(pattern, txt, re.M):
        a = txt.index("\n" + kname + ":")
        yield kname, [l.strip() for l in txt[a:txt.index(".Lfunc_end", a)].split("\n")]


def check_marks(txt):
    """Pending-pattern protocol (recurrent_rr.hip KH = 4, recurrent_fk.hip): a block of step t + 2 (fk: t + 1) is MARKED -- stored full of
    0xffffffff -- long before its data is stored, and the argument why a consumer can never take a stale block of an earlier launch for
    data needs the mark to have LEFT the wavefront (its vmcnt retired) before the same workgroup stores the data of the step in between.
    The prose says "the in-order vmcnt waits of the operand loads the wavefront consumes retire the marks"; this checks it in the ISA: a
    mark group = consecutive write-through stores whose data registers were all set to -1; on EVERY path from it, before the path has
    crossed two workgroup barriers (rr) or reached the next write-through store of other data (fk: the next publication), there is an
    `s_waitcnt vmcnt(N)` with N <= the vector-memory instructions issued on that path since the marks (so none of the <= N operations
    still in flight is a mark)."""
    bad = total = 0
    for kname, ins in list(kernels(txt, r'^(_Z1[34](?:lstm|gru)_rr_kernelILi4E\w+):')) + list(kernels(txt, r'^(_Z1[34](?:lstm|gru)_fk_kernel\w+):')):
        fk = "_fk_kernel" in kname
        labels = {m.group(1): i for i, t in enumerate(ins) for m in [re.match(r'(\.LBB\w+):', t)] if m}
        # registers that are ever set to -1 (flow-insensitive: the marks' data registers are constants, set once ahead of the loop or
        # right before the stores, directly or from an SGPR pair that holds -1)
        minus1, sminus1 = set(), set()
        for t in ins:
            m = re.match(r's_mov_b(?:32|64) (s\d+|s\[\d+:\d+\]), -1$', t)
            if m:
                sminus1 |= {("s", r) for r in regs_of(m.group(1).replace("s", "v"))}
        for t in ins:
            m = re.match(r'v_mov_b(?:32|64)(?:_e32)? (v\d+|v\[\d+:\d+\]), (-1|s\d+|s\[\d+:\d+\])$', t)
            if m and (m.group(2) == "-1" or {("s", r) for r in regs_of(m.group(2).replace("s", "v"))} <= sminus1):
                minus1 |= regs_of(m.group(1))
        def is_mark(t):
            m = re.match(r'buffer_store_dwordx[24] (v\[\d+:\d+\]), .* sc1', t)
            return bool(m) and regs_of(m.group(1)) <= minus1 and len(regs_of(m.group(1))) > 0
        def is_pub(t):
            return re.match(r'buffer_store_dwordx[24] ', t) is not None and " sc1" in t and not is_mark(t)
        for i, t in enumerate(ins):
            if not (is_mark(t) and not is_mark(ins[i - 1])):
                continue
            j = i
            while is_mark(ins[j]) or ins[j].startswith(";") or not ins[j]:
                j += 1
            total += 1
            ok, seen = True, {}
            stack = [(j, 0, 0)]                       # (instruction, vector-memory ops since the marks, barriers crossed)
            while stack and ok:
                k, cnt, bars = stack.pop()
                steps = 0
                while k < len(ins) and steps < 20000:
                    key = (k, bars)
                    if seen.get(key, 1 << 30) <= cnt:
                        break
                    seen[key] = cnt
                    u = ins[k]
                    steps += 1
                    m = re.match(r's_waitcnt .*vmcnt\((\d+)\)', u)
                    if m and int(m.group(1)) <= cnt:
                        break                          # covered on this path
                    if "s_barrier" in u:
                        bars += 1
                        if not fk and bars >= 2:
                            ok = False
                            print("%s: marks at +%d: a path crosses two barriers without a vmcnt wait that covers them" % (kname, i))
                            break
                    if fk and is_pub(u):
                        ok = False
                        print("%s: marks at +%d: a path reaches the next publication (+%d) without a vmcnt wait that covers them" % (kname, i, k))
                        break
                    if is_vmem(u):
                        cnt += 1
                    m = re.match(r's_branch\s+(\.LBB\w+)', u)
                    if m:
                        k = labels[m.group(1)]
                        continue
                    m = re.match(r's_cbranch_\w+\s+(\.LBB\w+)', u)
                    if m:
                        stack.append((labels[m.group(1)], cnt, bars))
                    if u.startswith("s_endpgm") or u.startswith("s_setpc"):
                        break
                    k += 1
            bad += not ok
    return bad, total


def check_split_fma(txt):
    """x = hi + mid + lo must be the split of the ROUNDED x.  Round 4's defect (r04k/t4.log): rr_split_pair inlined behind h = o * tanh(c) and
    hipcc contracted h - hi into fma(o, tanh c, -hi) -- the residual images then described the unrounded product and 5.7 % of the frag3
    elements were one ulp off their f32 twins.  Nothing but a full-size route comparison saw it.  Here: in every kernel of the units that
    split (recurrent_rr.hip, recurrent_fk.hip, frag3.hip), no fused multiply-add (v_fma / v_fmac / v_mad / v_pk_fma) may take as an operand
    a register that holds a bf16 image widened back to f32 (v_lshlrev_b32 .., 16, <cvt result> or v_and_b32 .., 0xffff0000, <cvt result>):
    an image may only be SUBTRACTED from the value it was rounded from."""
    bad = total = 0
    for kname, ins in kernels(txt, r'^(_Z\w+):'):
        cvt, img = set(), set()
        for i, t in enumerate(ins):
            if re.match(r'\.LBB', t):
                cvt, img = set(), set()               # (basic-block local: an image lives a few instructions)
                continue
            ops = re.split(r'[,\s]+', t)
            if len(ops) < 2:
                continue
            dst = regs_of(ops[1])
            srcs = set().union(*[regs_of(o) for o in ops[2:]]) if len(ops) > 2 else set()
            if ops[0].startswith("v_cvt_pk_bf16_f32"):
                total += 1
                cvt |= dst
                img -= dst
                continue
            if re.match(r'v_(fma_f32|fmac_f32|mad_f32|pk_fma_f32|fma_mix)', ops[0]) and (srcs & img):
                bad += 1
                print("%s: `%s` at +%d multiplies-and-adds a bf16 image (v%s) in one rounding" % (kname, t, i, sorted(srcs & img)))
            widened = (ops[0].startswith("v_lshlrev_b32") and len(ops) > 3 and ops[2] == "16" and (regs_of(ops[3]) & cvt)) or \
                      (ops[0].startswith("v_and_b32") and ("0xffff0000" in ops) and (srcs & cvt))
            if widened:
                img |= dst
            else:
                img -= dst
            cvt -= dst
    return bad, total


if __name__ == "__main__":
    sys.exit(main())
