"""Build-time ISA lint (no GPU): the compiled kernels must not contain the wide-store hazard the compiler does not
cover on gfx950 (tools/check_store_hazard.py) -- found the hard way in conv1d.hip's epilogue, where it corrupted four
lanes per 16 of some stores depending on timing."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_scanner_flags_the_hazard_pattern():
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    import check_store_hazard as L
    bad = """_Z1kv:
\tbuffer_store_dwordx4 v[64:67], v68, s[16:19], s10 offen
\tv_or_b32_e32 v64, s12, v96
"""
    ok_imm = bad.replace("s10 offen", "0 offen")
    ok_nop = bad.replace("\tv_or_b32", "\ts_nop 1\n\tv_or_b32")
    ok_other = bad.replace("v_or_b32_e32 v64", "v_or_b32_e32 v70")
    assert len(L.scan(bad)) == 1 and L.scan(bad)[0][0] == "_Z1kv"
    assert L.scan(ok_imm) == [] and L.scan(ok_nop) == [] and L.scan(ok_other) == []


def test_no_kernel_has_the_wide_store_sgpr_soffset_hazard():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_store_hazard.py")], capture_output=True, text=True,
                       timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr


def test_lstm_rr_counted_waits_match_the_isa():
    """recurrent_rr.hip raises a half's flag after a COUNTED vmcnt wait (the operand prefetch stays in flight): the count must
    equal the vector-memory instructions the compiler really emitted between the publication and the wait, in every
    instantiation and every peeled copy of the half-step (dead x loads in the last half-steps once made it 2 KX too lenient)."""
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "check_rr_waits.py")], capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout
    # ... and the flag poll's register is left alone between its asm load and its asm wait, no scratch in any rr kernel
    assert "0 violations" in r.stdout and " 0 flag polls" not in r.stdout
    # ... and the KH = 4 instantiations, which hand over without flags (pending pattern): every look covers all twelve words of a
    # k step's three fragments, no flag-protocol drain is left in them, no scratch
    assert "0 defects" in r.stdout and " 0 pending-pattern looks" not in r.stdout
    # ... and no kernel of these units (rr, fk, dense_frag3, conv1d) has a private segment: a spill still passes every test
    assert "0 with a private segment" in r.stdout
