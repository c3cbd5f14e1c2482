"""CPU-only: the C-ABI library loads, exports every symbol the header declares, and
its by-value struct layouts / config geometry / window functions equal what the
REAL reference returns (tests/golden/ref_probe.json, produced by oracle/ref_probe.c
from the reference's own sources compiled in place -- `make -C oracle ref`)."""
import ctypes as C
import json
import os
import re

import numpy as np
import pytest

from nntoolkitcore_amd import capi

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_probe.json")))

STRUCTS = {
    "SpectrogramConfig": capi.SpectrogramConfig, "Conv1dConfig": capi.Conv1dConfig,
    "BatchNormConfig": capi.BatchNormConfig, "RecurrentConfig": capi.RecurrentConfig,
    "GRUActivations": capi.GRUActivations, "GRUConfig": capi.GRUConfig,
    "LSTMActivations": capi.LSTMActivations, "LSTMConfig": capi.LSTMConfig,
    "RNNConfig": capi.RNNConfig,
    "DenseConfig": capi.DenseConfig, "TimeDistributedDenseConfig": capi.TimeDistributedDenseConfig,
    "DefaultWeights": capi.DefaultWeights, "RecurrentWeights": capi.RecurrentWeights,
    "BatchNormWeights": capi.BatchNormWeights, "MelFilterBankConfig": capi.MelFilterBankConfig,
}


def test_library_exports_every_declared_symbol(built_lib):
    header = open(os.path.join(ROOT, "include", "nntoolkitcore_hip.h")).read()
    header = re.sub(r"/\*.*?\*/", "", header, flags=re.S)
    header = re.sub(r"typedef[^;{]*(\{[^}]*\})?[^;]*;", "", header)      # drop typedefs (incl. fn-pointer types)
    declared = set(re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", header))
    assert len(declared) > 90
    missing = [s for s in sorted(declared) if not hasattr(built_lib, s)]
    assert not missing, missing
    assert declared == set(capi.SIGNATURES), declared ^ set(capi.SIGNATURES)


def test_struct_layouts_match_reference_headers():
    abi = GOLD["abi"]
    for name, st in STRUCTS.items():
        assert C.sizeof(st) == abi["sizeof_" + name], name
        for field, _ in st._fields_:
            key = "offsetof_%s_%s" % (name, field)
            if key in abi:
                assert getattr(st, field).offset == abi[key], key


def test_product_headers_compile_as_c_and_match_layouts(tmp_path):
    """Compile a C probe against include/ (the drop-in headers at the reference's include paths)."""
    src = tmp_path / "p.c"
    src.write_text(r'''
#include <stdio.h>
#include <stddef.h>
#include "nntoolkitcore/layers/conv_1d.h"
#include "nntoolkitcore/layers/batch_norm.h"
#include "nntoolkitcore/layers/gru.h"
#include "nntoolkitcore/layers/lstm.h"
#include "nntoolkitcore/layers/time_distributed_dense.h"
#include "nntoolkitcore/layers/activation_default.h"
#include "nntoolkitcore/signal/spectrogram.h"
int main(void){
 printf("%zu %zu %zu %zu %zu %zu %zu %zu %zu %zu\n", sizeof(SpectrogramConfig), sizeof(Conv1dConfig), sizeof(BatchNormConfig),
   sizeof(RecurrentConfig), sizeof(GRUConfig), sizeof(LSTMConfig), sizeof(DenseConfig), sizeof(TimeDistributedDenseConfig),
   offsetof(LSTMConfig, activations), offsetof(LSTMActivations, output_activation));
 return 0; }''')
    exe = tmp_path / "p"
    import subprocess
    subprocess.check_call(["gcc", "-std=c11", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe)])
    got = [int(v) for v in subprocess.check_output([str(exe)]).split()]
    a = GOLD["abi"]
    want = [a["sizeof_SpectrogramConfig"], a["sizeof_Conv1dConfig"], a["sizeof_BatchNormConfig"], a["sizeof_RecurrentConfig"],
            a["sizeof_GRUConfig"], a["sizeof_LSTMConfig"], a["sizeof_DenseConfig"], a["sizeof_TimeDistributedDenseConfig"],
            a["offsetof_LSTMConfig_activations"], a["offsetof_LSTMActivations_output_activation"]]
    assert got == want


def test_config_geometry_matches_reference(built_lib):
    for c in GOLD["spectrogram_config"]:
        got = built_lib.SpectrogramConfigCreate(c["nfft"], c["window_size"], c["noverlap"], c["input_size"], 1.0)
        assert (got.step, got.nfreq, got.ntime_series) == (c["step"], c["nfreq"], c["ntime_series"]), c
    for c in GOLD["conv1d_config"]:
        got = built_lib.Conv1dConfigCreate(c["cin"], c["cout"], c["k"], c["stride"], c["input_size"])
        assert got.output_size == c["output_size"], c
    r = built_lib.RecurrentConfigCreate(3, 4, True, 9)
    assert (r.input_feature_channels, r.output_feature_channels, r.return_sequences, r.timesteps) == (3, 4, True, 9)


@pytest.mark.parametrize("name,fn", [("ones", "ones"), ("hann", "hann_window"), ("hamming", "hamming_window"),
                                     ("periodic_hann", "periodic_hann_window"),
                                     ("periodic_hamming", "periodic_hamming_window"), ("blackman", "blackman_window")])
def test_window_functions_bit_exact_vs_reference(built_lib, name, fn):
    for key, n in (("windows16", 16), ("windows400", 400)):
        v = np.empty(n, np.float32)
        getattr(built_lib, fn)(v.ctypes.data_as(capi.fp), n)
        want = np.array(GOLD[key][name], np.float32)
        assert np.array_equal(v, want), (name, n)


def test_no_gpu_means_loud_failure_not_fallback(built_lib):
    """Without a device, creating a layer fails and reports why; nothing computes on the CPU."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    h = built_lib.Conv1dCreateForInference(built_lib.Conv1dConfigCreate(1, 16, 9, 1, 100))
    assert not h
    assert "HIP error" in capi.last_error()


def test_product_never_references_the_oracle():
    """The product path must not import, link or call anything under oracle/."""
    import subprocess
    pkg = os.path.join(ROOT, "nntoolkitcore_amd")
    for dirpath, _, files in os.walk(pkg):
        for f in files:
            if f.endswith((".py", ".c", ".h", ".hip", ".hpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "oracle" not in text.lower(), os.path.join(dirpath, f)
    syms = subprocess.check_output(["nm", "-D", capi.LIB_PATH], text=True)
    assert "ref_" not in syms


def test_mel_filterbank_weights_match_oracle_restatement(built_lib):
    """MelFilterBankCreate is pure host setup (no device): its matrix must equal the oracle's restatement of
    signal/mel_filterbank.c:43-102 bit for bit (same formulas, same libm)."""
    import oracle as O
    built_lib.nntk_mel_weights.restype = capi.fp
    built_lib.nntk_mel_weights.argtypes = [C.c_void_p]
    for (n_mels, n_fft, sr, lo, hi) in [(40, 512, 16000, 20.0, 8000.0), (13, 256, 8000, 0.0, 4000.0), (64, 1024, 44100, 50.0, 20000.0)]:
        bank = built_lib.MelFilterBankCreate(built_lib.MelFilterBankConfigCreate(n_mels, n_fft, sr, lo, hi))
        nb = n_fft // 2 + 1
        w = np.ctypeslib.as_array(built_lib.nntk_mel_weights(bank), shape=(nb, n_mels)).copy()
        assert np.array_equal(w, O.mel_filterbank_weights(n_mels, n_fft, sr, lo, hi))
        assert (w[0] == 0).all() and (w >= 0).all()
        built_lib.MelFilterBankDestroy(bank)


def test_current_stream_and_error_string_are_per_host_thread(built_lib):
    """SURVEY 8(b) Threading: no process-global mutable state a second thread could trip over.  Needs no GPU: the
    stream pointer is only stored."""
    import ctypes as C
    import threading
    L = built_lib
    L.nntk_hip_set_stream(C.c_void_p(0x1000))
    seen = {}

    def other():
        seen["initial"] = L.nntk_hip_get_stream()          # a fresh thread starts on the default stream
        L.nntk_hip_set_stream(C.c_void_p(0x2000))
        seen["own"] = L.nntk_hip_get_stream()
        L.nntk_hip_set_option(b"no_such_option", b"1")      # leaves an error string in THIS thread only
        seen["err"] = L.nntk_last_error().decode()

    t = threading.Thread(target=other)
    t.start(); t.join()
    assert seen["initial"] in (None, 0) and seen["own"] == 0x2000 and "unknown option" in seen["err"]
    assert L.nntk_hip_get_stream() == 0x1000                # untouched by the other thread
    assert L.nntk_last_error().decode() == ""
    L.nntk_hip_set_stream(None)


def test_options_by_name(built_lib):
    from nntoolkitcore_amd import capi
    assert capi.get_option("rec_persistent") == -1 and capi.get_option("weights_check") == -1
    capi.set_option("rec_persistent", 0)
    assert capi.get_option("rec_persistent") == 0
    capi.set_option("rec_persistent", "auto")
    assert capi.get_option("rec_persistent") == -1
    assert capi.get_option("rec_spin_us") == 1000000
    import pytest
    with pytest.raises(capi.NNTKError):
        capi.set_option("nope", 1)


def test_boundary_behaviour_pinned_by_the_real_reference_round2(built_lib):
    """tests/golden/ref_probe.json, round-2 additions: results of the REAL reference's op-free functions
    (oracle/ref_probe.c).  The product must behave the same at the C boundary."""
    import ctypes as C
    import json
    import os
    from nntoolkitcore_amd import capi
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "ref_probe.json")))
    L = built_lib
    # a custom ActivationFunction is the caller's host function: called with (implementer, input, output, size-at-create);
    # Destroy hands the implementer to the caller's destroy function (activation.c:23-45)
    assert gold["custom_activation_callback"] == {"implementer_passed": 1, "input_passed": 1, "output_passed": 1, "size": 7,
                                                   "destroy_called_with_implementer": 1}
    seen = {}
    token = C.c_int(42)

    @capi.ACT_IMPL_FN
    def cb(impl, i, o, n):
        seen.update(impl=impl, inp=C.addressof(i.contents), out=C.addressof(o.contents), size=n)

    DTOR = C.CFUNCTYPE(None, C.c_void_p)

    @DTOR
    def dtor(p):
        seen["destroyed_with"] = p

    x = np.arange(8, dtype=np.float32)
    y = np.full(8, -1, np.float32)
    h = L.ActivationFunctionCreate(7, C.cast(dtor, C.c_void_p), C.cast(C.pointer(token), C.c_void_p), C.cast(cb, C.c_void_p), None, None)
    L.ActivationFunctionApply(h, x.ctypes.data_as(capi.fp), y.ctypes.data_as(capi.fp))
    L.ActivationFunctionDestroy(h)
    assert seen["impl"] == C.addressof(token) and seen["inp"] == x.ctypes.data and seen["out"] == y.ctypes.data
    assert seen["size"] == gold["custom_activation_callback"]["size"] and seen["destroyed_with"] == C.addressof(token)
    # MelFilterBankConfig: by-value struct, field order and size are ABI (mel_filterbank.h:14-20)
    m = L.MelFilterBankConfigCreate(40, 512, 16000, C.c_float(20.0), C.c_float(8000.0))
    g = gold["mel_config"]
    assert (m.n_mels, m.n_fft, m.sample_rate, m.lower_hz, m.upper_hz, C.sizeof(capi.MelFilterBankConfig)) == \
        (g["n_mels"], g["n_fft"], g["sample_rate"], g["lower_hz"], g["upper_hz"], g["sizeof"])
    # recorded for INTEGRATION.md: the reference returns -1 from *ApplyInference on a training-mode handle; the
    # product has no training-mode handles (out of scope), its -1 cases are NULL handles and device errors
    assert set(gold["wrong_mode_apply_inference"].values()) == {-1}


def test_every_function_of_the_reference_headers_is_declared_and_exported(built_lib):
    """Drop-in completeness: every function the reference's layers/, signal/ and train/ headers declare is declared in
    include/nntoolkitcore_hip.h and exported by the library (the reference is only present in the build container)."""
    import glob
    import re
    ref_root = "/root/reference/nntoolkitcore"
    if not os.path.isdir(ref_root):
        pytest.skip("reference tree not present")
    names = set()
    for sub in ("layers", "signal", "train"):
        for f in glob.glob(os.path.join(ref_root, sub, "*.h")):
            t = re.sub(r"/\*.*?\*/", "", open(f).read(), flags=re.S)
            t = re.sub(r"//.*", "", t)
            names |= {m.group(1) for m in re.finditer(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", t)}
    names -= {"defined", "sizeof", "void", "float", "int"}           # typedef'd function pointers: `void (*Name)(...)`
    assert len(names) > 100
    header = open(os.path.join(ROOT, "include", "nntoolkitcore_hip.h")).read()
    undeclared = sorted(n for n in names if not re.search(r"\b" + n + r"\s*\(", header))
    assert not undeclared, undeclared
    missing = sorted(n for n in names if not hasattr(built_lib, n))
    assert not missing, missing
