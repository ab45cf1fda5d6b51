"""CPU-only checks of the drop-in boundary: the C-ABI library builds, loads, exports every symbol that
include/ofdm_hip.h declares, and its host-side entry points behave.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_functions():
    txt = open(os.path.join(ROOT, "include", "ofdm_hip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(ofdm_[a-z0-9_]+)\s*\(", txt)))


@pytest.fixture(scope="module")
def lib():
    from ofdm_amd import build

    return C.CDLL(build.build())


def test_header_and_loader_agree(lib):
    import ofdm_amd

    names = header_functions()
    assert len(names) >= 35
    assert sorted(ofdm_amd.SIGNATURES) == names  # the ctypes table binds exactly the declared surface
    for n in names:
        assert hasattr(lib, n), f"libofdm_hip.so does not export {n}"


def test_no_oracle_or_torch_in_the_boundary(lib):
    # the library is self-contained: it neither links the oracle nor exposes C++/torch types
    import subprocess

    import ofdm_amd

    needed = subprocess.run(["readelf", "-d", ofdm_amd.LIB_PATH], capture_output=True, text=True).stdout
    assert "liboracle" not in needed and "torch" not in needed and "libamdhip64" in needed
    for root, _, files in os.walk(os.path.join(ROOT, "ofdm_amd")):
        for f in files:
            if f.endswith((".py", ".hip", ".hpp")):
                src = open(os.path.join(root, f)).read()
                assert "import oracle" not in src and "from oracle" not in src and "ofdm_oracle" not in src, f


def test_host_entry_points(lib, orc):
    from ofdm_amd import Params

    lib.ofdm_strerror.restype = C.c_char_p
    assert lib.ofdm_abi_version() == 1
    assert lib.ofdm_strerror(0) == b"ok" and lib.ofdm_strerror(-1) == b"invalid argument"
    p = Params()
    assert lib.ofdm_default_params(C.byref(p)) == 0
    # reference defaults (src/transmitter.rs:16-17,33): 64 carriers, CP 16, Bpsk, no guard bands
    assert (p.n_fft, p.cp_len, p.modulation, p.guard_bands, p.ecc) == (64, 16, 1, 0, 0)
    assert (p.sync_window_reps, p.sync_backoff, p.cfo_mode) == (3, 4, 1) and abs(p.sync_threshold - 0.5) < 1e-7
    # default pilot tables are the documented SplitMix64 draws: identical to the oracle's independent restatement
    for n in (64, 1024):
        pre = np.zeros(n + n // 4, np.complex128)
        trn = np.zeros(n, np.complex128)
        assert lib.ofdm_default_pilots(n, n // 4, C.c_void_p(pre.ctypes.data), C.c_void_p(trn.ctypes.data)) == 0
        np.testing.assert_array_equal(pre, orc.default_preamble(n + n // 4))
        np.testing.assert_array_equal(trn, orc.default_training(n))
        assert np.abs(pre.real).max() <= 0.25 and np.abs(trn.real).max() <= 1.0
    assert lib.ofdm_default_pilots(100, 25, None, None) == -1


def test_create_rejects_bad_params_and_missing_gpu(lib):
    from ofdm_amd import Params

    h = C.c_void_p()
    p = Params()
    lib.ofdm_default_params(C.byref(p))
    bad = []
    for field, value in (("n_fft", 96), ("cp_len", 8), ("modulation", 3), ("guard_bands", 2), ("ecc", 7),
                         ("sync_window_reps", 0), ("cfo_mode", 9)):
        q = Params.from_buffer_copy(p)
        setattr(q, field, value)
        bad.append(lib.ofdm_create(C.byref(q), None, None, 0, None, C.byref(h)))
    assert bad == [-1] * len(bad)
    import torch

    if not torch.cuda.is_available():  # authoring container: no device -> a status code, never an abort
        assert lib.ofdm_create(C.byref(p), None, None, 0, None, C.byref(h)) == -3
        assert not h.value
    cnt = C.c_int(-1)
    assert lib.ofdm_device_count(C.byref(cnt)) in (0, -3)


def test_product_path_fails_loudly_without_gpu():
    import torch

    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from ofdm_amd import api

    with pytest.raises(api.OfdmError):
        api.Context()
    with pytest.raises(api.OfdmError):
        api.encode(b"abc")


def test_host_constants_match_reference_closed_forms(orc):
    from ofdm_amd import api

    np.testing.assert_array_equal(api.locking_signal(80), orc.locking_signal(80))  # transmitter.rs:60-72
    assert api.locking_signal(80)[0] == 0.375
    np.testing.assert_array_equal(api.preamble(80), orc.default_preamble(80))
    np.testing.assert_array_equal(api.training_signals(64), orc.default_training(64))
