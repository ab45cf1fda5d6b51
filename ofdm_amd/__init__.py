"""ofdm_amd -- MI355X-native OFDM modulate / demodulate hot path (HIP kernels behind a C ABI).

`load()` loads libofdm_hip.so (building it with hipcc when missing) and raises if that fails: there is no
CPU path in this package.  See include/ofdm_hip.h for the boundary and ofdm_amd/api.py for the host mirror
of the reference's function surface.
"""
from ._lib import load, LIB_PATH, Params, SIGNATURES  # noqa: F401


def __getattr__(name):
    # api imports torch and touches the GPU lazily; keep `import ofdm_amd` cheap
    if name in ("api", "Context", "encode", "decode", "decode_long", "pinned_empty", "channel", "default_pilots", "locking_signal", "preamble",
                "training_signals", "OfdmError", "DecodeError", "BPSK", "QPSK", "QAM16", "QAM64", "QAM256",
                "ECC_NONE", "ECC_HAMMING74", "CFO_OFF", "CFO_SIGNED", "CFO_ABS", "SYNC_SCHMIDL_COX", "SYNC_REFERENCE"):
        import importlib

        api = importlib.import_module(".api", __name__)
        return api if name == "api" else getattr(api, name)
    raise AttributeError(name)
