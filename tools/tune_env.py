"""Host-side convenience for tools/ ONLY (the library itself reads no environment variable, include/ofdm_hip.h):
  OFDM_TUNE="key=value,key=value"  -> merged into the tuning of every api.Context created afterwards (ofdm_set_tuning keys)
  OFDM_PROFILE=1                   -> load libofdm_hip_profile.so, the build with the kernels' ablation exits and section timers
                                      (the "debug_*" keys answer OFDM_ERR_UNSUPPORTED in the product build)
Call install() before the first Context is created."""
import os


def tuning() -> dict:
    out = {}
    for item in os.environ.get("OFDM_TUNE", "").split(","):
        if "=" in item:
            k, v = item.split("=", 1)
            out[k.strip()] = int(v)
    return out


def install():
    from ofdm_amd import _lib, api

    if os.environ.get("OFDM_PROFILE"):
        _lib.use_profile_build()
    api.DEFAULT_TUNING.update(tuning())
    return dict(api.DEFAULT_TUNING)
