"""Prints the ms / fraction rows of a bench_detail.json (tools/bench_cfg3.py, tools/bench_large_n.py blocks)."""
import json, sys
d = json.load(open(sys.argv[1]))
def row(name, b):
    if not isinstance(b, dict): return
    ms = b.get("ms", b.get("kernel_ms", b.get("ms_per_pass")))
    if ms is None: return
    ct = (b.get("capture_throughput") or {}).get("of_hbm_peak")
    fr = (b.get("roofline") or {}).get("frac")
    print(f"{name:58s} {ms:8.4f} ms  capture {ct if ct is None else round(ct, 4)}  roofline {fr if fr is None else round(fr, 4)}  {b.get('dispatch', '')}")
print("headline", d.get("value"), d.get("ms_per_step"), (d.get("roofline") or {}).get("frac"))
for cfg in ("cfg3", "cfg4", "cfg5"):
    c = d.get(cfg) or {}
    if "error" in c: print(cfg, "ERROR", c["error"])
    for k, v in c.items():
        row(cfg + "." + k, v)
        if k == "late_packets" and isinstance(v, dict):
            for k2, v2 in v.items(): row(cfg + ".late." + k2, v2)
            print(cfg, "late cpu_check", (v.get("cpu_check") or {}).get("gpu_bytes_equal_cpu_bytes"), "decoded", v.get("frames_decoded"))
    if cfg == "cfg5" and c: print("cfg5 tx", c.get("tx_ms"), c.get("roofline_tx", {}).get("frac"), "rx", c.get("rx_ms"), c.get("roofline_rx", {}).get("frac"))
    if "cpu_baseline" in c: print(cfg, "gpu==cpu", c["cpu_baseline"].get("gpu_bytes_equal_cpu_bytes"), "decoded", c.get("frames_decoded"))
for sh in d.get("shapes") or []:
    if isinstance(sh, dict): print("shape", sh.get("n_fft"), "tx", sh.get("tx_frac"), "rx", sh.get("rx_frac"), "enc", sh.get("encode_frac"))
