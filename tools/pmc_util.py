"""VALU / LDS occupancy of the timed kernels from the SQ passes of tools/profile_round.sh (profiles/rNN_pmc_sq_summary.tsv):
    valu_util = SQ_ACTIVE_INST_VALU x 4 cycles / (SIMD-cycles of the run) = ACTIVE_VALU / (8 x SQ_BUSY_CYCLES)
(SQ_BUSY_CYCLES is summed over the 32 shader engines, the chip has 1024 SIMDs; a wave64 VALU instruction occupies its SIMD for one
quad-cycle = 4 cycles, f64 ones for two: SQ_ACTIVE_INST_VALU / SQ_INSTS_VALU shows the mix).  python tools/pmc_util.py <tsv>"""
import collections, sys
d = collections.defaultdict(dict)
for line in open(sys.argv[1]):
    name, ctr, calls, mean, mx = line.rstrip("\n").split("\t")[:5]
    d[name.replace("void ofdm::", "").replace("ofdm::", "")][ctr] = float(mx.split("=")[1])   # the largest call (list-mode launches run near-empty)
print("kernel\tvalu_util\tquad_cycles_per_valu_inst\tlds_util\tlds_bank_conflict_share\twait_share_of_wave_cycles")
for k, c in sorted(d.items()):
    if "SQ_BUSY_CYCLES" not in c or c["SQ_BUSY_CYCLES"] < 1e6: continue
    print(f"{k[:48]}\t{c['SQ_ACTIVE_INST_VALU'] / (8 * c['SQ_BUSY_CYCLES']):.2f}\t{c['SQ_ACTIVE_INST_VALU'] / max(1.0, c['SQ_INSTS_VALU']):.2f}\t"
          f"{c['SQ_ACTIVE_INST_LDS'] / (8 * c['SQ_BUSY_CYCLES']):.2f}\t{c['SQ_LDS_BANK_CONFLICT'] / max(1.0, c['SQ_LDS_IDX_ACTIVE']):.2f}\t"
          f"{c['SQ_WAIT_ANY'] / max(1.0, c['SQ_WAVE_CYCLES']):.2f}")
