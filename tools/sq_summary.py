"""Summarise one rocprofv3 SQ counter pass over tools/dattn_bench.py (tools/refresh_profiles.sh, last step) per kernel instance
and grid: fractions of SQ_WAVE_CYCLES parked (WAIT_ANY), issue-stalled (WAIT_INST_ANY), issuing (ACTIVE_INST_ANY), issuing VALU
(ACTIVE_INST_VALU); matrix-pipe utilisation = SQ_VALU_MFMA_BUSY_CYCLES (cycles, summed over the SIMDs: MI355X_MICROARCH.md) /
(dispatch duration x 2.4 GHz x 1024 SIMDs) — a LOWER bound, the chip clocks below 2.4 GHz under load.   python tools/sq_summary.py <counter_collection.csv>"""
import collections
import csv
import sys

rows = collections.defaultdict(lambda: collections.defaultdict(float))
meta = {}
for r in csv.DictReader(open(sys.argv[1])):
    n = r["Kernel_Name"].replace("void ", "").split("(")[0]
    if not n.startswith(("dattn_", "sra_", "flashc_", "conv_wgrad", "conv_direct")):
        continue
    key = (n, int(r["Grid_Size"]))
    rows[key][r["Counter_Name"]] += float(r["Counter_Value"])
    rows[key]["_n_" + r["Counter_Name"]] += 1
    if r["Counter_Name"] == "SQ_WAVE_CYCLES":
        rows[key]["_dur_ns"] += float(r["End_Timestamp"]) - float(r["Start_Timestamp"])
    meta[key] = r["VGPR_Count"]
print("kernel | grid | vgpr | dispatches | us per dispatch | wait_any | wait_inst | active_any | active_valu | mfma util (>=) | insts_valu")
for key, c in rows.items():
    wc = c["SQ_WAVE_CYCLES"]
    if not wc:
        continue
    dur = c["_dur_ns"]
    nd = int(c["_n_SQ_WAVE_CYCLES"])
    util = c["SQ_VALU_MFMA_BUSY_CYCLES"] / (dur * 2.4 * 1024) if dur else float("nan")
    print(f"{key[0]} | {key[1]} | {meta[key]} | {nd} | {dur / nd / 1e3:.0f} | {c['SQ_WAIT_ANY'] / wc:.2f} | "
          f"{c['SQ_WAIT_INST_ANY'] / wc:.2f} | {c['SQ_ACTIVE_INST_ANY'] / wc:.2f} | {c['SQ_ACTIVE_INST_VALU'] / wc:.2f} | "
          f"{util:.3f} | {c['SQ_INSTS_VALU']:.3g}")
