"""Turns two rocprofv3 counter-collection passes (FETCH_SIZE, WRITE_SIZE; separate runs of the same command, as the TCC
block cannot hold both) into per-kernel HBM traffic per launch, with the gfx950 correction of MI355X_MICROARCH.md §HBM
(FETCH_SIZE counts 64 B per 128-B request: x2) and the counters' unit (KB).
    python tools/pmc_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [source note]
Writes {kernel name as rocprofv3 prints it, without "void " and the argument list: {hbm_bytes_per_launch, fetch_kb_raw,
write_kb, launches, source}} for every kernel; bench.py looks its dominant kernel up in profiles/r02_traffic.json."""
import collections
import csv
import json
import sys


def load(path, counter):
    tot, n = collections.defaultdict(float), collections.Counter()
    for r in csv.DictReader(open(path)):
        if r.get("Counter_Name") != counter:
            continue
        k = r["Kernel_Name"].replace("void ", "").split("(")[0].strip()
        tot[k] += float(r["Counter_Value"])
        n[k] += 1
    return tot, n


f, nf = load(sys.argv[1], "FETCH_SIZE")
w, nw = load(sys.argv[2], "WRITE_SIZE")
note = sys.argv[4] if len(sys.argv) > 4 else ""
out = {}
for k in f:
    if k not in w:
        continue
    fk, wk = f[k] / nf[k], w[k] / nw[k]
    out[k] = {"hbm_bytes_per_launch": int((2.0 * fk + wk) * 1024), "fetch_kb_raw_per_launch": round(fk, 1),
              "write_kb_per_launch": round(wk, 1), "launches_in_pass": nf[k],
              "source": ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE passes, 2 x FETCH + WRITE (gfx950 correction); " + note).strip()}
# which build these counters belong to: the hash of the kernel sources (bench.py reports the same hash of the sources it runs
# on, so a reader can tell whether `roofline.traffic` was measured on the build being benched)
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from cenet_amd.build import source_sha16  # noqa: E402
out["_meta"] = {"kernel_src_sha16": source_sha16()}
json.dump(out, open(sys.argv[3], "w"), indent=1, sort_keys=True)
out.pop("_meta")
top = sorted(out.items(), key=lambda kv: -kv[1]["hbm_bytes_per_launch"] * kv[1]["launches_in_pass"])[:12]
for k, v in top:
    print(f"{v['hbm_bytes_per_launch'] / 1e6:9.2f} MB/launch x {v['launches_in_pass']:5d}  {k}")
