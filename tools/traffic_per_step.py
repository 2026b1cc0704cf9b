"""HBM / fabric bytes of ONE training step from the committed profiles: per-launch bytes of every kernel (profiles/<tag>_traffic.json,
2 x FETCH_SIZE + WRITE_SIZE from the two PMC passes) x launches per step (profiles/<tag>_c_kernel_stats_noroofline.csv).
python tools/traffic_per_step.py r04 [r03 ...]"""
import csv
import json
import sys

for tag in sys.argv[1:] or ["r04"]:
    tr = json.load(open(f"profiles/{tag}_traffic.json"))
    rows = list(csv.DictReader(open(f"profiles/{tag}_c_kernel_stats_noroofline.csv")))
    steps = [int(r["Calls"]) for r in rows if r["Name"].startswith("sgd_kernel")][0]
    tot, miss, fam, n = 0.0, 0.0, {}, 0.0
    for r in rows:
        name = r["Name"].replace("void ", "").split("(")[0]
        calls = int(r["Calls"]) / steps
        e = tr.get(name)
        if e is None:
            miss += float(r["TotalDurationNs"]) / steps
            continue
        b = e["hbm_bytes_per_launch"] * calls
        tot += b
        n += calls
        f = name.split("<")[0]
        fam[f] = fam.get(f, 0.0) + b
    ms = sum(float(r["TotalDurationNs"]) for r in rows) / steps / 1e6
    print(f"{tag}: {tot / 1e9:.1f} GB per step over {n:.0f} launches ({ms:.2f} ms of kernels -> {tot / ms / 1e9:.2f} TB/s average); "
          f"kernels without a traffic entry: {miss / 1e3:.0f} us per step")
    for k, v in sorted(fam.items(), key=lambda kv: -kv[1])[:14]:
        print(f"    {v / 1e9:6.2f} GB  {k}")
