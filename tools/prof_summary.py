"""Summarise a rocprofv3 kernel trace: per-step wall/busy and time per kernel family (uses sgd_kernel as step delimiter)."""
import csv, glob, collections, sys
d = sys.argv[1]
f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
sg = [i for i, r in enumerate(rows) if "sgd_kernel" in r["Kernel_Name"]]
a, b = sg[-3], sg[-1]
seg = rows[a + 1:b + 1]
t0, t1 = int(seg[0]["Start_Timestamp"]), int(seg[-1]["End_Timestamp"])
busy = sum(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in seg)
print(f"2 steps: wall {(t1 - t0) / 2e6:.2f} ms/step; busy {busy / 2e6:.2f} ms/step; kernels/step {len(seg) / 2:.0f}")
grp, fam = collections.defaultdict(float), collections.defaultdict(float)
for r in seg:
    n = r["Kernel_Name"]; t = (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 2e6
    k = n.split("(")[0].replace("void ", "")
    grp[k] += t
    if "gemm_kernel<" in n:  # gemm_kernel<OpT, BM, BN, B_IM2COL, SWAP, KT>
        args = n.split("gemm_kernel<")[1].split(">")[0].split(",")
        fam["gemm_im2col" if args[3].strip() == "true" else "gemm_plain"] += t
    else:
        fam["flash" if "flash" in n else "other"] += t
print({k: round(v, 2) for k, v in fam.items()})
top = int(sys.argv[2]) if len(sys.argv) > 2 else 40
for k, t in sorted(grp.items(), key=lambda kv: -kv[1])[:top]:
    print(f"{t:7.2f}  {k[:100]}")

# gaps between consecutive kernels of the same hardware queue inside the two steps (hipGraph replay: dependency edges
# between nodes; eager: launch gaps)
byq = collections.defaultdict(list)
for r in seg:
    byq[r.get("Queue_Id", "0")].append((int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
for q, iv in sorted(byq.items(), key=lambda kv: -len(kv[1])):
    iv.sort()
    gaps = [max(0, iv[i + 1][0] - iv[i][1]) for i in range(len(iv) - 1)]
    qb = sum(e - s for s, e in iv)
    gs = sorted(gaps)
    if not gs:
        continue
    print(f"queue {q}: {len(iv) / 2:.0f} kernels/step, busy {qb / 2e6:.2f} ms/step, gaps {sum(gaps) / 2e6:.2f} ms/step "
          f"(median {gs[len(gs) // 2] / 1e3:.1f} us, p90 {gs[int(len(gs) * .9)] / 1e3:.1f} us, max {gs[-1] / 1e3:.0f} us)")
