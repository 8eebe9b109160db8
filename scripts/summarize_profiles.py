#!/usr/bin/env python3
"""gpurun_out/prof_r01 (scratch) -> profiles/ (tracked): kernel stats, PMC summary per dispatch, traffic.json."""
import collections, csv, json, os, shutil, sys
base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r01"
rnd = sys.argv[2] if len(sys.argv) > 2 else "r01"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{base}/single/s_kernel_stats.csv", f"profiles/{rnd}_single_stream_kernel_stats.csv")
shutil.copy(f"{base}/default/d_kernel_stats.csv", f"profiles/{rnd}_default_kernel_stats.csv")
shutil.copy(f"{base}/bench_default.json", f"profiles/{rnd}_bench_line.json")

def agg(f):
    rows = list(csv.DictReader(open(f)))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]; a[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(nd[k]) for c, v in d.items()} for k, d in a.items()}

f = agg(f"{base}/pmc_FETCH_SIZE/p_counter_collection.csv"); w = agg(f"{base}/pmc_WRITE_SIZE/p_counter_collection.csv")
s1 = agg(f"{base}/pmc_SQ_WAVES/p_counter_collection.csv"); s2 = agg(f"{base}/pmc_SQ_INSTS_MFMA/p_counter_collection.csv")
c1 = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
c2 = ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "GRBM_GUI_ACTIVE"]
out = ["# PMC summary per dispatch (rocprofv3 --pmc, one counter group per run; scripts/prof_r01.sh)",
       "# command: python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline  (B=64, T=1000, live baseline_m, fp32)",
       "# FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of a wide",
       "# coalesced streaming read -> hbm_read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE taken as is.",
       "kernel,FETCH_SIZE_KiB,hbm_read_MB_corrected,WRITE_SIZE_KiB,hbm_write_MB," + ",".join(c1 + c2)]
for k in f:
    if not any(t in k for t in ("gsn_scan", "spike_proj", "input_proj", "features_kernel", "deepfilter", "rowsum", "laplace")):
        continue
    fs = f[k].get("FETCH_SIZE", 0); ws = w.get(k, {}).get("WRITE_SIZE", 0)
    vals = [s1.get(k, {}).get(c, "") for c in c1] + [s2.get(k, {}).get(c, "") for c in c2]
    out.append('"%s",%.0f,%.1f,%.0f,%.1f,%s' % (k.replace('"', ""), fs, 2 * fs * 1024 / 1e6, ws, ws * 1024 / 1e6,
                                                ",".join("%.4g" % v if v != "" else "" for v in vals)))
open(f"profiles/{rnd}_pmc_summary.csv", "w").write("\n".join(out) + "\n")
# HBM traffic of one whole forward: every kernel's (2*FETCH + WRITE) per dispatch x its dispatches per forward
def counts(fn):
    c = collections.Counter()
    seen = set()
    for r in csv.DictReader(open(fn)):
        key = (r["Kernel_Name"], r["Dispatch_Id"])
        if key not in seen:
            seen.add(key); c[r["Kernel_Name"]] += 1
    return c
jf, jw = agg(f"{base}/job_FETCH_SIZE/p_counter_collection.csv"), agg(f"{base}/job_WRITE_SIZE/p_counter_collection.csv")
cf = counts(f"{base}/job_FETCH_SIZE/p_counter_collection.csv")
n_fwd = max(v for kk, v in cf.items() if "deepfilter" in kk)
job = 0.0
for kk, v in cf.items():
    if any(t in kk for t in ("gsn_scan", "spike_proj", "input_proj", "features", "deepfilter", "rowsum", "laplace", "spike_count")):
        job += (2 * jf[kk].get("FETCH_SIZE", 0) + jw.get(kk, {}).get("WRITE_SIZE", 0)) * 1024 * v / n_fwd
k = [x for x in f if "gsn_scan_kernel<1, 4, 16" in x][0]
tr = dict(B=64, T=1000, kernel=k, fetch_size_KiB=f[k]["FETCH_SIZE"], write_size_KiB=w[k]["WRITE_SIZE"],
          sb_scan_hbm_bytes_per_launch=int(2 * f[k]["FETCH_SIZE"] * 1024 + w[k]["WRITE_SIZE"] * 1024),
          forward_hbm_bytes=int(job),
          note=f"2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per dispatch (gfx950 FETCH_SIZE half-count correction); source profiles/{rnd}_pmc_summary.csv")
json.dump(tr, open("profiles/traffic.json", "w"), indent=1)
rows = list(csv.DictReader(open(f"profiles/{rnd}_single_stream_kernel_stats.csv")))
for r in rows[:12]:
    print("%-64s calls %4s avg_us %9.1f pct %s" % (r["Name"][:64], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(tr)
