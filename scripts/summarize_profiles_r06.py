#!/usr/bin/env python3
"""gpurun_out/prof_r06 (scratch, from scripts/prof_r06.sh) -> profiles/r06_* (tracked): kernel stats of the three commands, a PMC
summary per kernel, and profiles/r06_pmc.json -- the PMC-derived figures bench.py attaches to its line when the library build
(source hash) matches."""
import collections, csv, json, os, shutil, sys
base = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r06"
rnd = "r06"
os.makedirs("profiles", exist_ok=True)
shutil.copy(f"{base}/single/s_kernel_stats.csv", f"profiles/{rnd}_single_stream_kernel_stats.csv")
shutil.copy(f"{base}/single_whole/s_kernel_stats.csv", f"profiles/{rnd}_single_stream_whole_launch_kernel_stats.csv")
shutil.copy(f"{base}/default/d_kernel_stats.csv", f"profiles/{rnd}_default_kernel_stats.csv")
shutil.copy(f"{base}/bench_default.json", f"profiles/{rnd}_bench_line.json")
shutil.copy(f"{base}/bench_streaming.json", f"profiles/{rnd}_streaming_line.json")
for src, dst in (("bench_streaming_waveform_host.json", "streaming_waveform_host_line.json"), ("bench_training.json", "training_line.json"),
                 ("bench_training_b16.json", "training_line_b16.json"), ("training/t_kernel_stats.csv", "training_kernel_stats.csv"),
                 ("training64/t_kernel_stats.csv", "training_b64_kernel_stats.csv")):
    if os.path.exists(f"{base}/{src}"):
        shutil.copy(f"{base}/{src}", f"profiles/{rnd}_{dst}")
if os.path.exists("gpurun_out/parity_report.jsonl"):
    shutil.copy("gpurun_out/parity_report.jsonl", f"profiles/{rnd}_parity_report.jsonl")
src_hash = open(f"{base}/source_hash.txt").read().strip()

def agg(f):
    rows = list(csv.DictReader(open(f)))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]; a[k][r["Counter_Name"]] += float(r["Counter_Value"]); nd[k].add(r["Dispatch_Id"])
    return {k: {c: v / len(nd[k]) for c, v in d.items()} for k, d in a.items()}, {k: len(v) for k, v in nd.items()}

OURS = ("gsn_scan3", "gsn_scan", "gsn_stack", "spike_proj", "input_proj", "features_kernel", "deepfilter", "projdf", "rowsum", "laplace", "spike_count", "stack_setup")
PERFWD = ("projdf", "deepfilter")  # one launch per forward (whole-sequence launches): the divisor of the per-forward sums
f, nf = agg(f"{base}/pmc_FETCH_SIZE/p_counter_collection.csv"); w, _ = agg(f"{base}/pmc_WRITE_SIZE/p_counter_collection.csv")
s1, _ = agg(f"{base}/pmc_SQ_WAVES/p_counter_collection.csv"); s2, _ = agg(f"{base}/pmc_SQ_INSTS_MFMA/p_counter_collection.csv")
c1 = ["SQ_WAVES", "SQ_WAVE_CYCLES", "SQ_BUSY_CYCLES", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY"]
c2 = ["SQ_INSTS_MFMA", "SQ_VALU_MFMA_BUSY_CYCLES", "SQ_INSTS_LDS", "SQ_LDS_BANK_CONFLICT", "SQ_INSTS_VMEM_RD", "SQ_INSTS_VMEM_WR", "GRBM_GUI_ACTIVE"]
out = ["# PMC summary per dispatch (rocprofv3 --pmc, one counter group per run; scripts/prof_r06.sh), library source hash " + src_hash,
       "# command: SFSN_OVERLAP_CHUNKS=0 python bench.py --no-cpu-baseline --sequential --steps 2 --warmup 1 --no-phase-a  (B=64, T=1000, live baseline_m, fp32;",
       "#          one forward at a time, every scan one whole-sequence launch)",
       "# FETCH_SIZE / WRITE_SIZE are KiB.  gfx950 correction (MI355X_MICROARCH.md, HBM): FETCH_SIZE reports 1/2 of a wide",
       "# coalesced streaming read -> hbm_read_bytes = 2 * FETCH_SIZE * 1024; WRITE_SIZE taken as is.",
       "kernel,dispatches,FETCH_SIZE_KiB,hbm_read_MB_corrected,WRITE_SIZE_KiB,hbm_write_MB," + ",".join(c1 + c2)]
for k in f:
    if not any(t in k for t in OURS):
        continue
    fs = f[k].get("FETCH_SIZE", 0); ws = w.get(k, {}).get("WRITE_SIZE", 0)
    vals = [s1.get(k, {}).get(c, "") for c in c1] + [s2.get(k, {}).get(c, "") for c in c2]
    out.append('"%s",%d,%.0f,%.1f,%.0f,%.1f,%s' % (k.replace('"', ""), nf[k], fs, 2 * fs * 1024 / 1e6, ws, ws * 1024 / 1e6,
                                                  ",".join("%.4g" % v if v != "" else "" for v in vals)))
open(f"profiles/{rnd}_pmc_summary.csv", "w").write("\n".join(out) + "\n")

def hbm(fd, wd, k):
    return int(2 * fd[k].get("FETCH_SIZE", 0) * 1024 + wd.get(k, {}).get("WRITE_SIZE", 0) * 1024)

jf, njf = agg(f"{base}/job_FETCH_SIZE/p_counter_collection.csv"); jw, _ = agg(f"{base}/job_WRITE_SIZE/p_counter_collection.csv")
n_fwd = max(v for kk, v in njf.items() if any(t_ in kk for t_ in PERFWD))
job = sum(hbm(jf, jw, kk) * v / n_fwd for kk, v in njf.items() if any(t in kk for t in OURS))
# the strict schedule's sub-band layers: round 5 = ONE launch of the wide stack kernel for both layers (layer 1: IO-wave scan roles /
# FUSEDX3 for group 0; layer 2: FUSED3 roles); round 3's per-layer IO-wave scan at 4 rows per workgroup if that is what ran
kp = [x for x in f if "gsn_stack_wide_kernel<4" in x]
k4l = [x for x in f if "gsn_scan3_kernel<4, 4" in x]
k4 = (kp or k4l)[0]
kfu = ([x for x in jf if "gsn_scan_fused3_kernel" in x] or [x for x in jf if "gsn_scan_fused_kernel" in x])[0]
kpd = [x for x in f if "projdf_kernel" in x]
kst = ([x for x in s2 if "gsn_stack_fb_kernel" in x] or [x for x in s2 if "gsn_stack_kernel<5" in x])[0]  # (round 5: the IO-wave full-band kernel for whole-sequence launches)
m = s2[kst]
n_cu, simd = 256, 4
wg = 36  # full-band stack: 16 + 16 scan workgroups + 4 PROJ workgroups, one per CU
n_xcd = 8
cyc = m.get("GRBM_GUI_ACTIVE", 1) / n_xcd  # the counter is the sum over the 8 XCDs' GRBMs: cycles of the launch = sum / 8 (= duration x ~2.36 GHz)
mf = dict(kernel=kst, insts_mfma=m.get("SQ_INSTS_MFMA"), mfma_busy_cycles=m.get("SQ_VALU_MFMA_BUSY_CYCLES"), gui_active_cycles_sum_over_xcds=m.get("GRBM_GUI_ACTIVE"),
          launch_cycles=cyc, busy_cycles_per_mfma=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, m.get("SQ_INSTS_MFMA", 1)),
          util_chip_wide=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, cyc * n_cu * simd),
          util_on_occupied_cus=m.get("SQ_VALU_MFMA_BUSY_CYCLES", 0) / max(1.0, cyc * wg * simd),
          note="SQ_VALU_MFMA_BUSY_CYCLES / (launch cycles x SIMDs): all 1024 SIMDs of the chip / the 144 SIMDs of the 36 CUs the launch occupies; "
               "launch cycles = GRBM_GUI_ACTIVE / 8 (the counter sums the eight XCDs)")
def forward_bytes(tag):
    """HBM bytes of one forward from a FETCH_SIZE / WRITE_SIZE pass pair (all of this package's kernels, per deep-filter launch)"""
    try:
        ff, nff = agg(f"{base}/{tag}_FETCH_SIZE/p_counter_collection.csv"); fw, _ = agg(f"{base}/{tag}_WRITE_SIZE/p_counter_collection.csv")
    except Exception:
        return None, None
    nfw = max(v for kk, v in nff.items() if any(t_ in kk for t_ in PERFWD))
    per = {kk: hbm(ff, fw, kk) for kk in nff if any(t in kk for t in OURS)}
    tot = int(sum(per[kk] * nff[kk] / nfw for kk in per))
    pair = [kk for kk in per if "gsn_stack_wide_kernel<4" in kk]
    return tot, (per[pair[0]] if pair else None)
nl_single, nl_pair = forward_bytes("nl")
nl_job, _ = forward_bytes("nljob")
pj = dict(source_hash=src_hash, B=64, T=1000,
          workload=dict(B=64, T=1000, timed_region_rows_per_wg=[8, 16], model="live baseline_m", layer_outputs="api-faithful"),
          sb_scan_single_kernel=k4, sb_scan_single_hbm_bytes_per_launch=(None if kp else hbm(f, w, k4)),
          sb_pair_kernel=(kp[0] if kp else None), sb_pair_hbm_bytes_per_launch=(hbm(f, w, kp[0]) if kp else None),
          sb_pair_traffic_over_algorithmic=(round(hbm(f, w, kp[0]) / (2 * 14592 * 64 * 1000), 3) if kp else None),
          sb_fused_kernel=kfu, sb_fused_hbm_bytes_per_launch=hbm(jf, jw, kfu),
          projdf_kernel=(kpd[0] if kpd else None), projdf_hbm_bytes_per_launch=(hbm(f, w, kpd[0]) if kpd else None),
          forward_hbm_bytes=int(job), forward_hbm_bytes_single_forward_schedule=int(sum(hbm(f, w, kk) * v / max(vv for q, vv in nf.items() if any(t_ in q for t_ in PERFWD))
                                                                                      for kk, v in nf.items() if any(t in kk for t in OURS))),
          full_band_stack_mfma=mf,
          forward_hbm_bytes_no_layer_outputs=nl_job, forward_hbm_bytes_no_layer_outputs_single_forward_schedule=nl_single,
          sb_pair_hbm_bytes_per_launch_no_layer_outputs=nl_pair,
          note="HBM bytes = 2*FETCH_SIZE*1024 + WRITE_SIZE*1024 per dispatch (gfx950 FETCH_SIZE half-count correction); forward_hbm_bytes = all "
               "kernels of one forward in the timed region's geometry (--rpw 8,16: full-band stack at 8, sub-band scans at 16 rows per workgroup); sources: profiles/r06_pmc_summary.csv and gpurun_out/prof_r06/job_*")
json.dump(pj, open(f"profiles/{rnd}_pmc.json", "w"), indent=1)
rows = list(csv.DictReader(open(f"profiles/{rnd}_single_stream_whole_launch_kernel_stats.csv")))
for r in rows[:14]:
    print("%-70s calls %4s avg_us %9.1f pct %s" % (r["Name"][:70], r["Calls"], float(r["AverageNs"]) / 1e3, r["Percentage"]))
print(json.dumps(pj, indent=1))

# the strict forward's kernels in launch order (what gates what): profiles/r06_strict_timeline.txt
if os.path.exists(f"{base}/single/s_kernel_trace.csv"):
    import subprocess, sys
    subprocess.run([sys.executable, "scripts/strict_timeline_r06.py", f"{base}/single/s_kernel_trace.csv"], stdout=subprocess.DEVNULL, check=False)
