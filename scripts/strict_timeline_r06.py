"""profiles/r06_strict_timeline.txt: the kernels of ONE strict forward (bench.py --sequential under rocprofv3 --kernel-trace,
scripts/prof_r06.sh step 1) in launch order -- stream, start, end, duration (µs from the forward's first kernel), frames of the chunk
where the kernel name tells, and the per-frame time of the stack launches."""
import csv, sys
src = sys.argv[1] if len(sys.argv) > 1 else "gpurun_out/prof_r06/single/s_kernel_trace.csv"
tr = sorted(csv.DictReader(open(src)), key=lambda r: int(r["Start_Timestamp"]))
# the last complete forward: from the last-but-one 'CatArrayBatchedCopy' (the error-word reduction behind a forward) to the last one
ends = [i for i, r in enumerate(tr) if "CatArrayBatchedCopy" in r["Kernel_Name"]]
lo, hi = ends[-2] + 3, ends[-1]
rows = tr[lo:hi]
t0 = int(rows[0]["Start_Timestamp"])
T, fr = 1000, (240, 380, 380)
out = ["# one strict forward, B = 64, T = 1000, live baseline_m (chunks of 240 / 380 / 380 frames); times in µs from the forward's first kernel",
       "# stream  start    end     dur   kernel"]
seen = {}
for r in rows:
    s, e = (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - t0) / 1e3
    name = r["Kernel_Name"].split("(")[0].replace("void ", "")[:48]
    k = seen[name] = seen.get(name, 0) + 1
    note = ""
    if "gsn_stack" in name:
        n = fr[min(k, 3) - 1]
        note = f"   chunk {k}: {n} frames, {(e - s) / n:.3f} µs per frame"
    out.append(f"{r['Stream_Id']:>6} {s:8.1f} {e:8.1f} {e - s:7.1f}   {name}{note}")
out.append(f"# span {(int(rows[-1]['End_Timestamp']) - t0) / 1e3:.1f} µs")
open("profiles/r06_strict_timeline.txt", "w").write("\n".join(out) + "\n")
print("\n".join(out))
