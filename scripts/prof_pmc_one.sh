# PMC counters for kernels matching $1 (regex) on the single-stream forward (run on the GPU box)
cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
OUT=gpurun_out/pmc_one
export SFSN_OVERLAP_CHUNKS=0
rm -rf $OUT && mkdir -p $OUT
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU --kernel-trace --output-format csv -d $OUT/a -o p -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-phase-a > $OUT/a.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_WAIT_INST_LDS SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD SQ_INSTS_SALU SQ_INSTS_SMEM --kernel-trace --output-format csv -d $OUT/b -o p -- python bench.py --inflight 1 --steps 2 --warmup 1 --no-cpu-baseline --no-phase-a > $OUT/b.log 2>&1
python - "$1" <<'PY'
import csv, sys, re, collections
pat = re.compile(sys.argv[1])
for part in "ab":
    rows = list(csv.DictReader(open(f"gpurun_out/pmc_one/{part}/p_counter_collection.csv")))
    a = collections.defaultdict(lambda: collections.defaultdict(float)); nd = collections.defaultdict(set)
    for r in rows:
        k = r["Kernel_Name"]
        if not pat.search(k): continue
        key = (k[:40], r["Grid_Size"] if "Grid_Size" in r else "")
        a[key][r["Counter_Name"]] += float(r["Counter_Value"]); nd[key].add(r["Dispatch_Id"])
    for k, d in a.items():
        print(k, "dispatches", len(nd[k]), {c: "%.4g" % (v / len(nd[k])) for c, v in d.items()})
PY
