# single-stream forward time against the chunking of the full-band / sub-band overlap (SFSN_OVERLAP_CHUNKS, SFSN_OVERLAP_FIRST)
for ch in 2 3; do for first in 240 280 320 360 400 450; do
  r=$(SFSN_OVERLAP_CHUNKS=$ch SFSN_OVERLAP_FIRST=$first python bench.py --sequential --steps 10 --warmup 3 --no-cpu-baseline --no-streaming-leg --no-phase-a 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(d['ms_per_step'])")
  echo "chunks $ch first $first : $r ms"
done; done
