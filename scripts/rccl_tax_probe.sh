export TMPDIR=/tmp
for v in "X=1" "VQ_BENCH_REHEARSE_NATIVE=2"; do
( export $v
  rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r04y_prof -o p -- python3 bench.py --steps 60 --warmup 6 --no-cpu-baseline --no-sustained --no-search --no-preprocess --no-e2e > gpurun_out/r04y_bench.json 2>/dev/null
  f=$(find gpurun_out/r04y_prof -name "*kernel_stats.csv"); echo "== $v: $(python3 -c "import json;print(round(json.loads(open('gpurun_out/r04y_bench.json').read().strip().splitlines()[-1])['value']))") frames/s under the profiler"
  python3 - $f <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))[:8]
for r in rows: print("  %8.1f us x %5s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:100]))
PY
  rm -rf gpurun_out/r04y_prof )
done
