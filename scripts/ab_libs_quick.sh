# Same-box A/B of library build variants, the short form of ab_libs.sh: frames/s of the encode leg (two interleaved rounds, three
# batches in flight) and one rocprofv3 --kernel-trace --stats pass per variant with one batch at a time (per-kernel durations).
#   bash scripts/ab_libs_quick.sh <outdir> libvq_amd.so libvq_amd_x.so ...
set -e
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  ( export VQ_AMD_LIB=$PWD/video-quierer_amd/lib/$lib
    timeout -k 10 200 python bench.py --steps ${AB_STEPS:-60} --warmup 10 --no-search --no-preprocess --no-cpu-baseline --no-sustained --no-e2e $AB_ARGS 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$lib] 3 streams', round(d['value']), 'frames/s', round(d['ms_per_step'],4), 'ms/step')" )
done
done | tee $O/frames_per_s.txt
for lib in "$@"; do
  n=${lib%.so}
  ( export VQ_AMD_LIB=$PWD/video-quierer_amd/lib/$lib VQ_BENCH_CONCURRENT=1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n.stats -o p -- python3 bench.py --steps 30 --warmup 6 --streams 1 --no-cpu-baseline --no-sustained --no-search --no-preprocess --no-e2e $AB_ARGS > $O/$n.bench_1stream.json 2> $O/$n.stats.err
    find $O/$n.stats -name '*kernel_stats.csv' -exec cp {} $O/$n.kernel_stats.csv \;
    rm -rf $O/$n.stats
    echo "== $lib, one batch at a time (rocprofv3 averages, us)"
    python3 - $O/$n.kernel_stats.csv <<'PY'
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: -float(r["TotalDurationNs"]))[:9]
for r in rows: print("  %8.1f us x %5s  %s" % (float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:110]))
PY
  )
done | tee $O/kernels.txt
