# Per-kernel durations of bench.py's encode leg (one batch at a time, rocprofv3 --kernel-trace --stats) for a list of library builds:
#   KERNEL=attention AB_ARGS="--model l14 --batch 32" bash scripts/ab_kernel_time.sh <outdir> libvq_amd.so libvq_amd_x.so ...
set -e
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
for lib in "$@"; do
  n=${lib%.so}
  ( export VQ_AMD_LIB=$PWD/video-quierer_amd/lib/$lib VQ_BENCH_CONCURRENT=1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n.stats -o p -- python3 bench.py --steps ${AB_STEPS:-12} --warmup 3 --streams 1 --no-cpu-baseline --no-sustained --no-search --no-preprocess --no-e2e $AB_ARGS > $O/$n.bench_1stream.json 2> $O/$n.stats.err
    find $O/$n.stats -name '*kernel_stats.csv' -exec cp {} $O/$n.kernel_stats.csv \;
    rm -rf $O/$n.stats
    python3 - $O/$n.kernel_stats.csv "$lib" "${KERNEL:-attention}" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if sys.argv[3] in r["Name"]: print("%-28s %8.1f us x %5s  %s" % (sys.argv[2], float(r["AverageNs"]) / 1e3, r["Calls"], r["Name"][:90]))
PY
  )
done | tee $O/kernel_times.txt
