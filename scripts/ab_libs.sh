# Same-box A/B of library BUILD variants (EXTRA=-D... builds beside the product library, selected by $VQ_AMD_LIB):
# frames/s of bench.py's encode leg, two interleaved rounds, then for each variant one rocprofv3 --kernel-trace --stats pass
# (one batch at a time: per-kernel durations) and the two PMC passes (FETCH_SIZE / WRITE_SIZE: bytes from beyond L2 per launch).
#   bash scripts/ab_libs.sh <outdir> libvq_amd.so libvq_amd_ntw.so ...
set -e
export TMPDIR=/tmp
O=$1; shift
mkdir -p $O
for rep in 1 2; do
for lib in "$@"; do
  ( export VQ_AMD_LIB=$PWD/video-quierer_amd/lib/$lib
    timeout -k 10 200 python bench.py --steps ${AB_STEPS:-60} --warmup 10 --no-search --no-preprocess --no-cpu-baseline --no-sustained --no-e2e 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$lib] 3 streams', round(d['value']), 'frames/s', round(d['ms_per_step'],4), 'ms/step')" )
done
done | tee $O/frames_per_s.txt
for lib in "$@"; do
  n=${lib%.so}
  ( export VQ_AMD_LIB=$PWD/video-quierer_amd/lib/$lib VQ_BENCH_CONCURRENT=1
    timeout -k 10 300 rocprofv3 --kernel-trace --stats --output-format csv -d $O/$n.stats -o p -- python3 bench.py --steps 30 --warmup 6 --streams 1 --no-cpu-baseline --no-sustained --no-search --no-preprocess --no-e2e > $O/$n.bench_1stream.json 2> $O/$n.stats.err
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/$n.fetch -o p -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline --no-search --no-preprocess --no-e2e > /dev/null 2> $O/$n.fetch.err
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/$n.write -o p -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline --no-search --no-preprocess --no-e2e > /dev/null 2> $O/$n.write.err
    python3 scripts/pmc_traffic.py $O/$n.fetch $O/$n.write $O/$n.pmc_traffic.json "$lib, one batch at a time" > /dev/null
    find $O/$n.stats -name '*kernel_stats.csv' -exec cp {} $O/$n.kernel_stats.csv \;
    rm -rf $O/$n.stats $O/$n.fetch $O/$n.write
    echo "$lib profiled" )
done
python3 - "$O" "$@" <<'PY'
import csv, json, sys, re
O, libs = sys.argv[1], sys.argv[2:]
for lib in libs:
    n = lib[:-3]
    t = json.load(open(f"{O}/{n}.pmc_traffic.json"))["kernels"]
    dur = {}
    for r in csv.DictReader(open(f"{O}/{n}.kernel_stats.csv")):
        nm = r["Name"]
        for cls, pat in (("gemm_qkv", r"gemm_tn.*EpiLnH16<(true|false), false>"), ("gemm_fc1_quickgelu", r"gemm_tn.*EpiLnH16<(true|false), true>"),
                         ("gemm_out_proj_residual", r"gemm_tn.*EpiBiasResidual(Ln)?F32<0"), ("gemm_fc2_residual", r"gemm_tn.*EpiBiasResidual(Ln)?F32<1")):
            if re.search(pat, nm) and float(r["AverageNs"]) > dur.get(cls, 0):
                dur[cls] = float(r["AverageNs"])
    print(lib, {k: (round(v.get("hbm_bytes", 0) / 1e6, 1), "MB", round(dur.get(k, 0) / 1e3, 1), "us") for k, v in t.items() if k.startswith("gemm_")})
PY
