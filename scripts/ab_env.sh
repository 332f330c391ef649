# Same-box A/B of bench.py's encode leg under environment variants (boxes differ by up to 25 %, so only runs
# inside one gpurun call compare).  Each argument is a list of NAME=VAL settings for one variant; AB_ARGS inside
# a variant adds bench.py flags.  First variant = baseline, e.g.
#   bash scripts/ab_env.sh "X=0" "VQ_AMD_GEMM_MULTI=0" "AB_ARGS=--streams=4"
set -e
export TMPDIR=/tmp
for rep in 1 2; do
for v in "$@"; do
  ( export $v
    timeout -k 10 200 python bench.py --steps ${AB_STEPS:-40} --warmup 5 --no-search --no-preprocess --no-cpu-baseline --no-sustained $AB_ARGS 2>/dev/null |
      python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('[$v]', round(d['value']), round(d['ms_per_step'],4))" )
done
done
