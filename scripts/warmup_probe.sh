set -e
for rep in 1 2; do
for w in 5 20 80; do
python bench.py --steps 20 --warmup $w --no-search --no-preprocess --no-cpu-baseline --no-sustained --no-e2e 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('steps 20 warmup $w', round(d['value']), round(d['ms_per_step'],4))"
done
python bench.py --steps 196 --warmup 5 --no-search --no-preprocess --no-cpu-baseline --no-sustained --no-e2e 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('steps 196 warmup 5', round(d['value']), round(d['ms_per_step'],4))"
done
