run() { python bench.py --no-search --no-preprocess --no-cpu-baseline --no-e2e --no-sustained --steps 200 --warmup 10 "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(round(d['value']))"; }
echo "streams3 $(run)"; echo "streams2 $(run --streams 2)"; echo "streams4 $(run --streams 4)"; echo "streams3 $(run)"
echo "hwq16 $(GPU_MAX_HW_QUEUES=16 run)"; echo "tile2d $(VQ_AMD_TILE2D=1 run)"; echo "nomulti $(VQ_AMD_GEMM_MULTI=0 run)"; echo "streams3 $(run)"
echo "fulllast $(VQ_AMD_FULL_LAST_LAYER=1 run)"; echo "bf16 $(run --dtype bf16)"
