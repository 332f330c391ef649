set -e
export TMPDIR=/tmp
timeout -k 10 500 python -m pytest tests -m gpu -x -q -k "encoder or config1 or extractor or text or gemm or l14 or checkpoint" > gpurun_out/ab_tests.log 2>&1 || { tail -30 gpurun_out/ab_tests.log; exit 1; }
tail -3 gpurun_out/ab_tests.log
for i in 1 2; do
VQ_AMD_LIB=$PWD/video-quierer_amd/lib/libvq_amd_base.so timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('base', d['value'], d['ms_per_step'])"
timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('new ', d['value'], d['ms_per_step'])"
VQ_AMD_QKV_LAYOUT=rows timeout -k 10 200 python bench.py --steps 10 --warmup 3 2>/dev/null | python3 -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('rows', d['value'], d['ms_per_step'])"
done
