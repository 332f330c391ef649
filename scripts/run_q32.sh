set -e
export TMPDIR=/tmp
timeout -k 10 300 python -m pytest tests -m gpu -x -q -k "small_batch or fp16_scan or index or search or knn" 2>&1 | tail -3
VQ_AMD_LIB=$PWD/video-quierer_amd/lib/libvq_amd_stamps.so timeout -k 10 200 python scripts/rescore_stamps.py 2>&1 | grep RS_STAMP
for Q in 1 32; do
timeout -k 10 120 python scripts/scan_ab.py 1000000 $Q 10 | sed 's/^/new  /' | cut -c1-300
done
