set -e
export TMPDIR=/tmp
for S in 1 3; do
VQ_AMD_LIB=$PWD/video-quierer_amd/lib/libvq_amd_stamps.so STREAMS=$S PASSES=900 timeout -k 10 300 python scripts/gemm_tower_stamps.py 2> gpurun_out/stamps_s$S.txt
echo "== streams $S"; python scripts/gemm_tower_stamps.py --summarize gpurun_out/stamps_s$S.txt | tee gpurun_out/stamps_s${S}_summary.txt
done
