# MFMA-pipe busy cycles per kernel, in the tower and the search (one rocprofv3 --pmc pass, kernels serialised by the profiler).
set -e
export TMPDIR=/tmp
O=gpurun_out/pmc_mfma; mkdir -p $O
VQ_BENCH_CONCURRENT=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d $O -o p -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline --no-preprocess > /dev/null 2> $O/err.txt
python3 scripts/pmc_summary.py $O 'gemm_tn256|attention_t64|scan4|scan3|rescore' > $O/summary.txt
cat $O/summary.txt | head -90
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -delete
