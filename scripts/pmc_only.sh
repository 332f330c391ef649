set -e
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O; rm -rf $O/pmc_fetch $O/pmc_write
VQ_BENCH_CONCURRENT=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline > /dev/null 2> $O/pmc_fetch.err; echo "pmc fetch done"
VQ_BENCH_CONCURRENT=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline > /dev/null 2> $O/pmc_write.err; echo "pmc write done"
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "VQ_BENCH_CONCURRENT=1 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline (the kernels of the headline configuration, one batch at a time)" | tail -3
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -size +20M -delete
bash scripts/pmc_mfma.sh
