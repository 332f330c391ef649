# End-of-session evidence on ONE GPU box: full GPU suite, smoke, bench (default and driver flags), rocprofv3 kernel
# summaries (three batches in flight / one), the two PMC passes.  Everything lands in gpurun_out/final/; copy what is
# to be judged into profiles/.   usage (from the repo root): bash scripts/final_artifacts.sh
set -e
export TMPDIR=/tmp
O=gpurun_out/final; mkdir -p $O
timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }
tail -2 $O/pytest_gpu.log
timeout -k 10 300 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err; echo "bench default done"
timeout -k 10 600 python bench.py --gpus 1 --steps 20 --warmup 5 > $O/bench_driver_flags.json 2> $O/bench_driver.err; echo "bench driver flags done"
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof3 -o p -- python3 bench.py --steps 30 --warmup 6 --no-cpu-baseline --no-sustained --no-e2e > $O/bench_under_rocprof.json 2> $O/prof3.err; echo "rocprof 3-stream done"
VQ_BENCH_CONCURRENT=1 timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof1 -o p -- python3 bench.py --steps 30 --warmup 6 --streams 1 --no-cpu-baseline --no-sustained --no-search --no-preprocess --no-e2e > $O/bench_under_rocprof_1stream.json 2> $O/prof1.err; echo "rocprof 1-stream done"
VQ_BENCH_CONCURRENT=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/pmc_fetch -o p -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline --no-e2e > /dev/null 2> $O/pmc_fetch.err; echo "pmc fetch done"
VQ_BENCH_CONCURRENT=1 timeout -k 10 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $O/pmc_write -o p -- python3 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline --no-e2e > /dev/null 2> $O/pmc_write.err; echo "pmc write done"
python3 scripts/pmc_traffic.py $O/pmc_fetch $O/pmc_write $O/pmc_traffic.json "VQ_BENCH_CONCURRENT=1 bench.py --steps 3 --warmup 1 --streams 1 --no-sustained --no-cpu-baseline (the kernels of the headline configuration, one batch at a time)" | tail -3
# a LONE handle (no VQ_ENC_CONCURRENT): the dispatch a single-stream user gets — 160-row tiles for the N = 768 pair, tail split for fc1
timeout -k 10 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof0 -o p -- python3 bench.py --steps 30 --warmup 6 --streams 1 --no-cpu-baseline --no-sustained --no-search --no-preprocess --no-e2e > $O/bench_under_rocprof_single_handle.json 2> $O/prof0.err; echo "rocprof single-handle done"
timeout -k 10 300 python bench.py --workload config4 --steps 5 --warmup 2 > $O/bench_config4_1gpu.json 2>> $O/bench.err; echo "config4 done"
VQ_BENCH_DEVICE=0 VQ_BENCH_BACKEND=gloo timeout -k 10 300 python bench.py --gpus 2 --steps 10 --warmup 3 --no-cpu-baseline --no-e2e > $O/bench_2rank_gloo_rehearsal.json 2>> $O/bench.err; echo "2-rank gloo rehearsal done"
timeout -k 10 300 python scripts/search_latency.py > $O/search_latency.txt 2>&1; echo "search latency done"
timeout -k 10 300 python scripts/host_latency_probe.py 2>&1 | grep -v amdgpu.ids > $O/host_latency.txt; echo "host latency done"
timeout -k 10 300 python bench.py --model l14 --batch 32 --steps 12 --warmup 3 --no-search --no-preprocess --no-e2e > $O/bench_vit_l14_336.json 2>> $O/bench.err; echo "ViT-L/14@336 done"
find $O -name '*kernel_stats.csv' | while read f; do cp $f $O/$(echo $f | sed 's#.*/\(prof[013]\)/.*#\1#')_kernel_stats.csv; done
find $O -name '*_kernel_trace.csv' -delete; find $O -name '*counter_collection.csv' -size +20M -delete
ls -la $O
