export TMPDIR=/tmp
timeout -k 10 600 python -m pytest tests -m gpu -x -q -k "small_batch or fp16_scan or index or search or knn" 2>&1 | grep -E "Error|assert|^E |passed|failed" | head -30
