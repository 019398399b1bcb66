export TMPDIR=/tmp
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/bc -o t -- python /root/repo/scripts/dev/big_chunks_probe.py 2>&1 | grep "ms per"; head -8 $(find /tmp/bc -name "*kernel_stats.csv") | cut -c1-160
python -m pytest tests/test_gpu_scale_2g.py tests/test_gpu_topk_block.py tests/test_gpu_hi_maxsim.py -m gpu -q --timeout 800 2>&1 | grep -a "passed\|failed\|^E  \|^FAILED" | tail -5
