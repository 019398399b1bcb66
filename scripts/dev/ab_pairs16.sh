export TMPDIR=/tmp
for o in 2 1 2 1; do
  python bench.py --steps 20 --warmup 5 --no-configs --no-vendor-gemm --no-cpu-baseline --no-f16 --opt pairs_packed=$o 2>/dev/null | python scripts/bench_summary.py /dev/stdin | head -1
done
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp -o b -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm --opt pairs_packed=2 > /dev/null 2>&1; grep "pairs_packed" $(find /tmp/pp -name "*kernel_stats.csv") | cut -c1-200
