export TMPDIR=/tmp
python -m pytest tests/test_gpu_hi_few.py tests/test_gpu_hi_pivot.py tests/test_gpu_hi_maxsim.py tests/test_gpu_pairs_packed.py -m gpu -q --timeout 600 2>&1 | grep -a "passed\|failed" | tail -3
python scripts/time_one_query.py 200
python scripts/bench_embed.py 4000 | cut -c1-600
python bench.py --steps 20 --warmup 5 --no-configs --no-vendor-gemm --no-cpu-baseline --no-f16 2>/dev/null | python scripts/bench_summary.py /dev/stdin | head -1
python scripts/bench_configs.py beyond_shape | cut -c1-1200
cd /tmp && rocprofv3 --kernel-trace --output-format csv -d /tmp/tt -o t -- python /root/repo/bench.py --steps 10 --warmup 2 --no-cpu-baseline --no-configs --no-f16 --no-vendor-gemm > /dev/null 2>&1; python /root/repo/scripts/step_timeline.py $(find /tmp/tt -name "*kernel_trace.csv") query_planes_kernel maxsim_pp_kernel
