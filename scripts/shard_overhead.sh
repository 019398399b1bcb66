f() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value'], d['roofline']['kernel_ms'], d['roofline']['queries_per_launch'], d['roofline']['arithmetic'])"; }
echo pair; timeout 200 python bench.py --rows 125000 --no-cpu-baseline 2>/dev/null | f
echo single-split; RAGLITE_NO_QUERY_PAIRS=1 timeout 200 python bench.py --rows 125000 --no-cpu-baseline 2>/dev/null | f
echo single-exact; timeout 200 python bench.py --rows 125000 --no-cpu-baseline --exact-fp32 2>/dev/null | f
