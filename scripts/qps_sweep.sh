f() { python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['config']['queries_per_step'], d['value'], d['ms_per_step'])"; }
for q in 32 64 128; do timeout 200 python bench.py --rows 125000 --no-cpu-baseline --queries-per-step $q --steps 20 2>/dev/null | f; done
for q in 32 128; do timeout 200 python bench.py --no-cpu-baseline --queries-per-step $q --steps 10 2>/dev/null | f; done
