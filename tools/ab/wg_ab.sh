for i in 1 2; do
for v in 0 44; do
  export MI_DBG_WG=$v
  python bench.py --no-secondary --no-cpu-baseline --no-conv-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('WG=$v', d['ms_per_step'])"
done; done
