for i in 1 2; do
for v in 0 1; do
  if [ $v = 1 ]; then export MI_CUBE2_REDUCE=1; else unset MI_CUBE2_REDUCE; fi
  python bench.py --no-secondary --no-cpu-baseline --no-conv-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('REDUCE=$v', d['ms_per_step'])"
done; done
