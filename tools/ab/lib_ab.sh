# same-box A/B of two builds of the C-ABI library (CETPICK_HIP_LIB): tools/ab/lib_ab.sh <lib A> <lib B>
for i in 1 2 3; do
for l in $1 $2; do
  CETPICK_HIP_LIB=$GRAFT_REPO_ROOT/$l python bench.py --no-secondary --no-cpu-baseline --no-conv-profile 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('$l', d['ms_per_step'])"
done; done
