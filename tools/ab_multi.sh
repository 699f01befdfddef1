R=${GRAFT_REPO_ROOT:-$(pwd)}
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps 10 --warmup 2 --no-rmse-log"
ARGS="$1"; shift
N=$1; shift
for i in $(seq $N); do
  for L in "$@"; do
    CUMF_ALS_LIB=$R/$L python $R/bench.py $Q $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$L', '$ARGS', 'ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
  done
done
