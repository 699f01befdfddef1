#!/bin/bash
# A/B of two ENVIRONMENTS on one box, alternating bench runs: tools/ab_env.sh "<bench args>" "ENV_A" "ENV_B" [reps] [steps] [warmup]
# e.g. tools/ab_env.sh "--f 100 --solver lu" "CUMF_ALS_PRESPLIT=0" "CUMF_ALS_PRESPLIT=-" 3   ("X=-" leaves X unset)
R=${GRAFT_REPO_ROOT:-$(pwd)}
ARGS="$1"; A="$2"; B="$3"; N=${4:-3}; S=${5:-30}; W=${6:-10}
Q="--no-cpu-baseline --no-fast-leg --no-gram-leg --allow-missing-traffic --steps $S --warmup $W"
for i in $(seq $N); do
  for E in "$A" "$B"; do
    K=${E%%=*}; V=${E#*=}
    if [ "$V" = "-" ]; then CMD="env -u $K"; else CMD="env $E"; fi
    $CMD python $R/bench.py $Q $ARGS 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); r=d['roofline']; print('$E', '$ARGS', 'ms', round(d['ms_per_step'],3), 'x', round(r['x_side_ms'],3), 'theta', round(r['theta_side_ms'],3))"
  done
done
