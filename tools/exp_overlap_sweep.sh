#!/bin/bash
# sweep of the shadow / bounce overlap: the bounce it starts after, the compute units the bounces keep
for w in 8 1; do
  for cfg in "0 64" "0 96" "0 32" "1 32" "1 64"; do set -- $cfg
    echo "== world $w after $1 cus $2"
    RT_OPTS="6=1,7=$1,8=$2" timeout -k 10 100 python3 tools/rank_share.py $w || exit 1
  done
done
