#!/bin/bash
# sweep of the shadow / bounce overlap: the bounce it starts after, the compute units the bounces keep
for w in 8 1; do
  echo "== world $w: off"; RT_OPTS=6=0 timeout -k 10 100 python3 tools/rank_share.py $w || exit 1
  for after in 1 2; do for cus in 32 64; do
    echo "== world $w after $after cus $cus"
    RT_OPTS="6=1,7=$after,8=$cus" timeout -k 10 100 python3 tools/rank_share.py $w || exit 1
  done; done
done
