#!/bin/bash
# The pool's boxes differ by +-3.5 % (22.5k - 24.2k frames/s for the same binary): take the round's profile set on a box from
# the faster half, so that successive sets are comparable.   profile_if_fast.sh TAG [threshold frames/s]
TAG=$1; THR=${2:-23800}
cd $GRAFT_REPO_ROOT
V=$(python bench.py --no-cpu-baseline --no-host-path --no-gates --no-kernel-timing --files 0 2>/dev/null | python -c "import json,sys; print(int(json.loads(sys.stdin.read().strip().splitlines()[-1])['value']))")
echo "box speed $V frames/s (threshold $THR)"
if [ "$V" -lt "$THR" ]; then echo "slow box: skipped"; exit 0; fi
bash scripts/profile_round.sh $TAG > gpurun_out/prof_$TAG.log 2>&1 && bash scripts/profile_cfg4.sh ${TAG}_cfg4 > gpurun_out/prof_${TAG}_cfg4.log 2>&1
echo "profile set $TAG taken"
