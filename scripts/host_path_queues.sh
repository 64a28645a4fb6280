#!/bin/bash
# How HIP streams map onto hardware queues (GPU_MAX_HW_QUEUES, default 4) decides how the two compute lanes, the copy stream and
# the head stream interleave on the chip.  usage: host_path_queues.sh "1 2 3 4 5 6 8"
cd $GRAFT_REPO_ROOT
for q in ${1:-default 1 2 3 4 5 6 8}; do
  if [ $q = default ]; then unset GPU_MAX_HW_QUEUES; else export GPU_MAX_HW_QUEUES=$q; fi
  python bench.py --no-cpu-baseline --no-gates --no-kernel-timing --files 0 ${EXTRA} 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]);print('GPU_MAX_HW_QUEUES=$q ${EXTRA} value(host)', d['value'], 'hbm', d['hbm_resident']['value'], 'gap %.1f%%' % (100*(1-d['value']/d['hbm_resident']['value'])))"
done
