cd $GRAFT_REPO_ROOT
for v in default 1 0; do
  if [ $v = default ]; then unset HSA_ENABLE_SDMA; else export HSA_ENABLE_SDMA=$v; fi
  python bench.py --no-cpu-baseline --no-gates --no-kernel-timing --files 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]);print('HSA_ENABLE_SDMA=$v value(host)', d['value'], 'hbm', d['hbm_resident']['value'])"
done
unset HSA_ENABLE_SDMA
python scripts/host_path_probe.py
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --memory-copy-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/hp_trace -- python3 $GRAFT_REPO_ROOT/bench.py --no-cpu-baseline --no-gates --no-kernel-timing --files 0 --steps 40 > /dev/null 2>&1
ls $GRAFT_REPO_ROOT/gpurun_out/hp_trace/*/ | head; head -5 $GRAFT_REPO_ROOT/gpurun_out/hp_trace/*/*memory_copy_stats.csv 2>/dev/null; grep -i "copy\|blit" $GRAFT_REPO_ROOT/gpurun_out/hp_trace/*/*kernel_stats.csv | head -5
