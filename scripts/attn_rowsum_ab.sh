cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 3 4; do
CBAS_ATTN_STREAM=$v python bench.py --model vitl16 --hw 518 --batch 32 --steps 8 --warmup 2 --no-cpu-baseline --no-gates --no-host-path --files 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]);print('CBAS_ATTN_STREAM=$v', 'value', d['value'], 'attention us', d['roofline']['by_kernel']['attention']['avg_us'])"
done; done
