#!/bin/bash
# resident attention (T = 201): the r2 kernel (7 waves x 2 sequential query tiles, 2 workgroups per CU) against the paired form
# (CBAS_ATTN_PAIR=1: 7 waves x 2 interleaved tiles).  Alternating runs on one device; correctness via the ViT-B golden.
cd $GRAFT_REPO_ROOT
for rep in 1 2; do for v in 0 1; do
CBAS_ATTN_PAIR=$v python bench.py --steps 60 --no-cpu-baseline --no-host-path --files 0 2>/dev/null | python -c "
import json,sys;d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]);print('CBAS_ATTN_PAIR=$v', 'hbm value', d['value'], 'attention us', d['roofline']['by_kernel']['attention']['avg_us'], 'cls err', d['gates']['cls_rel_err_max'])"
done; done
