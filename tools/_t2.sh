python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convnext32_lean or block_bf16_mode" 2>&1 | tail -5
WL=c3 tools/ab_env.sh "fused:" "old:STY_NO_CNX_GX=1 STY_NO_CNX_XN16=1" "fused2:" "old2:STY_NO_CNX_GX=1 STY_NO_CNX_XN16=1"
python - <<'PY'
import json
for tag in ('fused','old'):
    d=json.load(open(f'gpurun_out/ab_c3_{tag}.json'))
    for r in d['single_stream_kernels']:
        if 'convnext32_bwd' in r['name'] or 'wgrad_cnx' in r['name']:
            print(tag, r['name'], r['launches'], round(1e3*r['ms_per_step']/r['launches'],1),'us', round(r['GBps']))
PY
