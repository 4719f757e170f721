python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convnext or block or layernorm32" 2>&1 | tail -12
python -m pytest tests/test_hip_parity.py tests/test_boundary.py -x -q -m gpu -k "bf16" 2>&1 | tail -12
WL=c3 tools/ab_env.sh "g16:" "nog16:STY_NO_GRAD16=1" "g16b:" "nog16b:STY_NO_GRAD16=1"
python - <<'PY'
import json
for tag in ('g16','nog16'):
    d=json.load(open(f'gpurun_out/ab_c3_{tag}.json'))
    for r in d['single_stream_kernels']:
        if 'convnext32_bwd' in r['name'] or 'wgrad_cnx' in r['name']:
            print(tag, r['name'], r['launches'], round(1e3*r['ms_per_step']/r['launches'],1),'us', round(r['GBps']))
PY
