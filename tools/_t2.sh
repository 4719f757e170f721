python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convnext or block" 2>&1 | tail -5
WL=c2 tools/ab_env.sh "fused:" "old:STY_NO_CNX_GX=1" "fused2:" "old2:STY_NO_CNX_GX=1"
