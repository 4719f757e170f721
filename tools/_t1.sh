python -m pytest tests/test_hip_parity.py -x -q -m gpu -k "convnext or block" 2>&1 | tail -25
