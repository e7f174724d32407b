#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_golden.py tests/test_parity_gpu.py -m gpu -q --timeout 1200 -k "golden or 2_to_32" 2>&1 | tail -6
