#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests/test_c_device_interface.py -m gpu -q --timeout 900 2>&1 | tail -15
