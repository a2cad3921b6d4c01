#!/bin/bash
python -m pytest tests/test_cli.py -x -q -m gpu 2>&1 | tail -4
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "grid_limit" 2>&1 | tail -15
tools/r5_cliprof.sh
