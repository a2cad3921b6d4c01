#!/bin/bash
# per-level times of phase 2 (KC_OV_TRACE): tools/ov_trace.sh [bench flags]
cd "$(dirname "$0")/.."
KC_OV_TRACE=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli "$@" 2>&1 | grep "^\[ov\]" | tail -40
