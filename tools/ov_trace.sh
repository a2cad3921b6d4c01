cd /root/repo
KC_OV_TRACE=1 python3 bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-cli 2>&1 | grep "^\[ov\]" | tail -40
