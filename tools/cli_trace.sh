#!/bin/bash
# CLI phase timing on the cfg[1] FASTA (run on the GPU box): tools/cli_trace.sh
python3 - <<'PY'
import os, sys
sys.path.insert(0, "tests")
import gen
d = "/dev/shm/kc_e2e"; os.makedirs(d, exist_ok=True)
with open(d + "/cfg2.fa", "wb") as f:
    for i, r in enumerate(gen.random_records(100000, 1030, 1)):
        f.write(b">r%d\n%s\n" % (i, r))
PY
for i in 1 2; do KC_CLI_TRACE=1 kmercamel_amd/bin/kmercamel compute -k 31 -o /dev/shm/kc_e2e/out.fa /dev/shm/kc_e2e/cfg2.fa 2>&1 | grep "cli"; done
rm -rf /dev/shm/kc_e2e
