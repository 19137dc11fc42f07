#!/bin/bash
for N in 32768 131072; do
timeout 900 python tools/icub_push_soak.py --envs $N --steps 900 2>&1 | tail -1 | cut -c1-330
PBRE_ICUB_LANE=0 timeout 900 python tools/icub_push_soak.py --envs $N --steps 900 2>&1 | tail -1 | cut -c1-330
done
