#!/bin/bash
bash tools/pmc_icub.sh r02 2>&1 | grep -E "valu_per_wave|valu_over|wait_any_over" | head -6
