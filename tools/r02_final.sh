#!/bin/bash
set -u
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -2
bash tools/profile_engine_ops.sh
bash tools/r02_pmc.sh | tail -3
