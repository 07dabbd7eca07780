#!/usr/bin/env bash
# Gate for every commit that touches cattus_amd/csrc: rebuild what is stale, warm the hazard-audit cache (kernels.hip: two minutes
# after a change, nothing otherwise) and run the CPU suite.  A kernel change is not committed while this is red.
#     scripts/precommit.sh            (from anywhere inside the repository)
set -euo pipefail
cd "$(dirname "$0")/.."
python -c "import __graft_entry__ as g; g.build()"
python -m pytest tests/ -x -q -m "not gpu" "$@"
