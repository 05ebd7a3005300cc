#!/bin/bash
# last call of the round: smoke() and the whole GPU suite on the final tree
cd $GRAFT_REPO_ROOT; mkdir -p gpurun_out
timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('SMOKE_OK')" 2>&1 | tail -2
( time timeout 600 python -m pytest tests -m gpu -q ) > gpurun_out/r02g_pytest.log 2>&1; echo "pytest rc=$?"; grep -E "passed|failed" gpurun_out/r02g_pytest.log | tail -1
