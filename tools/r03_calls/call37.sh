#!/bin/bash
mkdir -p gpurun_out/r03_c37
python -m pytest tests -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r03_c37/pytest_gpu.txt
python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')" 2>&1 | tail -2
