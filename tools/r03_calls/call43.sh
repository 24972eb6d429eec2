#!/bin/bash
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "side_streams or compact" 2>&1 | tail -5
