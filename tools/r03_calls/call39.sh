#!/bin/bash
python -m pytest tests/test_gpu_cli.py -m gpu -x -q 2>&1 | tail -12
