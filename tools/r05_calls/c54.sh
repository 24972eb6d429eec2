#!/bin/bash
# round 5, call 54: where a first pass loses against a planned one: one engine, 3 750 alignments, both ways in turn
out=gpurun_out/r05_c54; mkdir -p $out
timeout 600 python tools/first_pass_cost.py 3750 > $out/first_pass_cost_3750.txt 2>&1; tail -12 $out/first_pass_cost_3750.txt | cut -c1-250
