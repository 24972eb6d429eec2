#!/bin/bash
bash tools/measure_round.sh r03_m1 > gpurun_out/r03_m1.log 2>&1
tail -45 gpurun_out/r03_m1.log
