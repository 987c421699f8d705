#!/bin/bash
# soak: the fresh-process probe N times on one box -- every process must print the same two hash lines
ulimit -c 0
N=${1:-40}
mkdir -p gpurun_out/soak
for i in $(seq 1 $N); do
  timeout 300 python tests/fresh_process_probe.py > gpurun_out/soak/p_$i.log 2>&1; echo "rc=$?" >> gpurun_out/soak/p_$i.log
done
grep -h "^step 0" gpurun_out/soak/p_*.log | sed 's/alone-vs.*//' | sort | uniq -c
grep -h "^step 1" gpurun_out/soak/p_*.log | sed 's/alone-vs.*//' | sort | uniq -c
grep -h "^rc=" gpurun_out/soak/p_*.log | sort | uniq -c
