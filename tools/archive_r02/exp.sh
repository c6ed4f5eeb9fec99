#!/bin/bash
export TMPDIR=/tmp
for r in 1 2; do python bench.py --gpu-decode --no-cpu-baseline --steps 40 2>/dev/null | cut -c60-200; done
