#!/bin/bash
# Throughput of the BASELINE.json / SURVEY §8(d) configurations on one GPU (no CPU baseline, no kernel timing).
run() { echo -n "$* : "; python bench.py --no-cpu-baseline --no-kernel-timing --no-secondary "$@" 2>/dev/null | tail -1 | python -c "import sys,json; j=json.loads(sys.stdin.read()); print(j['value'], 'img/s', j['ms_per_step'], 'ms')"; }
run --arch ViT-B/32 --batch 32 --steps 30
run
run --cut
run --trim-eot
run --dtype bf16
run --method vpt --classes 1000 --steps 10
run --method upt --classes 2191 --steps 6 --warmup 2
run --method upt --classes 2191 --cut --steps 6 --warmup 2
run --arch ViT-L/14@336px --method upt --classes 1151 --batch 128 --steps 4 --warmup 2
run --arch ViT-L/14@336px --method upt --classes 1151 --batch 128 --cut --steps 4 --warmup 2
run --method upt --classes 2191 --multitask --steps 6 --warmup 2
