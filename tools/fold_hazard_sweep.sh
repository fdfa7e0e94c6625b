B=tools/_build/fold_hazard_repro
for busy in 0 32 64 96 128 192 256; do $B 3000 $busy 1 0 1 1; done
for busy in 64 128; do $B 3000 $busy 0 0 0 1; $B 3000 $busy 1 1 1 1; $B 3000 $busy 1 2 1 1; $B 3000 $busy 1 0 1 0; done
