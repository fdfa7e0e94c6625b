#!/bin/bash
# Round-2 measurement bundle (run on the GPU box through gpurun): bench line, rocprofv3 kernel stats of the same command,
# PMC passes (HBM traffic: FETCH_SIZE / WRITE_SIZE in separate passes; MFMA utilisation: SQ counters), config sweep.
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r02; mkdir -p $O
timeout 400 python bench.py --steps 30 --warmup 8 2>$O/bench.err | tail -1 > $O/bench_line.json
timeout 400 rocprofv3 --kernel-trace --stats -d $O/trace -o bench -- python bench.py --steps 12 --warmup 4 --no-cpu-baseline > $O/bench_under_rocprof.log 2>&1
grep "^{\"metric" $O/bench_under_rocprof.log | tail -1 > $O/bench_line_under_rocprof.json
python tools/rocpd_summary.py $(ls $O/trace/*.db | head -1) > $O/bench_kernel_stats.md
timeout 400 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pmc_fetch -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
timeout 400 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pmc_write -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python tools/traffic_from_pmc.py $(ls $O/pmc_fetch/*.db | head -1) $(ls $O/pmc_write/*.db | head -1) > $O/gemm_hbm_traffic.json
timeout 400 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d $O/pmc_sq -o p -- python bench.py --steps 3 --warmup 2 --no-cpu-baseline --no-kernel-timing > /dev/null 2>&1
python tools/mfma_util.py $(ls $O/pmc_sq/*.db | head -1) > $O/gemm_pmc.md
bash tools/config_sweep.sh > $O/config_sweep.txt 2>&1
timeout 400 python bench.py --no-cpu-baseline --grad-precision fast --steps 20 --warmup 5 2>/dev/null | tail -1 > $O/bench_line_fast_mode.json
rm -rf $O/trace $O/pmc_fetch $O/pmc_write $O/pmc_sq
ls -la $O
