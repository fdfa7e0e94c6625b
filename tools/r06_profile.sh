#!/bin/bash
# Round-6 measurement bundle (GPU box, via gpurun): driver-style bench line (with secondary_configs), rocprofv3 kernel stats of the
# headline step, of the two towers alone and of BASELINE configs [0], [2], [3], [4]; PMC passes of the headline (HBM traffic: FETCH_SIZE /
# WRITE_SIZE in separate passes; SQ counters); config sweep; parity reports; clock / power of the headline loop.
#   bash tools/r06_profile.sh [tag]   -> gpurun_out/<tag>/   (copy what is judged into profiles/)
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
TAG=${1:-r06}; O=gpurun_out/$TAG; mkdir -p $O
timeout 900 python bench.py 2>$O/bench.err | tail -1 > $O/bench_line.json
bash tools/r03_profile_headline.sh $TAG > /dev/null 2>&1
timeout 600 rocprofv3 --kernel-trace --stats -d $O/trace_img -o t -- python tools/image_bench.py > /dev/null 2>&1
python tools/rocpd_summary.py $(ls $O/trace_img/*.db | head -1) > $O/image_tower_kernels.md
rm -rf $O/trace_img
pmc() {   # name, bench args...
  local name=$1; shift
  timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf_$name -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --no-trim-extra --no-secondary "$@" > /dev/null 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw_$name -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --no-trim-extra --no-secondary "$@" > /dev/null 2>&1
  python tools/traffic_from_pmc.py $(ls $O/pf_$name/*.db | head -1) $(ls $O/pw_$name/*.db | head -1) "$name: bench.py $*" > $O/gemm_hbm_traffic_$name.json
  timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_LDS_IDX_ACTIVE SQ_LDS_BANK_CONFLICT SQ_BUSY_CYCLES --kernel-trace -d $O/ps_$name -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --no-trim-extra --no-secondary "$@" > /dev/null 2>&1
  python tools/mfma_util.py $(ls $O/ps_$name/*.db | head -1) > $O/kernel_pmc_$name.md
  rm -rf $O/pf_$name $O/pw_$name $O/ps_$name
}
pmc headline --steps 3 --warmup 2
if [ "${FULL:-1}" = "1" ]; then
  bash tools/r03_profile_cfgs.sh $TAG > /dev/null 2>&1
  pmc cfg3 --method vpt --classes 1000 --steps 2 --warmup 1
  bash tools/config_sweep.sh > $O/config_sweep.txt 2>&1
fi
python tools/parity_report.py fp16 2>&1 | grep -v amdgpu.ids > $O/parity_fp16.txt
python tools/inference_parity.py fp16 2>&1 | grep -v amdgpu.ids > $O/inference_parity.txt
# clock / power while the headline loop runs for ~20 s
bash tools/power_probe.sh $O/power_clock_step.txt python bench.py --steps 1400 --warmup 20 --no-cpu-baseline --no-kernel-timing --no-trim-extra --no-secondary
python tools/chain_probe.py 2>&1 | grep -v amdgpu.ids > $O/chain_probe.txt
ls -la $O
timeout 1200 python -m pytest tests -m gpu -q 2>&1 | tail -5 > $O/pytest_gpu.txt
tools/_build/pkfma_hazard 400 20 > $O/pkfma_hazard_micro.txt 2>&1
