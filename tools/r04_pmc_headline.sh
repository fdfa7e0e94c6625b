#!/bin/bash
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
O=gpurun_out/r04; mkdir -p $O
timeout 600 rocprofv3 --pmc FETCH_SIZE --kernel-trace -d $O/pf -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --no-trim-extra --steps 3 --warmup 2 > /dev/null 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --kernel-trace -d $O/pw -o p -- python bench.py --no-cpu-baseline --no-kernel-timing --no-trim-extra --steps 3 --warmup 2 > /dev/null 2>&1
python tools/traffic_from_pmc.py $(ls $O/pf/*.db | head -1) $(ls $O/pw/*.db | head -1) "headline: bench.py --steps 3 --warmup 2" > $O/gemm_hbm_traffic_headline.json
rm -rf $O/pf $O/pw
cat $O/gemm_hbm_traffic_headline.json
