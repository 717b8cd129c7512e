#!/bin/bash
# GPU box: the round's whole evidence set into gpurun_out/<tag>/ (copy what is judged into profiles/).
# Usage: tools/prof_round.sh <tag>
TAG=${1:-rXX}
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/$TAG
mkdir -p $O
cd $R
python -m pytest tests -m gpu -q > $O/pytest_gpu.log 2>&1; tail -1 $O/pytest_gpu.log
# the DRIVER's command line (round 3's verdict: the kept profile must come from it), then the same with --quick
bash tools/prof_bench.sh ${TAG}_bench --gpus 1 --steps 20 --warmup 5 > /dev/null 2>&1
bash tools/prof_bench.sh ${TAG}_bench_quick --gpus 1 --steps 20 --warmup 5 --quick > /dev/null 2>&1
python bench.py --gpus 1 --steps 20 --warmup 2 --quick --no-cpu-baseline > $O/bench_quick_warmup2.json 2> /dev/null
bash tools/actor_gaps.sh $O/actor_gaps.txt > /dev/null 2>&1
bash tools/prof_traffic.sh > $O/traffic.log 2>&1
bash tools/pmc_env.sh $O/env_pmc.log > /dev/null 2>&1
(python tools/emu_bench.py PongNoFrameskip-v4 1024,4096; python tools/emu_bench.py BreakoutNoFrameskip-v4 1024,4096) 2>&1 | grep "E=" > $O/emu_bench.log
if [ -f build_exp/regions.so ]; then (for g in PongNoFrameskip-v4 BreakoutNoFrameskip-v4; do PARL_HIP_LIB=$R/build_exp/regions.so python tools/env_regions.py $g 1024 40; done) 2>&1 | grep -v amdgpu > $O/env_regions.log; fi
python tools/microbench.py > $O/microbench_scans.json 2> $O/microbench.err
python tools/learner_bench.py --json $O/learner_bench.json > $O/learner_bench.log 2>&1
(bash tools/prof_heads_alone.sh tree; python tools/heads_beside_env.py; bash tools/pmc_heads.sh $O/heads_pmc_raw.log) > $O/heads_loss.log 2>&1
(python tools/ref_batch_probe.py; python tools/actor_chain_probe.py 1024 42; python tools/actor_chain_probe.py 1024 84) 2>&1 | grep -v amdgpu > $O/learner_actor_probes.log
(bash tools/learn84_prof.sh $O/learn84_kernels.txt 84 5120; bash tools/learn84_prof.sh $O/learn42_kernels.txt 42) > /dev/null 2>&1
if [ -f build_exp/convreg.so ]; then PARL_HIP_LIB=$R/build_exp/convreg.so python tools/conv_regions.py 51200 2>&1 | grep -v amdgpu > $O/conv12_bwd_regions.log; fi
python tools/ref_batch_probe.py 2>&1 | grep -v amdgpu | tail -5 > $O/ref_batch_probe.log
(python tools/conv12_scaling.py; python tools/conv84_scaling.py) 2>&1 | grep n_obs > $O/conv_scaling.log
if [ "${PROF_A2C:-0}" == "1" ]; then timeout 230 python examples/A2C/train.py --log-interval 10 --minutes 3.2 2>&1 | grep -v amdgpu > $O/learn_a2c_pong_256envs_first_3min.log; fi
ls $O
