#!/bin/bash
# Dev tool (GPU box): examples/IMPALA/train.py as a TWO-RANK data-parallel job on ONE GPU — the ranks share the device
# (gloo process group, the gradient bucket exchanged through HIP IPC slots: parl_amd.dist.SharedDeviceAllReduce), each
# with its own 512 envs.  Every update: each rank contributes <train_batch> rows, the gradients are SUMMED, the
# global-norm clip (40) is applied to the reduced gradient, every rank takes the same Adam step with the
# reference's lr / entropy schedule (the reference has ONE learner: train.py:90-121, impala_config.py:31).
# Usage: tools/dp_learning_run.sh <train_batch rows per rank> <minutes> <log prefix>
set -u
R=$(cd $(dirname $0)/.. && pwd)
TB=$1; MIN=$2; OUT=$3
PORT=$((29600 + RANDOM % 300))
for r in 0 1; do
  env PYTHONPATH=$R HSA_ENABLE_IPC_MODE_LEGACY=0 WORLD_SIZE=2 RANK=$r LOCAL_RANK=$r MASTER_ADDR=127.0.0.1 MASTER_PORT=$PORT \
      PARL_AMD_SHARE_GPU=1 PARL_AMD_DIST_BACKEND=gloo GPU_MAX_HW_QUEUES=2 \
      python $R/examples/IMPALA/train.py --minutes $MIN --env-num 512 --train-batch-size $TB --log-interval 10 --seed 1 \
      > ${OUT}_rank$r.log 2>&1 &
done
wait
grep -h "INFO\] {'sample_steps'" ${OUT}_rank0.log | sed -e "s/.*INFO\] //" > ${OUT}.log
echo "# rank 1:" >> ${OUT}.log
grep -h "INFO\] {'sample_steps'" ${OUT}_rank1.log | sed -e "s/.*INFO\] //" >> ${OUT}.log
tail -2 ${OUT}_rank0.log
