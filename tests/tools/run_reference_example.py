"""Runs one of the reference's OWN example directories UNMODIFIED (by path) through compat/{paddle,parl,gym} in
THIS process and prints one JSON line; meant to be started as a subprocess by tests/test_reference_scripts.py
(the examples' actor / learner threads never stop, the process ends with os._exit).

    python tests/tools/run_reference_example.py <impala|a2c_paddle> <script dir> [--cpu-doubles] [--steps N]

--cpu-doubles: a box without a GPU — the two GPU-only pieces (the env vector, calc_gae) are replaced by test
doubles backed by the CPU oracle, everything else (the scripts, compat layers, Agent / Algorithm / Model, learn)
runs for real.  Only the config DICT is shrunk (data, not code)."""
import argparse
import importlib
import json
import os
import sys
import time

sys.dont_write_bytecode = True  # modules are imported from /root/reference by path: never leave a __pycache__ there

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('kind', choices=['impala', 'a2c_paddle'])
    ap.add_argument('script_dir')
    ap.add_argument('--cpu-doubles', action='store_true')
    ap.add_argument('--steps', type=int, default=3)
    args = ap.parse_args()
    sys.path[:0] = [args.script_dir, os.path.join(ROOT, 'compat'), os.path.join(ROOT, 'tests'), ROOT]
    os.chdir('/tmp')
    import numpy as np
    import torch
    if args.cpu_doubles:
        sys.path.insert(0, os.path.join(ROOT, 'tests'))
        from test_reference_scripts import _OracleVecEnvDouble, _oracle_calc_gae
        import parl_amd.env.vector_env as ve
        import parl_amd.utils.rl_utils as ru
        ve.DeviceVectorEnv = _OracleVecEnvDouble
        ru.calc_gae = _oracle_calc_gae
        # IMPALA's V-trace kernel has no CPU form either: the oracle's restatement stands in, under the autograd
        # graph of the reference formulas (IMPALA.fused_loss = False)
        from oracle import c_oracle
        import parl_amd.ops as ops
        from parl_amd.algorithms.impala import impala as _imp

        def _vtrace_double(bl, tl, act, rew, dones, val, gamma, crho=1.0, cpg=1.0, time_major=True):
            vs, pg, _, _ = c_oracle.vtrace_from_logits(bl.numpy(), tl.numpy(), act.numpy(), rew.numpy(), dones.numpy(),
                                                       val.numpy(), gamma, crho, cpg, time_major=time_major)
            return torch.from_numpy(vs), torch.from_numpy(pg)

        ops.vtrace_from_logits = _vtrace_double
        _init = _imp.IMPALA.__init__

        def _init_unfused(self, *a, **k):
            _init(self, *a, **k)
            self.fused_loss = False

        _imp.IMPALA.__init__ = _init_unfused
        torch.cuda.is_available = lambda: False
        torch.set_num_threads(4)
    import paddle  # compat/paddle
    import parl  # compat/parl == parl_amd
    import parl_amd
    assert parl is parl_amd and paddle.__file__.startswith(os.path.join(ROOT, 'compat'))
    train = importlib.import_module('train')
    assert os.path.dirname(os.path.abspath(train.__file__)) == os.path.abspath(args.script_dir)
    out = {'kind': args.kind, 'script_dir': args.script_dir}
    if args.kind == 'impala':
        cfg = dict(importlib.import_module('impala_config').config)
        cfg.update(actor_num=2, env_num=2, sample_batch_steps=10, train_batch_size=40, sample_queue_max_size=4,
                   get_remote_metrics_interval=1, params_broadcast_interval=1)
        learner = train.Learner(cfg)  # starts its learn thread and one sampling thread per actor
        w0 = {k: v.copy() for k, v in learner.agent.get_weights().items()}
        t0 = time.time()
        while learner.total_loss_stat.count < args.steps and time.time() - t0 < 600:
            time.sleep(0.2)
        learner.log_metrics()
        w1 = learner.agent.get_weights()
        out.update(learn_steps=int(learner.total_loss_stat.count), sample_total_steps=int(learner.sample_total_steps),
                   total_loss=float(learner.total_loss_stat.mean), kl=float(learner.kl_stat.mean),
                   lr=float(learner.lr), entropy_coeff=float(learner.entropy_coeff),
                   weights_changed=bool(any(np.abs(w1[k] - w0[k]).max() > 0 for k in w0)),
                   total_params_sync=int(learner.total_params_sync),
                   device=str(next(learner.agent.alg.model.parameters()).device))
    else:
        cfg = dict(importlib.import_module('a2c_config').config)
        cfg.update(actor_num=2, env_num=2, sample_batch_steps=5, max_sample_steps=10**6)
        learner = train.Learner(cfg)
        w0 = {k: v.copy() for k, v in learner.agent.get_weights().items()}
        for _ in range(args.steps):
            learner.step()
        learner.log_metrics()
        w1 = learner.agent.get_weights()
        out.update(learn_steps=int(learner.total_loss_stat.count), sample_total_steps=int(learner.sample_total_steps),
                   total_loss=float(learner.total_loss_stat.mean), lr=float(learner.lr),
                   entropy_coeff=float(learner.entropy_coeff),
                   weights_changed=bool(any(np.abs(w1[k] - w0[k]).max() > 0 for k in w0)),
                   device=str(next(learner.agent.alg.model.parameters()).device))
    print('RESULT ' + json.dumps(out), flush=True)
    os._exit(0)


if __name__ == '__main__':
    main()
