"""IMPALA learner — the structure of examples/IMPALA/train.py:34-258 (Learner with a sample
queue, a learn thread, actor threads, schedulers, WindowStat metrics) on the device path.

    python examples/IMPALA/train.py [--env-num E] [--minutes M] [--train-batch-size 1000] [--threads]

Default: the learner hyper-parameters of the reference on parl_amd.rollout.AsyncActorLearner — actors and
learner on two HIP streams of one host thread, every `train_batch_size`-row update one hipGraph replay
(train_batch_size 1000 = the reference's, impala_config.py:31), the actors refreshing their weights inside
the rollout: 2.7 M frames/s, 680 updates/s on one MI355X.  --threads: the reference's own structure instead
(`Learner` below: a learn thread fed by a queue, one sampling thread per `@parl.remote_class` Actor).

The reference file imports paddle (`paddle.io.DataLoader.from_generator`, train.py:129-130); this
twin feeds `agent.learn` directly from the queue."""
import argparse
import os
import queue
import sys
import threading
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

import parl_amd as parl  # noqa: E402
from actor import Actor  # noqa: E402
from atari_agent import AtariAgent  # noqa: E402
from parl_amd.models import AtariModel42 as AtariModel  # noqa: E402  (torch twin of examples/IMPALA/atari_model.py:21-90)
from parl_amd.env import GAMES  # noqa: E402
from parl_amd.utils import logger, summary  # noqa: E402
from parl_amd.utils.scheduler import PiecewiseScheduler  # noqa: E402
from parl_amd.utils.time_stat import TimeStat  # noqa: E402
from parl_amd.utils.window_stat import WindowStat  # noqa: E402


class Learner(object):
    def __init__(self, config):
        self.config = config
        self.sample_data_queue = queue.Queue(maxsize=config['sample_queue_max_size'])
        self.device = torch.device('cuda')
        act_dim = 6 if GAMES[config['env_name']][0] == 'pong' else 4

        model = AtariModel(act_dim)
        algorithm = parl.algorithms.IMPALA(
            model, sample_batch_steps=config['sample_batch_steps'], gamma=config['gamma'],
            vf_loss_coeff=config['vf_loss_coeff'], clip_rho_threshold=config['clip_rho_threshold'],
            clip_pg_rho_threshold=config['clip_pg_rho_threshold'])
        self.agent = AtariAgent(algorithm, device=self.device)

        self.lr, self.entropy_coeff = None, None
        self.lr_scheduler = PiecewiseScheduler(config['lr_scheduler'])
        self.entropy_coeff_scheduler = PiecewiseScheduler(config['entropy_coeff_scheduler'])
        self.total_loss_stat = WindowStat(100)
        self.pi_loss_stat = WindowStat(100)
        self.vf_loss_stat = WindowStat(100)
        self.entropy_stat = WindowStat(100)
        self.kl_stat = WindowStat(100)
        self.learn_time_stat = TimeStat(100)
        self.start_time = None
        self.learn_steps = 0
        self.sample_total_steps = 0
        self.remote_metrics_queue = queue.Queue()
        self.stop = False

        self.learn_thread = threading.Thread(target=self.run_learn, daemon=True)
        self.learn_thread.start()
        self.create_actors()

    # ------------------------------------------------------------------ learner side
    def run_learn(self):
        T = self.config['sample_batch_steps']
        seqs = max(1, self.config['train_batch_size'] // T)
        while not self.stop:
            try:
                batch, consumed = self.sample_data_queue.get(timeout=0.5)
            except queue.Empty:
                continue
            E = batch['actions'].numel() // T
            view = lambda k: batch[k].reshape((T, E) + tuple(batch[k].shape[1:]))  # noqa: E731
            obs, act, bl, rew, done = [view(k) for k in ('obs', 'actions', 'behaviour_logits', 'rewards', 'dones')]
            for e0 in range(0, E, seqs):  # train_batch_size rows per update, whole sequences
                sl = slice(e0, min(E, e0 + seqs))
                self.lr = self.lr_scheduler.step()
                self.entropy_coeff = self.entropy_coeff_scheduler.step()
                flat = lambda t: t[:, sl].reshape((-1, ) + tuple(t.shape[2:]))  # noqa: E731
                with self.learn_time_stat:
                    total_loss, pi_loss, vf_loss, entropy, kl = self.agent.learn(
                        flat(obs), flat(act), flat(bl), flat(rew), flat(done), self.lr, self.entropy_coeff,
                        time_major=True)
                self.learn_steps += 1
                self.total_loss_stat.add(total_loss)
                self.pi_loss_stat.add(pi_loss)
                self.vf_loss_stat.add(vf_loss)
                self.entropy_stat.add(entropy)
                self.kl_stat.add(kl)
            self.sample_total_steps += T * E
            consumed.set()  # this actor's rollout buffers may be overwritten now

    # ------------------------------------------------------------------ actor side
    def create_actors(self):
        parl.connect(self.config['master_address'])
        # one event PER ACTOR: an actor's rollout buffers are reused, and the queued batch aliases
        # them, so only the learner's "done with YOUR batch" may release that actor
        self.consumed = [threading.Event() for _ in range(self.config['actor_num'])]
        for e in self.consumed:
            e.set()
        self.actors = []
        for i in range(self.config['actor_num']):
            if self.start_time is None:
                self.start_time = time.time()
            t = threading.Thread(target=self.run_remote_sample, args=(i, ), daemon=True)
            t.start()
            self.actors.append(t)

    def shutdown(self):
        """stop the actor / learner threads and wait for in-flight GPU work"""
        self.stop = True
        for e in self.consumed:
            e.set()
        for t in self.actors + [self.learn_thread]:
            t.join(timeout=30)
        torch.cuda.synchronize()

    def run_remote_sample(self, actor_id):
        remote_actor = Actor(self.config, actor_id, model=self.agent.alg.model, device=self.device)
        cnt = 0
        while not self.stop:
            # the rollout buffers are reused: wait until the learner has consumed the previous batch
            consumed = self.consumed[actor_id]
            consumed.wait()
            consumed.clear()
            batch = remote_actor.sample().get()
            self.sample_data_queue.put((batch, consumed))
            cnt += 1
            if cnt % self.config['get_remote_metrics_interval'] == 0:
                metrics = remote_actor.get_metrics().get()
                if metrics:
                    self.remote_metrics_queue.put(metrics)

    # ------------------------------------------------------------------ logging (train.py:196-247)
    def log_metrics(self):
        if self.start_time is None:
            return None
        rewards, steps = [], []
        while True:
            try:
                m = self.remote_metrics_queue.get_nowait()
            except queue.Empty:
                break
            rewards += m['episode_rewards']
            steps += m['episode_steps']
        elapsed = time.time() - self.start_time
        metric = {
            'sample_steps': self.sample_total_steps,
            'mean_episode_rewards': float(np.mean(rewards)) if rewards else None,
            'mean_episode_steps': float(np.mean(steps)) if steps else None,
            'episodes': len(rewards),
            'learn_steps': self.learn_steps,
            'total_loss': self.total_loss_stat.mean,
            'pi_loss': self.pi_loss_stat.mean,
            'vf_loss': self.vf_loss_stat.mean,
            'entropy': self.entropy_stat.mean,
            'kl': self.kl_stat.mean,
            'learn_time_s': self.learn_time_stat.mean,
            'elapsed_time_s': int(elapsed),
            'env_frames_per_s': 4 * self.sample_total_steps / max(elapsed, 1e-9),
            'lr': self.lr,
            'entropy_coeff': self.entropy_coeff,
        }
        for key, value in metric.items():
            if value is not None:
                summary.add_scalar(key, value, self.sample_total_steps)
        logger.info(metric)
        return metric


class PipelineLearner(object):
    """the Learner's hyper-parameters, schedulers and metrics on AsyncActorLearner(train_batch_size=...)"""

    def __init__(self, config):
        from parl_amd.env import DeviceVectorEnv
        from parl_amd.rollout import AsyncActorLearner
        from parl_amd import dist as pdist
        self.config = config
        # one process per GPU (torchrun env): envs sharded by rank, one gradient all-reduce per update
        # (parl_amd.dist; the reference has one learner GPU and no collectives, train.py:155-194)
        self.rank, local, self.world = pdist.init()
        if os.environ.get('PARL_AMD_SHARE_GPU'):  # fewer GPUs than ranks (a test box): ranks share devices
            local = local % torch.cuda.device_count()
        dev = torch.device('cuda', local)
        torch.cuda.set_device(dev)
        E, T = config['env_num'] * config['actor_num'], config['sample_batch_steps']
        elastic = config.get('elastic_launches', 'Breakout' in config['env_name'])
        self.env = DeviceVectorEnv(config['env_name'], E, dim=config['env_dim'], horizon=4 * T + 32 if elastic else T,
                                   seed=config.get('seed', 0), env_id0=self.rank * E, device=dev)
        if config['env_dim'] == 84:  # the north-star frame size: the A2C example's network (examples/A2C/atari_model.py)
            from parl_amd.models import AtariModel84
            model = AtariModel84(self.env.act_dim).to(dev)
        else:
            model = AtariModel(self.env.act_dim).to(dev)
        self.alg = parl.algorithms.IMPALA(
            model, sample_batch_steps=T, gamma=config['gamma'], vf_loss_coeff=config['vf_loss_coeff'],
            clip_rho_threshold=config['clip_rho_threshold'], clip_pg_rho_threshold=config['clip_pg_rho_threshold'])
        pdist.broadcast_model(model)
        if pdist.active():
            self.alg.grad_hook = pdist.FlatGradAllReduce(model)
        self.pipe = AsyncActorLearner(self.alg, [self.env], T, seed=config.get('seed', 0) + 1000, elastic=elastic,
                                      train_batch_size=config['train_batch_size'])
        self.lr_scheduler = PiecewiseScheduler(config['lr_scheduler'])
        self.entropy_coeff_scheduler = PiecewiseScheduler(config['entropy_coeff_scheduler'])
        self.sample_total_steps = 0
        self.start_time = time.time()
        self.T, self.E = T, E
        self.pipe.prime()

    def step(self):
        self.pipe.step(self.lr_scheduler, self.entropy_coeff_scheduler)
        self.sample_total_steps += self.T * self.E

    def log_metrics(self):
        stats, n_upd = self.pipe.pop_learn_stats()
        n, mean_r, mean_l = self.pipe.pop_episode_stats()
        elapsed = time.time() - self.start_time
        metric = {
            'sample_steps': self.sample_total_steps,
            'mean_episode_rewards': mean_r, 'mean_episode_steps': mean_l, 'episodes': int(n),
            'learn_steps': self.pipe.updates,
            'elapsed_time_s': int(elapsed),
            'env_frames_per_s': 4 * self.sample_total_steps / max(elapsed, 1e-9),
            'learner_updates_per_s': self.pipe.updates / max(elapsed, 1e-9),
            'lr': self.lr_scheduler.cur_value, 'entropy_coeff': self.entropy_coeff_scheduler.cur_value,
        }
        if stats:
            metric.update(dict(zip(('total_loss', 'pi_loss', 'vf_loss', 'entropy', 'kl'), stats)))
        for key, value in metric.items():  # the summary scalars the reference's Learner logs (train.py:214-238)
            if value is not None:
                summary.add_scalar(key, value, self.sample_total_steps)
        logger.info(metric)
        return metric

    def shutdown(self):
        self.pipe.synchronize()
        self.env.check_faults()


if __name__ == '__main__':
    from impala_config import config
    ap = argparse.ArgumentParser()
    ap.add_argument('--minutes', type=float, default=None, help='stop after this many minutes')
    ap.add_argument('--env-num', type=int, default=None)
    ap.add_argument('--env-name', default=None, help='PongNoFrameskip-v4 (config default) or BreakoutNoFrameskip-v4')
    ap.add_argument('--train-batch-size', type=int, default=None)
    ap.add_argument('--env-dim', type=int, default=None, choices=(42, 84),
                    help='observation size: 42 (config default, AtariModel42) or 84 (AtariModel84; pipeline mode)')
    ap.add_argument('--log-interval', type=float, default=None)
    ap.add_argument('--seed', type=int, default=None,
                    help='seed of the network initialisation, the envs and the sampler (default: the envs / sampler use 0, '
                    'the initialisation is unseeded as in the reference); the pipeline mode with synchronous launches '
                    '(Pong) is then one reproducible run, elastic launches (Breakout) still follow the host\'s polling')
    ap.add_argument('--threads', action='store_true',
                    help='the reference\'s thread-per-actor structure (class Learner) instead of the stream pipeline')
    ap.add_argument('--pipeline', action='store_true', help='(default; kept for older command lines)')
    args = ap.parse_args()
    if args.env_name:
        config['env_name'] = args.env_name
        config['experiment_name'] = args.env_name.split('NoFrameskip')[0]
    if args.env_num:
        config['env_num'] = args.env_num
    if args.train_batch_size:
        config['train_batch_size'] = args.train_batch_size
    if args.log_interval:
        config['log_metrics_interval_s'] = args.log_interval
    if args.env_dim:
        assert args.env_dim == 42 or not args.threads, '--env-dim 84 runs in the pipeline mode'
        config['env_dim'] = args.env_dim
    if args.seed is not None:
        config['seed'] = args.seed
        torch.manual_seed(args.seed)
        np.random.seed(args.seed)
    if not args.threads:
        from parl_amd import dist as pdist
        learner = PipelineLearner(config)
        t0 = t_log = time.time()
        # Every update holds a collective in a data-parallel run, so all ranks must stop in the SAME iteration: the
        # stop flag is the max over ranks.  That all-reduce ends in a blocking .item() behind everything the rank
        # has enqueued — taken every iteration it would drain the learner pass each step and remove the
        # host-runs-ahead overlap of actors and learner — so it is taken every `stop_check_every`
        # iterations (the ranks count iterations identically; one iteration is ~40 ms); a single process reads
        # its own clock every iteration as before.
        every = int(config.get('stop_check_every', 16)) if pdist.active() else 1
        it = 0

        def time_is_up():
            return float(args.minutes is not None and time.time() - t0 >= args.minutes * 60)

        while not (it % every == 0 and pdist.all_reduce_max_scalar(time_is_up())):
            it += 1
            learner.step()
            if time.time() - t_log >= config['log_metrics_interval_s']:
                learner.log_metrics()
                t_log = time.time()
        learner.log_metrics()
        learner.shutdown()
        sys.stdout.flush()
        sys.exit(0)  # no proxy threads in this mode: a normal interpreter exit (loggers and summary files flushed)
    learner = Learner(config)
    assert config['log_metrics_interval_s'] > 0
    t0 = time.time()
    while args.minutes is None or time.time() - t0 < args.minutes * 60:
        time.sleep(config['log_metrics_interval_s'])
        learner.log_metrics()
    learner.shutdown()
    sys.stdout.flush()
    os._exit(0)  # daemon proxy threads (parl_amd.remote) never return; skip interpreter teardown
