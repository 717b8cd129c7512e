"""IMPALA on the device env.  Keys and learner hyper-parameters are those the reference's example reads
(examples/IMPALA/impala_config.py:15-46).  `actor_num` actors live in THIS process and each owns `env_num`
GPU-resident envs (the reference: 32 CPU actor processes x 5 envs)."""

_where = dict(
    experiment_name='Pong',
    master_address='localhost:8010',     # only handed to parl.connect(); the actors are in-process
    env_name='PongNoFrameskip-v4',
    env_dim=42,                          # AtariModel42
)

_actors = dict(
    actor_num=1,
    env_num=1024,                        # BASELINE configs[2]: 1024 actors on one MI355X
    sample_batch_steps=50,
    params_broadcast_interval=1,
)

_learner = dict(
    # rows per learner update, whole sequences of sample_batch_steps: the reference's 1000.  A rollout of
    # env_num sequences is consumed as env_num // 20 updates.
    train_batch_size=1000,
    sample_queue_max_size=8,
    gamma=0.99,
    vf_loss_coeff=0.5,
    clip_rho_threshold=1.0,
    clip_pg_rho_threshold=1.0,
    lr_scheduler=[(0, 0.001), (20000, 0.0005), (40000, 0.0001)],     # (train_step, learning rate)
    entropy_coeff_scheduler=[(0, -0.01)],                            # (train_step, coefficient)
    get_remote_metrics_interval=1,
    log_metrics_interval_s=10,
)

config = {**_where, **_actors, **_learner}
