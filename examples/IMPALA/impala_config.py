"""IMPALA on the device env — same keys and learner hyper-parameters as the reference's
examples/IMPALA/impala_config.py:15-46.  Differences: `actor_num` actors live in THIS process, each
owning `env_num` GPU-resident envs (the reference: 32 CPU actor processes x 5 envs)."""
config = {
    'experiment_name': 'Pong',

    # ==========  remote config ==========
    'master_address': 'localhost:8010',  # kept for parl.connect(); actors are in-process

    # ==========  env config ==========
    'env_name': 'PongNoFrameskip-v4',
    'env_dim': 42,

    # ==========  actor config ==========
    'actor_num': 1,
    'env_num': 1024,
    'sample_batch_steps': 50,

    # ==========  learner config ==========
    # rows per learner update (whole sequences of sample_batch_steps): the reference's value; the rollout of
    # env_num sequences is consumed as env_num // 20 updates
    'train_batch_size': 1000,
    'sample_queue_max_size': 8,
    'gamma': 0.99,

    # learning rate adjustment schedule: (train_step, learning_rate)
    'lr_scheduler': [(0, 0.001), (20000, 0.0005), (40000, 0.0001)],

    # coefficient of policy entropy adjustment schedule: (train_step, coefficient)
    'entropy_coeff_scheduler': [(0, -0.01)],
    'vf_loss_coeff': 0.5,
    'clip_rho_threshold': 1.0,
    'clip_pg_rho_threshold': 1.0,
    'get_remote_metrics_interval': 1,
    'log_metrics_interval_s': 10,
    'params_broadcast_interval': 1,
}
