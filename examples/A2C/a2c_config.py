"""A2C on the device env.  The dictionary has the keys the reference's example reads
(examples/A2C/a2c_config.py:15-44) with its learner hyper-parameters; what differs is where the actors live:
`actor_num` in-process actors, each with `env_num` GPU-resident envs (BASELINE configs[1]: 256 vectorised
envs on one MI355X) instead of 5 CPU processes x 5 envs."""

_where = dict(
    master_address='localhost:8110',     # only handed to parl.connect(); nothing listens
    env_name='PongNoFrameskip-v4',
    env_dim=84,                          # the A2C model's input (AtariModel84)
)

_actors = dict(
    actor_num=1,
    env_num=256,
    sample_batch_steps=20,               # n of the n-step return
)

_learner = dict(
    # The reference's value (examples/A2C/a2c_config.py:33): end of the linear learning-rate decay and of training,
    # in sample steps of ITS actor pool (5 actors x 5 envs x 20 steps = 500 rows per update -> 20,000 updates).
    # train.py keeps the number of UPDATES, not of samples, when the pool is bigger (`--horizon`, see there).
    max_sample_steps=int(1e7),
    start_lr=0.001,
    gamma=0.99,
    vf_loss_coeff=0.5,
    entropy_coeff_scheduler=[(0, -0.01)],    # (train_step, coefficient) pairs, piecewise constant
    get_remote_metrics_interval=10,
    log_metrics_interval_s=10,
)
_learner['lambda'] = 1.0                 # GAE lambda; 1.0 = plain n-step returns

config = {**_where, **_actors, **_learner}
