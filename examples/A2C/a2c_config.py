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
    # End of the linear learning-rate decay and of training.  The reference's 1e7 sample steps are 20,000 updates
    # of its 5 x 5 envs x 20 steps = 500 rows; an update here has env_num x 20 = 5,120 rows, so the same number
    # of updates is 1e8 sample steps.  Measured on one MI355X (profiles/r03_a2c_pong_256envs_*.log): with 1e7 the
    # 1,953 updates end at -20.2 (the policy never leaves uniform); with 1e8 Pong crosses 0 after 1.85e7 steps
    # (130 s) and stands at +20.2 after 3.3e7 (230 s, 570 k frames/s).
    max_sample_steps=int(1e8),
    start_lr=0.001,
    gamma=0.99,
    vf_loss_coeff=0.5,
    entropy_coeff_scheduler=[(0, -0.01)],    # (train_step, coefficient) pairs, piecewise constant
    get_remote_metrics_interval=10,
    log_metrics_interval_s=10,
)
_learner['lambda'] = 1.0                 # GAE lambda; 1.0 = plain n-step returns

config = {**_where, **_actors, **_learner}
