"""examples/PPO/mujoco_config.py of the reference (same keys / values)."""
mujoco_config = {
    'env': 'HalfCheetah-v2', 'continuous_action': True, 'env_num': 1, 'seed': None, 'xparl_addr': None,
    'train_total_steps': int(1e6), 'step_nums': 2048, 'num_minibatches': 32, 'update_epochs': 10,
    'eval_episode': 3, 'test_every_steps': int(5e3),
    'initial_lr': 3e-4, 'lr_decay': True, 'clip_param': 0.2, 'entropy_coef': 0.0,
}
