"""examples/PPO/atari_config.py of the reference (same keys / values)."""
atari_config = {
    'env': 'PongNoFrameskip-v4', 'continuous_action': False, 'env_num': 8, 'seed': None, 'xparl_addr': None,
    'train_total_steps': int(1e7), 'step_nums': 128, 'num_minibatches': 4, 'update_epochs': 4,
    'eval_episode': 3, 'test_every_steps': int(5e3),
    'initial_lr': 2.5e-4, 'lr_decay': True, 'clip_param': 0.1, 'entropy_coef': 0.01,
}
