"""reference settings/deepmimic/deepmimic_base.py:3-24 (values only; TF callables become names)."""
network_opts = {
    "common_net_shape": [],
    "critic_net_shape": [1024, 512],
    "actor_net_shape": [1024, 512],
    "weight_initializer": None,      # truncated_normal(0.0, 0.01), deepmimic_base.py:7
    "activator": "relu6",            # tf.nn.relu6, deepmimic_base.py:8
    "value_loss_coef": 0.5,

    "gamma": 0.95,

    "normalize_state": True,
    "clip_state": 5.,
    "normalize_value": False,
    "clip_value": None,
    "normalize_advantage": False,
    "clip_advantage": None,
}

worker_opts = {
    "lr_critic": 1e-4,
    "lr_actor": 5e-6
}
