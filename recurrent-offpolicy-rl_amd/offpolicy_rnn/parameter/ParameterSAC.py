"""`Parameter`: the flat flag set of the reference trainer (offpolicy_rnn/parameter/ParameterSAC.py:15-308).

Flag names, types, defaults and store_true semantics are the API surface that `main.py` and the `gen_tmuxp_*`
launchers rely on; they are declared once in FLAGS and turned into both an argparse parser and instance attributes.
"""
import argparse

from .._compat import ParameterTemplate, smart_logger


def str_or_int(value):
    try:
        return int(value)
    except ValueError:
        return value


S, I, F = str, int, float
# (name, type | 'flag' | ('list', type), default, help)
FLAGS = [
    ('env_name', S, 'HalfCheetah-v2', 'name of the environment to run'),
    ('alg_name', S, 'sac_mlp', 'name of the algorithm'),
    ('seed', I, 1, 'seed'),
    ('policy_lr', F, 3e-4, 'learning rate of the policy'),
    ('rnn_policy_lr', F, 1e-5, 'learning rate of the context-encoder part of the policy'),
    ('policy_l2_norm', F, 0.0, 'weight decay of the policy optimizer'),
    ('policy_update_per', I, 1, 'actor update every N critic updates'),
    ('policy_max_gradnorm', F, None, 'clip_grad_norm_ bound for the policy'),
    ('policy_embedding_max_gradnorm', F, None, 'clip_grad_value_ bound for the policy context encoder'),
    ('alpha_lr', F, 1e-2, 'learning rate of the entropy coefficient'),
    ('value_lr', F, 1e-3, 'learning rate of the critic'),
    ('rnn_value_lr', F, 1e-4, 'learning rate of the context-encoder part of the critic'),
    ('value_max_gradnorm', F, None, 'clip_grad_norm_ bound for the critic'),
    ('value_embedding_max_gradnorm', F, None, 'clip_grad_value_ bound for the critic context encoder'),
    ('value_l2_norm', F, 0.0, 'weight decay of the critic optimizer'),
    ('cuda_inference', 'flag', False, 'sample actions on the GPU'),
    ('backing_log', 'flag', False, 'back logs up to a remote machine'),
    ('reward_input', 'flag', False, 'feed the last reward to the context encoder'),
    ('last_state_input', 'flag', False, 'feed the last state to the context encoder'),
    ('randomize_mask', 'flag', False, 'keep only a random subset of valid loss positions'),
    ('random_trunc_traj', 'flag', False, 'randomly truncate sampled trajectories'),
    ('valid_number_post_randomized', I, 256, 'loss positions kept after mask randomisation'),
    ('policy_uni_model_input_mapping_dim', str_or_int, 0, 'state mapping width in front of the policy MLP'),
    ('value_uni_model_input_mapping_dim', str_or_int, 0, 'state/action mapping width in front of the critic MLP'),
    ('randomize_first_hidden', 'flag', False, 'random instead of zero initial hidden state'),
    ('randomize_training_initial_hidden', 'flag', False, 'perturb the pre-computed initial hidden state (slice trainer)'),
    ('no_alpha_auto_tune', 'flag', False, 'keep the entropy coefficient fixed'),
    ('no_last_action_input', 'flag', False, 'do not feed the last action to the context encoder'),
    ('state_action_encoder', 'flag', False, 'separate Linear encoders for state / action / reward inputs'),
    ('value_hidden_size', ('list', I), [256, 128], 'hidden widths of the critic MLP'),
    ('value_activations', ('list', S), ['relu', 'relu', 'linear'], 'activations of the critic MLP'),
    ('value_layer_type', ('list', S), ['fc', 'fc', 'fc'], 'layer ids of the critic MLP'),
    ('value_net_num', I, 2, 'number of critic networks'),
    ('utd', I, 1, 'update-to-data ratio'),
    ('policy_utd', I, 1, 'update-to-data ratio of the policy'),
    ('redq_m', I, 2, 'REDQ subset size'),
    ('value_embedding_hidden_size', ('list', I), [256, 128, 64], 'hidden widths of the critic context encoder'),
    ('value_embedding_activations', ('list', S), ['relu', 'linear', 'relu', 'tanh'], 'activations of the critic context encoder'),
    ('value_embedding_layer_type', ('list', S), ['fc', 'gru', 'fc', 'fc'], 'layer ids of the critic context encoder'),
    ('value_embedding_dim', str_or_int, 16, 'critic context embedding width'),
    ('policy_hidden_size', ('list', I), [256, 128], 'hidden widths of the policy MLP'),
    ('policy_activations', ('list', S), ['relu', 'relu', 'linear'], 'activations of the policy MLP'),
    ('policy_layer_type', ('list', S), ['fc', 'fc', 'fc'], 'layer ids of the policy MLP'),
    ('policy_embedding_hidden_size', ('list', I), [256, 128, 64], 'hidden widths of the policy context encoder'),
    ('policy_embedding_activations', ('list', S), ['relu', 'linear', 'relu', 'tanh'], 'activations of the policy context encoder'),
    ('policy_embedding_layer_type', ('list', S), ['fc', 'gru', 'fc', 'fc'], 'layer ids of the policy context encoder'),
    ('policy_embedding_dim', str_or_int, 16, 'policy context embedding width'),
    ('test_nprocess', I, 5, 'evaluation worker processes'),
    ('test_nrollout', I, 2, 'rollouts per evaluation worker'),
    ('total_iteration', I, 5000, 'training iterations'),
    ('gamma', F, 0.99, 'discount factor'),
    ('information', S, 'None', 'free-form run tag'),
    ('rnn_sample_max_batch_size', I, 300000, 'cap on transitions sampled per batch'),
    ('max_buffer_traj_num', I, 10000, 'replay capacity in trajectories'),
    ('max_buffer_transition_num', I, int(1e6), 'replay capacity in transitions'),
    ('sac_tau', F, 0.995, 'target-network retention factor'),
    ('sac_alpha', F, 0.2, 'initial / fixed entropy coefficient'),
    ('target_entropy_ratio', F, 1.5, 'target entropy = -act_dim * ratio'),
    ('rnn_fix_length', I, 0, 'fixed RNN memory length (0 = whole trajectory)'),
    ('rnn_slice_length', I, 0, 'slice length of the slice trainer'),
    ('step_per_iteration', I, 1000, 'environment steps per iteration'),
    ('random_num', I, 20000, 'uniformly random warm-up steps'),
    ('start_train_num', I, 1000, 'environment steps before the first update'),
    ('update_interval', I, 1, 'environment steps between updates'),
    ('sac_batch_size', I, 1024, 'valid transitions per sampled batch'),
    ('base_algorithm', S, 'sac', 'sac or td3'),
    ('sample_std', F, 0.1, 'TD3 exploration noise'),
    ('target_action_noise_std', F, 0.04, 'TD3 target smoothing noise'),
    ('target_action_noise_clip', F, 0.12, 'TD3 target smoothing clip'),
]


class Parameter(ParameterTemplate):
    def __init__(self, config_path=None, debug=False):
        super().__init__(config_path, debug)

    def parse(self):
        parser = argparse.ArgumentParser(description=smart_logger.experiment_config.EXPERIMENT_TARGET)
        for name, kind, default, help_ in FLAGS:
            setattr(self, name, default)
            if kind == 'flag':
                parser.add_argument(f'--{name}', action='store_true', help=help_)
            elif isinstance(kind, tuple):
                parser.add_argument(f'--{name}', nargs='+', type=kind[1], default=default, help=help_)
            else:
                parser.add_argument(f'--{name}', type=kind, default=default, help=help_)
        args, _unknown = parser.parse_known_args()
        return args
