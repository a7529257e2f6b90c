"""Minimal stand-in for the external `smart_logger` package (not installed in this image; the reference imports it
at algorithm/sac.py:26-27 and parameter/ParameterSAC.py:1).  Only what the hot path touches: a callable Logger with
tabular no-ops and an output directory, the ParameterTemplate base class, and the two module-level helpers."""
import json
import os
import time


class ParameterTemplate:
    short_name = 'resel'

    def __init__(self, config_path=None, debug=False, silence=False):
        self._config_path = config_path
        args = self.parse()
        if args is not None:
            for k, v in vars(args).items():
                setattr(self, k, v)

    def parse(self):
        return None

    def set_config_path(self, path):
        self._config_path = path

    def save_config(self):
        if not self._config_path:
            return
        os.makedirs(self._config_path, exist_ok=True)
        pub = {k: v for k, v in vars(self).items() if not k.startswith('_') and isinstance(v, (int, float, str, bool, list, type(None)))}
        with open(os.path.join(self._config_path, 'parameter.json'), 'w') as f:
            json.dump(pub, f, indent=1, sort_keys=True)

    def __str__(self):
        return '\n'.join(f'{k}: {v}' for k, v in sorted(vars(self).items()) if not k.startswith('_'))


class Logger:
    def __init__(self, log_name=None, log_to_file=False, **kw):
        root = os.environ.get('RESEL_LOG_DIR', os.path.join('/tmp', 'resel_logs'))
        self.output_dir = os.path.join(root, f'{log_name or "run"}-{os.getpid()}')
        self._tab = {}
        self.quiet = os.environ.get('RESEL_QUIET', '1') == '1'

    def __call__(self, *msg):
        if not self.quiet:
            print(time.strftime('[%H:%M:%S]'), *msg, flush=True)

    def add_tabular_data(self, tb_prefix=None, **kw):
        for k, v in kw.items():
            self._tab.setdefault(f'{tb_prefix}/{k}' if tb_prefix else k, []).append(v)

    def log_tabular(self, key, val, tb_prefix=None):
        self._tab[f'{tb_prefix}/{key}' if tb_prefix else key] = [val]

    def dump_tabular(self):
        if not self.quiet:
            for k, v in sorted(self._tab.items()):
                try:
                    print(f'{k:48s} {sum(map(float, v)) / max(len(v), 1):.6g}')
                except (TypeError, ValueError):
                    pass
        self._tab = {}

    def sync_log_to_remote(self, *a, **k):
        pass


class _ExperimentConfig:
    EXPERIMENT_TARGET = 'RESeL on MI355X'


experiment_config = _ExperimentConfig()
_CUSTOM = {'MAX_TRAJ_STEP': 1000}


def init_config(*a, **k):
    pass


def get_customized_value(name):
    return _CUSTOM[name]
