"""init_smart_logger(): same entry point as the reference (offpolicy_rnn/config/load_config.py:5-11)."""
import os

from .._compat import smart_logger


def init_smart_logger():
    here = os.path.dirname(os.path.abspath(__file__))
    base = os.path.dirname(os.path.dirname(here))
    rel = os.path.relpath(here, base)
    smart_logger.init_config(os.path.join(rel, 'common_config.yaml'), os.path.join(rel, 'experiment_config.yaml'), base)
