"""Library-GEMM solution selection for the update's projection / critic GEMMs.

The fp32 GEMMs of the update are plain library calls (rocBLAS / hipBLASLt through torch.mm).  The default heuristics of
both libraries pick poor solutions for the weight-gradient shapes of this path - [M, 66752] x [66752, N] with M, N <= 1024,
i.e. a tiny output and a huge reduction - leaving them 3-20x off the fp32 MFMA / HBM bound (measured: 237 us -> 12 us for
the [128, 66752] x [66752, 17] first-layer gradient, 620 us -> 248 us for the [1024, 66752] x [66752, 256] in_proj
gradient).  `gemm_tuning/gfx950.csv` holds the per-shape winners found on an MI355X by PyTorch's TunableOp
(`tools/tune_gemms.py` regenerates it); this module loads them with tuning switched OFF, so a run never spends time
searching and unseen shapes fall back to the library default.  All candidates are fp32-in / fp32-accumulate solutions
of the same two libraries - no precision change.  RESEL_GEMM_SELECT=0 disables the table.

The table is keyed by exact GEMM shapes, i.e. by rows x row length of the training batch.  A run with another horizon or
batch size misses it (measured: the untuned `smamba_b1_c8_s64_ff` configuration ran 20 % slower than after tuning).
RESEL_GEMM_AUTOTUNE=1 keeps the table AND lets TunableOp search every shape it has not seen, once, on first use (seconds per
shape during the first updates); the winners are written to RESEL_GEMM_AUTOTUNE_FILE (default ./resel_gemm_tuned.csv) at
exit and can be passed back as RESEL_GEMM_TABLE or merged with `tools/tune_gemms.py merge`.
"""
import os
import tempfile

import torch

TABLE = os.environ.get('RESEL_GEMM_TABLE') or os.path.join(os.path.dirname(os.path.abspath(__file__)), 'gemm_tuning', 'gfx950.csv')
_state = {'loaded': None}


def _read_with_current_validators(tunable, path):
    """The table is rejected when any validator string differs (e.g. a point release of the same library build).  Solution
    names are stable across such releases, so retry once with this process' validator lines; entries the libraries no
    longer know are ignored by TunableOp.  Only when the architecture is still gfx950."""
    try:
        cur = dict(tunable.get_validators())
        old = dict(ln.strip().split(',')[1:3] for ln in open(path) if ln.startswith('Validator'))
        if 'gfx950' not in cur.get('GCN_ARCH_NAME', '') or cur == old:
            return False
        tmp = os.path.join(tempfile.gettempdir(), f'resel_gemm_table_{os.getpid()}.csv')
        with open(tmp, 'w') as fh:
            for k, v in cur.items():
                fh.write(f'Validator,{k},{v}\n')
            fh.writelines(ln for ln in open(path) if not ln.startswith('Validator'))
        return bool(tunable.read_file(tmp))
    except Exception:
        return False


def enable_tuned_gemms(path=TABLE):
    """Idempotent; returns True when the table was accepted (validators - torch / rocBLAS / hipBLASLt versions and the
    gfx950 arch string - must match the running stack, otherwise the libraries' defaults stay in force)."""
    if _state['loaded'] is not None:
        return _state['loaded']
    ok = False
    if os.environ.get('RESEL_GEMM_SELECT', '1') != '0' and torch.cuda.is_available() and os.path.exists(path) \
            and os.environ.get('PYTORCH_TUNABLEOP_TUNING', '0') != '1':
        import torch.cuda.tunable as tunable
        tunable.enable(True)
        tunable.tuning_enable(False)
        # results are written back at exit to get_filename(): point that away from the tracked table
        tunable.set_filename(os.path.join(tempfile.gettempdir(), f'resel_gemm_select_{os.getpid()}.csv'))
        try:
            torch._C._cuda_tunableop_write_file_on_exit(False)
        except AttributeError:
            pass
        ok = bool(tunable.read_file(path))
        if not ok:
            ok = _read_with_current_validators(tunable, path)
        if os.environ.get('RESEL_GEMM_AUTOTUNE', '0') == '1':       # search unseen shapes on first use, keep the winners
            tunable.enable(True)
            tunable.tuning_enable(True)
            tunable.set_filename(os.environ.get('RESEL_GEMM_AUTOTUNE_FILE', os.path.join(os.getcwd(), 'resel_gemm_tuned.csv')))
            try:
                torch._C._cuda_tunableop_write_file_on_exit(True)
            except AttributeError:
                pass
            ok = True
        if not ok:
            tunable.enable(False)
    _state['loaded'] = ok
    return ok
