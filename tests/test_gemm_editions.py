"""The third edition of the split GEMM (`gemm_ws_kernel` in csrc/gemm_bf3.hip: producer / consumer waves, hand-placed tile loads).

Its producer loop leans on things the compiler must not undo - no scratch access among the hand-counted loads of the steady state (an
extra vector-memory wait there drains the pipeline), and above all no COPY of a register that a load is still writing (right answers on
small grids, garbage on large ones).  `test_producer_loop_isa` compiles the file to ISA and checks exactly that (no GPU needed), and the same for
the hand-counted Y loads of the dact epilogue's consumer waves (no compiler-emitted vector-memory instruction among them, no use of a Y register
before its wait: advisor r05);
`test_large_grids_against_mode6` runs every layout on grids that fill the chip, where the unit tests' sizes do not."""
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_producer_loop_isa(tmp_path):
    hipcc = '/opt/rocm/bin/hipcc'
    if not os.path.exists(hipcc):
        pytest.skip('no hipcc')
    src = os.path.join(ROOT, 'recurrent-offpolicy-rl_amd', 'csrc', 'gemm_bf3.hip')
    out = str(tmp_path / 'gemm_bf3.s')
    subprocess.run([hipcc, '--offload-arch=gfx950', '-O3', '-std=c++17', '-fPIC', '-fno-slp-vectorize', '-S', '--cuda-device-only', src, '-o', out],
                   check=True, capture_output=True, timeout=600)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'check_ws_isa.py'), out], capture_output=True, text=True)
    print(r.stdout)
    assert r.returncode == 0, r.stdout + r.stderr


@pytest.mark.gpu
@pytest.mark.parametrize('edition', ['3', '4'])
def test_large_grids_against_mode6(edition):
    """edition 4 = the experimental 256 x 256 block tile (`gemm_w8_kernel`: takes the tall A-[rows][K] shapes of the sweep, the rest falls
    through to edition 3); off by default, kept selectable for the A/B in profiles/r05_gemm.md."""
    if not torch.cuda.is_available():
        pytest.skip('needs a GPU')
    env = dict(os.environ, RESEL_GEMM_EDITION=edition)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'tools', 'gemm_sweep.py')], capture_output=True, text=True, env=env, timeout=900)
    print(r.stdout[-2000:])
    assert r.returncode == 0 and 'cases ok' in r.stdout and 'BAD' not in r.stdout, r.stdout[-2000:] + r.stderr[-2000:]
