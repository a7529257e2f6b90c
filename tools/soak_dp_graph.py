"""Soak of the data-parallel update graphs on ONE GPU: a one-rank RCCL group (RESEL_DP_FORCE_COLLECTIVES=1) issues every collective of the
data-parallel update; every update goes through GraphedUpdate.step() - three graphs per update, cut at the two gradient exchanges - with the
reference's published cadence (policy_update_per = 2).  python tools/soak_dp_graph.py [rnn] [updates]"""
import math, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [ROOT, os.path.join(ROOT, 'recurrent-offpolicy-rl_amd')]
os.environ.update(RESEL_DP_FORCE_COLLECTIVES='1', RANK='0', WORLD_SIZE='1', LOCAL_RANK='0', MASTER_ADDR='127.0.0.1')
os.environ.setdefault('MASTER_PORT', '29533')
import torch
import torch.distributed as dist
from offpolicy_rnn.parallel.data_parallel import init_from_env
init_from_env()
from bench import build_trainer
rnn = sys.argv[1] if len(sys.argv) > 1 else 'smamba_s32_c16_b2_nln'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 200
alg = build_trainer(rnn, 64, 1024)
alg.grad_sync.__init__()
alg.parameter.policy_update_per = 2
from offpolicy_rnn.algorithm.graphed_update import GraphedUpdate
gu = GraphedUpdate(alg, warmup=1)
for i in range(n):
    log = dict(gu.step())
    alg.grad_num += 1
    assert all(math.isfinite(float(v[0] if isinstance(v, tuple) else v)) for v in log.values()), (i, log)
torch.cuda.synchronize()
print('dp-graph soak ok:', {str(k[-1]): len(v['segs']) for k, v in gu.graphs.items()}, 'segments per recording (actor due: graphs);', alg.grad_sync.calls,
      'reserved GB %.1f' % (torch.cuda.memory_reserved() / 2 ** 30), flush=True)
dist.destroy_process_group()
