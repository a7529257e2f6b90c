"""`Linear`: torch.nn.Linear's parameters, initialisation and state-dict keys (checkpoints of the reference load unchanged), with the product
itself on the hand-written GEMMs.  Every fully connected layer of the package is built from this class, so a direct module call -
`self.out_proj(x)` in a layer file, an encoder outside the fused input node - never reaches the vendor GEMM library: `ops.linear` is the
`LinearAct` node (`resel_gemm_f32x` forward, input gradient and weight gradient) for whole trajectories and single rollout tokens alike.
Reference: the `fc` layers of models/rnn_base.py:131-136 and the projections inside the layer files are plain torch.nn.Linear there."""
import torch

from ..hip import ops


class Linear(torch.nn.Linear):
    def forward(self, x: torch.Tensor) -> torch.Tensor:
        return ops.linear(x, self.weight, self.bias)
