"""Input encoders shared by actor and critic: `[state, last_state, last_action, reward]` filtered by flags, each
through its own Linear(., 128) when `separate_encoder` (reference contextual_sac_value.py:27-48,90-99)."""
import torch

from ..hip import ops
from ..models.linear import Linear

BASIC_EMBEDDING_DIM = 128


def build_encoders(owner, state_dim, action_dim, reward_input, last_action_input, last_state_input, separate_encoder):
    owner.reward_input, owner.last_action_input, owner.last_state_input = reward_input, last_action_input, last_state_input
    owner.reward_dim = 1 if reward_input else 0
    owner.last_act_dim = action_dim if last_action_input else 0
    owner.last_obs_dim = state_dim if last_state_input else 0
    owner.separate_encoder = separate_encoder
    if separate_encoder:
        d = BASIC_EMBEDDING_DIM
        owner.state_encoder = Linear(state_dim, d)
        owner.last_act_encoder = Linear(owner.last_act_dim, d) if owner.last_act_dim else None
        owner.reward_encoder = Linear(owner.reward_dim, d) if owner.reward_dim else None
        owner.last_obs_encoder = Linear(owner.last_obs_dim, d) if owner.last_obs_dim else None
        return d * (1 + sum(e is not None for e in (owner.last_act_encoder, owner.last_obs_encoder, owner.reward_encoder)))
    ident = torch.nn.Identity()
    owner.state_encoder = owner.last_act_encoder = owner.reward_encoder = owner.last_obs_encoder = ident
    return state_dim + owner.reward_dim + owner.last_act_dim + owner.last_obs_dim


def register_encoders(owner):
    if not owner.separate_encoder:
        return
    owner.contextual_register_rnn_base_module(owner.state_encoder, 'state_encoder')
    for mod, name in ((owner.last_act_encoder, 'last_act_encoder'), (owner.last_obs_encoder, 'last_obs_encoder'),
                      (owner.reward_encoder, 'reward_encoder')):
        if mod is not None:
            owner.contextual_register_rnn_base_module(mod, name)


def _fusable(act_mod, x, mods) -> bool:
    """The activation module behind the encoders is a plain ELU and the pass is a long fp32 GPU pass of Linear encoders: it rides in
    the encoder GEMM's epilogue (and its derivative in the backward's bias pass) instead of two element-wise ATen passes."""
    return (isinstance(act_mod, torch.nn.ELU) and act_mod.alpha == 1.0 and x.is_cuda and x.dtype == torch.float32
            and x.numel() // x.shape[-1] >= ops.GEMM_F32_MIN_ROWS and all(isinstance(m, torch.nn.Linear) and m.bias is not None for m in mods))


def encode_concat(pairs, act_mod=None, dest=None) -> torch.Tensor:
    """cat([enc_i(x_i)], -1) for Linear encoders as ONE library GEMM: the (narrow) inputs are concatenated instead of the
    (wide) outputs and multiplied by the block-diagonal of the encoder weights with the biases fused (addmm epilogue) -
    per update this removes three skinny GEMMs, three bias-add passes and the 384-wide cat per call.  The zero blocks
    add exact zeros to every dot product; autograd splits the gradients back through block_diag / cat (long GPU passes: through
    `ops.place_blocks`, one launch per operand, gradients as views).
    act_mod: the activation module applied to the result (None: none) - fused into the GEMM when `_fusable`.
    dest: an `ops.ColDest` the fused GEMM writes its output to in place (a column block of the head's row buffer)."""
    mods = [m for m, _ in pairs]
    xs = [x for _, x in pairs]
    fuse = act_mod is not None and _fusable(act_mod, xs[0], mods)
    post = (lambda t: t) if (act_mod is None or fuse) else act_mod
    if not all(isinstance(m, torch.nn.Linear) and m.bias is not None for m in mods):
        return post(torch.cat([m(x) for m, x in pairs], dim=-1))
    # no activation behind the encoders (`linear`): the same node without an epilogue, so that `dest` is honoured there too
    plain = (act_mod is None or isinstance(act_mod, torch.nn.Identity)) and _fusable(torch.nn.ELU(), xs[0], mods)
    if len(pairs) == 1:                       # one encoder alone (the actor step's action encoding): still the hand-written GEMM
        if (fuse or plain) and mods[0].weight.shape[0] >= ops.GEMM_F32_MIN_DIM and mods[0].weight.shape[1] >= ops.GEMM_F32_MIN_K:
            return ops.linear_act(xs[0], mods[0].weight, mods[0].bias, 'elu' if fuse else None, dest=dest)
        return (act_mod if fuse else post)(ops.linear(xs[0], mods[0].weight, mods[0].bias))
    ks, ns = [m.weight.shape[1] for m in mods], [m.weight.shape[0] for m in mods]
    kp = sum(ks) + (-sum(ks)) % 4            # 17 + 17 + 6 + 1 = 41 input columns: three zero columns make the rows 16-byte multiples,
    if xs[0].is_cuda and xs[0].dtype == torch.float32 and xs[0].numel() // xs[0].shape[-1] >= ops.GEMM_F32_MIN_ROWS and len(mods) <= 8:
        # which the hand-written GEMM needs (35 us against the library's 97 at 66 752 tokens).  Long GPU passes: the three operands are
        # assembled by ONE launch each (`ops.place_blocks`) and their gradients come back as views
        c0 = [sum(ks[:i]) for i in range(len(ks))]
        r0 = [sum(ns[:i]) for i in range(len(ns))]
        w = ops.place_blocks(sum(ns), kp, list(zip(r0, c0)), *[m.weight for m in mods])
        b = ops.place_blocks(1, sum(ns), [(0, r) for r in r0], *[m.bias for m in mods]).view(-1)
        rows = xs[0].numel() // xs[0].shape[-1]
        x = ops.place_blocks(rows, kp, [(0, c) for c in c0], *xs).view(*xs[0].shape[:-1], kp)
    else:
        w = torch.block_diag(*[m.weight for m in mods])
        b = torch.cat([m.bias for m in mods])
        pad = (-w.shape[1]) % 4
        if pad and xs[0].is_cuda:
            xs = xs + [torch.zeros(*xs[0].shape[:-1], pad, dtype=xs[0].dtype, device=xs[0].device)]
            w = torch.nn.functional.pad(w, (0, pad))
        x = torch.cat(xs, dim=-1)
    x2 = x.reshape(-1, x.shape[-1])
    if (fuse or plain) and w.shape[0] >= ops.GEMM_F32_MIN_DIM and w.shape[1] >= ops.GEMM_F32_MIN_K:
        return ops.linear_act(x2, w, b, 'elu' if fuse else None, dest=dest).view(*x.shape[:-1], w.shape[0])
    y = ops.linear(x2, w, b)
    return (act_mod if fuse else post)(y.view(*x.shape[:-1], w.shape[0]))


def embedding_input(owner, state, lst_state, lst_action, reward) -> torch.Tensor:
    pairs = [(owner.state_encoder, state)]
    if owner.last_state_input:
        pairs.append((owner.last_obs_encoder, lst_state))
    if owner.last_action_input:
        pairs.append((owner.last_act_encoder, lst_action))
    if owner.reward_input:
        pairs.append((owner.reward_encoder, reward))
    return encode_concat(pairs)
