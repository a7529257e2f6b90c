"""EnsembleLinear: E independent Linear layers evaluated as batched GEMMs (reference
offpolicy_rnn/models/ensemble_linear_model.py:8-69).  Parameter names / shapes (`weight` [E, in, out], `bias`
[E, 1, out]) and the shape polymorphism steered by `desire_ndim` follow the reference.

Every contraction is `ops.gemm_f32` (the hand-written GEMMs behind `resel_gemm_f32x`; there is no library GEMM in this file since round 6),
arranged so that no operand is ever copied:
  * shared input (every member sees the same rows): ONE GEMM  [M, in] x [in, E*out]  with bias / ELU in its epilogue; the result
    is handed on as an [E, M, out] *view* (strides (out, E*out, 1)) that the next layer's strided-batched GEMM consumes
    directly; the backward is again one GEMM each for dX (which also sums over the ensemble) and dW;
  * per-member input [E, M, in]: one strided-batched GEMM with the epilogue; dW = X^T G and dX = G W^T likewise, on views.
torch's generic einsum/bmm autograd materialised a transposed [E, in, M] copy of the activations per layer and step
(547 MB at config 2) plus separate bias-add kernels; these Functions remove both."""
import torch
import torch.nn as nn

from ..hip import ops


def _unit(t):
    """t with unit stride along its last axis (what `ops.gemm_f32` describes by row / member strides); a copy only for exotic views."""
    return t if t.stride(-1) == 1 else t.contiguous()


class _SharedInput(torch.autograd.Function):
    """x_part / col0: optional differentiable column block of x2 (x2[:, col0:col0 + k] holds the values of x_part): the
    backward then forms ONLY that block of dX - the actor step differentiates the critic's first layer with respect to the
    action encoding alone (128 of 384 input columns at config 2: a third of the dX GEMM)."""

    @staticmethod
    def forward(ctx, x2, weight, bias, act, x_part=None, col0=0, ax=None):
        """ax: magnitude handle of x2 (ops.amax_of) when the caller has one - saved tensors and reshaped views lose their tags."""
        E, n_in, n_out = weight.shape
        w_cat = weight.permute(1, 0, 2).reshape(n_in, E * n_out)                 # [in, E*out] (weights only: tiny copy)
        ops.LAST_AMAX = None
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        # bias + ELU in the GEMM's epilogue (837 us against the library's 995 at 66 752 x 384 -> 2048)
        y2 = ops.gemm_f32(x2, w_cat, True, False, None if bias is None else bias.reshape(E * n_out), act, amax_a=ax,
                          amax_b=ops.weight_amax(weight))              # w_cat holds the same values as the parameter
        ax = ax if ax is not None else ops.amax_of(x2)
        ctx.save_for_backward(x2, w_cat, y2 if act is not None else None)
        ctx.ax = ops.keep_handles(ax)[0]
        ctx.dims = (E, n_in, n_out, bias is not None, act)
        ctx.part = None if x_part is None else (col0, x_part.shape[-1], tuple(x_part.shape))
        return y2.view(-1, E, n_out).transpose(0, 1)                              # [E, M, out] view

    @staticmethod
    def backward(ctx, g):
        x2, w_cat, y2 = ctx.saved_tensors
        E, n_in, n_out, has_bias, act = ctx.dims
        g2 = g.transpose(0, 1).reshape(-1, E * n_out)                             # a view when g kept y's strides
        need_db = has_bias and ctx.needs_input_grad[2]
        if act is not None:
            g2, db = ops.bias_act_bwd(g2, y2, g2.shape[0], act, need_db)
        else:
            db = g2.sum(dim=0, keepdim=True) if need_db else None
        g2 = g2 if g2.stride(-1) == 1 else g2.contiguous()
        dx = None
        if ctx.needs_input_grad[0]:                                               # sums over the ensemble
            dx = ops.gemm_f32(g2, w_cat, True, True)
        dw = None
        if ctx.needs_input_grad[1]:
            dw = ops.gemm_f32(x2, g2, False, False, amax_a=ops.handle_alive(ctx.ax))
            dw = dw.view(n_in, E, n_out).permute(1, 0, 2)
        dpart = None
        if ctx.part is not None and ctx.needs_input_grad[4]:
            col0, k, shape = ctx.part
            wp = w_cat[col0:col0 + k]
            dpart = ops.gemm_f32(g2, wp, True, True).view(shape)
        return dx, dw, None if db is None else db.view(E, 1, n_out), None, dpart, None, None


def _member_wgrad(x3, gy, ax=None):
    """dW[e] = x3[e]^T gy[e] for the per-member layers: the hand-written K-split fp32 MFMA kernel (676 us against the tuned
    strided-batched library GEMM's 833 us at 8 x 66 752 x 256 x 256, `tools/bench_gemm_f32.py`) when the layout allows."""
    return ops.gemm_f32(_unit(x3), _unit(gy), False, False, amax_a=ax)


def _member_dgrad(gy, weight, like):
    """dx[e] = gy[e] W[e]^T.  When the layer's input was the [E, M, in] VIEW of a shared-input layer's [M, E*in] output
    (`like.stride(0) < like.stride(1)`), dx is written in that same memory layout, so that the producer's backward reads it as
    the [M, E*in] matrix it needs - no transposing copy.  649 us hand-written against 674 (8 x 66 752 x 256 x 256)."""
    E, M, _ = gy.shape
    n_in = weight.shape[1]
    if like.stride(0) < like.stride(1):
        dx = torch.empty(M, E, n_in, dtype=gy.dtype, device=gy.device).transpose(0, 1)
    else:
        dx = torch.empty(E, M, n_in, dtype=gy.dtype, device=gy.device)
    return ops.gemm_f32(_unit(gy), weight, True, True, out=dx)


class _PerMember(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x3, weight, bias, act, ax=None):
        E, M, _ = x3.shape
        ops.LAST_AMAX = None
        x3 = _unit(x3)
        # bias + ELU in the GEMM epilogue: 583 us against the library's 605 + a 140 us tail pass (8 x 66 752 x 256 x 256)
        y = ops.gemm_f32(x3, weight, True, False, bias, act, amax_a=ax)
        ax = ax if ax is not None else ops.amax_of(x3)
        ctx.save_for_backward(x3, weight, y if act is not None else None)
        ctx.ax = ops.keep_handles(ax)[0]
        ctx.cfg = (bias is not None, act)
        return y

    @staticmethod
    def backward(ctx, g):
        x3, weight, y = ctx.saved_tensors
        has_bias, act = ctx.cfg
        E, M, n_out = g.shape
        need_db = has_bias and ctx.needs_input_grad[2]
        if act is not None or (need_db and n_out % 4 == 0):
            gy, db = ops.bias_act_bwd(g.reshape(E * M, n_out), None if y is None else y.view(E * M, n_out), M, act, need_db)
            gy = gy.view(E, M, n_out)
            db = None if db is None else db.view(E, 1, n_out)
        else:
            gy, db = g, (g.sum(dim=1, keepdim=True) if need_db else None)
        dx = _member_dgrad(gy, weight, x3) if ctx.needs_input_grad[0] else None
        dw = _member_wgrad(x3, gy, ops.handle_alive(ctx.ax)) if ctx.needs_input_grad[1] else None
        return dx, dw, db, None, None


class _Head(torch.autograd.Function):
    """Per-member hidden layer + ELU + width-1 output layer as one node: q = elu(x W2 + b2) W3 + b3.  The hidden
    activation is produced in place on the GEMM output together with q; the backward reads it once to form the hidden
    layer's gradient, its bias gradient and the output layer's weight gradient (no [E, M, H] outer-product tensor, no
    k = M GEMV through the GEMM library - rocBLAS' best solution for that shape runs at 1.1 TB/s)."""

    @staticmethod
    def forward(ctx, x3, w2, b2, w3, b3, ax=None):
        E, M, _ = x3.shape
        H = w2.shape[2]
        x3 = _unit(x3)
        a = ops.gemm_f32(x3, w2, True, False, amax_a=ax)
        ax = ax if ax is not None else ops.amax_of(x3)
        ctx.ax = ops.keep_handles(ax)[0]
        w3v = w3.reshape(E, H)
        q = ops.ensemble_head_fwd_(a, b2.reshape(E, H), w3v, None if b3 is None else b3.reshape(E))
        ctx.save_for_backward(x3, w2, w3v, a)
        ctx.has_b3 = b3 is not None
        return q.view(E, M, 1)

    @staticmethod
    def backward(ctx, gq):
        x3, w2, w3v, a = ctx.saved_tensors
        E, M, H = a.shape
        gq2 = gq.reshape(E, M)
        gy, db2, dw3 = ops.ensemble_head_bwd(gq2, a, w3v)
        dx = _member_dgrad(gy, w2, x3) if ctx.needs_input_grad[0] else None
        dw2 = _member_wgrad(x3, gy, ops.handle_alive(ctx.ax))
        db3 = gq2.sum(dim=1).view(E, 1, 1) if ctx.has_b3 else None
        return dx, dw2, db2.view(E, 1, H), dw3.view(E, H, 1), db3, None


class _CriticMLP(torch.autograd.Function):
    """The whole efc-E critic head as ONE node: q = efc3(elu(efc2(elu(efc1(x))))) with a shared input x2 [M, in], H1 = efc1's width,
    H2 = efc2's, efc3 of width 1 (reference: three EnsembleLinear layers with their activation modules, rnn_base.py:461-469 over
    ensemble_linear_model.py:36-49).  One node instead of `_SharedInput` + `_Head` lets the kernels between the layers fold:
      forward   GEMM 1 (bias + ELU in the epilogue) -> a1 [M, E H1];  GEMM 2 `resel_gemm_f32_head`: a2 = elu(a1 W2 + b2) AND q = a2 . w3 + b3
                from the same epilogue (the separate head pass read and rewrote a2: 547 MB each way at config 2);
      backward  head backward (gy2, db2, dw3); dW2 = a1^T gy2;  `resel_gemm_f32_dact`: g1 = (gy2 W2^T) * elu'(a1) AND db1 = column sums of g1
                from the epilogue (the separate ELU-backward pass read g and a1 and wrote g1: 3 x 547 MB); dX = g1 W1^T (only the x_part
                block when the actor differentiates the action encoding alone), dW1 = x2^T g1."""

    @staticmethod
    def forward(ctx, x2, W1, b1, W2, b2, W3, b3, x_part, col0, ax):
        E, n_in, H1 = W1.shape
        H2 = W2.shape[2]
        M = x2.shape[0]
        w_cat = W1.permute(1, 0, 2).reshape(n_in, E * H1)                          # [in, E H1] (weights only: tiny copy)
        y2 = ops.gemm_f32(x2, w_cat, True, False, b1.reshape(E * H1), 'elu', amax_a=ax, amax_b=ops.weight_amax(W1))
        h_x = ax if ax is not None else ops.amax_of(x2)
        h_a1 = ops.amax_of(y2)
        a1 = y2.view(M, E, H1).transpose(0, 1)                                     # [E, M, H1] view of the shared layer's output
        w3v = W3.reshape(E, H2)
        a2, q = ops.gemm_f32_head(a1, W2, False, b2.reshape(E, H2), w3v, None if b3 is None else b3.reshape(E), amax_a=h_a1,
                                  amax_b=ops.weight_amax(W2))
        ctx.save_for_backward(x2, w_cat, y2, W2, w3v, a2)
        ctx.handles = ops.keep_handles(h_x, h_a1)
        ctx.dims = (E, n_in, H1, H2, M, b3 is not None)
        ctx.part = None if x_part is None else (col0, x_part.shape[-1], tuple(x_part.shape))
        return q.view(E, M, 1)

    @staticmethod
    def backward(ctx, gq):
        x2, w_cat, y2, W2, w3v, a2 = ctx.saved_tensors
        E, n_in, H1, H2, M, has_b3 = ctx.dims
        h_x, h_a1 = ops.live_handles(ctx.handles)
        need = ctx.needs_input_grad
        gq2 = gq.reshape(E, M)
        gy2, db2, dw3 = ops.ensemble_head_bwd(gq2, a2, w3v)                        # gy2 [E, M, H2] = gq w3 elu'(a2); tagged with its magnitude
        a1 = y2.view(M, E, H1).transpose(0, 1)
        dw2 = _member_wgrad(a1, gy2, h_a1) if need[3] else None
        db3 = gq2.sum(dim=1).view(E, 1, 1) if (has_b3 and need[6]) else None
        g1 = torch.empty(M, E, H1, dtype=torch.float32, device=x2.device).transpose(0, 1)      # written in the shared layer's [M, E H1] layout
        _, db1 = ops.gemm_f32_dact(gy2, W2, True, a1, g1, need_dbias=bool(need[2]), amax_b=ops.weight_amax(W2))
        g2 = g1.transpose(0, 1).reshape(M, E * H1)                                  # a view
        ops.tag_amax(g2, ops.amax_of(g1))
        dx = ops.gemm_f32(g2, w_cat, True, True) if need[0] else None              # sums over the ensemble
        dw1 = None
        if need[1]:
            dw1 = ops.gemm_f32(x2, g2, False, False, amax_a=h_x).view(n_in, E, H1).permute(1, 0, 2)
        dpart = None
        if ctx.part is not None and need[7]:
            col0, k, shape = ctx.part
            dpart = ops.gemm_f32(g2, w_cat[col0:col0 + k], True, True).view(shape)
        return (dx, dw1, None if db1 is None else db1.view(E, 1, H1), dw2, db2.view(E, 1, H2) if need[4] else None,
                dw3.view(E, H2, 1) if need[5] else None, db3, dpart, None, None)


def critic_mlp_fusable(l0, act0, l1, act1, l2, act2, x) -> bool:
    """efc-E (shared input) ELU -> efc-E ELU -> efc-E(1) on a long GPU pass whose GEMMs the fused-epilogue forms take."""
    if not (isinstance(l0, EnsembleLinear) and isinstance(l1, EnsembleLinear) and isinstance(l2, EnsembleLinear)):
        return False
    elu = lambda a: isinstance(a, nn.ELU) and a.alpha == 1.0
    if not (elu(act0) and elu(act1) and isinstance(act2, nn.Identity) and l0.use_bias and l1.use_bias):
        return False
    E = l0.active_members()
    if l1.active_members() != E or l2.active_members() != E or x.dtype != torch.float32 or not x.is_cuda:
        return False
    n_in, H1 = l0.weight.shape[1:]
    H2 = l1.weight.shape[2]
    if l1.weight.shape[1] != H1 or l2.weight.shape[1] != H2 or l2.weight.shape[2] != 1 or x.shape[-1] != n_in:
        return False
    nd = x.dim()
    if nd >= 5 or (nd in (3, 4) and (l0.desire_ndim is None or l0.desire_ndim == nd) and x.shape[0] == E):
        return False                                                               # x already carries the ensemble axis: not the shared-input form
    if not all(l.desire_ndim is None or l.desire_ndim == nd + 1 for l in (l1, l2)):
        return False
    M = x.numel() // n_in
    x2 = x.reshape(-1, n_in)
    return (min(n_in, H1, H2) >= 32 and H1 % 32 == 0 and H2 % 32 == 0 and ops.gemm_f32_ok(M, x2) and ops.rows_aligned16(x2)
            and ops.gemm_fused_ok(5, M, H2, H1, l1.weight) and ops.gemm_fused_ok(4, M, H1, H2, l1.weight))


def critic_mlp(l0: 'EnsembleLinear', l1: 'EnsembleLinear', l2: 'EnsembleLinear', x: torch.Tensor, grad_part=None) -> torch.Tensor:
    W1, b1 = l0.active_params()
    W2, b2 = l1.active_params()
    W3, b3 = l2.active_params()
    E, n_in, _ = W1.shape
    lead = tuple(x.shape[:-1])
    ax = ops.amax_of(x)
    if grad_part is not None:
        q = _CriticMLP.apply(x.detach().reshape(-1, n_in), W1, b1, W2, b2, W3, b3, grad_part[0], grad_part[1], ax)
    else:
        q = _CriticMLP.apply(x.reshape(-1, n_in), W1, b1, W2, b2, W3, b3, None, 0, ax)
    return q.reshape((E,) + lead + (1,))


def ensemble_head(hidden: 'EnsembleLinear', out: 'EnsembleLinear', x: torch.Tensor) -> torch.Tensor:
    """`out(elu(hidden(x)))` for per-member x [E, ..., in] and out.weight [E, H, 1] (RNNBase.forward routes the last two
    layers of an efc-E critic head here)."""
    w2, b2 = hidden.active_params()
    w3, b3 = out.active_params()
    E, n_in, H = w2.shape
    lead = tuple(x.shape[1:-1])
    q = _Head.apply(x.reshape(E, -1, n_in), w2, b2, w3, b3, ops.amax_of(x))
    return q.reshape((E,) + lead + (1,))


def head_fusable(hidden, hidden_act, out, out_act, x) -> bool:
    return (isinstance(hidden, EnsembleLinear) and isinstance(out, EnsembleLinear) and isinstance(hidden_act, nn.ELU)
            and hidden_act.alpha == 1.0 and isinstance(out_act, nn.Identity) and hidden.use_bias
            and out.weight.shape[2] == 1 and out.weight.shape[1] == hidden.weight.shape[2] and hidden.weight.shape[2] % 4 == 0
            and x.dim() >= 3 and x.shape[0] == hidden.active_members() and x.dtype == torch.float32
            and (hidden.desire_ndim is None or hidden.desire_ndim == x.dim()))


class EnsembleLinear(nn.Module):
    def __init__(self, input_dim: int, output_dim: int, num_ensemble: int, bias: bool = True, desire_ndim: int = None):
        super().__init__()
        self.use_bias = bias
        self.desire_ndim = desire_ndim
        self.num_ensemble = num_ensemble
        self.weight = nn.Parameter(torch.zeros(num_ensemble, input_dim, output_dim))
        if bias:
            self.bias = nn.Parameter(torch.zeros(num_ensemble, 1, output_dim))
        nn.init.trunc_normal_(self.weight, std=1 / (2 * input_dim ** 0.5))
        self.member_index = None            # optional int64 device vector: evaluate only these ensemble members

    def active_members(self) -> int:
        return self.num_ensemble if self.member_index is None else int(self.member_index.numel())

    def active_params(self):
        """(weight, bias) of the members being evaluated: all of them, or the `member_index` subset (REDQ's target only
        needs its M sampled critics - reference sac_full_length_rnn_redq.py:16-34 evaluates all E and discards E - M)."""
        b = self.bias if self.use_bias else None
        if self.member_index is None:
            return self.weight, b
        return self.weight.index_select(0, self.member_index), None if b is None else b.index_select(0, self.member_index)

    def forward(self, x: torch.Tensor, act: str = None, grad_part=None) -> torch.Tensor:
        """act: None, or 'elu' to fuse the layer's activation module into the bias pass (RNNBase.forward does so).
        grad_part = (x_part, col0): x carries no graph of its own; only its column block x_part is differentiated."""
        W, b = self.active_params()
        E, n_in, n_out = W.shape
        nd = x.dim()
        shared = True                       # True: every member sees the same input (x has no ensemble axis)
        if nd == 3:
            shared = not ((self.desire_ndim is None or self.desire_ndim == 3) and x.shape[0] == E)
        elif nd == 4:
            shared = not ((self.desire_ndim is None or self.desire_ndim == 4) and x.shape[0] == E)
        elif nd == 5:
            shared = False
        ax = ops.amax_of(x)                  # magnitude handle of the input, if its producer left one (reshapes below drop the tag)
        if shared:
            lead = tuple(x.shape[:-1])
            if grad_part is not None:
                y = _SharedInput.apply(x.detach().reshape(-1, n_in), W, b, act, grad_part[0], grad_part[1], ax)
            else:
                y = _SharedInput.apply(x.reshape(-1, n_in), W, b, act, None, 0, ax)
        else:
            assert grad_part is None, 'grad_part is a shared-input feature'
            lead = tuple(x.shape[1:-1])
            y = _PerMember.apply(x.reshape(E, -1, n_in), W, b, act, ax)
        return ops.tag_amax(y.reshape((E,) + lead + (n_out,)), ops.LAST_AMAX, whole=True)      # a view: only the row axis is split
