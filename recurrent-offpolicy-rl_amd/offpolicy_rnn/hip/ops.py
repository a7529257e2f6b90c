"""torch.autograd wrappers over the C-ABI kernels (include/resel_hip.h).

Every function here requires CUDA (ROCm) tensors and the in-tree `libresel_hip.so`; there is deliberately NO
CPU / eager fallback - a missing library or a CPU tensor raises.  PyTorch only provides device memory, the
current stream and autograd bookkeeping.

Layout: activations are token-major `[B, L, C]` (channel stride 1).  Column slices of a wider row-major
tensor (e.g. the `x` / `z` halves of `in_proj`'s output) are passed by stride, never copied.
"""
import ctypes
import os
from typing import Optional

import torch

from ._lib import check, lib


def _need_cuda(name, *ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError(f'RESeL-HIP: {name} needs CUDA tensors (got a {t.device} tensor); the HIP kernels are the '
                               f'only implementation of this op - there is no CPU fallback.')


def _p(t: Optional[torch.Tensor]):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


if hasattr(torch._C, '_cuda_getCurrentRawStream') and hasattr(torch._C, '_cuda_getDevice'):
    def _stream():
        """Raw handle of the current HIP stream of the current device (the two C calls are what
        `torch.cuda.current_stream().cuda_stream` wraps in a Stream object: 11 us per call there, ~1700 calls per update)."""
        return ctypes.c_void_p(torch._C._cuda_getCurrentRawStream(torch._C._cuda_getDevice()))
else:                                                   # another torch build: the public (slower) spelling
    def _stream():
        return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _tok_major(t: torch.Tensor) -> torch.Tensor:
    """Return a [B, L, C] view/copy with channel stride 1, batch stride L*ld, ld % 4 == 0 and a 16-byte aligned base."""
    if t.dtype != torch.float32:
        t = t.float()
    B, L, C = t.shape
    ok = (C == 1 or t.stride(2) == 1) and t.stride(1) % 4 == 0 and (B == 1 or t.stride(0) == L * t.stride(1)) \
        and t.data_ptr() % 16 == 0 and t.stride(1) >= C
    return t if ok else t.contiguous()


def _flags(t: Optional[torch.Tensor], B: int, L: int) -> Optional[torch.Tensor]:
    """[B, L] / [B, L, 1] float flags -> dense fp32 [B*L]."""
    if t is None:
        return None
    return t.reshape(B, L).to(torch.float32).contiguous()


def _ws(nbytes: int, device) -> torch.Tensor:
    return torch.empty(max(int(nbytes), 16), dtype=torch.uint8, device=device)


PROF_SLOTS = ('sscan_fwd_kernel', 'sscan_bwd_kernel', 'attn_fwd_kernel', 'attn_dq_kernel', 'attn_dkv_kernel', 'linrec_real_fwd_kernel',
              'linrec_real_bwd_kernel', 'linrec_complex_fwd_kernel', 'linrec_complex_bwd_kernel', 'gru_fwd_kernel', 'gru_bwd_kernel',
              'conv_fwd_kernel', 'conv_bwd_kernel', 'gemm_f32_kernel', 'sscan_fwd_local_kernel',
              'sscan_bwd_local_kernel')                                         # RESEL_PROF_* of include/resel_hip.h, in id order
GEMM_FLOPS = [0.0]            # 2 M N K of every resel_gemm_f32 call since the caller last reset it (bench.py's GEMM line)


def profile_enable(on: bool):
    """Bind a (start, stop) HIP event pair to every sequence-layer kernel dispatch (bench.py's roofline measurement)."""
    check(lib().resel_profile_enable(int(bool(on))), 'profile_enable')


def profile_collect():
    """-> {kernel: (launches, avg_us)} for the dispatches recorded since the last call."""
    out = {}
    for kid, name in enumerate(PROF_SLOTS):
        tot, n = ctypes.c_double(0.0), ctypes.c_int(0)
        check(lib().resel_profile_collect(kid, ctypes.byref(tot), ctypes.byref(n)), 'profile_collect')
        if n.value:
            out[name] = (n.value, tot.value / n.value)
    return out


# ---------------------------------------------------------------------------------------------- selective scan
# 0: the library cuts small batches into time segments scanned in parallel (resel_hip.h); 1: never; k > 1: k segments (tests)
SSCAN_TIME_SEGMENTS = int(os.environ.get('RESEL_SSCAN_TIME_SEGMENTS', '0'))


class SelectiveScanFn(torch.autograd.Function):
    """Token-major selective scan with start resets.  Interface counterpart of the reference's
    `SelectiveScanFn` (mamba_ssm/ops/selective_scan_interface_new.py:19-84)."""

    @staticmethod
    def forward(ctx, u, delta, A, Bm, Cm, D, z, delta_bias, start, delta_softplus, return_last_state):
        _need_cuda('selective_scan', u, delta, A, Bm, Cm)
        u, delta, Bm, Cm = _tok_major(u), _tok_major(delta), _tok_major(Bm), _tok_major(Cm)
        z = None if z is None else _tok_major(z)
        A = A.float().contiguous()
        D = None if D is None else D.float().contiguous()
        delta_bias = None if delta_bias is None else delta_bias.float().contiguous()
        Bsz, L, Di = u.shape
        N = A.shape[1]
        start = _flags(start, Bsz, L)
        out = torch.empty(Bsz, L, Di, dtype=torch.float32, device=u.device)
        need_grad = any(ctx.needs_input_grad)
        ck = None
        if need_grad:
            ck = _ws(lib().resel_selective_scan_ckpt_bytes(Bsz, L, Di, N), u.device)
        last = torch.empty(Bsz, Di, N, dtype=torch.float32, device=u.device) if return_last_state else None
        nb = lib().resel_selective_scan_fwd_workspace_bytes(Bsz, L, Di, N, SSCAN_TIME_SEGMENTS)
        check(lib().resel_selective_scan_fwd(
            _p(u), u.stride(1), _p(delta), delta.stride(1), _p(z), 0 if z is None else z.stride(1), _p(A),
            _p(Bm), Bm.stride(1), _p(Cm), Cm.stride(1), _p(D), _p(delta_bias), _p(start),
            _p(out), out.stride(1), _p(ck), _p(last), _p(_ws(nb, u.device) if nb else None), Bsz, L, Di, N, int(bool(delta_softplus)),
            SSCAN_TIME_SEGMENTS, None, 0, _stream()), 'selective_scan_fwd')
        ctx.save_for_backward(u, delta, A, Bm, Cm, D, z, delta_bias, start, ck)
        ctx.softplus = bool(delta_softplus)
        if return_last_state:
            ctx.mark_non_differentiable(last)
            return out, last
        return out

    @staticmethod
    def backward(ctx, dout, *_):
        u, delta, A, Bm, Cm, D, z, delta_bias, start, ck = ctx.saved_tensors
        Bsz, L, Di = u.shape
        N = A.shape[1]
        dout = _tok_major(dout)
        dev = u.device
        du = torch.empty(Bsz, L, Di, dtype=torch.float32, device=dev)
        ddelta = torch.empty_like(du)
        dz = torch.empty_like(du) if z is not None else None
        dB = torch.empty(Bsz, L, N, dtype=torch.float32, device=dev)
        dC = torch.empty_like(dB)
        dA = torch.empty(Di, N, dtype=torch.float32, device=dev)
        dD = torch.empty(Di, dtype=torch.float32, device=dev) if D is not None else None
        dbias = torch.empty(Di, dtype=torch.float32, device=dev) if delta_bias is not None else None
        ws = _ws(lib().resel_selective_scan_bwd_workspace_bytes(Bsz, L, Di, N, SSCAN_TIME_SEGMENTS), dev)
        check(lib().resel_selective_scan_bwd(
            _p(u), u.stride(1), _p(delta), delta.stride(1), _p(z), 0 if z is None else z.stride(1), _p(A),
            _p(Bm), Bm.stride(1), _p(Cm), Cm.stride(1), _p(D), _p(delta_bias), _p(start),
            _p(dout), dout.stride(1), _p(ck),
            _p(du), du.stride(1), _p(ddelta), ddelta.stride(1), _p(dz), 0 if dz is None else dz.stride(1),
            _p(dB), dB.stride(1), _p(dC), dC.stride(1), _p(dA), _p(dD), _p(dbias), _p(ws),
            Bsz, L, Di, N, int(ctx.softplus), SSCAN_TIME_SEGMENTS, None, None, 0, _stream()), 'selective_scan_bwd')
        return du, ddelta, dA, dB, dC, dD, dz, dbias, None, None, None


def selective_scan_tm(u, delta, A, Bm, Cm, D=None, z=None, delta_bias=None, start=None, delta_softplus=True,
                      return_last_state=False):
    """Token-major entry: u, delta, z [B, L, Di]; Bm, Cm [B, L, N]; start [B, L] or [B, L, 1]."""
    return SelectiveScanFn.apply(u, delta, A, Bm, Cm, D, z, delta_bias, start, delta_softplus, return_last_state)


def selective_scan_fn(u, delta, A, B, C, start, D=None, z=None, delta_bias=None, delta_softplus=False,
                      return_last_state=False):
    """Reference signature (selective_scan_interface_new.py:87-93): u, delta, z, start (B, D, L); B, C (B, N, L).
    Channel-major VIEWS of token-major storage are accepted without copies; anything else is re-laid-out."""
    tm = lambda t: None if t is None else t.transpose(1, 2)
    st = None if start is None else start[:, 0, :]
    res = selective_scan_tm(tm(u), tm(delta), A, tm(B), tm(C), D, tm(z), delta_bias, st, delta_softplus, return_last_state)
    if return_last_state:
        return res[0].transpose(1, 2), res[1]
    return res.transpose(1, 2)


# ---------------------------------------------------------------------------------------------- fused Mamba mixer


class MambaInnerFn(torch.autograd.Function):
    """in_proj -> masked causal conv + SiLU -> x_proj -> dt_proj -> selective scan (gate, skip, resets) -> out_proj as ONE
    autograd node (interface counterpart of the reference's `MambaInnerFn`, selective_scan_interface_new.py:169-335, which
    only exists for d_conv <= 4).  Everything is token-major; the backward writes every gradient straight into its slot of
    two buffers - dxz [M, 2Di] (conv backward fills the x half, scan backward the z half) and dx_dbl [M, R+2N] (scan
    backward fills the B / C columns) - through the kernels' token strides, so autograd's slice-backward zero-fill + copy +
    add passes (8 x 273 MB per update at config 2) disappear.  GEMMs: `mm_nt` / `mm_nn` / `wgrad` (hand-written for long passes, library otherwise)."""

    @staticmethod
    def forward(ctx, x, in_w, conv_w, conv_b, xproj_w, dt_w, dt_b, A_log, D, out_w, mask, start):
        _need_cuda('mamba_inner', x, in_w, conv_w, xproj_w, dt_w, A_log, out_w)
        Bsz, L, Dm = x.shape
        Di, N = A_log.shape
        R = dt_w.shape[1]
        K = conv_w.shape[-1]
        M = Bsz * L
        x2 = x.reshape(M, Dm)
        maskf, startf = _flags(mask, Bsz, L), _flags(start, Bsz, L)
        xz = mm_nt(x2, in_w)                                               # [M, 2Di] = (x | z)
        ctx.ax = amax_of(x2)                                               # magnitude handle of the block input (for d in_proj.weight)
        cw = conv_w.reshape(Di, K).contiguous()
        xc = torch.empty(M, Di, dtype=torch.float32, device=x.device)
        track = amax_tracking() and M * Di >= (1 << 20)                    # publish the magnitudes of the tensors the projections read
        h_xc, p_xc, e_xc = _slot_args(track, x.device)
        check(lib().resel_causal_conv1d_fwd(_p(xz), 2 * Di, _p(cw), _p(conv_b), _p(maskf), _p(xc), Di, Bsz, L, Di, K, 1, p_xc, e_xc, _stream()),
              'causal_conv1d_fwd')
        tag_amax(xc, h_xc)
        x_dbl = mm_nt(xc, xproj_w)                                         # [M, R + 2N] = (delta_r | B | C)
        fold = True    # (rounds 4-5: only where the hand-written GEMM took dt_proj; it takes every shape now)
        if fold and R < 32 <= R + 2 * N and os.environ.get('RESEL_DT_PAD', '1') != '0' and gemm_split() == 2:
            # dt_proj has K = dt_rank = 16: one PARTIAL K step, which only the fp32-MFMA first edition takes (58 us for a product whose floor
            # is its 137 MB of output: 28 us).  Read 32 columns of x_dbl instead - the 16 behind delta_r are B's, finite - against the weight
            # padded with 16 zero columns (one launch): a whole K step, so the producer / consumer edition with its full-line stores runs it.
            w32 = place_blocks(Di, 32, [(0, 0)], dt_w.detach())
            dt = gemm_f32(x_dbl[:, :32], w32, True, True, dt_b, GEMM_SOFTPLUS, amax_a=amax_of(x_dbl), amax_b=weight_amax(dt_w))
        elif fold:     # delta = softplus(dt_proj(.) + bias) leaves the GEMM epilogue: the scan kernels spend no vector issue on it
            dt = gemm_f32(x_dbl[:, :R], dt_w, True, True, dt_b, GEMM_SOFTPLUS)
        else:
            dt = mm_nt(x_dbl[:, :R], dt_w)                                  # [M, Di]; bias enters the scan as delta_bias
        A = A_log.float().contiguous()                                     # the kernels form A = -exp(A_log) themselves (delta_softplus + 4)
        need_grad = any(ctx.needs_input_grad)
        ck = _ws(lib().resel_selective_scan_ckpt_bytes(Bsz, L, Di, N), x.device) if need_grad else None
        y = torch.empty(M, Di, dtype=torch.float32, device=x.device)
        zptr = ctypes.c_void_p(xz.data_ptr() + 4 * Di)
        bptr, cptr = ctypes.c_void_p(x_dbl.data_ptr() + 4 * R), ctypes.c_void_p(x_dbl.data_ptr() + 4 * (R + N))
        nb = lib().resel_selective_scan_fwd_workspace_bytes(Bsz, L, Di, N, SSCAN_TIME_SEGMENTS)
        h_y, p_y, e_y = _slot_args(track, x.device)
        check(lib().resel_selective_scan_fwd(_p(xc), Di, _p(dt), Di, zptr, 2 * Di, _p(A), bptr, R + 2 * N, cptr, R + 2 * N,
                                             _p(D), None if fold else _p(dt_b), _p(startf), _p(y), Di, _p(ck), None, _p(_ws(nb, x.device) if nb else None),
                                             Bsz, L, Di, N, 4 + (2 if fold else 1), SSCAN_TIME_SEGMENTS, p_y, e_y, _stream()), 'selective_scan_fwd')
        tag_amax(y, h_y)
        out = mm_nt(y, out_w)
        ctx.handles = keep_handles(ctx.ax, h_xc, h_y)                      # saved tensors come back untagged
        ctx.save_for_backward(x2, in_w, cw, conv_b, xproj_w, dt_w, dt_b, A, D, out_w, maskf, startf, xz, xc, x_dbl, dt, y, ck)
        ctx.dims = (Bsz, L, Dm, Di, N, R, K, conv_w.shape)
        ctx.fold = fold
        return out.view(Bsz, L, -1)

    @staticmethod
    def backward(ctx, dout):
        (x2, in_w, cw, conv_b, xproj_w, dt_w, dt_b, A, D, out_w, maskf, startf, xz, xc, x_dbl, dt, y, ck) = ctx.saved_tensors
        Bsz, L, Dm, Di, N, R, K, cw_shape = ctx.dims
        M = Bsz * L
        dev = x2.device
        do2 = dout.reshape(M, -1)
        if not do2.is_contiguous():
            do2 = do2.contiguous()
        ax, h_xc, h_y = live_handles(ctx.handles)
        d_out_w = wgrad(do2, y, amax_x=h_y)
        dy = mm_nn(do2, out_w)                                             # [M, Di]
        dxz = torch.empty(M, 2 * Di, dtype=torch.float32, device=dev)      # fully written: conv bwd -> [:, :Di], scan bwd -> [:, Di:]
        dx_dbl = torch.empty(M, R + 2 * N, dtype=torch.float32, device=dev)
        dxc = torch.empty(M, Di, dtype=torch.float32, device=dev)
        ddt = torch.empty(M, Di, dtype=torch.float32, device=dev)
        dA = torch.empty(Di, N, dtype=torch.float32, device=dev)
        dD = torch.empty(Di, dtype=torch.float32, device=dev)
        ddt_b = torch.empty(Di, dtype=torch.float32, device=dev)
        ws = _ws(lib().resel_selective_scan_bwd_workspace_bytes(Bsz, L, Di, N, SSCAN_TIME_SEGMENTS), dev)
        P = lambda t, off: ctypes.c_void_p(t.data_ptr() + 4 * off)
        track = amax_tracking() and M * Di >= (1 << 20)
        h_dxz, p_dxz, e_b = _slot_args(track, dev)                         # ONE handle for dxz: the scan fills its z half, the conv its x half
        h_ddt, p_ddt, _ = _slot_args(track, dev)                           # (published under the same epoch e_b)
        check(lib().resel_selective_scan_bwd(
            _p(xc), Di, _p(dt), Di, P(xz, Di), 2 * Di, _p(A), P(x_dbl, R), R + 2 * N, P(x_dbl, R + N), R + 2 * N,
            _p(D), None if ctx.fold else _p(dt_b), _p(startf), _p(dy), Di, _p(ck),
            _p(dxc), Di, _p(ddt), Di, P(dxz, Di), 2 * Di, P(dx_dbl, R), R + 2 * N, P(dx_dbl, R + N), R + 2 * N,
            _p(dA), _p(dD), _p(ddt_b), _p(ws), Bsz, L, Di, N, 4 + (2 if ctx.fold else 1), SSCAN_TIME_SEGMENTS, p_dxz, p_ddt, e_b, _stream()), 'selective_scan_bwd')
        tag_amax(ddt, h_ddt)
        # [Di, R] with a 66 752-long reduction: hand-written MFMA kernel (the library reaches 7 TFLOP/s on this shape)
        d_dt_w = atb(ddt, x_dbl[:, :R]) if R <= 32 and Di % 4 == 0 and ddt.stride(1) == 1 and ddt.stride(0) % 4 == 0 \
            else gemm_f32(ddt, x_dbl[:, :R], False, False)
        mm_nn(ddt, dt_w, out=dx_dbl[:, :R])                                  # straight into its column block of dx_dbl
        d_xproj_w = wgrad(dx_dbl, xc, amax_x=h_xc)
        # the conv output's gradient = the scan's du (in dxc) + the x_proj input gradient: handed to the conv backward as TWO tensors
        # (summed on load) - the accumulating GEMM epilogue cost 150-214 us per call against 67 for the plain product
        gx = mm_nn(dx_dbl, xproj_w)
        dcw = torch.empty(Di, K, dtype=torch.float32, device=dev)
        dcb = torch.empty(Di, dtype=torch.float32, device=dev) if conv_b is not None else None
        ws2 = _ws(lib().resel_causal_conv1d_bwd_workspace_bytes(Bsz, L, Di, K), dev)
        check(lib().resel_causal_conv1d_bwd2(_p(xz), 2 * Di, _p(cw), _p(conv_b), _p(maskf), _p(dxc), Di, _p(gx), gx.stride(0) if gx is not None else 0, _p(dxz), 2 * Di,
                                             _p(dcw), _p(dcb), _p(ws2), Bsz, L, Di, K, 1, p_dxz, e_b, _stream()), 'causal_conv1d_bwd')
        tag_amax(dxz, h_dxz)
        d_in_w = wgrad(dxz, x2, amax_x=ax)
        dx = mm_nn(dxz, in_w).view(Bsz, L, Dm) if ctx.needs_input_grad[0] else None
        return (dx, d_in_w, dcw.reshape(cw_shape), dcb, d_xproj_w, d_dt_w, ddt_b, dA, dD, d_out_w, None, None)      # dA: already dL/dA_log


def mamba_inner_fn(x, in_w, conv_w, conv_b, xproj_w, dt_w, dt_b, A_log, D, out_w, mask=None, start=None):
    """x [B, L, D] -> [B, L, D]: the whole Mamba mixer of the training path (no biases on in/out projections)."""
    return MambaInnerFn.apply(x.float().contiguous(), in_w, conv_w, conv_b, xproj_w, dt_w, dt_b, A_log, D, out_w, mask, start)


# ---------------------------------------------------------------------------------------------- causal conv1d
class CausalConv1dFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, weight, bias, mask, activation):
        _need_cuda('causal_conv1d', x, weight)
        x = _tok_major(x)
        Bsz, L, Di = x.shape
        w = weight.float().reshape(Di, -1).contiguous()
        K = w.shape[1]
        bias = None if bias is None else bias.float().contiguous()
        mask = _flags(mask, Bsz, L)
        y = torch.empty(Bsz, L, Di, dtype=torch.float32, device=x.device)
        check(lib().resel_causal_conv1d_fwd(_p(x), x.stride(1), _p(w), _p(bias), _p(mask), _p(y), y.stride(1),
                                            Bsz, L, Di, K, int(bool(activation)), None, 0, _stream()), 'causal_conv1d_fwd')
        ctx.save_for_backward(x, w, bias, mask)
        ctx.act = bool(activation)
        ctx.wshape = weight.shape
        return y

    @staticmethod
    def backward(ctx, dy):
        x, w, bias, mask = ctx.saved_tensors
        Bsz, L, Di = x.shape
        K = w.shape[1]
        dy = _tok_major(dy)
        dx = torch.empty(Bsz, L, Di, dtype=torch.float32, device=x.device)
        dw = torch.empty(Di, K, dtype=torch.float32, device=x.device)
        db = torch.empty(Di, dtype=torch.float32, device=x.device) if bias is not None else None
        ws = _ws(lib().resel_causal_conv1d_bwd_workspace_bytes(Bsz, L, Di, K), x.device)
        check(lib().resel_causal_conv1d_bwd(_p(x), x.stride(1), _p(w), _p(bias), _p(mask), _p(dy), dy.stride(1),
                                            _p(dx), dx.stride(1), _p(dw), _p(db), _p(ws), Bsz, L, Di, K, int(ctx.act),
                                            None, 0, _stream()), 'causal_conv1d_bwd')
        return dx, dw.reshape(ctx.wshape), db, None, None


def causal_conv1d_fn(x, weight, bias=None, mask=None, activation=True):
    """x [B, L, Di] token-major; weight [Di, K] or Conv1d's [Di, 1, K]; mask [B, L(,1)]: y = silu(conv(mask * x) + b)."""
    return CausalConv1dFn.apply(x, weight, bias, mask, activation)


# ---------------------------------------------------------------------------------------------- add + norm
class AddNormFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, residual, weight, bias, eps, rms, prenorm, act=None):
        _need_cuda('add_layernorm', x, weight)
        assert act in (None, 'elu')
        shape = x.shape
        C = shape[-1]
        x2 = x.float().reshape(-1, C).contiguous()
        r2 = None if residual is None else residual.float().reshape(-1, C).contiguous()
        M = x2.shape[0]
        w = weight.float().contiguous()
        b = None if bias is None else bias.float().contiguous()
        y = torch.empty_like(x2)
        need_res = residual is not None or prenorm
        res_out = torch.empty_like(x2) if need_res else None
        stats = torch.empty(M, 2, dtype=torch.float32, device=x.device)
        global LAST_AMAX
        slot, slot_p, epoch = _slot_args(amax_tracking() and M * C >= (1 << 20), x.device)
        check(lib().resel_add_layernorm_fwd(_p(x2), _p(r2), _p(w), _p(b), _p(y), _p(res_out), _p(stats), M, C, float(eps),
                                            int(bool(rms)), int(act == 'elu'), slot_p, epoch, _stream()), 'add_layernorm_fwd')
        LAST_AMAX = slot
        ctx.save_for_backward(res_out if res_out is not None else x2, w, stats, b if act else None)
        ctx.rms, ctx.has_bias, ctx.has_res, ctx.prenorm, ctx.shape = bool(rms), b is not None, residual is not None, prenorm, shape
        ctx.act = int(act == 'elu')
        if prenorm:
            return y.reshape(shape), res_out.reshape(shape)
        return y.reshape(shape)

    @staticmethod
    def backward(ctx, dy, dres=None):
        res, w, stats, b = ctx.saved_tensors
        M, C = res.shape
        dy2 = dy.float().reshape(M, C).contiguous()
        dr2 = None if (dres is None or not ctx.prenorm) else dres.float().reshape(M, C).contiguous()
        dx = torch.empty_like(res)
        dw = torch.empty(C, dtype=torch.float32, device=res.device)
        db = torch.empty(C, dtype=torch.float32, device=res.device) if ctx.has_bias else None
        ws = _ws(lib().resel_add_layernorm_bwd_workspace_bytes(M, C), res.device)
        slot, slot_p, epoch = _slot_args(amax_tracking() and M * C >= (1 << 20), res.device)
        check(lib().resel_add_layernorm_bwd(_p(dy2), _p(dr2), _p(res), _p(w), _p(b), _p(stats), _p(dx), _p(dw), _p(db), _p(ws),
                                            M, C, int(ctx.rms), int(ctx.has_bias), ctx.act, slot_p, epoch, _stream()), 'add_layernorm_bwd')
        dx = tag_amax(dx.reshape(ctx.shape), slot, whole=True)      # the gradient the block below multiplies with its out_proj weight
        return dx, (dx if ctx.has_res else None), dw, db, None, None, None, None


def _add_norm(x, residual, weight, bias, eps, rms, prenorm, act=None):
    global LAST_AMAX
    LAST_AMAX = None
    out = AddNormFn.apply(x, residual, weight, bias, eps, rms, prenorm, act)
    tag_amax(out[0] if prenorm else out, LAST_AMAX, whole=True)    # the normalised output feeds a projection: its magnitude came out of the same pass
    return out


def layer_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False, act=None):
    """Signature of the reference's fused add+LayerNorm (mamba_ssm/ops/triton/layernorm.py `layer_norm_fn`).  act='elu' (this library's
    addition): the plain ELU that follows the layer, applied where the normalised output is stored."""
    return _add_norm(x, residual, weight, bias, eps, False, prenorm, act)


def rms_norm_fn(x, weight, bias, residual=None, eps=1e-6, prenorm=False, residual_in_fp32=False):
    return _add_norm(x, residual, weight, bias, eps, True, prenorm)


# ---------------------------------------------------------------------------------------------- linear recurrences
def _token_rows(*ts):
    """[B, L, C] fp32 operands of a scan -> (operands usable in place, ld): their (b, t) axes collapse to token rows `ld` floats apart with unit
    column stride (dense tensors, or column blocks of one wider token-major matrix); anything else is copied to dense."""
    t0 = ts[0]
    ld = t0.stride(1)
    if all(t.dtype == torch.float32 and t.dim() == 3 and t.stride(2) == 1 and t.stride(1) == ld and t.stride(0) == t.shape[1] * ld and ld >= t.shape[2]
           for t in ts):
        return ts, ld
    return tuple(t.float().contiguous() for t in ts), t0.shape[2]


def _member_rows(u):
    """u [E, B, L, C]: the members as token-row operands (see `_token_rows`) - the [E, B, T, C] view of a shared-input EnsembleLinear's
    [M, E C] output qualifies as it is (ld = E C), so does a dense tensor (ld = C).  Returns (u or a dense copy, ld)."""
    E, Bsz, L, C = u.shape
    if not (u.dtype == torch.float32 and u.stride(3) == 1 and u.stride(1) == L * u.stride(2) and u.stride(2) >= C):
        u = u.float().contiguous()
    return u, u.stride(2)


def _like_members(u):
    """Uninitialised tensor with u's shape AND memory layout (gradients of the members go back in the layout the producer reads)."""
    return torch.empty_strided(u.shape, u.stride(), dtype=torch.float32, device=u.device)


def _real_fwd(v, f, ld, start, h0, fuse_act):
    Bsz, L, C = v.shape
    h = torch.empty(Bsz, L, C, dtype=torch.float32, device=v.device)
    slot, slot_p, epoch = _slot_args(amax_tracking() and h.numel() >= (1 << 20), v.device)
    check(lib().resel_linrec_real_fwd(_p(v), _p(f), ld, _p(start), _p(h0), _p(h), Bsz, L, C, int(bool(fuse_act)), slot_p, epoch, _stream()),
          'linrec_real_fwd')
    return h, slot


class GilrScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, v, f, start, h0, fuse_act):
        _need_cuda('linrec_real', v, f)
        (v, f), ld = _token_rows(v, f)
        Bsz, L, C = v.shape
        start = _flags(start, Bsz, L)
        h0 = None if h0 is None else h0.float().reshape(Bsz, C).contiguous()
        global LAST_AMAX
        h, LAST_AMAX = _real_fwd(v, f, ld, start, h0, fuse_act)
        ctx.save_for_backward(v, f, start, h0, h)
        ctx.act, ctx.ld = bool(fuse_act), ld
        return h

    @staticmethod
    def backward(ctx, dh):
        v, f, start, h0, h = ctx.saved_tensors
        Bsz, L, C = v.shape
        dh = dh.float().contiguous()
        dv = torch.empty(Bsz, L, C, dtype=torch.float32, device=v.device)
        df = torch.empty_like(dv)
        check(lib().resel_linrec_real_bwd(_p(v), _p(f), ctx.ld, _p(start), _p(h0), _p(h), _p(dh), _p(dv), _p(df), C, Bsz, L, C,
                                          int(ctx.act), None, 0, _stream()), 'linrec_real_bwd')
        return dv, df, None, None, None


class GilrMembersFn(torch.autograd.Function):
    """The gilr recurrence on u = (v | f) [2, B, T, C] as its producer left it (reference gilr.py:60-62 slices u[0], u[1]): the kernels
    read the two members through their row stride and write both gradients into ONE tensor of u's layout - no dense copies of the
    members going in (2 x 2 per call), no zero-filled [2, B, T, C] per member plus an add coming back (autograd's select_backward)."""

    @staticmethod
    def forward(ctx, u, start, h0, fuse_act):
        _need_cuda('linrec_real', u)
        assert u.dim() == 4 and u.shape[0] == 2
        u, ld = _member_rows(u)
        _, Bsz, L, C = u.shape
        start = _flags(start, Bsz, L)
        h0 = None if h0 is None else h0.float().reshape(Bsz, C).contiguous()
        global LAST_AMAX
        h, LAST_AMAX = _real_fwd(u[0], u[1], ld, start, h0, fuse_act)
        ctx.save_for_backward(u, start, h0, h)
        ctx.act, ctx.ld = bool(fuse_act), ld
        return h

    @staticmethod
    def backward(ctx, dh):
        u, start, h0, h = ctx.saved_tensors
        _, Bsz, L, C = u.shape
        dh = dh.float().contiguous()
        du = _like_members(u)
        slot, slot_p, epoch = _slot_args(amax_tracking() and du.numel() >= (1 << 20), u.device)
        check(lib().resel_linrec_real_bwd(_p(u[0]), _p(u[1]), ctx.ld, _p(start), _p(h0), _p(h), _p(dh), _p(du[0]), _p(du[1]), du.stride(2),
                                          Bsz, L, C, int(ctx.act), slot_p, epoch, _stream()), 'linrec_real_bwd')
        return tag_amax(du, slot), None, None, None


def gilr_scan(v, f, start=None, h0=None, fuse_act=True):
    """h_t = f'_t h_{t-1} + (1 - f'_t) v'_t with v' = tanh(v), f' = sigmoid(f) (1 - start) when fuse_act."""
    global LAST_AMAX
    LAST_AMAX = None
    return tag_amax(GilrScanFn.apply(v, f, start, h0, fuse_act), LAST_AMAX)


def gilr_scan_members(u, start=None, h0=None, fuse_act=True):
    """`gilr_scan(u[0], u[1], ...)` for u [2, B, T, C], reading the members in place (see `GilrMembersFn`)."""
    global LAST_AMAX
    LAST_AMAX = None
    return tag_amax(GilrMembersFn.apply(u, start, h0, fuse_act), LAST_AMAX)


def real_scan_tie_input_gate(v, f):
    """Reference name (gilr/scan_triton/real_rnn_tie_input_gate.py:170-214): post-activation v, f; zero initial state."""
    return GilrScanFn.apply(v, f, None, None, False)


def _complex_fwd(vr, vi, ld, lam_re, lam_im, gamma, start, h0r, h0i):
    """-> h2 [2, B, L, C] = (Re h | Im h) stacked (the layer's next product wants exactly that), its magnitude handle."""
    Bsz, L, C = vr.shape
    h2 = torch.empty(2, Bsz, L, C, dtype=torch.float32, device=vr.device)
    slot, slot_p, epoch = _slot_args(amax_tracking() and h2.numel() >= (1 << 20), vr.device)
    check(lib().resel_linrec_complex_fwd(_p(vr), _p(vi), ld, _p(lam_re), _p(lam_im), _p(gamma), _p(start), _p(h0r), _p(h0i),
                                         _p(h2[0]), _p(h2[1]), Bsz, L, C, slot_p, epoch, _stream()), 'linrec_complex_fwd')
    return h2, slot


def _complex_args(lam_re, lam_im, gamma, start, h0r, h0i, Bsz, L, C):
    lam_re, lam_im = lam_re.float().contiguous(), lam_im.float().contiguous()
    gamma = None if gamma is None else gamma.float().contiguous()
    start = _flags(start, Bsz, L)
    h0r = None if h0r is None else h0r.float().reshape(Bsz, C).contiguous()
    h0i = None if h0i is None else h0i.float().reshape(Bsz, C).contiguous()
    return lam_re, lam_im, gamma, start, h0r, h0i


def _complex_bwd(vr, vi, ld, lam_re, lam_im, gamma, start, h0r, h0i, h2, dh2, dvr, dvi, ld_du):
    Bsz, L, C = vr.shape
    dlr, dli = torch.empty_like(lam_re), torch.empty_like(lam_im)
    dg = torch.empty_like(lam_re) if gamma is not None else None
    ws = _ws(lib().resel_linrec_complex_bwd_workspace_bytes(Bsz, L, C), vr.device)
    check(lib().resel_linrec_complex_bwd(_p(vr), _p(vi), ld, _p(lam_re), _p(lam_im), _p(gamma), _p(start), _p(h0r), _p(h0i),
                                         _p(h2[0]), _p(h2[1]), _p(dh2[0]), _p(dh2[1]), _p(dvr), _p(dvi), ld_du, _p(dlr), _p(dli), _p(dg),
                                         _p(ws), Bsz, L, C, _stream()), 'linrec_complex_bwd')
    return dlr, dli, dg


class LruScanFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, vr, vi, lam_re, lam_im, gamma, start, h0r, h0i):
        _need_cuda('linrec_complex', vr, vi, lam_re, lam_im)
        (vr, vi), ld = _token_rows(vr, vi)
        Bsz, L, C = vr.shape
        lam_re, lam_im, gamma, start, h0r, h0i = _complex_args(lam_re, lam_im, gamma, start, h0r, h0i, Bsz, L, C)
        h2, _ = _complex_fwd(vr, vi, ld, lam_re, lam_im, gamma, start, h0r, h0i)
        ctx.save_for_backward(vr, vi, lam_re, lam_im, gamma, start, h0r, h0i, h2)
        ctx.ld = ld
        return h2[0], h2[1]

    @staticmethod
    def backward(ctx, dhr, dhi):
        vr, vi, lam_re, lam_im, gamma, start, h0r, h0i, h2 = ctx.saved_tensors
        Bsz, L, C = vr.shape
        dh2 = (dhr.float().contiguous(), dhi.float().contiguous())
        dvr = torch.empty(Bsz, L, C, dtype=torch.float32, device=vr.device)
        dvi = torch.empty_like(dvr)
        dlr, dli, dg = _complex_bwd(vr, vi, ctx.ld, lam_re, lam_im, gamma, start, h0r, h0i, h2, dh2, dvr, dvi, C)
        return dvr, dvi, dlr, dli, dg, None, None, None


class LruParamsFn(torch.autograd.Function):
    """params_log [3, C] = (nu_log | theta_log | gamma_log) -> lam3 [3, C] = (lam_re | lam_im | gamma), lambda = exp(-exp(nu_log)) e^{i exp(theta_log)},
    gamma = exp(gamma_log) (reference lru.py:104-110): one launch each way instead of ~25 element-wise launches on [C] tensors per layer call."""

    @staticmethod
    def forward(ctx, params_log):
        _need_cuda('lru_params', params_log)
        p = params_log.float().contiguous()
        out = torch.empty_like(p)
        check(lib().resel_lru_params_fwd(_p(p), _p(out), p.shape[1], _stream()), 'lru_params_fwd')
        ctx.save_for_backward(p)
        return out

    @staticmethod
    def backward(ctx, dout):
        (p,) = ctx.saved_tensors
        dout = dout.float().contiguous()
        dp = torch.empty_like(p)
        check(lib().resel_lru_params_bwd(_p(p), _p(dout), _p(dp), p.shape[1], _stream()), 'lru_params_bwd')
        return dp


def lru_params(params_log):
    return LruParamsFn.apply(params_log)


class LruMembersFn(torch.autograd.Function):
    """The lru recurrence on u [E >= 2, B, T, C] as its producer left it (reference lru.py:112-120: members 0 / 1 are Re / Im of the
    input, member 2 the skip term): reads the members through their row stride, returns h2 = (Re h | Im h) stacked [2, B, T, C] (what
    `middle_proj` multiplies - no `torch.stack` copy) and, for E = 3, member 2 as a pass-through output so that ALL of u's gradient
    comes back through this node as one tensor of u's layout (blocks 0 / 1 written by the kernel, block 2 one copy).
    lam3 [3, C] = (lam_re | lam_im | gamma) (`lru_params`); its gradient comes back as one [3, C] tensor too."""

    @staticmethod
    def forward(ctx, u, lam3, start, h0r, h0i):
        _need_cuda('linrec_complex', u, lam3)
        assert u.dim() == 4 and u.shape[0] in (2, 3) and lam3.shape == (3, u.shape[3])
        u, ld = _member_rows(u)
        E, Bsz, L, C = u.shape
        lam3 = lam3.float().contiguous()
        _, _, _, start, h0r, h0i = _complex_args(lam3[0], lam3[1], lam3[2], start, h0r, h0i, Bsz, L, C)
        global LAST_AMAX
        h2, LAST_AMAX = _complex_fwd(u[0], u[1], ld, lam3[0], lam3[1], lam3[2], start, h0r, h0i)
        ctx.save_for_backward(u, lam3, start, h0r, h0i, h2)
        ctx.ld = ld
        return (h2, u[2]) if E == 3 else (h2, None)

    @staticmethod
    def backward(ctx, dh2, du2):
        u, lam3, start, h0r, h0i, h2 = ctx.saved_tensors
        E, Bsz, L, C = u.shape
        dh2 = dh2.float().contiguous()
        du = _like_members(u)
        dlam3 = torch.empty_like(lam3)
        ws = _ws(lib().resel_linrec_complex_bwd_workspace_bytes(Bsz, L, C), u.device)
        check(lib().resel_linrec_complex_bwd(_p(u[0]), _p(u[1]), ctx.ld, _p(lam3[0]), _p(lam3[1]), _p(lam3[2]), _p(start), _p(h0r), _p(h0i),
                                             _p(h2[0]), _p(h2[1]), _p(dh2[0]), _p(dh2[1]), _p(du[0]), _p(du[1]), du.stride(2),
                                             _p(dlam3[0]), _p(dlam3[1]), _p(dlam3[2]), _p(ws), Bsz, L, C, _stream()), 'linrec_complex_bwd')
        if E == 3:
            if du2 is None:
                du[2].zero_()
            else:
                du[2].copy_(du2)
        return du, dlam3, None, None, None


def complex_scan(vr, vi, lam_re, lam_im, gamma=None, start=None, h0r=None, h0i=None):
    """h_t = lambda (1 - start_t) h_{t-1} + gamma (vr_t + i vi_t); lambda, gamma per channel [C]."""
    return LruScanFn.apply(vr, vi, lam_re, lam_im, gamma, start, h0r, h0i)


def complex_scan_members(u, lam3, start=None, h0r=None, h0i=None):
    """`complex_scan(u[0], u[1], lam3[0], lam3[1], lam3[2], ...)` for u [2 or 3, B, T, C] read in place and lam3 = `lru_params(params_log)`
    -> (h2 [2, B, T, C] = (Re h | Im h), u[2] or None); see `LruMembersFn`."""
    global LAST_AMAX
    LAST_AMAX = None
    h2, u2 = LruMembersFn.apply(u, lam3, start, h0r, h0i)
    return tag_amax(h2, LAST_AMAX), u2


class SubAddMembers(torch.autograd.Function):
    """m [2, ...] , r [...] -> m[0] - m[1] + r (lru.py:120 `mid[0] - mid[1] + u[2]`); the backward writes (g | -g) into ONE tensor of m's
    shape (autograd: a negation pass, two zero-filled [2, ...] tensors with one member copied in, and their sum)."""

    @staticmethod
    def forward(ctx, m, r):
        out = torch.sub(m[0], m[1])
        if r is not None:
            out.add_(r)
        ctx.has_r = r is not None
        return out

    @staticmethod
    def backward(ctx, g):
        dm = torch.empty((2,) + tuple(g.shape), dtype=g.dtype, device=g.device)
        dm[0].copy_(g)
        torch.neg(g, out=dm[1])
        return dm, (g if ctx.has_r else None)


# ---------------------------------------------------------------------------------------------- GRU
class GruSeqFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, gi, w_hh, b_hh, h0):
        _need_cuda('gru_seq', gi, w_hh, b_hh)
        gi = gi.float().contiguous()
        Bsz, L, H3 = gi.shape
        H = H3 // 3
        w_hh, b_hh = w_hh.float().contiguous(), b_hh.float().contiguous()
        h0 = None if h0 is None else h0.float().reshape(Bsz, H).contiguous()
        need_grad = any(ctx.needs_input_grad)
        h_all = torch.empty(Bsz, L, H, dtype=torch.float32, device=gi.device)
        gates = torch.empty(Bsz, L, 4 * H, dtype=torch.float32, device=gi.device) if need_grad else None
        ws = _ws(lib().resel_gru_workspace_bytes(Bsz, L, H), gi.device)
        check(lib().resel_gru_seq_fwd(_p(gi), _p(w_hh), _p(b_hh), _p(h0), _p(h_all), _p(gates), _p(ws), Bsz, L, H, _stream()),
              'gru_seq_fwd')
        ctx.save_for_backward(w_hh, h0, h_all, gates)
        return h_all

    @staticmethod
    def backward(ctx, dh_all):
        w_hh, h0, h_all, gates = ctx.saved_tensors
        Bsz, L, H = h_all.shape
        dh_all = dh_all.float().contiguous()
        dgi = torch.empty(Bsz, L, 3 * H, dtype=torch.float32, device=h_all.device)
        dgh = torch.empty_like(dgi)
        ws = _ws(lib().resel_gru_workspace_bytes(Bsz, L, H), h_all.device)
        check(lib().resel_gru_seq_bwd(_p(w_hh), _p(h0), _p(h_all), _p(gates), _p(dh_all), _p(dgi), _p(dgh), _p(ws),
                                      Bsz, L, H, _stream()), 'gru_seq_bwd')
        # dW_hh = dgh^T h_prev over the B*L rows (`wgrad`: the K-split hand-written GEMM on long passes) and db_hh = sum dgh
        h_prev = torch.cat((torch.zeros(Bsz, 1, H, device=h_all.device) if h0 is None else h0.unsqueeze(1), h_all[:, :-1]), dim=1)
        dw_hh = wgrad(dgh.reshape(-1, 3 * H), h_prev.reshape(-1, H))
        db_hh = dgh.sum(dim=(0, 1))
        return dgi, dw_hh, db_hh, None


def gru_seq(gi, w_hh, b_hh, h0=None):
    """gi = x W_ih^T + b_ih [B, L, 3H] -> all hidden states [B, L, H] (torch.nn.GRU gate order / formula)."""
    return GruSeqFn.apply(gi, w_hh, b_hh, h0)


# ---------------------------------------------------------------------------------------------- attention (cgpt)
class AttnVarlenFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, qkv, cu_seqlens, max_seqlen, slopes, scale, p_drop, seed, offset):
        _need_cuda('attn_varlen', qkv, cu_seqlens)
        assert qkv.dtype == torch.bfloat16 and qkv.dim() == 4 and qkv.shape[1] == 3
        qkv = qkv.contiguous()
        T, _, H, hd = qkv.shape
        cu = cu_seqlens.to(torch.int32).contiguous()
        S = cu.numel() - 1
        slopes = None if slopes is None else slopes.float().contiguous()
        out = torch.empty(T, H, hd, dtype=torch.bfloat16, device=qkv.device)
        lse = torch.empty(H, T, dtype=torch.float32, device=qkv.device)
        ws = _ws(lib().resel_attn_varlen_fwd_workspace_bytes(S, int(max_seqlen)), qkv.device)
        check(lib().resel_attn_varlen_fwd(_p(qkv), _p(cu), _p(slopes), _p(out), _p(lse), _p(ws), T, S, H, hd, int(max_seqlen), float(scale),
                                          float(p_drop), int(seed), int(offset), _stream()), 'attn_varlen_fwd')
        ctx.save_for_backward(qkv, cu, slopes, out, lse)
        ctx.max_seqlen, ctx.scale, ctx.drop = int(max_seqlen), float(scale), (float(p_drop), int(seed), int(offset))
        return out

    @staticmethod
    def backward(ctx, dout):
        qkv, cu, slopes, out, lse = ctx.saved_tensors
        T, _, H, hd = qkv.shape
        dout = dout.to(torch.bfloat16).contiguous()
        dqkv = torch.empty_like(qkv)
        ws = _ws(lib().resel_attn_varlen_bwd_workspace_bytes(T, cu.numel() - 1, H, hd, ctx.max_seqlen), qkv.device)
        check(lib().resel_attn_varlen_bwd(_p(qkv), _p(cu), _p(slopes), _p(out), _p(lse), _p(dout), _p(dqkv), _p(ws), T, cu.numel() - 1, H, hd,
                                          ctx.max_seqlen, ctx.scale, *ctx.drop, _stream()), 'attn_varlen_bwd')
        return dqkv, None, None, None, None, None, None, None


_MASK64 = (1 << 64) - 1


DROP_OVERRIDE = None           # [seed, next offset] while a GraphedUpdate body runs (recorded or eager): see dropout_offset_base
_DROP_BASE_KEEP = []           # the device words handed to resel_dropout_offset_base stay alive as long as the process


def dropout_offset_base(word):
    """Install (None: remove) the device int64 word that every counter-keyed mask kernel adds to its offset when it runs
    (include/resel_hip.h `resel_dropout_offset_base`): a captured update advances it with a node of its own graph, so that a replay
    - whose kernel nodes carry the host-drawn offsets of the recording - draws fresh masks."""
    if word is not None:
        assert word.is_cuda and word.dtype == torch.int64 and word.numel() == 1
        _DROP_BASE_KEEP.append(word)
    check(lib().resel_dropout_offset_base(_p(word)), 'dropout_offset_base')


def dropout_counter(device):
    """(seed, offset) for one counter-keyed dropout mask, taken from the device's default torch generator the way ATen's
    own dropout kernels reserve Philox offsets: host-side bookkeeping only (no sync), deterministic under
    `torch.manual_seed`, and every call gets a fresh offset.  Inside a GraphedUpdate body (torch's generator may not be
    queried while a stream is capturing) the draws count up from 0 per update and the device-side base makes updates differ."""
    if DROP_OVERRIDE is not None:
        off = DROP_OVERRIDE[1]
        DROP_OVERRIDE[1] = off + 4
        return DROP_OVERRIDE[0], off
    if not torch.cuda.default_generators:            # first CUDA touch of the process: the generator tuple is filled by the lazy init
        torch.cuda.init()
    gen = torch.cuda.default_generators[device.index if device.index is not None else torch.cuda.current_device()]
    off = gen.get_offset()
    gen.set_offset(off + 4)
    return gen.initial_seed() & _MASK64, off & _MASK64


def attn_varlen(qkv, cu_seqlens, max_seqlen, slopes=None, scale=None, p_drop=0.0, seed=0, offset=0):
    """qkv [T, 3, H, hd] bf16 packed tokens -> out [T, H, hd] bf16: causal softmax(q k^T * scale - slope_h (i - j)) v per sequence.
    p_drop > 0: dropout on the attention probabilities with the keep mask keyed on (seed, offset) (see resel_hip.h)."""
    scale = qkv.shape[-1] ** -0.5 if scale is None else scale
    return AttnVarlenFn.apply(qkv, cu_seqlens, max_seqlen, slopes, scale, p_drop, seed, offset)


class CounterDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p_drop, seed, offset):
        _need_cuda('dropout', x)
        assert x.dtype == torch.float32
        x = x.contiguous()
        y = torch.empty_like(x)
        check(lib().resel_dropout(_p(x), _p(y), x.numel(), float(p_drop), int(seed), int(offset), _stream()), 'dropout')
        ctx.drop = (float(p_drop), int(seed), int(offset))
        return y

    @staticmethod
    def backward(ctx, dy):
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        check(lib().resel_dropout(_p(dy), _p(dx), dy.numel(), *ctx.drop, _stream()), 'dropout')
        return dx, None, None, None


def counter_dropout(x, p_drop, seed=None, offset=None):
    """Training-mode dropout of an fp32 tensor with a mask keyed on (seed, offset, flat element index); seed / offset default
    to a fresh `dropout_counter` draw."""
    if p_drop <= 0.0:
        return x
    if seed is None:
        seed, offset = dropout_counter(x.device)
    return CounterDropoutFn.apply(x, p_drop, seed, offset)


class GeluDropoutFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, x, p_drop, seed, offset):
        _need_cuda('gelu_dropout', x)
        assert x.dtype == torch.float32
        x = x.contiguous()
        y = torch.empty_like(x)
        global LAST_AMAX
        slot, slot_p, epoch = _slot_args(amax_tracking() and x.numel() >= (1 << 20), x.device)
        check(lib().resel_gelu_dropout_fwd(_p(x), _p(y), x.numel(), float(p_drop), int(seed), int(offset), slot_p, epoch, _stream()), 'gelu_dropout_fwd')
        LAST_AMAX = slot
        ctx.save_for_backward(x)
        ctx.drop = (float(p_drop), int(seed), int(offset))
        return y

    @staticmethod
    def backward(ctx, dy):
        x, = ctx.saved_tensors
        dy = dy.contiguous()
        dx = torch.empty_like(dy)
        slot, slot_p, epoch = _slot_args(amax_tracking() and dy.numel() >= (1 << 20), dy.device)
        check(lib().resel_gelu_dropout_bwd(_p(x), _p(dy), _p(dx), dy.numel(), *ctx.drop, slot_p, epoch, _stream()), 'gelu_dropout_bwd')
        return tag_amax(dx, slot), None, None, None


def gelu_dropout(x, p_drop, seed=None, offset=None):
    """dropout(gelu(x)) (erf GELU) as one kernel forward and one backward; the mask is the `counter_dropout` mask of the same
    (seed, offset) - one `dropout_counter` draw when p_drop > 0, none otherwise (plain GELU)."""
    if p_drop > 0.0 and seed is None:
        seed, offset = dropout_counter(x.device)
    global LAST_AMAX
    LAST_AMAX = None
    return tag_amax(GeluDropoutFn.apply(x, p_drop if p_drop > 0.0 else 0.0, seed or 0, offset or 0), LAST_AMAX)


# ---------------------------------------------------------------------------------------------- SAC / TD3 arithmetic
class TanhGaussianFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, out2, noise):
        _need_cuda('tanh_gaussian', out2, noise)
        shape = out2.shape
        A = shape[-1] // 2
        o2 = out2.float().reshape(-1, 2 * A).contiguous()
        nz = noise.float().reshape(-1, A).contiguous()
        M = o2.shape[0]
        mean, samp = torch.empty_like(nz), torch.empty_like(nz)
        logp = torch.empty(M, dtype=torch.float32, device=o2.device)
        check(lib().resel_tanh_gaussian_fwd(_p(o2), _p(nz), _p(mean), _p(samp), _p(logp), M, A, _stream()), 'tanh_gaussian_fwd')
        ctx.save_for_backward(o2, nz)
        ctx.shape = shape
        ctx.mark_non_differentiable(mean)
        out_shape = shape[:-1] + (A,)
        return mean.reshape(out_shape), samp.reshape(out_shape), logp.reshape(shape[:-1] + (1,))

    @staticmethod
    def backward(ctx, _dmean, dsamp, dlogp):
        o2, nz = ctx.saved_tensors
        M, A = nz.shape
        ds = None if dsamp is None else dsamp.float().reshape(M, A).contiguous()
        dl = None if dlogp is None else dlogp.float().reshape(M).contiguous()
        d2 = torch.empty_like(o2)
        check(lib().resel_tanh_gaussian_bwd(_p(o2), _p(nz), _p(ds), _p(dl), _p(d2), M, A, _stream()), 'tanh_gaussian_bwd')
        return d2.reshape(ctx.shape), None


def tanh_gaussian(out2, noise):
    """(logstd | mean) [..., 2A], noise [..., A] -> tanh(mean) (no grad), tanh(sample), log_prob [..., 1]."""
    return TanhGaussianFn.apply(out2, noise)


@torch.no_grad()
def sac_target(q, subset, next_logp, log_alpha, reward, done, mask, gamma, guard, stats=None, reduce_max=None, local_ext=None):
    """q [E, ...]; subset int32 [m] (device); guard fp32[4] device state {min, max, initialised, decay}.
    Data parallel, either
      local_ext (fp32 [4], default of the trainer): the target of the rank's own rows with the guard as it stands; the rank-local
        extrema go to local_ext, travel in the gradient bucket and reach the guard through `guard_apply_slots` - no collective here;
      reduce_max: in-place MAX all-reduce of a small device tensor - three phases with two 2-float exchanges inside the target."""
    _need_cuda('sac_target', q, reward, done, guard)
    E = q.shape[0]
    M = reward.numel()
    qf = q.float().reshape(E, M).contiguous()
    target = torch.empty(M, dtype=torch.float32, device=q.device)
    ws = _ws(lib().resel_sac_target_workspace_bytes(M), q.device)
    f = lambda t: None if t is None else t.float().reshape(-1).contiguous()
    nl, rw, dn, mk = f(next_logp), f(reward), f(done), f(mask)
    if local_ext is not None:
        check(lib().resel_sac_target_local(_p(qf), _p(subset), int(subset.numel()), _p(nl), _p(log_alpha), _p(rw), _p(dn), _p(mk),
                                           float(gamma), _p(guard), _p(target), _p(stats), _p(local_ext), _p(ws), E, M, _stream()), 'sac_target_local')
        return target.reshape(reward.shape)
    if reduce_max is not None:
        ext = torch.empty(4, dtype=torch.float32, device=q.device)
        for phase in range(3):
            check(lib().resel_sac_target_phase(phase, _p(qf), _p(subset), int(subset.numel()), _p(nl), _p(log_alpha), _p(rw), _p(dn), _p(mk),
                                               float(gamma), _p(guard), _p(target), _p(stats), _p(ext), _p(ws), E, M, _stream()), 'sac_target_phase')
            if phase < 2:
                reduce_max(ext[2 * phase:2 * phase + 2])
        return target.reshape(reward.shape)
    check(lib().resel_sac_target(_p(qf), _p(subset), int(subset.numel()), _p(nl), _p(log_alpha), _p(rw), _p(dn), _p(mk),
                                 float(gamma), _p(guard), _p(target), _p(stats), _p(ws), E, M, _stream()), 'sac_target')
    return target.reshape(reward.shape)


class MaskedQLossFn(torch.autograd.Function):
    """sum_m mask[m] sum_e (q[e, m] - y[m])^2 as ONE node (include/resel_hip.h `resel_q_loss_*`)."""

    @staticmethod
    def forward(ctx, q, y, mask):
        _need_cuda('q_loss', q, y)
        E = q.shape[0]
        q2, y1, m1 = q.float().reshape(E, -1).contiguous(), y.float().reshape(-1).contiguous(), None if mask is None else mask.float().reshape(-1).contiguous()
        M = y1.numel()
        out = torch.empty(2, dtype=torch.float32, device=q.device)
        ws = _ws(lib().resel_masked_loss_workspace_bytes(), q.device)
        check(lib().resel_q_loss_fwd(_p(q2), _p(y1), _p(m1), _p(out), _p(ws), E, M, _stream()), 'q_loss_fwd')
        ctx.save_for_backward(q2, y1, m1)
        ctx.shape = q.shape
        return out[0]

    @staticmethod
    def backward(ctx, g):
        q2, y1, m1 = ctx.saved_tensors
        E, M = q2.shape
        dq = torch.empty_like(q2)
        check(lib().resel_q_loss_bwd(_p(q2), _p(y1), _p(m1), _p(g.float().reshape(1).contiguous()), _p(dq), E, M, _stream()), 'q_loss_bwd')
        return dq.view(ctx.shape), None, None


def masked_q_loss(q, y, mask):
    return MaskedQLossFn.apply(q, y, mask)


class MaskedActorLossFn(torch.autograd.Function):
    """(sum_m mask (use_logp alpha logp - red_e q[e]), sum_m mask logp) as ONE node; red = mean (REDQ) or min (ensemble-min); alpha =
    exp(log_alpha) is a constant of the objective (the entropy coefficient has its own loss).  The second output carries no gradient."""

    @staticmethod
    def forward(ctx, logp, q, mask, log_alpha, use_logp, reduce_min):
        _need_cuda('actor_loss', q)
        E = q.shape[0]
        q2 = q.float().reshape(E, -1).contiguous()
        M = q2.shape[1]
        lp = None if logp is None else logp.float().reshape(-1).contiguous()
        m1 = None if mask is None else mask.float().reshape(-1).contiguous()
        la = log_alpha.detach().float().reshape(-1).contiguous()
        out = torch.empty(2, dtype=torch.float32, device=q.device)
        ws = _ws(lib().resel_masked_loss_workspace_bytes(), q.device)
        check(lib().resel_actor_loss_fwd(_p(lp), _p(q2), _p(m1), _p(la), _p(out), _p(ws), E, M, int(bool(use_logp)), int(bool(reduce_min)), _stream()),
              'actor_loss_fwd')
        ctx.save_for_backward(q2, m1, la)
        ctx.cfg = (bool(use_logp), bool(reduce_min), q.shape, None if logp is None else logp.shape)
        return out                                    # ONE output [2] = (objective sum, sum mask logp): the caller indexes it

    @staticmethod
    def backward(ctx, g2):
        q2, m1, la = ctx.saved_tensors
        g = g2[0:1]                                   # the second entry is a statistic (used detached)
        use_logp, reduce_min, qshape, lshape = ctx.cfg
        E, M = q2.shape
        dq = torch.empty_like(q2)
        dlp = torch.empty(M, dtype=torch.float32, device=q2.device) if use_logp else None
        check(lib().resel_actor_loss_bwd(_p(q2), _p(m1), _p(la), _p(g.float().reshape(1).contiguous()), _p(dlp), _p(dq), E, M, int(use_logp),
                                         int(reduce_min), _stream()), 'actor_loss_bwd')
        return (None if dlp is None or lshape is None else dlp.view(lshape)), dq.view(qshape), None, None, None, None


def masked_actor_loss(logp, q, mask, log_alpha, use_logp=True, reduce_min=False):
    out = MaskedActorLossFn.apply(logp, q, mask, log_alpha, use_logp, reduce_min)
    return out[0], out[1].detach()


@torch.no_grad()
def guard_apply_slots(slots, world, guard):
    """slots fp32 [world * 4] (every rank's extrema, delivered by the gradient all-reduce) -> first-call initialisation + running
    update of the Q guard from the extrema of the global batch (include/resel_hip.h `resel_guard_apply_slots`)."""
    _need_cuda('guard_apply_slots', slots, guard)
    check(lib().resel_guard_apply_slots(_p(slots), int(world), _p(guard), _stream()), 'guard_apply_slots')


# ---------------------------------------------------------------------------------------------- bias + activation tail
ACT_IDS = {None: 0, 'linear': 0, 'elu': 1}
GEMM_SOFTPLUS = 'softplus'           # resel_gemm_f32x only: softplus(product + bias) (epilogue code 3)
GEMM_ACCUMULATE = 'accumulate'       # resel_gemm_f32 only: C += product (epilogue code 2)


@torch.no_grad()
def bias_act_(y2, bias2, rows_per_seg, act):
    """In place on the GEMM output y2 [rows, C] (contiguous): y2 <- act(y2 + bias2[row // rows_per_seg]); bias2 [nseg, C] or None."""
    _need_cuda('bias_act', y2, bias2)
    assert y2.is_contiguous() and y2.dtype == torch.float32 and (bias2 is None or bias2.is_contiguous())
    check(lib().resel_bias_act_fwd(_p(y2), _p(bias2), y2.shape[0], y2.shape[1], int(rows_per_seg), ACT_IDS[act], _stream()), 'bias_act_fwd')
    tag_amax(y2, None)                                # rewritten in place
    return y2


@torch.no_grad()
def bias_act_bwd(g2, a2, rows_per_seg, act, need_dbias):
    """gy = g2 * act'(.) computed from the forward OUTPUT a2, dbias [nseg, C] = per-segment column sums of gy (or None)."""
    _need_cuda('bias_act_bwd', g2, a2)
    # a column block of a wider gradient (the halves of a `cat` backward) is read in place through its row stride
    if not (g2.dim() == 2 and g2.stride(1) == 1 and g2.stride(0) >= g2.shape[1] and g2.stride(0) % 4 == 0 and g2.data_ptr() % 16 == 0):
        g2 = g2.contiguous()
    rows, C = g2.shape
    aid = ACT_IDS[act]
    if aid == 0 and not need_dbias:
        return g2, None
    if aid and not (a2.dim() == 2 and a2.stride(1) == 1 and a2.stride(0) >= C and a2.stride(0) % 4 == 0 and a2.data_ptr() % 16 == 0):
        a2 = a2.contiguous()                          # (a block of a row buffer qualifies as it is)
    gy = torch.empty(rows, C, dtype=torch.float32, device=g2.device) if aid else g2
    nseg = rows // int(rows_per_seg)
    db = torch.empty(nseg, C, dtype=torch.float32, device=g2.device) if need_dbias else None
    ws = _ws(lib().resel_bias_act_bwd_workspace_bytes(rows, C, int(rows_per_seg)), g2.device) if need_dbias else None
    global LAST_AMAX
    slot, slot_p, epoch = _slot_args(amax_tracking() and rows * C >= (1 << 20), g2.device)
    check(lib().resel_bias_act_bwd(_p(g2), g2.stride(0), _p(a2) if aid else None, a2.stride(0) if aid else 0, _p(gy) if aid else None, _p(db), _p(ws), rows, C, int(rows_per_seg), aid,
                                   slot_p, epoch, _stream()), 'bias_act_bwd')
    tag_amax(gy, slot)
    LAST_AMAX = slot
    return gy, db


@torch.no_grad()
def ensemble_head_fwd_(y3, b2, w3, b3):
    """y3 [E, M, H] (GEMM output, overwritten with a = elu(y + b2)), b2 / w3 [E, H], b3 [E] -> q [E, M]."""
    _need_cuda('ensemble_head', y3, b2, w3, b3)
    E, M, H = y3.shape
    q = torch.empty(E, M, dtype=torch.float32, device=y3.device)
    check(lib().resel_ensemble_head_fwd(_p(y3), _p(b2), _p(w3), _p(b3), _p(q), E * M, H, M, _stream()), 'ensemble_head_fwd')
    tag_amax(y3, None)                                # rewritten in place: the GEMM's magnitude no longer describes it
    return q


@torch.no_grad()
def ensemble_head_bwd(gq, a3, w3):
    """gq [E, M], a3 [E, M, H], w3 [E, H] -> gy [E, M, H], db2 [E, H], dw3 [E, H]."""
    _need_cuda('ensemble_head_bwd', gq, a3, w3)
    E, M, H = a3.shape
    gq = gq if gq.is_contiguous() else gq.contiguous()
    gy = torch.empty_like(a3)
    db2 = torch.empty(E, H, dtype=torch.float32, device=a3.device)
    dw3 = torch.empty(E, H, dtype=torch.float32, device=a3.device)
    ws = _ws(lib().resel_ensemble_head_bwd_workspace_bytes(E * M, H, M), a3.device)
    global LAST_AMAX
    slot, slot_p, epoch = _slot_args(amax_tracking() and E * M * H >= (1 << 20), a3.device)
    check(lib().resel_ensemble_head_bwd(_p(gq), _p(a3), _p(w3), _p(gy), _p(db2), _p(dw3), _p(ws), E * M, H, M, slot_p, epoch, _stream()),
          'ensemble_head_bwd')
    tag_amax(gy, slot)
    LAST_AMAX = slot
    return gy, db2, dw3


class LinearAct(torch.autograd.Function):
    """act(x W^T + b) for nn.Linear weights (reference rnn_base.py:462-474: `fc` layer followed by its activation module):
    one GEMM with the bias / activation tail (`mm_nt`); the backward needs the layer OUTPUT only.  On long GPU passes an input
    width (17- / 6-wide encoders) or an output width (6-wide TD3 action head) that is not a multiple of 4 is zero-padded inside
    the node - exact zeros in every dot product, the padding rows / columns of the gradients are dropped - so that every
    contraction of the update runs on the hand-written GEMM."""

    @staticmethod
    def forward(ctx, x, weight, bias, act, dest=None):
        """dest: a `ColDest` - the column block of a row buffer the output should be written to in place (when the hand-written GEMM runs
        and nothing is padded; otherwise the output is a fresh tensor and whoever assembles the buffer copies it in)."""
        x2 = x.reshape(-1, x.shape[-1])
        x2 = x2 if x2.stride(-1) == 1 else x2.contiguous()
        long_pass = x2.is_cuda                         # (rounds 2-5: passes of 4 096 tokens and more)
        ctx.kpad = (-x2.shape[1]) % 4 if long_pass else 0
        ctx.npad = (-weight.shape[0]) % 4 if long_pass else 0
        n_out = weight.shape[0]
        if ctx.kpad or ctx.npad:
            x2 = torch.nn.functional.pad(x2, (0, ctx.kpad)) if ctx.kpad else x2
            weight = torch.nn.functional.pad(weight, (0, ctx.kpad, 0, ctx.npad))
            bias = torch.nn.functional.pad(bias, (0, ctx.npad)) if (bias is not None and ctx.npad) else bias
        if dest is not None and long_pass and not ctx.npad and dest.fits(x2.shape[0], weight.shape[0]):
            y2 = mm_nt(x2, weight, bias, act, out=dest.view(), amax_out=dest.rb.amax if dest.rb.handle() is not None else None)
            dest.note(y2)
        else:
            y2 = mm_nt(x2, weight, bias, act)
        ctx.ax = keep_handles(amax_of(x2))[0]         # the input's magnitude handle, for the weight gradient (saved tensors come back untagged)
        ctx.save_for_backward(x2, weight, y2)
        ctx.act, ctx.has_bias, ctx.xshape = act, bias is not None, x.shape
        out = y2[:, :n_out] if ctx.npad else y2
        return out.reshape(*x.shape[:-1], n_out)

    @staticmethod
    def backward(ctx, g):
        x2, weight, y2 = ctx.saved_tensors
        n_out = y2.shape[1] - ctx.npad
        g2 = g.reshape(y2.shape[0], n_out)
        if ctx.npad:
            g2 = torch.nn.functional.pad(g2, (0, ctx.npad))
        need_db = ctx.has_bias and ctx.needs_input_grad[2]
        if y2.shape[1] % 4:                                           # short passes keep odd widths: plain torch tail
            g2 = g2 if g2.is_contiguous() else g2.contiguous()
            aid = ACT_IDS[ctx.act]                                     # 0: identity ('linear' / None), 1: ELU; anything else is a KeyError
            gy = g2 if aid == 0 else g2 * torch.where(y2 > 0, torch.ones_like(y2), y2 + 1.0)          # elu'(x) from the output
            db = gy.sum(dim=0) if need_db else None
        else:
            gy, db = bias_act_bwd(g2, y2, y2.shape[0], ctx.act, need_db)
        dx = mm_nn(gy, weight) if ctx.needs_input_grad[0] else None
        dw = wgrad(gy, x2, amax_x=handle_alive(ctx.ax)) if ctx.needs_input_grad[1] else None
        if ctx.kpad or ctx.npad:                      # drop the padding rows / columns again
            k = x2.shape[1] - ctx.kpad
            dx = None if dx is None else dx[:, :k]
            dw = None if dw is None else dw[:n_out, :k]
            db = None if db is None else db.reshape(-1)[:n_out]
        dx = None if dx is None else dx.reshape(ctx.xshape)
        return dx, dw, None if db is None else db.reshape(-1), None, None


class PlaceBlocksFn(torch.autograd.Function):
    """out [rows, cols] = zeros with the source blocks copied in at their (r0, c0) (include/resel_hip.h `resel_place_blocks`): ONE launch for what
    block_diag / cat / pad assemble in several; the backward hands every source the matching slice of the gradient as a VIEW.
    srcs: fp32 tensors [..., nc_i]; their leading axes collapse to the nr_i rows of the block."""

    @staticmethod
    def forward(ctx, rows, cols, origins, *srcs):
        _need_cuda('place_blocks', *srcs)
        n = len(srcs)
        assert 1 <= n <= 8 and len(origins) == n
        s2 = []
        for t in srcs:
            t2 = t.reshape(-1, t.shape[-1])                                    # a view whenever the leading axes collapse
            s2.append(t2 if (t2.dtype == torch.float32 and t2.stride(1) == 1) else t2.float().contiguous())
        out = torch.empty(rows, cols, dtype=torch.float32, device=srcs[0].device)
        VP, LL, II = ctypes.c_void_p * n, ctypes.c_int64 * n, ctypes.c_int * n
        check(lib().resel_place_blocks(_p(out), cols, rows, cols, n, VP(*[t.data_ptr() for t in s2]), LL(*[t.stride(0) for t in s2]),
                                       II(*[o[0] for o in origins]), II(*[t.shape[0] for t in s2]), II(*[o[1] for o in origins]),
                                       II(*[t.shape[1] for t in s2]), _stream()), 'place_blocks')
        ctx.meta = [(o[0], o[1], t2.shape[0], t2.shape[1], tuple(t.shape)) for o, t2, t in zip(origins, s2, srcs)]
        return out

    @staticmethod
    def backward(ctx, g):
        outs = []
        for need, (r0, c0, nr, nc, shape) in zip(ctx.needs_input_grad[3:], ctx.meta):
            outs.append(g[r0:r0 + nr, c0:c0 + nc].reshape(shape) if need else None)    # a view of g (rows of a block keep g's row stride)
        return (None, None, None) + tuple(outs)


def place_blocks(rows, cols, origins, *srcs):
    return PlaceBlocksFn.apply(int(rows), int(cols), tuple((int(r), int(c)) for r, c in origins), *srcs)


def linear_act(x, weight, bias, act, dest=None):
    global LAST_AMAX
    LAST_AMAX = None
    # (an output written into a row buffer is a view of it: its magnitude must not be taken for the whole buffer's)
    return tag_amax(LinearAct.apply(x, weight, bias, act, dest), LAST_AMAX, whole=dest is None)


# ---- row buffers: the column blocks of ONE token-major matrix written in place by the GEMMs that produce them -----------------------
class RowBuffer:
    """[M, width] fp32 with M = prod(lead): what `torch.cat([...], -1)` of several layer outputs would build, allocated up front so that
    each producer's GEMM writes its column block directly (`dest=` of `linear_act`; reference contextual_model.py cat of the input
    encoding and the embedding, contextual_sac_value.py:101-107).  The producers publish into ONE magnitude handle (the buffer's);
    `cat_into` hands the buffer on - copying in whatever did not land in place - as one autograd node whose backward slices."""

    def __init__(self, lead, width, device):
        self.lead, self.width = tuple(int(v) for v in lead), int(width)
        self.rows = 1
        for v in self.lead:
            self.rows *= v
        self.buf = torch.empty(self.rows, self.width, dtype=torch.float32, device=device)
        self.amax = _slot_args(amax_tracking() and self.rows * self.width >= (1 << 20), self.buf.device)     # (handle, pointer, epoch)
        # the handle's tenant serial AT ALLOCATION: the arena has 2 048 slots and a deep embedding tower (or RESEL_AMAX_VERIFY) may hand
        # out more than that between this constructor and `cat_into` - the slot then belongs to another tensor and no tag may name it
        self.serial = getattr(self.amax[0], '_resel_serial', None)
        self.published = 0                               # columns whose producer published into the handle

    def handle(self):
        """The buffer's magnitude handle while its arena slot still has the tenant it had at allocation, else None."""
        h = self.amax[0]
        return h if h is not None and getattr(h, '_resel_serial', None) == self.serial else None

    def block(self, col0, n):
        return ColDest(self, col0, n)


class ColDest:
    def __init__(self, rb, col0, n):
        self.rb, self.col0, self.n = rb, int(col0), int(n)

    def view(self):
        return self.rb.buf[:, self.col0:self.col0 + self.n]

    def fits(self, rows, n):
        return rows == self.rb.rows and n == self.n and self.col0 % 4 == 0 and self.n % 4 == 0

    def holds(self, t):
        """t (any leading shape) IS this block of the buffer."""
        if t.shape[-1] != self.n or t.numel() != self.rb.rows * self.n or t.dtype != torch.float32:
            return False
        v = self.view()
        return t.data_ptr() == v.data_ptr() and t.stride(-1) == 1 and (t.dim() < 2 or t.stride(-2) == self.rb.width) and \
            t.untyped_storage().data_ptr() == v.untyped_storage().data_ptr()

    def note(self, y2):
        if self.holds(y2):
            self.rb.published += self.n


class CatInto(torch.autograd.Function):
    """pieces (tensor_i at column col0_i of the row buffer) -> the whole buffer as [lead..., width]; pieces that are already in place cost
    nothing, the others one strided copy each.  Backward: column slices of the gradient (views)."""

    @staticmethod
    def forward(ctx, rb, col0s, *pieces):
        for c0, t in zip(col0s, pieces):
            d = rb.block(c0, t.shape[-1])
            if not d.holds(t):
                # copied by this library's own kernel: an ATen copy_ would bump the version counter the buffer shares with the views that the
                # in-place producers saved for their backward
                t2 = t.reshape(rb.rows, t.shape[-1])
                t2 = t2 if (t2.dtype == torch.float32 and t2.stride(1) == 1) else t2.float().contiguous()
                _need_cuda('cat_into', t2)
                check(lib().resel_place_blocks(_p(d.view()), rb.width, rb.rows, d.n, 1, (ctypes.c_void_p * 1)(t2.data_ptr()), (ctypes.c_int64 * 1)(t2.stride(0)),
                                               (ctypes.c_int * 1)(0), (ctypes.c_int * 1)(rb.rows), (ctypes.c_int * 1)(0), (ctypes.c_int * 1)(d.n), _stream()),
                      'cat_into')
        ctx.meta = [(c0, t.shape[-1], tuple(t.shape)) for c0, t in zip(col0s, pieces)]
        return rb.buf.detach().view(*rb.lead, rb.width)

    @staticmethod
    def backward(ctx, g):
        g2 = g.reshape(-1, g.shape[-1])
        outs = [g2[:, c0:c0 + n].view(shape) if need else None for need, (c0, n, shape) in zip(ctx.needs_input_grad[2:], ctx.meta)]
        return (None, None) + tuple(outs)


def cat_into(rb, pieces):
    """pieces: [(tensor, col0), ...] covering the buffer's columns."""
    complete = all(rb.block(c, t.shape[-1]).holds(t) for t, c in pieces) and sum(t.shape[-1] for t, _ in pieces) == rb.width \
        and rb.published == rb.width
    out = CatInto.apply(rb, tuple(int(c) for _, c in pieces), *[t for t, _ in pieces])
    return tag_amax(out, rb.handle() if complete else None, whole=True)


def linear(x, weight, bias=None):
    """x W^T + b for an fp32 nn.Linear: the `LinearAct` node (hand-written GEMM forward, input and weight gradients) for every pass -
    the whole trajectories of an update and the single token of a rollout step alike (`resel_gemm_f32x` takes every shape: the matrix-core
    editions where the layout allows them, csrc/gemm_any.hip for the rest).  An input width that is not a multiple of 4 (the 17-wide
    observation / 6-wide action encoders) is zero-padded to 16-byte rows inside the node.  There is no library GEMM behind this module
    (round 5 kept `F.linear` for passes under 4 096 tokens); CPU tensors raise in `gemm_f32`."""
    return linear_act(x, weight, bias, None)


# Every CUDA fp32 pass takes the hand-written kernels since round 6 (the constant is what remains of the round 2-5 switch between them
# and the GEMM library: 4 096 tokens then).  Code that asked "is this a long pass?" now asks "is this a CUDA pass?".
GEMM_F32_MIN_ROWS = 1


def gemm_f32_ok(rows, *mats):
    """True when `resel_gemm_f32` takes these operands as they are: on the GPU, fp32, unit column stride (any row alignment: the C entry
    routes shapes the matrix-core editions cannot read to csrc/gemm_any.hip)."""
    return rows >= 1 and all(t.is_cuda and t.dtype == torch.float32 and t.stride(-1) == 1 for t in mats)


def rows_aligned16(*mats):
    """Every row of every operand starts on a 16-byte boundary and is a whole number of float4s: what the matrix-core editions (and their
    fused epilogues, which have no other form) read."""
    return all(t.data_ptr() % 16 == 0 and t.shape[-1] % 4 == 0 and all(st % 4 == 0 for st in t.stride()[:-1])
               and (t.dim() < 2 or t.stride(-2) < (1 << 22)) for t in mats)


GEMM_F32_MIN_DIM = 1        # (rounds 2-5: outputs / reductions narrower than 4 stayed with the library)
GEMM_F32_MIN_K = 1


def _unit(t):
    """t with unit column stride (what the C entry's row-stride description needs); a copy only for exotic views."""
    return t if t.stride(-1) == 1 else t.contiguous()


def mm_nt(x2, w, bias=None, act=None, out=None, amax_out=None):
    """act(x2 [M, K] w[N, K]^T + bias): forward of an nn.Linear-shaped layer over the tokens of a pass, bias / ELU in the GEMM epilogue.
    out (with amax_out): a column block of a row buffer to write in place."""
    ACT_IDS[act]                                                       # 0: identity ('linear' / None), 1: ELU; anything else is a KeyError
    x2, w = _unit(x2), _unit(w)
    if out is not None and out.stride(-1) == 1:
        return gemm_f32(x2, w, True, True, bias, act, out=out, amax_out=amax_out)
    return gemm_f32(x2, w, True, True, bias, act)


def mm_nn(g2, w, out=None):
    """g2 [M, N] w[N, K] -> [M, K]: input gradient of such a layer.  out: a (possibly row-strided) destination, e.g. a column block of a
    wider buffer."""
    return gemm_f32(_unit(g2), _unit(w), True, False, out=out)


def wgrad(gy, x2, amax_x=None):
    """dW [out, in] = gy[M, out]^T x2[M, in] over the M tokens of a pass: the hand-written K-split GEMM (at 66 752 tokens:
    [128, 256] 49 us against the library's 234, [2048, 384] 731 against 1217, [256, 256] 86 against 123; `tools/bench_gemm_f32.py`)."""
    return gemm_f32(_unit(gy), _unit(x2), False, False, amax_b=amax_x)


@torch.no_grad()
def gather_trajs(buffer, segments, max_len, skip, rows, row_len, c_mask, c_start, c_done, c_timeout, pre_pairs):
    """Packed batch [rows, row_len, W + 3] from the device ring `buffer` [capacity, W] and the int32 plan `segments` [nseg, 4]
    (row, first slot, length incl. skip, first transition); see include/resel_hip.h `resel_gather_trajs`."""
    _need_cuda('gather_trajs', buffer, segments, pre_pairs)
    assert buffer.dtype == torch.float32 and buffer.is_contiguous() and segments.dtype == torch.int32 and segments.is_contiguous()
    W = buffer.shape[1]
    out = torch.empty(rows, row_len, W + 3, dtype=torch.float32, device=buffer.device)
    check(lib().resel_gather_trajs(_p(buffer), W, buffer.shape[0], _p(segments), segments.shape[0], int(max_len), int(skip), int(rows), int(row_len),
                                   int(c_mask), int(c_start), int(c_done), int(c_timeout), _p(pre_pairs), pre_pairs.shape[0], _p(out),
                                   _stream()), 'gather_trajs')
    return out


@torch.no_grad()
def atb(wide, narrow, transposed=False):
    """wide [K, Wd] (column stride 1, Wd % 4 == 0), narrow [K, Nd <= 96] -> wide^T narrow [Wd, Nd] (or [Nd, Wd] if transposed):
    the long-reduction weight-gradient GEMMs of the narrow Mamba projections on a hand-written fp32 MFMA kernel."""
    _need_cuda('atb', wide, narrow)
    K, Wd = wide.shape
    Nd = narrow.shape[1]
    assert narrow.shape[0] == K and wide.stride(1) == 1 and narrow.stride(1) == 1
    out = torch.empty((Nd, Wd) if transposed else (Wd, Nd), dtype=torch.float32, device=wide.device)
    ws = _ws(lib().resel_atb_workspace_bytes(K, Wd, Nd), wide.device)
    check(lib().resel_atb(_p(wide), wide.stride(0), Wd, _p(narrow), narrow.stride(0), Nd, _p(out), int(bool(transposed)), _p(ws), K,
                          _stream()), 'atb')
    return out


# ---------------------------------------------------------------------------------------------- bf16 projections (cgpt)
@torch.no_grad()
def gemm_bf16(A, B, a_kcontig=True, b_kcontig=True, bias=None, out_dtype=torch.bfloat16, round_out=False):
    """C [M, N] = bf16(A) (.) bf16(B) + bf16(bias), fp32 accumulation (include/resel_hip.h `resel_gemm_bf16`).  A: [M, K]
    (a_kcontig) or [K, M]; B: [N, K] (b_kcontig) or [K, N]; each fp32 or bf16 (rounded to bf16 on the way into LDS); bias fp32.
    round_out (fp32 output only): store the bf16-ROUNDED value as fp32 - what `F.linear(...).to(float32)` leaves under bf16 autocast."""
    _need_cuda('gemm_bf16', A, B)
    assert A.dim() == 2 and B.dim() == 2 and A.stride(-1) == 1 and B.stride(-1) == 1
    assert A.dtype in (torch.float32, torch.bfloat16) and B.dtype in (torch.float32, torch.bfloat16)
    M, K = (A.shape[0], A.shape[1]) if a_kcontig else (A.shape[1], A.shape[0])
    N = B.shape[0] if b_kcontig else B.shape[1]
    assert (B.shape[1] if b_kcontig else B.shape[0]) == K
    out = torch.empty((M, N), dtype=out_dtype, device=A.device)
    L = lib()
    ws = _ws(L.resel_gemm_bf16_workspace_bytes(M, N, K), A.device)
    check(L.resel_gemm_bf16(_p(A), A.stride(0), int(a_kcontig), int(A.dtype == torch.bfloat16), _p(B), B.stride(0), int(b_kcontig),
                            int(B.dtype == torch.bfloat16), _p(bias), _p(out), out.stride(0), 1 if out_dtype == torch.bfloat16 else (2 if round_out else 0), _p(ws),
                            M, N, K, _stream()), 'gemm_bf16')
    return out


def gemm_bf16_ok(x, w):
    """Operands the mixed-precision GEMM node takes: token rows on the GPU, fp32 master weight (any row count and alignment since round 6:
    `resel_gemm_bf16` routes what its matrix-core kernel cannot read, and the few rows of a decode step, to csrc/gemm_any.hip)."""
    return (x.is_cuda and x.dim() == 2 and x.stride(-1) == 1 and x.dtype in (torch.float32, torch.bfloat16) and w.dtype == torch.float32
            and w.is_contiguous())


@torch.no_grad()
def colsum_bf16(x):
    """x [M, N] bf16 -> column sums fp32 [N] (include/resel_hip.h `resel_colsum_bf16`); shapes the kernel does not take go to ATen."""
    M, N = x.shape
    if not (x.is_cuda and x.dtype == torch.bfloat16 and x.stride(1) == 1 and N % 8 == 0 and N <= 2048 and x.stride(0) % 8 == 0
            and x.data_ptr() % 16 == 0 and M >= 256):
        return torch.sum(x, 0, dtype=torch.float32)
    out = torch.empty(N, dtype=torch.float32, device=x.device)
    ws = _ws(lib().resel_colsum_bf16_workspace_bytes(M, N), x.device)
    check(lib().resel_colsum_bf16(_p(x), x.stride(0), M, N, _p(out), _p(ws), _stream()), 'colsum_bf16')
    return out


class LinearBf16(torch.autograd.Function):
    """F.linear under the reference's bf16 autocast (flash-attn MHA's Wqkv / out_proj, TransformerFlashAttention.py:67-70) as ONE
    node on `resel_gemm_bf16`: x fp32 or bf16, fp32 master weight and bias; forward output in `out_dtype`; the input gradient comes
    back in x's dtype, the weight / bias gradients in fp32 - the casts of the autocast graph (activation -> bf16, weight -> bf16 per
    call, gradients back to fp32) happen inside the GEMMs."""

    @staticmethod
    def forward(ctx, x, weight, bias, out_dtype, round_out=False):
        ctx.save_for_backward(x, weight)
        ctx.has_bias = bias is not None
        return gemm_bf16(x, weight, True, True, bias, out_dtype, round_out)

    @staticmethod
    def backward(ctx, gy):
        x, weight = ctx.saved_tensors
        gy = gy if gy.stride(-1) == 1 and gy.stride(0) % 4 == 0 and gy.data_ptr() % 16 == 0 else gy.contiguous()
        dx = gemm_bf16(gy, weight, True, False, None, x.dtype) if ctx.needs_input_grad[0] else None
        dw = gemm_bf16(gy, x, False, False, None, torch.float32) if ctx.needs_input_grad[1] else None
        db = None
        if ctx.has_bias and ctx.needs_input_grad[2]:
            if gy.dtype == torch.float32 and gy.shape[1] % 4 == 0 and gy.shape[0] >= 256:
                db = bias_act_bwd(gy, gy, gy.shape[0], None, True)[1].reshape(-1)          # fp32 column sums (one read of gy, fixed order)
            else:
                db = colsum_bf16(gy)
        return dx, dw, db, None, None


def linear_bf16(x, weight, bias, out_dtype=torch.bfloat16, round_out=False):
    return LinearBf16.apply(x, weight, bias, out_dtype, round_out)


# product formation of resel_gemm_f32 (include/resel_hip.h): 0 fp32 MFMA; 9 / 6 exact three-way bf16 split on the bf16 MFMA; 2 fp16 planes
# of the scaled operands, three products (all of these fp32-accurate against fp64, tests/test_hip_ops.py); 3 two bf16 planes per operand
# ("bf16x3").  None (no RESEL_GEMM_SPLIT): follow torch.get_float32_matmul_precision() the way torch's own GEMMs do - 'highest' (torch's
# default, the reference's setting) = 2 where the operand magnitudes are at hand and 6 elsewhere, 'high' / 'medium' = 3.
GEMM_SPLIT = int(os.environ['RESEL_GEMM_SPLIT']) if os.environ.get('RESEL_GEMM_SPLIT') else None


def gemm_split():
    if GEMM_SPLIT is not None:
        return GEMM_SPLIT
    return 2 if torch.get_float32_matmul_precision() == 'highest' else 3



_GEMM_WS_BYTES = {}           # (M, N, K, batch) -> workspace bytes of resel_gemm_f32 (a pure function of the shape)


# ---- operand magnitudes for GEMM mode 2 (include/resel_hip.h "magnitude slots") ---------------------------------------------
# A tensor written by one of this library's kernels carries a HANDLE to max |x| on the device - `t._resel_amax = (handle, version)` -
# so that a later `gemm_f32` on it (or on a view of it: `_base`) needs no extra pass; handles are 8-byte slots of a per-device
# arena ({float bits | epoch}; never zeroed: every producer call takes a fresh epoch) or the float32 [1] result of `amax()`.
# Tags die with an in-place modification that torch sees (`_version`); this module's own in-place kernels re-tag or clear.
AMAX_SLOTS = 2048             # handles per device arena (1 KiB each: eight 8-byte words, 128 bytes apart)
AMAX_WORDS = 128              # int64 words per handle
AMAX_PREPASS_FRACTION = 0.16  # of the estimated GEMM time one may spend on reading an untagged operand (measured: profiles/r04_gemm.md)
_AMAX_ARENA = {}              # device -> [int64 tensor [AMAX_SLOTS], next index]
_AMAX_EPOCH = [0]
LAST_AMAX = None              # handle of the magnitude published by the most recent producer call (wrappers tag Function outputs with it)
LAST_SPLIT = [None]


_AMAX_SERIAL = [0]            # handles handed out so far; every arena handle object carries the serial of its current tenant (`_resel_serial`)
AMAX_GENERATION = [0]         # bumped by `amax_maintenance` when the epoch counter starts over: captured updates hold epochs of the old generation
_EPOCH_RESET_AT = 0x60000000  # epochs are 31-bit device words; start over at an update boundary long before they run out
_STORES = None                # weak set of FlatParameterStores (their weight handles carry epochs too); filled by `register_store`


def register_store(store):
    global _STORES
    if _STORES is None:
        import weakref
        _STORES = weakref.WeakSet()
    _STORES.add(store)


def amax_slot(device):
    """A fresh (handle: int64 view [AMAX_WORDS], epoch) pair for a producer kernel."""
    _AMAX_SERIAL[0] += 1
    ar = _AMAX_ARENA.get(device)
    if ar is None:
        buf = torch.zeros(AMAX_SLOTS * AMAX_WORDS, dtype=torch.int64, device=device)
        ar = _AMAX_ARENA[device] = [buf, 0, list(buf.view(AMAX_SLOTS, AMAX_WORDS).unbind(0))]     # handle views made once (this sits on the launch path)
    if _AMAX_EPOCH[0] >= _EPOCH_RESET_AT and not torch.cuda.is_current_stream_capturing():
        # a long-lived user of this module that never reaches an update boundary (evaluation / inference loops, soak tools): start the
        # epochs over HERE, long before the 31-bit device word runs out, instead of failing after weeks of uptime.  Safe between two
        # producer calls: the reset synchronises the device first (see amax_maintenance), so no kernel with an old epoch is in flight.
        # Handles the caller still holds become void (`handle_alive` / `amax_of` see the changed tenant serial).
        amax_maintenance(force=True)
    if _AMAX_EPOCH[0] >= 0x7ffffff0:
        raise RuntimeError('RESeL-HIP: magnitude epochs exhausted inside one captured region')
    _AMAX_EPOCH[0] += 1
    i = ar[1]
    ar[1] = (i + 1) % AMAX_SLOTS
    h = ar[2][i]
    h._resel_serial = _AMAX_SERIAL[0]               # whoever kept this handle for its previous tenant sees the change (`handle_alive`)
    return h, _AMAX_EPOCH[0]


def amax_maintenance(force=False):
    """Call at an UPDATE BOUNDARY (the trainers do, eager and graphed).  Epochs are 31 bits on the device: long before they run out
    everything that carries one starts over together - arenas zeroed, every arena handle evicted (tags and kept handles become void),
    the weight handles of every FlatParameterStore dropped (re-published on next use), and `AMAX_GENERATION` bumped so that holders of
    captured updates (epochs baked into kernel nodes) drop their graphs.  Returns True when it reset."""
    if not force and _AMAX_EPOCH[0] < _EPOCH_RESET_AT:
        return False
    _AMAX_EPOCH[0] = 0
    AMAX_GENERATION[0] += 1
    PARAM_EPOCH[0] += 1
    for ar in _AMAX_ARENA.values():
        # every stream of the device must be idle before the words are zeroed: a producer still running on a side / target stream with an
        # old-generation epoch (>= 0x60000000) that published AFTER the zero would outrank every new small epoch in its slot, and readers
        # would scale with a stale magnitude.  Once per ~1.6e9 producer calls: the stall does not matter.
        torch.cuda.synchronize(ar[0].device)
        ar[0].zero_()
        for h in ar[2]:
            h._resel_serial = None
    for st in (list(_STORES) if _STORES is not None else []):
        st._amax = None
    return True


def amax_arena_zero(device):
    """Zero the arena of `device` on the launch stream: FIRST node of a captured update - a replay publishes into the slots and
    epochs baked into the graph, which must not inherit the previous replay's maxima."""
    ar = _AMAX_ARENA.get(device)
    if ar is not None:
        ar[0].zero_()


def keep_handles(*hs):
    """For `ctx`: (handle, tenant serial) pairs - saved tensors come back untagged, and between a layer's forward and its backward a
    deep model may hand out more handles than the arena holds, so a kept handle is only used while its slot still has the same tenant."""
    return tuple((h, getattr(h, '_resel_serial', None)) for h in hs)


def handle_alive(kept):
    """The handle of a `keep_handles` pair, or None once the arena has given its slot to another tensor."""
    h, serial = kept
    return h if h is not None and getattr(h, '_resel_serial', None) == serial else None


def live_handles(kept):
    return tuple(handle_alive(k) for k in kept)


def _taggable(t):
    """Tags live on activations and gradients - tensors that are written once and die with the update.  Parameters are NOT tagged:
    this library's optimizer and soft-update kernels rewrite them through the flat buffers without bumping `_version`, so a tag
    would outlive the values it describes (their magnitudes come from a pre-pass per call: a few KB)."""
    return not (isinstance(t, torch.nn.Parameter) or (t.is_leaf and t.requires_grad))


def tag_amax(t, handle, whole=False):
    """whole: t covers ALL of the tensor it is a view of (a reshape of a fresh output) - tag that base too, so that other views of it
    (the [M, C] form of a [B, L, C] activation) find the magnitude.  A tag = (handle, tensor version, tenant serial of the handle)."""
    if t is not None:
        ok = handle is not None and _taggable(t)
        serial = getattr(handle, '_resel_serial', None) if ok else None
        t._resel_amax = (handle, t._version, serial) if ok else None
        base = getattr(t, '_base', None)
        if whole and base is not None:
            base._resel_amax = (handle, base._version, serial) if (ok and _taggable(base)) else None
    return t


def amax_of(t):
    """Handle of a bound on max |t| if one is known (t itself or the tensor t is a view of), else None.  A tag is void once torch has
    seen an in-place write to its tensor, or once the arena has handed the handle's slot to another tensor."""
    for obj in (t, getattr(t, '_base', None)):
        if obj is not None:
            tg = getattr(obj, '_resel_amax', None)
            if tg is not None and tg[1] == obj._version and getattr(tg[0], '_resel_serial', None) == tg[2]:
                return tg[0]
    return None


# ---- verify mode: RESEL_AMAX_VERIFY=1 checks, in front of EVERY mode-2 product, that the handles it scales with really bound its
# operands (one extra pass per operand: slow, for tests and debugging).  Violations are collected in a device word block; the trainers
# read it at the end of each update (`amax_verify_raise`) - also after a replayed update, whose graph then contains the check kernels.
AMAX_VERIFY = os.environ.get('RESEL_AMAX_VERIFY', '0') == '1'
_VERIFY_ERR = {}              # device -> int32 [4] (resel_amax_check)
_VERIFY_CALLS = []            # ring of (tag, description) of the most recent checks
_VERIFY_TAG = [0]


class AmaxBoundError(RuntimeError):
    pass


@torch.no_grad()
def amax_check(x, handle, what=''):
    """Queue a device-side check `max |x| <= value(handle)` (x as in `amax`)."""
    assert x.dtype == torch.float32 and x.stride(-1) == 1 and x.dim() in (2, 3)
    err = _VERIFY_ERR.get(x.device)
    if err is None:
        err = _VERIFY_ERR[x.device] = torch.zeros(4, dtype=torch.int32, device=x.device)
    _VERIFY_TAG[0] += 1
    _VERIFY_CALLS.append((_VERIFY_TAG[0], f'{what} {tuple(x.shape)} strides {tuple(x.stride())}'))
    del _VERIFY_CALLS[:-4096]
    batch = x.shape[0] if x.dim() == 3 else 1
    check(lib().resel_amax_check(_p(x), x.stride(-2), x.stride(0) if x.dim() == 3 and batch > 1 else 0, x.shape[-2], x.shape[-1], batch,
                                 _p(handle), _p(err), _VERIFY_TAG[0], _stream()), 'amax_check')


def amax_verify_raise(device=None):
    """Synchronise and raise AmaxBoundError if any check since the last call found an operand above its handle's bound."""
    for dev, err in list(_VERIFY_ERR.items()):
        if device is not None:
            d = torch.device(device)
            if d.type != dev.type or (d.index is not None and d.index != dev.index):       # 'cuda' names every cuda device
                continue
        e = err.cpu()
        if int(e[0]) == 0:
            continue
        err.zero_()
        tag = int(e[2])
        what = dict(_VERIFY_CALLS).get(tag, '?')
        got = float(e[1:2].view(torch.float32)[0])
        bound = float(e[3:4].view(torch.float32)[0])
        raise AmaxBoundError(f'RESeL-HIP: GEMM mode 2 was given a magnitude handle BELOW its operand: max |x| = {got:.6g} > bound {bound:.6g} '
                             f'(first report: check #{tag}: {what}; {int(e[0])} reporting waves) - the fp16 planes would overflow to inf')


@torch.no_grad()
def amax(x):
    """max |x| of a GEMM operand (2-D, or 3-D with a leading batch axis; last axis contiguous and a multiple of 4) as a magnitude
    handle on the device: one HBM-bound pass, no host synchronisation (include/resel_hip.h `resel_amax`)."""
    _need_cuda('amax', x)
    assert x.dtype == torch.float32 and x.stride(-1) == 1 and x.dim() in (2, 3)
    out, epoch = amax_slot(x.device)
    batch = x.shape[0] if x.dim() == 3 else 1
    # stateless since ABI 7: pre-passes on different streams (the trainer's target / side streams) cannot disturb each other
    check(lib().resel_amax(_p(x), x.stride(-2), x.stride(0) if x.dim() == 3 and batch > 1 else 0, x.shape[-2], x.shape[-1], batch,
                           _p(out), epoch, None, _stream()), 'amax')
    return out


def amax_value(handle):
    """Host value of a magnitude handle (tests / tools: synchronises)."""
    w = handle.view(torch.int64).cpu()[::16][:8]
    ep = (w >> 32) & 0xffffffff
    lo = (w[ep == ep.max()] & 0xffffffff).to(torch.int32)
    return float(lo.view(torch.float32).max())


PARAM_EPOCH = [0]             # bumped by every kernel of this module that rewrites parameters in place (flat AdamW, soft update)


@torch.no_grad()
def amax_segments(flat, begin, length, handles):
    """One launch: max |.| of every segment flat[begin[g] : begin[g] + length[g]] into the g-th handle of `handles` (int64 [nseg * AMAX_WORDS])."""
    _need_cuda('amax_segments', flat, begin, length, handles)
    _AMAX_EPOCH[0] += 1
    check(lib().resel_amax_segments(_p(flat), _p(begin), _p(length), int(begin.numel()), _p(handles), _AMAX_EPOCH[0], _stream()), 'amax_segments')


def weight_amax(p):
    """Magnitude handle of a parameter held in a FlatParameterStore (models/flat_params.py keeps one handle per tensor, refreshed by ONE
    launch after the buffer was rewritten), or None (the caller's GEMM then decides about a pre-pass)."""
    ref = getattr(p, '_resel_store', None)
    if ref is None or not amax_tracking():
        return None
    return ref[0].amax_handle(ref[1], p)


def _slot_args(want, device):
    """(slot tensor or None, pointer, epoch) for a producer kernel's amax output."""
    if not want:
        return None, None, 0
    slot, epoch = amax_slot(device)
    return slot, _p(slot), epoch


def amax_tracking():
    """Producers publish magnitudes only while the GEMMs would use them (product mode 2)."""
    return gemm_split() == 2


@torch.no_grad()
def gemm_f32(A, B, a_kcontig=True, b_kcontig=True, bias=None, act=None, out=None, split=None, amax_a=None, amax_b=None, amax_out=None):
    """C[b] = act(A[b] (.) B[b] + bias[b]) on the matrix cores, fp32 in / out (include/resel_hip.h `resel_gemm_f32`).
    A: [M, K] (a_kcontig) or [K, M]; B: [N, K] (b_kcontig) or [K, N]; optionally a leading batch (ensemble) dimension on all of
    A, B, bias [N] / [batch, N], out.  Row stride free (column stride 1); returns C [M, N] / [batch, M, N].
    (The wrapper is on the host's critical path at small batches: 79 calls per update - no temporary views, cached sizes.)"""
    _need_cuda('gemm_f32', A, B)
    batched = A.dim() == 3
    assert A.stride(-1) == 1 and B.stride(-1) == 1 and A.dtype == torch.float32 and B.dtype == torch.float32
    sa_, sb_ = A.shape, B.shape
    batch = sa_[0] if batched else 1
    M, K = (sa_[-2], sa_[-1]) if a_kcontig else (sa_[-1], sa_[-2])
    N = sb_[-2] if b_kcontig else sb_[-1]
    assert (sb_[-1] if b_kcontig else sb_[-2]) == K and (not batched or sb_[0] == batch)
    if out is None:
        out = torch.empty((batch, M, N) if batched else (M, N), dtype=torch.float32, device=A.device)
    assert out.stride(-1) == 1
    GEMM_FLOPS[0] += 2.0 * M * N * K * batch
    L = lib()
    key = (M, N, K, batch)
    nb = _GEMM_WS_BYTES.get(key)
    if nb is None:
        nb = _GEMM_WS_BYTES[key] = L.resel_gemm_f32_workspace_bytes(M, N, K, batch)
    ws = _ws(nb, A.device) if nb else None
    bs = 0
    if bias is not None:
        bias = bias.reshape(batch, N) if batched else bias.reshape(N)
        bs = bias.stride(0) if batched else 0
    multi = batch > 1
    split = gemm_split() if split is None else int(split)
    global LAST_AMAX
    ha = hb = None
    if split == 2:
        # mode 2 (fp16 planes of the scaled operands) needs a bound on max |A|, max |B| on the device: a producer's tag, the
        # caller's handle, or - when reading the operand once more is cheap next to the GEMM - one resel_amax pass (tagged for reuse)
        if K < 32 or M <= 128:
            split = 6
        else:
            ha = amax_a if amax_a is not None else amax_of(A)
            hb = amax_b if amax_b is not None else amax_of(B)
            ha = weight_amax(A) if ha is None else ha
            hb = weight_amax(B) if hb is None else hb
            if ha is None or hb is None:
                t_gemm = 2.0 * M * N * K * batch / 1.5e8                       # us at 150 TFLOP/s
                cost = (0.0 if ha is not None else 4.0 * M * K * batch / 4.5e6 + 2.5) + (0.0 if hb is not None else 4.0 * N * K * batch / 4.5e6 + 2.5)
                if cost <= AMAX_PREPASS_FRACTION * t_gemm:
                    if ha is None:
                        ha = amax(A)
                        tag_amax(A, ha)
                    if hb is None:
                        hb = amax(B)
                        tag_amax(B, hb)
                else:
                    split = 6
    if AMAX_VERIFY and split == 2:
        amax_check(A, ha, f'A of gemm M={M} N={N} K={K} batch={batch}')
        amax_check(B, hb, f'B of gemm M={M} N={N} K={K} batch={batch}')
    # max |C| for whoever multiplies C next (only while mode 2 is the product mode, and only for outputs worth a pass)
    # (amax_out: the (handle, pointer, epoch) of a row buffer this product fills a column block of - its producers share one handle)
    slot, slot_p, epoch = amax_out if amax_out is not None else \
        _slot_args(amax_tracking() and act != GEMM_ACCUMULATE and M * N * batch >= (1 << 20), A.device)
    check(L.resel_gemm_f32x(_p(A), A.stride(-2), A.stride(0) if multi else 0, int(a_kcontig), _p(B), B.stride(-2),
                            B.stride(0) if multi else 0, int(b_kcontig), _p(bias), bs, 2 if act == GEMM_ACCUMULATE else 3 if act == GEMM_SOFTPLUS else ACT_IDS[act], _p(out), out.stride(-2),
                            out.stride(0) if multi else 0, _p(ws), M, N, K, batch, split, _p(ha) if split == 2 else None,
                            _p(hb) if split == 2 else None, slot_p, epoch, _stream()), 'gemm_f32')
    tag_amax(out, slot)
    LAST_AMAX = slot
    LAST_SPLIT[0] = split                            # tools/gemm_census.py: which product mode the call took
    return out


# ---- fused epilogues of the producer / consumer GEMM (include/resel_hip.h `resel_gemm_f32_dact` / `resel_gemm_f32_head`) ----------
def _forced_handles(A, B, amax_a, amax_b):
    """Magnitude handles of both operands for a product that exists in mode 2 only: the caller's, a producer's tag, the parameter
    store's - or one pre-pass (tagged for reuse)."""
    ha = amax_a if amax_a is not None else amax_of(A)
    hb = amax_b if amax_b is not None else amax_of(B)
    ha = weight_amax(A) if ha is None else ha
    hb = weight_amax(B) if hb is None else hb
    if ha is None:
        ha = amax(A)
        tag_amax(A, ha)
    if hb is None:
        hb = amax(B)
        tag_amax(B, hb)
    return ha, hb


def gemm_fused_ok(kind, M, N, K, *mats):
    """True when a fused-epilogue form (kind 4: dact, 5: head) may take a product of this shape: product mode 2, a long pass, the
    producer / consumer edition's shape rules (K a multiple of 32, M > 128; dact: N a multiple of 128), 16-byte aligned rows."""
    if os.environ.get('RESEL_GEMM_FUSED', '1') == '0' or gemm_split() != 2 or not gemm_f32_ok(M, *mats) or not rows_aligned16(*mats):
        return False
    ld = max(int(t.stride(-2)) for t in mats)
    return bool(lib().resel_gemm_f32_fused_supported(int(kind), int(M), int(N), int(K), ld, ld))


@torch.no_grad()
def gemm_f32_dact(A, B, b_kcontig, Y, out, need_dbias=True, amax_a=None, amax_b=None):
    """out[b] = (A[b] (.) B[b]) * elu'(Y[b]) with elu' taken from the ELU OUTPUT Y (y > 0 ? 1 : y + 1), and dbias[b][n] = column sums of
    out[b]: the input gradient of a layer whose input was the ELU output of the layer below, with that ELU's backward and that layer's
    bias gradient in the GEMM epilogue.  A [batch, M, K] / [M, K] (K contiguous); B [batch, N, K] (b_kcontig) or [batch, K, N]; Y and out
    [batch, M, N] with free row / batch strides.  -> (out, dbias [batch, N] or None)."""
    _need_cuda('gemm_f32_dact', A, B, Y, out)
    batched = A.dim() == 3
    batch = A.shape[0] if batched else 1
    M, K = A.shape[-2], A.shape[-1]
    N = B.shape[-2] if b_kcontig else B.shape[-1]
    assert (B.shape[-1] if b_kcontig else B.shape[-2]) == K and Y.shape[-2:] == (M, N) and out.shape[-2:] == (M, N)
    assert A.stride(-1) == 1 and B.stride(-1) == 1 and Y.stride(-1) == 1 and out.stride(-1) == 1
    ha, hb = _forced_handles(A, B, amax_a, amax_b)
    if AMAX_VERIFY:
        amax_check(A, ha, f'A of gemm_dact M={M} N={N} K={K} batch={batch}')
        amax_check(B, hb, f'B of gemm_dact M={M} N={N} K={K} batch={batch}')
    GEMM_FLOPS[0] += 2.0 * M * N * K * batch
    L = lib()
    ws = _ws(L.resel_gemm_f32_fused_workspace_bytes(M, N, K, batch, 4), A.device)
    dbias = torch.empty(batch, N, dtype=torch.float32, device=A.device) if need_dbias else None
    multi = batched and batch > 1
    slot, slot_p, epoch = _slot_args(amax_tracking() and M * N * batch >= (1 << 20), A.device)
    check(L.resel_gemm_f32_dact(_p(A), A.stride(-2), A.stride(0) if multi else 0, 1, _p(B), B.stride(-2), B.stride(0) if multi else 0, int(b_kcontig),
                                _p(Y), Y.stride(-2), Y.stride(0) if multi else 0, _p(out), out.stride(-2), out.stride(0) if multi else 0,
                                _p(dbias), _p(ws), M, N, K, batch, _p(ha), _p(hb), slot_p, epoch, _stream()), 'gemm_f32_dact')
    tag_amax(out, slot)
    global LAST_AMAX
    LAST_AMAX = slot
    LAST_SPLIT[0] = 2
    return out, dbias


@torch.no_grad()
def gemm_f32_head(A, B, b_kcontig, bias, w3, b3, amax_a=None, amax_b=None):
    """a[b] = elu(A[b] (.) B[b] + bias[b]) and q[b][m] = sum_n a[b][m][n] w3[b][n] + b3[b]: the hidden layer and the width-1 output layer of
    an efc-E critic head in one GEMM.  A [batch, M, K]; B [batch, N, K] / [batch, K, N]; bias, w3 [batch, N]; b3 [batch] or None.
    -> (a [batch, M, N], q [batch, M])."""
    _need_cuda('gemm_f32_head', A, B, bias, w3)
    batch, M, K = A.shape
    N = B.shape[-2] if b_kcontig else B.shape[-1]
    assert A.stride(-1) == 1 and B.stride(-1) == 1 and w3.shape == (batch, N) and w3.is_contiguous() and bias.shape == (batch, N) and bias.stride(-1) == 1
    ha, hb = _forced_handles(A, B, amax_a, amax_b)
    if AMAX_VERIFY:
        amax_check(A, ha, f'A of gemm_head M={M} N={N} K={K} batch={batch}')
        amax_check(B, hb, f'B of gemm_head M={M} N={N} K={K} batch={batch}')
    GEMM_FLOPS[0] += 2.0 * M * N * K * batch
    L = lib()
    ws = _ws(L.resel_gemm_f32_fused_workspace_bytes(M, N, K, batch, 5), A.device)
    a = torch.empty(batch, M, N, dtype=torch.float32, device=A.device)
    q = torch.empty(batch, M, dtype=torch.float32, device=A.device)
    multi = batch > 1
    slot, slot_p, epoch = _slot_args(amax_tracking() and M * N * batch >= (1 << 20), A.device)
    check(L.resel_gemm_f32_head(_p(A), A.stride(-2), A.stride(0) if multi else 0, 1, _p(B), B.stride(-2), B.stride(0) if multi else 0, int(b_kcontig),
                                _p(bias), bias.stride(0) if multi else 0, _p(w3), N, _p(b3), _p(a), a.stride(-2), a.stride(0) if multi else 0, _p(q), _p(ws),
                                M, N, K, batch, _p(ha), _p(hb), slot_p, epoch, _stream()), 'gemm_f32_head')
    tag_amax(a, slot)
    global LAST_AMAX
    LAST_AMAX = slot
    LAST_SPLIT[0] = 2
    return a, q


# ---- one-token rollout step (T = 1, no autograd) ----------------------------------------------------------------------
@torch.no_grad()
def conv_step(x, window, weight, bias, d_conv, layout, act):
    """Depthwise conv on one token with a rolling window.  x [B, Di] (row-strided view), window [B, *] rows holding
    layout 'dk': [Di, K] per row (smamba) or 'kd': time-major [K - 1, Di] (s6 mamba / conv1d).  -> (xc [B, Di], new window)."""
    _need_cuda('conv_step', x, window)
    B, Di, K = x.shape[0], x.shape[1], d_conv
    W, sd, sk = (K, K, 1) if layout == 'dk' else (K - 1, 1, Di)
    assert window.shape == (B, Di * W) and window.stride(1) == 1 and x.stride(1) == 1
    new = torch.empty((B, Di * W), dtype=torch.float32, device=x.device)
    xc = torch.empty((B, Di), dtype=torch.float32, device=x.device)
    check(lib().resel_mamba_conv_step(_p(x), x.stride(0), _p(window), window.stride(0), _p(new), new.stride(0), sd, sk, W,
                                      _p(weight), _p(bias), _p(xc), B, Di, K, int(bool(act)), _stream()), 'mamba_conv_step')
    return xc, new


@torch.no_grad()
def selective_state_update(state, xc, x_db, dt_w, dt_b, A_log, D, z=None):
    """state [B, Di*N] (row-strided view), xc [B, Di], x_db [B, R + 2N] = (dt low-rank | B | C), z [B, Di] view or None.
    -> (y [B, Di], new state [B, Di*N])   (reference selective_state_update.py:123-154 with dt_proj folded in)."""
    _need_cuda('selective_state_update', state, xc, x_db)
    B, Di = xc.shape
    R = dt_w.shape[1]
    N = (x_db.shape[1] - R) // 2
    assert state.shape == (B, Di * N) and state.stride(1) == 1 and x_db.stride(1) == 1 and xc.is_contiguous()
    new = torch.empty((B, Di * N), dtype=torch.float32, device=xc.device)
    y = torch.empty_like(xc)
    check(lib().resel_selective_state_update(_p(state), state.stride(0), _p(new), new.stride(0), _p(xc), _p(x_db), x_db.stride(0),
                                             _p(dt_w), _p(dt_b), _p(A_log), _p(D), _p(z), 0 if z is None else z.stride(0), _p(y),
                                             B, Di, N, R, _stream()), 'selective_state_update')
    return y, new


@torch.no_grad()
def mamba_step(hidden, xz, conv_w, conv_b, xproj_w, dt_w, dt_b, A_log, D, d_conv, d_state):
    """smamba mixer state update for one token (reference smamba/mamba.py:257-305).
    hidden [B, Di*K + Di*N] = (conv window | ssm state), xz [B, 2 Di] = in_proj output.  -> (y [B, Di], new hidden)."""
    Di, K = xz.shape[1] // 2, d_conv
    xc, window = conv_step(xz[:, :Di], hidden[:, :Di * K], conv_w, conv_b, K, 'dk', True)
    x_db = gemm_f32(xc, xproj_w, True, True)                                         # [B, R + 2N]: the rows form of resel_gemm_f32x
    y, state = selective_state_update(hidden[:, Di * K:], xc, x_db, dt_w, dt_b, A_log, D, xz[:, Di:])
    return y, torch.cat((window, state), dim=-1)


@torch.no_grad()
def attn_decode(qkv, kv_cache, pos, slopes, scale):
    """Append this token's k, v to kv_cache [Bmax, S, 2, H, hd] (bf16, in place) and attend over positions 0..pos.
    qkv [B, 3, H, hd] bf16; pos: host int, or an int32 device tensor (graph replay).  -> [B, H, hd] bf16."""
    _need_cuda('attn_decode', qkv, kv_cache)
    B, _, H, hd = qkv.shape
    assert qkv.dtype == torch.bfloat16 and kv_cache.dtype == torch.bfloat16 and qkv.is_contiguous() and kv_cache.is_contiguous()
    assert kv_cache.shape[0] >= B and kv_cache.shape[2:] == (2, H, hd)
    out = torch.empty((B, H, hd), dtype=torch.bfloat16, device=qkv.device)
    dev = pos if torch.is_tensor(pos) else None
    check(lib().resel_attn_decode(_p(qkv), 3 * H * hd, _p(kv_cache), _p(dev), 0 if dev is not None else int(pos), _p(slopes), _p(out),
                                  float(scale), B, H, hd, kv_cache.shape[1], _stream()), 'attn_decode')
    return out


@torch.no_grad()
def soft_update_(target_flat, online_flat, tau):
    _need_cuda('soft_update', target_flat, online_flat)
    PARAM_EPOCH[0] += 1
    check(lib().resel_soft_update(_p(target_flat), _p(online_flat), float(tau), target_flat.numel(), _stream()), 'soft_update')


@torch.no_grad()
def adamw_flat_(p, g, m, v, seg_end, seg_lr, seg_wd, step, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=None):
    _need_cuda('adamw_flat', p, g, m, v, seg_end, seg_lr, seg_wd)
    PARAM_EPOCH[0] += 1
    check(lib().resel_adamw_flat(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(seg_end), _p(seg_lr), _p(seg_wd), int(seg_end.numel()),
                                 float(beta1), float(beta2), float(eps), int(step), _p(grad_scale), _stream()), 'adamw_flat')


@torch.no_grad()
def adamw_flat_dev_(p, g, m, v, seg_end, seg_lr, seg_wd, bias_corrections, beta1=0.9, beta2=0.999, eps=1e-8, grad_scale=None):
    """`adamw_flat_` with (1 - beta1^t, sqrt(1 - beta2^t)) read from the 2-element device tensor `bias_corrections` (captured updates)."""
    _need_cuda('adamw_flat_dev', p, g, m, v, seg_end, seg_lr, seg_wd, bias_corrections)
    PARAM_EPOCH[0] += 1
    check(lib().resel_adamw_flat_dev(_p(p), _p(g), _p(m), _p(v), p.numel(), _p(seg_end), _p(seg_lr), _p(seg_wd), int(seg_end.numel()),
                                     float(beta1), float(beta2), float(eps), _p(bias_corrections), _p(grad_scale), _stream()), 'adamw_flat_dev')


@torch.no_grad()
def sumsq(x, out=None):
    _need_cuda('sumsq', x)
    out = torch.empty(1, dtype=torch.float32, device=x.device) if out is None else out
    ws = _ws(lib().resel_sumsq_workspace_bytes(x.numel()), x.device)
    check(lib().resel_sumsq(_p(x), x.numel(), _p(out), _p(ws), _stream()), 'sumsq')
    return out
