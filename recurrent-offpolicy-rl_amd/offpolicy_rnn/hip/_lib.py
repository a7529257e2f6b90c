"""ctypes binding of the C-ABI library `libresel_hip.so` (include/resel_hip.h).

The product path has NO CPU fallback: every op in `ops.py` goes through `lib()`, which raises a loud
RuntimeError when the shared library is missing or was built against another ABI.
"""
import ctypes
import os
from ctypes import c_float, c_int, c_int64, c_size_t, c_uint64, c_void_p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get('RESEL_HIP_LIBRARY') or os.path.join(_HERE, 'libresel_hip.so')      # override: ablation builds (tools/gemm_ablate.sh)
ABI_VERSION = 8
_lib = None

P, I, L, F, S, U = c_void_p, c_int, c_int64, c_float, c_void_p, c_uint64
E = ctypes.c_uint                # epoch of a magnitude slot

# name -> (restype, argtypes); order and types mirror include/resel_hip.h
SIGNATURES = {
    'resel_abi_version': (c_int, []),
    'resel_build_info': (ctypes.c_char_p, []),
    'resel_profile_enable': (c_int, [I]),
    'resel_profile_collect': (c_int, [I, P, P]),
    'resel_selective_scan_ckpt_bytes': (c_size_t, [I, I, I, I]),
    'resel_selective_scan_fwd_workspace_bytes': (c_size_t, [I, I, I, I, I]),
    'resel_selective_scan_fwd': (c_int, [P, L, P, L, P, L, P, P, L, P, L, P, P, P, P, L, P, P, P, I, I, I, I, I, I, P, E, S]),
    'resel_selective_scan_bwd_workspace_bytes': (c_size_t, [I, I, I, I, I]),
    'resel_selective_scan_bwd': (c_int, [P, L, P, L, P, L, P, P, L, P, L, P, P, P, P, L, P,
                                         P, L, P, L, P, L, P, L, P, L, P, P, P, P, I, I, I, I, I, I, P, P, E, S]),
    'resel_causal_conv1d_fwd': (c_int, [P, L, P, P, P, P, L, I, I, I, I, I, P, E, S]),
    'resel_causal_conv1d_bwd_workspace_bytes': (c_size_t, [I, I, I, I]),
    'resel_causal_conv1d_bwd': (c_int, [P, L, P, P, P, P, L, P, L, P, P, P, I, I, I, I, I, P, E, S]),
    'resel_causal_conv1d_bwd2': (c_int, [P, L, P, P, P, P, L, P, L, P, L, P, P, P, I, I, I, I, I, P, E, S]),
    'resel_add_layernorm_fwd': (c_int, [P, P, P, P, P, P, P, I, I, F, I, I, P, E, S]),
    'resel_add_layernorm_bwd_workspace_bytes': (c_size_t, [I, I]),
    'resel_add_layernorm_bwd': (c_int, [P, P, P, P, P, P, P, P, P, P, I, I, I, I, I, P, E, S]),
    'resel_linrec_real_fwd': (c_int, [P, P, L, P, P, P, I, I, I, I, P, E, S]),
    'resel_linrec_real_bwd': (c_int, [P, P, L, P, P, P, P, P, P, L, I, I, I, I, P, E, S]),
    'resel_linrec_complex_fwd': (c_int, [P, P, L, P, P, P, P, P, P, P, P, I, I, I, P, E, S]),
    'resel_linrec_complex_bwd_workspace_bytes': (c_size_t, [I, I, I]),
    'resel_place_blocks': (c_int, [P, L, I, I, I, P, P, P, P, P, P, S]),
    'resel_lru_params_fwd': (c_int, [P, P, I, S]),
    'resel_lru_params_bwd': (c_int, [P, P, P, I, S]),
    'resel_linrec_complex_bwd': (c_int, [P, P, L, P, P, P, P, P, P, P, P, P, P, P, P, L, P, P, P, P, I, I, I, S]),
    'resel_gru_workspace_bytes': (c_size_t, [I, I, I]),
    'resel_gru_seq_fwd': (c_int, [P, P, P, P, P, P, P, I, I, I, S]),
    'resel_gru_seq_bwd': (c_int, [P, P, P, P, P, P, P, P, I, I, I, S]),
    'resel_attn_varlen_fwd_workspace_bytes': (c_size_t, [I, I]),
    'resel_attn_varlen_fwd': (c_int, [P, P, P, P, P, P, I, I, I, I, I, F, F, U, U, S]),
    'resel_attn_varlen_bwd_workspace_bytes': (c_size_t, [I, I, I, I, I]),
    'resel_attn_varlen_bwd': (c_int, [P, P, P, P, P, P, P, P, I, I, I, I, I, F, F, U, U, S]),
    'resel_dropout': (c_int, [P, P, L, F, U, U, S]),
    'resel_dropout_offset_base': (c_int, [P]),
    'resel_gelu_dropout_fwd': (c_int, [P, P, L, F, U, U, P, E, S]),
    'resel_gelu_dropout_bwd': (c_int, [P, P, P, L, F, U, U, P, E, S]),
    'resel_tanh_gaussian_fwd': (c_int, [P, P, P, P, P, I, I, S]),
    'resel_tanh_gaussian_bwd': (c_int, [P, P, P, P, P, I, I, S]),
    'resel_sac_target': (c_int, [P, P, I, P, P, P, P, P, F, P, P, P, P, I, I, S]),
    'resel_sac_target_phase': (c_int, [I, P, P, I, P, P, P, P, P, F, P, P, P, P, P, I, I, S]),
    'resel_masked_loss_workspace_bytes': (c_size_t, []),
    'resel_q_loss_fwd': (c_int, [P, P, P, P, P, I, I, S]),
    'resel_q_loss_bwd': (c_int, [P, P, P, P, P, I, I, S]),
    'resel_actor_loss_fwd': (c_int, [P, P, P, P, P, P, I, I, I, I, S]),
    'resel_actor_loss_bwd': (c_int, [P, P, P, P, P, P, I, I, I, I, S]),
    'resel_sac_target_local': (c_int, [P, P, I, P, P, P, P, P, F, P, P, P, P, P, I, I, S]),
    'resel_guard_apply_slots': (c_int, [P, I, P, S]),
    'resel_sac_target_workspace_bytes': (c_size_t, [I]),
    'resel_soft_update': (c_int, [P, P, F, L, S]),
    'resel_adamw_flat': (c_int, [P, P, P, P, L, P, P, P, I, F, F, F, I, P, S]),
    'resel_adamw_flat_dev': (c_int, [P, P, P, P, L, P, P, P, I, F, F, F, P, P, S]),
    'resel_sumsq_workspace_bytes': (c_size_t, [L]),
    'resel_sumsq': (c_int, [P, L, P, P, S]),
    'resel_bias_act_fwd': (c_int, [P, P, L, I, L, I, S]),
    'resel_bias_act_bwd_workspace_bytes': (c_size_t, [L, I, L]),
    'resel_bias_act_bwd': (c_int, [P, L, P, L, P, P, P, L, I, L, I, P, E, S]),
    'resel_ensemble_head_fwd': (c_int, [P, P, P, P, P, L, I, L, S]),
    'resel_ensemble_head_bwd_workspace_bytes': (c_size_t, [L, I, L]),
    'resel_ensemble_head_bwd': (c_int, [P, P, P, P, P, P, P, L, I, L, P, E, S]),
    'resel_gemm_f32_workspace_bytes': (c_size_t, [I, I, I, I]),
    'resel_gemm_f32': (c_int, [P, L, L, I, P, L, L, I, P, L, I, P, L, L, P, I, I, I, I, I, S]),
    'resel_gemm_f32x': (c_int, [P, L, L, I, P, L, L, I, P, L, I, P, L, L, P, I, I, I, I, I, P, P, P, E, S]),
    'resel_amax_segments': (c_int, [P, P, P, I, P, E, S]),
    'resel_amax_state_bytes': (c_size_t, []),
    'resel_amax': (c_int, [P, L, L, I, I, I, P, E, P, S]),
    'resel_gemm_f32_fused_supported': (c_int, [I, I, I, I, L, L]),
    'resel_gemm_f32_fused_workspace_bytes': (c_size_t, [I, I, I, I, I]),
    'resel_gemm_f32_dact': (c_int, [P, L, L, I, P, L, L, I, P, L, L, P, L, L, P, P, I, I, I, I, P, P, P, E, S]),
    'resel_gemm_f32_head': (c_int, [P, L, L, I, P, L, L, I, P, L, P, L, P, P, L, L, P, P, I, I, I, I, P, P, P, E, S]),
    'resel_amax_check': (c_int, [P, L, L, I, I, I, P, P, I, S]),
    'resel_gemm_bf16_workspace_bytes': (c_size_t, [I, I, I]),
    'resel_gemm_bf16': (c_int, [P, L, I, I, P, L, I, I, P, P, L, I, P, I, I, I, S]),
    'resel_colsum_bf16_workspace_bytes': (c_size_t, [I, I]),
    'resel_colsum_bf16': (c_int, [P, L, I, I, P, P, S]),
    'resel_gather_trajs': (c_int, [P, I, L, P, I, I, I, I, I, I, I, I, I, P, I, P, S]),
    'resel_mamba_conv_step': (c_int, [P, L, P, L, P, L, L, L, I, P, P, P, I, I, I, I, S]),
    'resel_selective_state_update': (c_int, [P, L, P, L, P, P, L, P, P, P, P, P, L, P, I, I, I, I, S]),
    'resel_atb_workspace_bytes': (c_size_t, [L, I, I]),
    'resel_atb': (c_int, [P, L, I, P, L, I, P, I, P, L, S]),
    'resel_attn_decode': (c_int, [P, L, P, P, I, P, P, F, I, I, I, I, S]),
}


class ReselHipUnavailable(RuntimeError):
    pass


def lib():
    """Load (once) and return the C-ABI library; fail loudly if it is missing or stale."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise ReselHipUnavailable(
            f'RESeL-HIP: {LIB_PATH} not found. The MI355X kernels are the only implementation of the hot path; '
            f'build them with `python -c "import __graft_entry__ as g; g.build()"` (or `make -C recurrent-offpolicy-rl_amd/csrc`).')
    handle = ctypes.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(handle, name, None)
        if fn is None:
            raise ReselHipUnavailable(f'RESeL-HIP: symbol {name} missing from {LIB_PATH} (stale build?)')
        fn.restype = res
        fn.argtypes = args
    if handle.resel_abi_version() != ABI_VERSION:
        raise ReselHipUnavailable(f'RESeL-HIP: ABI {handle.resel_abi_version()} != expected {ABI_VERSION}; rebuild the library')
    _lib = handle
    return _lib


def check(rc: int, what: str):
    if rc != 0:
        raise RuntimeError(f'RESeL-HIP: {what} failed with code {rc} '
                           f'({ {-1: "RESEL_EINVAL (shape/alignment/null)", -2: "RESEL_ELAUNCH"}.get(rc, "?")})')
