"""`cgpt` layer - pre-norm causal Transformer decoder with ALiBi over packed variable-length sequences
(reference offpolicy_rnn/models/flash_attention/TransformerFlashAttention.py:64-121).

The reference delegates attention to the un-vendored `flash_attn.modules.mha.MHA` under bf16 autocast; here the same
block structure runs on `ops.attn_varlen` (hand-written bf16 MFMA kernels, fp32 softmax).  Parameter names follow the
reference / flash-attn (`decoder_layers.{i}.mha.Wqkv.{weight,bias}`, `mha.out_proj`, `ffn.fc1/fc2`, `mha_norm`,
`ffn_norm`, `output_ln`, `output_fc`).  Semantics restated, parity unpinned (see oracle/kernels.py).
Deviation: no dropout on the attention probabilities (residual / FFN dropout is applied); published cgpt runs use p = 0.0.
T == 1 rollout steps append to a per-layer bf16 KV cache held by `InferenceParams` and attend with `ops.attn_decode`."""
import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import ops


@dataclass
class InferenceParams:
    """Rollout-time KV-cache handle of the reference (TransformerFlashAttention.py:13-27).  `key_value_memory_dict[layer]`
    is a bf16 cache [max_batch_size, max_seqlen, 2, H, hd], allocated on first use as flash-attn's MHA does; the caller
    (RNNBase.meta_forward, reference rnn_base.py:451-452) advances `seqlen_offset` after every step.
    `device_offset` (int32 [1] on the device) is this build's addition for hipGraph replay: when set, the kernels take the
    position from it instead of the host integer and the decoder advances it on the stream."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: Optional[torch.Tensor] = None
    device_offset: Optional[torch.Tensor] = None

    def reset(self, max_seqlen, max_batch_size):
        self.max_seqlen, self.max_batch_size, self.seqlen_offset = max_seqlen, max_batch_size, 0
        if self.lengths_per_sample is not None:
            self.lengths_per_sample.zero_()
        if self.device_offset is not None:
            self.device_offset.zero_()

    def __deepcopy__(self, memo):
        out = InferenceParams(self.max_seqlen, self.max_batch_size, self.seqlen_offset, self.batch_size_offset)
        out.key_value_memory_dict = {k: v.clone() for k, v in self.key_value_memory_dict.items()}
        out.lengths_per_sample = None if self.lengths_per_sample is None else self.lengths_per_sample.clone()
        out.device_offset = None if self.device_offset is None else self.device_offset.clone()
        return out


class PackedSeqs:
    """Host-built description of the sequences packed into the rows of a batch: the per-row length table of the
    reference (`attention_concat_mask`, sac_full_length_rnn_ensembleQ.py:358-366) plus the token indices / cu_seqlens
    that flash-attn's `unpad_input_for_concatenated_sequences` would derive from it on the device."""

    def __init__(self, table, row_len: int, device):
        import numpy as np
        table = np.asarray(table).astype(np.int64)
        idx, cu, mx = [], [0], 1
        for b in range(table.shape[0]):
            pos = 0
            for n in table[b]:
                n = int(n)
                if n <= 0:
                    continue
                idx.append(np.arange(b * row_len + pos, b * row_len + pos + n))
                cu.append(cu[-1] + n)
                mx = max(mx, n)
                pos += n
        self.indices = torch.from_numpy(np.concatenate(idx) if idx else np.zeros(0, dtype=np.int64)).to(device)
        self.cu_seqlens = torch.tensor(cu, dtype=torch.int32, device=device)
        self.max_seqlen = mx
        self.table = table


def alibi_slopes(nheads: int) -> torch.Tensor:
    def pow2(n):
        start = 2.0 ** (-(2.0 ** -(math.log2(n) - 3)))
        return [start * (start ** i) for i in range(n)]
    if math.log2(nheads).is_integer():
        return torch.tensor(pow2(nheads), dtype=torch.float32)
    c = 2 ** math.floor(math.log2(nheads))
    return torch.tensor(pow2(c) + pow2(2 * c)[0::2][: nheads - c], dtype=torch.float32)


class RMSNorm(nn.Module):
    def __init__(self, d_model: int, eps: float = 1e-5):
        super().__init__()
        self.eps = eps
        self.weight = nn.Parameter(torch.ones(d_model))

    def forward(self, x):
        return x * torch.rsqrt(x.pow(2).mean(-1, keepdim=True) + self.eps) * self.weight


def _norm(mod, x, residual, prenorm):
    fn = ops.rms_norm_fn if isinstance(mod, RMSNorm) else ops.layer_norm_fn
    return fn(x, mod.weight, getattr(mod, 'bias', None), residual=residual, eps=mod.eps, prenorm=prenorm)


class MHA(nn.Module):
    """Packed-QKV multi-head attention, causal + ALiBi, computed in bf16 (parameters stay fp32)."""

    def __init__(self, embed_dim, num_heads):
        super().__init__()
        assert embed_dim % num_heads == 0
        self.num_heads, self.head_dim = num_heads, embed_dim // num_heads
        self.Wqkv = nn.Linear(embed_dim, 3 * embed_dim)
        self.out_proj = nn.Linear(embed_dim, embed_dim)
        self._slopes = None

    def forward(self, x, cu_seqlens, max_seqlen):
        if self._slopes is None or self._slopes.device != x.device:
            self._slopes = alibi_slopes(self.num_heads).to(x.device)
        bf = torch.bfloat16
        qkv = F.linear(x.to(bf), self.Wqkv.weight.to(bf), self.Wqkv.bias.to(bf))
        ctx = ops.attn_varlen(qkv.view(-1, 3, self.num_heads, self.head_dim), cu_seqlens, max_seqlen, self._slopes,
                              self.head_dim ** -0.5)
        return F.linear(ctx.reshape(-1, self.num_heads * self.head_dim), self.out_proj.weight.to(bf), self.out_proj.bias.to(bf))

    def step(self, x, inference_params: InferenceParams, layer_idx: int):
        """One rollout token per row, x [B, D]: k, v go to the layer's cache at the current position, the query attends
        over the cache (flash-attn MHA._update_kvcache_attention under the reference's bf16 autocast)."""
        if self._slopes is None or self._slopes.device != x.device:
            self._slopes = alibi_slopes(self.num_heads).to(x.device)
        ip, bf, B = inference_params, torch.bfloat16, x.shape[0]
        cache = ip.key_value_memory_dict.get(layer_idx)
        if cache is None:
            cache = torch.zeros((ip.max_batch_size, ip.max_seqlen, 2, self.num_heads, self.head_dim), dtype=bf, device=x.device)
            ip.key_value_memory_dict[layer_idx] = cache
        if B > cache.shape[0]:
            raise RuntimeError(f'cgpt rollout: batch {B} exceeds the KV cache batch {cache.shape[0]}')
        qkv = F.linear(x.to(bf), self.Wqkv.weight.to(bf), self.Wqkv.bias.to(bf))
        pos = ip.device_offset if ip.device_offset is not None else ip.seqlen_offset
        ctx = ops.attn_decode(qkv.view(B, 3, self.num_heads, self.head_dim), cache, pos, self._slopes, self.head_dim ** -0.5)
        return F.linear(ctx.reshape(B, -1), self.out_proj.weight.to(bf), self.out_proj.bias.to(bf))


class PositionWiseFeedForward(nn.Module):
    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.fc1 = nn.Linear(d_model, d_ff)
        self.fc2 = nn.Linear(d_ff, d_model)
        self.act = nn.GELU()
        self.dropout = nn.Dropout(dropout)

    def forward(self, x):
        return self.fc2(self.dropout(self.act(self.fc1(x))))


class DecoderLayer(nn.Module):
    def __init__(self, d_model, nhead, d_ff, dropout=0.1, layer_idx=None, ln=True):
        super().__init__()
        self.mha = MHA(d_model, nhead)
        self.layer_idx = layer_idx
        self.ffn = PositionWiseFeedForward(d_model, d_ff, dropout)
        self.dropout = nn.Dropout(dropout)
        self.mha_norm = nn.LayerNorm(d_model) if ln else RMSNorm(d_model)
        self.ffn_norm = nn.LayerNorm(d_model) if ln else RMSNorm(d_model)

    def forward(self, h, residual, cu_seqlens, max_seqlen):
        """Pre-norm block on a (branch output h, running residual) pair: every `x = branch + x` of the reference block
        (TransformerFlashAttention.py:76-95) is folded into the following norm (fused add+norm kernel, fp32 residual)."""
        normed, residual = _norm(self.mha_norm, h, residual, prenorm=True)
        a = self.mha(normed, cu_seqlens, max_seqlen).to(torch.float32)
        normed, residual = _norm(self.ffn_norm, self.dropout(a), residual, prenorm=True)
        return self.dropout(self.ffn(normed)), residual

    def step(self, h, residual, inference_params):
        normed, residual = _norm(self.mha_norm, h, residual, prenorm=True)
        a = self.mha.step(normed, inference_params, self.layer_idx).to(torch.float32)
        normed, residual = _norm(self.ffn_norm, self.dropout(a), residual, prenorm=True)
        return self.dropout(self.ffn(normed)), residual


class TransformerDecoder(nn.Module):
    def __init__(self, d_model, n_head, d_ff, n_layer, dropout=0.1, ln=True):
        super().__init__()
        self.d_model, self.n_head, self.d_ff, self.n_layer = d_model, n_head, d_ff, n_layer
        self.decoder_layers = nn.ModuleList([DecoderLayer(d_model, n_head, d_ff, dropout=dropout, layer_idx=i, ln=ln) for i in range(n_layer)])
        self.output_ln = nn.LayerNorm(d_model) if ln else RMSNorm(d_model)
        self.output_fc = nn.Linear(d_model, d_model)

    def forward(self, x, inference_params=None, seqlens=None):
        if inference_params is not None:
            return self._step(x, inference_params)
        batch, row_len, dim = x.shape
        if seqlens is None:
            packed = PackedSeqs([[row_len]] * batch, row_len, x.device)
        elif isinstance(seqlens, PackedSeqs):
            packed = seqlens
        else:                                   # raw per-row length table on the device: needs one device->host copy
            packed = PackedSeqs(seqlens.detach().cpu().numpy(), row_len, x.device)
        flat = x.reshape(batch * row_len, dim)
        t = flat.index_select(0, packed.indices)
        residual = None
        for layer in self.decoder_layers:
            t, residual = layer(t, residual, packed.cu_seqlens, packed.max_seqlen)
        t = self.output_fc(_norm(self.output_ln, t, residual, prenorm=False))
        out = torch.zeros_like(flat).index_copy(0, packed.indices, t)
        return out.view(batch, row_len, dim)

    def _step(self, x, ip: InferenceParams):
        """Rollout step x [B, 1, D] against the KV caches in `ip` (reference :104-121 with inference_params; the caller
        advances ip.seqlen_offset).  A full cache raises, as flash-attn's cache update asserts."""
        if x.dim() != 3 or x.shape[-2] != 1:
            raise NotImplementedError('cgpt rollouts decode one token per call (reference rnn_base.py:437-452 only passes T == 1)')
        if ip.seqlen_offset >= ip.max_seqlen:
            raise RuntimeError(f'cgpt rollout: KV cache is full ({ip.seqlen_offset} tokens, max_seqlen {ip.max_seqlen}); '
                               f'raise `ml` in the layer id or reset the hidden state')
        B, _, dim = x.shape
        t, residual = x.reshape(B, dim), None
        for layer in self.decoder_layers:
            t, residual = layer.step(t, residual, ip)
        t = self.output_fc(_norm(self.output_ln, t, residual, prenorm=False))
        if ip.device_offset is not None:
            ip.device_offset.add_(1)
        return t.view(B, 1, dim)
