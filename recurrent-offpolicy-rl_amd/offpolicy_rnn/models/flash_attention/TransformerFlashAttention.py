"""`cgpt` layer - pre-norm causal Transformer decoder with ALiBi over packed variable-length sequences
(reference offpolicy_rnn/models/flash_attention/TransformerFlashAttention.py:64-121).

The reference delegates attention to the un-vendored `flash_attn.modules.mha.MHA` under bf16 autocast; here the same
block structure runs on `ops.attn_varlen` (hand-written bf16 MFMA kernels, fp32 softmax).  Parameter names follow the
reference / flash-attn (`decoder_layers.{i}.mha.Wqkv.{weight,bias}`, `mha.out_proj`, `ffn.fc1/fc2`, `mha_norm`,
`ffn_norm`, `output_ln`, `output_fc`).  Semantics restated, parity unpinned (see oracle/kernels.py).
Deviation: no dropout on the attention probabilities (residual / FFN dropout is applied); published cgpt runs use p = 0.0.
T == 1 rollout decoding with a KV cache is outside the training hot path and not provided."""
import math
from dataclasses import dataclass, field
from typing import Optional

import torch
import torch.nn as nn
import torch.nn.functional as F

from ...hip import ops


@dataclass
class InferenceParams:
    """Rollout-time KV-cache handle of the reference (TransformerFlashAttention.py:13-27); a placeholder here."""
    max_seqlen: int
    max_batch_size: int
    seqlen_offset: int = 0
    batch_size_offset: int = 0
    key_value_memory_dict: dict = field(default_factory=dict)
    lengths_per_sample: Optional[torch.Tensor] = None


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


class TransformerDecoder(nn.Module):
    def __init__(self, d_model, n_head, d_ff, n_layer, dropout=0.1, ln=True):
        super().__init__()
        self.d_model, self.n_head, self.d_ff, self.n_layer = d_model, n_head, d_ff, n_layer
        self.decoder_layers = nn.ModuleList([DecoderLayer(d_model, n_head, d_ff, dropout=dropout, layer_idx=i, ln=ln) for i in range(n_layer)])
        self.output_ln = nn.LayerNorm(d_model) if ln else RMSNorm(d_model)
        self.output_fc = nn.Linear(d_model, d_model)

    def forward(self, x, inference_params=None, seqlens=None):
        if inference_params is not None or x.shape[-2] == 1:
            raise NotImplementedError('cgpt single-step decoding with a KV cache (rollouts) is outside the training hot path of this build')
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
