"""RNNBase: a stack of layers selected by layer-id strings, with a per-layer (norm, activation) tail
(reference offpolicy_rnn/models/rnn_base.py:31-532).

Layer-id grammar kept from the reference (:101-247): `fc`, `efc-<E>`, `gru`, `gilr`, `lru`, `gilr_lstm`,
`smamba[_s<N>][_c<K>][_b<blocks>][_n<ln|...>][_ff]`, `mamba[_s<N>][_c<K>][_noff]`, `conv1d[_<K>]`,
`cgpt[_h<H>][_l<L>][_p<drop>][_ml<M>][_rms]`.  Ids of reference layers that are outside this build (`lstm`, `e<rnn>-<E>`,
`econv1d*`, `gpt*`, `transformer*`, `cgru`) are recognised and rejected with an explicit message.  Module / parameter naming (`layer_list.<i>.…`, `activation_list.<i>.0.…`) matches the
reference so that its per-module checkpoints load."""
import copy
import os
from typing import List, Optional, Tuple, Union

import torch

from ..hip import ops
from .RNNHidden import RNNHidden
from .ensemble_linear_model import EnsembleLinear, critic_mlp, critic_mlp_fusable, ensemble_head, head_fusable
from .flash_attention.TransformerFlashAttention import InferenceParams, TransformerDecoder
from .conv1d.conv1d import Conv1d
from .gilr.gilr import GILRLayer
from .gilr_lstm.gilr_lstm import GILRLSTMLayer
from .gru import GRU
from .s6.mamba import MambaResidualBlock
from .lru.lru import LRULayer
from .smamba.mamba import BlockList as MambaBlockList
from .linear import Linear

ACTIVATIONS = {'tanh': torch.nn.Tanh, 'relu': torch.nn.ReLU, 'sigmoid': torch.nn.Sigmoid, 'leaky_relu': torch.nn.LeakyReLU,
               'linear': torch.nn.Identity, 'elu': torch.nn.ELU, 'gelu': torch.nn.GELU}
_UNSUPPORTED_PREFIXES = ('lstm', 'econv1d', 'gpt', 'transformer', 'cgru', 'elru', 'egilr')


def parse_smamba_id(layer_id: str) -> dict:
    cfg = dict(d_conv=4, d_state=16, block_num=2, rms_norm=True, use_ff=False)
    for tok in layer_id.split('_')[1:]:
        if tok.startswith('s'):
            cfg['d_state'] = int(tok[1:])
        elif tok.startswith('c'):
            cfg['d_conv'] = int(tok[1:])
        elif tok.startswith('b'):
            cfg['block_num'] = int(tok[1:])
        elif tok.startswith('n'):
            cfg['rms_norm'] = tok[1:] != 'ln'
        elif tok.startswith('f'):
            cfg['use_ff'] = cfg['use_ff'] or tok[1:] == 'f'
        else:
            raise ValueError(f'Pattern {tok} has not been implemented!')
    return cfg


def parse_mamba_id(layer_id: str) -> dict:
    """`mamba[_s<N>][_c<K>][_noff]` (reference rnn_base.py:118-135)."""
    cfg = dict(d_conv=4, d_state=16, use_ff=True)
    for tok in layer_id.split('_')[1:]:
        if tok.startswith('s'):
            cfg['d_state'] = int(tok[1:])
        elif tok.startswith('c'):
            cfg['d_conv'] = int(tok[1:])
        elif tok.startswith('no'):
            if tok[2:] == 'ff':
                cfg['use_ff'] = False
        else:
            raise ValueError(f'Pattern {tok} has not been implemented!')
    return cfg


def parse_cgpt_id(layer_id: str) -> dict:
    cfg = dict(nhead=8, nlayer=4, pdrop=0.1, maxlength=1024, ln=True)
    for tok in layer_id.split('_')[1:]:
        if tok.startswith('h'):
            cfg['nhead'] = int(tok[1:])
        elif tok.startswith('l'):
            cfg['nlayer'] = int(tok[1:])
        elif tok.startswith('p'):
            cfg['pdrop'] = float(tok[1:])
        elif tok.startswith('ml'):
            cfg['maxlength'] = int(tok[2:])
        elif tok.startswith('rms'):
            cfg['ln'] = False
        else:
            raise ValueError(f'Pattern {tok} has not been implemented!')
    return cfg


# Set by the trainers around an update (`training_pass()`): nobody reads the per-layer output sequences (`full_rnn_memory`) of a training
# pass, so a sequence layer may hand back its output with the following plain ELU already applied by its last GEMM's epilogue (the
# reference's full hidden holds the PRE-activation sequence, rnn_base.py:456-460: outside a training pass nothing changes).  The entry of
# such a layer in the returned `full` RNNHidden is None.
# Per THREAD: a rollout / evaluation forward running on another thread while an update is in flight (the reference samples from the policy
# between updates; a user may do so concurrently) must keep getting the pre-activation sequences.  Autograd's backward threads never read it.
import threading
_TRAINING_PASS = threading.local()


def _in_training_pass() -> bool:
    return getattr(_TRAINING_PASS, 'depth', 0) > 0


class training_pass:
    def __enter__(self):
        _TRAINING_PASS.depth = getattr(_TRAINING_PASS, 'depth', 0) + 1

    def __exit__(self, *exc):
        _TRAINING_PASS.depth -= 1
        return False


def is_rnn_layer(layer_id: str) -> bool:
    return layer_id != 'fc' and not layer_id.startswith('efc')


class RNNBase(torch.nn.Module):
    def __init__(self, input_size: int, output_size: int, hidden_size_list: List[int], activation: List[str],
                 layer_type: List[str]):
        super().__init__()
        assert len(activation) - 1 == len(hidden_size_list), 'number of activation should be larger by 1 than size of hidden layers.'
        assert len(activation) == len(layer_type), 'number of layer type should equal to the activate'
        self.activation_dict = ACTIVATIONS
        self.check_is_rnn = is_rnn_layer
        self.layer_type = copy.deepcopy(layer_type)
        self.activation_type = copy.deepcopy(activation)
        self.layer_list = torch.nn.ModuleList()
        self.activation_list = torch.nn.ModuleList()
        self.rnn_hidden_state_input_size, self.rnn_layer_type = [], []
        self.rnn_num = 0
        self.input_size = input_size
        self.output_size = output_size
        width_in = input_size
        for ind, width in enumerate(list(hidden_size_list) + [output_size]):
            lid = self.layer_type[ind]
            layer, hidden_width = self._make_layer(lid, width_in, width)
            self.layer_list.append(layer)
            if hidden_width is not None:
                self.rnn_num += 1
                self.rnn_hidden_state_input_size.append(hidden_width)
                self.rnn_layer_type.append(lid)
            self.activation_list.append(self._make_activation(activation[ind], width))
            width_in = width
        self.xavier_initialize_weights()

    @staticmethod
    def _make_layer(lid: str, n_in: int, n_out: int):
        if lid == 'fc':
            return Linear(n_in, n_out), None
        if lid.startswith('efc'):
            return EnsembleLinear(n_in, n_out, int(lid.split('-')[-1])), None
        if lid == 'gru':
            return GRU(n_in, n_out, batch_first=True), n_out
        if lid == 'gilr':
            return GILRLayer(n_in, n_out, batch_first=True), n_out
        if lid == 'lru':
            return LRULayer(n_in, n_out, batch_first=True), n_out * 2
        if lid == 'gilr_lstm':
            return GILRLSTMLayer(n_in, n_out, batch_first=True), n_out * 2
        if lid.startswith('mamba'):
            cfg = parse_mamba_id(lid)
            layer = MambaResidualBlock(n_in, n_out, d_conv=cfg['d_conv'], d_state=cfg['d_state'], use_ff=cfg['use_ff'])
            return layer, layer.mixer.desired_hidden_dim
        if lid.startswith('conv1d'):
            layer = Conv1d(n_in, n_out, d_conv=int(lid.split('_')[-1]) if '_' in lid else 4)
            return layer, layer.desired_hidden_dim
        if lid.startswith('smamba'):
            cfg = parse_smamba_id(lid)
            assert n_in == n_out, f'mamba_simple require input_dim == output_dim, while got {n_in} and {n_out}'
            layer = MambaBlockList(cfg['block_num'], n_in, d_conv=cfg['d_conv'], d_state=cfg['d_state'], rms_norm=cfg['rms_norm'],
                                   use_ff=cfg['use_ff'])
            return layer, layer.desired_hidden_dim
        if lid.startswith('cgpt'):
            cfg = parse_cgpt_id(lid)
            return TransformerDecoder(n_in, cfg['nhead'], 4 * n_in, cfg['nlayer'], cfg['pdrop'], cfg['ln']), cfg['maxlength']
        if lid == 'lstm':
            # the reference lists `lstm` (rnn_base.py:58) but cannot run it on this path either: ContextualModel.meta_forward asks every
            # embedding network for its full hidden sequence (contextual_model.py:92-94, 113-115) and RNNHidden refuses to hold one for an
            # LSTM (RNNHidden.py:23-25: 'It is not supported to store full RNN output of LSTM!!!')
            raise NotImplementedError("layer id 'lstm': the reference's full-trajectory path rejects it as well (RNNHidden.py:23-25 asserts "
                                      "when ContextualModel.meta_forward requests the full hidden sequence, contextual_model.py:92-94)")
        if lid.startswith(_UNSUPPORTED_PREFIXES):
            raise NotImplementedError(f'layer id {lid!r} exists in the reference but is outside the MI355X hot path of this build '
                                      f'(supported: fc, efc-<E>, gru, gilr, lru, gilr_lstm, smamba_*, mamba_*, conv1d_*, cgpt_*)')
        raise NotImplementedError(f'unknown layer id {lid!r}')

    @staticmethod
    def _make_activation(spec: str, width: int):
        if '+' not in spec:
            return ACTIVATIONS[spec]()
        norm, act = spec.split('+')
        if norm.startswith('eln'):
            shape = [int(norm.split('-')[-1]), width]
        elif norm == 'ln':
            shape = width
        else:
            raise NotImplementedError(f'norm {norm}')
        return torch.nn.ModuleList([torch.nn.LayerNorm(shape), ACTIVATIONS[act]()])

    def xavier_initialize_weights(self):
        """Xavier-uniform weights / zero biases for the plain layers; smamba keeps its own init (reference :267-354)."""
        def xavier_ensemble(efc):
            with torch.no_grad():
                for i in range(efc.weight.shape[0]):
                    torch.nn.init.xavier_uniform_(efc.weight[i].transpose(0, 1))
                if getattr(efc, 'bias', None) is not None:
                    efc.bias.zero_()
        for m in self.layer_list:
            if isinstance(m, torch.nn.Linear):
                torch.nn.init.xavier_uniform_(m.weight)
                if m.bias is not None:
                    torch.nn.init.constant_(m.bias, 0)
            elif isinstance(m, EnsembleLinear):
                xavier_ensemble(m)
            elif isinstance(m, LRULayer):
                xavier_ensemble(m.in_proj)
                xavier_ensemble(m.middle_proj)
            elif isinstance(m, GILRLayer):
                torch.nn.init.xavier_uniform_(m.out_proj.weight)
                torch.nn.init.constant_(m.out_proj.bias, 0)
                xavier_ensemble(m.in_proj)
            elif isinstance(m, GILRLSTMLayer):
                torch.nn.init.xavier_uniform_(m.out_proj.weight)
                torch.nn.init.constant_(m.out_proj.bias, 0)
                xavier_ensemble(m.in_proj)
                xavier_ensemble(m.middle_proj)
            elif isinstance(m, (MambaBlockList, TransformerDecoder, MambaResidualBlock, Conv1d)):
                pass
            else:
                for name, param in m.named_parameters():
                    if 'weight' in name:
                        torch.nn.init.xavier_uniform_(param.data)
                    elif 'bias' in name:
                        torch.nn.init.constant_(param.data, 0)

    def rnn_parameters(self, recursive=True):
        out = []
        for lid, layer in zip(self.layer_type, self.layer_list):
            if is_rnn_layer(lid):
                out += list(layer.rnn_parameters()) if hasattr(layer, 'rnn_parameters') else list(layer.parameters(recursive))
        return out

    # ------------------------------------------------------------------------------------------ hidden state
    def _make_state(self, batch_size, device, random: bool) -> RNNHidden:
        st = RNNHidden(self.rnn_num, self.rnn_layer_type, device)
        for width, lid in zip(self.rnn_hidden_state_input_size, self.rnn_layer_type):
            if lid.startswith('cgpt'):
                st.append(InferenceParams(max_seqlen=width, max_batch_size=batch_size))
                continue
            make = st.init_random_hidden_by_type if random else st.init_hidden_by_type
            st.append(make(lid, batch_size, width, device))
        return st

    def make_init_state(self, batch_size: int, device: Union[str, torch.device] = torch.device('cpu')) -> RNNHidden:
        return self._make_state(batch_size, device, False)

    def make_rnd_init_state(self, batch_size: int, device: Union[str, torch.device] = torch.device('cpu')) -> RNNHidden:
        return self._make_state(batch_size, device, True)

    # ------------------------------------------------------------------------------------------ forward
    def _fuse_out_act(self, ind, x) -> bool:
        """The activation module behind sequence layer `ind` is a plain ELU and this is a training pass over whole fp32 GPU sequences: the
        layer may apply it in its last kernel (`out_act='elu'`)."""
        act_mod = self.activation_list[ind]
        return (_in_training_pass() and isinstance(act_mod, torch.nn.ELU) and act_mod.alpha == 1.0 and x.is_cuda
                and x.dtype == torch.float32 and x.dim() == 3 and x.shape[-2] > 1)

    def meta_forward(self, x: torch.Tensor, hidden_state: Optional[RNNHidden] = None, require_full_hidden: bool = False,
                     first_grad_part=None, out_dest=None) -> Tuple[torch.Tensor, RNNHidden, Optional[RNNHidden]]:
        """first_grad_part = (x_part, col0): the first layer (a shared-input efc layer) differentiates x only through that
        column block (EnsembleLinear.forward); anything else keeps the ordinary path.
        out_dest: an `ops.ColDest` - if the LAST layer is an `fc` on the hand-written GEMM, its output is written straight into that column
        block of a row buffer (the caller checks with `dest.holds`)."""
        assert x.shape[-1] == self.input_size, f'inputting size does not match!!!! input is {x.shape[-1]}, expected: {self.input_size}'
        if hidden_state is None:
            hidden_state = self.make_init_state(x.shape[0], x.device)
        assert len(hidden_state) == self.rnn_num, f'rnn num does not match, input is {len(hidden_state)}, expected: {self.rnn_num}'
        squeeze = x.dim() == 2 and self.rnn_num > 0
        if squeeze:
            x = x.unsqueeze(0)
        out_state = RNNHidden(self.rnn_num, self.rnn_layer_type, device=x.device, batch_first=False)
        full = RNNHidden(self.rnn_num, self.rnn_layer_type, device=x.device, batch_first=True) if require_full_hidden else None
        k = 0
        n_layers = len(self.layer_list)
        fused_next = 0
        for ind, layer in enumerate(self.layer_list):
            if fused_next:                          # consumed by the ensemble head / critic MLP node below
                fused_next -= 1
                continue
            lid = self.layer_type[ind]
            if is_rnn_layer(lid):
                if lid in ('gilr', 'lru') and self._fuse_out_act(ind, x) and layer.use_ff:
                    # the plain ELU behind the layer rides in its closing add + LayerNorm kernel (training passes only, see _TRAINING_PASS)
                    if lid == 'gilr':
                        x, h = layer(x, hidden_state[k], hidden_state.rnn_start, out_act='elu')
                    else:
                        x, h = layer(x, hidden_state[k], hidden_state.rnn_start, hidden_state.grad_detach, out_act='elu')
                    k += 1
                    out_state.append(h)
                    if require_full_hidden:
                        full.append(None)
                    continue
                if lid in ('gilr', 'gilr_lstm'):
                    x, h = layer(x, hidden_state[k], hidden_state.rnn_start)
                elif lid == 'lru':
                    x, h = layer(x, hidden_state[k], hidden_state.rnn_start, hidden_state.grad_detach)
                elif lid.startswith('smamba'):
                    fuse_act = self._fuse_out_act(ind, x) and not layer.use_ff
                    x, h = layer(x, hidden_state[k], hidden_state.rnn_start, hidden_state.mask, out_act='elu' if fuse_act else None)
                    if fuse_act:                       # the activation rode in the layer's last GEMM: skip the module below
                        k += 1
                        out_state.append(h)
                        if require_full_hidden:
                            full.append(None)
                        continue
                elif lid.startswith('mamba'):
                    x, h = layer(x, hidden_state[k], hidden_state.rnn_start, hidden_state.mask, hidden_state.grad_detach)
                elif lid.startswith('conv1d'):
                    x, h = layer(x, hidden_state[k], hidden_state.mask)
                elif lid.startswith('cgpt'):
                    multi = x.dim() == 3 and x.shape[-2] > 1          # whole packed rows (training) vs one rollout step
                    fuse_act = multi and self._fuse_out_act(ind, x) and x.shape[0] * x.shape[1] >= ops.GEMM_F32_MIN_ROWS
                    x = layer(x, inference_params=None if multi else hidden_state[k],
                              seqlens=hidden_state.attention_concat_mask if multi else None, out_act='elu' if fuse_act else None)
                    h = hidden_state[k]
                    if not multi:
                        h.seqlen_offset += x.shape[-2]        # reference :451-452
                    if fuse_act:                       # the activation rode in the decoder's output GEMM: skip the module below
                        k += 1
                        out_state.append(h)
                        if require_full_hidden:
                            full.append(None)
                        continue
                else:                                   # gru: no reset / mask handling (reference :453-454)
                    x, h = layer(x, hidden_state[k])
                k += 1
                out_state.append(h)
                if require_full_hidden:
                    full.append(x)
                act = self.activation_list[ind]
            else:
                act = self.activation_list[ind]
                # efc-E ELU -> efc-E(H) ELU -> efc-E(1) = the published critic: ONE node whose GEMM epilogues carry the head and the
                # ELU backward / bias gradient between the layers (ensemble_linear_model._CriticMLP)
                if ind + 3 == n_layers and critic_mlp_fusable(layer, act, self.layer_list[ind + 1], self.activation_list[ind + 1],
                                                              self.layer_list[ind + 2], self.activation_list[ind + 2], x):
                    x = critic_mlp(layer, self.layer_list[ind + 1], self.layer_list[ind + 2], x, grad_part=first_grad_part if ind == 0 else None)
                    fused_next = 2
                    continue
                # efc-E(H) ELU -> efc-E(1) at the end of the stack (the critic head): one fused node
                if ind + 2 == n_layers and head_fusable(layer, act, self.layer_list[ind + 1], self.activation_list[ind + 1], x):
                    x = ensemble_head(layer, self.layer_list[ind + 1], x)
                    fused_next = 1
                    continue
                # plain ELU behind fc / efc-E: fused into the layer's bias pass (one in-place kernel; backward from the output)
                if isinstance(act, torch.nn.ELU) and act.alpha == 1.0 and isinstance(layer, (EnsembleLinear, torch.nn.Linear)) \
                        and x.dtype == torch.float32:
                    if isinstance(layer, EnsembleLinear):
                        x = layer(x, act='elu', grad_part=first_grad_part if ind == 0 else None)
                    else:
                        assert not (ind == 0 and first_grad_part is not None)
                        x = ops.linear_act(x, layer.weight, layer.bias, 'elu', dest=out_dest if ind == n_layers - 1 else None)
                    continue
                assert not (ind == 0 and first_grad_part is not None), 'first_grad_part needs an efc first layer with ELU'

                if isinstance(layer, torch.nn.Linear):                    # the bias rides in the GEMM epilogue (every pass: whole trajectories and rollout steps)
                    x = ops.linear_act(x, layer.weight, layer.bias, None, dest=out_dest if ind == n_layers - 1 else None)
                else:
                    x = layer(x)
            if isinstance(act, torch.nn.ModuleList):
                if self.activation_type[ind].startswith('eln'):
                    x = act[0](x.transpose(-2, 0)).transpose(-2, 0)
                else:
                    x = act[0](x)
                x = act[1](x)
            else:
                x = act(x)
        if squeeze:
            x = x.squeeze(0)
        return x, out_state, full

    # ------------------------------------------------------------------------------------------ weights
    @staticmethod
    def _copy_weight_from(dst_net: torch.nn.Module, src_net: torch.nn.Module, tau: float):
        """dst <- tau * dst + (1 - tau) * src  (tau = 0: hard copy).  Per-tensor form; ContextualModel overrides this
        with one flat-buffer kernel for whole networks."""
        with torch.no_grad():
            if tau == 0.0:
                dst_net.load_state_dict(src_net.state_dict())
                return
            if tau == 1.0:
                return
            src, dst = list(src_net.parameters(True)), list(dst_net.parameters(True))
            assert len(src) == len(dst), 'parameter number show be equal!'
            for s, d in zip(src, dst):
                d.data.mul_(tau).add_(s.data, alpha=1 - tau)

    def copy_weight_from(self, src_net: 'RNNBase', tau: float):
        RNNBase._copy_weight_from(self, src_net, tau)

    def save(self, path):
        os.makedirs(os.path.dirname(path), exist_ok=True)
        torch.save(self.state_dict(), path)

    def load(self, path, **kwargs):
        self.load_state_dict(torch.load(path, map_location=kwargs.get('map_location')))

    def l2_norm_square(self) -> torch.Tensor:
        return sum(torch.sum(p ** 2) for p in self.parameters(True))
