"""ContextualModel: context-encoder tower ("embedding_model") + universal tower over [input mapping, embedding]
(reference offpolicy_rnn/models/contextual_model.py:9-228), with every parameter of the network living in ONE flat
fp32 buffer (see flat_params.py) once `finalize_parameters()` has run."""
import os
from collections import OrderedDict
from typing import Dict, List, Optional, Tuple

import torch

from .RNNHidden import RNNHidden
from .flat_params import FlatParameterStore
from .mlp_base import MLPBase
from .rnn_base import RNNBase
from ..hip import ops


class ContextualModel:
    def __init__(self, embedding_input_size: int, embedding_size: int, embedding_hidden: List[int], embedding_activations: List[str],
                 embedding_layer_type: List[str], uni_model_input_size: int, uni_model_output_size: int, uni_model_hidden: List[int],
                 uni_model_activations: List[str], uni_model_layer_type: List[str], fix_rnn_length: int, name: str,
                 uni_model_input_mapping_dim: int = 0, uni_model_input_mapping_activation: str = 'linear'):
        if fix_rnn_length and fix_rnn_length > 0:
            raise NotImplementedError('rnn_fix_length > 0 (fixed-window forward) is outside the full-trajectory hot path')
        self.name = name
        self.fix_rnn_length = self._fix_rnn_length = fix_rnn_length
        self.embedding_size = embedding_size
        self.uni_model_input_mapping_dim = uni_model_input_mapping_dim
        self.embedding_network = RNNBase(embedding_input_size, embedding_size, embedding_hidden, embedding_activations,
                                         embedding_layer_type)
        uni_in = uni_model_input_size if uni_model_input_mapping_dim == 0 else uni_model_input_mapping_dim
        self.uni_network = RNNBase(embedding_size + uni_in, uni_model_output_size, uni_model_hidden, uni_model_activations,
                                   uni_model_layer_type)
        self.contextual_modules: 'OrderedDict[str, torch.nn.Module]' = OrderedDict()
        self.contextual_register_rnn_base_module(self.embedding_network, 'embedding_model')
        self.contextual_register_rnn_base_module(self.uni_network, 'universal_model')
        if uni_model_input_mapping_dim > 0:
            self.uni_input_mapping_network = MLPBase(uni_model_input_size, uni_model_input_mapping_dim, [],
                                                     [uni_model_input_mapping_activation])
            self.contextual_register_rnn_base_module(self.uni_input_mapping_network, 'uni_input_mapping_network')
        else:
            self.uni_input_mapping_network = torch.nn.Identity()
        self.rnn_num = self.embedding_network.rnn_num + self.uni_network.rnn_num
        self.device = torch.device('cpu')
        self.dtype = torch.float32
        self.store: Optional[FlatParameterStore] = None

    def contextual_register_rnn_base_module(self, module, module_name: str):
        self.contextual_modules[module_name] = module

    # ------------------------------------------------------------------------------------------ parameters
    def finalize_parameters(self):
        """Move every parameter into one flat buffer (call once, after the sub-class registered all its modules)."""
        if self.store is None:
            self.store = FlatParameterStore(self.contextual_modules)
        return self.store

    def parameters(self, recursive=True) -> List[torch.Tensor]:
        return [p for m in self.contextual_modules.values() for p in m.parameters(recursive)]

    def rnn_parameters(self, recursive=True):
        out = []
        for m in self.contextual_modules.values():
            if hasattr(m, 'rnn_parameters'):
                out += list(m.rnn_parameters(recursive))
        return out

    def to(self, device: torch.device = None, dtype: torch.dtype = None) -> None:
        if device is not None and self.device != torch.device(device):
            self.device = torch.device(device)
            if self.store is not None:
                self.store.to(self.device)
            else:
                for m in self.contextual_modules.values():
                    m.to(self.device)
        if dtype is not None and dtype != self.dtype:
            raise NotImplementedError('the MI355X build keeps parameters in fp32')

    # ------------------------------------------------------------------------------------------ forward
    def head_row_buffer(self, lead, device, dtype):
        """A row buffer [prod(lead), uni input width + embedding width] for the head input of a long fp32 GPU pass (None otherwise): the
        GEMMs that produce the input encoding and the embedding write their column blocks in place instead of a `cat` afterwards."""
        rows = 1
        for v in lead:
            rows *= int(v)
        if torch.device(device).type != 'cuda' or dtype != torch.float32 or rows < ops.GEMM_F32_MIN_ROWS or len(lead) == 0:
            return None
        return ops.RowBuffer(lead, self.uni_network.input_size, device)

    def meta_forward(self, embedding_input: torch.Tensor, uni_model_input, rnn_memory=None, detach_embedding=False,
                     uni_grad_part=None, row_buffer=None) -> Tuple[torch.Tensor, RNNHidden, torch.Tensor, RNNHidden]:
        """uni_grad_part = (x_part, col0): the head input is differentiated only through that column block of
        `uni_model_input` (which must carry no graph of its own, like the detached embedding beside it).
        row_buffer (`head_row_buffer`): the caller had the head input's leading columns produced into it; `uni_model_input` may then be a
        list of (tensor, col0) pieces.  Without one, a buffer is made here when the input mapping network and the embedding can fill it."""
        if rnn_memory is None:
            rnn_memory = self.make_init_state(1 if embedding_input.dim() == 2 else embedding_input.shape[0], embedding_input.device)
        n_emb = self.embedding_network.rnn_num
        pre, self._prefetched = getattr(self, '_prefetched', None), None
        if pre is not None:
            # the (graph-free) embedding pass already ran on a side stream (prefetch_embedding): join it here
            assert detach_embedding or not torch.is_grad_enabled(), 'a prefetched embedding carries no graph'
            emb, emb_mem, emb_full, event = pre
            main = torch.cuda.current_stream(emb.device)
            main.wait_event(event)
            for t in [emb] + [h for h in list(emb_mem._data) + list(emb_full._data) if torch.is_tensor(h)]:
                t.record_stream(main)
        else:
            mapped = not isinstance(self.uni_input_mapping_network, torch.nn.Identity)
            if row_buffer is None and mapped and torch.is_tensor(uni_model_input) and uni_model_input.dim() == embedding_input.dim():
                row_buffer = self.head_row_buffer(embedding_input.shape[:-1], embedding_input.device, embedding_input.dtype)
            w_emb = self.embedding_network.output_size
            dest = None if row_buffer is None else row_buffer.block(row_buffer.width - w_emb, w_emb)
            # a detached embedding is computed without a graph: same values, but no scan checkpoints / saved activations
            with torch.set_grad_enabled(torch.is_grad_enabled() and not detach_embedding):
                emb, emb_mem, emb_full = self.embedding_network.meta_forward(embedding_input, rnn_memory[:n_emb], require_full_hidden=True,
                                                                             out_dest=dest)
        if detach_embedding:
            emb = emb.detach()
        if row_buffer is not None and emb.shape[:-1] == row_buffer.lead:
            w_uni = row_buffer.width - emb.shape[-1]
            if isinstance(uni_model_input, (list, tuple)):
                pieces = list(uni_model_input)
            elif isinstance(self.uni_input_mapping_network, torch.nn.Identity):
                pieces = [(uni_model_input, 0)]
            else:
                pieces = [(self.uni_input_mapping_network(uni_model_input, out_dest=row_buffer.block(0, w_uni)), 0)]
            head_in = ops.cat_into(row_buffer, pieces + [(emb, w_uni)])
        else:
            assert torch.is_tensor(uni_model_input), 'piece lists need the row buffer they were produced into'
            uni_in = self.uni_input_mapping_network(uni_model_input)
            if emb.dim() - uni_in.dim() == 1:
                uni_in = uni_in.unsqueeze(0).repeat_interleave(repeats=emb.shape[0], dim=0)
            head_in = torch.cat((uni_in, emb), dim=-1)
        out, uni_mem, uni_full = self.uni_network.meta_forward(head_in, rnn_memory[n_emb:], require_full_hidden=True, first_grad_part=uni_grad_part)
        return out, emb_mem + uni_mem, emb, emb_full + uni_full

    def prefetch_embedding(self, embedding_args, rnn_memory, stream) -> None:
        """Run the embedding pass of the NEXT no-grad / detached-embedding forward on `stream` (forked from the current
        stream); that forward then only waits for it.  For latency-bound recurrent layers (gru: ~3 us per step whatever the
        batch) this lets an independent pass - the actor's - use the otherwise idle chip at the same time."""
        main = torch.cuda.current_stream(self.device)
        stream.wait_stream(main)
        with torch.cuda.stream(stream), torch.no_grad():
            emb, mem, full = self.get_embedding(self.get_embedding_input(*embedding_args), rnn_memory)
            event = torch.cuda.Event()
            event.record(stream)
        self._prefetched = (emb, mem, full, event)

    def get_embedding(self, x, rnn_memory):
        n_emb = self.embedding_network.rnn_num
        mem = rnn_memory[:n_emb] if rnn_memory is not None and len(rnn_memory) > 0 else None
        return self.embedding_network.meta_forward(x, mem, require_full_hidden=True)

    def make_init_state(self, batch_size: int, device: torch.device) -> RNNHidden:
        return self.embedding_network.make_init_state(batch_size, device) + self.uni_network.make_init_state(batch_size, device)

    def make_rnd_init_state(self, batch_size, device):
        return self.embedding_network.make_rnd_init_state(batch_size, device) + self.uni_network.make_rnd_init_state(batch_size, device)

    # ------------------------------------------------------------------------------------------ weights / io
    def copy_weight_from(self, src: 'ContextualModel', tau: float) -> None:
        """self <- tau * self + (1 - tau) * src (tau = 0: hard copy); one kernel over the flat buffers."""
        if self.store is not None and src.store is not None and self.store.numel == src.store.numel:
            with torch.no_grad():
                if tau == 0.0:
                    self.store.flat.copy_(src.store.flat.to(self.store.flat.device))
                    ops.PARAM_EPOCH[0] += 1             # a raw write into the parameter buffer: weight magnitudes (GEMM mode 2) are refreshed on next use
                elif tau != 1.0:
                    ops.soft_update_(self.store.flat, src.store.flat, tau)
            return
        for k, v in self.contextual_modules.items():
            RNNBase._copy_weight_from(v, src.contextual_modules[k], tau)

    def state_dict(self, destination=None, prefix='', keep_vars=False):
        return {k: v.state_dict(destination=destination, prefix=prefix, keep_vars=keep_vars) for k, v in self.contextual_modules.items()}

    def load_state_dict(self, state_dict):
        for k, v in self.contextual_modules.items():
            v.load_state_dict(state_dict[k])

    def save(self, path: str, index=0) -> None:
        os.makedirs(path, exist_ok=True)
        for k, v in self.contextual_modules.items():
            torch.save(v.state_dict(), os.path.join(path, f'{self.name}-{index}-{k}.pt'))

    def load(self, path: str, index=0, **kwargs) -> None:
        for k, v in self.contextual_modules.items():
            v.load_state_dict(torch.load(os.path.join(path, f'{self.name}-{index}-{k}.pt'), **kwargs))

    def train(self, mode=True):
        """Same effect as `module.train(mode)` on every registered module (reference contextual_model.py: train / eval), without
        nn.Module's recursive walk: `training` is a plain instance attribute, and an update toggles the modes four times - at small
        batches the walks were ≈ 0.5 ms of a host-bound update.  The module list is rebuilt when the registry changes size."""
        mode = bool(mode)
        cache = self.__dict__.get('_mode_modules')
        if cache is None or cache[0] != len(self.contextual_modules):
            cache = (len(self.contextual_modules), [m for v in self.contextual_modules.values() for m in v.modules()])
            self.__dict__['_mode_modules'] = cache
        for m in cache[1]:
            m.__dict__['training'] = mode

    def eval(self):
        self.train(False)

    def set_fix_length(self, enable: bool):
        pass

    def l2_norm_square(self) -> torch.Tensor:
        """Sum of squares over the RNNBase modules only - plain nn.Linear encoders have no `l2_norm_square`
        in the reference (contextual_model.py:227-228).  RNNBase modules are registered first, i.e. a prefix of the flat buffer."""
        if self.store is not None:
            end = 0
            for k, v in self.contextual_modules.items():
                if isinstance(v, RNNBase):
                    end = max(end, self.store.module_range[k][1])
            return ops.sumsq(self.store.flat[:end])[0]
        return sum(m.l2_norm_square() for m in self.contextual_modules.values() if hasattr(m, 'l2_norm_square'))
