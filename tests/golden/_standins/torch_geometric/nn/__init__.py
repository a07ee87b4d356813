"""Minimal stand-in for torch_geometric.nn (PyG 1.6.1 semantics used by subgraph_mpn.py)."""
import inspect
import torch


class _Inspector:
    def __init__(self, owner):
        self.params = {}
        for name in ('message', 'aggregate', 'update'):
            sig = inspect.signature(getattr(owner, name))
            self.params[name] = [p for p in sig.parameters]

    def distribute(self, func_name, kwargs):
        out = {}
        for p in self.params[func_name]:
            if p in kwargs:
                out[p] = kwargs[p]
        return out


class MessagePassing(torch.nn.Module):
    def __init__(self, aggr='add', flow='source_to_target', node_dim=-2):
        super().__init__()
        assert aggr == 'add' and flow == 'source_to_target'
        self.aggr, self.flow, self.node_dim = aggr, flow, node_dim
        self.inspector = _Inspector(self)
        msg_args = set(self.inspector.params['message'])
        agg_args = set(self.inspector.params['aggregate']) - {'inputs', 'index', 'ptr', 'dim_size'}
        upd_args = set(self.inspector.params['update']) - {'aggr_out', 'inputs'}
        self.__user_args__ = msg_args | agg_args | upd_args

    def __check_input__(self, edge_index, size):
        assert edge_index.dtype == torch.long and edge_index.dim() == 2 and edge_index.size(0) == 2
        return [None, None] if size is None else list(size)

    def __collect__(self, args, edge_index, size, kwargs):
        i, j = 1, 0                     # source_to_target: x_j = source = edge_index[0]
        out = {}
        for arg in args:
            if arg[-2:] not in ('_i', '_j'):
                out[arg] = kwargs.get(arg, inspect.Parameter.empty)
            else:
                idx = j if arg[-2:] == '_j' else i
                data = kwargs.get(arg[:-2], inspect.Parameter.empty)
                if isinstance(data, torch.Tensor):
                    if size[idx] is None:
                        size[idx] = data.size(self.node_dim)
                    data = data.index_select(self.node_dim, edge_index[idx])
                out[arg] = data
        size[0] = size[1] if size[0] is None else size[0]
        size[1] = size[0] if size[1] is None else size[1]
        out['index'] = edge_index[i]
        out['ptr'] = None
        out['size'] = size
        out['dim_size'] = size[1]
        return out

    def aggregate(self, inputs, index, ptr=None, dim_size=None):
        # torch_scatter.scatter(inputs, index, dim=-2, dim_size=dim_size, reduce='sum')
        out = torch.zeros((dim_size, inputs.size(-1)), dtype=inputs.dtype, device=inputs.device)
        return out.index_add_(0, index, inputs)

    def message(self, x_j):
        return x_j

    def update(self, inputs):
        return inputs


class GINConv(torch.nn.Module):      # import-only in SubGNN.py:39
    pass
