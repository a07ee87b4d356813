"""SubGNN module with the reference's LightningModule surface (mirrors reference
SubGNN/SubGNN.py): same constructor ``SubGNN(hparams, graph_path, subgraph_path, embedding_path,
similarities_path, shortest_paths_path, degree_dict_path, ego_graph_path)``, same hooks
(prepare_data, *_dataloader, training_step, validation_step/test_step, *_epoch_end,
configure_optimizers, backward), same ``forward`` signature, same state-dict key names, same
``metric_scores`` / ``test_results`` attributes -- so the reference's ``train_config.py`` can
drive it unchanged.

What differs is where the work runs: the base graph, all component / anchor / similarity
tensors and the embedding table stay resident in HBM; connected components, border sets,
shortest-path and DTW similarities, walks, anchor draws and the per-step gather-weight-
aggregate layers are HIP kernels (subgnn_amd.ops); the LSTM, the Linear layers, BatchNorm and
the loss are torch modules (MIOpen / rocBLAS).  Randomness is the draw tape keyed by
``hparams['seed']`` (subgnn_amd.tape).
"""
import json
from pathlib import Path

import numpy as np
import torch
import torch.nn as nn
import torch.nn.functional as F
from torch.nn.parameter import Parameter

from . import config, gamma, ops, subgraph_utils
from . import anchor_patch_samplers as aps
from .datasets import SubgraphDataset
from .graph import load_graph
from .subgraph_mpn import SG_MPN

CHANNELS = (('neighborhood', 'N', 'use_neighborhood', 'neighborhood_mpns'),
            ('position', 'P', 'use_position', 'position_mpns'),
            ('structure', 'S', 'use_structure', 'structure_mpns'))
CC_SLOTS = ('N_I', 'N_B', 'S_I', 'S_B', 'P_I', 'P_B')


class LSTM(nn.Module):
    """bidirectional LSTM + Linear(2h -> n_features) (S.py:60-88); keys lstm.lstm.* / lstm.linear.*"""

    def __init__(self, n_features, h, dropout=0.0, num_layers=1, batch_first=True, aggregator='last'):
        super().__init__()
        if aggregator not in ('last', 'sum'):
            raise NotImplementedError(aggregator)
        self.num_layers, self.aggregator = num_layers, aggregator
        self.lstm = nn.LSTM(n_features, h, num_layers=num_layers, batch_first=batch_first, dropout=dropout,
                            bidirectional=True)
        self.linear = nn.Linear(h * 2, n_features)

    def _fused(self, input):
        """The recurrence in one launch per layer and direction pair (ops.bilstm_layer) when the sizes
        are ones the kernel is built for; the parameters are nn.LSTM's own (state-dict compatible)."""
        m = self.lstm
        out = input
        for l in range(self.num_layers):
            params = [getattr(m, '%s_l%d%s' % (n, l, sfx)) for sfx in ('', '_reverse')
                      for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            out = ops.bilstm_layer(out, params)
            if l < self.num_layers - 1 and m.dropout > 0 and self.training:
                out = nn.functional.dropout(out, m.dropout, True)
        return out

    def forward_walks(self, table, walk_ids, n_walks):
        """The walk aggregator in one chain of this library's launches (aps:413-433): embedding lookup of ``walk_ids`` (B, T) ->
        the LSTM layers -> last step / sum over steps -> Linear -> sum over every ``n_walks`` consecutive sequences -> (B /
        n_walks, n_features).  The lookup is the first layer's operand load (ops.bilstm_layer_fused), the tail one launch
        (ops.lstm_tail).  None where the fused form does not apply (the caller takes forward())."""
        m = self.lstm
        if not (table.is_cuda and table.dtype == torch.float32 and m.batch_first and m.bias and walk_ids.dim() == 2
                and getattr(table, '_sgnn_half', None) is None and table.shape[1] == m.input_size
                and ops.lstm_fused_supported(m.input_size, m.hidden_size) and walk_ids.shape[0] % n_walks == 0):
            return None
        out = None
        for l in range(self.num_layers):
            params = [getattr(m, '%s_l%d%s' % (n, l, sfx)) for sfx in ('', '_reverse')
                      for n in ('weight_ih', 'weight_hh', 'bias_ih', 'bias_hh')]
            out = ops.bilstm_layer_fused(table, params, ids=walk_ids) if l == 0 else ops.bilstm_layer_fused(out, params)
            if l < self.num_layers - 1 and m.dropout > 0 and self.training:
                out = nn.functional.dropout(out, m.dropout, True)
        return ops.lstm_tail(out, self.linear.weight, self.linear.bias, n_walks, self.aggregator == 'last')

    def forward(self, input):
        m = self.lstm
        if (input.is_cuda and input.dtype == torch.float32 and m.batch_first and m.bias and input.dim() == 3
                and ops.lstm_supported(m.input_size, m.hidden_size)):
            out = self._fused(input)
        else:
            out, _ = m(input)                     # hidden sizes other than 32 / 64 / 128: the library LSTM
        agg = out[:, -1, :] if self.aggregator == 'last' else out.sum(dim=1)
        return self.linear(agg)


class _DeviceLoader:
    """Batches assembled by whole-tensor index_select on the device (the tensors never leave
    HBM); yields the same dict ``_pad_collate`` builds (S.py:1112-1114)."""

    def __init__(self, model, split, batch_size, shuffle, drop_last):
        self.m, self.split, self.bs, self.shuffle, self.drop_last = model, split, batch_size, shuffle, drop_last
        self.n = len(getattr(model, split + '_sub_G'))

    def __len__(self):
        return self.n // self.bs if self.drop_last else (self.n + self.bs - 1) // self.bs

    def index_batches(self):
        """The subgraph indices of each batch (what graph_step.CapturedTrainStep.replay takes)."""
        order = torch.randperm(self.n) if self.shuffle else torch.arange(self.n)
        for i in range(len(self)):
            yield order[i * self.bs:(i + 1) * self.bs]

    def __iter__(self):
        for idx in self.index_batches():
            yield self.m.make_batch(self.split, idx)


class SubGNN(nn.Module):
    def __init__(self, hparams, graph_path, subgraph_path, embedding_path, similarities_path, shortest_paths_path,
                 degree_dict_path, ego_graph_path, _memory=None):
        super().__init__()
        self._memory = _memory
        self.device = torch.device('cuda' if torch.cuda.is_available() else 'cpu')
        self.hparams = hparams
        self.graph_path, self.subgraph_path, self.embedding_path = graph_path, subgraph_path, embedding_path
        self.similarities_path = Path(similarities_path)
        self.shortest_paths_path, self.degree_dict_path, self.ego_graph_path = \
            shortest_paths_path, degree_dict_path, ego_graph_path
        self.read_data()

        hp = self.hparams
        D, nl = hp['node_embed_size'], hp['n_layers']
        hid_dim = D
        for channel, tag, flag, attr in CHANNELS:
            layers = nn.ModuleList()
            if hp[flag]:
                if tag == 'N':
                    hid_dim += nl * 2 * D
                elif tag == 'P':
                    hid_dim += (hp['n_anchor_patches_pos_in'] + hp['n_anchor_patches_pos_out']) * nl
                else:
                    hid_dim += 2 * hp['n_anchor_patches_structure'] * nl
                for _ in range(nl):
                    layer = nn.ModuleDict({'internal': SG_MPN(hp), 'border': SG_MPN(hp)})
                    if hp.get('batch_norm', False):
                        layer['batch_norm'] = nn.BatchNorm1d(D)
                        layer['batch_norm_out'] = nn.BatchNorm1d(D)
                    layers.append(layer)
            setattr(self, attr, layers)
        self.hid_dim = hid_dim
        self.lin = nn.Linear(hid_dim, hp['linear_hidden_dim_1'])
        self.lin2 = nn.Linear(hp['linear_hidden_dim_1'], hp['linear_hidden_dim_2'])
        self.lin3 = nn.Linear(hp['linear_hidden_dim_2'], self.num_classes)
        self.lin_dropout = nn.Dropout(p=hp['lin_dropout'])
        self.lin_dropout2 = nn.Dropout(p=hp['lin_dropout'])
        self.loss = nn.BCEWithLogitsLoss() if self.multilabel else nn.CrossEntropyLoss()
        self.lstm = LSTM(D, D, dropout=hp['lstm_dropout'], num_layers=hp['lstm_n_layers'],
                         aggregator=hp['lstm_aggregator'])
        if hp.get('ff_attn', False):                                   # S.py:179-183
            from .attention import AdditiveAttention
            self.attn_vector = Parameter(torch.zeros((hid_dim, 1), dtype=torch.float))
            nn.init.xavier_uniform_(self.attn_vector)
            self.attention = AdditiveAttention(hid_dim, hid_dim, half_operands=str(hp.get('embedding_dtype', 'fp32')).lower()
                                               in ('fp16', 'float16', 'half'))
        hp.setdefault('structure_similarity_fn', 'dtw')
        # predecessor rule of fastdtw's DP (0: the published pure-Python module; 1 / 2: the variants a compiled build may
        # implement; default config.DTW_TIE_ORDER = 2, argued there); reaches every DTW launch of the model, dense and sparse path
        hp.setdefault('dtw_tie_order', config.DTW_TIE_ORDER)
        if hp['dtw_tie_order'] not in (0, 1, 2):
            raise ValueError("hparams['dtw_tie_order'] must be 0, 1 or 2")
        # hparams['deterministic'] (default True): gradients by sorted segmented sums and per-row partials -- bit-
        # reproducible; False: float atomics (fewer launches per batch-sized step, sums in arbitrary order).
        # The model's own choice: stated for the duration of its forward (ops.deterministic), recorded by every op's
        # autograd context -- a second model with another setting does not change this one's backward.
        self._deterministic = bool(hp.get('deterministic', True))
        self.metric_scores = []
        self.to(self.device)
        # one-time start-up of the device libraries and of this library's code objects: paid here, once per process, not
        # in the middle of prepare_data (hparams['warm_up'] = False leaves it to the first pass)
        self.warm_up_s = ops.warm_up(self.device) if (self.device.type == 'cuda' and hp.get('warm_up', True)) else 0.0

    # ------------------------------------------------------------------ data -------------
    def read_data(self):
        """S.py:519-570: base graph -> CSR in HBM, subgraphs + labels, embedding table with the
        zero PAD row."""
        if self._memory is not None:
            return self._read_memory(self._memory)
        root = Path(config.PROJECT_ROOT)
        self.networkx_graph = load_graph(root / self.graph_path, self.device, root / self.degree_dict_path)
        (self.train_sub_G, self.train_sub_G_label, self.val_sub_G, self.val_sub_G_label, self.test_sub_G,
         self.test_sub_G_label) = subgraph_utils.read_subgraphs(root / self.subgraph_path)
        self.multilabel = isinstance(self.train_sub_G_label, list)
        self.multilabel_binarizer = None
        if self.multilabel:
            from sklearn.preprocessing import MultiLabelBinarizer
            self.multilabel_binarizer = MultiLabelBinarizer().fit(
                self.train_sub_G_label + self.val_sub_G_label + self.test_sub_G_label)
        if self.hparams.get('subset_data', False):
            bs = self.hparams['batch_size']
            for sp in ('train', 'val', 'test'):
                setattr(self, sp + '_sub_G', getattr(self, sp + '_sub_G')[:bs])
                setattr(self, sp + '_sub_G_label', getattr(self, sp + '_sub_G_label')[:bs])
        if self.multilabel:
            self.num_classes = max(max(l) for l in self.train_sub_G_label + self.val_sub_G_label + self.test_sub_G_label) + 1
        else:
            self.num_classes = int(torch.max(torch.cat((self.train_sub_G_label.view(-1), self.val_sub_G_label.view(-1),
                                                        self.test_sub_G_label.view(-1))))) + 1
        for sp in ('train', 'val', 'test'):                                   # ids become 1-based
            setattr(self, sp + '_sub_G', [[n + 1 for n in sg] for sg in getattr(self, sp + '_sub_G')])
        pre = torch.load(root / self.embedding_path, map_location='cpu')
        self.hparams['node_embed_size'] = pre.shape[1]
        table = torch.cat((torch.zeros(1, pre.shape[1]), pre.float()), 0)
        self.node_embeddings = nn.Embedding.from_pretrained(table, freeze=self.hparams['freeze_node_embeds'],
                                                            padding_idx=config.PAD_VALUE)

    def _read_memory(self, mem):
        """In-memory twin of read_data for synthetic inputs: ``mem`` = dict(graph=DeviceGraph,
        sub_G={'train': [...], 'val': [...], 'test': [...]} (1-based ids), labels={split: LongTensor},
        embeddings=(N, D) float tensor without the PAD row)."""
        self.networkx_graph = mem['graph']
        for sp in ('train', 'val', 'test'):
            setattr(self, sp + '_sub_G', mem['sub_G'].get(sp, []))
            setattr(self, sp + '_sub_G_label', mem['labels'].get(sp, torch.zeros(0, dtype=torch.int64)))
        self.multilabel, self.multilabel_binarizer = False, None
        self.num_classes = mem.get('num_classes') or int(max(int(l.max()) for l in mem['labels'].values() if l.numel() > 0)) + 1
        pre = mem['embeddings']
        self.hparams['node_embed_size'] = pre.shape[1]
        table = torch.cat((torch.zeros(1, pre.shape[1], device=pre.device), pre.float()), 0)
        self.node_embeddings = nn.Embedding.from_pretrained(table, freeze=self.hparams['freeze_node_embeds'],
                                                            padding_idx=config.PAD_VALUE)

    @classmethod
    def from_memory(cls, hparams, graph, sub_G, labels, embeddings, num_classes=None):
        """``num_classes``: give it when the labels at hand are one shard of the data (a class may be absent)."""
        return cls(hparams, None, None, None, 'similarities', None, None, None,
                   _memory=dict(graph=graph, sub_G=sub_G, labels=labels, embeddings=embeddings, num_classes=num_classes))

    # ------------------------------------------------------------------ components -------
    def initialize_cc_ids(self, subgraph_ids):
        """S.py:575-607 -> (S, max_n_cc, max_len_cc) int64 on the device.  Component order:
        by first node; nodes in subgraph order (the reference's is CPython-set order)."""
        subs = ops.Ragged.from_lists(subgraph_ids, self.device)
        labels = ops.cc_labels(self.networkx_graph, subs)
        return subgraph_utils.components_from_labels(subs.ptr, subs.nodes, labels, subs.max_len)

    def _half_table(self):
        """hparams['embedding_dtype'] in ('fp16', 'float16', 'half'): the table is additionally kept in
        IEEE half and the fused kernels read that copy (fp32 accumulate); the fp32 parameter stays the
        master the optimizer updates.  The copy is ONE persistent buffer refreshed in place.  The master's
        version counter cannot be trusted to say when: torch's fused Adam updates the parameter without
        moving it, and so does a replayed hipGraph.  Hence: a read with gradients enabled on a trainable
        table (a training forward: an optimizer step is expected to follow) always refreshes and leaves
        the copy marked dirty; a read without (validation, init_all_embeddings) refreshes a dirty copy
        once; while a step is being recorded the refresh is part of the recording, so every replay starts
        from the current master; ``invalidate_half_table`` marks the copy dirty by hand.  One conversion
        of the table (N x D read + half of it written) per training forward."""
        if str(self.hparams.get('embedding_dtype', 'fp32')).lower() not in ('fp16', 'float16', 'half'):
            return None
        w = self.node_embeddings.weight
        st = self.__dict__.get('_half_state')
        if st is None or st[1].device != w.device or st[1].shape != w.shape:
            st = self.__dict__['_half_state'] = [None, torch.empty(w.shape, dtype=torch.float16, device=w.device), True]
        capturing = w.is_cuda and torch.cuda.is_current_stream_capturing()
        training_read = torch.is_grad_enabled() and w.requires_grad
        if st[2] or st[0] != w._version or capturing or training_read:
            st[1].copy_(w.detach())
            st[0] = w._version
        st[2] = training_read or capturing
        return st[1]

    def invalidate_half_table(self):
        """The master table changed in a way nothing above can see (``.data`` edits, foreign in-place
        kernels): the next reader refreshes the half copy."""
        st = self.__dict__.get('_half_state')
        if st is not None:
            st[2] = True

    def _table(self):
        """The embedding table as the fused ops read it: inside ``forward`` a tapped alias whose
        consumers add their gradients into one shared buffer (ops.tap_table), else the parameter."""
        t = self.__dict__.get('_tapped_table')         # kept out of nn.Module's parameter registry
        if t is not None:
            return t
        half = self._half_table()
        return self.node_embeddings.weight if half is None else ops.tap_table(self.node_embeddings.weight, half)

    def initialize_cc_embeddings(self, cc_id_list, aggregator='sum'):
        """S.py:609-622 -> (S, C, D).  The padded rows are handed to the kernel as fixed-stride sets
        (PAD entries included): row 0 of the table is zero, so PAD adds nothing to a sum and competes
        in a max exactly as it does in the reference -- and no host synchronisation (a compaction
        would need the number of non-PAD entries on the host) sits in the per-step path."""
        S, C, L = cc_id_list.shape
        # (hotpath.prepare_pass converts the ids beside the sampling stages and hangs them on the tensor)
        ids = getattr(cc_id_list, '_sgnn_ids32', None)
        if ids is None or ids.numel() != S * C * L:
            ids = cc_id_list.to(self.device).reshape(S * C * L).to(torch.int32)
        # the fixed-stride set pointers depend on the padded shape only: kept
        memo = self.__dict__.setdefault('_fixed_stride_ptr', {})
        ptr = memo.get((S * C, L))
        if ptr is None:
            if len(memo) > 64:
                memo.clear()
            ptr = memo[(S * C, L)] = torch.arange(S * C + 1, dtype=torch.int64, device=self.device) * L
        sets = ops.Ragged(ptr, ids.contiguous() if ids.numel() else torch.zeros(1, dtype=torch.int32, device=self.device),
                          max_len=L)
        # (hotpath.prepare_sparse attaches the members' sorted order to the split's cc_ids tensor: kept across passes)
        pre = getattr(cc_id_list, '_sgnn_member_order', None)
        if pre is not None and pre[0].numel() != ids.numel():
            pre = None
        return ops.cc_embed(self._table(), sets, aggregator, padded_len=0, stride=L, presorted=pre).view(S, C, -1)

    def initialize_channel_embeddings(self, cc_embeddings, trainable=False):
        if trainable:
            return tuple(Parameter(cc_embeddings.detach().clone()) for _ in CC_SLOTS)
        return tuple(cc_embeddings for _ in CC_SLOTS)

    def init_all_embeddings(self, split='all', trainable=False, lazy=False):
        """S.py:624-650.  ``lazy`` (the per-pass caller, hotpath.install_pass): without ``trainable_cc`` nothing on the
        path reads the six per-split copies (forward recomputes the component embeddings from the table, S.py:238-247),
        so they are computed when somebody asks for the attribute instead of on every pass."""
        which = {'all': ('train', 'val', 'test'), 'train_val': ('train', 'val')}.get(split, (split,))
        pending = self.__dict__.setdefault('_lazy_cc_embed', set())
        for sp in which:
            if lazy and not (trainable and sp == 'train'):
                for nm in CC_SLOTS:
                    self.__dict__.pop('%s_%s_cc_embed' % (sp, nm), None)
                pending.add(sp)
                continue
            pending.discard(sp)
            with torch.no_grad():
                emb = self.initialize_cc_embeddings(getattr(self, sp + '_cc_ids'), self.hparams['cc_aggregator'])
            six = self.initialize_channel_embeddings(emb, trainable and sp == 'train')
            for nm, t in zip(CC_SLOTS, six):
                setattr(self, '%s_%s_cc_embed' % (sp, nm), t)

    def __getattr__(self, name):
        if name.endswith('_cc_embed'):
            sp = name.split('_', 1)[0]
            if sp in self.__dict__.get('_lazy_cc_embed', ()):
                self.init_all_embeddings(split=sp, trainable=False)
                return self.__dict__[name]
        return super().__getattr__(name)

    # ------------------------------------------------------------------ border sets -------
    def _sim_dir(self):
        d = Path(config.PROJECT_ROOT) / self.similarities_path
        d.mkdir(parents=True, exist_ok=True)
        return d

    def _cached(self, fname, compute, dtype=None):
        """the reference's .npy cache (S.py:726-742, 852-873, 893-978): load unless
        compute_similarities, else compute on the GPU and save."""
        if fname.exists() and not self.hparams['compute_similarities']:
            t = torch.from_numpy(np.load(fname, allow_pickle=True))
            return t.to(self.device) if dtype is None else t.to(self.device, dtype)
        t = compute()
        np.save(fname, t.detach().cpu().numpy())
        return t

    def initialize_border_sets(self, fname, cc_ids, radius, ego_graph_dict=None):
        t = subgraph_utils.border_sets(self.networkx_graph, cc_ids, radius, ego_dict_mode=ego_graph_dict is not None)
        np.save(fname, t.cpu().numpy())
        return t

    def get_border_sets(self, split):
        hp = self.hparams
        need = hp['use_neighborhood'] or (hp['use_structure'] and hp['structure_similarity_fn'] == 'edit_distance')
        splits = ('test',) if split == 'test' else ('train', 'val')
        if not need:
            for sp in splits:
                setattr(self, sp + '_N_border', None)
            return
        ego = (Path(config.PROJECT_ROOT) / self.ego_graph_path).exists()
        for sp in splits:
            f = self._sim_dir() / ('%d_%d_%s_border_set.npy' % (hp['neigh_sample_border_size'], config.PAD_VALUE, sp))
            cc = getattr(self, sp + '_cc_ids')
            setattr(self, sp + '_N_border', self._cached(
                f, lambda: subgraph_utils.border_sets(self.networkx_graph, cc, hp['neigh_sample_border_size'], ego)))

    # ------------------------------------------------------------------ similarities ------
    def compute_shortest_path_similarities(self, fname, shortest_paths, cc_ids):
        """S.py:752-781: (S, C, N) float32 = column-min of the APSP rows of each component."""
        S, C, L = cc_ids.shape
        sets = ops.Ragged.from_padded(cc_ids.reshape(S * C, L))
        sims = ops.sp_similarity_dense(shortest_paths, sets).view(S, C, -1)
        np.save(fname, sims.cpu().numpy())
        return sims

    def compute_structure_patch_similarities(self, degree_dict, fname, internal, cc_ids, sim_path=None,
                                             dataset_type=None, border_set=None):
        """S.py:783-833: (S, C, n_patches) = 1/(1+fastdtw(deg seq of CC, deg seq of anchor))."""
        if self.hparams['structure_similarity_fn'] != 'dtw':
            raise NotImplementedError
        S, C, L = cc_ids.shape
        g = self.networkx_graph
        use_dict = g.full_degree is not None
        a_sets, a_seq = gamma.degree_sequences(g, self.structure_anchors, internal, use_dict)
        c_sets, c_seq = gamma.degree_sequences(g, cc_ids.reshape(S * C, L), internal, use_dict)
        sims = gamma.dtw_similarity_matrix(c_sets, c_seq, a_sets, a_seq, self.hparams['dtw_tie_order']).view(S, C, -1)
        np.save(fname, sims.cpu().numpy())
        return sims

    def get_similarities(self, split):
        hp = self.hparams
        d = self._sim_dir()
        splits = ('test',) if split == 'test' else ('train', 'val')
        pad = config.PAD_VALUE
        if hp['use_position'] or hp['use_neighborhood']:
            apsp = None
            for sp in splits:
                f = d / ('%d_%s_similarities.npy' % (pad, sp))

                def compute(sp=sp, f=f):
                    nonlocal apsp
                    if apsp is None:
                        apsp = torch.from_numpy(np.load(Path(config.PROJECT_ROOT) / self.shortest_paths_path,
                                                        allow_pickle=True)).to(self.device, torch.float64).contiguous()
                    return self.compute_shortest_path_similarities(f, apsp, getattr(self, sp + '_cc_ids'))
                setattr(self, sp + '_neigh_pos_similarities', self._cached(f, compute))
        else:
            for sp in splits:
                setattr(self, sp + '_neigh_pos_similarities', None)
        if not hp['use_structure']:
            self.structure_anchors = None
            for sp in splits:
                setattr(self, sp + '_int_struc_similarities', None)
                setattr(self, sp + '_bor_struc_similarities', None)
            return
        g = self.networkx_graph
        tagp = '%d_%s_%d' % (hp['sample_walk_len'], hp['structure_patch_type'], hp['max_sim_epochs'])
        self.structure_anchors = self._cached(
            d / ('struc_patches_%s.npy' % tagp),
            lambda: aps.sample_structure_anchor_patches(hp, g, self.device, hp['max_sim_epochs']))
        tagw = '%d_%d_%s' % (hp['n_triangular_walks'], hp['random_walk_len'], tagp)
        views = aps.patch_node_views(self.structure_anchors)
        self.bor_structure_anchor_random_walks = self._cached(
            d / ('bor_struc_patch_random_walks_%s.npy' % tagw),
            lambda: aps.perform_random_walks(hp, g, self.structure_anchors, inside=False, views=views))
        self.int_structure_anchor_random_walks = self._cached(
            d / ('int_struc_patch_random_walks_%s.npy' % tagw),
            lambda: aps.perform_random_walks(hp, g, self.structure_anchors, inside=True, views=views))
        fn = '' if hp['structure_similarity_fn'] == 'dtw' else '_' + hp['structure_similarity_fn']
        for side, internal in (('int', True), ('bor', False)):
            for sp in splits:
                f = d / ('%s_struc_%s_%d%s_%s_similarities.npy' % (side, tagp, pad, fn, sp))
                cc = getattr(self, sp + '_cc_ids')
                setattr(self, '%s_%s_struc_similarities' % (sp, side), self._cached(
                    f, lambda f=f, cc=cc, internal=internal: self.compute_structure_patch_similarities(None, f, internal, cc)))

    # ------------------------------------------------------------------ prepare -----------
    def _prepare(self, split):
        hp, g = self.hparams, self.networkx_graph
        splits = ('test',) if split == 'test' else ('train', 'val')
        for sp in splits:
            setattr(self, sp + '_cc_ids', self.initialize_cc_ids(getattr(self, sp + '_sub_G')))
        self.init_all_embeddings(split=split, trainable=hp['trainable_cc'])
        self.get_border_sets(split=split)
        self.get_similarities(split=split)
        cc = {sp: getattr(self, sp + '_cc_ids', None) for sp in ('train', 'val', 'test')}
        nb = {sp: getattr(self, sp + '_N_border', None) for sp in ('train', 'val', 'test')}
        sub = {sp: getattr(self, sp + '_sub_G') for sp in ('train', 'val', 'test')}
        if hp['use_neighborhood']:
            ni, nbo = aps.init_anchors_neighborhood(split, hp, g, self.device, cc['train'], cc['val'], cc['test'],
                                                    nb['train'], nb['val'], nb['test'])
            if split == 'test' and getattr(self, 'anchors_neigh_int', None) is not None:
                self.anchors_neigh_int.update(ni)
                self.anchors_neigh_border.update(nbo)
            else:
                self.anchors_neigh_int, self.anchors_neigh_border = ni, nbo
        else:
            self.anchors_neigh_int, self.anchors_neigh_border = None, None
        if hp['use_position']:
            pi = aps.init_anchors_pos_int(split, hp, g, self.device, sub['train'], sub['val'], sub['test'])
            if split == 'test' and getattr(self, 'anchors_pos_int', None) is not None:
                self.anchors_pos_int.update(pi)
            else:
                self.anchors_pos_int = pi
            if split != 'test':
                self.anchors_pos_ext = aps.init_anchors_pos_ext(hp, g, self.device)
        elif split != 'test':
            self.anchors_pos_int, self.anchors_pos_ext = None, None
        if split != 'test':
            if hp['use_structure']:
                self.anchors_structure = aps.init_anchors_structure(hp, self.structure_anchors,
                                                                    self.int_structure_anchor_random_walks,
                                                                    self.bor_structure_anchor_random_walks)
            else:
                self.anchors_structure = None
        self._build_sim_cols()
        self.__dict__.pop('_resident', None)
        self._bump_generation()

    def prepare_data(self):
        """S.py:1024-1063."""
        self._prepare('train_val')
        self.__dict__['_sparse_prepared'] = False          # (border sets and similarity slabs are kept: a resample re-draws from them)

    def prepare_test_data(self):
        """S.py:994-1022."""
        self._prepare('test')

    # ------------------------------------------------------------------ batches -----------
    def _split_tensors(self, split):
        return (getattr(self, split + '_cc_ids'), getattr(self, split + '_N_border', None),
                getattr(self, split + '_neigh_pos_similarities', None),
                getattr(self, split + '_int_struc_similarities', None),
                getattr(self, split + '_bor_struc_similarities', None))

    def _resident_split(self, split):
        """Per-split tensors that never change (padded subgraph ids, labels), uploaded once."""
        cache = self.__dict__.setdefault('_resident', {})
        if split not in cache:
            subs = getattr(self, split + '_sub_G')
            L = max((len(s) for s in subs), default=1)
            ids = torch.zeros((len(subs), max(L, 1)), dtype=torch.int64)
            for i, s_ in enumerate(subs):
                ids[i, :len(s_)] = torch.as_tensor(s_)
            labels = getattr(self, split + '_sub_G_label')
            if self.multilabel:
                lab = torch.LongTensor(self.multilabel_binarizer.transform(labels))
            else:
                lab = labels.view(-1)
            # which columns of the padded rows hold a non-PAD entry, per subgraph, on the HOST: dropping a
            # batch's all-PAD columns (S.py:1098-1099,1109-1110) then needs no device round trip.  Not
            # just a width: with ego_graphs.txt a border row can hold the id 0 itself (su:168-174), which
            # the trim cannot tell from PAD -- an all-zero column is dropped wherever it lies.
            cc = getattr(self, split + '_cc_ids')
            nb = getattr(self, split + '_N_border', None)
            # (numpy on purpose: a torch CPU reduction of this size starts an intra-op parallel region, and on a
            # 128-thread host that costs ~10 ms per call -- measured: 40 ms eager steps instead of 5)
            w_cc = (cc != 0).any(dim=1).cpu().numpy()
            w_nb = (nb != 0).any(dim=1).cpu().numpy() if nb is not None else None
            cache[split] = (ids.to(self.device), lab.to(self.device), w_cc, w_nb)
        return cache[split]

    def make_batch(self, split, idx, trim=True):
        """Batch dict for subgraph indices ``idx`` (same keys as _pad_collate, S.py:1112-1114); every
        tensor is gathered on the device and nothing in here waits for the GPU.
        ``subgraph_ids`` keeps the split's padded width (forward never reads it).  ``trim=False``
        keeps the split's full padded widths too (``idx`` may then be a device tensor whose values
        the host never sees -- the form a recorded step needs)."""
        idx = torch.as_tensor(idx, dtype=torch.int64)
        didx = idx.to(self.device)
        cc, nb, npsim, isim, bsim = self._split_tensors(split)
        if not idx.is_cuda and idx.numel() and (int(idx.min()) < 0 or int(idx.max()) >= cc.shape[0]):
            raise IndexError('subgraph index out of range for split %r (%d subgraphs)' % (split, cc.shape[0]))
        sub_ids, lab, w_cc, w_nb = self._resident_split(split)

        # every per-subgraph tensor of the split that the batch reads, gathered in ONE launch (ops.index_rows_many): component
        # ids, border ids, the similarity rows (one slab or a dict of per-layer edge weights), subgraph ids, labels
        wanted = {'cc': cc, 'nb': nb, 'np': npsim, 'is': isim, 'bs': bsim, 'ids': sub_ids, 'lab': lab}
        flat = [(name, key, v) for name, t in wanted.items()
                for key, v in (t.items() if isinstance(t, dict) else [(None, t)]) if isinstance(v, torch.Tensor)]
        got = ops.index_rows_many([v for _, _, v in flat], didx) if didx.is_cuda else [v.index_select(0, didx) for _, _, v in flat]
        gathered = {(name, key): g for (name, key, _), g in zip(flat, got)}

        def pick(name):
            t = wanted[name]
            if t is None:
                return None
            if isinstance(t, dict):
                return {k: (gathered[(name, k)] if isinstance(v, torch.Tensor) else v.index_select(0, didx)) for k, v in t.items()}
            return gathered[(name, None)] if isinstance(t, torch.Tensor) else t.index_select(0, didx)
        batch_cc, batch_nb = pick('cc'), pick('nb')
        if trim:
            hidx = idx.cpu().numpy()

            def drop_pad_columns(t, nz):
                keep = nz[hidx].any(axis=0) if idx.numel() else np.zeros(t.shape[2], dtype=bool)
                k = int(keep.sum())
                if bool(keep[:k].all()):                       # left-justified rows: the kept columns are a prefix
                    return t[:, :, :k].contiguous()
                return t.index_select(2, torch.from_numpy(np.nonzero(keep)[0]).to(t.device))
            batch_cc = drop_pad_columns(batch_cc, w_cc)
            if nb is not None:
                batch_nb = drop_pad_columns(batch_nb, w_nb)
        return {'subgraph_ids': pick('ids'), 'cc_ids': batch_cc, 'N_border': batch_nb,
                'NP_sim': pick('np'), 'I_S_sim': pick('is'), 'B_S_sim': pick('bs'),
                'subgraph_idx': didx.view(-1, 1), 'label': pick('lab')}

    def _pad_collate(self, batch):
        """S.py:1068-1114 for a list of SubgraphDataset items."""
        sub, cc, nb, npsim, isim, bsim, idx, labels = zip(*batch)
        L = max(s.numel() for s in sub)
        sub_ids = torch.zeros((len(sub), L), dtype=torch.int64)
        for i, s in enumerate(sub):
            sub_ids[i, :s.numel()] = s
        st = lambda xs: None if any(x is None for x in xs) else torch.stack(xs)
        nbs = st(nb)
        return {'subgraph_ids': sub_ids, 'cc_ids': subgraph_utils.trim_zero_columns(torch.stack(cc)),
                'N_border': subgraph_utils.trim_zero_columns(nbs) if nbs is not None else None,
                'NP_sim': st(npsim), 'I_S_sim': st(isim), 'B_S_sim': st(bsim),
                'subgraph_idx': torch.stack(idx), 'label': torch.stack(labels).squeeze()}

    def _dataset(self, split):
        cc, nb, npsim, isim, bsim = self._split_tensors(split)
        return SubgraphDataset(getattr(self, split + '_sub_G'), getattr(self, split + '_sub_G_label'), cc, nb, npsim,
                               isim, bsim, self.multilabel, self.multilabel_binarizer)

    def train_dataloader(self):
        bs = self.hparams['batch_size']
        return _DeviceLoader(self, 'train', bs, shuffle=True, drop_last=bs <= len(self.train_sub_G))

    def val_dataloader(self):
        return _DeviceLoader(self, 'val', self.hparams['batch_size'], shuffle=False, drop_last=False)

    def test_dataloader(self):
        self.prepare_test_data()
        return _DeviceLoader(self, 'test', self.hparams['batch_size'], shuffle=False, drop_last=False)

    # ------------------------------------------------------------------ forward -----------
    def run_mpn_layer(self, dataset_type, mpn_fn, subgraph_ids, subgraph_idx, cc_ids, cc_embeds, cc_embed_mask, sims,
                      layer_num, channel, inside=True):
        """S.py:195-223, reference-shaped: materialise (B,C,A,D) anchor embeddings, then SG_MPN."""
        ap, am, ae = aps.get_anchor_patches(dataset_type, self.hparams, self.networkx_graph, self.node_embeddings,
                                            subgraph_idx, cc_ids, cc_embed_mask, self.lstm, self.anchors_neigh_int,
                                            self.anchors_neigh_border, self.anchors_pos_int, self.anchors_pos_ext,
                                            self.anchors_structure, layer_num, channel, inside, self.device)
        idx = self.anchors_structure[layer_num][1] if channel == 'structure' else None
        return mpn_fn(self.networkx_graph, sims, cc_ids, cc_embeds, cc_embed_mask, ap, ae, am, idx)

    def _run_mpn_layer_fused(self, dataset_type, mpn_fn, sidx, cc_embeds, cc_embed_mask, sims, layer_num, channel,
                             inside, need_out=True, defer=False, need_pos=True, defer_update=False):
        """Same layer without the (B,C,A,D) tensor: anchors are gathered inside the kernel."""
        B, C, _ = cc_embeds.shape
        E = self._table()
        whole = getattr(sidx, '_sgnn_identity', None)

        def rows_of(t):
            return t if whole is not None and t.shape[0] == whole else t.index_select(0, sidx)

        def layer_rows(src, tag):
            """rows_of(src[dataset_type][layer_num]) -- for a batch, the rows of ALL layers' anchor tensors in one gather (they
            are stacked once per preparation): a 4-layer step made 12 gathers of a few KB each, now 3."""
            t = src[dataset_type][layer_num]
            L = self.hparams['n_layers']
            cache = self.__dict__.get('_fwd_cache')
            if (whole is not None and t.shape[0] == whole) or L < 2 or cache is None:
                return rows_of(t)
            key = ('anchor_rows', tag, dataset_type)
            if key not in cache:
                gen = self.__dict__.get('_prep_generation', 0)
                stacks = self.__dict__.setdefault('_anchor_stacks', {})
                ent = stacks.get(key)
                per_layer = [src[dataset_type][l] for l in range(L)]
                if ent is None or ent[0] != gen or any(a is not b for a, b in zip(ent[1], per_layer)):
                    if any(p.shape != per_layer[0].shape or p.dtype != per_layer[0].dtype for p in per_layer) \
                            or (t.is_cuda and torch.cuda.is_current_stream_capturing()):
                        # (a stack made while a step is being recorded would live in the recording's memory pool and be re-made
                        # by every replay: the per-layer gathers serve that recording; eager steps before it build the stack)
                        cache[key] = None
                        return rows_of(t)
                    ent = stacks[key] = (gen, per_layer, torch.stack(per_layer, 0))
                cache[key] = ent[2].index_select(1, sidx.reshape(-1))                 # (L, B, ...)
            got = cache[key]
            return rows_of(t) if got is None else got[layer_num]
        # NP_sim is either the reference's dense (B,C,N) slab (column = anchor id - 1) or, for
        # graphs where that slab cannot exist, a dict of already-gathered (B,C,A) edge weights
        # keyed (channel tag, side, layer) -- see hotpath.py
        per_edge = isinstance(sims, dict)
        if per_edge:
            sims = sims[(channel[0].upper(), 'in' if inside else 'out', layer_num)]
        if channel == 'neighborhood':
            src = self.anchors_neigh_int if inside else self.anchors_neigh_border
            anchors = src[dataset_type][layer_num]
            ids = layer_rows(src, 'N_in' if inside else 'N_out').reshape(B * C, -1).contiguous()
            # (hotpath.prepare_pass: the gradient-independent half of this layer's table-gradient scatter, for these anchors)
            plan = self.__dict__.get('_mpn_edge_plans', {}).get(dataset_type, {}).get(('N', inside, layer_num))
            if plan is not None and not (whole is not None and plan['anchors'] is anchors and per_edge):
                plan = None
            return mpn_fn.forward_fused(sims, cc_embeds, cc_embed_mask, src=ops.SRC_GATHER, x=E, ids=ids,
                                        sims_per_edge=per_edge, need_out=need_out, defer_readout=defer, need_pos=need_pos,
                                        edge_plan=plan, defer_update=defer_update)
        if channel == 'position':
            if inside:
                ids = layer_rows(self.anchors_pos_int, 'P_in').contiguous()
                return mpn_fn.forward_fused(sims, cc_embeds, cc_embed_mask, src=ops.SRC_GATHER, x=E, ids=ids, id_div=C,
                                            sims_per_edge=per_edge, need_out=need_out, defer_readout=defer, need_pos=need_pos,
                                            defer_update=defer_update)
            ids = self.anchors_pos_ext[layer_num]
            X = ops.gather_rows(E, ids)
            return mpn_fn.forward_fused(sims, cc_embeds, cc_embed_mask, src=ops.SRC_SHARED, x=X, ids=ids,
                                        sims_per_edge=per_edge, need_out=need_out, defer_readout=defer, need_pos=need_pos,
                                        defer_update=defer_update)
        X = self._structure_anchor_embeddings(layer_num, E)[0 if inside else 1]
        return mpn_fn.forward_fused(sims, cc_embeds, cc_embed_mask, src=ops.SRC_SHARED, x=X,
                                    sim_col=self._sim_col_cache[layer_num], need_out=need_out, defer_readout=defer,
                                    need_pos=need_pos, defer_update=defer_update)

    def _structure_anchor_embeddings(self, layer_num, E):
        """aps:413-433 for the internal and the border walks of a layer's structure patches in ONE pass over the LSTM (same
        parameters, independent sequences): (X_internal, X_border), each (patches, D); kept for the forward in progress."""
        cache = self.__dict__.get('_fwd_cache')
        key = ('S_X', layer_num)
        if cache is not None and key in cache:
            return cache[key]
        L = self.hparams['n_layers']
        if L > 1 and cache is not None:
            # The walks of ALL layers in one pass over the LSTM: the anchor embeddings of a layer depend on the table and the LSTM's
            # parameters only (aps:413-433), not on what the layers below computed, and the LSTM is shared by the layers -- so the
            # n_layers x (internal, border) batches are one batch.  Same arithmetic per sequence; one recurrence launch per LSTM
            # layer and direction pair instead of n_layers of them, one set of weight-gradient contractions, no accumulation of
            # n_layers partial weight gradients (HPO-METAB stand-in, 4 layers: 8 -> 2 recurrences each way).
            if ('S_X_all',) not in cache:
                walks = self._all_structure_walks()
                Xall = aps.aggregate_structure_anchor_patch(self.hparams, self.networkx_graph, self.lstm, self.node_embeddings,
                                                            walks, walks, None, self.device, table=E)
                off = 0
                for l in range(L):
                    n_l = self.anchors_structure[l][0].shape[0]
                    cache[('S_X', l)] = (Xall[off:off + n_l], Xall[off + n_l:off + 2 * n_l])
                    off += 2 * n_l
                cache[('S_X_all',)] = True
            return cache[key]
        patches, indices, int_rw, bor_rw = self.anchors_structure[layer_num]
        n = patches.shape[0]
        both = getattr(int_rw, '_sgnn_both', None)               # hotpath.prepare_pass: stacked, with the lookup's sort
        if both is None or both.shape[0] != 2 * n or both.device != E.device:
            both = torch.cat([int_rw.to(self.device), bor_rw.to(self.device)], 0)
        X = aps.aggregate_structure_anchor_patch(self.hparams, self.networkx_graph, self.lstm, self.node_embeddings, both, both,
                                                 None, self.device, table=E)
        out = (X[:n], X[n:])
        if cache is not None:
            cache[key] = out
        return out

    def _all_structure_walks(self):
        """[layer 0: internal walks, border walks | layer 1: ... ] of the sampled structure patches, stacked once per set of
        anchors (hung on layer 0's internal walks as ``_sgnn_all``, with the sort its embedding lookup's backward needs when the
        backward is the deterministic one): built outside any recording by ``_build_sim_cols`` -- or here, the first time."""
        a = self.anchors_structure
        first = a[0][2]
        walks = getattr(first, '_sgnn_all', None)
        want = sum(2 * a[l][0].shape[0] for l in range(self.hparams['n_layers']))
        if walks is None or walks.shape[0] != want or walks.device != self.node_embeddings.weight.device:
            walks = torch.cat([t.to(self.device) for l in range(self.hparams['n_layers']) for t in (a[l][2], a[l][3])], 0)
            if self._deterministic and walks.is_cuda:
                ops.presort_ids(walks, self.networkx_graph.max_id)
            try:
                first._sgnn_all = walks
            except AttributeError:
                pass
        return walks

    def forward(self, dataset_type, N_I_cc_embed, N_B_cc_embed, S_I_cc_embed, S_B_cc_embed, P_I_cc_embed,
                P_B_cc_embed, subgraph_ids, cc_ids, subgraph_idx, NP_sim, I_S_sim, B_S_sim):
        """S.py:225-312."""
        hp = self.hparams
        fused = hp.get('fused_forward', True)
        self.__dict__['_tapped_table'] = ops.tap_table(self.node_embeddings.weight, self._half_table()) if fused else None
        self.__dict__['_fwd_cache'] = {}
        try:
            with ops.deterministic(self._deterministic):
                return self._forward(dataset_type, N_I_cc_embed, N_B_cc_embed, S_I_cc_embed, S_B_cc_embed, P_I_cc_embed,
                                     P_B_cc_embed, subgraph_ids, cc_ids, subgraph_idx, NP_sim, I_S_sim, B_S_sim)
        finally:
            self.__dict__['_tapped_table'] = None
            self.__dict__['_fwd_cache'] = None

    def _forward(self, dataset_type, N_I_cc_embed, N_B_cc_embed, S_I_cc_embed, S_B_cc_embed, P_I_cc_embed,
                 P_B_cc_embed, subgraph_ids, cc_ids, subgraph_idx, NP_sim, I_S_sim, B_S_sim):
        hp = self.hparams
        fused = hp.get('fused_forward', True)
        ops.drop_lazy_mpn()                                 # (launches a forward that did not finish left queued)
        init_cc_embeds = self.initialize_cc_embeddings(cc_ids, hp['cc_aggregator'])
        sidx = subgraph_idx.view(-1)
        # hotpath.full_split_batch: the batch IS the split, in order -- selecting its rows would be a copy
        sidx._sgnn_identity = getattr(subgraph_idx, '_sgnn_identity', None)
        given = {'N_I': N_I_cc_embed, 'N_B': N_B_cc_embed, 'S_I': S_I_cc_embed, 'S_B': S_B_cc_embed,
                 'P_I': P_I_cc_embed, 'P_B': P_B_cc_embed}
        state = {}
        for nm in CC_SLOTS:
            state[nm] = torch.index_select(given[nm], 0, sidx) if hp['trainable_cc'] else init_cc_embeds
        B, C, _ = init_cc_embeds.shape
        made = getattr(cc_ids, '_sgnn_mask', None)                    # (hotpath.prepare_pass makes both beside the sampling stages)
        made8 = getattr(cc_ids, '_sgnn_mask_u8', None)
        if made is not None and made8 is not None and tuple(made.shape) == (B, C) and made8.numel() == B * C and made.device == cc_ids.device:
            cc_embed_mask = made.view(B, C)                            # (a fresh view object: the attribute below must not land on the kept tensor)
            cc_embed_mask._sgnn_u8 = made8
        else:
            cc_embed_mask = cc_ids[:, :, 0] != config.PAD_VALUE      # (the first column only: a component is real iff it has a member)
            cc_embed_mask._sgnn_u8 = cc_embed_mask.reshape(-1).to(torch.uint8)
        bn = hp.get('batch_norm', False)
        # without an attention read-out or a gathered head the channel outputs are consumed only as their masked sum over a
        # subgraph's components: every piece is summed straight into its slot of the subgraph embedding
        slots = fused and not hp.get('ff_attn', False) and not hp.get('dp_gather_embeddings', False)
        outputs = []
        for l in range(hp['n_layers']):
            # the bodies of a layer (up to three channels x two sides) read the layer below only: their message-passing kernels
            # run one after the other, their update layers TOGETHER (ops.update_layers: one launch each way for a batch-sized step)
            bodies = []
            for channel, tag, flag, attr in CHANNELS:
                if not hp[flag]:
                    continue
                layer = getattr(self, attr)[l]
                pick = 0 if tag == 'N' else 1                      # N adds CC embeddings, P/S their read-outs
                # the updated component embeddings of a P / S layer only feed the next layer: not computed for the last
                need_out = pick == 0 or l + 1 < hp['n_layers']
                for inside, side, name, bnname in ((True, 'I', 'internal', 'batch_norm'), (False, 'B', 'border', 'batch_norm_out')):
                    sims = NP_sim if tag != 'S' else (I_S_sim if inside else B_S_sim)
                    slot = tag + '_' + side
                    if fused:
                        o, p = self._run_mpn_layer_fused(dataset_type, layer[name], sidx, state[slot], cc_embed_mask,
                                                         sims, l, channel, inside, need_out=need_out, defer=slots, need_pos=pick == 1,
                                                         defer_update=True)
                    else:
                        o, p = self.run_mpn_layer(dataset_type, layer[name], subgraph_ids, subgraph_idx, cc_ids,
                                                  state[slot], cc_embed_mask, sims, layer_num=l, channel=channel,
                                                  inside=inside)
                    bodies.append([slot, pick, o, p, layer[bnname] if bn else None])
            pending = [b for b in bodies if isinstance(b[2], ops.PendingUpdate)]
            ops.flush_lazy_mpn()                       # the layer's queued message-passing launches, as one
            for b, o in zip(pending, ops.update_layers([b[2] for b in pending])):
                b[2] = o.view(b[2].shape[0], b[2].shape[1], -1)
            for slot, pick, o, p, bn_layer in bodies:
                if bn and o is not None:
                    o = bn_layer(o.reshape(B * C, -1)).view(B, C, -1)
                state[slot] = o
                outputs.append((o, p)[pick])
        if slots:
            subgraph_embedding = ops.subgraph_embedding([init_cc_embeds] + outputs, cc_embed_mask._sgnn_u8, B, C)
            return self._head(subgraph_embedding)
        all_cc_embeds = torch.cat([init_cc_embeds] + outputs, dim=-1)
        if hp.get('dp_gather_embeddings', False):
            # data parallelism over subgraph shards: the one exchange of the data path.  Every rank has
            # computed the per-component channel embeddings of ITS subgraphs; the all-gather (RCCL over xGMI)
            # assembles the global batch on every rank and the read-out + MLP head + loss below run
            # replicated on it -- identical on all ranks, so head gradients need no reduction, and each
            # rank's slice of d loss / d embeddings flows back into its own channels
            # (dist.gather_rows_replicated).  The caller supplies the labels of the global batch.
            from . import dist as sdist
            H = all_cc_embeds.shape[-1]
            all_cc_embeds = sdist.gather_rows_replicated(all_cc_embeds.reshape(B * C, H)).view(-1, C, H)
            cc_embed_mask = sdist.all_gather_rows(cc_embed_mask.reshape(B * C, 1).to(torch.uint8),
                                                  equal_rows=True).view(-1, C).bool()
        if hp.get('ff_attn', False):                                    # S.py:298-301
            batched_attn = self.attn_vector.squeeze().unsqueeze(0).repeat(all_cc_embeds.shape[0], 1)
            attn_weights = self.attention(batched_attn, all_cc_embeds, cc_embed_mask)
            subgraph_embedding = subgraph_utils.weighted_sum(all_cc_embeds, attn_weights)
        else:
            subgraph_embedding = subgraph_utils.masked_sum(all_cc_embeds, cc_embed_mask.unsqueeze(-1), dim=1)
        return self._head(subgraph_embedding)

    def _dropout_rng(self):
        """{seed, step} of the fused head's dropout masks, int64 (2,) on the device: the seed is torch's CUDA seed at the first
        training forward (``torch.manual_seed`` / ``torch.cuda.manual_seed`` as the caller set it: two same-seed models draw the
        same masks, whatever ran before them in the process), the step is advanced by every training forward ON THE DEVICE, so
        a step replayed from a hipGraph draws fresh masks.  Created outside any recording (an upload)."""
        st = self.__dict__.get('_head_rng')
        if st is None or st.device != self.node_embeddings.weight.device:
            if torch.cuda.is_current_stream_capturing():
                raise RuntimeError('the dropout state of the head must exist before a step is recorded (run one eager training step first)')
            seed = int(torch.cuda.initial_seed()) & ((1 << 63) - 1)
            st = self.__dict__['_head_rng'] = torch.tensor([seed, 0], dtype=torch.int64, device=self.node_embeddings.weight.device)
        return st

    def _head(self, subgraph_embedding):
        """S.py:304-312.  One fused launch behind the first layer's GEMM (ops.fused_head: csrc/head.hip) when the widths fit;
        with the step's labels at hand (training_step leaves them in ``_head_labels``) the same launch computes the loss and
        the accuracy and leaves them in ``_head_result``."""
        hp = self.hparams
        labels = self.__dict__.pop('_head_labels', None)
        x = subgraph_embedding
        if (hp.get('fused_forward', True) and hp.get('fused_head', True) and x.is_cuda and x.dtype == torch.float32 and x.dim() == 2
                and x.shape[0] > 0 and ops.head_supported(self.lin.out_features, self.lin2.out_features, self.lin3.out_features)
                and all(l.bias is not None for l in (self.lin, self.lin2, self.lin3))):
            p = float(self.lin_dropout.p) if self.training else 0.0
            if labels is not None and (labels.dim() != 1 or labels.shape[0] != x.shape[0] or labels.dtype != torch.int64):
                labels = None
            logits, loss, acc = ops.fused_head(x, self.lin, self.lin2, self.lin3, labels, p, self._dropout_rng() if p > 0 else None)
            if labels is not None:
                self.__dict__['_head_result'] = (loss, acc)
            return logits
        h = self.lin_dropout(F.relu(ops.linear(subgraph_embedding, self.lin.weight, self.lin.bias)))
        h = self.lin_dropout2(F.relu(ops.linear(h, self.lin2.weight, self.lin2.bias)))
        return ops.linear(h, self.lin3.weight, self.lin3.bias)

    # ------------------------------------------------------------------ steps -------------
    def _forward_batch(self, split, batch):
        if self.hparams['trainable_cc']:
            e = {nm: getattr(self, '%s_%s_cc_embed' % (split, nm)) for nm in CC_SLOTS}
        else:                                   # not read by forward (S.py:238-247)
            e = {nm: None for nm in CC_SLOTS}
        return self.forward(split, e['N_I'], e['N_B'], e['S_I'], e['S_B'], e['P_I'], e['P_B'], batch['subgraph_ids'],
                            batch['cc_ids'], batch['subgraph_idx'], batch['NP_sim'], batch['I_S_sim'], batch['B_S_sim'])

    def _loss(self, logits, labels):
        if labels.dim() == 0:
            labels = labels.unsqueeze(-1)
        if self.multilabel:
            return self.loss(logits.squeeze(1), labels.type_as(logits)), labels
        return self.loss(logits, labels), labels

    def training_step(self, train_batch, batch_idx):
        labels = train_batch['label'].squeeze(-1)
        fusable = not self.multilabel and labels.is_cuda and labels.dim() == 1 and type(self.loss) is nn.CrossEntropyLoss \
            and self.hparams.get('fused_forward', True)
        self.__dict__.pop('_head_result', None)
        if fusable:
            self.__dict__['_head_labels'] = labels          # the head computes loss + accuracy in its own launch (_head)
        try:
            logits = self._forward_batch('train', train_batch)
        finally:
            self.__dict__.pop('_head_labels', None)
        done = self.__dict__.pop('_head_result', None)
        if done is not None:
            loss, acc = done
            return {'loss': loss, 'log': {'train_loss': loss, 'train_acc': acc}}
        if not self.multilabel and logits.is_cuda and logits.dim() == 2 and labels.dim() == 1 and logits.dtype == torch.float32 \
                and type(self.loss) is nn.CrossEntropyLoss and self.hparams.get('fused_forward', True):
            # cross entropy + accuracy in one pass over the logits (ops.cross_entropy_with_accuracy): the library's nll
            # reduction runs on one workgroup (48 us at 50k rows) and the pair is 12 launches
            loss, acc = ops.cross_entropy_with_accuracy(logits, labels)
            return {'loss': loss, 'log': {'train_loss': loss, 'train_acc': acc}}
        loss, labels = self._loss(logits, labels)
        acc = subgraph_utils.calc_accuracy(logits, labels, multilabel_binarizer=self.multilabel_binarizer)
        return {'loss': loss, 'log': {'train_loss': loss, 'train_acc': acc}}

    def val_test_step(self, batch, batch_idx, is_test=False):
        p = 'test' if is_test else 'val'
        labels = batch['label'].squeeze(-1)
        logits = self._forward_batch(p, batch)
        return self.val_test_outputs(p, logits, labels)

    def val_test_outputs(self, p, logits, labels):
        """What val_test_step makes of a batch's logits and labels (S.py:380-406): loss, accuracy, macro F1 + the logits and
        labels themselves for the epoch's metrics.  Device tensors or host copies alike (the trainer's recorded validation
        forward hands over host copies of a whole epoch's batches: one transfer instead of three read-backs per batch)."""
        loss, labels = self._loss(logits, labels)
        acc = subgraph_utils.calc_accuracy(logits, labels, multilabel_binarizer=self.multilabel_binarizer)
        f1 = subgraph_utils.calc_f1(logits, labels, avg_type='macro', multilabel_binarizer=self.multilabel_binarizer)
        return {p + '_loss': loss, p + '_acc': acc, p + '_macro_f1': f1, p + '_logits': logits, p + '_labels': labels}

    def validation_step(self, val_batch, batch_idx):
        return self.val_test_step(val_batch, batch_idx, is_test=False)

    def test_step(self, test_batch, batch_idx):
        return self.val_test_step(test_batch, batch_idx, is_test=True)

    def _epoch_metrics(self, outputs, p):
        roc_auc_score = subgraph_utils.roc_auc              # sklearn's values without its per-call argument validation
        if outputs and outputs[0][p + '_logits'].is_cuda:
            ops.poll_index_errors(block=True)           # (a device-side batch index outside its split: reported here at the latest)
        logits = torch.cat([x[p + '_logits'] for x in outputs], dim=0).detach()
        labels = torch.cat([x[p + '_labels'] for x in outputs], dim=0)
        mb = self.multilabel_binarizer
        logs = {p + '_loss': torch.stack([x[p + '_loss'] for x in outputs]).mean().detach().cpu(),
                p + '_micro_f1': subgraph_utils.calc_f1(logits, labels, 'micro', mb).squeeze(),
                p + '_macro_f1': subgraph_utils.calc_f1(logits, labels, 'macro', mb).squeeze(),
                p + '_acc': subgraph_utils.calc_accuracy(logits, labels, mb).squeeze().cpu()}
        avg_acc = torch.stack([x[p + '_acc'] for x in outputs]).mean().cpu()
        avg_f1 = torch.stack([x[p + '_macro_f1'] for x in outputs]).mean()
        if p == 'val':
            logs['avg_val_acc'], logs['avg_macro_f1'] = avg_acc, avg_f1
        else:
            logs['avg_test_acc'], logs['test_avg_macro_f1'] = avg_acc, avg_f1
        lc, gc = labels.cpu(), logits.cpu()
        try:
            if self.multilabel:
                logs[p + '_auroc'] = roc_auc_score(lc.numpy(), torch.sigmoid(gc).numpy(), multi_class='ovr')
            elif len(torch.unique(lc)) == 2:
                logs[p + '_auroc'] = roc_auc_score(lc.numpy(), F.softmax(gc, dim=1)[:, 1].numpy())
            else:
                logs[p + '_auroc'] = roc_auc_score(lc.numpy(), F.softmax(gc, dim=1).numpy(), multi_class='ovr')
            onehot = (lc if self.multilabel else F.one_hot(lc, num_classes=gc.shape[1])).numpy()
            score = (torch.sigmoid(gc) if self.multilabel else gc).numpy()
            for c in range(gc.shape[1]):
                logs['%s_auroc_class_%d' % (p, c)] = roc_auc_score(onehot[:, c], score[:, c])
        except ValueError:
            logs.setdefault(p + '_auroc', float('nan'))       # a class absent from the split
        return logs

    def validation_epoch_end(self, outputs):
        """S.py:408-464."""
        logs = self._epoch_metrics(outputs, 'val')
        hp = self.hparams
        if not hp['trainable_cc']:
            # (S.py:446-448 recomputes the six per-split copies of the component embeddings here; without trainable_cc nothing on
            # the path reads them -- forward recomputes from the table, S.py:238-247 -- so they are made when somebody asks)
            self.init_all_embeddings(split='train_val', trainable=False, lazy=True)
        if hp['resample_anchor_patches']:
            # a fresh tape stream for the new draws; hparams['seed'] (already written to hyperparams.json
            # by the caller, and what prepare_test_data draws from) stays what the caller set
            self.__dict__['_resample_epoch'] = self.__dict__.get('_resample_epoch', 0) + 1
            self._prepare_anchors_only()
        self.metric_scores.append(logs)
        return {'avg_val_loss': logs['val_loss'], 'log': logs}

    def _prepare_anchors_only(self):
        """S.py:453-460: new anchor draws on the prepared sets (resample_anchor_patches); the draws read
        the tape streams of resample epoch ``_resample_epoch``."""
        hp, g = self.hparams, self.networkx_graph
        ep = self.__dict__.get('_resample_epoch', 0)
        if self.__dict__.get('_sparse_prepared', False):
            # prepared by hotpath.prepare_sparse: no border sets and no N x N similarity slabs are kept to re-draw from -- the
            # similarities exist for the drawn anchors only -- so a resample is a new sparse pass over each prepared split, its
            # N / P draws and structure picks keyed by the resample epoch (the structure patches and walks: the same tape items)
            from . import hotpath
            for sp in ('train', 'val'):
                if getattr(self, sp + '_cc_ids', None) is not None and len(getattr(self, sp + '_sub_G', [])) > 0:
                    hotpath.prepare_sparse(self, sp)
            return
        new = {}
        if hp['use_neighborhood']:
            new['anchors_neigh_int'], new['anchors_neigh_border'] = aps.init_anchors_neighborhood(
                'train_val', hp, g, self.device, self.train_cc_ids, self.val_cc_ids, None, self.train_N_border,
                self.val_N_border, None, epoch=ep)
        if hp['use_position']:
            new['anchors_pos_int'] = aps.init_anchors_pos_int('train_val', hp, g, self.device, self.train_sub_G,
                                                               self.val_sub_G, self.test_sub_G, epoch=ep)
            new['anchors_pos_ext'] = aps.init_anchors_pos_ext(hp, g, self.device, epoch=ep)
        if hp['use_structure']:
            new['anchors_structure'] = aps.init_anchors_structure(hp, self.structure_anchors,
                                                                  self.int_structure_anchor_random_walks,
                                                                  self.bor_structure_anchor_random_walks, epoch=ep)
        if hp.get('resample_in_place', True) and self.device.type == 'cuda' and self._resample_in_place(new):
            return                                     # same tensors, new draws: recorded steps stay valid (no generation bump)
        for k, v in new.items():
            setattr(self, k, v)
        self._build_sim_cols()
        self._bump_generation()

    def _resample_in_place(self, new):
        """The new draws COPIED into the tensors of the old ones where every shape agrees -- anchor counts and padded widths do
        not change between resamples -- so that a recorded training / validation step (graph_step.py), which reads those
        addresses, stays valid: a resample then costs its sampling launches + a few copies instead of a device synchronisation
        and two recordings per epoch.  Also refreshed, in place: what was DERIVED from the old draws and is read by address -- the
        structure similarity columns, the per-layer anchor stacks, the stacked walks of all layers with their sorted order.
        -> False (nothing touched beyond copies that are about to be replaced anyway) when some shape changed: the caller
        assigns the new containers and bumps the generation, as before."""
        from . import hotpath
        for k, v in new.items():
            if getattr(self, k, None) is None:
                return False
        replaced, memo = [], {}
        kept = {k: hotpath._copy_into(getattr(self, k), v, k, replaced, memo) for k, v in new.items()}
        if replaced:
            return False
        for k, v in kept.items():
            setattr(self, k, v)
        src = getattr(self, 'anchors_structure', None)
        if src is not None and 'anchors_structure' in new:
            cols = self.sim_cols_of(src)
            cur = getattr(self, '_sim_col_cache', None) or {}
            if set(cols) != set(cur) or any(cols[l].shape != cur[l].shape for l in cols):
                return False
            for l in cols:
                cur[l].copy_(cols[l])
            self.__dict__['_sim_cols_src'] = src
            first = src[0][2]
            L = self.hparams['n_layers']
            stacked = getattr(first, '_sgnn_all', None)
            if stacked is not None:
                fresh = torch.cat([t for l in range(L) for t in (src[l][2], src[l][3])], 0)
                if fresh.shape != stacked.shape:
                    return False
                stacked.copy_(fresh)
                old = getattr(stacked, '_sgnn_sorted', None)
                if old is not None:
                    k32 = stacked.reshape(-1).to(torch.int32)
                    sk, order = ops.sort_edges_by_key(k32.contiguous(), self.networkx_graph.max_id)
                    old[0].copy_(sk), old[1].copy_(order), old[2].copy_(k32)
            for l in range(L):
                both = getattr(src[l][2], '_sgnn_both', None)
                if both is not None:
                    fresh = torch.cat([src[l][2], src[l][3]], 0)
                    if fresh.shape != both.shape:
                        return False
                    both.copy_(fresh)
                    old = getattr(both, '_sgnn_sorted', None)
                    if old is not None:
                        k32 = both.reshape(-1).to(torch.int32)
                        sk, order = ops.sort_edges_by_key(k32.contiguous(), self.networkx_graph.max_id)
                        old[0].copy_(sk), old[1].copy_(order), old[2].copy_(k32)
        for key, ent in (self.__dict__.get('_anchor_stacks') or {}).items():
            ent[2].copy_(torch.stack(ent[1], 0))          # (ent[1]: the per-layer tensors, refreshed in place above)
        for name in ('anchors_pos_ext',):
            for v in (getattr(self, name, None) or {}).values():
                old = getattr(v, '_sgnn_sorted', None)
                if old is not None:
                    k32 = v.reshape(-1).to(torch.int32)
                    sk, order = ops.sort_edges_by_key(k32.contiguous(), self.networkx_graph.max_id)
                    old[0].copy_(sk), old[1].copy_(order), old[2].copy_(k32)
        return True

    def _bump_generation(self):
        """Every replacement of tensors a recorded step reads (prepared sets, anchors, similarity rows)
        moves this counter; graph_step.CapturedTrainStep compares it instead of object ids."""
        self.__dict__['_prep_generation'] = self.__dict__.get('_prep_generation', 0) + 1

    def sim_cols_of(self, anchors_structure):
        """{layer: device index tensor} for a set of sampled structure patches (an upload from host lists)."""
        return {l: torch.as_tensor(indices, dtype=torch.int64, device=self.device)
                for l, (_, indices, _, _) in anchors_structure.items()}

    def set_sim_cols(self, anchors_structure, cols):
        self.__dict__['_sim_cols_src'] = anchors_structure
        self._sim_col_cache = cols

    def _build_sim_cols(self):
        """Per layer, the columns of the S similarity rows its sampled patches read (S.py:206-210),
        resident on the device so that forward never uploads anything."""
        src = getattr(self, 'anchors_structure', None)
        if src is not None and self.hparams['n_layers'] > 1 and self.hparams.get('fused_forward', True) and torch.is_tensor(src[0][2]) \
                and not (torch.cuda.is_available() and torch.cuda.is_current_stream_capturing()):
            self._all_structure_walks()                       # (the layers' walks stacked for the one LSTM pass of forward)
        if src is not None and self.__dict__.get('_sim_cols_src') is src:
            return                                            # already uploaded for these patches
        self.set_sim_cols(src, self.sim_cols_of(src) if src is not None else {})

    def test_epoch_end(self, outputs):
        """S.py:466-504."""
        logs = self._epoch_metrics(outputs, 'test')
        self.test_results = logs
        return {'avg_test_loss': logs['test_loss'], 'log': logs}

    # ------------------------------------------------------------------ optimisation ------
    def configure_optimizers(self):
        # S.py:1156-1161; the fused implementation is one kernel over all parameters instead of ~10
        # multi-tensor launches per step (same update rule)
        return torch.optim.Adam(self.parameters(), lr=self.hparams['learning_rate'], fused=self.device.type == 'cuda')

    def backward(self, trainer, loss, optimizer, optimizer_idx):
        loss.backward(retain_graph=True)


def dataset_paths(task, embedding_type='gin'):
    """The seven dataset paths train_config.py derives from ``data.task`` (train_config.py:213-230)."""
    t = Path(task)
    return dict(graph_path=str(t / 'edge_list.txt'), subgraph_path=str(t / 'subgraphs.pth'),
                embedding_path=str(t / ('%s_embeddings.pth' % embedding_type)),
                similarities_path=str(t / 'similarities/'), shortest_paths_path=str(t / 'shortest_path_matrix.npy'),
                degree_dict_path=str(t / 'degree_sequence.txt'), ego_graph_path=str(t / 'ego_graphs.txt'))


__all__ = ['SubGNN', 'LSTM', 'dataset_paths', 'json']
