"""Shared helpers for the parity tests: golden -> tensors, relative-error check."""
import json

import numpy as np
import torch

REL_TOL = 1e-4      # BASELINE.json north_star: "within 1e-4 rel on the float channel embeddings"


def T(x):
    return torch.from_numpy(np.ascontiguousarray(x))


ABS_FLOOR = 1e-2    # x tol x max|b|: the absolute slack an element much smaller than the tensor's largest is allowed


def _np(a):
    return a.detach().cpu().numpy() if isinstance(a, torch.Tensor) else np.asarray(a)


def rel_err(a, b):
    """Worst element of |a - b| / (|b| + ABS_FLOOR * max|b|): an ELEMENT-WISE relative error (round 3; rounds 1-2
    divided the largest absolute difference by the largest reference value, which let an element 100x smaller than
    the largest be off by 1 % unnoticed).  The floor keeps elements that are small by cancellation -- whose error is a
    rounding error of the terms, not of the result -- from dominating: with tol = 1e-4 the test is
    |a - b| <= 1e-4 |b| + 1e-6 max|b| for every element."""
    a, b = _np(a).astype(np.float64), _np(b).astype(np.float64)
    if a.shape != b.shape:
        a, b = np.broadcast_arrays(a, b)
    scale = np.abs(b).max() if b.size else 0.0
    if scale == 0:
        return float(np.abs(a).max()) if a.size else 0.0
    return float((np.abs(a - b) / (np.abs(b) + ABS_FLOOR * scale)).max())


def norm_err(a, b):
    """max|a - b| / max|b|: the whole-tensor measure of rounds 1-2, kept as an ADDITIONAL, tighter bound where a test
    had one below 1e-4 (it says how good the bulk is; the element-wise test says that no element is off)."""
    a, b = _np(a).astype(np.float64), _np(b).astype(np.float64)
    scale = np.abs(b).max() if b.size else 0.0
    if scale == 0:
        return float(np.abs(a).max()) if a.size else 0.0
    return float(np.abs(a - b).max() / scale)


def assert_close(a, b, what, tol=REL_TOL, norm_tol=None):
    e = rel_err(a, b)
    if not e < tol:
        an, bn = _np(a).astype(np.float64), _np(b).astype(np.float64)
        an, bn = np.broadcast_arrays(an, bn)
        scale = np.abs(bn).max()
        q = np.abs(an - bn) / (np.abs(bn) + ABS_FLOOR * scale)
        i = np.unravel_index(int(np.argmax(q)), q.shape)
        raise AssertionError('%s: element-wise relative error %.3e >= %.1e at %r: got %.9g, want %.9g (max|want| %.3g)'
                             % (what, e, tol, tuple(int(x) for x in i), an[i], bn[i], scale))
    if norm_tol is not None:
        n = norm_err(a, b)
        assert n < norm_tol, '%s: max|a-b| / max|b| = %.3e >= %.1e' % (what, n, norm_tol)


def g11_case(g, variant):
    """Unpack one g11 variant: hparams, params (requires_grad), batch, anchors, cc params, labels."""
    from oracle import integer_half as IH
    t = 'g11_%s/' % variant
    hp = json.loads(str(g[t + 'hparams']))
    params = {}
    for k in g.files:
        if k.startswith(t + 'sd/'):
            v = T(g[k]).clone()
            if v.dtype == torch.float32:
                v.requires_grad_(True)
            params[k[len(t) + 3:]] = v
    idx = T(g[t + 'idx'])
    cc_ids = T(IH.trim_zero_columns(g[t + 'cc_ids_train'][g[t + 'idx']]))
    batch = {'cc_ids': cc_ids, 'subgraph_idx': idx.view(-1, 1), 'NP_sim': T(g[t + 'np_sim_train'])[idx],
             'I_S_sim': T(g[t + 'int_sim_train'])[idx], 'B_S_sim': T(g[t + 'bor_sim_train'])[idx]}
    L = hp['n_layers']
    anchors = {
        'N_int': {'train': {l: T(g[t + 'N_int_train_%d' % l]) for l in range(L)}},
        'N_bor': {'train': {l: T(g[t + 'N_bor_train_%d' % l]) for l in range(L)}},
        'P_int': {'train': {l: T(g[t + 'P_int_train_%d' % l]) for l in range(L)}},
        'P_ext': {l: T(g[t + 'P_ext_%d' % l]) for l in range(L)},
        'S': {l: (T(g[t + 'S_patches_%d' % l]), [int(i) for i in g[t + 'S_idx_%d' % l]],
                  T(g[t + 'S_int_rw_%d' % l]), T(g[t + 'S_bor_rw_%d' % l])) for l in range(L)},
    }
    ccp = None
    if hp['trainable_cc']:
        ccp = {nm: T(g[t + 'cc_param/' + nm]).clone().requires_grad_(True)
               for nm in ('N_I', 'N_B', 'S_I', 'S_B', 'P_I', 'P_B')}
    labels = T(g['labels_train'])[idx]
    return t, hp, params, batch, anchors, ccp, labels


G11_VARIANTS = ('sum', 'max_trainable', 'bn', 'sumlstm_norm', 'ff_attn')


def write_dataset_from_golden(g, root, name='ds', with_ego=None):
    """Recreate the on-disk dataset (SURVEY.md 8f-1 formats) a golden fixture was made from."""
    import json
    import os
    d = os.path.join(str(root), name)
    os.makedirs(os.path.join(d, 'similarities'), exist_ok=True)
    with open(os.path.join(d, 'edge_list.txt'), 'w') as f:
        for u, v in g['edge_list']:
            f.write('%d %d\n' % (u, v))
    with open(os.path.join(d, 'subgraphs.pth'), 'w') as f:
        for sp in ('train', 'val', 'test'):
            for row, lab in zip(g['subgraphs_' + sp], g['labels_' + sp]):
                nodes = [int(v) - 1 for v in row if v != 0]
                f.write('-'.join(str(n) for n in nodes) + '\t' + str(int(lab)) + '\t' + sp + '\t\n')
    torch.save(T(g['embeddings'][1:]).clone(), os.path.join(d, 'gin_embeddings.pth'))
    np.save(os.path.join(d, 'shortest_path_matrix.npy'), g['apsp'])
    rp, col = g['g1_rowptr'], g['g1_col']
    deg, ego = {}, {}
    for v in range(1, len(rp) - 1):
        nb = col[rp[v]:rp[v + 1]]
        deg[str(v - 1)] = int(len(nb) + (nb == v).sum())
        ego[str(v - 1)] = [int(w) - 1 for w in nb]
    with open(os.path.join(d, 'degree_sequence.txt'), 'w') as f:
        json.dump(deg, f)
    if bool(g['has_ego']) if with_ego is None else with_ego:
        with open(os.path.join(d, 'ego_graphs.txt'), 'w') as f:
            json.dump(ego, f)
    return name


SLOTS = ('N_I', 'N_B', 'S_I', 'S_B', 'P_I', 'P_B')


def oracle_inputs(m, batch, idx):
    """The product's prepared state as the containers oracle/float_half.py takes (CPU tensors)."""
    hp = m.hparams
    Lr = hp['n_layers']
    cpu = lambda t: t.detach().cpu()
    params = {}
    for k, v in m.state_dict().items():
        v = cpu(v).clone()
        if v.dtype == torch.float32:
            v.requires_grad_(True)
        params[k] = v
    if str(hp.get('embedding_dtype', 'fp32')) == 'fp16':
        # the fused kernels gather from the IEEE-half copy of the table (fp32 accumulate)
        params['node_embeddings.weight'] = params['node_embeddings.weight'].detach().half().float().requires_grad_(True)
    anchors = {'N_int': {}, 'N_bor': {}, 'P_int': {}, 'P_ext': {}, 'S': {}}
    if hp['use_neighborhood']:
        anchors['N_int'] = {'train': {l: cpu(m.anchors_neigh_int['train'][l]) for l in range(Lr)}}
        anchors['N_bor'] = {'train': {l: cpu(m.anchors_neigh_border['train'][l]) for l in range(Lr)}}
    if hp['use_position']:
        anchors['P_int'] = {'train': {l: cpu(m.anchors_pos_int['train'][l]) for l in range(Lr)}}
        anchors['P_ext'] = {l: cpu(m.anchors_pos_ext[l]) for l in range(Lr)}
    if hp['use_structure']:
        for l in range(Lr):
            p, i, a, b = m.anchors_structure[l]
            anchors['S'][l] = (cpu(p), [int(v) for v in (i.tolist() if torch.is_tensor(i) else i)], cpu(a), cpu(b))
    npsim = batch['NP_sim']
    if isinstance(npsim, dict):
        npsim = {k: cpu(v.dense() if hasattr(v, 'dense') else v) for k, v in npsim.items()}
    elif npsim is not None:
        npsim = cpu(npsim)
    ob = {'cc_ids': cpu(batch['cc_ids']), 'subgraph_idx': cpu(batch['subgraph_idx']), 'NP_sim': npsim,
          'I_S_sim': cpu(batch['I_S_sim']) if batch['I_S_sim'] is not None else None,
          'B_S_sim': cpu(batch['B_S_sim']) if batch['B_S_sim'] is not None else None}
    ccp = {nm: params['train_%s_cc_embed' % nm] for nm in SLOTS} if hp['trainable_cc'] else None
    return params, anchors, ob, ccp
