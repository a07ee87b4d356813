"""The float-half oracle (dense torch-CPU restatement) against goldens g9/g10/g11 from the
imported reference.  CPU only."""
import numpy as np
import pytest
import torch

from oracle import float_half as FH
from helpers import T, assert_close, g11_case, G11_VARIANTS


@pytest.mark.parametrize('tag', ['N', 'P', 'S'])
@pytest.mark.parametrize('side', ['in', 'out'])
def test_g10_sg_mpn(tiny, tag, side):
    g, t = tiny, 'g10_%s_%s_' % (tag, side)
    W, b, wp, bp = [T(g[t + k]).requires_grad_(True) for k in ('W', 'b', 'wp', 'bp')]
    cc = T(g[t + 'cc_embeds']).requires_grad_(True)
    ae = T(g[t + 'anchor_embeds']).requires_grad_(True)
    idx = [int(i) for i in g[t + 'sim_index']] if tag == 'S' else None
    o, p = FH.sg_mpn_forward(W, b, wp, bp, T(g[t + 'sims']), cc, T(g[t + 'patches']), ae, T(g[t + 'mask']), idx)
    ((o * T(g[t + 'gout_cc'])).sum() + (p * T(g[t + 'gout_pos'])).sum()).backward()
    assert_close(o, g[t + 'out_cc'], 'cc out')
    assert_close(p, g[t + 'out_pos'], 'pos out')
    for ten, k in ((cc, 'grad_cc_embeds'), (ae, 'grad_anchor_embeds'), (W, 'grad_W'), (b, 'grad_b'),
                   (wp, 'grad_wp'), (bp, 'grad_bp')):
        assert_close(ten.grad, g[t + k], k)


def test_g10_padded_cc_rows_are_relu_bias(tiny):
    """mpn:168,239: padded CC rows go through update() too and come out as ReLU(W[x||0]+b)."""
    g, t = tiny, 'g10_N_in_'
    mask_cc = T(g['g12_cc_ids'])[:, :, 0] != 0
    assert (~mask_cc).any()
    out = T(g[t + 'out_cc'])
    x = T(g[t + 'cc_embeds'])
    W, b = T(g[t + 'W']), T(g[t + 'b'])
    D = x.shape[-1]
    expect = torch.relu(x @ W[:, :D].T + b)
    assert_close(out[~mask_cc], expect[~mask_cc], 'padded rows')


@pytest.mark.parametrize('ch,inside', [(c, i) for c in ('N', 'P', 'S') for i in (True, False)])
def test_g9_get_anchor_patches(tiny, ch, inside):
    g = tiny
    hp = g.hp
    params = {k[3:]: T(g[k]) for k in g.files if k.startswith('sd/')}
    L = hp['n_layers']
    anchors = {
        'N_int': {'train': {l: T(g['g8_N_int_train_%d' % l]) for l in range(L)}},
        'N_bor': {'train': {l: T(g['g8_N_bor_train_%d' % l]) for l in range(L)}},
        'P_int': {'train': {l: T(g['g8_P_int_train_%d' % l]) for l in range(L)}},
        'P_ext': {l: T(g['g8_P_ext_%d' % l]) for l in range(L)},
        'S': {l: (T(g['g8_S_patches_%d' % l]), [int(i) for i in g['g8_S_idx_%d' % l]],
                  T(g['g8_S_int_rw_%d' % l]), T(g['g8_S_bor_rw_%d' % l])) for l in range(L)},
    }
    cc_ids = T(g['g12_cc_ids'])
    mask = (cc_ids != 0)[:, :, 0]
    channel = {'N': 'neighborhood', 'P': 'position', 'S': 'structure'}[ch]
    with torch.no_grad():
        ap, am, ae = FH.get_anchor_patches(params, hp, params['node_embeddings.weight'], T(g['g12_subgraph_idx']),
                                           cc_ids, mask, anchors, 'train', 1, channel, inside)
    tag = 'g9_%s_%s_' % (ch, 'in' if inside else 'out')
    assert np.array_equal(ap.numpy(), g[tag + 'patches'])
    assert np.array_equal(am.numpy(), g[tag + 'mask'])
    assert_close(ae, g[tag + 'embeds'], 'anchor embeds', 1e-5)


@pytest.mark.parametrize('variant', G11_VARIANTS)
def test_g11_full_forward_and_grads(tiny, variant):
    g = tiny
    t, hp, params, batch, anchors, ccp, labels = g11_case(g, variant)
    logits = FH.forward(params, hp, 'train', batch, anchors, ccp)
    loss = torch.nn.functional.cross_entropy(logits, labels)
    loss.backward()
    assert_close(logits, g[t + 'logits'], 'logits', 1e-5)
    assert_close(loss, g[t + 'loss'], 'loss', 1e-5)
    n = 0
    for k in g.files:
        if not k.startswith(t + 'grad/'):
            continue
        nm = k[len(t) + 5:]
        if nm.startswith('train_'):
            continue                    # trainable CC Parameters: checked through cc_grad below
        ref = g[k]
        got = params[nm].grad
        if got is None:
            assert np.abs(ref).max() == 0, nm
        else:
            assert_close(got, ref, 'grad ' + nm)
            n += 1
    assert n > 10
    if ccp is not None:
        for nm, p in ccp.items():
            key = t + 'cc_grad/' + nm
            if key in g.files:
                if p.grad is None:
                    assert np.abs(g[key]).max() == 0
                else:
                    assert_close(p.grad, g[key], 'cc grad ' + nm)
