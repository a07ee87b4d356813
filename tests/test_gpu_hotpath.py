"""-m gpu: the sparse precompute (hotpath.prepare_sparse) gives the model the same numbers the
dense reference-shaped prepare_data gathers from its slabs, and a DP-sharded pass reproduces
the single-rank pass."""
import numpy as np
import pytest
import torch

from helpers import assert_close, write_dataset_from_golden

pytestmark = pytest.mark.gpu
DEV = 'cuda:0'


def _models(golden, tmp_path, over=None):
    from subgnn_amd import config
    from subgnn_amd.SubGNN import SubGNN, dataset_paths
    name = write_dataset_from_golden(golden, tmp_path, with_ego=False)   # true k-hop border in both paths
    config.PROJECT_ROOT = tmp_path
    out = []
    for _ in range(2):
        hp = dict(golden.hp)
        hp.update({'seed': golden.seed, 'neigh_sample_border_size': 2, 'lin_dropout': 0.0})
        if over:
            hp.update(over)
        torch.manual_seed(0)
        out.append(SubGNN(hp, **dataset_paths(name)))
    return out


@pytest.mark.parametrize('name', ['tiny', 'density'])
def test_sparse_prepare_equals_dense_prepare(name, tmp_path):
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden(name)                 # no ego dict: true k-hop border in both paths
    dense, sparse = _models(golden, tmp_path)
    dense.prepare_data()
    hotpath.prepare_sparse(sparse, 'train')
    hp = dense.hparams
    assert torch.equal(dense.train_cc_ids, sparse.train_cc_ids)
    slab = dense.train_neigh_pos_similarities
    S, C, _ = slab.shape
    for l in range(hp['n_layers']):
        # anchors: identical draws (same tape; ragged sampler == padded sampler)
        assert torch.equal(dense.anchors_neigh_int['train'][l], sparse.anchors_neigh_int['train'][l])
        assert torch.equal(dense.anchors_neigh_border['train'][l], sparse.anchors_neigh_border['train'][l])
        assert torch.equal(dense.anchors_pos_int['train'][l], sparse.anchors_pos_int['train'][l])
        assert torch.equal(dense.anchors_pos_ext[l], sparse.anchors_pos_ext[l])
        real = (dense.train_cc_ids[:, :, 0] != 0)
        for key, ids in ((('N', 'in', l), dense.anchors_neigh_int['train'][l]),
                         (('N', 'out', l), dense.anchors_neigh_border['train'][l]),
                         (('P', 'in', l), dense.anchors_pos_int['train'][l].unsqueeze(1).expand(S, C, -1)),
                         (('P', 'out', l), dense.anchors_pos_ext[l].view(1, 1, -1).expand(S, C, -1))):
            edge = (ids != 0) & real.unsqueeze(-1)
            want = torch.gather(slab, 2, (ids - 1).clamp(min=0)) * edge
            got = sparse.train_neigh_pos_similarities[key]
            got = (got.dense() if hasattr(got, 'dense') else got) * edge          # ops.ZeroSims: known-zero weights
            assert torch.equal(got, want), key
    assert torch.equal(dense.structure_anchors, sparse.structure_anchors)
    assert torch.equal(dense.train_int_struc_similarities, sparse.train_int_struc_similarities)
    assert torch.equal(dense.train_bor_struc_similarities, sparse.train_bor_struc_similarities)
    # same logits through the per-edge similarity dict as through the dense slab
    idx = torch.arange(6)
    dense.eval(); sparse.eval()
    sparse.load_state_dict(dense.state_dict())
    with torch.no_grad():
        a = dense._forward_batch('train', dense.make_batch('train', idx))
        b = sparse._forward_batch('train', sparse.make_batch('train', idx))
    assert_close(b, a, 'logits sparse vs dense', 1e-5)


def test_full_split_step_runs_and_is_deterministic(tmp_path):
    from conftest import load_golden
    from subgnn_amd import hotpath
    golden = load_golden('density')
    (m, _) = _models(golden, tmp_path)
    hotpath.prepare_sparse(m, 'train')
    batch = hotpath.full_split_batch(m, 'train')
    m.eval()
    with torch.no_grad():
        a = m._forward_batch('train', batch)
        hotpath.prepare_sparse(m, 'train')          # same seed -> same tape -> same draws
        b = m._forward_batch('train', hotpath.full_split_batch(m, 'train'))
    assert torch.equal(a, b)
    assert a.shape[0] == len(m.train_sub_G)
