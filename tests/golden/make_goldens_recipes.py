#!/usr/bin/env python3
"""Fixtures for the four synthetic-dataset recipes, produced by IMPORTING AND RUNNING the reference's
prepare_dataset/prepare_dataset.py (build container only) with Python's global ``random`` seeded first -- the
reference draws everything from it and never seeds it itself:

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens_recipes.py       -> tests/golden/recipes.npz

Per recipe (density / cut_ratio / coreness / cc, at sizes that run in seconds): the keyword arguments given to
SyntheticGraph, the final graph (edges in ``graph.edges()`` order, nodes in ``graph.nodes()`` order), the subgraph lists and
labels exactly as the object holds them, the split mask of ``generate_mask`` drawn right afterwards, and the next
``random.random()`` -- the position of the stream after everything the recipe consumed.
Stand-ins: tests/golden/_standins (torch_geometric is imported by the reference module and unused on this path);
``train_node_emb`` (PyG pre-training) is an empty module.  No reference source text is stored: arrays and JSON only.
"""
import io
import json
import os
import random
import sys
import types
import warnings
from contextlib import redirect_stdout
from pathlib import Path

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
warnings.simplefilter('ignore')

HERE = Path(__file__).resolve().parent
REF = Path(os.environ.get('SUBGNN_REFERENCE', '/root/reference'))
sys.path.insert(0, str(HERE / '_standins'))

import numpy as np                   # noqa: E402

SEED = 42
CASES = {
    'density': dict(base_graph_type='barabasi_albert', subgraph_type='bfs', n_subgraphs=40, n_connected_components=1,
                    n_subgraph_nodes=12, features_type='one_hot', n=300, p=0.5, q=0, m=4, n_bins=3,
                    subgraph_generator='complete', modify_graph_for_properties=True, desired_property='density'),
    'cut_ratio': dict(base_graph_type='barabasi_albert', subgraph_type='plant', n_subgraphs=30, n_connected_components=1,
                      n_subgraph_nodes=10, features_type='one_hot', n=300, p=0.5, q=0, m=4, n_bins=3,
                      subgraph_generator='complete', modify_graph_for_properties=True, desired_property='cut_ratio'),
    'coreness': dict(base_graph_type='duplication_divergence_graph', subgraph_type='plant', n_subgraphs=4,
                     n_connected_components=1, n_subgraph_nodes=8, features_type='one_hot', n=300, p=0.7, q=0, m=1, n_bins=3,
                     subgraph_generator='duplication_divergence_graph', modify_graph_for_properties=True,
                     desired_property='coreness'),
    'cc': dict(base_graph_type='barabasi_albert', subgraph_type='staple', n_subgraphs=30, n_connected_components=None,
               n_subgraph_nodes=8, features_type='one_hot', n=200, p=0.5, q=0, m=3, n_bins=2,
               subgraph_generator='extended_barabasi_albert', modify_graph_for_properties=True, desired_property='cc'),
    # a second density case in which no node is cut off: the final relabelling is the identity there
    'density_b': dict(base_graph_type='barabasi_albert', subgraph_type='bfs', n_subgraphs=25, n_connected_components=1,
                      n_subgraph_nodes=10, features_type='one_hot', n=400, p=0.5, q=0, m=6, n_bins=3,
                      subgraph_generator='complete', modify_graph_for_properties=True, desired_property='density'),
}


def ragged(lists):
    w = max([len(s) for s in lists] + [1])
    out = np.full((len(lists), w), -1, dtype=np.int64)
    for i, s in enumerate(lists):
        out[i, :len(s)] = s
    return out


def main():
    fake = types.ModuleType('config')
    fake.PROJECT_ROOT = Path('/tmp/subgnn_recipe_goldens')
    fake.PAD_VALUE = 0
    sys.modules['config'] = fake                                  # config_prepare_dataset creates DATASET_DIR at import
    sys.modules['train_node_emb'] = types.ModuleType('train_node_emb')
    sys.path.insert(0, str(REF / 'prepare_dataset'))
    import prepare_dataset as PD
    import config_prepare_dataset as C
    assert C.RANDOM_SEED == SEED
    out = {'seed': np.array(SEED)}
    for name, kw in CASES.items():
        random.seed(SEED)
        np.random.seed(SEED)
        with redirect_stdout(io.StringIO()):
            sg = PD.SyntheticGraph(**kw)
            mask = PD.generate_mask(len(sg.subgraph_labels))
        nxt = random.random()
        t = name + '/'
        out[t + 'kwargs'] = np.array(json.dumps(kw))
        out[t + 'edges'] = np.array(list(sg.graph.edges()), dtype=np.int64).reshape(-1, 2)
        out[t + 'nodes'] = np.array(list(sg.graph.nodes()), dtype=np.int64)
        out[t + 'subgraphs'] = ragged([list(s) for s in sg.subgraphs])
        out[t + 'labels'] = np.array([str(l) for l in sg.subgraph_labels])
        out[t + 'mask'] = np.array(mask, dtype=np.int64)
        out[t + 'next_random'] = np.array(nxt, dtype=np.float64)
        print(name, 'nodes', sg.graph.number_of_nodes(), 'edges', sg.graph.number_of_edges(), 'subgraphs', len(sg.subgraphs),
              'labels', sorted(set(out[t + 'labels'].tolist())))
    np.savez_compressed(HERE / 'recipes.npz', **out)
    print('recipes written:', len(out), 'arrays,', (HERE / 'recipes.npz').stat().st_size // 1024, 'KiB')


if __name__ == '__main__':
    main()
