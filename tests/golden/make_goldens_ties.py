#!/usr/bin/env python3
"""g7 under fastdtw's predecessor rules 1 and 2 (build container only; needs /root/reference).

    PYTHONDONTWRITEBYTECODE=1 python tests/golden/make_goldens_ties.py

tests/golden/tiny.npz (and density.npz) pin the structure similarities under rule 0 -- the restated pure-Python module, the
fastdtw stand-in's default.  The product's default rule is 2 (subgnn_amd/config.py: the pure-Python module raises on the
reference's padded component rows, so the reference ran the compiled variant), so the same stage boundary is pinned here for
rules 1 and 2: the reference's own ``SubGNN.compute_structure_patch_similarities`` (SubGNN/SubGNN.py:783-833) and
``gamma.get_degree_sequence`` / ``gamma.calc_dtw`` (SubGNN/gamma.py:21-59) are executed unmodified on the components and
structure patches of the ``tiny`` fixture, with the stand-in told the rule through SGNN_STANDIN_FASTDTW_TIE.  Like g7 this is a
self-consistency pin (the stand-in IS the restatement: PARITY UNPINNED for the DTW values).  Output: tests/golden/ties.npz."""
import os
import sys
import tempfile
from pathlib import Path

os.environ['PYTHONDONTWRITEBYTECODE'] = '1'
sys.dont_write_bytecode = True
HERE = Path(__file__).resolve().parent
REPO = HERE.parent.parent
REF = Path('/root/reference')
sys.path[:0] = [str(HERE / '_standins'), str(REF / 'SubGNN'), str(REF), str(REPO)]

import numpy as np          # noqa: E402
import networkx as nx       # noqa: E402
import torch                # noqa: E402

import SubGNN as S          # noqa: E402  (the reference module)


class _Holder:
    """What compute_structure_patch_similarities reads from ``self``."""


def main():
    out = {}
    for name in ('tiny', 'density'):
        g = np.load(HERE / (name + '.npz'), allow_pickle=False)
        G = nx.Graph()
        rp, col = g['g1_rowptr'], g['g1_col']
        G.add_nodes_from(int(v) for v in g['g1_node_order'])
        for v in range(1, len(rp) - 1):
            for w in col[rp[v]:rp[v + 1]]:
                G.add_edge(v, int(w))
        degree_dict = {v - 1: int(rp[v + 1] - rp[v] + (col[rp[v]:rp[v + 1]] == v).sum()) for v in range(1, len(rp) - 1)}
        h = _Holder()
        h.networkx_graph = G
        h.structure_anchors = torch.from_numpy(g['g5_structure_anchors'])
        h.hparams = {'structure_similarity_fn': 'dtw', 'n_processes': 2}
        cc = torch.from_numpy(g['g2_cc_ids_train'])
        for tie in (0, 1, 2):
            os.environ['SGNN_STANDIN_FASTDTW_TIE'] = str(tie)
            for internal, key in ((True, 'int'), (False, 'bor')):
                with tempfile.TemporaryDirectory() as d:
                    sims = S.SubGNN.compute_structure_patch_similarities(h, degree_dict, Path(d) / 'x' / 'sims.npy', internal, cc, None, 'train')
                arr = sims.numpy().astype(np.float32)
                if tie == 0:
                    assert np.array_equal(arr, g['g7_%s_struc_sim_train' % key]), 'rule 0 must reproduce g7'
                else:
                    out['%s/g7_tie%d_%s_struc_sim_train' % (name, tie, key)] = arr
    np.savez_compressed(HERE / 'ties.npz', **out)
    print('wrote', HERE / 'ties.npz', {k: v.shape for k, v in out.items()})


if __name__ == '__main__':
    main()
