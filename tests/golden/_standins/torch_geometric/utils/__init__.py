def add_self_loops(*a, **k):        # import-only in subgraph_mpn.py:13
    raise NotImplementedError


def from_networkx(*a, **k):         # import-only in prepare_dataset/prepare_dataset.py:18
    raise NotImplementedError
