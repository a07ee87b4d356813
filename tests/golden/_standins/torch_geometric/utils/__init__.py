def add_self_loops(*a, **k):        # import-only in subgraph_mpn.py:13
    raise NotImplementedError
