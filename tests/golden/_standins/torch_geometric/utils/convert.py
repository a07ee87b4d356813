def to_networkx(*a, **k):           # import-only in SubGNN.py:38
    raise NotImplementedError
