class Data:                          # import-only in prepare_dataset/prepare_dataset.py:17 (node-embedding pre-training)
    def __init__(self, *a, **k):
        raise NotImplementedError
