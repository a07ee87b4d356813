"""Per-subgraph batching (mirrors reference SubGNN/datasets.py:9-57)."""
import torch
from torch.utils.data import Dataset


class SubgraphDataset(Dataset):
    """One item = one subgraph: its node ids, component ids, border set row, the three
    precomputed similarity rows, its index and label."""

    def __init__(self, subgraph_list, labels, cc_ids, N_border, NP_sim, I_S_sim, B_S_sim, multilabel,
                 multilabel_binarizer):
        self.subgraph_list, self.labels, self.cc_ids = subgraph_list, labels, cc_ids
        self.N_border, self.NP_sim, self.I_S_sim, self.B_S_sim = N_border, NP_sim, I_S_sim, B_S_sim
        self.multilabel, self.multilabel_binarizer = multilabel, multilabel_binarizer

    def __len__(self):
        return len(self.subgraph_list)

    def __getitem__(self, idx):
        pick = lambda t: t[idx] if t is not None else None
        if self.multilabel:
            label = torch.LongTensor(self.multilabel_binarizer.transform([self.labels[idx]]))
        else:
            label = torch.LongTensor([int(self.labels[idx])])
        return (torch.LongTensor(self.subgraph_list[idx]), self.cc_ids[idx], pick(self.N_border), pick(self.NP_sim),
                pick(self.I_S_sim), pick(self.B_S_sim), torch.LongTensor([idx]), label)
