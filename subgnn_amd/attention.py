"""Optional attention read-out over a subgraph's components (mirrors reference SubGNN/attention.py,
used when hparams['ff_attn'], SubGNN/SubGNN.py:179-183,298-301).  Same class and parameter names
(``_w_matrix``, ``_u_matrix``, ``_v_vector``: state-dict compatible); the score contraction runs on
the matrix cores (ops.attn_scores -> sgnn_attn_scores_fwd)."""
import torch
import torch.nn as nn
from torch.nn.parameter import Parameter

from . import ops


def tiny_value_of_dtype(dtype):
    if dtype in (torch.float, torch.double):
        return 1e-13
    if dtype == torch.half:
        return 1e-4
    raise TypeError('Does not support dtype ' + str(dtype))


def masked_softmax(vector, mask, dim=-1, memory_efficient=False):
    """attention.masked_softmax (attention.py:22-57): softmax over the un-masked entries; an
    all-masked row gives zeros."""
    if mask is None:
        return torch.nn.functional.softmax(vector, dim=dim)
    while mask.dim() < vector.dim():
        mask = mask.unsqueeze(1)
    if memory_efficient:
        return torch.nn.functional.softmax(vector.masked_fill(~mask, torch.finfo(vector.dtype).min), dim=dim)
    result = torch.nn.functional.softmax(vector * mask, dim=dim) * mask
    return result / (result.sum(dim=dim, keepdim=True) + tiny_value_of_dtype(result.dtype))


class AdditiveAttention(nn.Module):
    """score = V tanh(W x + U y) (attention.py:102-139), softmax-normalised over the rows of y."""

    def __init__(self, vector_dim, matrix_dim, normalize=True, half_operands=False):
        super().__init__()
        self._normalize = normalize
        self.half_operands = half_operands           # embedding_dtype = 'fp16': the score GEMM takes half operands
        self._w_matrix = Parameter(torch.Tensor(vector_dim, vector_dim))
        self._u_matrix = Parameter(torch.Tensor(matrix_dim, vector_dim))
        self._v_vector = Parameter(torch.Tensor(vector_dim, 1))
        self.reset_parameters()

    def reset_parameters(self):
        torch.nn.init.xavier_uniform_(self._w_matrix)
        torch.nn.init.xavier_uniform_(self._u_matrix)
        torch.nn.init.xavier_uniform_(self._v_vector)

    def forward(self, vector, matrix, matrix_mask=None):
        B, C, H = matrix.shape
        qW = vector.matmul(self._w_matrix)                                   # (B, H): tiny
        scores = ops.attn_scores(matrix.reshape(B * C, H), self._u_matrix, qW, self._v_vector, C,
                                 half_operands=self.half_operands).view(B, C)
        return masked_softmax(scores, matrix_mask) if self._normalize else scores
