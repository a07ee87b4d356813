"""SG_MPN: one subgraph-level message-passing layer (mirrors reference SubGNN/subgraph_mpn.py).

Same constructor, same parameter names (``linear``, ``linear_position``: state-dict
compatible), same ``forward`` signature and return values.  The body -- edge construction,
similarity lookup, message = sim * x_anchor, add-aggregation per component, read-out -- is one
HIP kernel (ops.mpn -> sgnn_mpn_fwd / sgnn_mpn_bwd; a shard-sized layer over SHARED anchors is a dense contraction and
goes to the library); the Linear(2D -> D) + ReLU update is a
fused fp32-MFMA kernel (ops.update_layer -> sgnn_update_fwd / sgnn_update_bwd; widths other than 32 / 64 / 128
keep the library GEMM).
"""
import torch
import torch.nn as nn
import torch.nn.functional as F

from . import ops


class SG_MPN(nn.Module):
    def __init__(self, hparams):
        super().__init__()
        self.hparams = hparams
        D = hparams['node_embed_size']
        self.linear = nn.Linear(D * 2, D)
        self.linear_position = nn.Linear(D, 1)

    # -- shared tail: update() and the read-out non-linearity (mpn:122-131, 233-241) -------
    def _finish(self, cc_embeds, agg, z, need_out=True, need_pos=True, activated=False, defer_update=False):
        B, C, D = cc_embeds.shape
        if not need_out:
            # the caller reads only the position read-out of this layer (the last layer of the position / structure
            # channels: their updated component embeddings would feed a next layer that does not exist)
            out = None
        elif self.hparams['use_mpn_projection'] and defer_update:
            # the caller collects the bodies of this layer and runs their update layers together (ops.update_layers)
            out = ops.PendingUpdate(cc_embeds.reshape(B * C, D), agg, self.linear.weight, self.linear.bias, (B, C))
        elif self.hparams['use_mpn_projection']:
            # applied to every component row, padded ones included (mpn:168,239)
            out = ops.update_layer(cc_embeds.reshape(B * C, D), agg, self.linear.weight, self.linear.bias)
        else:
            out = agg if agg.dim() == 2 else (agg[0] if agg.shape[0] == 1 else agg.sum(0))
        shaped = (lambda o: o if (o is None or isinstance(o, ops.PendingUpdate)) else o.view(B, C, -1))
        if not need_pos:                     # the neighbourhood channel hands on its component embeddings only
            return shaped(out), None
        z = z.view(B, C, -1)
        if self.hparams.get('norm_pos_struc_embed', False):
            pos = F.normalize(z, p=2, dim=-1)
        else:
            pos = z if activated else F.relu(z)       # (activated: the layer kernel wrote relu(z) itself)
        return shaped(out), pos

    def forward(self, networkx_graph, sims, cc_ids, cc_embeds, cc_embed_mask, anchor_patches, anchor_embeds,
                anchor_mask, anchors_sim_index):
        """Reference signature (mpn:133-174).  ``networkx_graph`` and ``cc_ids`` are unused by
        the arithmetic, as in the reference.  anchor_patches (B,C,A,Lp) ids, anchor_embeds
        (B,C,A,D), anchor_mask (B,C,A,Lp) bool, anchors_sim_index: list (structure) or None."""
        B, C, D = cc_embeds.shape
        A = anchor_patches.shape[2]
        R = B * C
        ids = anchor_patches[..., 0].reshape(R, A).contiguous()
        edge = anchor_mask[..., 0].reshape(R, A).to(torch.uint8).contiguous()
        sim_col = None
        if anchors_sim_index is not None:
            sim_col = torch.as_tensor([int(i) for i in anchors_sim_index], dtype=torch.int64, device=cc_embeds.device)
        agg, z = ops.mpn(anchor_embeds.reshape(R, A, D), self.linear_position.weight, self.linear_position.bias, sims,
                         src=ops.SRC_DENSE, R=R, A=A, ids=ids, edge_mask=edge, sim_col=sim_col)
        return self._finish(cc_embeds, agg, z)

    def forward_fused(self, sims, cc_embeds, cc_embed_mask, *, src, x, ids=None, id_div=1, sim_col=None,
                      sims_per_edge=False, need_out=True, defer_readout=False, need_pos=True, edge_plan=None, defer_update=False):
        """Fast path used by SubGNN.forward: the anchor rows are gathered inside the kernel
        (src GATHER: x = embedding table, ids (R/id_div, A)) or shared by all rows (src SHARED:
        x (A,D)), so the (B,C,A,D) tensor of get_anchor_patches is never materialised."""
        B, C, D = cc_embeds.shape
        R = B * C
        A = ids.shape[-1] if ids is not None else x.shape[0]
        row_mask = getattr(cc_embed_mask, '_sgnn_u8', None)
        if row_mask is None or row_mask.numel() != R:
            row_mask = cc_embed_mask.reshape(R).to(torch.uint8).contiguous()
        if defer_readout and not need_out and A > 0 and not self.hparams.get('norm_pos_struc_embed', False) \
                and (isinstance(sims, ops.ZeroSims) or src == ops.SRC_SHARED):
            # only the read-out of this layer is consumed, and only summed over a subgraph's components: it is written
            # straight into its slot of the subgraph embedding (ops.subgraph_embedding)
            wp, bp = self.linear_position.weight.view(-1), self.linear_position.bias
            # (the scores s = x wp -- 0 for a PAD anchor -- are computed inside ops.subgraph_embedding's launches, for all such
            # pieces of the step together, and so is their backward: no matrix-vector / mask launches per piece)
            if isinstance(sims, ops.ZeroSims):
                return None, ops.ReadoutPiece(None, None, None, bp, A, row_mask, R)
            sims2 = sims.reshape(R, -1)
            if not sims2.is_contiguous():
                sims2 = sims2.contiguous()
            if sim_col is None and not sims_per_edge:
                sim_col = (ids - 1).clamp(min=0)
            return None, ops.ReadoutPiece(sims2, sim_col, None, bp, A, row_mask, R, X=x, wp=wp,
                                          ids=ids.reshape(-1).contiguous() if ids is not None else None)
        if isinstance(sims, ops.ZeroSims):          # all edge weights 0: messages vanish, read-out = bias
            agg = ops.zeros_cached((R, D), cc_embeds.dtype, cc_embeds.device)       # (read-only: nothing writes an aggregate)
            z = self.linear_position.bias.view(1, 1).expand(R, A)
            return self._finish(cc_embeds, agg, z, need_out, need_pos, defer_update=defer_update)
        # (SubGNN._forward converts the mask once per forward and hangs it on the tensor: one launch instead of one per layer)
        relu_z = bool(need_pos and not self.hparams.get('norm_pos_struc_embed', False))
        # (the anchor-chunk partials of a batch-sized call go to the update layer as they are: it adds them while loading)
        agg, z = ops.mpn(x, self.linear_position.weight, self.linear_position.bias, sims, src=src, R=R, A=A, ids=ids,
                         id_div=id_div, row_mask=row_mask, sim_col=sim_col, sims_per_edge=sims_per_edge,
                         need_agg=need_out, edge_plan=edge_plan,
                         keep_chunks=bool(need_out and self.hparams['use_mpn_projection']), relu_z=relu_z,
                         # (queued with the other bodies of the layer when the caller collects them: it runs ops.update_layers /
                         # ops.flush_lazy_mpn before anything reads agg or the read-out; a read-out that is normalised here is read here)
                         lazy=bool(defer_update and (relu_z or not need_pos)))
        return self._finish(cc_embeds, agg, z, need_out, need_pos, activated=relu_z, defer_update=defer_update)
