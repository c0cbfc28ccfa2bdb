"""Shared by the GPU codec tests: the reference's own top-1 / top-2 margin of every RVQ decision (fp32 near-ties)."""
import torch


def code_margins(sd, cfg, emb, codes):
    """top-1 minus top-2 score of every RVQ decision along the reference's own residual path."""
    B, D, T = emb.shape
    res = emb.clone()
    out = torch.zeros(B, cfg.n_q, T)
    for q in range(cfg.n_q):
        E = sd[f"quantizer.vq.layers.{q}._codebook.embed"]
        x = res.permute(0, 2, 1).reshape(-1, D)
        dist = -(x.pow(2).sum(1, keepdim=True) - 2 * x @ E.t() + E.t().pow(2).sum(0, keepdim=True))
        top2 = dist.topk(2, dim=-1).values
        out[:, q] = (top2[:, 0] - top2[:, 1]).view(B, T)
        res = res - torch.nn.functional.embedding(codes[:, q], E).permute(0, 2, 1)
    return out
