"""Cached visual features: the producer of scripts/get_clip_vit_embeddings.py:55-78 and the reader of
dataset_loaders/dataset_loaders.py:162-184, for the 2-D "precomputed feature" fast path of the wrappers
(model/model.py:328-330, :460-462).

On-disk format (unchanged): ``torch.save({"reddit_ids": int64[N], "embeddings": float32[N, 512]}, path)``;
embeddings are the UN-normalised ``encode_image`` outputs, exactly what the reference's script stores."""
from __future__ import annotations

import time
from typing import Iterable, Sequence

import torch


@torch.no_grad()
def cache_clip_vit_embeddings(model, batches: Iterable[torch.Tensor], ids: Sequence[int], out_path: str, device="cuda",
                              verbose: bool = False) -> dict:
    """``batches`` yields image tensors [b,3,224,224] (fp32 / bf16 pre-normalised, or raw uint8 -- then the
    CLIP ToTensor+Normalize is fused into the patch gather).  Returns the saved dict."""
    model.eval()
    out = []
    for bi, imgs in enumerate(batches):
        tic = time.time()
        y = model.encode_image(imgs.to(device))              # get_clip_vit_embeddings.py:61
        out.append(y.float().cpu())
        if verbose:
            print(bi, "%.1fHz" % (imgs.shape[0] / max(1e-9, time.time() - tic)), tuple(y.shape))
    stacked = torch.cat(out)
    ids_t = torch.tensor(list(ids), dtype=torch.int64)
    assert ids_t.numel() == stacked.shape[0]
    save_dict = {"reddit_ids": ids_t, "embeddings": stacked.to(torch.float32)}
    torch.save(save_dict, out_path)
    return save_dict


def load_features(ids: Sequence[int], path: str) -> torch.Tensor:
    """dataset_loaders.py:176-184 (the non-comment branch): rows of ``embeddings`` in the order of ``ids``."""
    stored = torch.load(path)
    assert stored["reddit_ids"].dtype is torch.int64
    assert stored["embeddings"].dtype is torch.float32
    lookup = {int(el): i for i, el in enumerate(stored["reddit_ids"])}
    sel = [lookup[int(r)] for r in ids]
    feats = stored["embeddings"][sel]
    assert feats.shape[0] == len(sel)
    return feats
