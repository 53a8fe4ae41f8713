"""Generate tests/golden/tokenise_cases.npz by running the REFERENCE's own ``_tokenise`` (dataset_loaders/dataset_loaders.py:224-248,
unmodified) on constructed texts.

Run in the build container only (needs the read-only reference checkout):

    python tests/golden/make_tokenise_golden.py [/root/reference]

``dataset_loaders/dataset_loaders.py`` imports third-party packages that are not installed here (torchvision, ffmpeg, rake_nltk, and the
un-vendored ``clip``): permissive stand-in modules let the file import unmodified -- none of them is touched by ``_tokenise`` itself,
which uses only ``self.tokenizer`` (``.encoder`` dict, ``.encode``) and ``self.rake`` (``extract_keywords_from_text`` /
``get_ranked_phrases``).  Those two are stand-ins too, for this script only: a deterministic word-level "BPE" (one or two ids per word
from a hash) and a "RAKE" that keeps every other word.  What the fixture pins is everything ``_tokenise`` itself decides: SOT / EOT
framing, the ``len(tokens) >= max_len`` test (:236), re-encoding of the summary (:238-240), the hard cut ``tokens[:max_len-1] + [eot]``
(:241-244), zero padding, int64.  The fixture stores the encoder's outputs (id lists, as offsets + a flat array), the summariser's
re-encodings, and the reference's result array -- the boundary vtc_pack_tokens / oracle.tokens_ref take over at."""
from __future__ import annotations

import os
import sys
import types
import zlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REF = sys.argv[1] if len(sys.argv) > 1 else "/root/reference"


class _Anything:
    """callable / attribute sink for the module-level code of the reference file (transforms.Compose([...]) etc.)"""

    def __init__(self, *a, **k):
        pass

    def __call__(self, *a, **k):
        return _Anything()

    def __getattr__(self, name):
        return _Anything()


class _StubModule(types.ModuleType):
    def __getattr__(self, name):
        if name.startswith("__"):
            raise AttributeError(name)
        return _Anything()


for name in ("torchvision", "torchvision.transforms", "ffmpeg", "rake_nltk", "clip", "clip.simple_tokenizer", "PIL", "PIL.Image"):
    if name not in sys.modules:
        try:
            __import__(name)
        except Exception:      # noqa: BLE001 -- absent here: a sink module
            sys.modules[name] = _StubModule(name)
sys.path.insert(0, REF)
import importlib  # noqa: E402

DL = importlib.import_module("dataset_loaders.dataset_loaders")
assert DL.__file__.startswith(REF), DL.__file__
tokenise = DL.VisionTitleCommentDatasetBase._tokenise          # the reference's function object, unmodified

SOT, EOT = 49406, 49407


class Tok:
    encoder = {"<|startoftext|>": SOT, "<|endoftext|>": EOT}

    @staticmethod
    def encode(text):
        ids = []
        for w in text.split():
            h = zlib.crc32(w.encode())
            ids.append(1 + h % 49000)
            if len(w) > 6:                      # long words split into two ids, as BPE would
                ids.append(1 + (h >> 7) % 49000)
        return ids


class Rake:
    def extract_keywords_from_text(self, text):
        self.words = text.split()

    def get_ranked_phrases(self):
        return self.words[::2]                  # half the words: long texts may still be too long afterwards


rng = np.random.default_rng(11)
vocab = ["w%d" % i for i in range(400)] + ["longerword%d" % i for i in range(100)]


def text(n):
    return " ".join(vocab[int(i)] for i in rng.integers(0, len(vocab), n))


cases = {
    "short_mixed": [text(n) for n in (1, 3, 10, 40)] + [""],
    "boundary": [text(n) for n in (60, 70, 74, 75, 76, 77, 78)],          # around max_len with SOT/EOT
    "summarised_fits": [text(100), text(120)],                              # too long, the summary fits
    "summarised_still_long": [text(200), text(400)],                        # too long even after the summary: hard cut + EOT
    "single_string": text(12),                                              # a str, not a list (:225-226)
}
arrays, desc = {}, {}
for name, texts in cases.items():
    for max_len in (77, 24):
        self = types.SimpleNamespace(tokenizer=Tok(), rake=Rake())
        got = tokenise(self, texts, max_len=max_len)
        assert got.dtype == torch.long
        tl = [texts] if isinstance(texts, str) else texts
        enc = [Tok.encode(t) for t in tl]
        summ = []
        for t in tl:
            r = Rake(); r.extract_keywords_from_text(t)
            summ.append(Tok.encode(" ".join(r.get_ranked_phrases())))
        key = f"{name}.{max_len}"
        arrays[key + ".enc"] = np.array([i for e in enc for i in e], dtype=np.int64)
        arrays[key + ".enc_off"] = np.cumsum([0] + [len(e) for e in enc]).astype(np.int64)
        arrays[key + ".sum"] = np.array([i for e in summ for i in e], dtype=np.int64)
        arrays[key + ".sum_off"] = np.cumsum([0] + [len(e) for e in summ]).astype(np.int64)
        arrays[key + ".out"] = got.numpy()
        desc[key] = {"n": len(tl), "max_len": max_len}
import json  # noqa: E402

np.savez_compressed(os.path.join(HERE, "tokenise_cases.npz"), case=np.array(json.dumps(desc)), **arrays)
print("wrote tests/golden/tokenise_cases.npz:", json.dumps(desc))
