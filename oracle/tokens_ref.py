"""TEST INFRASTRUCTURE (never imported by the product path): the array-building half of the reference's `_tokenise`
(dataset_loaders/dataset_loaders.py:224-248) restated in plain Python.  The BPE encoder (`clip.simple_tokenizer`, un-vendored) and
the RAKE keyword summariser in front of it (`rake_nltk`, absent here) are host text processing and out of scope: this takes the
encoder's output -- one list of ids per text -- and builds the [n, max_len] int64 array the text tower reads.
PINNED (round 5): tests/golden/tokenise_cases.npz = outputs of the reference's own `_tokenise`, run unmodified under stand-in BPE / RAKE
objects by tests/golden/make_tokenise_golden.py; tests/test_token_packing.py holds this file and vtc_pack_tokens to them."""
import numpy as np

SOT, EOT = 49406, 49407


def pack_tokens(token_lists, max_len=77, sot=SOT, eot=EOT, summarise=None):
    """:228-231 `[sot] + encode(text) + [eot]`; :233 zeros [n, max_len]; :235-239 a sequence with len >= max_len is first
    replaced by `[sot] + encode(" ".join(rake phrases of text i)) + [eot]` -- `summarise(i)` returns that re-encoding's ids (the
    RAKE + BPE steps themselves are host text processing; None: no summariser, the ids stay); :240-243 if still >= max_len it
    becomes `tokens[: max_len - 1] + [eot]`; :244-247 otherwise the ids followed by zeros."""
    out = np.zeros((len(token_lists), max_len), dtype=np.int64)
    for i, enc in enumerate(token_lists):
        tokens = [sot] + [int(t) for t in enc] + [eot]
        if len(tokens) >= max_len:
            if summarise is not None:
                tokens = [sot] + [int(t) for t in summarise(i)] + [eot]
            if len(tokens) >= max_len:
                tokens = tokens[: max_len - 1] + [eot]
        out[i, : len(tokens)] = tokens
    return out
