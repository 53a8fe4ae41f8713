"""TEST INFRASTRUCTURE (never imported by the product path): the array-building half of the reference's `_tokenise`
(dataset_loaders/dataset_loaders.py:224-248) restated in plain Python.  The BPE encoder (`clip.simple_tokenizer`, un-vendored) and
the RAKE keyword summariser in front of it (`rake_nltk`, absent here) are host text processing and out of scope: this takes the
encoder's output -- one list of ids per text -- and builds the [n, max_len] int64 array the text tower reads."""
import numpy as np

SOT, EOT = 49406, 49407


def pack_tokens(token_lists, max_len=77, sot=SOT, eot=EOT):
    """:228-231 `[sot] + encode(text) + [eot]`; :233 zeros [n, max_len]; :236-243 a sequence with len >= max_len (after the
    summarisation attempt, which this restatement does not model) becomes `tokens[: max_len - 1] + [eot]`; :244-247 otherwise
    the ids followed by zeros."""
    out = np.zeros((len(token_lists), max_len), dtype=np.int64)
    for i, enc in enumerate(token_lists):
        tokens = [sot] + [int(t) for t in enc] + [eot]
        if len(tokens) >= max_len:
            tokens = tokens[: max_len - 1] + [eot]
        out[i, : len(tokens)] = tokens
    return out
