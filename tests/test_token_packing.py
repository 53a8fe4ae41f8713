"""Token packing: the array-building half of the reference's `_tokenise`
(dataset_loaders/dataset_loaders.py:224-248).  CPU: the oracle restatement against the outputs of the reference's OWN `_tokenise`
(tests/golden/tokenise_cases.npz, made by tests/golden/make_tokenise_golden.py) and against hand-written known answers of that code's
rules; GPU: vtc_pack_tokens (through the C ABI) against the same fixture and against the oracle on ragged batches, empty /
exactly-full / over-long texts."""
import json
import os
import numpy as np
import pytest
import torch

from oracle import tokens_ref as TR


def _tokenise_cases():
    z = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "tokenise_cases.npz"), allow_pickle=False)
    for key, c in json.loads(str(z["case"])).items():
        eo, so = z[key + ".enc_off"], z[key + ".sum_off"]
        enc = [z[key + ".enc"][eo[i]:eo[i + 1]].tolist() for i in range(c["n"])]
        summ = [z[key + ".sum"][so[i]:so[i + 1]].tolist() for i in range(c["n"])]
        yield key, c["max_len"], enc, summ, z[key + ".out"]


def test_oracle_equals_the_references_own_tokenise():
    """The fixture holds what the reference's `_tokenise` (run unmodified, stand-in BPE / RAKE) returned: SOT / EOT framing, the
    `len(tokens) >= max_len` test, the re-encoded summary, the hard cut `tokens[:max_len - 1] + [eot]`, zero padding, a bare string."""
    n = 0
    for key, max_len, enc, summ, want in _tokenise_cases():
        got = TR.pack_tokens(enc, max_len=max_len, summarise=lambda i: summ[i])
        assert got.dtype == np.int64 and np.array_equal(got, want), key
        n += 1
    assert n >= 10


@pytest.mark.gpu
def test_pack_tokens_kernel_equals_the_references_own_tokenise():
    from vtc_amd.host import datasets as DS
    for key, max_len, enc, summ, want in _tokenise_cases():
        got = DS.pack_token_lists(enc, max_len=max_len, device="cuda", summarise=lambda i: summ[i])
        assert got.dtype == torch.int64 and np.array_equal(got.cpu().numpy(), want), key


def test_oracle_packing_rules():
    # short: [sot] + ids + [eot] + zeros (:244-247)
    out = TR.pack_tokens([[5, 6, 7], []], max_len=8)
    assert out.tolist() == [[TR.SOT, 5, 6, 7, TR.EOT, 0, 0, 0], [TR.SOT, TR.EOT, 0, 0, 0, 0, 0, 0]]
    # len(tokens) == max_len takes the truncation branch and comes out unchanged; longer keeps max_len - 1 ids + eot (:236-243)
    full = list(range(10, 16))                # 6 ids + sot + eot = 8
    over = list(range(10, 30))
    out = TR.pack_tokens([full, over], max_len=8)
    assert out[0].tolist() == [TR.SOT] + full + [TR.EOT]
    assert out[1].tolist() == [TR.SOT] + over[:6] + [TR.EOT]
    # the text tower finds the EOT with argmax: it is the largest id and the first maximum of every row
    assert (out.argmax(-1) == np.array([7, 7])).all()
    assert TR.pack_tokens([[1]] * 3).shape == (3, 77) and TR.pack_tokens([[1]]).dtype == np.int64
    # :235-239 an over-long text is first replaced by the re-encoded keyword summary (hook), and truncated only if still too long
    summ = {1: [7, 8], 2: list(range(40, 60))}
    out = TR.pack_tokens([[5], over, over], max_len=8, summarise=lambda i: summ[i])
    assert out[0].tolist() == [TR.SOT, 5, TR.EOT, 0, 0, 0, 0, 0]
    assert out[1].tolist() == [TR.SOT, 7, 8, TR.EOT, 0, 0, 0, 0]
    assert out[2].tolist() == [TR.SOT] + summ[2][:6] + [TR.EOT]


def test_pack_token_lists_flags_overlong_texts_without_a_summariser():
    """silently truncating the original ids differs from the reference for texts that reach max_len."""
    from vtc_amd.host import datasets as DS
    with pytest.raises(ValueError, match="summaris"):
        DS.pack_token_lists([[1, 2], list(range(1, 100))], max_len=77, device="cpu", strict=True)


@pytest.mark.gpu
def test_pack_tokens_kernel_matches_the_oracle():
    from vtc_amd import ops
    from vtc_amd.host import datasets as DS
    rng = np.random.default_rng(11)
    for ctx in (77, 24, 8):
        lens = [0, 1, ctx - 3, ctx - 2, ctx - 1, ctx, 3 * ctx] + [int(v) for v in rng.integers(0, ctx + 5, size=200)]
        lists = [[int(v) for v in rng.integers(1, 49405, size=n)] for n in lens]
        want = TR.pack_tokens(lists, max_len=ctx)
        with pytest.warns(UserWarning, match="summaris"):
            got = DS.pack_token_lists(lists, max_len=ctx, device="cuda")
        summ = {i: [int(v) for v in rng.integers(1, 49405, size=int(rng.integers(0, ctx + 3)))] for i, t in enumerate(lists) if len(t) + 2 >= ctx}
        got_s = DS.pack_token_lists(lists, max_len=ctx, device="cuda", summarise=lambda i: summ[i])
        assert np.array_equal(got_s.cpu().numpy(), TR.pack_tokens(lists, max_len=ctx, summarise=lambda i: summ[i]))
        assert got.dtype == torch.int64 and got.is_cuda and np.array_equal(got.cpu().numpy(), want)
        # device-resident inputs through the op itself
        flat = torch.tensor([v for t in lists for v in t], dtype=torch.int32, device="cuda")
        offs = torch.tensor(np.concatenate([[0], np.cumsum(lens)]), dtype=torch.int32, device="cuda")
        assert np.array_equal(ops.pack_tokens(flat, offs, ctx).cpu().numpy(), want)
    # an all-empty batch (no token at all)
    assert np.array_equal(DS.pack_token_lists([[], []], device="cuda").cpu().numpy(), TR.pack_tokens([[], []]))
    # and the packed ids run through the text tower like any other (EOT position = argmax)
    assert (got.argmax(-1).cpu().numpy() == want.argmax(-1)).all()
