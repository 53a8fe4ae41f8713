"""oracle/ -- CPU restatement of the VTC retrieval forward/eval hot path.

TEST INFRASTRUCTURE ONLY.  Nothing under ``vtc_amd/`` (the product) may import
this package.  The only legitimate importers are ``tests/``,
``__graft_entry__.smoke()`` and the ``cpu_baseline`` leg of ``bench.py`` --
always as the checker, never as the thing that is measured or shipped.

What it is: a plain PyTorch fp32 (optionally fp64) functional restatement of
the reference algorithm for the path named in BASELINE.json:north_star.  Every
function cites the reference ``file:line`` it follows (paths are relative to
the read-only reference checkout).  The path is floating point, hence torch
rather than C; the integer parts (token argmax, empty-comment mask, Recall@K
rank bookkeeping) are exact.

Pinning status (see DESIGN.md "Oracle"):
  * first-party reference arithmetic (model/timesformer_clip_alt.py,
    model/timesformer_clip.py, model/model.py CAM glue + wrappers,
    model/loss.py) -- PINNED: ``tests/golden/make_golden.py`` imports the
    reference's own Python unmodified (with a tests-only stand-in for the
    un-vendored ``clip`` package) and commits input/output vectors under
    ``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement
    against them.
  * third-party ``openai/CLIP`` arithmetic (environment.yml:30, un-pinned git
    HEAD, source NOT under the reference tree) -- restated from the public
    architecture and cross-checked against HuggingFace ``transformers`` CLIP
    (an independent implementation) with mapped weights; the reference's own
    tests for that boundary need a weights download: parity unpinned vs the
    real upstream weights.
  * ``model/metric.py`` RecallAtK (faiss-gpu, not importable, nothing in the
    reference's tests pins it): restated literally; parity unpinned by the
    reference, pinned only by constructed rank cases.
"""

from .arch import ClipArch, VIT_B32, TINY  # noqa: F401
