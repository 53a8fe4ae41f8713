"""Drop-in shim: ``import model.model as module_arch`` etc. resolve to the MI355X implementation
(vtc_amd.host), mirroring the reference's ``model`` package for the hot path only."""
from . import loss, metric, model, timesformer_clip, timesformer_clip_alt  # noqa: F401
from .loss import clip_loss  # noqa: F401
from .metric import RecallAtK  # noqa: F401
from .model import (PretrainedCLIP, PretrainedCLIP_finaltf, PretrainedCLIP_TimeSformer,  # noqa: F401
                    PretrainedCLIP_TimeSformer_finaltf, PretrainedCLIPBase)
from .timesformer_clip_alt import make_timesformer_clip_vit_alt  # noqa: F401
from .timesformer_clip import make_timesformer_clip_vit  # noqa: F401
