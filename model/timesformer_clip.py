"""Drop-in for the reference's ``model.timesformer_clip`` (model/timesformer_clip.py): the older TimeSformer variant."""
from vtc_amd.host.clip_arch import VisualTransformerV1 as VisualTransformer  # noqa: F401
from vtc_amd.host.clip_arch import make_timesformer_clip_vit  # noqa: F401
