"""Drop-in for the reference's ``model.timesformer_clip_alt`` (model/timesformer_clip_alt.py): the tower class and its
factory, forward on libvtc_hip.so."""
from vtc_amd.host.clip_arch import VisualTransformer, make_timesformer_clip_vit_alt  # noqa: F401
