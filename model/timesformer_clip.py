from vtc_amd.host.clip_arch import make_timesformer_clip_vit  # noqa: F401
