from vtc_amd.host.loss import clip_loss  # noqa: F401
