"""``python train.py -c configs/pretrained_clip_comments_attn_frozen.jsonc`` -- the reference's training entry point,
served by the MI355X implementation for the adapter-only slice (vtc_amd/host/train.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from vtc_amd.host.train import cli, main  # noqa: E402,F401

if __name__ == "__main__":
    cli(sys.argv[1:])
