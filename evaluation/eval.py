"""``python evaluation/eval.py -c configs/<cfg>.jsonc`` -- the reference's eval entry point, served by
the MI355X implementation (vtc_amd/host/eval.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtc_amd.host.eval import cli, main  # noqa: E402,F401

if __name__ == "__main__":
    cli(sys.argv[1:])
