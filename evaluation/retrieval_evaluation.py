"""``from evaluation.retrieval_evaluation import retrieval_evaluation`` / ``python evaluation/retrieval_evaluation.py -c MSRVTT_videos
-m clip_timesformer_finaltf`` -- the reference's video-benchmark evaluation (evaluation/retrieval_evaluation.py, called by
trainer/trainer.py:159-173), served by the MI355X implementation (vtc_amd/host/retrieval_evaluation.py)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from vtc_amd.host.retrieval_evaluation import (  # noqa: E402,F401
    cli, compute_recall, image_models, load_model, models_needing_comments, retrieval_evaluation, video_models)

if __name__ == "__main__":
    cli(sys.argv[1:])
