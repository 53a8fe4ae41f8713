from vtc_amd.host.metric import BaseMetric, RecallAtK  # noqa: F401
