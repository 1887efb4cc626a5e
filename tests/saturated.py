"""Re-export of the confident-head fixture (oracle/confident_head.py) under the name the tests import."""
from oracle.confident_head import metric_counts, saturate_head  # noqa: F401
