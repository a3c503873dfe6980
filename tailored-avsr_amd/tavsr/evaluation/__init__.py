from .bootstrap_wer import compute_bootstrap_wer, error_rate  # noqa: F401
