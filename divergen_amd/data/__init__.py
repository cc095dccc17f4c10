from .synthetic import synthetic_batch  # noqa
