from .synthetic import synthetic_batch  # noqa
from .samplers import InferenceSampler, RepeatFactorTrainingSampler, TrainingSampler  # noqa
