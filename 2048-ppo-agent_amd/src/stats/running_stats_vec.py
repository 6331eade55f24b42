"""Streaming mean/variance of row-vectors (API of the reference src/stats/running_stats_vec.py:4-99)."""
import numpy as np


class RunningStatsVec:
    """Chan et al. pairwise merge of (count, mean, population variance), one row per feature.

    ``push(x)`` takes ``x[num_features, num_samples]``; rows never seen before start at zero count.
    """

    def __init__(self):
        self.clear()

    def clear(self):
        self.num_samples = np.zeros((1, 1), dtype=np.int64)
        self._mean = np.zeros((1, 1), dtype=np.float64)
        self._variance = np.zeros((1, 1), dtype=np.float64)

    def _ensure_rows(self, rows: int):
        have = self.num_samples.shape[0]
        if rows > have:
            pad = ((0, rows - have), (0, 0))
            self.num_samples = np.pad(self.num_samples, pad)
            self._mean = np.pad(self._mean, pad)
            self._variance = np.pad(self._variance, pad)

    def push(self, x: np.ndarray):
        x = np.asarray(x)
        if x.ndim != 2:
            raise ValueError("Input array should have 2 dimensions.")
        rows, m = x.shape
        self._ensure_rows(rows)
        n = self.num_samples[:rows]
        mean_b = x.mean(axis=1, keepdims=True)
        var_b = x.var(axis=1, keepdims=True)
        total = n + m
        delta = mean_b - self._mean[:rows]
        new_mean = (self._mean[:rows] * n + mean_b * m) / total
        new_var = (var_b * m + self._variance[:rows] * n + delta ** 2.0 * (n * m) / total) / total
        self._mean[:rows] = new_mean
        self._variance[:rows] = new_var
        self.num_samples[:rows] = total

    @property
    def mean(self):
        return self._mean if self.num_samples.sum() else 0.0

    @property
    def variance(self):
        return self._variance if self.num_samples.sum() else 0.0

    @property
    def std(self):
        return np.sqrt(self._variance) if self.num_samples.sum() else 0.0
