"""Online mean / population-variance accumulator (``SuperresolutionNetwork/utils/mv.py:1-29``,
Welford update; ``var`` divides by n, matching ``numpy.var``)."""


class MeanVariance:
    def __init__(self):
        self._n = 0
        self._mean = 0.0
        self._m2 = 0.0

    def append(self, x):
        self._n += 1
        d = x - self._mean
        self._mean += d / self._n
        self._m2 += d * (x - self._mean)

    def count(self):
        return self._n

    def mean(self):
        return self._mean

    def var(self):
        return self._m2 / self._n if self._n > 0 else 0.0
