"""Parameter initialisers with Lasagne's semantics (reference use:
gcnmodel.py:270-274,348,359).  Lasagne's RNG is the global ``np.random`` -- which is why the
reference seeds it with ``np.random.seed(seed)`` at gcnmodel.py:336 -- so these draw from
``np.random`` too, in the same order, to keep seed-for-seed comparable initial weights.
[RECALL: lasagne/init.py is not in the reference tree; formulas restated from Lasagne 0.1/0.2.]"""
from __future__ import annotations

import numpy as np


def floatX(a):
    return np.asarray(a, dtype=np.float32)


class Initializer:
    def __call__(self, shape):
        return self.sample(shape)

    def sample(self, shape):
        raise NotImplementedError


class Uniform(Initializer):
    def __init__(self, range=0.01, std=None, mean=0.0):
        if std is not None:
            a = mean - np.sqrt(3) * std
            b = mean + np.sqrt(3) * std
        else:
            try:
                a, b = range
            except TypeError:
                a, b = -range, range
        self.range = (a, b)

    def sample(self, shape):
        return floatX(np.random.uniform(low=self.range[0], high=self.range[1], size=shape))


class Normal(Initializer):
    def __init__(self, std=0.01, mean=0.0):
        self.std, self.mean = std, mean

    def sample(self, shape):
        return floatX(np.random.normal(self.mean, self.std, size=shape))


class Glorot(Initializer):
    """std = gain * sqrt(2 / ((n_in + n_out) * receptive_field))."""

    def __init__(self, initializer, gain=1.0, c01b=False):
        if gain == 'relu':
            gain = np.sqrt(2)
        self.initializer, self.gain, self.c01b = initializer, gain, c01b

    def sample(self, shape):
        if len(shape) < 2:
            raise RuntimeError("This initializer only works with shapes of length >= 2")
        n1, n2 = shape[:2]
        receptive_field_size = np.prod(shape[2:])
        std = self.gain * np.sqrt(2.0 / ((n1 + n2) * receptive_field_size))
        return self.initializer(std=std).sample(shape)


class GlorotUniform(Glorot):
    def __init__(self, gain=1.0, c01b=False):
        super().__init__(Uniform, gain, c01b)


class GlorotNormal(Glorot):
    def __init__(self, gain=1.0, c01b=False):
        super().__init__(Normal, gain, c01b)


class Constant(Initializer):
    def __init__(self, val=0.0):
        self.val = val

    def sample(self, shape):
        return floatX(np.ones(shape) * self.val)


class Orthogonal(Initializer):
    """SVD of a gaussian matrix (Saxe et al.); reference gate weights gcnmodel.py:359."""

    def __init__(self, gain=1.0):
        if gain == 'relu':
            gain = np.sqrt(2)
        self.gain = gain

    def sample(self, shape):
        if len(shape) < 2:
            raise RuntimeError("Only shapes of length 2 or more are supported.")
        flat_shape = (shape[0], int(np.prod(shape[1:])))
        a = np.random.normal(0.0, 1.0, flat_shape)
        u, _, v = np.linalg.svd(a, full_matrices=False)
        q = u if u.shape == flat_shape else v
        q = q.reshape(shape)
        return floatX(self.gain * q)
