import numpy as np


def same_padding(kernel_size, dilation=1):
    k = np.atleast_1d(kernel_size)
    d = np.atleast_1d(dilation)
    if np.any((k - 1) * d % 2 == 1):
        raise NotImplementedError("same padding not available for this kernel/dilation")
    p = tuple(int(v) for v in (k - 1) / 2 * d)
    return p if len(p) > 1 else p[0]


def stride_minus_kernel_padding(kernel_size, stride):
    k = np.atleast_1d(kernel_size)
    s = np.atleast_1d(stride)
    p = tuple(int(v) for v in (s - k))
    return p if len(p) > 1 else p[0]
