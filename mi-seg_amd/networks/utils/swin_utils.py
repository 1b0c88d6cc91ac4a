"""Window geometry helpers (reference networks/utils/swin_utils.py:80-104).  window_partition / window_reverse /
compute_mask have no counterpart here: they are index arithmetic inside csrc/attention.hip."""


def get_window_size(x_size, window_size, shift_size=None):
    use_window_size = list(window_size)
    if shift_size is not None:
        use_shift_size = list(shift_size)
    for i in range(len(x_size)):
        if x_size[i] <= window_size[i]:
            use_window_size[i] = x_size[i]
            if shift_size is not None:
                use_shift_size[i] = 0
    if shift_size is None:
        return tuple(use_window_size)
    return tuple(use_window_size), tuple(use_shift_size)
