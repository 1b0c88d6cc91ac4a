"""Window geometry of a Swin stage (behaviour of reference networks/utils/swin_utils.py:80-104).  window_partition / window_reverse /
compute_mask have no counterpart here: they are index arithmetic inside csrc/attention.hip."""


def get_window_size(x_size, window_size, shift_size=None):
    """The window an attention layer really uses on a token grid of extent `x_size`: along an axis whose grid does not exceed the configured
    window the window shrinks to the grid and that axis is not shifted (one window covers it, a cyclic shift would be a no-op with a mask).
    Returns the window, or (window, shift) when a shift is given - the call shapes of swin_transformer_block.py:103 and
    swin_transformer.py:233."""
    fits = [g > w for g, w in zip(x_size, window_size)]          # False: the axis is covered by a single, clamped window
    window = tuple(w if f else g for f, g, w in zip(fits, x_size, window_size))
    if shift_size is None:
        return window
    return window, tuple(s if f else 0 for f, s in zip(fits, shift_size))
