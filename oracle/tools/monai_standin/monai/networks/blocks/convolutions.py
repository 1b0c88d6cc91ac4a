def __getattr__(name):  # lazy to survive the circular import with the reference's convolutions.py
    import networks.blocks.convolutions as ref
    return getattr(ref, name)
