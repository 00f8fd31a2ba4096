def imread(*a, **k):
    raise RuntimeError("imageio stub: no file I/O in the oracle")
