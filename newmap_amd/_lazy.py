"""numpy on first use.  `newmap search` through the native driver needs no array on the Python side, and importing
numpy is a sixth of that process's wall time on a 3 Gbp genome."""


class _LazyNumpy:
    def __getattr__(self, name):
        import numpy
        return getattr(numpy, name)


np = _LazyNumpy()
