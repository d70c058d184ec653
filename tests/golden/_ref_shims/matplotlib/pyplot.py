"""No-op pyplot (see __init__.py)."""


class _Axes:
    def __getattr__(self, name):
        return lambda *a, **k: None


class _Figure:
    def add_subplot(self, *a, **k):
        return _Axes()

    def __getattr__(self, name):
        return lambda *a, **k: None


def figure(*a, **k):
    return _Figure()


def __getattr__(name):
    return lambda *a, **k: None
