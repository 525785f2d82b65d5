"""Minimal ``dnnlib`` surface the operator modules need (reference dnnlib/util.py:40-54)."""


class EasyDict(dict):
    """dict whose items are also attributes."""

    def __getattr__(self, name):
        try:
            return self[name]
        except KeyError:
            raise AttributeError(name)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]
