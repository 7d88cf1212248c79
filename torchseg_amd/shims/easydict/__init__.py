"""Minimal stand-in for the `easydict` package the reference's config.py imports
(model/*/config.py:11): a dict whose keys are attributes, recursively."""


class EasyDict(dict):
    def __init__(self, d=None, **kwargs):
        super().__init__()
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v

    @staticmethod
    def _wrap(v):
        if isinstance(v, dict) and not isinstance(v, EasyDict):
            return EasyDict(v)
        if isinstance(v, (list, tuple)):
            return type(v)(EasyDict._wrap(x) for x in v)
        return v

    def __setitem__(self, k, v):
        super().__setitem__(k, self._wrap(v))

    def __setattr__(self, k, v):
        self[k] = v

    def __getattr__(self, k):
        try:
            return self[k]
        except KeyError:
            raise AttributeError(k)

    def __delattr__(self, k):
        del self[k]

    def update(self, d=None, **kwargs):
        for k, v in dict(d or {}, **kwargs).items():
            self[k] = v
