"""Stand-in for `addict.Dict` (attribute-access dict with recursive conversion of
nested dicts).  Used only by oracle/make_golden.py; see ../README.md."""


class Dict(dict):
    def __init__(self, *args, **kwargs):
        super().__init__()
        for arg in args:
            if not arg:
                continue
            if isinstance(arg, dict):
                for k, v in arg.items():
                    self[k] = self._hook(v)
            elif isinstance(arg, tuple) and not isinstance(arg[0], tuple):
                self[arg[0]] = self._hook(arg[1])
            else:
                for k, v in iter(arg):
                    self[k] = self._hook(v)
        for k, v in kwargs.items():
            self[k] = self._hook(v)

    @classmethod
    def _hook(cls, item):
        if isinstance(item, dict):
            return cls(item)
        if isinstance(item, (list, tuple)):
            return type(item)(cls._hook(e) for e in item)
        return item

    def __getattr__(self, item):
        try:
            return self[item]
        except KeyError:
            raise AttributeError(item)

    def __setattr__(self, name, value):
        self[name] = value

    def __delattr__(self, name):
        del self[name]
